#!/usr/bin/env python
"""Session-less sampling runner with the reference's command line (scripts/generate.py of alexlee-gk/video_prediction).

Preserved from the reference (file:line under /root/reference/scripts/generate.py): every flag and default of :20-49, options /
hparams read back from the checkpoint directory (:57-79), output directory naming (:77-86), the three side JSONs written next to
the results (:140-148), the sampling loop -- num_stochastic_samples prior samples per batch, only the future frames kept
(:154-170) -- and the file names `gen_image_%05d_%02d_%0Nd.png` (:183-189).  PNGs are written by a ~30-line zlib encoder (no
cv2 / PIL here); GIFs (ffmpeg in the reference, :176-181) are out of scope, the per-frame PNGs carry the same pixels.
"""
from __future__ import absolute_import, division, print_function

import argparse
import errno
import json
import os
import random
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# (flag, keyword arguments) in the reference's order, generate.py:20-49; --synthetic_shape is this repository's addition
_FLAGS = (
    ('input_dir', dict(type=str, required=True,
                       help="either a directory containing subdirectories train, val, test, etc, or a directory containing the tfrecords")),
    ('results_dir', dict(type=str, default='results', help="ignored if output_gif_dir is specified")),
    ('results_gif_dir', dict(type=str, help="default is results_dir. ignored if output_gif_dir is specified")),
    ('results_png_dir', dict(type=str, help="default is results_dir. ignored if output_png_dir is specified")),
    ('output_gif_dir', dict(help="output directory where samples are saved as gifs. default is results_gif_dir/model_fname")),
    ('output_png_dir', dict(help="output directory where samples are saved as pngs. default is results_png_dir/model_fname")),
    ('checkpoint', dict(help="directory with checkpoint or checkpoint name (e.g. checkpoint_dir/model-200000)")),
    ('mode', dict(type=str, choices=['val', 'test'], default='val', help='mode for dataset, val or test.')),
    ('dataset', dict(type=str, help="dataset class name")),
    ('dataset_hparams', dict(type=str, help="a string of comma separated list of dataset hyperparameters")),
    ('model', dict(type=str, help="model class name")),
    ('model_hparams', dict(type=str, help="a string of comma separated list of model hyperparameters")),
    ('batch_size', dict(type=int, default=8, help="number of samples in batch")),
    ('num_samples', dict(type=int, help="number of samples in total (all of them by default)")),
    ('num_epochs', dict(type=int, default=1)),
    ('num_stochastic_samples', dict(type=int, default=5)),
    ('gif_length', dict(type=int, help="default is sequence_length")),
    ('fps', dict(type=int, default=4)),
    ('gpu_mem_frac', dict(type=float, default=0, help="fraction of gpu memory to use")),
    ('seed', dict(type=int, default=7)),
    ('synthetic_shape', dict(type=str, default='64,64,3', help="H,W,C of --dataset synthetic (not in the reference)")),
)


def build_parser():
    parser = argparse.ArgumentParser()
    for name, kw in _FLAGS:
        parser.add_argument('--' + name, **kw)
    return parser


def _side_json(directory, name):
    """One of the side files train.py leaves beside a checkpoint; a missing hparams file is reported, not fatal (generate.py:68-76)."""
    try:
        with open(os.path.join(directory, name)) as f:
            return json.load(f)
    except FileNotFoundError:
        print("%s was not loaded because it does not exist" % name)
        return {}


def resolve_options(args):
    """generate.py:57-86: result directories, and dataset / model / hparams read back from the checkpoint's directory."""
    for kind in ('gif', 'png'):
        if not getattr(args, 'results_%s_dir' % kind):
            setattr(args, 'results_%s_dir' % kind, args.results_dir)
    hparams = {'dataset': {}, 'model': {}}
    if args.checkpoint:
        ckpt_dir = os.path.normpath(args.checkpoint)
        if not os.path.isdir(args.checkpoint):                  # a checkpoint PREFIX (dir/model-200000) names its directory
            ckpt_dir = os.path.dirname(ckpt_dir)
        if not os.path.exists(ckpt_dir):
            raise FileNotFoundError(errno.ENOENT, os.strerror(errno.ENOENT), ckpt_dir)
        print("loading options from checkpoint %s" % args.checkpoint)
        with open(os.path.join(ckpt_dir, "options.json")) as f:      # this one is required
            saved = json.load(f)
        for key in ('dataset', 'model'):
            if not getattr(args, key):
                setattr(args, key, saved[key])
            hparams[key] = _side_json(ckpt_dir, key + '_hparams.json')
        leaf = os.path.basename(ckpt_dir)
    else:
        for key in ('dataset', 'model'):
            if not getattr(args, key):
                raise ValueError('%s is required when checkpoint is not specified' % key)
        leaf = 'model.%s' % args.model
    for kind in ('gif', 'png'):
        if not getattr(args, 'output_%s_dir' % kind):
            setattr(args, 'output_%s_dir' % kind, os.path.join(getattr(args, 'results_%s_dir' % kind), leaf))
    return hparams['dataset'], hparams['model']


def write_png(path, image):
    """image uint8 [H, W, 1 | 3] -> 8-bit grayscale / RGB PNG (zlib + CRC from the standard library)."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w, c = image.shape
    if c not in (1, 3):
        raise ValueError('PNG writer handles 1 or 3 channels, got %d' % c)

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    raw = b''.join(b'\x00' + image[y].tobytes() for y in range(h))            # filter type 0 per scanline
    png = b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0 if c == 1 else 2, 0, 0, 0)) + \
        chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b'')
    with open(path, 'wb') as f:
        f.write(png)


def write_gif(path, frames, fps):
    """frames uint8 [T, H, W, 1 | 3] -> an animated GIF at `fps`, looping (the reference's save_gif through moviepy, utils/ffmpeg_gif.py /
    generate.py:170-176; here Pillow's GIF encoder: adaptive palette per frame)."""
    from PIL import Image
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    imgs = [Image.fromarray(f[..., 0], 'L') if f.shape[-1] == 1 else Image.fromarray(f, 'RGB') for f in frames]
    imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=max(1, int(round(1000.0 / max(fps, 1)))), loop=0)


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    if args.seed is not None:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
        random.seed(args.seed)
    dataset_hparams_dict, model_hparams_dict = resolve_options(args)
    print('----------------------------------- Options ------------------------------------')
    for k, v in args._get_kwargs():
        print(k, "=", v)
    print('------------------------------------- End --------------------------------------')
    if not torch.cuda.is_available():
        raise SystemExit('scripts/generate.py needs an MI355X: the SAVP hot path has no CPU fallback')
    device = 'cuda:0'

    from scripts.train import get_dataset_class
    from video_prediction_amd import models
    VideoDataset = get_dataset_class(args.dataset, args.synthetic_shape)
    dataset = VideoDataset(args.input_dir, mode=args.mode, num_epochs=args.num_epochs, seed=args.seed,
                           hparams_dict=dataset_hparams_dict, hparams=args.dataset_hparams)
    VideoPredictionModel = models.get_model_class(args.model)
    hparams_dict = dict(model_hparams_dict)
    hparams_dict.update({'context_frames': dataset.hparams.context_frames, 'sequence_length': dataset.hparams.sequence_length,
                         'repeat': dataset.hparams.time_shift})
    model = VideoPredictionModel(mode=args.mode if args.mode == 'train' else 'test', hparams_dict=hparams_dict, hparams=args.model_hparams)
    sequence_length = model.hparams.sequence_length
    context_frames = model.hparams.context_frames
    future_length = sequence_length - context_frames

    if args.num_samples:
        if args.num_samples > dataset.num_examples_per_epoch():
            raise ValueError('num_samples cannot be larger than the dataset')
        num_examples_per_epoch = args.num_samples
    else:
        num_examples_per_epoch = dataset.num_examples_per_epoch()
    if num_examples_per_epoch % args.batch_size != 0:
        raise ValueError('batch_size should evenly divide the dataset size %d' % num_examples_per_epoch)

    batches = iter(dataset.make_batch(args.batch_size, device=device))
    inputs = next(batches)
    model.build_graph(inputs, device=device)

    side = (("options.json", vars(args)), ("dataset_hparams.json", dataset.hparams.values()), ("model_hparams.json", model.hparams.values()))
    for output_dir in (args.output_gif_dir, args.output_png_dir):                             # generate.py:140-148
        os.makedirs(output_dir, exist_ok=True)
        for fname, content in side:
            with open(os.path.join(output_dir, fname), "w") as f:
                f.write(json.dumps(content, sort_keys=True, indent=4))
    if args.checkpoint:
        model.restore(args.checkpoint)

    def to_uint8(x):
        return (x * 255.0).to(torch.uint8).cpu().numpy()

    first = 0                                                    # index of the batch's first sample (generate.py:154-189)
    while inputs is not None and not (args.num_samples and first >= args.num_samples):
        print("evaluation samples from %d to %d" % (first, first + args.batch_size))
        context = to_uint8(inputs['images'][:, :context_frames])                                  # [B, context, H, W, C]
        for draw in range(args.num_stochastic_samples):
            future = to_uint8(model.generate(inputs)['gen_images'][:, -future_length:])      # [B, T - context, H, W, C]: the future frames only
            digits = max(2, len(str(future.shape[1] - 1)))
            for b, clip in enumerate(future):
                # generate.py:170-176: context frames + generated frames as one GIF, truncated to --gif_length, at --fps
                frames = np.concatenate([context[b], clip], axis=0)
                if args.gif_length:
                    frames = frames[:args.gif_length]
                write_gif(os.path.join(args.output_gif_dir, 'gen_image_%05d_%02d.gif' % (first + b, draw)), frames, args.fps)
                for t, frame in enumerate(clip):
                    name = 'gen_image_%05d_%02d_%0*d.png' % (first + b, draw, digits, t)
                    write_png(os.path.join(args.output_png_dir, name), frame)
        first += args.batch_size
        inputs = next(batches, None)


if __name__ == '__main__':
    main()
