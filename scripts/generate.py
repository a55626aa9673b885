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


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--input_dir", type=str, required=True, help="either a directory containing subdirectories "
                                                                     "train, val, test, etc, or a directory containing "
                                                                     "the tfrecords")
    parser.add_argument("--results_dir", type=str, default='results', help="ignored if output_gif_dir is specified")
    parser.add_argument("--results_gif_dir", type=str, help="default is results_dir. ignored if output_gif_dir is specified")
    parser.add_argument("--results_png_dir", type=str, help="default is results_dir. ignored if output_png_dir is specified")
    parser.add_argument("--output_gif_dir", help="output directory where samples are saved as gifs. default is "
                                                 "results_gif_dir/model_fname")
    parser.add_argument("--output_png_dir", help="output directory where samples are saved as pngs. default is "
                                                 "results_png_dir/model_fname")
    parser.add_argument("--checkpoint", help="directory with checkpoint or checkpoint name (e.g. checkpoint_dir/model-200000)")

    parser.add_argument("--mode", type=str, choices=['val', 'test'], default='val', help='mode for dataset, val or test.')

    parser.add_argument("--dataset", type=str, help="dataset class name")
    parser.add_argument("--dataset_hparams", type=str, help="a string of comma separated list of dataset hyperparameters")
    parser.add_argument("--model", type=str, help="model class name")
    parser.add_argument("--model_hparams", type=str, help="a string of comma separated list of model hyperparameters")

    parser.add_argument("--batch_size", type=int, default=8, help="number of samples in batch")
    parser.add_argument("--num_samples", type=int, help="number of samples in total (all of them by default)")
    parser.add_argument("--num_epochs", type=int, default=1)

    parser.add_argument("--num_stochastic_samples", type=int, default=5)
    parser.add_argument("--gif_length", type=int, help="default is sequence_length")
    parser.add_argument("--fps", type=int, default=4)

    parser.add_argument("--gpu_mem_frac", type=float, default=0, help="fraction of gpu memory to use")
    parser.add_argument("--seed", type=int, default=7)
    parser.add_argument("--synthetic_shape", type=str, default='64,64,3', help="H,W,C of --dataset synthetic (not in the reference)")
    return parser


def resolve_options(args):
    """generate.py:57-86."""
    args.results_gif_dir = args.results_gif_dir or args.results_dir
    args.results_png_dir = args.results_png_dir or args.results_dir
    dataset_hparams_dict, model_hparams_dict = {}, {}
    if args.checkpoint:
        checkpoint_dir = os.path.normpath(args.checkpoint)
        if not os.path.isdir(args.checkpoint):
            checkpoint_dir, _ = os.path.split(checkpoint_dir)
        if not os.path.exists(checkpoint_dir):
            raise FileNotFoundError(errno.ENOENT, os.strerror(errno.ENOENT), checkpoint_dir)
        with open(os.path.join(checkpoint_dir, "options.json")) as f:
            print("loading options from checkpoint %s" % args.checkpoint)
            options = json.loads(f.read())
            args.dataset = args.dataset or options['dataset']
            args.model = args.model or options['model']
        try:
            with open(os.path.join(checkpoint_dir, "dataset_hparams.json")) as f:
                dataset_hparams_dict = json.loads(f.read())
        except FileNotFoundError:
            print("dataset_hparams.json was not loaded because it does not exist")
        try:
            with open(os.path.join(checkpoint_dir, "model_hparams.json")) as f:
                model_hparams_dict = json.loads(f.read())
        except FileNotFoundError:
            print("model_hparams.json was not loaded because it does not exist")
        args.output_gif_dir = args.output_gif_dir or os.path.join(args.results_gif_dir, os.path.split(checkpoint_dir)[1])
        args.output_png_dir = args.output_png_dir or os.path.join(args.results_png_dir, os.path.split(checkpoint_dir)[1])
    else:
        if not args.dataset:
            raise ValueError('dataset is required when checkpoint is not specified')
        if not args.model:
            raise ValueError('model is required when checkpoint is not specified')
        args.output_gif_dir = args.output_gif_dir or os.path.join(args.results_gif_dir, 'model.%s' % args.model)
        args.output_png_dir = args.output_png_dir or os.path.join(args.results_png_dir, 'model.%s' % args.model)
    return dataset_hparams_dict, model_hparams_dict


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

    for output_dir in (args.output_gif_dir, args.output_png_dir):
        if not os.path.exists(output_dir):
            os.makedirs(output_dir)
        with open(os.path.join(output_dir, "options.json"), "w") as f:
            f.write(json.dumps(vars(args), sort_keys=True, indent=4))
        with open(os.path.join(output_dir, "dataset_hparams.json"), "w") as f:
            f.write(json.dumps(dataset.hparams.values(), sort_keys=True, indent=4))
        with open(os.path.join(output_dir, "model_hparams.json"), "w") as f:
            f.write(json.dumps(model.hparams.values(), sort_keys=True, indent=4))
    if args.checkpoint:
        model.restore(args.checkpoint)

    sample_ind = 0
    while inputs is not None:
        if args.num_samples and sample_ind >= args.num_samples:
            break
        print("evaluation samples from %d to %d" % (sample_ind, sample_ind + args.batch_size))
        context = (inputs['images'] * 255.0).to(torch.uint8).cpu().numpy()                 # [B, T, H, W, C]
        for stochastic_sample_ind in range(args.num_stochastic_samples):
            gen_images = model.generate(inputs)['gen_images']                               # [B, T-1, H, W, C]
            gen_images = (gen_images[:, -future_length:] * 255.0).to(torch.uint8).cpu().numpy()      # only keep the future frames
            for i, gen_images_ in enumerate(gen_images):
                frames = list(context[i][:context_frames]) + list(gen_images_)
                if args.gif_length:
                    frames = frames[:args.gif_length]
                pattern = 'gen_image_%%05d_%%02d_%%0%dd.png' % max(2, len(str(len(gen_images_) - 1)))
                for t, gen_image in enumerate(gen_images_):
                    write_png(os.path.join(args.output_png_dir, pattern % (sample_ind + i, stochastic_sample_ind, t)), gen_image)
        sample_ind += args.batch_size
        try:
            inputs = next(batches)
        except StopIteration:
            inputs = None


if __name__ == '__main__':
    main()
