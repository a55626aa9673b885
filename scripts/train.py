#!/usr/bin/env python
"""Session-less training runner with the reference's command line (scripts/train.py of alexlee-gk/video_prediction).

Preserved from the reference (file:line under /root/reference/scripts/train.py):
  * every flag, default and help string of :30-61, the output-directory naming from --model / --model_hparams (:68-83),
    --resume / --checkpoint with options.json + dataset_hparams.json + model_hparams.json read back from the checkpoint
    directory (:85-118) and written to the output directory (:193-200);
  * dataset / model construction: the dataset's context_frames / sequence_length / time_shift override the model hparams
    (:151-160), batch size from the model hparams (:162);
  * the step loop: steps run from -1 (log without training) to max_steps - start_step, timing skips steps -1 and 0 (:241-245),
    the progress block every --progress_freq steps (:322-345, same lines: "progress  global step", "image/sec", d_loss / g_loss and
    their terms, learning_rate), checkpoints every --save_freq steps as <output_dir>/model-<global_step> (:347-350, max_to_keep 2).
What TensorFlow did implicitly is explicit here: one `model.train_step(inputs)` is one `sess.run(model.train_op)`; summaries are
JSON lines in <output_dir>/summaries.jsonl (scalars + best-of-N eval metrics) instead of TensorBoard event files; image / GIF
summaries are not produced.  Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N` (one process per GPU,
RCCL gradient exchange = --aggregate_nccl 1 of the reference; the per-GPU batch is batch_size / N like tf.split, base_model.py:523-527).
`--dataset synthetic` (not in the reference) feeds seeded uniform video of --synthetic_shape without record files.
"""
from __future__ import absolute_import, division, print_function

import argparse
import errno
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--input_dir", type=str, required=True, help="either a directory containing subdirectories "
                                                                     "train, val, test, etc, or a directory containing "
                                                                     "the tfrecords")
    parser.add_argument("--val_input_dir", type=str, help="directories containing the tfrecords. default: input_dir")
    parser.add_argument("--logs_dir", default='logs', help="ignored if output_dir is specified")
    parser.add_argument("--output_dir", help="output directory where json files, summary, model, gifs, etc are saved. "
                                             "default is logs_dir/model_fname, where model_fname consists of "
                                             "information from model and model_hparams")
    parser.add_argument("--output_dir_postfix", default="")
    parser.add_argument("--checkpoint", help="directory with checkpoint or checkpoint name (e.g. checkpoint_dir/model-200000)")
    parser.add_argument("--resume", action='store_true', help='resume from lastest checkpoint in output_dir.')

    parser.add_argument("--dataset", type=str, help="dataset class name")
    parser.add_argument("--dataset_hparams", type=str, help="a string of comma separated list of dataset hyperparameters")
    parser.add_argument("--dataset_hparams_dict", type=str, help="a json file of dataset hyperparameters")
    parser.add_argument("--model", type=str, help="model class name")
    parser.add_argument("--model_hparams", type=str, help="a string of comma separated list of model hyperparameters")
    parser.add_argument("--model_hparams_dict", type=str, help="a json file of model hyperparameters")

    parser.add_argument("--summary_freq", type=int, default=1000, help="save frequency of summaries (except for image and eval summaries) for train/validation set")
    parser.add_argument("--image_summary_freq", type=int, default=5000, help="save frequency of image summaries for train/validation set")
    parser.add_argument("--eval_summary_freq", type=int, default=25000, help="save frequency of eval summaries for train/validation set")
    parser.add_argument("--accum_eval_summary_freq", type=int, default=100000, help="save frequency of accumulated eval summaries for validation set only")
    parser.add_argument("--progress_freq", type=int, default=100, help="display progress every progress_freq steps")
    parser.add_argument("--save_freq", type=int, default=5000, help="save frequence of model, 0 to disable")

    parser.add_argument("--aggregate_nccl", type=int, default=0, help="whether to use nccl or cpu for gradient aggregation in multi-gpu training")
    parser.add_argument("--gpu_mem_frac", type=float, default=0, help="fraction of gpu memory to use")
    parser.add_argument("--seed", type=int)
    # not in the reference: a record-free input for smoke runs
    parser.add_argument("--synthetic_shape", type=str, default='64,64,3', help="H,W,C of --dataset synthetic")
    return parser


def model_fname_from(model, model_hparams):
    """train.py:68-83: 'model=savp,lr=0.1,a=[1,2]' -> 'model.savp.lr.0.1.a.1..2'."""
    list_depth = 0
    out = ''
    for t in ('model=%s,%s' % (model, model_hparams)):
        if t == '[':
            list_depth += 1
        if t == ']':
            list_depth -= 1
        if list_depth and t == ',':
            t = '..'
        if t in '=,':
            t = '.'
        if t in '[]':
            t = ''
        out += t
    return out


def resolve_options(args):
    """train.py:68-118: output directory, --resume, hparams dicts from files and from the checkpoint directory."""
    if args.output_dir is None:
        args.output_dir = os.path.join(args.logs_dir, model_fname_from(args.model, args.model_hparams)) + args.output_dir_postfix
    if args.resume:
        if args.checkpoint:
            raise ValueError('resume and checkpoint cannot both be specified')
        args.checkpoint = args.output_dir
    dataset_hparams_dict, model_hparams_dict = {}, {}
    if args.dataset_hparams_dict:
        with open(args.dataset_hparams_dict) as f:
            dataset_hparams_dict.update(json.loads(f.read()))
    if args.model_hparams_dict:
        with open(args.model_hparams_dict) as f:
            model_hparams_dict.update(json.loads(f.read()))
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
                dataset_hparams_dict.update(json.loads(f.read()))
        except FileNotFoundError:
            print("dataset_hparams.json was not loaded because it does not exist")
        try:
            with open(os.path.join(checkpoint_dir, "model_hparams.json")) as f:
                model_hparams_dict.update(json.loads(f.read()))
        except FileNotFoundError:
            print("model_hparams.json was not loaded because it does not exist")
    return dataset_hparams_dict, model_hparams_dict


class SyntheticVideoDataset(object):
    """Seeded uniform[0,1) video with the dataset-class surface the runner touches (SURVEY.md 8(d) synthetic inputs)."""

    def __init__(self, input_dir, mode='train', num_epochs=None, seed=None, hparams_dict=None, hparams=None, shape=(64, 64, 3)):
        from video_prediction_amd.hparams import HParams
        self.mode, self.seed, self.image_shape = mode, seed or 0, tuple(shape)
        hp = HParams(context_frames=2, sequence_length=12, long_sequence_length=12, time_shift=2, frame_skip=0, force_time_shift=False,
                     shuffle_on_val=False, use_state=False, crop_size=0, scale_size=0)
        hp.override_from_dict(hparams_dict or {})
        if hparams:
            hp.parse(hparams)
        self.hparams = hp

    def num_examples_per_epoch(self):
        return 256

    def set_sequence_length(self, sequence_length):
        self.hparams.sequence_length = sequence_length

    def make_batch(self, batch_size, device='cuda:0', rank=0, world=1):
        import torch
        g = torch.Generator().manual_seed(1234 + 7919 * rank + (0 if self.mode == 'train' else 1))
        T = self.hparams.sequence_length
        while True:
            batch = {'images': torch.rand((batch_size, T) + self.image_shape, generator=g).to(device)}
            if self.hparams.use_state:          # the robot datasets' extra inputs (softmotion_dataset.py:62-64): 3-d end-effector states, 4-d actions
                batch['states'] = torch.randn(batch_size, T, 3, generator=g).cumsum(1).mul(0.1).to(device)
                batch['actions'] = torch.randn(batch_size, T - 1, 4, generator=g).to(device)
            yield batch


def get_dataset_class(name, synthetic_shape):
    if name == 'synthetic':
        import functools
        return functools.partial(SyntheticVideoDataset, shape=tuple(int(v) for v in synthetic_shape.split(',')))
    from video_prediction_amd import datasets
    return datasets.get_dataset_class(name)


def should(step, freq, max_steps, start_step):
    """train.py:232-236."""
    if freq is None:
        return (step + 1) == (max_steps - start_step)
    return bool(freq and ((step + 1) % freq == 0 or (step + 1) in (0, max_steps - start_step)))


def prune_checkpoints(output_dir, keep=2):
    """tf.train.Saver(max_to_keep=2), train.py:207."""
    import glob
    import re
    found = {}
    for f in glob.glob(os.path.join(output_dir, 'model-*.index')):
        m = re.search(r'model-(\d+)\.index$', f)
        if m:
            found[int(m.group(1))] = f[:-len('.index')]
    for step in sorted(found)[:-keep]:
        for f in glob.glob(found[step] + '.*'):
            os.remove(f)


def main(argv=None):
    args = build_parser().parse_args(argv)
    import torch
    if args.seed is not None:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
        random.seed(args.seed)
    dataset_hparams_dict, model_hparams_dict = resolve_options(args)

    rank, world, local_rank = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))
    chief = rank == 0
    if chief:
        print('----------------------------------- Options ------------------------------------')
        for k, v in args._get_kwargs():
            print(k, "=", v)
        print('------------------------------------- End --------------------------------------')
    if not torch.cuda.is_available():
        raise SystemExit('scripts/train.py needs an MI355X: the SAVP hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = 'cuda:%d' % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist_mod.init_process_group(backend='nccl', rank=rank, world_size=world)       # "nccl" is RCCL on ROCm
        dist = dist_mod

    from video_prediction_amd import models
    VideoDataset = get_dataset_class(args.dataset, args.synthetic_shape)
    train_dataset = VideoDataset(args.input_dir, mode='train', seed=args.seed, hparams_dict=dataset_hparams_dict, hparams=args.dataset_hparams)
    val_dataset = VideoDataset(args.val_input_dir or args.input_dir, mode='val', seed=args.seed, hparams_dict=dataset_hparams_dict,
                               hparams=args.dataset_hparams)

    VideoPredictionModel = models.get_model_class(args.model)
    hparams_dict = dict(model_hparams_dict)
    hparams_dict.update({'context_frames': train_dataset.hparams.context_frames,
                         'sequence_length': train_dataset.hparams.sequence_length,
                         'repeat': train_dataset.hparams.time_shift})                  # train.py:151-156
    model = VideoPredictionModel(hparams_dict=hparams_dict, hparams=args.model_hparams, aggregate_nccl=args.aggregate_nccl)
    batch_size = model.hparams.batch_size
    if batch_size % world:
        raise ValueError('batch_size %d is not divisible by %d GPUs' % (batch_size, world))
    per_gpu = batch_size // world                                                       # tf.split over the towers (base_model.py:523-527)
    train_iter = iter(train_dataset.make_batch(per_gpu, device=device, rank=rank, world=world))
    val_iter = iter(val_dataset.make_batch(per_gpu, device=device, rank=rank, world=world))
    inputs = next(train_iter)
    model.build_graph(inputs, device=device)
    if dist is not None:
        model.engine.attach_process_group(dist)

    if chief:
        if not os.path.exists(args.output_dir):
            os.makedirs(args.output_dir)
        with open(os.path.join(args.output_dir, "options.json"), "w") as f:
            f.write(json.dumps(vars(args), sort_keys=True, indent=4))
        with open(os.path.join(args.output_dir, "dataset_hparams.json"), "w") as f:
            f.write(json.dumps(train_dataset.hparams.values(), sort_keys=True, indent=4))
        with open(os.path.join(args.output_dir, "model_hparams.json"), "w") as f:
            f.write(json.dumps(model.hparams.values(), sort_keys=True, indent=4))
        store = model.engine.store
        print("parameter_count =", sum(int(np.prod(store[n].shape)) for n in store.names() if store.group_of[n] != 'aux'))
    if args.checkpoint:
        model.restore(args.checkpoint)
        if dist is not None:                                                            # every replica continues from rank 0's state
            for g in model.engine.store.groups.values():
                dist.broadcast(g.p, src=0)
    summaries = open(os.path.join(args.output_dir, 'summaries.jsonl'), 'a') if chief else None

    def scalars(info):
        out = {'d_loss': float(info['d_loss']), 'g_loss': float(info['g_loss'])}
        for k, (l, w) in list(info['d_losses'].items()) + list(info['g_losses'].items()):
            out[k] = float(l)
        return out

    max_steps = model.hparams.max_steps
    start_step = model.global_step
    start_time = time.time()
    info = None
    # start at one step earlier to log everything without doing any training; step is relative to start_step (train.py:240-242)
    def have_batch(it_inputs):
        """End of data (finite num_epochs) must be a collective decision: replicas read different file shards and may run dry at
        different steps -- a rank that left the loop alone would leave the others blocked in the gradient all-reduce."""
        ok = it_inputs is not None
        if dist is not None and getattr(train_dataset, 'num_epochs', None) is not None:     # endless data: no per-step collective / sync
            flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        return ok

    for step in range(-1, max_steps - start_step):
        if step == 1:
            start_time = time.time()            # skip step -1 and 0 for timing purposes (train.py:243-245)
        global_step = model.global_step
        if step >= 0:
            if step > 0:                        # the batch of step 0 was fetched for build_graph; later ones at the top of their own step,
                inputs = next(train_iter, None)  # so that the summary / progress / save blocks below always run for the step just trained
            if not have_batch(inputs):
                break
            run_start_time = time.time()
            info = model.train_step(inputs)
            if should(step, args.progress_freq, max_steps, start_step):
                torch.cuda.synchronize()
            run_elapsed_time = time.time() - run_start_time
            if run_elapsed_time > 1.5 and step > 0 and not should(step, args.progress_freq, max_steps, start_step):
                print('running train_op took too long (%0.1fs)' % run_elapsed_time)
        if chief and info is not None and should(step, args.summary_freq, max_steps, start_step):
            print("recording summary")
            summaries.write(json.dumps(dict(scalars(info), global_step=global_step, tag='summary')) + '\n')
            summaries.flush()
            print("done")
        if step >= 0 and should(step, args.eval_summary_freq, max_steps, start_step):
            # eval summary (train.py:264-265,301-305): best / mean / worst of eval_num_samples prior samples on a validation batch
            print("recording eval summary")
            _, metrics = model.eval_outputs_and_metrics_fn(next(val_iter))
            model.engine.set_images(inputs)                     # back to the training batch (images and, if the model has them, actions / states)
            if chief:
                summaries.write(json.dumps(dict({k: float(v.mean()) for k, v in metrics.items()}, global_step=global_step,
                                                tag='eval_summary_1')) + '\n')
                summaries.flush()
            print("done")
        if chief and should(step, args.progress_freq, max_steps, start_step):
            # global_step will have the correct step count if we resume from a checkpoint; it is read before it's incremented
            steps_per_epoch = train_dataset.num_examples_per_epoch() / batch_size
            train_epoch = global_step / steps_per_epoch
            print("progress  global step %d  epoch %0.1f" % (global_step + 1, train_epoch))
            if step > 0:
                elapsed_time = time.time() - start_time
                average_time = elapsed_time / step
                images_per_sec = batch_size / average_time
                remaining_time = (max_steps - (start_step + step + 1)) * average_time
                print("          image/sec %0.1f  remaining %dm (%0.1fh) (%0.1fd)" %
                      (images_per_sec, remaining_time / 60, remaining_time / 60 / 60, remaining_time / 60 / 60 / 24))
            if info is not None:
                if info['d_losses']:
                    print("d_loss", float(info["d_loss"]))
                for name, (loss, _) in info['d_losses'].items():
                    print("  ", name, float(loss))
                if info['g_losses']:
                    print("g_loss", float(info["g_loss"]))
                for name, (loss, _) in info['g_losses'].items():
                    print("  ", name, float(loss))
                print("learning_rate", info["learning_rate"])
        if chief and step >= 0 and should(step, args.save_freq, max_steps, start_step):
            print("saving model to", args.output_dir)
            model.save(os.path.join(args.output_dir, "model-%d" % model.global_step))
            prune_checkpoints(args.output_dir)
            print("done")
    if summaries:
        summaries.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
