"""SoftmotionVideoDataset (BAIR robot pushing) with the reference's class surface
(video_prediction/datasets/softmotion_dataset.py:11-82, base_dataset.py:12-232,235-353) on libsavp_io.so:
C++ TFRecord reading / Example parsing / sub-sequence sampling / shuffling / batching / prefetch, uint8 over PCIe,
conversion to float32 [0,1] on the GPU.  Not supported (raise): crop_size / scale_size (resizing), jpeg encoding, object_pos
pixel distributions."""
import glob
import itertools
import os
import re

import numpy as np
import torch

from .. import io as sio
from ..hparams import HParams


class SoftmotionVideoDataset(object):
    def __init__(self, input_dir, mode='train', num_epochs=None, seed=None, hparams_dict=None, hparams=None):
        """base_dataset.py:13-58: input_dir holds train/ val/ test/ sub-directories of *.tfrecord* files (or is one of them)."""
        self.input_dir = os.path.normpath(os.path.expanduser(input_dir))
        self.mode = mode
        self.num_epochs = num_epochs
        self.seed = seed
        if self.mode not in ('train', 'val', 'test'):
            raise ValueError('Invalid mode %s' % self.mode)
        if not os.path.exists(self.input_dir):
            raise FileNotFoundError('input_dir %s does not exist' % self.input_dir)
        self.filenames = None
        # look for tfrecords in input_dir and input_dir/mode directories (base_dataset.py:36-43)
        for d in (self.input_dir, os.path.join(self.input_dir, self.mode)):
            filenames = glob.glob(os.path.join(d, '*.tfrecord*'))
            if filenames:
                self.input_dir = d
                self.filenames = sorted(filenames)
                break
        if not self.filenames:
            raise FileNotFoundError('No tfrecords were found in %s.' % self.input_dir)
        self.dataset_name = os.path.basename(os.path.split(self.input_dir)[0])
        self.hparams = self.parse_hparams(hparams_dict, hparams)
        # infer the image feature name, frames per example and image shape from the first example (softmotion_dataset.py:15-43,
        # base_dataset.py:264-312)
        first = sio.read_records(self.filenames[0])[0]
        self._first = first
        names = self._feature_names(first)
        image_names = set(m.group(1) for m in (re.search(r'\d+/(\w+)/encoded', n) for n in names) if m)
        image_name = next((n for n in ('image_aux1', 'image_view0') if n in image_names), None)
        if not image_name:
            if len(image_names) == 1:
                image_name = image_names.pop()
            else:
                raise ValueError('The examples have images under more than one name.')
        self.image_key_fmt = '%%d/%s/encoded' % image_name
        self._max_sequence_length = 1 + max(int(m.group(1)) for m in (re.match(r'(\d+)/%s/encoded' % image_name, n) for n in names) if m)
        _, buf = sio.example_feature(first, self.image_key_fmt % 0)
        side = int(round((len(buf) // 3) ** 0.5))
        if side * side * 3 != len(buf):
            raise ValueError('cannot infer a square RGB image shape from %d bytes' % len(buf))
        self.image_shape = (side, side, 3)
        self.state_like_names_and_shapes = {'images': (self.image_key_fmt, self.image_shape)}
        self.action_like_names_and_shapes = {}
        if self.hparams.use_state:
            self.state_like_names_and_shapes['states'] = ('%d/endeffector_pos', (3,))
            self.action_like_names_and_shapes['actions'] = ('%d/action', (4,))
        if self.hparams.crop_size or self.hparams.scale_size:
            raise NotImplementedError('crop_size / scale_size are not supported by the HIP input path')

    @staticmethod
    def _feature_names(example):
        """Feature keys of a serialized tf.train.Example (minimal wire-format walk)."""
        def varint(b, i):
            v = s = 0
            while True:
                c = b[i]; i += 1
                v |= (c & 0x7f) << s; s += 7
                if not c & 0x80:
                    return v, i
        names, i = [], 0
        _, i = varint(example, i)
        n, i = varint(example, i)
        feats, j = example[i:i + n], 0
        while j < len(feats):
            _, j = varint(feats, j)
            m, j = varint(feats, j)
            entry, j = feats[j:j + m], j + m
            _, k = varint(entry, 0)
            ln, k = varint(entry, k)
            names.append(entry[k:k + ln].decode())
        return names

    def get_default_hparams_dict(self):
        """base_dataset.py:60-101 + softmotion_dataset.py:45-53."""
        base = dict(crop_size=0, scale_size=0, context_frames=1, sequence_length=0, long_sequence_length=0, frame_skip=0,
                    time_shift=1, force_time_shift=False, shuffle_on_val=False, use_state=False)
        over = dict(context_frames=2, sequence_length=12, long_sequence_length=30, time_shift=2)
        return dict(itertools.chain(base.items(), over.items()))

    def get_default_hparams(self):
        return HParams(**self.get_default_hparams_dict())

    def parse_hparams(self, hparams_dict, hparams):
        parsed = self.get_default_hparams().override_from_dict(hparams_dict or {})
        if hparams:
            if not isinstance(hparams, (list, tuple)):
                hparams = [hparams]
            for h in hparams:
                parsed.parse(h)
        if parsed.long_sequence_length == 0:
            parsed.long_sequence_length = parsed.sequence_length
        return parsed

    @property
    def jpeg_encoding(self):
        return False

    def num_examples_per_epoch(self):
        """softmotion_dataset.py:70-82: trajectory ranges are encoded in the file names."""
        count = 0
        for filename in self.filenames:
            match = re.search(r'traj_(\d+)_to_(\d+).tfrecords', os.path.basename(filename))
            if not match:
                return sum(len(sio.read_records(f)) for f in self.filenames)
            count += int(match.group(2)) - int(match.group(1)) + 1
        return count

    def _shard(self, rank, world):
        """Files (and the seed) of one data-parallel replica: every replica reads its own share of the record files -- the
        reference feeds all towers from ONE iterator and tf.split()s the batch (base_model.py:523-527), so distinct towers see
        distinct sequences; with one process per GPU that becomes distinct files per rank (round-robin; with fewer files than
        ranks every rank reads everything in its own shuffled order)."""
        files = self.filenames[rank::world] if len(self.filenames) >= world else self.filenames
        seed = ((self.seed or 0) * 1000003 + rank * 7919) & 0xffffffffffffffff
        if self.seed is None and world == 1:
            seed = 0
        return files, seed

    def make_pipeline(self, batch_size, prefetch_batches=2, rank=0, world=1):
        hp = self.hparams
        shuffle = self.mode == 'train' or (self.mode == 'val' and hp.shuffle_on_val)        # base_dataset.py:131
        time_shift = hp.time_shift if ((hp.time_shift and self.mode == 'train') or hp.force_time_shift) else 0   # :198
        float_keys = []
        if hp.use_state:
            float_keys = [('%d/endeffector_pos', 3, 0), ('%d/action', 4, 1)]
        files, seed = self._shard(rank, world)
        return sio.VideoPipeline(files, self.image_key_fmt, self._max_sequence_length, self.image_shape,
                                 hp.sequence_length, batch_size, frame_skip=hp.frame_skip, time_shift=time_shift, shuffle=shuffle,
                                 num_epochs=self.num_epochs, seed=seed, prefetch_batches=prefetch_batches,
                                 float_keys=float_keys, var_len=self.var_len)

    var_len = False          # one feature per frame (softmotion); KTHVideoDataset: one bytes_list per sequence

    def make_batch(self, batch_size, device='cuda:0', rank=0, world=1):
        """base_dataset.py:153-156: an iterator of input dicts {'images': float32 [B,T,H,W,C] in [0,1] on the device, ('states',
        'actions')}.  Frames cross PCIe as uint8 from pinned memory; conversion + layout change happen in one HIP kernel.
        rank / world: the data-parallel replica this iterator feeds (see _shard)."""
        return _BatchIterator(self, batch_size, device, rank, world)

    def set_sequence_length(self, sequence_length):
        """base_dataset.py:103-104."""
        self.hparams.sequence_length = sequence_length


class _BatchIterator(object):
    def __init__(self, ds, batch_size, device, rank=0, world=1):
        from .. import kernels as K
        self.K = K
        self.ds, self.device = ds, torch.device(device)
        self.pipe = ds.make_pipeline(batch_size, rank=rank, world=world)
        B, T = batch_size, ds.hparams.sequence_length
        self.host = torch.empty((B, T) + ds.image_shape, dtype=torch.uint8).pin_memory()
        self.dev_u8 = torch.empty((B, T) + ds.image_shape, dtype=torch.uint8, device=self.device)
        self.copied = None           # event recorded behind the H2D copy out of the pinned buffer

    def __iter__(self):
        return self

    def __next__(self):
        if self.copied is not None:
            self.copied.synchronize()     # the previous batch has left the pinned buffer before the reader refills it
        got = self.pipe.next(self.host.numpy())
        if got is None:
            raise StopIteration
        _, floats = got
        self.dev_u8.copy_(self.host, non_blocking=True)
        if self.device.type == 'cuda':
            self.copied = torch.cuda.Event()
            self.copied.record(torch.cuda.current_stream(self.device))
        B, T = self.dev_u8.shape[:2]
        images_tm = torch.empty((T, B) + self.ds.image_shape, device=self.device)
        self.K.u8_frames_to_f32(self.dev_u8, images_tm)
        out = {'images': images_tm.transpose(0, 1)}                      # batch-major view, like the reference's iterator
        if floats:
            out['states'] = torch.from_numpy(floats[0]).to(self.device)
            out['actions'] = torch.from_numpy(floats[1]).to(self.device)
        return out

    next = __next__
