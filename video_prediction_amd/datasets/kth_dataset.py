"""KTHVideoDataset with the reference's class surface (video_prediction/datasets/kth_dataset.py:16-46 on
base_dataset.py:394-453 VarLenFeatureVideoDataset) on libsavp_io.so.

Record layout (written by the reference's own preprocessing, kth_dataset.py:60-100): ONE tf.train.Example per sequence with int64
features 'sequence_length', 'height', 'width', 'channels' and a bytes_list 'images/encoded' holding one raw uint8 frame per entry
(jpeg_encoding False, :39-41).  Sequences shorter than hparams.sequence_length are dropped (filter, base_dataset.py:401-407); the
sub-sequence is sampled per example (slice_sequences, :189-229) -- both inside the C++ pipeline (SavpVideoPipelineArgs.var_len)."""
import glob
import itertools
import os

import numpy as np

from .. import io as sio
from .softmotion_dataset import SoftmotionVideoDataset


class KTHVideoDataset(SoftmotionVideoDataset):
    var_len = True

    def __init__(self, input_dir, mode='train', num_epochs=None, seed=None, hparams_dict=None, hparams=None):
        self.input_dir = os.path.normpath(os.path.expanduser(input_dir))
        self.mode = mode
        self.num_epochs = num_epochs
        self.seed = seed
        if self.mode not in ('train', 'val', 'test'):
            raise ValueError('Invalid mode %s' % self.mode)
        if not os.path.exists(self.input_dir):
            raise FileNotFoundError('input_dir %s does not exist' % self.input_dir)
        self.filenames = None
        for d in (self.input_dir, os.path.join(self.input_dir, self.mode)):          # base_dataset.py:36-43
            filenames = glob.glob(os.path.join(d, '*.tfrecord*'))
            if filenames:
                self.input_dir = d
                self.filenames = sorted(filenames)
                break
        if not self.filenames:
            raise FileNotFoundError('No tfrecords were found in %s.' % self.input_dir)
        self.dataset_name = os.path.basename(os.path.split(self.input_dir)[0])
        self.hparams = self.parse_hparams(hparams_dict, hparams)
        first = sio.read_records(self.filenames[0])[0]                                # kth_dataset.py:19-24
        self.image_shape = tuple(sio.example_int64(first, key) for key in ('height', 'width', 'channels'))
        self.image_key_fmt = 'images/encoded'
        self._max_sequence_length = 0                                                 # per example ('sequence_length' feature)
        self.state_like_names_and_shapes = {'images': (self.image_key_fmt, self.image_shape)}
        self.action_like_names_and_shapes = {}
        if self.hparams.use_state:
            raise NotImplementedError('KTH records carry no states / actions')
        if self.hparams.crop_size or self.hparams.scale_size:
            raise NotImplementedError('crop_size / scale_size are not supported by the HIP input path')

    def get_default_hparams_dict(self):
        """base_dataset.py:60-101 + kth_dataset.py:26-36."""
        base = dict(crop_size=0, scale_size=0, context_frames=1, sequence_length=0, long_sequence_length=0, frame_skip=0,
                    time_shift=1, force_time_shift=False, shuffle_on_val=False, use_state=False)
        over = dict(context_frames=10, sequence_length=20, long_sequence_length=40, force_time_shift=True, shuffle_on_val=True,
                    use_state=False)
        return dict(itertools.chain(base.items(), over.items()))

    def num_examples_per_epoch(self):
        """kth_dataset.py:43-47: sequences at least sequence_length long, from sequence_lengths.txt next to the records (falls back
        to reading the records' own 'sequence_length' features when the side file is absent)."""
        path = os.path.join(self.input_dir, 'sequence_lengths.txt')
        if os.path.exists(path):
            with open(path, 'r') as f:
                lengths = [int(line.strip()) for line in f.readlines() if line.strip()]
        else:
            lengths = [sio.example_int64(ex, 'sequence_length') for fn in self.filenames for ex in sio.read_records(fn)]
        return int(np.sum(np.array(lengths) >= self.hparams.sequence_length))
