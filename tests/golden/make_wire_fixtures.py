"""Generate the committed wire-format fixtures (run in the build container: `python tests/golden/make_wire_fixtures.py`):
  tiny.tfrecords                 two tf.train.Example records (3 frames of 4x4x3 uint8 + float features) in TFRecord framing
  tiny_ckpt.index / .data-*      a TensorFlow V2 checkpoint with five small tensors
  metrics_golden.npz             frames + psnr / mse / ssim of the fp64 oracle (tf.image.psnr / ssim restatement)
All are written by the ORACLE (oracle/tfrecord.py, oracle/tf_checkpoint.py, oracle/metrics.py): they pin the product readers and
kernels against committed bytes / numbers and the oracle writers against drift.  PARITY UNPINNED with respect to TensorFlow
itself (none of these files was produced by TensorFlow)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import metrics as OM, tf_checkpoint as OC, tfrecord as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def fixture_arrays():
    rng = np.random.default_rng(2024)
    frames = rng.integers(0, 256, (2, 3, 4, 4, 3), dtype=np.uint8)
    states = rng.standard_normal((2, 3, 3)).astype(np.float32)
    tensors = {'generator/encoder/z_mu/dense/bias': rng.standard_normal(8).astype(np.float32),
               'generator/encoder/z_mu/dense/kernel': rng.standard_normal((4, 8)).astype(np.float32),
               'generator/rnn/savp_cell/h0/conv2d/kernel': rng.standard_normal((3, 3, 2, 4)).astype(np.float32),
               'discriminator/video/sn_conv0_0/conv3d/u': rng.standard_normal((1, 4)).astype(np.float32),
               'global_step': np.asarray(300000, dtype=np.int64)}
    return frames, states, tensors


def main():
    frames, states, tensors = fixture_arrays()
    exs = []
    for i in range(2):
        f = {'%d/image_aux1/encoded' % t: frames[i, t].tobytes() for t in range(3)}
        f.update({'%d/endeffector_pos' % t: [float(x) for x in states[i, t]] for t in range(3)})
        exs.append(R.encode_example(f))
    R.write_records(os.path.join(HERE, 'tiny.tfrecords'), exs)
    OC.write(os.path.join(HERE, 'tiny_ckpt'), tensors, entries_per_block=2)
    rng = np.random.default_rng(7)
    a = rng.random((2, 2, 16, 16, 3))
    b = np.clip(a + 0.08 * rng.standard_normal(a.shape), 0, 1)
    ta, tb = torch.tensor(a), torch.tensor(b)
    np.savez_compressed(os.path.join(HERE, 'metrics_golden.npz'), a=a.astype(np.float32), b=b.astype(np.float32),
                        psnr=OM.psnr(ta.float().double(), tb.float().double()).numpy(),
                        mse=OM.mse(ta.float().double(), tb.float().double()).numpy(),
                        ssim=OM.ssim(ta.float().double(), tb.float().double()).numpy())


if __name__ == '__main__':
    main()
