"""libsavp_io.so (C++ TFRecord / tf.train.Example reader + batched video pipeline, include/savp_io.h) against the pure-Python
restatement of the wire formats (oracle/tfrecord.py) on fixture files written by the oracle.  CPU only."""
import os
import struct

import numpy as np
import pytest

from oracle import tfrecord as R
from video_prediction_amd import io as sio

H = W = 16
C = 3
FRAMES = 10


def _example(rng, idx):
    feats = {}
    frames = rng.integers(0, 256, (FRAMES, H, W, C), dtype=np.uint8)
    frames[:, 0, 0, 0] = idx                                           # tag
    for t in range(FRAMES):
        feats['%d/image_aux1/encoded' % t] = frames[t].tobytes()
        feats['%d/endeffector_pos' % t] = [float(idx), float(t), 0.5]
        if t < FRAMES - 1:
            feats['%d/action' % t] = [float(idx), float(t), 1.0, -1.0]
    feats['traj_id'] = ('int64', [idx, 7])
    return frames, R.encode_example(feats)


@pytest.fixture(scope='module')
def records(tmp_path_factory):
    d = tmp_path_factory.mktemp('bair') / 'train'
    d.mkdir()
    rng = np.random.default_rng(0)
    frames, paths = [], []
    idx = 0
    for f in range(3):
        exs = []
        for _ in range(5):
            fr, ex = _example(rng, idx)
            frames.append(fr); exs.append(ex); idx += 1
        p = str(d / ('traj_%d_to_%d.tfrecords' % (f * 5, f * 5 + 4)))
        R.write_records(p, exs)
        paths.append(p)
    return dict(dir=str(d.parent), paths=paths, frames=np.stack(frames))


def test_crc32c_known_answers_rfc3720():
    for data, want in ((b'123456789', 0xE3069283), (bytes(32), 0x8A9136AA), (bytes([0xff] * 32), 0x62A8AB43),
                       (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C)):
        assert sio.crc32c(data) == want == R.crc32c(data)
    rng = np.random.default_rng(1)
    for n in (0, 1, 7, 8, 9, 63, 1000):
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert sio.crc32c(b) == R.crc32c(b) and sio.masked_crc32c(b) == R.masked_crc32c(b)


def test_record_reader_matches_oracle_and_detects_corruption(records, tmp_path):
    for p in records['paths']:
        assert sio.read_records(p) == R.read_records(p)
    raw = bytearray(open(records['paths'][0], 'rb').read())
    raw[40] ^= 1                                                        # flip a payload bit
    bad = str(tmp_path / 'bad.tfrecords')
    open(bad, 'wb').write(raw)
    with pytest.raises(RuntimeError, match='corrupt'):
        sio.read_records(bad)
    empty = str(tmp_path / 'empty.tfrecords')
    open(empty, 'wb').close()
    assert sio.read_records(empty) == []


def test_example_features(records):
    ex = sio.read_records(records['paths'][1])[2]                        # global example 7
    kind, buf = sio.example_feature(ex, '3/image_aux1/encoded')
    assert kind == 1 and buf == records['frames'][7, 3].tobytes()
    assert sio.example_feature(ex, '4/endeffector_pos') == (2, [7.0, 4.0, 0.5])
    assert sio.example_feature(ex, 'traj_id') == (3, 2)
    with pytest.raises(RuntimeError, match='not found'):
        sio.example_feature(ex, '99/image_aux1/encoded')


def test_pipeline_in_order_with_frame_skip_and_float_features(records):
    seq, fs, B = 4, 1, 4
    pipe = sio.VideoPipeline(records['paths'], '%d/image_aux1/encoded', FRAMES, (H, W, C), seq, B, frame_skip=fs, time_shift=0,
                             shuffle=False, num_epochs=1, float_keys=[('%d/endeffector_pos', 3, 0), ('%d/action', 4, 1)])
    state_t, action_t = R.slice_times(FRAMES, seq, fs, 0)
    seen = 0
    while True:
        got = pipe.next()
        if got is None:
            break
        images, (states, actions) = got
        for b in range(B):
            i = seen + b
            assert np.array_equal(images[b], records['frames'][i][state_t])
            assert np.array_equal(states[b], np.array([[i, t, 0.5] for t in state_t], dtype=np.float32))
            want = np.array([[i, t, 1.0, -1.0] for t in action_t], dtype=np.float32).reshape(seq - 1, -1)    # base_dataset.py:223-226
            assert np.array_equal(actions[b], want)
        seen += B
    assert seen == 12                                                    # 15 examples, drop_remainder (base_dataset.py:149)
    pipe.close()


def test_pipeline_shuffle_epochs_and_time_shift(records):
    seq, B, epochs = 5, 5, 2
    pipe = sio.VideoPipeline(records['paths'], '%d/image_aux1/encoded', FRAMES, (H, W, C), seq, B, time_shift=2, shuffle=True,
                             shuffle_buffer=4, num_epochs=epochs, seed=3)
    tags, starts = [], set()
    while True:
        got = pipe.next()
        if got is None:
            break
        images, _ = got
        for b in range(B):
            i = int(images[b, 0, 0, 0, 0])
            tags.append(i)
            # which t_start reproduces this clip? (base_dataset.py:198-213: multiples of time_shift up to num_shifts)
            ok = [t0 for t0 in range(0, FRAMES - seq + 1, 2) if np.array_equal(images[b, :, 1:], records['frames'][i][t0:t0 + seq, 1:])]
            assert len(ok) == 1
            starts.add(ok[0])
    assert sorted(tags) == sorted(list(range(15)) * epochs)              # every example once per epoch
    assert tags[:15] != list(range(15))                                  # shuffled
    assert starts == {0, 2, 4}
    pipe.close()


def test_pipeline_reports_missing_feature(records):
    pipe = sio.VideoPipeline(records['paths'], '%d/image_view9/encoded', FRAMES, (H, W, C), 4, 2)
    with pytest.raises(RuntimeError, match='not found'):
        pipe.next()
    pipe.close()


def test_dataset_class_surface(records):
    from video_prediction_amd.datasets import get_dataset_class
    DS = get_dataset_class('bair')
    ds = DS(records['dir'], mode='train', num_epochs=1, seed=1, hparams='sequence_length=6,time_shift=0')
    assert ds.image_shape == (H, W, C) and ds.image_key_fmt == '%d/image_aux1/encoded' and ds._max_sequence_length == FRAMES
    assert ds.hparams.context_frames == 2 and ds.hparams.long_sequence_length == 30      # softmotion_dataset.py:47-52
    assert ds.num_examples_per_epoch() == 15 and not ds.jpeg_encoding
    pipe = ds.make_pipeline(3)
    images, _ = pipe.next()
    assert images.shape == (3, 6, H, W, C)
    pipe.close()
    with pytest.raises(ValueError):
        DS(records['dir'], mode='bogus')
