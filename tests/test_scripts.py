"""Host logic of scripts/train.py / scripts/generate.py against the reference's command line (flags, defaults, directory naming,
step-frequency rules, side files) -- CPU only; the GPU smoke of the runners is tests/test_gpu_model.py::test_train_and_generate_scripts."""
import json
import os
import re
import struct
import zlib

import numpy as np
import pytest

from scripts import generate as G
from scripts import train as T

REF = '/root/reference/scripts'


def _ref_flags(path):
    src = open(path).read()
    return set(re.findall(r'add_argument\(\s*"(--[a-z_]+)"', src))


@pytest.mark.skipif(not os.path.exists(REF), reason='reference not mounted')
def test_flags_match_the_reference_scripts():
    ours = {a.option_strings[0] for a in T.build_parser()._actions if a.option_strings and a.option_strings[0] != '-h'}
    assert _ref_flags(os.path.join(REF, 'train.py')) <= ours, _ref_flags(os.path.join(REF, 'train.py')) - ours
    assert ours - _ref_flags(os.path.join(REF, 'train.py')) == {'--synthetic_shape'}
    ours = {a.option_strings[0] for a in G.build_parser()._actions if a.option_strings and a.option_strings[0] != '-h'}
    assert _ref_flags(os.path.join(REF, 'generate.py')) <= ours
    assert ours - _ref_flags(os.path.join(REF, 'generate.py')) == {'--synthetic_shape'}


def test_flag_defaults():
    a = T.build_parser().parse_args(['--input_dir', 'x'])
    assert (a.logs_dir, a.summary_freq, a.image_summary_freq, a.eval_summary_freq, a.accum_eval_summary_freq, a.progress_freq,
            a.save_freq, a.aggregate_nccl, a.gpu_mem_frac, a.seed) == ('logs', 1000, 5000, 25000, 100000, 100, 5000, 0, 0, None)
    g = G.build_parser().parse_args(['--input_dir', 'x'])
    assert (g.results_dir, g.mode, g.batch_size, g.num_epochs, g.num_stochastic_samples, g.fps, g.seed) == ('results', 'val', 8, 1, 5, 4, 7)


def test_output_dir_naming_like_train_py_68_83():
    assert T.model_fname_from('savp', 'lr=0.001,schedule_sampling_steps=[0,100]') == 'model.savp.lr.0.001.schedule_sampling_steps.0..100'
    assert T.model_fname_from('savp', None) == 'model.savp.None'
    a = T.build_parser().parse_args(['--input_dir', 'x', '--model', 'savp', '--model_hparams', 'nz=8', '--output_dir_postfix', '_run1'])
    T.resolve_options(a)
    assert a.output_dir == os.path.join('logs', 'model.savp.nz.8') + '_run1'


def test_resume_reads_options_and_hparams_back(tmp_path):
    d = tmp_path / 'run'
    d.mkdir()
    (d / 'options.json').write_text(json.dumps({'dataset': 'bair', 'model': 'savp'}))
    (d / 'model_hparams.json').write_text(json.dumps({'nz': 8, 'lr': 0.0002}))
    (d / 'dataset_hparams.json').write_text(json.dumps({'sequence_length': 12}))
    a = T.build_parser().parse_args(['--input_dir', 'x', '--output_dir', str(d), '--resume'])
    ds, mh = T.resolve_options(a)
    assert (a.checkpoint, a.dataset, a.model) == (str(d), 'bair', 'savp') and mh == {'nz': 8, 'lr': 0.0002} and ds == {'sequence_length': 12}
    with pytest.raises(ValueError):
        T.resolve_options(T.build_parser().parse_args(['--input_dir', 'x', '--output_dir', str(d), '--resume', '--checkpoint', str(d)]))
    # a checkpoint PREFIX names its directory (train.py:98-101); generate.py derives the result directories from it (:77-79)
    g = G.build_parser().parse_args(['--input_dir', 'x', '--checkpoint', str(d / 'model-100')])
    ds, mh = G.resolve_options(g)
    assert (g.dataset, g.model, mh['nz']) == ('bair', 'savp', 8)
    assert g.output_png_dir == os.path.join('results', 'run') and g.output_gif_dir == os.path.join('results', 'run')
    with pytest.raises(ValueError):
        G.resolve_options(G.build_parser().parse_args(['--input_dir', 'x']))


def test_should_rule_like_train_py_232_236():
    max_steps, start = 1000, 0
    assert T.should(-1, 100, max_steps, start)                      # step -1: log everything before training
    assert T.should(99, 100, max_steps, start) and not T.should(100, 100, max_steps, start)
    assert T.should(999, 7, max_steps, start)                       # the last step always
    assert not T.should(5, 0, max_steps, start)                     # 0 disables
    assert T.should(999, None, max_steps, start) and not T.should(99, None, max_steps, start)


def test_checkpoint_pruning_keeps_two(tmp_path):
    for step in (5, 10, 15):
        for ext in ('.index', '.data-00000-of-00001'):
            (tmp_path / ('model-%d%s' % (step, ext))).write_bytes(b'x')
    T.prune_checkpoints(str(tmp_path), keep=2)
    left = sorted(os.listdir(str(tmp_path)))
    assert left == ['model-10.data-00000-of-00001', 'model-10.index', 'model-15.data-00000-of-00001', 'model-15.index']


def test_png_writer_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for c in (1, 3):
        img = rng.integers(0, 256, (5, 7, c), dtype=np.uint8)
        p = str(tmp_path / ('a%d.png' % c))
        G.write_png(p, img)
        raw = open(p, 'rb').read()
        assert raw[:8] == b'\x89PNG\r\n\x1a\n'
        pos, chunks = 8, {}
        while pos < len(raw):
            n, tag = struct.unpack('>I4s', raw[pos:pos + 8])
            data = raw[pos + 8:pos + 8 + n]
            assert struct.unpack('>I', raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xffffffff
            chunks[tag] = data
            pos += 12 + n
        w, h, depth, ctype = struct.unpack('>IIBB', chunks[b'IHDR'][:10])
        assert (w, h, depth, ctype) == (7, 5, 8, 0 if c == 1 else 2)
        rows = zlib.decompress(chunks[b'IDAT'])
        got = np.frombuffer(rows, dtype=np.uint8).reshape(5, 1 + 7 * c)[:, 1:].reshape(5, 7, c)
        assert np.array_equal(got, img)
