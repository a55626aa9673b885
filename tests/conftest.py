import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EVIDENCE_DIR = os.path.join(ROOT, 'gpurun_out', 'pytest_evidence')
_STATE = {'fingerprint': None, 'poison': False}


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'soak: long bit-identity repeat runs (part of -m gpu)')
    _install_poison()


def _install_poison():
    """SAVP_POISON=1 (or a list of alloc,scratch,lds,arena): the adversarial-memory mode of video_prediction_amd/debug.py.  'alloc' is done
    here, on the test side: torch.empty / empty_like / empty_strided / Tensor.new_empty hand out device memory whose every byte is 0xFF (NaN
    as fp32, bf16 and fp64), so a kernel that reads an element nobody wrote fails the parity check instead of passing on a box whose fresh
    VRAM pages happen to be zero."""
    from video_prediction_amd import debug
    if not debug.configure_from_env():
        return
    _STATE['poison'] = True
    if not debug.POISON['alloc']:
        return
    import torch

    def wrap(fn):
        def poisoned(*args, **kwargs):
            t = fn(*args, **kwargs)
            if t.is_cuda and t.numel():
                debug.poison_tensor(t)
                debug.COUNTS['alloc'] += 1
            return t
        poisoned.__wrapped__ = fn
        return poisoned

    for name in ('empty', 'empty_like', 'empty_strided'):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.new_empty = wrap(torch.Tensor.new_empty)


def _fingerprint():
    if _STATE['fingerprint'] is None:
        from video_prediction_amd import debug
        try:
            fp = debug.box_fingerprint()
            fp['id'] = debug.fingerprint_id(fp)
        except Exception as e:                  # never take the session down
            fp = {'id': 'unknown', 'error': repr(e)}
        _STATE['fingerprint'] = fp
    return _STATE['fingerprint']


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_sessionstart(session):
    """The full fingerprint goes next to the logs at the START of every GPU session (-q suppresses the header hook below)."""
    if not _gpu_present():
        return
    fp = _fingerprint()
    try:
        os.makedirs(EVIDENCE_DIR, exist_ok=True)
        with open(os.path.join(EVIDENCE_DIR, 'box_fingerprint_%s_pid%d.json' % (time.strftime('%Y%m%d_%H%M%S'), os.getpid())), 'w') as f:
            json.dump(dict(fp, poison=os.environ.get('SAVP_POISON')), f, indent=1, sort_keys=True, default=str)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    """One line that survives -q and `tail`: which box, which poison modes, how often each fired."""
    if not _gpu_present():
        return
    from video_prediction_amd import debug
    fp = _fingerprint()
    card = fp.get('visible_card') or {}
    terminalreporter.write_line('box: id=%s serial=%s vbios=%s partition=%s/%s host=%s | poison=%s fired=%s' % (
        fp.get('id'), card.get('serial_number'), card.get('vbios_version'), card.get('current_compute_partition'),
        card.get('current_memory_partition'), fp.get('host'), os.environ.get('SAVP_POISON') if _STATE['poison'] else 'off',
        json.dumps(debug.COUNTS)))


def pytest_report_header(config):
    """Which box is this?  (round 5: one lease failed 14 of 64 GPU tests with the binary that was green on four other boxes, and no log said
    which box that had been.)  Printed in every session header; written in full next to the logs when a GPU is present."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        return ['box: no GPU (CPU-side suite)']
    fp = _fingerprint()
    card = fp.get('visible_card') or (fp.get('cards') or [{}])[0]
    ras = card.get('ras') or {}
    bad_ras = {k: v for k, v in ras.items() if v and any(ch.isdigit() and ch != '0' for ch in v.replace('\n', ' ').split(':')[-1])}
    return ['box: id=%s host=%s amdgpu=%s vbios=%s partition=%s/%s torch=%s' % (
                fp.get('id'), fp.get('host'), fp.get('amdgpu_version'), card.get('vbios_version'), card.get('current_compute_partition'),
                card.get('current_memory_partition'), (fp.get('torch') or {}).get('name')),
            'box: ras error counters non-zero: %s' % (json.dumps(bad_ras) if bad_ras else 'none (or not exposed)'),
            'poison mode: %s' % (os.environ.get('SAVP_POISON') if _STATE['poison'] else 'off')]


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    """A failing GPU test leaves what is needed to judge it next to the log: the box, the library's option table, what the live tuner chose
    (kernels.AUTOTUNE['log']) and the full assertion text (the first bad indices are part of it: tests/gpu_checks.first_bad)."""
    outcome = yield
    rep = outcome.get_result()
    if rep.when != 'call' or not rep.failed or item.get_closest_marker('gpu') is None:
        return
    try:
        from video_prediction_amd import kernels as K, lib
        dump = {'test': item.nodeid, 'box': _fingerprint().get('id'), 'poison': os.environ.get('SAVP_POISON'),
                'source_id': lib.source_id(), 'longrepr': str(rep.longrepr)[-20000:]}
        try:
            dump['options'] = {n: lib.get_option(n) for n in lib.OPTION_NAMES}
        except Exception as e:
            dump['options'] = repr(e)
        dump['autotune'] = {'enabled': K.AUTOTUNE['enabled'], 'log': [[repr(k), list(v)] for k, v in K.AUTOTUNE['log'][-400:]],
                            'cache_entries': len(K.AUTOTUNE['cache']), 'rejected': [[repr(k), r] for k, r in K.AUTOTUNE.get('rejected', [])[-100:]]}
        os.makedirs(EVIDENCE_DIR, exist_ok=True)
        name = item.nodeid.replace('/', '_').replace('::', '__').replace('[', '_').replace(']', '')
        with open(os.path.join(EVIDENCE_DIR, 'FAILED_%s.json' % name[-150:]), 'w') as f:
            json.dump(dump, f, indent=1, default=str)
    except Exception:                           # evidence is best effort; the failure itself is already reported
        pass


@pytest.fixture(scope='session', autouse=True)
def _oracle_thread_cap():
    """The CPU oracle is thousands of small torch ops: on a 128-thread GPU box torch's default pool spends its time waking workers
    (bench.py's cpu_baseline measured 0.16 frames/s with 128 threads against 9.7 with 16).  The GPU parity tests spend most of their
    wall time in the oracle, so the whole session runs it on at most 16 threads."""
    import torch
    torch.set_num_threads(min(16, torch.get_num_threads()))
    yield


@pytest.fixture(scope='session')
def hip_lib():
    from video_prediction_amd import lib
    return lib.get()
