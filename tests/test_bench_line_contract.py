"""The bench.py contract, checked on the committed lines of the round (profiles/r05_bench*.json; bench.py itself needs the GPU): the keys the
driver reads, the roofline / cpu_baseline objects of the tier, internal consistency of the numbers, and that what the line quotes from
profiles/ (counter traffic, kernel families) belongs to the checkout's kernel sources."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'roofline')


def _line(path):
    rows = [l for l in open(path) if l.startswith('{')]
    assert rows, path
    return json.loads(rows[-1])


@pytest.mark.parametrize('name', ['r05_bench.json', 'r05_bench_final.json', 'r05_bench_c4_kth.json', 'r05_bench_c5_128.json',
                                  'r05_bench_c1_det.json', 'r05_bench_rccl_world1_forced.json', 'r05_bench_2ranks_one_gpu_gloo.json'])
def test_committed_bench_lines_follow_the_contract(name):
    d = _line(os.path.join(ROOT, 'profiles', name))
    for k in REQUIRED:
        assert k in d, (name, k)
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert d['unit'] == 'frames/s' and d['dtype'] in ('bf16', 'f32') and 'workload' in d['config'] and 'model' not in d['config']
    cfg = d['config']
    # value = whole-job frames per second: global batch x sequence length / step time
    assert abs(d['value'] - cfg['global_batch'] * cfg['seq_len'] / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value'], name
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-9
    assert 0.0 < r['frac'] < 1.0 and 'traffic' in r


def test_default_line_carries_the_tier_objects_and_same_source_evidence():
    from video_prediction_amd import lib
    d = _line(os.path.join(ROOT, 'profiles', 'r05_bench_final.json'))
    assert d['n_gpus'] == 1 and d['config']['workload'].startswith('c2') and d['config']['global_batch'] == 16 and d['config']['seq_len'] == 30
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'frames/s' and c['sample']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 2500.0                      # dense bf16 MFMA peak (MI355X_MICROARCH.md), not the sparse figure
    # counter traffic and kernel families are quoted only from files of the same kernel sources + tuning tables
    # (the id recorded in the committed files, not the working tree's: a later kernel change makes bench.py stop quoting them by itself)
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r05_convlstm_cell_pmc_bf16.json')))
    fam = json.load(open(os.path.join(ROOT, 'profiles', 'r05_kernel_families.json')))
    sid = pmc['source_id']
    assert fam['source_id'] == sid and sid in r['traffic_unit'] and len(sid) == len(lib.source_id())
    assert r['traffic'] == pmc['avg_hbm_bytes_per_launch_five_layers'] and r['traffic'] >= r['algorithmic_bytes'] > 0
    assert d['kernel_families_ms']['families'] == {k: v['ms_per_step'] for k, v in fam['families'].items()} or \
        set(d['kernel_families_ms']['families']) == set(fam['families'])
    for extra in ('roofline_step', 'roofline_cell', 'f32'):
        assert extra in d
    assert abs(d['roofline_step']['frac'] - d['roofline_step']['achieved'] / d['roofline_step']['peak']) <= 1e-9
