"""Bit-identity soak (-m gpu): the same launch sequence from the same state must produce the same BITS, every time, in every engine
instance, whatever the LDS and the allocator's free blocks held before.

Why: round 5's suite failed once on one box with a binary that was green on four others, and the bf16 bench's `losses` differed run to run
(fp32 atomics in the weight-gradient / loss partials; the next step rounds the updated weights to bf16 and the trajectories fork).  A
timing-dependent race or a read of unwritten memory has nowhere to hide in a run whose every output is required to repeat exactly; between
repeats all 160 KB of LDS of every CU and the caching allocator's free blocks are filled with NaN bit patterns (video_prediction_amd/debug.py).

What is compared (reference semantics: one sess.run(train_op), base_model.py:486-510; ConvLSTM cell rnn_ops.py:115-126): generated frames,
masks, losses, and after the train step every variable, both Adam moments of both optimiser groups and the spectral-norm u vectors."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'gpurun_out', 'pytest_evidence')


def _poison_between_repeats():
    from video_prediction_amd import debug
    debug.poison_lds()
    debug.poison_free_blocks(big_gb=2, small_mb=32)


def _bits(t):
    """A tensor's bit pattern as int32 / int16 / int64 (NaN-safe equality, -0.0 != +0.0)."""
    t = t.detach().contiguous()
    return t.view({2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()])


def _diff(name, got, ref, report):
    gb, rb = _bits(got), _bits(ref)
    if torch.equal(gb, rb):
        return
    ne = (gb != rb)
    idx = ne.reshape(-1).nonzero()[:4, 0].tolist()
    g, r = got.detach().double().reshape(-1), ref.detach().double().reshape(-1)
    report.append({'what': name, 'elements': int(ne.sum()), 'of': int(ne.numel()), 'first_bad_flat_indices': idx,
                   'got': [float(g[i]) for i in idx], 'ref': [float(r[i]) for i in idx],
                   'max_abs': float((g - r).abs().max()), 'nonfinite': int((~torch.isfinite(g)).sum())})


def _write(name, payload):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, name), 'w') as f:
            json.dump(payload, f, indent=1)
    except OSError:
        pass


def test_the_poison_tools_reach_what_the_next_kernel_and_the_next_allocation_find():
    """The soak's poison is only worth something if it lands: after savp_debug_poison_lds a kernel that READS its LDS without writing it
    finds the pattern in (nearly) every word on every CU it runs on; after poison_free_blocks a fresh torch.empty of a just-released
    block holds NaN; the scratch hand-out of POISON['scratch'] is NaN throughout."""
    from video_prediction_amd import debug, kernels as K, lib
    out = torch.zeros(2, dtype=torch.int64, device='cuda')
    debug.poison_lds()
    lib.check(lib.get_raw().savp_debug_probe_lds(lib.stream(), debug.NAN_WORD, lib.ptr(out)), 'savp_debug_probe_lds')
    torch.cuda.synchronize()
    hit, seen = int(out[0]), int(out[1])
    assert seen == 512 * 16384 and hit >= 0.99 * seen, (hit, seen)
    out.zero_()
    lib.check(lib.get_raw().savp_debug_poison_lds(lib.stream(), 0x12345678, None), 'savp_debug_poison_lds')
    lib.check(lib.get_raw().savp_debug_probe_lds(lib.stream(), debug.NAN_WORD, lib.ptr(out)), 'savp_debug_probe_lds')
    torch.cuda.synchronize()
    assert int(out[0]) <= 0.01 * int(out[1]), out.tolist()           # ... and the probe does see what is there, not what it hopes for
    x = torch.zeros(1 << 20, device='cuda')
    del x
    debug.poison_free_blocks(big_gb=1, small_mb=8)
    raw_empty = getattr(torch.empty, '__wrapped__', torch.empty)      # underneath SAVP_POISON's own wrapper, if that is on
    y = raw_empty(1 << 20, device='cuda')
    assert bool(torch.isnan(y).all())
    prev = debug.POISON['scratch']
    debug.POISON['scratch'] = True
    try:
        ws = K.scratch(torch.device('cuda:0'), 4096)
        torch.cuda.synchronize()
        assert bool(torch.isnan(ws).all())
    finally:
        debug.POISON['scratch'] = prev


def test_fp32_generator_forward_repeats_bit_identically_in_two_engines_under_lds_and_vram_poison():
    """The exact-fp32 generator forward (nz = 8: posterior + prior unrolls, encoder) x 50: 25 repeats in each of two engine instances,
    NaN-poisoned LDS and free blocks between repeats; frames, masks, CDNA kernels and the encoder's (mu, log sigma^2) repeat bit for bit --
    and the first one matches the fp64 oracle (so "identical" is not "identically wrong")."""
    from tests import gpu_model_checks as G
    from video_prediction_amd import kernels as K, variables as V
    from video_prediction_amd.models.savp_model import SAVPEngine
    K.set_conv_precision('f32')
    res = G.check_generator_forward(nz=8, B=2, T=6, tag='soak_gen_fwd')
    assert all(e <= t for _, e, t in res), [r for r in res if not r[1] <= r[2]]
    hp = G.make_hparams(context_frames=2, sequence_length=6, nz=8, schedule_sampling='inverse_sigmoid')
    specs = V.variable_specs(hp, (64, 64, 3), mode='test')
    vals = V.init_variables(specs, seed=4)
    images = G.synth(hp, 2, 64, 64, 3, 0).float().cuda()
    noise = G.make_noise(hp, 2, sampling=True)
    ref, report, repeats = None, [], 0
    for inst in range(2):
        eng = SAVPEngine(hp, (64, 64, 3), 2, mode='test', values=vals, device='cuda:0')
        eng.mode = 'train'
        eng.set_images(images, time_major=True)
        for rep in range(25):
            eng.prep_generator_weights()
            gen = eng.forward_generator(noise, collect_masks=True)
            torch.cuda.synchronize()
            out = {'gen_images': gen.clone(), 'masks': eng.gen.masks.clone(), 'cdna_kernels': eng.gen.cdna_kern.v.clone(),
                   'zs_mu': eng.enc.mu.clone(), 'zs_log_sigma_sq': eng.enc.ls.clone()}
            assert all(bool(torch.isfinite(v.float()).all()) for v in out.values())
            if ref is None:
                ref = out
            else:
                for k, v in out.items():
                    _diff('engine %d repeat %d: %s' % (inst, rep, k), v, ref[k], report)
            repeats += 1
            _poison_between_repeats()
        del eng
        torch.cuda.empty_cache()
    _write('soak_gen_fwd_f32.json', {'repeats': repeats, 'engines': 2, 'differences': report[:50]})
    assert not report, report[:5]


def _state(eng):
    G_ = eng.store.groups
    s = {'aux.p': G_['aux'].p.clone()}
    for g in ('g', 'd'):
        s[g + '.p'], s[g + '.m'], s[g + '.v'] = G_[g].p.clone(), G_[g].m.clone(), G_[g].v.clone()
    return s


def _restore(eng, s):
    G_ = eng.store.groups
    G_['aux'].p.copy_(s['aux.p'])
    for g in ('g', 'd'):
        G_[g].p.copy_(s[g + '.p'])
        G_[g].m.copy_(s[g + '.m'])
        G_[g].v.copy_(s[g + '.v'])
        G_[g].t = 0
    eng.step = 0


def _per_variable(eng, group, got, ref, what, report):
    """Name the variables whose slice of a flat arena differs (the question a failing soak has to answer: WHICH reduction is not repeatable)."""
    arena = eng.store.groups[group].arena
    for name in arena.names():
        a, b = arena.view_of(got, name), arena.view_of(ref, name)
        if not torch.equal(_bits(a), _bits(b)):
            _diff('%s of %s' % (what, name), a, b, report)


@pytest.mark.parametrize('replayed', [False, True], ids=['eager', 'replayed'])
def test_c2_bf16_train_step_repeats_bit_identically_in_two_engines_under_lds_and_vram_poison(replayed):
    """The benchmarked step (c2: B = 16, T = 30, bf16 datapath, shipped tuning table, the recipe's learning rate) x 20 from identical state:
    10 repeats in each of two engine instances, launched one by one or replayed as the captured hipGraph.  Every repeat's generated frames,
    losses, updated variables, Adam moments (m = (1 - beta1) g: the gradients themselves) and spectral-norm vectors equal the first
    repeat's bit for bit; a difference is reported per variable."""
    import gc
    from tests import gpu_model_checks as G
    from tests.test_gpu_model import _bench_engine
    from video_prediction_amd import kernels as K
    case = G.BENCH_CASES['c2']
    saved = dict(K.AUTOTUNE, cache=dict(K.AUTOTUNE['cache']))
    ref, report, repeats = None, [], 0
    try:
        for inst in range(2):
            eng, gold, noise = _bench_engine('c2_step_golden.npz', case, graph=replayed)
            s0 = _state(eng)
            n = 10 + (2 if replayed else 0)          # replayed: repeat 0 runs eagerly, repeat 1 captures + replays, the rest replay
            for rep in range(n):
                _restore(eng, s0)
                info = eng.train_step(noise)
                torch.cuda.synchronize()
                if replayed and rep >= 1:
                    assert eng.graph is not None and eng.graph.segments == 1
                out = _state(eng)
                out['gen_images'] = eng.gen.gen.v.clone()
                out['losses'] = torch.stack([info['d_loss'].reshape(()).double(), info['g_loss'].reshape(()).double()] +
                                            [l.reshape(()).double() for l, w in info['g_losses'].values()] +
                                            [l.reshape(()).double() for l, w in info['d_losses'].values()]).clone()
                assert all(bool(torch.isfinite(v.float()).all()) for v in out.values()), [k for k, v in out.items() if not bool(torch.isfinite(v.float()).all())]
                if ref is None:
                    ref = out
                else:
                    tag = 'engine %d repeat %d' % (inst, rep)
                    for k in ('gen_images', 'losses', 'aux.p'):
                        _diff('%s: %s' % (tag, k), out[k], ref[k], report)
                    for g in ('g', 'd'):
                        for kind in ('m', 'v', 'p'):
                            _per_variable(eng, g, out[g + '.' + kind], ref[g + '.' + kind], '%s: %s.%s' % (tag, g, kind), report)
                repeats += 1
                _poison_between_repeats()
                if len(report) > 200:
                    break
            del eng, s0
            gc.collect()
            torch.cuda.empty_cache()
        names = sorted(set(r['what'].split(' of ')[-1] for r in report if ' of ' in r['what']))
        _write('soak_c2_bf16_%s.json' % ('replayed' if replayed else 'eager'),
               {'repeats': repeats, 'engines': 2, 'variables_that_differ': names[:200], 'differences': report[:60],
                'losses_first_repeat': [float(x) for x in ref['losses']]})
        assert not report, (names[:20], report[:3])
    finally:
        K.set_conv_precision('f32')
        K.AUTOTUNE.update(enabled=saved['enabled'], cache=saved['cache'])
