"""CPU-side checks: the C-ABI library loads and exports every symbol include/savp_hip.h declares (no compute calls
without a GPU), the product never imports the oracle, hparams / model-class behaviour mirrors the reference."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'savp_hip.h')).read()
    return sorted(set(re.findall(r'\b(savp_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol(hip_lib):
    from video_prediction_amd import lib
    syms = _declared_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), 'libsavp_hip.so does not export %s' % s
    # every symbol the Python side binds is declared in the header, and vice versa (savp_version is bound separately)
    bound = set(lib.EXPORTS) | {'savp_version'}
    assert bound == set(syms), (sorted(bound - set(syms)), sorted(set(syms) - bound))
    assert hip_lib.savp_version().startswith(b'savp_hip')


def test_io_library_exports_every_declared_symbol():
    """libsavp_io.so (host C++ input pipeline) exports everything include/savp_io.h declares."""
    from video_prediction_amd import io as sio
    txt = open(os.path.join(ROOT, 'include', 'savp_io.h')).read()
    syms = sorted(set(re.findall(r'\b(savp_[a-z0-9_]+)\s*\(', txt)))
    assert len(syms) >= 11
    sio.get()
    raw = ctypes.CDLL(sio.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), 'libsavp_io.so does not export %s' % s


def test_ops_fail_loudly_without_device():
    import torch
    from video_prediction_amd import kernels as K, lib
    x = torch.zeros(1, 4, 4, 4)
    with pytest.raises(RuntimeError, match='no CPU path'):
        K.conv(lib.CONV_FPROP, K.ConvGeom((3, 3), (1, 1), (1, 1)), x, x, torch.zeros(9 * 16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'video_prediction_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py') or f.endswith('.hip') or f.endswith('.h') or f.endswith('.cpp'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), '%s imports the oracle' % f


def test_hparams_parsing_like_contrib_hparams():
    from video_prediction_amd.hparams import HParams
    hp = HParams(a=1, b=2.0, c='x', d=(1, 2), e=True)
    hp.override_from_dict({'a': 3, 'd': [5, 6]})
    hp.parse('b=0.5,c=hello,d=[7,8],e=false')
    assert hp.a == 3 and hp.b == 0.5 and hp.c == 'hello' and hp.d == (7, 8) and hp.e is False
    assert hp.values()['a'] == 3
    with pytest.raises(ValueError):
        hp.override_from_dict({'nope': 1})
    with pytest.raises(ValueError):
        hp.parse('nope=1')


def test_model_class_surface():
    from video_prediction_amd.models import get_model_class
    M = get_model_class('savp')
    with pytest.raises(ValueError, match='mode must be train or test'):
        M(mode='val', hparams_dict=dict(context_frames=2, sequence_length=12))
    with pytest.raises(ValueError, match='context_frames'):
        M(mode='train')
    with pytest.raises(ValueError, match='Invalid model'):
        get_model_class('nope')
    # published ours_savp recipe (hparams/bair_action_free/ours_savp/model_hparams.json) incl. a deprecated key
    recipe = {"batch_size": 16, "lr": 0.0002, "beta1": 0.5, "beta2": 0.999, "l1_weight": 100.0, "l2_weight": 0.0,
              "kl_weight": 1.0, "video_sn_vae_gan_weight": 0.1, "video_sn_gan_weight": 0.1,
              "vae_gan_feature_cdist_weight": 10.0, "gan_feature_cdist_weight": 0.0, "state_weight": 0.0, "gan_weight": 1.0}
    m = M(mode='train', hparams_dict=dict(recipe, context_frames=2, sequence_length=30), hparams='nz=8,kernel_size=[5,5]')
    assert m.hparams.lr == 0.0002 and m.hparams.nz == 8 and m.hparams.kernel_size == (5, 5)
    assert m.hparams.transformation == 'cdna' and m.hparams.conv_rnn == 'lstm' and m.hparams.ngf == 32
    assert not m.deterministic
    t = M(mode='test', hparams_dict=dict(recipe, context_frames=2, sequence_length=30))
    assert t.discriminator_fn is None


def test_reference_recipes_load_unchanged_when_reference_is_mounted():
    import glob
    import json
    from video_prediction_amd.models import get_model_class
    files = glob.glob('/root/reference/hparams/*/ours_*/model_hparams.json')
    if not files:
        pytest.skip('reference not mounted (GPU box)')
    M = get_model_class('savp')
    for f in files:
        d = json.load(open(f))
        m = M(mode='train', hparams_dict=dict(d, context_frames=2, sequence_length=12))
        for k, v in d.items():
            assert getattr(m.hparams, k) == v


def test_variable_inventory_matches_survey_counts():
    # BAIR SAVP = 79 G+E tensors / 7.35 M params, 32 D tensors / 10.29 M params (trainable)
    import numpy as np
    from video_prediction_amd import variables as V
    from video_prediction_amd.models import get_model_class
    m = get_model_class('savp')(mode='train', hparams_dict=dict(context_frames=2, sequence_length=30, video_sn_gan_weight=0.1,
                                                                  video_sn_vae_gan_weight=0.1))
    specs = V.variable_specs(m.hparams, (64, 64, 3))
    g = [(n, s) for n, (s, _) in specs.items() if n.startswith('generator/')]
    d = [(n, s) for n, (s, _) in specs.items() if n.startswith('discriminator/') and V.is_trainable(n)]
    assert len(g) == 79 and len(d) == 32
    assert abs(sum(int(np.prod(s)) for _, s in g) / 1e6 - 7.35) < 0.01
    assert abs(sum(int(np.prod(s)) for _, s in d) / 1e6 - 10.29) < 0.01
    assert specs['generator/rnn/savp_cell/lstm_h2/basic_conv2dlstm_cell/kernel'][0] == (5, 5, 264, 512)
    assert specs['discriminator/encoder/video/sn_fc4/dense/kernel'][0] == (65536, 1)


def test_tuning_table_round_trip(tmp_path):
    """The shipped conv tuning tables parse, and save/load reproduces the cache (keys are reprs of the problem tuples)."""
    import glob
    import os
    from video_prediction_amd import kernels as K
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved = dict(K.AUTOTUNE['cache'])
    try:
        for path in glob.glob(os.path.join(root, 'video_prediction_amd', 'tuning_gfx950_*.json')):
            K.AUTOTUNE['cache'].clear()
            n = K.load_tuning(path)
            assert n == len(K.AUTOTUNE['cache']) and n > 20
            for key, (tile, sk) in K.AUTOTUNE['cache'].items():
                assert isinstance(key, tuple) and key[0] in (0, 1, 2) and 0 <= tile < 0x4000 and 0 <= sk <= 64
            out = str(tmp_path / 'table.json')
            K.save_tuning(out)
            before = dict(K.AUTOTUNE['cache'])
            K.AUTOTUNE['cache'].clear()
            K.load_tuning(out)
            assert K.AUTOTUNE['cache'] == before
    finally:
        K.AUTOTUNE['cache'].clear()
        K.AUTOTUNE['cache'].update(saved)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Every argument struct mirrored in lib.py has the size and field offsets gcc gives the struct of the same name in
    include/savp_hip.h (a field appended on one side only would silently shift everything behind it)."""
    import ctypes
    import shutil
    import subprocess
    from video_prediction_amd import lib
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    header = open(os.path.join(ROOT, 'include', 'savp_hip.h')).read()
    structs = [(n, c) for n, c in vars(lib).items()
               if isinstance(c, type) and issubclass(c, ctypes.Structure) and n.startswith('Savp') and ('} %s;' % n) in header]
    assert len(structs) >= 5
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "savp_hip.h"', 'int main() {']
    for n, c in structs:
        lines.append('printf("%s %%zu", sizeof(%s));' % (n, n))
        for f in c._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (n, f[0]))
        lines.append('printf("\\n");')
    lines += ['return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = {ln.split()[0]: [int(v) for v in ln.split()[1:]] for ln in subprocess.check_output([str(exe)], text=True).splitlines()}
    for n, c in structs:
        want = [ctypes.sizeof(c)] + [getattr(c, f[0]).offset for f in c._fields_]
        assert got[n] == want, (n, got[n], want)


def test_library_allocates_nothing_and_never_reads_the_environment():
    """The header's contract (include/savp_hip.h: 'allocate no device memory ... never read the environment'): no hipMalloc /
    hipFree / getenv in the shipped sources (SAVP_CONV_ABLATE is a developer build that is not compiled into the library)."""
    csrc = os.path.join(ROOT, 'video_prediction_amd', 'csrc')
    bad = []
    for f in sorted(os.listdir(csrc)):
        if not (f.endswith('.hip') or f.endswith('.h')):
            continue
        src = open(os.path.join(csrc, f)).read()
        src = re.sub(r'#ifdef SAVP_CONV_ABLATE.*?#else', '', src, flags=re.S)
        for m in re.finditer(r'\b(hipMalloc\w*|hipFree\w*|getenv)\s*\(', src):
            bad.append((f, m.group(1)))
    assert not bad, bad


def test_no_memset_nodes_in_capturable_launch_sequences():
    """Every launch sequence of the library may be replayed from a hipGraph, and a hipMemsetAsync NODE of a replayed graph is not
    ordered with its neighbours on this ROCm build (csrc/zero_fill.h; tests/tools/ab_calls/graph_memset_probe.py: 29 of 30 replays
    wrong): buffers are cleared by savp_zero_async (a kernel) only."""
    csrc = os.path.join(ROOT, 'video_prediction_amd', 'csrc')
    bad = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith('.hip') or f.endswith('.h'):
            src = re.sub(r'//[^\n]*', '', open(os.path.join(csrc, f)).read())
            bad += [(f, m.group(1)) for m in re.finditer(r'\b(hipMemset\w*)\s*\(', src)]
    assert not bad, bad


def test_option_table_round_trip(hip_lib):
    """savp_set_option / savp_get_option: known names round-trip, unknown names are refused (host code, runs without a GPU)."""
    from video_prediction_amd import lib
    for name in lib.OPTION_NAMES:
        old = lib.get_option(name)
        lib.set_option(name, old + 5)
        assert lib.get_option(name) == old + 5
        lib.set_option(name, old)
    assert hip_lib.savp_set_option(b'no_such_option', 1) != 0
    assert lib.get_option('lstm_fused') == 1 and lib.get_option('thin') == 1 and lib.get_option('conv_ring') == 0


def test_conv_workspace_query_and_special_probe(hip_lib):
    """savp_conv_workspace_bytes / savp_conv_special are pure host-side predicates: the RGB-side weight gradient asks for its
    partial-sum rows, the generic weight gradient with a bias gradient for the column-sum scratch, a forward gate conv for nothing;
    the discriminators' first layer (3 -> 32 channels, 3x3x3) is a problem-specific kernel under tile 0 only."""
    from video_prediction_amd import lib
    a = lib.SavpConvArgs()
    a.mode = lib.CONV_WGRAD
    a.N, a.D, a.H, a.W, a.Cx = 32, 10, 64, 64, 3
    a.Do, a.Ho, a.Wo, a.Cy = 10, 64, 64, 32
    a.kd, a.kh, a.kw, a.sd, a.sh, a.sw, a.pd, a.ph, a.pw = 3, 3, 3, 1, 1, 1, 1, 1, 1
    a.precision = 1
    a.x, a.y, a.w = 0x10000, 0x20000, 0x30000
    a.x_sw, a.x_sh, a.x_sd, a.x_sn = 3, 64 * 3, 64 * 64 * 3, 10 * 64 * 64 * 3
    a.y_sw, a.y_sh, a.y_sd, a.y_sn = 32, 64 * 32, 64 * 64 * 32, 10 * 64 * 64 * 32
    need = hip_lib.savp_conv_workspace_bytes(ctypes.byref(a))
    assert need > 0 and need % 4 == 0
    assert hip_lib.savp_conv_special(ctypes.byref(a)) == 0              # no workspace handed over -> the general kernel would run
    a.ws, a.ws_bytes = 0x80000, need
    assert hip_lib.savp_conv_special(ctypes.byref(a)) == 1
    a.tile = 0x111                                                      # forced algorithm: the caller gets that kernel
    assert hip_lib.savp_conv_special(ctypes.byref(a)) == 0
    a.tile, a.mode = 0, lib.CONV_FPROP
    assert hip_lib.savp_conv_workspace_bytes(ctypes.byref(a)) == 0 and hip_lib.savp_conv_special(ctypes.byref(a)) == 1
    a.Cx, a.x_sw = 72, 72                                               # a gate-conv-like problem: nothing special, no scratch
    assert hip_lib.savp_conv_special(ctypes.byref(a)) == 0


def _conv2d_args(lib, mode, N, H, W, Cx, Ho, Wo, Cy, k, s, p, precision=1):
    a = lib.SavpConvArgs()
    a.mode = mode
    a.N, a.D, a.H, a.W, a.Cx = N, 1, H, W, Cx
    a.Do, a.Ho, a.Wo, a.Cy = 1, Ho, Wo, Cy
    a.kd, a.kh, a.kw, a.sd, a.sh, a.sw, a.pd, a.ph, a.pw = 1, k, k, 1, s, s, 0, p, p
    a.precision = precision
    a.x, a.y, a.w, a.w_bf16 = 0x100000, 0x200000, 0x300000, 0x400000
    a.x_sw, a.x_sh, a.x_sd, a.x_sn = Cx, W * Cx, H * W * Cx, H * W * Cx
    a.y_sw, a.y_sh, a.y_sd, a.y_sn = Cy, Wo * Cy, Ho * Wo * Cy, Ho * Wo * Cy
    return a


def test_conv_statistics_probe_is_a_host_side_predicate(hip_lib):
    """savp_conv_stats_ok (include/savp_hip.h): answers without a device -- yes for the generator's down / upsample convolutions of
    the bf16 datapath (whole 8-column tiles), no for the fp32 datapath, for planes that leave partial tiles, for an activation or an
    accumulating epilogue, and for the problems the RGB-side kernels take."""
    from video_prediction_amd import lib
    ok = lambda a: hip_lib.savp_conv_stats_ok(ctypes.byref(a))
    down = _conv2d_args(lib, lib.CONV_FPROP, 32, 32, 32, 40, 16, 16, 64, 4, 2, 1)              # conv_pool2d 3x3 folded to 4x4 stride 2
    assert ok(down) == 1
    up = _conv2d_args(lib, lib.CONV_DGRAD, 32, 32, 32, 32, 16, 16, 136, 6, 2, 2)               # upsample_conv2d as a DGRAD-mode launch
    assert ok(up) == 1
    down.precision = 0
    assert ok(down) == 0                                                                         # exact-fp32 datapath: no ring kernel
    down.precision, down.act = 1, lib.ACT_LRELU
    assert ok(down) == 0
    down.act, down.beta = 0, 1
    assert ok(down) == 0
    down.beta, down.w_bf16 = 0, None
    assert ok(down) == 0                                                                         # no packed bf16 weights
    ragged = _conv2d_args(lib, lib.CONV_FPROP, 2, 12, 12, 32, 12, 12, 64, 3, 1, 1)             # 12 columns: partial 8-column tiles
    assert ok(ragged) == 0
    rgb = _conv2d_args(lib, lib.CONV_FPROP, 32, 64, 64, 3, 64, 64, 32, 3, 1, 1)                # taken by conv_thin.hip under tile 0
    assert ok(rgb) == 0
    assert hip_lib.savp_conv_stats_ok(None) == 0


def test_allreduce_bucket_entry_point_validates_its_arguments(hip_lib):
    """savp_allreduce_bucket: exported, refuses a missing communicator / buffer, and an empty bucket is a no-op
    that does not even load RCCL.  (The collective itself needs >= 2 GPUs: the driver's scaling run.)"""
    assert hip_lib.savp_allreduce_bucket(None, None, 0x1000, 16) != 0
    assert hip_lib.savp_allreduce_bucket(0x1000, None, None, 16) != 0
    assert hip_lib.savp_allreduce_bucket(0x1000, None, 0x2000, 0) == 0


def test_conditioning_inputs_are_read_off_the_inputs_dict_and_pix_distribs_are_refused_not_ignored():
    """SAVPCell.call consumes inputs['actions'] / ['states'] (reference savp_model.py:413-421,655-658): the model is built for the widths
    the inputs dict carries (round 6; GPU parity: test_gpu_model.py::test_action_and_state_conditioned_cell_vs_oracle).  inputs['pix_distribs']
    (:408-410,647-653) is not on the HIP path: every entry that takes the dataset's inputs dict raises instead of running as if the key
    were absent."""
    import numpy as np
    import pytest
    from video_prediction_amd.models import get_model_class
    from video_prediction_amd.models import savp_model as M
    from video_prediction_amd import variables as V
    images = np.zeros((2, 4, 64, 64, 3), np.float32)
    model = get_model_class('savp')(mode='test', hparams_dict=dict(context_frames=2, sequence_length=4))
    inputs = {'images': images, 'pix_distribs': np.zeros((2, 4, 64, 64, 1), np.float32)}
    with pytest.raises(NotImplementedError, match='pix_distribs'):
        model.build_graph(inputs)
    with pytest.raises(NotImplementedError, match='pix_distribs'):
        M.generator_fn(inputs, 'test', model.hparams)
    with pytest.raises(NotImplementedError, match='pix_distribs'):
        M.SAVPEngine.set_images(None, inputs)
    M.refuse_conditioning_inputs({'images': images, 'pix_distribs': None})       # an absent / None entry is fine
    assert M.cond_of({'images': images}) == (0, 0)
    assert M.cond_of({'images': images, 'actions': np.zeros((2, 3, 4), np.float32), 'states': np.zeros((2, 4, 3), np.float32)}) == (4, 3)
    assert M.cond_of({'images': images, 'actions': None, 'states': np.zeros((2, 4, 5), np.float32)}) == (0, 5)
    # the variable table of the conditioned model: BAIR with use_state (4 actions, 3 states, nz = 8)
    specs = V.variable_specs(model.hparams, (64, 64, 3), mode='test', cond=(4, 3))
    assert specs['generator/rnn/savp_cell/h0/conv_pool2d/kernel'][0] == (5, 5, 6 + 15, 32)
    assert specs['generator/rnn/savp_cell/lstm_h0/basic_conv2dlstm_cell/kernel'][0] == (5, 5, 32 + 15 + 32, 128)
    assert specs['generator/encoder/layer_1/conv2d/kernel'][0] == (4, 4, 6 + 4, 64)
    assert specs['generator/rnn/savp_cell/state_pred/dense/kernel'][0] == (7, 3)
    # state_weight without states: the reference reads inputs['states'] (base_model.py:758-760)
    hp = get_model_class('savp')(mode='train', hparams_dict=dict(context_frames=2, sequence_length=4, state_weight=1.0)).hparams
    with pytest.raises(KeyError):
        M.SAVPEngine(hp, (64, 64, 3), 2, mode='train', device='cpu')


def test_fused_operator_entries_refuse_halves_that_do_not_fit(hip_lib):
    """csrc/fused_ops.hip (one entry per fused op): the hand-overs between the two launches of a fused operator are
    checked by the library -- a convolution that does not write what the per-sample pass reads, or a statistics buffer only one half
    knows about, is SAVP_EINVAL before anything is launched (no GPU needed)."""
    import ctypes
    from video_prediction_amd import lib
    c = lib.SavpConvLstmCellArgs()
    c.conv.mode = lib.CONV_DGRAD                                   # forward wants an FPROP
    assert hip_lib.savp_convlstm_cell_fwd(None, ctypes.byref(c)) == -1
    c.conv.mode = lib.CONV_FPROP
    c.conv.y, c.gates.gates = 0x1000, 0x2000                        # the gate block would not read the conv's output
    assert hip_lib.savp_convlstm_cell_fwd(None, ctypes.byref(c)) == -1
    c.gates.gates = 0x1000
    c.conv.Cy, c.gates.F, c.conv.N, c.gates.N, c.conv.Do, c.conv.Ho, c.conv.Wo, c.gates.HW = 128, 32, 2, 2, 1, 8, 8, 64
    c.conv.stats = 0x3000                                           # statistics written, but the gate block not told
    assert hip_lib.savp_convlstm_cell_fwd(None, ctypes.byref(c)) == -1
    c.gates.stats1_ready, c.gates.ws_stats = 1, 0x4000              # ... or told about another buffer
    assert hip_lib.savp_convlstm_cell_fwd(None, ctypes.byref(c)) == -1
    assert hip_lib.savp_convlstm_cell_bwd(None, ctypes.byref(c)) == -1          # backward wants the DGRAD
    n = lib.SavpConvNormArgs()
    n.conv.mode = lib.CONV_WGRAD
    assert hip_lib.savp_conv_in_act_fwd(None, ctypes.byref(n)) == -1 and hip_lib.savp_conv_in_act_bwd(None, ctypes.byref(n)) == -1
    n.conv.mode, n.conv.y, n.norm.x.p = lib.CONV_FPROP, 0x1000, 0x2000
    assert hip_lib.savp_conv_in_act_fwd(None, ctypes.byref(n)) == -1
    assert hip_lib.savp_convlstm_cell_fwd(None, None) == -1 and hip_lib.savp_conv_in_act_bwd(None, None) == -1


def test_zero_arena_offsets_after_a_replay():
    """kernels.ZeroArena on the host side (CPU tensors: no launch involved): fresh slices are disjoint and zero, the high-water mark
    follows the takes, a full arena wraps through reset(), and replayed(mark) puts eager takes behind a captured sequence's slices
    whatever the host offset said (models/savp_model._StepProgram.run)."""
    import torch
    from video_prediction_amd import kernels as K
    a = K.ZeroArena(torch.device('cpu'), floats=1024)
    s1, s2 = a.take(100), a.take(1)
    assert s1.numel() == 128 and s2.numel() == 64 and a.off == 192 and a.hi == 192          # rounded up to 64 floats
    assert s1.data_ptr() + 128 * 4 == s2.data_ptr()
    s1.fill_(3.0)
    a.hi = 0                                  # what a capture does before it runs the body
    a.reset()
    assert a.off == 0 and float(a.buf.abs().sum()) == 0.0
    a.take(300)
    a.take(200)
    mark = a.hi
    assert mark == 320 + 256 and a.off == mark
    # an eager caller runs on, wraps (reset: memset + rewind) and leaves a small offset behind ...
    for _ in range(3):
        a.take(256).fill_(1.0)
    assert a.off < mark
    # ... then the captured sequence is replayed: device memory is zero from `mark` on and used below it
    a.buf.zero_()
    a.buf[:mark].fill_(7.0)
    a.replayed(mark)
    nxt = a.take(64)
    assert a.off == mark + 64 and float(nxt.abs().sum()) == 0.0
    with pytest.raises(ValueError):
        a.take(4096)


def test_split_k_scratch_query_is_a_host_side_predicate(hip_lib):
    """Deterministic split-K (include/savp_hip.h SavpConvArgs.ws): a FPROP / DGRAD call stores every split's share in its own slice of the
    caller's scratch; savp_conv_workspace_bytes says how much -- exactly splitk slices of the dense destination block for an explicit
    split count, an upper bound of the heuristic's choice for the automatic one, nothing for an unsplit call or a large problem."""
    from video_prediction_amd import lib
    a = _conv2d_args(lib, lib.CONV_DGRAD, 32, 8, 8, 264, 8, 8, 512, 5, 1, 2)      # the 8x8 gate convolution's data gradient
    block = 32 * 8 * 8 * 264 * 4
    a.splitk = 4
    assert hip_lib.savp_conv_workspace_bytes(ctypes.byref(a)) == 4 * block
    a.splitk = 1
    assert hip_lib.savp_conv_workspace_bytes(ctypes.byref(a)) == 0
    a.splitk = 0
    auto = hip_lib.savp_conv_workspace_bytes(ctypes.byref(a))
    assert auto % block == 0 and 2 * block <= auto <= 16 * block
    a.dst_gap_at, a.dst_gap, a.Cx, a.splitk = 128, 8, 256, 2                      # gapped destination: the block spans the physical channels
    assert hip_lib.savp_conv_workspace_bytes(ctypes.byref(a)) == 2 * block
    big = _conv2d_args(lib, lib.CONV_FPROP, 32, 64, 64, 32, 64, 64, 32, 3, 1, 1)   # 131 072 output rows: never split automatically
    assert hip_lib.savp_conv_workspace_bytes(ctypes.byref(big)) == 0


def test_weight_gradient_scratch_query_is_the_launchers_own_plan(hip_lib):
    """Deterministic weight gradients (round 6; include/savp_hip.h SavpConvArgs.ws): every pixel split of a WGRAD launch leaves its dW tiles in
    its own slice of the caller's scratch and a fold adds the slices in split order.  savp_conv_workspace_bytes asks the SAME planner the
    launcher runs (conv_wgrad_patch_try(plan) / wgrad_generic_plan): whole slices of taps * Cx * Cy floats, one round of 256 workgroups for
    the LDS-patch kernel, never an empty split, a bias gradient's rows on top."""
    from video_prediction_amd import lib
    # the 32x32 ConvLSTM gate convolution's weight gradient over one timestep of both unrolls: 5x5, 72 -> 128, both operands bf16
    a = _conv2d_args(lib, lib.CONV_WGRAD, 32, 32, 32, 72, 32, 32, 128, 5, 1, 2)
    a.src_bf16 = a.out_bf16 = 1
    nW = 25 * 72 * 128 * 4
    need = hip_lib.savp_conv_workspace_bytes(ctypes.byref(a))
    assert need % nW == 0 and 2 <= need // nW <= 256
    S = need // nW
    tiles = 32 * 4 * 4                                            # 8x8 pixel tiles: N x (32 / 8)^2
    per = -(-tiles // S)
    assert -(-tiles // per) == S                                  # no split is left without tiles
    # fp32 operands + a bias gradient: each split also keeps one row of Cy floats per wave
    b = _conv2d_args(lib, lib.CONV_WGRAD, 4, 16, 16, 32, 16, 16, 64, 3, 1, 1)
    b.bias = 0x500000
    need_b = hip_lib.savp_conv_workspace_bytes(ctypes.byref(b))
    row = 9 * 32 * 64
    assert need_b >= 2 * (row + 4 * 64) * 4                      # (at least the column-sum pass's workspace of the generic path: the larger of the two is reported)
    # the generic kernel (forced: tile bits 8-9 = 1; exact-fp32 datapath): splitk slices of the im2col GEMM's M x Cy block
    c = _conv2d_args(lib, lib.CONV_WGRAD, 8, 16, 16, 32, 16, 16, 64, 3, 1, 1, precision=0)
    c.tile = 0x122
    need_c = hip_lib.savp_conv_workspace_bytes(ctypes.byref(c))
    assert need_c % (row * 4) == 0 and need_c // (row * 4) >= 2
    c.splitk = 1
    assert hip_lib.savp_conv_workspace_bytes(ctypes.byref(c)) == 0


def test_thin_head_routing_is_a_host_side_predicate(hip_lib):
    """savp_conv_special (include/savp_hip.h): which calls the problem-specific kernels of csrc/conv_thin.hip take under tile 0 -- round 5's
    wide -> thin FPROP (scratch-image head 32 -> 4, mask convolution 56 -> 8) and the mask convolution's data gradient (56 <- 8), and the
    conditions that send a call to the general kernels instead."""
    from video_prediction_amd import lib
    hip_lib.savp_conv_special.argtypes = [ctypes.c_void_p]
    sp = lambda a: hip_lib.savp_conv_special(ctypes.byref(a))
    head = _conv2d_args(lib, lib.CONV_FPROP, 32, 64, 64, 32, 64, 64, 4, 3, 1, 1)
    head.act, head.y_sw = lib.ACT_SIGMOID, 56                                  # into a channel slice of the mask convolution's input
    masks = _conv2d_args(lib, lib.CONV_FPROP, 32, 64, 64, 56, 64, 64, 8, 3, 1, 1)
    assert sp(head) == 1 and sp(masks) == 1
    dg = _conv2d_args(lib, lib.CONV_DGRAD, 32, 64, 64, 56, 64, 64, 8, 3, 1, 1)
    dg.beta = 1
    assert sp(dg) == 1
    for field, value in (('Cx', 44), ('Cy', 9), ('precision', 0), ('w_bf16', None), ('src_bf16', 1), ('beta', 1), ('sh', 2)):
        a = _conv2d_args(lib, lib.CONV_FPROP, 32, 64, 64, 56, 64, 64, 8, 3, 1, 1)
        setattr(a, field, value)
        assert sp(a) == 0, field                                                # Cx % 8, Cy <= 8, bf16 precision, packed bf16 weights, fp32 tensors, no accumulation, stride 1
    dg.Cy = 4                                                                    # the 8-channel kernel is for Cy == 8 exactly ...
    dg.Cx = 40
    assert sp(dg) == 0
    dg.Cx, dg.beta = 32, 0                                                       # ... 32 <- 4 is the RGB-side kernel's problem (round 2)
    assert sp(dg) == 1


def test_bench_dry_run_prints_the_eight_rank_launch_plan():
    """`bench.py --gpus 8 --dry-run` on a box without a GPU: the launcher line the script becomes (= the driver's torch.distributed.run
    line), one core slice per rank, the weak-scaling workload."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '5', '--warmup', '2', '--dry-run'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d['n_gpus'] == 8 and d['global_batch'] == 128 and d['per_gpu_batch'] == 16 and d['scaling'] == 'weak' and d['backend'] == 'nccl'
    L = d['launcher']
    assert L[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and L[L.index('--nproc-per-node') + 1] == '8'
    assert L[L.index('--master-addr') + 1] == '127.0.0.1' and '--dry-run' not in L
    assert d['binding'] == 'none' or len(d['binding']) == 8
