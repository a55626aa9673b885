#!/usr/bin/env python
"""bench.py -- SAVP training throughput on MI355X (BASELINE.json metric: train frames/sec, BAIR 64x64 seq30 SAVP).

One "step" = one full reference train step (D Adam update, then G+E Adam update against the updated D:
/root/reference/video_prediction/models/base_model.py:486-510) on a synthetic BAIR-shaped batch already resident in
HBM.  Workload at every N: configs[1] of BASELINE.json per GPU (full VAE-GAN, action-free BAIR 64x64x3, seq 30,
batch 16 per GPU, published ours_savp recipe) -> weak scaling; one process per GPU, gradients all-reduced by RCCL.

Timed region: W untimed + exactly K timed steps between (barrier + synchronize) pairs, MAX over ranks; un-instrumented; on one
GPU the step body (no host input) is replayed as a captured hipGraph (--eager: launch by launch); with replicas it is replayed as
hipGraph SEGMENTS with the collectives issued by the host between them (models/savp_model.py:_StepProgram).  Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- the dominant kernel family (LDS-patch / implicit-GEMM conv on the MFMA pipe): 8 further EAGER steps after the
                  timed region carry HIP events / dispatch stamps around the five ConvLSTM gate-conv FPROP launches; algorithmic
                  FLOPs per launch from SURVEY.md 8(d) (2*M*N*K of each layer) / measured duration, against the dense MFMA peak
                  of the datapath in use (bf16: 2.5 PFLOP/s; --precision f32: 157.3 TFLOP/s).  `traffic` is read from the
                  committed rocprofv3 PMC pass of the round (profiles/), not collected by this run.
  roofline_cell-- the fused cell (gate conv + ONE gate-block launch) against both roofs, same instrumented steps.
  roofline_step-- the whole train step: SURVEY.md 8(d)'s algorithmic TFLOP per sequence x sequences / measured time, against the same peak.
  roofline_cell-- also `kernel_only`: the same cells on the two launches' own begin / end stamps (gate conv + gate block).
  kernel_families_ms -- kernel time per step by family, quoted from profiles/r05_kernel_families.json (like `roofline.traffic` from the
                  PMC file) ONLY when that file's source id equals this checkout's (video_prediction_amd.lib.source_id).
  config       -- besides the workload: `submission` (hipGraph replay / ... in N segments with replicas / eager launches), `eager_ms_per_step`
                  (the same step launch by launch), `host_issue_ms_per_step` (host time to ISSUE one eager step with an idle GPU) and, with a
                  process group, `dist` (backend, world, chunks issued, side stream; SAVP_FORCE_DIST=1 keeps the collectives at world size 1).
  f32          -- (N=1) the exact-fp32 datapath, the reference's own arithmetic, on the same workload, timed the same way.
  cpu_baseline -- the CPU oracle (a torch-CPU restatement of the reference step, kind "port") timed on this host's
                  cores on a bounded sample (one sequence), rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on these hosts: RCCL / tensor sharing across processes needs it

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

RECIPE = dict(  # hparams/bair_action_free/ours_savp/model_hparams.json of the reference
    batch_size=16, lr=0.0002, beta1=0.5, beta2=0.999, l1_weight=100.0, l2_weight=0.0, kl_weight=1.0,
    video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1, vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0,
    state_weight=0.0)
# Workloads (BASELINE.json configs / SURVEY.md 8(d)); the headline metric is quoted on c2, the default.
#   c2: configs[1] -- BAIR action-free 64x64x3, seq 30, context 2 (softmotion_dataset.py:46-54), batch 16 per GPU
#   c4: configs[3] -- KTH 64x64x1, seq 40, context 10 (kth_dataset.py:26-36), nz 32 / kl 0.01 (hparams/kth/ours_savp), batch 16 per GPU
#   c5: configs[4] -- synthetic 128x128x3, seq 30, context 2, batch 8 per GPU (the >= 128 layer table, savp_model.py:198-210)
CONFIGS = {
    # c1: configs[0] -- the reference's CPU-runnable plumbing case (ours_deterministic_l1: nz=0, l1 only), BAIR, context 2 + 10, batch 4
    'c1': dict(name='BAIR action-free 64x64x3 (deterministic)', shape=(64, 64, 3), seq=12, context=2, batch=4,
               over=dict(nz=0, lr=0.001, beta1=0.9, l1_weight=1.0, kl_weight=0.0, video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                         vae_gan_feature_cdist_weight=0.0)),
    'c2': dict(name='BAIR action-free 64x64x3', shape=(64, 64, 3), seq=30, context=2, batch=16, over={}),
    'c4': dict(name='KTH 64x64x1', shape=(64, 64, 1), seq=40, context=10, batch=16, over=dict(nz=32, kl_weight=0.01)),
    'c5': dict(name='synthetic 128x128x3', shape=(128, 128, 3), seq=30, context=2, batch=8, over={}),
}
H, W, C = CONFIGS['c2']['shape']
SEQ, CONTEXT = CONFIGS['c2']['seq'], CONFIGS['c2']['context']
CPU_SEQ = 12              # frames of the cpu_baseline sample (>= clip_length + 1 = 11 for the video discriminator)
PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}  # MI355X_MICROARCH.md: fp32 MFMA / vector peak; dense bf16 MFMA peak


def make_hparams(batch, seq=SEQ, context=CONTEXT, over=None):
    from video_prediction_amd.models import get_model_class
    d = dict(RECIPE)
    d.update(over or {})
    d.update(context_frames=context, sequence_length=seq, batch_size=batch)
    model = get_model_class('savp')(mode='train', hparams_dict=d)
    return model


def synthetic_batch(batch, seed, device, seq=SEQ, shape=(H, W, C)):
    """Seeded synthetic video of the workload's shape, uniform[0,1) like convert_image_dtype'd uint8 frames."""
    rng = np.random.default_rng(seed)
    x = rng.random((batch, seq) + tuple(shape), dtype=np.float32)
    return torch.from_numpy(x).to(device)


def convlstm_flops(engine):
    """Algorithmic FLOPs (2*M*N*K) of the ConvLSTM gate convolutions instrumented below, per launch list."""
    out = []
    for L in engine.gen.layers:
        if L['rnn']:
            h, w = L['hw']
            a = L['a'].v
            cin, f = a.shape[-1], L['f']
            out.append((L['rconv'], 2.0 * engine.N * h * w * (4 * f) * (25 * cin)))
    return out


def cpu_baseline(seconds_budget=30.0):
    """Time the CPU oracle (kind 'port': not TensorFlow, a torch-CPU restatement; see oracle/__init__.py) on a bounded
    sample of the same workload: one full train step, fp32, B=1, the first CPU_SEQ frames of a sequence (the per-frame
    cost of the unrolled model is constant, so frames/s carries over; a full 30-frame step takes ~1 min here)."""
    from oracle import train as OT
    from video_prediction_amd import variables as V
    SEQ = CPU_SEQ
    model = make_hparams(1, SEQ)
    hp = model.hparams
    specs = V.variable_specs(hp, (H, W, C), mode='train')
    vals = V.init_variables(specs, seed=4)
    P = {k: torch.tensor(v) for k, v in vals.items()}
    st = OT.init_opt_state(P)
    rng = np.random.default_rng(0)
    images = torch.tensor(rng.random((SEQ, 1, H, W, C), dtype=np.float32))
    T1 = SEQ - 1

    def noise(seed):
        r = np.random.default_rng(seed)
        n = {'eps': torch.tensor(r.standard_normal((T1, 1, hp.nz)).astype(np.float32)),
             'prior': torch.tensor(r.standard_normal((SEQ - CONTEXT, 1, hp.nz)).astype(np.float32))}
        for ph in ('pre', 'post'):
            n['d_indices_' + ph] = {k: (r.integers(0, T1, 1), r.integers(0, T1 - hp.clip_length + 1, 1))
                                    for k in ('enc_real', 'enc_fake', 'real', 'fake')}
        return n
    # The oracle is thousands of small torch-CPU ops: on the GPU box's 128 hardware threads the default thread pool spends its time
    # waking workers (0.16-0.5 frames/s measured) while 8 threads of this container reach 3.3 frames/s on the same step
    # (profiles/r02_cpu_baseline_full_T30.json).  16 threads is the honest operating point; `cores` reports what was used.
    prev_threads = torch.get_num_threads()
    threads = min(prev_threads, 16)
    torch.set_num_threads(threads)
    t0 = time.time()
    nsteps = 0
    while True:
        n = noise(nsteps)
        P, st, _ = OT.train_step(P, st, {'images': images}, hp, n, n['d_indices_pre'], n['d_indices_post'], step=nsteps)
        nsteps += 1
        el = time.time() - t0
        if el > seconds_budget * 0.4 or nsteps >= 3:
            break
    el = time.time() - t0
    torch.set_num_threads(prev_threads)
    return {'value': nsteps * SEQ / el, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': '%d full train step(s) of 1 sequence (B=1, T=%d, 64x64x3, fp32 torch-CPU oracle), %.1f s' % (nsteps, SEQ, el)}


def pin_rank(local_rank, local_world):
    """One process per GPU: give rank r the r-th contiguous slice of the cores this job may use (sched_setaffinity) and size torch's
    host thread pool to it.  Eight Python interpreters that each issue ~2.5 k launches (or 8 graph segments + 7 collectives) per step
    otherwise migrate across both sockets and share cores with each other's RCCL proxy threads.  Contiguous slices follow the usual
    MI300-class node layout (GPUs 0-3 on socket 0, 4-7 on socket 1; cores numbered socket by socket); SAVP_PIN=0 leaves the
    scheduler alone.  Returns a description for the bench line (config.dist.binding)."""
    if os.environ.get('SAVP_PIN', '1') != '1' or not hasattr(os, 'sched_setaffinity'):
        return 'none'
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // max(local_world, 1)
        if per < 1:
            return 'none (%d cores for %d ranks)' % (len(cpus), local_world)
        mine = cpus[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, 8)))
        desc = 'rank %d -> cores %d-%d (%d of %d), %d torch threads' % (local_rank, mine[0], mine[-1], per, len(cpus), torch.get_num_threads())
        print('bench.py: ' + desc, file=sys.stderr)
        return desc
    except OSError as ex:
        return 'failed: %r' % (ex,)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=170, help='timed steps (default: a timed region of >= 10 s on the c2 workload)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='c2', help='workload (see CONFIGS); the headline metric is c2')
    ap.add_argument('--batch', type=int, default=0, help='per-GPU batch (default: the workload\'s: 16 for c2 / c4, 8 for c5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=('bf16', 'f32'), default='bf16',
                    help='conv multiply precision: bf16 operands / fp32 accumulate (BASELINE configs[1]) or exact fp32')
    ap.add_argument('--no-autotune', action='store_true')
    ap.add_argument('--eager', action='store_true', help='submit the timed steps launch by launch instead of replaying a captured hipGraph')
    ap.add_argument('--graph', action='store_true', help='(default on one GPU; kept for older command lines)')
    ap.add_argument('--inst-steps', type=int, default=8, help='instrumented eager steps after the timed region (0: none, roofline objects empty)')
    ap.add_argument('--no-f32', action='store_true', help='skip the second object: the exact-fp32 datapath on the same workload')
    ap.add_argument('--no-workloads', action='store_true', help='skip the `workloads` object: the other single-GPU configs (c1, c4, c5), 20 replayed steps each')
    ap.add_argument('--tuning-table', default=None, help='developer: a tuning table other than the shipped one (A/B of re-tuned entries)')
    ap.add_argument('--retune', action='store_true', help='ignore the shipped tuning table and time every conv problem again')
    ap.add_argument('--save-tuning', default=None, help='write the tuning table found during this run to this path')
    ap.add_argument('--dry-run', action='store_true', help='print the launch plan (launcher command line, per-rank core binding, workload) as JSON and exit; needs no GPU')
    args = ap.parse_args()

    if args.dry_run:
        cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
        per = len(cpus) // max(args.gpus, 1)
        cfg = CONFIGS[args.config]
        plan = {'n_gpus': args.gpus, 'workload': args.config, 'per_gpu_batch': args.batch or cfg['batch'], 'global_batch': (args.batch or cfg['batch']) * args.gpus,
                'steps': args.steps, 'warmup': args.warmup, 'scaling': 'weak', 'backend': os.environ.get('SAVP_DIST_BACKEND', 'nccl'),
                'launcher': ([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
                              '--master-port', '<free port>', os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--dry-run']) if args.gpus > 1 else None,
                'binding': ['rank %d -> cores %d-%d' % (r, cpus[r * per], cpus[(r + 1) * per - 1]) for r in range(args.gpus)] if (args.gpus > 1 and per >= 1) else 'none'}
        print(json.dumps(plan))
        return

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the SAVP hot path has no CPU fallback')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run (the driver's own
        # launch line), rank 0 prints the JSON line
        import socket
        import subprocess
        if args.gpus > torch.cuda.device_count() and os.environ.get('SAVP_DIST_BACKEND', 'nccl') == 'nccl':
            raise SystemExit('bench.py --gpus %d: this node has %d GPU(s)' % (args.gpus, torch.cuda.device_count()))
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and rank == 0:
        print('bench.py: --gpus %d but launched with WORLD_SIZE=%d; reporting n_gpus=%d' % (args.gpus, world, world), file=sys.stderr)
    # SAVP_DIST_BACKEND=gloo: several ranks may share one GPU (tests/test_gpu_dp.py runs the whole multi-rank path of this script
    # on a one-GPU box: launcher, rendezvous, tuning broadcast, chunked exchange, MAX-over-ranks clock, one JSON line)
    backend = os.environ.get('SAVP_DIST_BACKEND', 'nccl')
    binding = pin_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world))) if world > 1 else None
    dev_index = local_rank % torch.cuda.device_count() if backend != 'nccl' else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    # SAVP_FORCE_DIST=1: keep the process group and every collective of the step at world size 1 -- RCCL, the side-stream exchange
    # and the segmented replay run on a one-GPU box exactly as a rank of the 8-GPU job runs them (tests/test_gpu_dp.py)
    force_dist = os.environ.get('SAVP_FORCE_DIST', '0') == '1'
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist_mod.init_process_group(backend=backend, rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
        dist = dist_mod

    from video_prediction_amd import kernels as K
    from video_prediction_amd.models.savp_model import SAVPEngine
    cfg = CONFIGS[args.config]
    if not args.batch:
        args.batch = cfg['batch']
    shape, seq = cfg['shape'], cfg['seq']

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def build_engine(precision, cfg=cfg, batch=None, shape=shape, seq=seq):
        batch = batch or args.batch
        K.set_conv_precision(precision)
        K.enable_autotune(not args.no_autotune)      # tile / split-K selection happens during the warm-up steps
        # shipped table of the BASELINE workload (measured on MI355X by an earlier run of this script with --save-tuning):
        # problems found in it are not timed again, anything else is tuned live during the warm-up
        table = args.tuning_table or os.path.join(ROOT, 'video_prediction_amd', 'tuning_gfx950_%s.json' % precision)
        if not args.no_autotune and not args.retune and os.path.exists(table):
            K.load_tuning(table)
        model = make_hparams(batch, seq, cfg['context'], cfg['over'])
        eng = SAVPEngine(model.hparams, shape, batch, mode='train', seed=4, device=str(device))
        if dist is not None:
            eng.attach_process_group(dist)
        eng.set_images(synthetic_batch(batch, 1234 + rank, device, seq, shape))      # inputs resident in HBM before timing
        return eng, model.hparams

    def timed_steps(eng, warmup, steps):
        """W untimed steps, then EXACTLY K steps between (barrier + synchronize) pairs; returns (seconds = MAX over ranks, last info)."""
        info = None
        for _ in range(warmup):
            info = eng.train_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            info = eng.train_step()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, info

    engine, hp = build_engine(args.precision)
    # The timed region runs the step the way a training job does: un-instrumented, and on one GPU as a replayed hipGraph (the step
    # body has no host input; --eager keeps per-launch submission; with replicas over RCCL: graph segments between the
    # collectives; over gloo, whose collectives block the host, the engine keeps launch-by-launch submission).  The roofline
    # numbers come from INST_STEPS further steps AFTER the timed region, eager, with HIP events / dispatch stamps around the
    # ConvLSTM gate-conv launches and around whole cells -- instrumentation never sits inside the timed region.
    engine.use_graph = engine.use_graph and not args.eager     # the engine's own choice stands (eager under a host-blocking backend)
    dt, info = timed_steps(engine, args.warmup, args.steps)
    mode = 'eager launches'
    if engine.use_graph and engine.graph is not None:
        nseg = engine.graph.segments
        mode = 'hipGraph replay' if nseg == 1 else 'hipGraph replay in %d segments, collectives issued between them' % nseg
        if nseg == 1 and getattr(engine, 'graph_collectives', False):
            mode = 'hipGraph replay, collectives captured in the graph (SAVP_GRAPH_COLLECTIVES=1)'
    eager_ms = None
    host_issue_ms = None
    INST_STEPS = max(0, args.inst_steps)
    engine.use_graph = False
    inst = convlstm_flops(engine)
    cells = [L for L in engine.gen.layers if L['rnn']]
    prof_lists = {id(layer): [] for layer, _ in inst}
    ktimers = {id(layer): K.KernelTimer() for layer, _ in inst}       # kernel-only durations of the same launches
    cell_lists = {id(L): [] for L in cells}
    gate_timers = {id(L): K.KernelTimer() for L in cells}              # the gate-block launch of every cell, kernel-only clock
    for layer, _ in inst:
        layer.prof, layer.ktimer = prof_lists[id(layer)], ktimers[id(layer)]
    for L in cells:
        L['cell_prof'] = cell_lists[id(L)]
        L['gate_ktimer'] = gate_timers[id(L)]
    sync()
    t1 = time.perf_counter()
    for _ in range(INST_STEPS):
        engine.train_step()
    sync()
    inst_ms = ((time.perf_counter() - t1) / INST_STEPS * 1e3) if INST_STEPS else None
    if mode != 'eager launches' and INST_STEPS:                     # the same step submitted launch by launch, un-instrumented, for comparison
        for layer, _ in inst:
            layer.prof = layer.ktimer = None
        for L in cells:
            L['cell_prof'] = None
            L['gate_ktimer'] = None
        e_dt, _ = timed_steps(engine, 1, min(args.steps, 10))
        eager_ms = e_dt / min(args.steps, 10) * 1e3
        # host time to ISSUE one eager step (GPU idle at the start of each measurement, no sync before the clock stops): the head-room
        # of launch-by-launch submission -- what a replica of a multi-GPU run, which cannot replay a graph around its collectives, needs
        acc = 0.0
        for _ in range(3):
            torch.cuda.synchronize()
            th = time.perf_counter()
            engine.train_step()
            acc += time.perf_counter() - th
        torch.cuda.synchronize()
        host_issue_ms = acc / 3 * 1e3
        for layer, _ in inst:
            layer.prof = prof_lists[id(layer)]
    # roofline of the dominant kernel family from the live event pairs
    # Two clocks on the same launches: (1) the dispatch's own begin / end stamps (hipExtLaunchKernelGGL
    # events handed over with savp_prof_arm: the kernel alone, the duration rocprofv3's kernel trace reports) -- this is
    # `achieved`; (2) hipEventRecord markers in front of / behind the launch, which also hold the command processor's
    # hand-over between packets -- reported beside it as avg_launch_us_with_gaps.  Kernels that do not take the pair (the
    # fp32 datapath) fall back to (2).
    tot_flops, tot_s, launches = 0.0, 0.0, 0
    k_flops, k_s, k_launches = 0.0, 0.0, 0
    conv_us_of = {}                               # gate conv layer -> its kernel-only durations (for the cell's kernel-only clock)
    per_layer = {}                                # variable scope of the layer -> its mean kernel-only duration / fraction of the peak
    gate_native = 0                               # how many of the timed layers ran the gate convolution's own kernel (csrc/conv_gate.hip)
    for layer, fl in inst:
        for e0, e1 in prof_lists[id(layer)]:
            tot_s += e0.elapsed_time(e1) * 1e-3
            tot_flops += fl
            launches += 1
        conv_us_of[id(layer)] = ktimers[id(layer)].durations_us()
        for us in conv_us_of[id(layer)]:
            k_s += us * 1e-6
            k_flops += fl
            k_launches += 1
        if conv_us_of[id(layer)]:
            mean_us = sum(conv_us_of[id(layer)]) / len(conv_us_of[id(layer)])
            from video_prediction_amd import lib as _lib0
            native = bool(getattr(layer, 'wfrag', None) is not None and args.precision == 'bf16' and _lib0.get_option('gate_kernel'))
            gate_native += int(native)
            per_layer[layer.kernel_name.split('/')[-3] if layer.kernel_name.count('/') >= 3 else layer.kernel_name] = {
                'avg_us': mean_us, 'gflop': fl / 1e9, 'frac': fl / mean_us / 1e6 / PEAK_TFLOPS[args.precision],
                'kernel': 'conv_gate_kernel' if native else ('conv_ring_kernel' if args.precision == 'bf16' else 'conv_fd_kernel')}
        ktimers[id(layer)].close()
        layer.prof = None
        layer.ktimer = None
    gaps_us = (tot_s / launches * 1e6) if launches else None
    clock = 'dispatch begin/end stamps (kernel alone)'
    if k_launches == 0:
        k_flops, k_s, k_launches, clock = tot_flops, tot_s, launches, 'event markers around the launch (includes dispatch gaps)'
    achieved = k_flops / k_s / 1e12 if k_s > 0 else None
    tot_s, launches = k_s, k_launches
    # the fused ConvLSTM cell as a unit (gate conv with statistics epilogue + the one-launch gate block): HIP events around the whole cell.
    # Algorithmic bytes (read x, h, c and W once, write c', h'; fp32 activations, bf16
    # weights in bf16 mode) and FLOPs (2*M*N*K of the gate conv) per cell launch-set, against both roofs.
    cell_s, cell_flops, cell_bytes, cell_n = 0.0, 0.0, 0.0, 0
    ck_s, ck_flops, ck_bytes, ck_n = 0.0, 0.0, 0.0, 0          # the same cells on the kernel-only clock: gate conv + gate block durations
    two_launch_gate_us = {}
    for L in cells:
        h_, w_ = L['hw']
        cin, f = L['a'].v.shape[-1], L['f']
        wbytes = 25 * cin * 4 * f * (2 if args.precision == 'bf16' else 4)
        ab = L['a'].v.element_size()                                  # the cell input [x | z | h] is bf16 on the bf16 datapath
        abytes = engine.N * h_ * w_ * (cin * ab + f * 4 + f * 4 + f * ab)   # x|z|h + c read, c' + h' written
        fl = 2.0 * engine.N * h_ * w_ * (4 * f) * (25 * cin)
        for e0, e1 in cell_lists[id(L)]:
            cell_s += e0.elapsed_time(e1) * 1e-3
            cell_flops += fl
            cell_bytes += wbytes + abytes
            cell_n += 1
        g_us = gate_timers[id(L)].durations_us()
        two_launch_gate_us[id(L)] = g_us
        c_us = conv_us_of.get(id(L['rconv']), [])
        if g_us and len(g_us) == len(c_us):
            ck_s += (sum(g_us) + sum(c_us)) * 1e-6
            ck_flops += fl * len(g_us)
            ck_bytes += (wbytes + abytes) * len(g_us)
            ck_n += len(g_us)
        gate_timers[id(L)].close()
        L['cell_prof'] = None
        L['gate_ktimer'] = None
    # ... and the cell the way the timed region runs it: where the gate convolution's tile holds whole images (16 x 16, 8 x 8) the whole cell
    # forward is ONE kernel (csrc/conv_gate.hip, CELL instantiations); a second set of instrumented steps times that kernel by its own dispatch
    # stamps, the other layers as gate conv + gate block as above.
    one_s, one_flops, one_n, one_layers = 0.0, 0.0, 0, 0
    if INST_STEPS and args.precision == 'bf16':
        fused_layers = [L for L in cells if getattr(L['rconv'], 'wfrag_il', None) is not None]
        if fused_layers:
            ct = {id(L): K.KernelTimer() for L in fused_layers}
            for L in fused_layers:
                L['cell_ktimer'] = ct[id(L)]
            sync()
            for _ in range(max(1, INST_STEPS // 2)):
                engine.train_step()
            sync()
            for L in cells:
                h_, w_ = L['hw']
                cin, f = L['a'].v.shape[-1], L['f']
                fl = 2.0 * engine.N * h_ * w_ * (4 * f) * (25 * cin)
                if id(L) in ct:
                    us = ct[id(L)].durations_us()
                    ct[id(L)].close()
                    L['cell_ktimer'] = None
                    if us:
                        one_s += sum(us) * 1e-6; one_flops += fl * len(us); one_n += len(us); one_layers += 1
            # two-launch layers enter with the mean of their (gate conv + gate block) pairs measured above
            two = [L for L in cells if id(L) not in ct]
            for L in two:
                c_us = conv_us_of.get(id(L['rconv']), [])
                g_us = two_launch_gate_us.get(id(L), [])
                if c_us and len(c_us) == len(g_us):
                    h_, w_ = L['hw']
                    cin, f = L['a'].v.shape[-1], L['f']
                    fl = 2.0 * engine.N * h_ * w_ * (4 * f) * (25 * cin)
                    k = len(c_us)
                    one_s += (sum(c_us) + sum(g_us)) * 1e-6; one_flops += fl * k; one_n += k
    # HBM traffic of the same kernel / shapes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (FETCH_SIZE x2 on gfx950 per
    # MI355X_MICROARCH.md).  bench.py cannot collect counters itself: the number is read from the committed PMC pass of the
    # round (profiles/, written by tests/tools/collect_profiles.sh on the same build) and labelled with its source.
    traffic, traffic_src, traffic_alg, traffic_note = None, None, None, None
    from video_prediction_amd import lib as _lib
    src_id = _lib.source_id()
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02'):
        pmc_path = os.path.join(ROOT, 'profiles', '%s_convlstm_cell_pmc_%s.json' % (rnd, args.precision))
        if os.path.exists(pmc_path) and args.batch == 16 and args.config == 'c2':
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get('source_id') != src_id:
                    # a counter profile of OTHER kernel sources / tuning tables is not this build's traffic: say so instead of quoting it
                    traffic_note = ('no counter profile of this build: %s was taken on source id %s, this checkout is %s (re-run '
                                    'tests/tools/collect_profiles.sh)' % (os.path.basename(pmc_path), pmc.get('source_id'), src_id))
                    break
                traffic = pmc['avg_hbm_bytes_per_launch_five_layers']
                traffic_alg = pmc.get('avg_algorithmic_bytes_five_layers')
                traffic_src = 'profiles/' + os.path.basename(pmc_path)
                break
            except Exception:
                traffic = None
    frames = world * args.batch * seq * args.steps
    result = {
        'metric': 'train frames/sec (whole node), %s seq%d SAVP' % ({'c1': 'BAIR 64x64 deterministic', 'c2': 'BAIR 64x64', 'c4': 'KTH 64x64', 'c5': 'synthetic 128x128'}[args.config], seq),
        'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': ('%s: SAVP full VAE-GAN (ours_savp recipe), %s, seq=%d, context=%d, nz=%d, batch=%d per GPU, D step + G/E step '
                                'per train step' if engine.has_d else
                                '%s: SAVP deterministic generator (ours_deterministic_l1 recipe), %s, seq=%d, context=%d, nz=%d, batch=%d per GPU, '
                                'one Adam step on the L1 loss per train step') % (args.config, cfg['name'], seq, cfg['context'], hp.nz, args.batch),
                   'global_batch': world * args.batch, 'seq_len': seq, 'parallelism': 'dp%d' % world,
                   'sequences_per_s': world * args.batch * args.steps / dt,
                   'conv_problems_tuned_live': len(K.AUTOTUNE['log']), 'tuner_rejected': len(K.AUTOTUNE['rejected']),
                   'submission': mode, 'eager_ms_per_step': eager_ms, 'host_issue_ms_per_step': host_issue_ms, 'instrumented_ms_per_step': inst_ms},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_TFLOPS[args.precision], 'unit': 'TFLOP/s',
                     'frac': (achieved / PEAK_TFLOPS[args.precision]) if achieved else None, 'traffic': traffic, 'algorithmic_bytes': traffic_alg,
                     'traffic_unit': ('HBM bytes per launch, mean of the 5 layers (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; '
                                     'read from %s -- a counter pass of the same kernel sources + tuning tables (source id %s), not collected by this '
                                     'run); algorithmic bytes: avg_algorithmic_bytes_five_layers of the same file' % (traffic_src, src_id)) if traffic is not None else traffic_note,
                     'kernel': '%s, ConvLSTM gate conv FPROP x%d layers' % (
                         ('conv_gate_kernel (csrc/conv_gate.hip: shape-specialised, LDS patch by LDS-DMA, weights in B-fragment order straight from L2, one K slice per wave, '
                          'bf16 MFMA, fused cell epilogue: bf16 gates + instance-norm statistics)' + ('' if gate_native == len(per_layer) else
                          ' on %d of the layers, conv_ring_kernel on the others' % gate_native)) if (args.precision == 'bf16' and gate_native) else
                         ('conv_ring_kernel (LDS patch + LDS-DMA weight ring, bf16 MFMA, fused cell epilogue: bf16 gates + instance-norm statistics)'
                          if args.precision == 'bf16' else 'conv_fd_kernel (implicit GEMM, fp32 MFMA)'), len(per_layer)),
                     'per_layer': per_layer,
                     'launches_timed': launches, 'avg_launch_us': (tot_s / launches * 1e6) if launches else None, 'clock': clock,
                     'avg_launch_us_with_gaps': gaps_us,
                     'measured_on': '%d instrumented eager steps after the timed region' % INST_STEPS},
        'roofline_cell': {'what': 'fused ConvLSTM cell = gate conv (instance-norm statistics + bf16 gates in its epilogue) + ONE gate-block launch, '
                                  'five layers, HIP events around each cell of the instrumented steps',
                          'avg_cell_us': (cell_s / cell_n * 1e6) if cell_n else None, 'cells_timed': cell_n,
                          'clock': 'HIP event markers around the cell (two launches, three markers: half of it is marker / hand-over gaps)',
                          'kernel_only': {'what': 'the same cells, sum of the two launches\' own begin / end stamps (gate conv + gate block)',
                                          'avg_cell_us': (ck_s / ck_n * 1e6) if ck_n else None, 'cells_timed': ck_n,
                                          'mfma_frac': (ck_flops / ck_s / 1e12 / PEAK_TFLOPS[args.precision]) if ck_s else None,
                                          'hbm_frac': (ck_bytes / ck_s / 1e9 / 8000.0) if ck_s else None},
                          'as_timed': {'what': 'the cell as the timed region runs it: ONE kernel per cell on %d of the %d layers (16 x 16 and 8 x 8: conv_gate_kernel CELL '
                                               'instantiations, dispatch stamps), gate conv + gate block on the others' % (one_layers, len(cells)),
                                       'avg_cell_us': (one_s / one_n * 1e6) if one_n else None, 'cells_timed': one_n,
                                       'mfma_frac': (one_flops / one_s / 1e12 / PEAK_TFLOPS[args.precision]) if one_s else None},
                          'mfma': {'achieved': (cell_flops / cell_s / 1e12) if cell_s else None, 'peak': PEAK_TFLOPS[args.precision], 'unit': 'TFLOP/s',
                                   'frac': (cell_flops / cell_s / 1e12 / PEAK_TFLOPS[args.precision]) if cell_s else None},
                          'hbm': {'achieved': (cell_bytes / cell_s / 1e9) if cell_s else None, 'peak': 8000.0, 'unit': 'GB/s',
                                  'frac': (cell_bytes / cell_s / 1e9 / 8000.0) if cell_s else None,
                                  'algorithmic_bytes_per_cell': (cell_bytes / cell_n) if cell_n else None}},
        'losses': {'d_loss': float(info['d_loss']), 'g_loss': float(info['g_loss'])},
    }
    # kernel time by family: not measurable from inside the run; quoted from the committed rocprofv3 kernel stats of the SAME kernel
    # sources + tuning tables (tests/tools/kernel_families.py stamps the source id), else left out
    if args.config == 'c2' and args.batch == 16:
        for rnd in ('r06', 'r05', 'r04'):
            fam_path = os.path.join(ROOT, 'profiles', '%s_kernel_families.json' % rnd)
            try:
                fam = json.load(open(fam_path))
                if fam.get('source_id') == src_id:
                    result['kernel_families_ms'] = {'source': 'profiles/%s_kernel_families.json (rocprofv3 --kernel-trace --stats, 6 eager steps, same source id)' % rnd,
                                                    'launches_per_step': fam['launches_per_step'], 'kernel_ms_per_step': fam['kernel_ms_per_step'],
                                                    'families': {k: v['ms_per_step'] for k, v in fam['families'].items()}}
                    break
            except Exception:
                pass
    # whole step against the conv roofline (algorithmic FLOPs per sequence and train step, fwd + data-grad + weight-grad)
    step_tflop = {'c2': 0.684, 'c4': 1.004, 'c5': 4.81}.get(args.config)
    if step_tflop:
        tf = step_tflop * world * args.batch * args.steps / dt
        result['roofline_step'] = {'what': 'whole train step: SURVEY.md 8(d) algorithmic TFLOP per sequence x sequences / measured time',
                                   'tflop_per_sequence': step_tflop, 'achieved': tf, 'peak': PEAK_TFLOPS[args.precision] * world, 'unit': 'TFLOP/s',
                                   'frac': tf / (PEAK_TFLOPS[args.precision] * world)}
    if dist is not None:
        st = engine.replicas.stats
        result['config']['dist'] = {'backend': backend + (' (RCCL)' if backend == 'nccl' else ''), 'world': world, 'forced_at_world_1': bool(force_dist and world == 1),
                                    'allreduce_chunks_issued': st['chunks'], 'allreduce_elements': st['elements'], 'aux_broadcasts': st['aux_broadcasts'],
                                    'side_stream': engine.replicas.comm_stream is not None, 'binding': binding,
                                    'collectives_in_graph': bool(getattr(engine, 'graph_collectives', False)),
                                    'ranks_seen_by_backend': dist.get_world_size(), 'allreduce_bytes_per_step': 4 * st['elements'] // max(1, args.warmup + args.steps + INST_STEPS)}
    if dist is not None and os.environ.get('SAVP_BENCH_CHECK_REPLICAS', '0') == '1':
        result['replicas_identical'] = bool(engine.replicas.checksum_identical())      # collective: every rank calls it
    if rank == 0 and args.save_tuning:
        K.save_tuning(args.save_tuning)
    if world == 1 and not args.no_f32 and args.precision == 'bf16':
        # The reference's own arithmetic is fp32: the exact-fp32 datapath (fp32 MFMA pipe, bit-equal to an fmaf chain; the parity
        # mode of the tests) on the same workload, timed the same way (W_f untimed + K_f timed steps, graph replay) as a second object.
        try:
            del engine
            torch.cuda.empty_cache()
            eng32, _ = build_engine('f32')
            eng32.use_graph = not args.eager
            k32 = max(4, min(args.steps, 12))
            dt32, info32 = timed_steps(eng32, 3, k32)
            fl_step = 10.9e12 * args.batch / 16.0 if args.config == 'c2' else None         # 0.684 TFLOP per sequence
            result['f32'] = {'dtype': 'f32', 'steps': k32, 'warmup': 3, 'ms_per_step': dt32 / k32 * 1e3,
                             'value': args.batch * seq * k32 / dt32, 'unit': 'frames/s',
                             'whole_step_tflops': (fl_step * k32 / dt32 / 1e12) if fl_step else None, 'peak_tflops': PEAK_TFLOPS['f32'],
                             'losses': {'d_loss': float(info32['d_loss']), 'g_loss': float(info32['g_loss'])}}
            del eng32
            K.set_conv_precision(args.precision)
        except Exception as ex:            # never lose the headline line to the second datapath
            result['f32'] = {'error': repr(ex)}
    if world == 1 and not args.no_workloads and args.config == 'c2' and args.precision == 'bf16' and not args.eager:
        # The other single-GPU configurations of BASELINE.json (configs[0], [3], [4] at their per-GPU batch), AFTER the headline's timed region:
        # each a fresh engine, 3 untimed steps (untuned conv problems are timed there), then 20 steps replayed as the captured hipGraph between
        # synchronisations -- the same clock as the headline.  Parity of these very steps at these shapes: tests/test_gpu_model.py
        # (test_the_replayed_bench_step_is_the_eager_step_and_matches_the_golden[c4|c5], test_config_c1_...).
        result['workloads'] = {}
        step_tflop_of = {'c2': 0.684, 'c4': 1.004, 'c5': 4.81}
        for name in ('c1', 'c4', 'c5'):
            wc = CONFIGS[name]
            try:
                try:
                    del engine
                except NameError:
                    pass
                torch.cuda.empty_cache()
                tuned0, rej0 = len(K.AUTOTUNE['log']), len(K.AUTOTUNE['rejected'])
                engw, hpw = build_engine('bf16', cfg=wc, batch=wc['batch'], shape=wc['shape'], seq=wc['seq'])
                engw.use_graph = True
                kw = 20
                dtw, infow = timed_steps(engw, 3, kw)
                obj = {'metric': 'train frames/sec, %s seq%d SAVP' % (wc['name'], wc['seq']), 'value': wc['batch'] * wc['seq'] * kw / dtw, 'unit': 'frames/s',
                       'ms_per_step': dtw / kw * 1e3, 'steps': kw, 'warmup': 3, 'dtype': 'bf16', 'data': 'synthetic', 'n_gpus': 1,
                       'submission': 'hipGraph replay' if (engw.graph is not None and engw.graph.segments == 1) else 'eager launches',
                       'config': {'workload': '%s: %s, seq=%d, context=%d, nz=%d, batch=%d per GPU' % (name, wc['name'], wc['seq'], wc['context'], hpw.nz, wc['batch'])},
                       'losses': {'d_loss': float(infow['d_loss']), 'g_loss': float(infow['g_loss'])},
                       'conv_problems_tuned_live': len(K.AUTOTUNE['log']) - tuned0, 'tuner_rejected': len(K.AUTOTUNE['rejected']) - rej0}
                if name in step_tflop_of:
                    tfw = step_tflop_of[name] * wc['batch'] * kw / dtw
                    obj['roofline_step'] = {'tflop_per_sequence': step_tflop_of[name], 'achieved': tfw, 'peak': PEAK_TFLOPS['bf16'], 'unit': 'TFLOP/s',
                                            'frac': tfw / PEAK_TFLOPS['bf16']}
                result['workloads'][name] = obj
                del engw
            except Exception as ex:        # never lose the headline line to an extra workload
                result['workloads'][name] = {'error': repr(ex)}
        torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and args.config == 'c2':     # the CPU sample has the c2 workload's shape
            try:
                result['cpu_baseline'] = cpu_baseline()
            except Exception as ex:    # the oracle is only a reported baseline; never fail the bench on it
                result['cpu_baseline'] = {'value': None, 'unit': 'frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                          'sample': 'failed: %r' % (ex,)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
