"""Data-parallel host logic on CPU with world_size 2 and 4 over gloo: flat-bucket all-reduce, 1/K scaling, rank-0 broadcast,
batch sharding.  Gradients come from the oracle (the HIP kernels cannot run without a GPU); what is under test is
video_prediction_amd.parallel.ReplicaGroup and the ParamStore arenas."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    from video_prediction_amd import variables as V
    from video_prediction_amd.hparams import HParams
    from video_prediction_amd.models.hparam_defaults import savp_defaults
    hp = HParams(**savp_defaults())
    hp.override_from_dict(dict(context_frames=2, sequence_length=3, nz=0, ngf=8, l1_weight=1.0, schedule_sampling='none'))
    H = W = 32
    specs = V.variable_specs(hp, (H, W, 3), mode='train')
    return hp, specs, (H, W, 3)


def _oracle_grads(hp, values, images):
    from oracle import savp as OS, train as OT
    P = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in values.items()}
    out = OS.generator_fn(OS.Scope(P).sub('generator'), {'images': images}, 'train', hp, {})
    loss = OT.l1_loss(out['gen_images'], images[1:])
    names = list(P)
    grads = torch.autograd.grad(loss, [P[n] for n in names], allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(P[n])) for n, g in zip(names, grads)}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from video_prediction_amd import variables as V
    from video_prediction_amd.engine import ParamStore
    from video_prediction_amd.parallel import ReplicaGroup
    from oracle import tf_ops as TF
    hp, specs, shape = _setup()
    vals = V.init_variables(specs, seed=4 + rank)          # replicas start different on purpose
    store = ParamStore(specs, vals, 'cpu')
    rg = ReplicaGroup(store, dist)                          # broadcast from rank 0
    rng = np.random.default_rng(0)
    global_images = torch.tensor(rng.random((3, world, 32, 32, 3)))     # [T, global B = one sample per rank, ...]
    mine = rg.shard(global_images, dim=1)
    assert mine.shape[1] == 1 and torch.equal(mine[:, 0], global_images[:, rank])
    cur = {n: store[n].numpy().copy() for n in store.names()}
    grads = _oracle_grads(hp, cur, mine)
    g = store.groups['g']
    for n, gr in grads.items():
        store.grad(n).copy_(gr.float())
    # chunked exchange as SAVPEngine issues it: one network's contiguous chunk first, the rest of the arena at finish
    lo, hi = store.chunk_of('g', 'generator/rnn/savp_cell/lstm_h2/')
    assert 0 <= lo < hi <= g.g.numel()
    rg.begin_allreduce('g', lo, hi)
    rg.finish_allreduce('g')
    assert rg.stats['chunks'] == 3 and rg.stats['elements'] == g.g.numel(), rg.stats      # [0,lo) [lo,hi) [hi,n): every element once
    avg = g.g * rg.grad_scale
    p, m, v = TF.adam_update(g.p, avg, g.m, g.v, 1e-3, 0.9, 0.999, 1)
    g.p.copy_(p)
    same = rg.checksum_identical()
    if rank == 0:
        q.put({'avg_grads': {n: (store.grad(n) * rg.grad_scale).numpy().copy() for n in grads}, 'same': same,
               'start_vals': cur})
    else:
        q.put({'same': same})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 4])
def test_replicas_average_gradients_and_stay_identical(world):
    import multiprocessing
    ctx = multiprocessing.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r['same'] for r in results), 'replicas diverged'
    r0 = [r for r in results if 'avg_grads' in r][0]
    # the averaged shard gradients equal the single-process gradient of the global batch (per-sample ops only)
    hp, specs, shape = _setup()
    rng = np.random.default_rng(0)
    global_images = torch.tensor(rng.random((3, world, 32, 32, 3)))
    ref = _oracle_grads(hp, r0['start_vals'], global_images)
    gmax = max(float(v.abs().max()) for v in ref.values())
    for n, gref in ref.items():
        err = float((torch.tensor(r0['avg_grads'][n]).double() - gref).abs().max())
        assert err <= 1e-5 * gmax + 1e-9, (n, err)
