"""Probe (round-5 verdict item 7): can the step's RCCL collectives be captured INTO its hipGraph on this ROCm / RCCL / PyTorch build?
One rank, backend 'nccl', collectives forced (a sum over one replica is the identity).  Arm A: SAVP_GRAPH_COLLECTIVES=1 -- the side-stream
all-reduces, the u broadcast and their event fork / join are captured; arm B: the shipped segmented replay (graph | host action | graph ...).
Prints one JSON line: segments / host actions per arm, step time of each, and whether the two arms end at bit-identical variables.
Run under `timeout`: a capture that deadlocks must not hold the box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch


def run(arm, steps, timed):
    import torch.distributed as dist
    from tests import gpu_model_checks as G
    from video_prediction_amd.models.savp_model import SAVPEngine
    os.environ['SAVP_GRAPH_COLLECTIVES'] = '1' if arm == 'captured' else '0'
    B, T = int(os.environ.get('B', 16)), int(os.environ.get('T', 30))
    hp, vals, images, _ = G.recipe_case(B, T)
    from video_prediction_amd import kernels as K
    K.set_conv_precision('bf16')
    eng = SAVPEngine(hp, (64, 64, 3), B, mode='train', values=vals, device='cuda:0')
    eng.attach_process_group(dist, force=True)
    eng.set_images(images.float().to('cuda:0'), time_major=True)
    for i in range(steps):
        eng.train_step(G.make_noise(hp, B, seed=100 + i, sampling=True))
    torch.cuda.synchronize()
    noise = G.make_noise(hp, B, seed=7, sampling=True)
    t0 = time.perf_counter()
    for _ in range(timed):
        eng.train_step(noise)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / timed * 1e3
    g = eng.graph
    return {'segments': g.segments if g is not None else 0,
            'host_ops': sum(1 for it in g.items if not isinstance(it, torch.cuda.CUDAGraph)) if g is not None else -1,
            'ms_per_step': ms, 'chunks': eng.replicas.stats['chunks'], 'aux_broadcasts': eng.replicas.stats['aux_broadcasts']}, eng.store.to_numpy()


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29617')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    out = {'torch': torch.__version__}
    try:
        steps, timed = int(os.environ.get('STEPS', 4)), int(os.environ.get('TIMED', 30))
        try:
            out['captured'], pa = run('captured', steps, timed)
        except Exception as ex:                                  # the limitation, written down
            out['captured'] = {'error': repr(ex)[:600]}
            pa = None
        out['segmented'], pb = run('segmented', steps, timed)
        if pa is not None:
            out['variables_bit_identical'] = all((pa[k] == pb[k]).all() for k in pb)
    finally:
        print(json.dumps(out), flush=True)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
