"""tests/golden/c2_step_golden.npz (CONFIG=c2; round 3's b16_step_golden.npz without projections is what a run without CONFIG writes):
ONE train step of the CPU oracle at exactly the benchmarked problem (BASELINE.json configs[1]:
B=16, T=30, 64x64x3, nz=8, clip_length=10, recipe loss weights), reduced to what a parity test needs -- every loss term, sampled
generated pixels, and per variable the gradient's L2 norm + a seeded sample of its elements (the full gradients are 70 MB).
The GPU test (tests/test_gpu_model.py::test_bench_problem_b16_t30_bf16_step_vs_oracle_golden) re-creates the inputs from the same
seeds (tests/gpu_model_checks.recipe_case) and runs the bf16 datapath with the shipped tuning table.

Run in the build container (no GPU):  python tests/golden/make_b16_step_golden.py        (~6 min on 8 threads, ~25 GB)
CONFIG=c4 / CONFIG=c5 write c4_step_golden.npz (KTH 64x64x1, B=16, T=40, context 10, nz=32) / c5_step_golden.npz (128x128x3, B=8, T=30):
bench.py's other two workloads at their bench shapes (tests/gpu_model_checks.BENCH_CASES); these files also carry, for every variable
above 4096 elements, the gradient's L2 norm per output channel and per (tap, input channel) row -- projections that cover the WHOLE
tensor, so an error confined to elements outside the seeded sample still shows (~10-14 min, 35-50 GB).
The oracle runs in fp32 here (fp64 autograd state of a B=16, T=30 step does not fit the container); its own rounding error,
~1e-5 relative on these quantities, is three orders below the bf16 datapath's tolerances.  PARITY UNPINNED: see oracle/__init__.py.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import train as OT  # noqa: E402
from tests import gpu_model_checks as G  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES = 4096


def sample_index(name, numel):
    """Seeded element sample of one variable's gradient (whole tensor when it is small)."""
    if numel <= SAMPLES:
        return np.arange(numel)
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b'\0'), 'little') % (2 ** 31)
    return np.sort(np.random.default_rng(seed).choice(numel, SAMPLES, replace=False))


def projections(g):
    """L2 norms of a gradient tensor [..., C_out] per output channel and per leading row: 2 small vectors that see every element."""
    g2 = g.detach().double().reshape(-1, g.shape[-1])
    return g2.norm(dim=0).numpy(), g2.norm(dim=1).numpy()


def main(B=16, T=30, config=None):
    case = dict(G.BENCH_CASES[config]) if config else dict(B=B, T=T)
    B, T = case['B'], case['T']
    hp, vals, images, noise = G.recipe_case(**case)
    P = {k: torch.tensor(v, dtype=torch.float32) for k, v in vals.items()}
    n32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in noise.items()}
    t0 = time.time()
    _, _, ref = OT.train_step(P, OT.init_opt_state(P), {'images': images.float()}, hp, n32, noise['d_indices_pre'], noise['d_indices_post'],
                              step=0)
    print('oracle step: %.0f s' % (time.time() - t0))
    save = {'B': np.int32(B), 'T': np.int32(T), 'd_loss': np.float64(ref['d_loss']), 'g_loss': np.float64(ref['g_loss'])}
    for nm, v in ref['g_losses'].items():
        save['g_losses/' + nm] = np.float64(v)
    for nm, v in ref.get('d_losses', {}).items():
        save['d_losses/' + nm] = np.float64(v)
    ts, bs = np.array([0, 1, T // 2, T - 2]), np.array([0, B // 2, B - 1])
    save['gen_t'], save['gen_b'] = ts, bs
    save['gen_images_enc'] = ref['gen_images_enc'][ts][:, bs].numpy().astype(np.float32)
    save['gen_images'] = ref['gen_images'][ts][:, bs].numpy().astype(np.float32)
    for key in ('d_grads', 'g_grads'):
        for name, g in ref[key].items():
            g = g.detach().reshape(-1)
            idx = sample_index(name, g.numel())
            save['%s/%s/norm' % (key, name)] = np.float64(g.double().norm())
            save['%s/%s/max' % (key, name)] = np.float64(g.abs().max())
            save['%s/%s/sample' % (key, name)] = g[torch.from_numpy(idx)].numpy().astype(np.float32)
            if config and g.numel() > SAMPLES and ref[key][name].dim() >= 2:
                pc, pr = projections(ref[key][name])
                save['%s/%s/colnorm' % (key, name)], save['%s/%s/rownorm' % (key, name)] = pc.astype(np.float32), pr.astype(np.float32)
    out = os.path.join(HERE, '%s_step_golden.npz' % (config or 'b16'))
    np.savez_compressed(out, **save)
    print(out, os.path.getsize(out) / 1e6, 'MB', len(save), 'arrays; d_loss', save['d_loss'], 'g_loss', save['g_loss'])


if __name__ == '__main__':
    torch.set_num_threads(int(os.environ.get('THREADS', 8)))
    main(int(os.environ.get('B', 16)), int(os.environ.get('T', 30)), os.environ.get('CONFIG') or None)
