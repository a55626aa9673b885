"""Developer check: does anything on the HIP path read memory it has not written?  The caching allocator's free blocks are filled with a
poison value (a moderate finite number, then NaN) before a parity check runs, so every torch.empty() of the engine hands out poisoned bytes.
A fresh VRAM page is zero on most boxes, which would mask such a read.  Prints one JSON line per (poison, check)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def poison(value, big_gb=12, small_mb=1024):
    blocks = [torch.full((big_gb * (1 << 28),), value, device='cuda')]                    # fp32: big_gb GB in one large-pool block
    blocks += [torch.full((1 << 17,), value, device='cuda') for _ in range(small_mb * 2)]   # 512 KB each: the small pool's 2 MB segments
    blocks += [torch.full((1 << 8,), value, device='cuda') for _ in range(4096)]
    torch.cuda.synchronize()
    del blocks


def main():
    from tests import gpu_model_checks as G
    t0 = time.time()
    plan = [(37.0, 'gen_fwd'), (37.0, 'train'), (float('nan'), 'gen_fwd'), (float('nan'), 'train')]
    budget = float(os.environ.get('POISON_BUDGET_S', '45'))
    for value, what in plan:
        if time.time() - t0 > budget:
            print(json.dumps({'skipped': [str(value), what]}), flush=True)
            continue
        poison(value)
        if what == 'gen_fwd':
            res = G.check_generator_forward(nz=0, B=2, T=5)
        else:
            res = G.check_train_step(B=2, T=6, nz=8, steps=1, tag='train_poison')
        bad = [(n, float(e), float(t)) for (n, e, t) in res if not (e <= t)]
        print(json.dumps({'poison': str(value), 'check': what, 'checked': len(res), 'bad': bad[:12], 'elapsed_s': round(time.time() - t0, 1)}), flush=True)


if __name__ == '__main__':
    main()
