"""Run every GPU parity check, never stop at the first failure, dump results to gpurun_out/checks.json."""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from tests import gpu_checks
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    only = set(sys.argv[1:])
    results = {}
    nbad = 0
    from tests import gpu_model_checks
    checks = list(gpu_checks.ALL_CHECKS) + [('model_fwd', gpu_model_checks.check_model_small), ('model_train', gpu_model_checks.check_train_small), ('model_bf16', gpu_model_checks.check_model_bf16)]
    for name, fn in checks:
        if only and name not in only:
            continue
        t0 = time.time()
        try:
            res = fn()
            torch.cuda.synchronize()
            results[name] = [dict(name=n, err=e, tol=t, ok=bool(e <= t)) for (n, e, t) in res]
            for n, e, t in res:
                ok = e <= t
                nbad += (not ok)
                print('%-6s %-40s err=%.3e tol=%.1e' % ('ok' if ok else 'FAIL', n, e, t), flush=True)
        except Exception:
            nbad += 1
            results[name] = {'exception': traceback.format_exc()}
            print('EXC in %s:\n%s' % (name, traceback.format_exc()), flush=True)
        print('-- %s took %.1fs' % (name, time.time() - t0), flush=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'checks.json'), 'w') as f:
        json.dump(results, f, indent=1)
    print('TOTAL FAILURES: %d' % nbad)
    return 1 if nbad else 0


if __name__ == '__main__':
    sys.exit(main())
