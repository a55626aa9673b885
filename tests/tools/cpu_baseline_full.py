"""One-off: bench.py's CPU baseline on the FULL sequence length of the headline workload (T=30 instead of the bounded 12-frame sample
that the default bench run times).  Writes profiles/<tag>_cpu_baseline_full_T30.json; run on any host (no GPU needed)."""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

if __name__ == '__main__':
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
    bench.CPU_SEQ = 30
    res = bench.cpu_baseline(seconds_budget=1e9)            # at least one full step, at most three
    res['host'] = '%s, %d torch threads' % (platform.processor() or platform.machine(), torch.get_num_threads())
    out = os.path.join(ROOT, 'profiles', '%s_cpu_baseline_full_T30.json' % tag)
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))
