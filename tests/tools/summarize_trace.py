"""Group a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup) and print time per group (per-step view)."""
import csv, sys, collections
path = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
window_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None      # only the last window_ms of the trace
agg = collections.defaultdict(lambda: [0, 0.0])
rows_all = list(csv.DictReader(open(path)))
tmax = max(int(r['End_Timestamp']) for r in rows_all)
with open(path) as f:
    for r in rows_all:
        if window_ms is not None and int(r['Start_Timestamp']) < tmax - window_ms * 1e6:
            continue
        name = r['Kernel_Name'].split('(')[0][:60]
        grid = (r.get('Grid_Size_X'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        k = (name, grid)
        agg[k][0] += 1; agg[k][1] += d
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print('total %.1f ms per step' % (tot / steps / 1e3))
for (name, grid), (cnt, us) in rows[:45]:
    print('%8.2f ms  %6.1f calls  avg %8.1f us  %-52s grid=%s' % (us / steps / 1e3, cnt / steps, us / cnt, name, 'x'.join(str(g) for g in grid)))
