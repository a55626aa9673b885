"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (same workload, two builds).  usage: compare_stats.py a.csv b.csv [steps]"""
import csv, re, sys
def load(fn):
    d = {}
    for r in csv.DictReader(open(fn)):
        n = re.sub(r'\(.*', '', r['Name']).replace('void ', '')
        c, t = d.get(n, (0, 0.0))
        d[n] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
rows = []
for n in set(a) | set(b):
    ca, ta = a.get(n, (0, 0.0)); cb, tb = b.get(n, (0, 0.0))
    rows.append(((tb - ta) / steps / 1e3, n, ca / steps, ta / max(ca, 1) / 1e3, cb / steps, tb / max(cb, 1) / 1e3))
rows.sort()
print('total a %.2f ms/step, b %.2f ms/step' % (sum(v[1] for v in a.values()) / steps / 1e6, sum(v[1] for v in b.values()) / steps / 1e6))
for d, n, ca, ua, cb, ub in rows[:12] + rows[-12:]:
    print('%+8.1f us/step  a %6.1f x %7.1f us   b %6.1f x %7.1f us   %s' % (d, ca, ua, cb, ub, n[:70]))
