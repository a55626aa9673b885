"""Per-kernel-family time of the last `window_ms` of a rocprofv3 kernel trace.  usage: family_summary.py trace.csv steps window_ms"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); win = float(sys.argv[3])
tmax = max(int(r['End_Timestamp']) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r['Start_Timestamp']) < tmax - win * 1e6:
        continue
    n = re.sub(r'<.*', '', r['Kernel_Name'].split('(')[0]).replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    agg[n][0] += 1; agg[n][1] += d
tot = sum(v[1] for v in agg.values())
print('total %.2f ms/step' % (tot / steps / 1e3))
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print('%7.2f ms %7.1f calls avg %7.1f us  %s' % (us / steps / 1e3, c / steps, us / c, n))
