"""python tests/tools/trace_report.py gpurun_out/X_last_step_trace.csv [N] : per (kernel, grid) time of one step."""
import csv, collections, re, sys, statistics
st = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = int(st[0]['Start_Timestamp']); t1 = int(st[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in st)
print('wall ms %.2f launches %d busy ms %.2f' % ((t1 - t0) / 1e6, len(st), busy / 1e6))
agg = collections.defaultdict(lambda: [0, 0]); fam = collections.defaultdict(lambda: [0, 0])
for r in st:
    nm = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = (nm, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'])
    agg[k][0] += 1; agg[k][1] += d
    f = re.sub(r'<.*', '', nm); fam[f][0] += 1; fam[f][1] += d
print('--- families')
for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%7.3f ms %5d x %7.1f us  %s' % (t / 1e6, n, t / n / 1e3, k))
print('--- kernels')
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%7.3f ms %4d x %7.1f us  %s' % (t / 1e6, n, t / n / 1e3, k))
