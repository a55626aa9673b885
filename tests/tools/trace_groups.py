"""Run-length summary of a rocprofv3 kernel trace: consecutive dispatches of the same kernel/grid -> mean duration.
usage: python tests/tools/trace_groups.py <kernel_trace.csv> [min_count]"""
import csv, sys, re

def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    minc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    groups = []
    for r in rows:
        name = re.sub(r'\(.*', '', r['Kernel_Name'])
        key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Z', ''), r.get('LDS_Block_Size', ''))
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if groups and groups[-1][0] == key:
            groups[-1][1].append(d)
        else:
            groups.append((key, [d]))
    for key, ds in groups:
        if len(ds) < minc:
            continue
        ds2 = sorted(ds)[:max(1, len(ds) * 3 // 4)]
        print('%-60s grid=%-8s z=%-3s lds=%-7s n=%-3d mean=%8.1f us  min=%8.1f us' % (key[0][:60], key[1], key[2], key[3], len(ds), sum(ds2) / len(ds2), ds2[0]))

if __name__ == '__main__':
    main()
