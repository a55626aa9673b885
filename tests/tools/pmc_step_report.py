"""Aggregate the two PMC passes of tests/pmc_step.sh into per-kernel HBM bytes per dispatch (JSON on stdout)."""
import collections, csv, json, os, sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith('counter_collection.csv'):
                per = collections.defaultdict(float)
                names = {}
                for row in csv.DictReader(open(os.path.join(r, f))):
                    if row['Counter_Name'] != counter:
                        continue
                    key = row.get('Dispatch_Id') or row.get('Correlation_Id')
                    per[key] += float(row['Counter_Value'])            # one row per instance (XCD): sum over the chip
                    names[key] = row['Kernel_Name']
                for k, v in per.items():
                    agg[names[k].split('(')[0]].append(v)
    return agg


def main():
    fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes with --kernel-trace only, over 4 steps of bench.py (c2 workload); '
                   'values are KB per dispatch as reported; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of '
                   'a wide coalesced read, MI355X_MICROARCH.md HBM section)', 'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
        out['kernels'][k] = {'dispatches': max(len(f), len(w)), 'fetch_kb_raw': fm, 'write_kb_raw': wm, 'hbm_bytes': (2 * fm + wm) * 1024}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
