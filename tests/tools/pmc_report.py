"""Build profiles/<round>_convlstm_fprop_pmc_<prec>.json from two rocprofv3 --pmc passes of tests/pmc_conv.py run.
usage: python tests/tools/pmc_report.py FETCH_DIR WRITE_DIR TRACE_DIR out.json
FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md) -> fetch x2."""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.tools.pmc_conv import LAYERS, N


def find(d, suffix):
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith(suffix):
                return os.path.join(r, f)
    raise FileNotFoundError(suffix + ' under ' + d)


def counter_per_dispatch(d, counter):
    rows = list(csv.DictReader(open(find(d, 'counter_collection.csv'))))
    per = collections.OrderedDict()
    for r in rows:
        if r['Counter_Name'] != counter or 'conv_' not in r['Kernel_Name']:
            continue
        k = int(r['Dispatch_Id'])
        per[k] = per.get(k, 0.0) + float(r['Counter_Value'])
    return [per[k] for k in sorted(per)]


def durations(d):
    rows = [r for r in csv.DictReader(open(find(d, 'kernel_trace.csv'))) if 'conv_' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    return [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3 for r in rows], [r['Kernel_Name'].split('(')[0] for r in rows]


def main():
    fd, wd, td, out = sys.argv[1:5]
    fetch, write = counter_per_dispatch(fd, 'FETCH_SIZE'), counter_per_dispatch(wd, 'WRITE_SIZE')
    dur, names = durations(td)
    assert len(fetch) == len(write) == len(dur) == 3 * len(LAYERS), (len(fetch), len(write), len(dur))
    res, tot = {}, 0.0
    for i, (name, H, W, Cx, Cy) in enumerate(LAYERS):
        f = sum(fetch[3 * i:3 * i + 3]) / 3.0
        w = sum(write[3 * i:3 * i + 3]) / 3.0
        alg = N * H * W * (Cx + Cy) * 4 + 25 * Cx * Cy * 2          # fp32 activations in/out + bf16 packed weights
        res[name] = {'kernel': names[3 * i].replace('void ', ''), 'fetch_kb_raw': f, 'write_kb_raw': w,
                     'hbm_bytes_corrected': (2.0 * f + w) * 1024.0, 'algorithmic_bytes': alg,
                     'avg_us': sum(dur[3 * i + 1:3 * i + 3]) / 2.0}
        tot += res[name]['hbm_bytes_corrected']
    json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), ConvLSTM gate conv FPROP, '
                       'N=32, bf16, autotuned tiles; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 correction)',
               'layers': res, 'avg_hbm_bytes_per_launch_five_layers': tot / len(LAYERS)}, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
