"""One conv configuration for PMC collection: SHAPE=lstm_h0:fprop TILE=0x712 [CELL=1] [SK=1] python tests/tools/pmc_one.py
   rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python tests/tools/pmc_one.py ; python tests/tools/pmc_one.py report DIR"""
import collections, csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    import torch
    from video_prediction_amd import kernels as K, lib
    from tests.tools.bench_ring_ab import SHAPES
    K.set_conv_precision('bf16')
    name, mname = os.environ.get('SHAPE', 'lstm_h0:fprop').split(':')
    sh = [s for s in SHAPES if s[0] == name and s[1] == mname][0]
    _, _, N, H, W, Cx, Cy, k = sh
    mode = lib.CONV_FPROP if mname == 'fprop' else lib.CONV_DGRAD
    tile = int(os.environ.get('TILE', '0x712'), 16)
    cell = os.environ.get('CELL', '0') == '1'
    x = torch.randn(N, H, W, Cx, device='cuda')
    if os.environ.get('SRC16', '0') == '1':          # the engine's default: the cell input [x | z | h] is a bf16 tensor
        x = x.to(torch.bfloat16)
    if os.environ.get('TABLE', '0') == '1':           # the exact instantiation bench.py launches: shipped tuning table, tile 0
        K.enable_autotune(True)
        K.load_tuning(os.path.join(ROOT, 'video_prediction_amd', 'tuning_gfx950_bf16.json'))
        tile = 0
    y = torch.empty(N, H, W, Cy, device='cuda', dtype=torch.bfloat16 if cell else torch.float32)
    if mode == lib.CONV_DGRAD:
        y = torch.randn(N, H, W, Cy, device='cuda')
    w = torch.randn(k * k * Cx * Cy, device='cuda') * 0.05
    st = torch.zeros(N, Cy, 2, device='cuda', dtype=torch.float64) if cell else None
    geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
    frag = None
    if cell and mode == lib.CONV_FPROP and k == 5 and x.dtype == torch.bfloat16 and os.environ.get('GATE', '1') == '1':
        # what the engine launches since round 6: the gate convolution's own kernel, given the B-fragment pack (ConvLayer.enable_gate_pack)
        n = K.gate_weights_elems(k * k, Cx, Cy)
        if n:
            frag = torch.empty(n, device='cuda', dtype=torch.bfloat16)
            K.pack_gate_weights(w.reshape(k, k, Cx, Cy), frag)
    for _ in range(6):
        K.conv(mode, geom, x, y, w, tile=tile, w16=w.to(torch.bfloat16), splitk=0 if tile == 0 else int(os.environ.get('SK', '1')), stats=st, w_frag=frag)
    torch.cuda.synchronize()


def report(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith('counter_collection.csv'):
                for row in csv.DictReader(open(os.path.join(r, f))):
                    if os.environ.get('KFILTER', 'conv_') in row['Kernel_Name']:
                        agg[row['Kernel_Name'].split('(')[0][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
    for kn, cs in agg.items():
        print(kn)
        for c, v in sorted(cs.items()):
            # counters come per dispatch (already summed over the chip by rocprofv3's CSV: one row per dimension instance)
            print('   %-28s mean/dispatch %.4g  (n=%d)' % (c, sum(v) / max(1, len(v)) , len(v)))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == 'report':
        report(sys.argv[2])
    else:
        run()
