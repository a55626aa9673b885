"""ConvLSTM gate-conv FPROP launches (the five layers at N=32, BAIR) for PMC collection.
  python tests/tools/pmc_conv.py tune    -> writes gpurun_out/pmc_tuned.json (autotuned tile/split per layer)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tests/tools/pmc_conv.py run
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib

LAYERS = [('lstm_h0', 32, 32, 72, 128), ('lstm_h1', 16, 16, 136, 256), ('lstm_h2', 8, 8, 264, 512),
          ('lstm_h3', 16, 16, 136, 256), ('lstm_h4', 32, 32, 72, 128)]
N = 32
PATH = os.path.join(ROOT, 'gpurun_out', 'pmc_tuned.json')


def main():
    mode = sys.argv[1]
    prec = os.environ.get('PREC', 'bf16')
    K.set_conv_precision(prec)
    geom = K.ConvGeom((5, 5), (1, 1), (2, 2))
    bufs = []
    for name, H, W, Cx, Cy in LAYERS:
        x = torch.randn(N, H, W, Cx, device='cuda')
        y = torch.empty(N, H, W, Cy, device='cuda')
        w = torch.randn(Cy, 25 * Cx, device='cuda') * 0.05
        bufs.append((name, x, y, w, w.to(torch.bfloat16) if prec == 'bf16' else None))
    if mode == 'tune':
        K.enable_autotune(True)
        out = {}
        for name, x, y, w, w16 in bufs:
            K.conv(lib.CONV_FPROP, geom, x, y, w, w16=w16)
        log = list(K.AUTOTUNE['log'])
        for i, (name, *_r) in enumerate(bufs):            # h3/h4 repeat the shapes of h1/h0 (cached): reuse their entry
            cfg = log[min(i, len(log) - 1)][1] if i < 3 else out[{'lstm_h3': 'lstm_h1', 'lstm_h4': 'lstm_h0'}[name]]
            out[name] = list(cfg)
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        json.dump({prec: out}, open(PATH, 'w'))
        print(out)
    else:
        cfgs = json.load(open(PATH))[prec]
        for name, x, y, w, w16 in bufs:
            tile, sk = cfgs[name]
            for _ in range(3):
                K.conv(lib.CONV_FPROP, geom, x, y, w, tile=tile, splitk=sk, w16=w16)
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
