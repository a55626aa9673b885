"""Inference throughput of the generator unroll (scripts/generate.py:166 = model.outputs['gen_images']; the sampling loop of
eval_outputs_and_metrics_fn, base_model.py:176-198) on bench.py's workloads: one call = weight preparation + posterior encoder +
the N = 2B batched unroll (posterior and prior halves) of T-1 steps, inputs resident in HBM, noise staged per call.  Timed twice in one
process: launch by launch (SAVP_INFER_GRAPH=0 behaviour) and as the replayed hipGraph (default), K calls each between synchronizes.
One JSON line on stdout; `value` counts the prior half only (B x (T-1) predicted frames per call -- what generate.py keeps).
usage: bench_generate.py [--config c2|c4|c5|c1] [--calls 40] [--warmup 3] [--precision bf16|f32]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench as B
from video_prediction_amd import kernels as K, lib
from video_prediction_amd.models.savp_model import SAVPEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--calls', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default='bf16')
    args = ap.parse_args()
    cfg = B.CONFIGS[args.config]
    dev = torch.device('cuda:0')
    K.set_conv_precision(args.precision)
    K.enable_autotune(True)
    table = os.path.join(ROOT, 'video_prediction_amd', 'tuning_gfx950_%s.json' % args.precision)
    if os.path.exists(table):
        K.load_tuning(table)
    hp = B.make_hparams(cfg['batch'], cfg['seq'], cfg['context'], cfg['over']).hparams
    eng = SAVPEngine(hp, cfg['shape'], cfg['batch'], mode='test', seed=4, device=str(dev))
    eng.set_images(B.synthetic_batch(cfg['batch'], 1234, dev, cfg['seq'], cfg['shape']))
    noises = [eng.default_noise(torch.Generator().manual_seed(100 + i)) for i in range(8)]

    def timed(replay):
        eng.infer_graph = replay
        for i in range(args.warmup):
            eng.generate(noises[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.calls):
            eng.generate(noises[i % 8])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.calls * 1e3

    eager_ms = timed(False)
    replay_ms = timed(True)
    assert eng.gen_graph is not None, 'the unroll was not captured'
    frames = cfg['batch'] * (cfg['seq'] - 1)
    print(json.dumps({'metric': 'inference frames/sec (prior unroll), %s seq%d SAVP' % (cfg['name'], cfg['seq']),
                      'value': frames / replay_ms * 1e3, 'unit': 'frames/s', 'n_gpus': 1, 'calls': args.calls, 'warmup': args.warmup,
                      'ms_per_call': replay_ms, 'eager_ms_per_call': eager_ms, 'higher_is_better': True, 'dtype': args.precision,
                      'data': 'synthetic', 'source_id': lib.source_id(),
                      'config': {'workload': '%s: generator unroll, batch %d (N = %d with the posterior half), %d steps' %
                                 (args.config, cfg['batch'], eng.N, cfg['seq'] - 1), 'submission': 'hipGraph replay'}}))


if __name__ == '__main__':
    main()
