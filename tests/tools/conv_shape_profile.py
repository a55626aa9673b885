"""Per-shape time of the conv launches of one train step: run under
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python tests/tools/conv_shape_profile.py run
then  python tests/tools/conv_shape_profile.py report DIR  (on the same box; the call log is written to /tmp/conv_calls.json)."""
import sys, os, json, csv, collections, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def run():
    import torch
    from tests.gpu_model_checks import make_hparams
    from video_prediction_amd.models.savp_model import SAVPEngine
    from video_prediction_amd import kernels as K
    os.environ['SAVP_GRAPH'] = '0'
    K.set_conv_precision('bf16'); K.enable_autotune(True)
    K.load_tuning(os.path.join(ROOT, 'video_prediction_amd', 'tuning_gfx950_bf16.json'))
    B, T, HW_ = int(os.environ.get('B', 16)), int(os.environ.get('T', 30)), int(os.environ.get('HW', 64))     # c5: B=8 HW=128
    hp = make_hparams(context_frames=2, sequence_length=T, batch_size=B, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                      l2_weight=0.0, kl_weight=1.0, video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                      vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    eng = SAVPEngine(hp, (HW_, HW_, 3), B, mode='train')
    eng.set_images(torch.rand(T, B, HW_, HW_, 3, device='cuda:0'), time_major=True)
    for _ in range(2):
        eng.train_step()
    torch.cuda.synchronize()
    K.CONV_CALL_LOG = []
    eng.train_step()
    torch.cuda.synchronize()
    json.dump(K.CONV_CALL_LOG, open('/tmp/conv_calls.json', 'w'))

def report(d):
    calls = json.load(open('/tmp/conv_calls.json'))
    path = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith('kernel_trace.csv')][0]
    # one kernel per logged call: the main kernel of every conv algorithm (fold / reduce / clear launches are companions, not calls)
    rows = [r for r in csv.DictReader(open(path))
            if re.search(r'conv_fd_kernel|conv_patch_kernel|conv_ring_kernel|conv_wgrad|wgrad_patch_kernel|thin_fprop_kernel|thin_wgrad_kernel|s2dgrad_kernel|thin_dgrad',
                         r['Kernel_Name'])]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[-len(calls):]
    print('logged calls %d, conv kernels in the trace tail %d' % (len(calls), len(rows)))
    agg = collections.defaultdict(lambda: [0, 0.0, ''])
    for c, r in zip(calls, rows):
        k = tuple(tuple(x) if isinstance(x, list) else x for x in c[:12])
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        agg[k][0] += 1; agg[k][1] += dur
        agg[k][2] = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '') + ' tile=%s sk=%d' % (hex(c[12]), c[13])
    tot = sum(v[1] for v in agg.values())
    print('conv total %.2f ms over %d calls' % (tot / 1e3, len(calls)))
    names = {0: 'fprop', 1: 'dgrad', 2: 'wgrad'}
    for k, (n, us, kn) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        mode, N, D, H, W, Cx, Do, Ho, Wo, Cy, kk, ss = k
        fl = 2.0 * N * Do * Ho * Wo * Cx * Cy * kk[0] * kk[1] * kk[2] * n
        print('%7.2f ms %4d x %7.1f us %6.1f TF  %-5s N=%-4d in=%dx%dx%dx%-3d out=%dx%dx%dx%-3d k=%s s=%s  %s' % (
            us / 1e3, n, us / n, fl / us / 1e6, names[mode], N, D, H, W, Cx, Do, Ho, Wo, Cy, 'x'.join(map(str, kk)), 'x'.join(map(str, ss)), kn))

if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else report(sys.argv[2])
