"""Diagnostic for lease 15: eager generate() between train-step replays (lr = 0) -- what changes when the frames change?"""
import os, sys
sys.path.insert(0, '.')
import torch
from tests.gpu_model_checks import make_hparams
from video_prediction_amd import kernels as K
from video_prediction_amd.models.savp_model import SAVPEngine

small = len(sys.argv) > 1 and sys.argv[1] == 'small'
MODE = sys.argv[4] if len(sys.argv) > 4 else ''
if MODE == 'sidestream':
    torch.cuda.set_stream(torch.cuda.Stream())
if small:
    K._ARENAS['cuda:0'] = K.ZeroArena(torch.device('cuda:0'), floats=1 << 20)
hp = make_hparams(context_frames=2, sequence_length=4, nz=8, lr=0.0, l1_weight=100.0, kl_weight=1.0, video_sn_gan_weight=0.0,
                  video_sn_vae_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
eng = SAVPEngine(hp, (64, 64, 3), 1, mode='train', seed=4)
eng.set_images(torch.rand(4, 1, 64, 64, 3).cuda(), time_major=True)
arena = K.zero_arena(eng.device)
noise = eng.default_noise()
eng.infer_graph = False
ref = eng.generate(noise).clone()
p0 = {n: eng.store[n].clone() for n in eng.store.names()}
snap = lambda: dict(zs=eng.zs_all.clone(), gt=eng.d_gt.clone(), mu=eng.enc.mu.clone(), masks=eng.gen.logits.v.clone(),
                    h0=eng.gen.layers[0]['pre'].v.clone(), a0=eng.gen.layers[0]['a'].v.float().clone())
s0 = snap()
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nosync = len(sys.argv) > 3
for _ in range(pre):
    eng.train_step()
if MODE == 'sync_in_stage':          # host synchronize in front of every staging copy
    _orig = eng._stage_noise
    def _st(n):
        torch.cuda.synchronize()
        return _orig(n)
    eng._stage_noise = _st
def nanmap(tag):
    G = eng.store.groups['g']
    L0, L1 = eng.gen.layers[0], eng.gen.layers[1]
    items = [('g', G.g), ('m', G.m), ('v', G.v), ('p', G.p), ('loss_buf', eng.loss_buf), ('d_scal', eng.d_scal), ('gen', eng.gen.gen.v),
             ('gen.g', eng.gen.gen.g), ('pre0', L0['pre'].v), ('pre0.g', L0['pre'].g), ('gates0', L0['gates'].v), ('gates0.g', L0['gates'].g),
             ('n1mean', L0['n1'].mean), ('n1rstd', L0['n1'].rstd), ('n2rstd', L0['n2'].rstd), ('normrstd0', L0['norm'].rstd), ('c0', L0['c'].v),
             ('a1', L1['a'].v), ('a1.g', L1['a'].g), ('enc_mu', eng.enc.mu), ('zs', eng.zs_all), ('arena_tail', K.zero_arena(eng.device).buf[85248:]),
             ('logits', eng.gen.logits.v), ('maskin.g', eng.gen.maskin.g)]
    flags = torch.stack([(~torch.isfinite(t.float())).any() for _, t in items]).tolist()
    big = torch.stack([t.float().abs().max() for _, t in items]).tolist()
    bad = ['%s' % n for (n, _), f in zip(items, flags) if f]
    print('   %s nonfinite: %s | max|g| %.2e max|v| %.2e d_scal %s arena_tail %.2e' % (tag, ','.join(bad) or '-', big[0], big[2],
          ['%.3g' % q for q in eng.d_scal.tolist()], big[21]))
for i in range(10):
    eng.train_step()
    if MODE == 'trace':
        nanmap('after train')
    if not nosync:
        torch.cuda.synchronize()
    drift = max(float((eng.store[n] - p0[n]).abs().max()) for n in p0)
    off_before = arena.off
    if MODE == 'event':            # order the eager launches behind the replay explicitly
        ev = torch.cuda.Event(); ev.record(); torch.cuda.current_stream().wait_event(ev)
    if MODE == 'sync_before_generate':
        torch.cuda.synchronize()
    if os.environ.get('DIAG_NOGEN') == '1':
        got = ref
    elif os.environ.get('DIAG_NOGEN') == '2':          # generator unroll without its staging copies (the noise is already on the device)
        eng.prep_generator_weights(); got = eng.forward_generator(None)
    else:
        got = eng.generate(noise)
    if MODE == 'trace':
        nanmap('after generate')
    if MODE == 'sync_after_generate':
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    s = snap()
    print('i=%d step=%d graph=%s off %d -> %d mark %s  |gen-ref| %.3e  param drift %.3e  ' % (
        i, eng.step, eng.graph is not None, off_before, arena.off, getattr(eng.graph, 'arena_mark', None), float((got - ref).abs().max()), drift) +
        ' '.join('%s %.2e' % (k, float((s[k].float() - s0[k].float()).abs().max())) for k in s0))
