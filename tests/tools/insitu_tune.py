"""In-step tuning of the conv problems that carry the train step (developer tool; writes a tuning table).

The shipped table comes from kernels._tune: every candidate (tile, split-K) of a problem timed as back-to-back launches of that one
kernel.  In the step the same kernel runs between other kernels (cold L2, different neighbours): round 3 measured the 16x16 gate
DGRAD at 34.9 us isolated vs 48.6 us in the step.  This tool ranks the candidates of the TOP problems (by in-step time) with
the isolated tuner, then times the best few of each INSIDE the eager train step (HIP events around their launches, all target
problems switched together) and keeps, per problem, the candidate with the smallest in-step total.

usage: insitu_tune.py OUT.json [TOP=24] [CANDS=6]"""
import collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from video_prediction_amd import kernels as K, lib
from video_prediction_amd.models.savp_model import SAVPEngine

out_path = sys.argv[1]
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 24
NC = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = bench.CONFIGS[os.environ.get('CONFIG', 'c2')]
dev = torch.device('cuda', 0)
K.set_conv_precision('bf16')
K.enable_autotune(True)
table = os.path.join(ROOT, 'video_prediction_amd', 'tuning_gfx950_bf16.json')
K.load_tuning(table)
model = bench.make_hparams(cfg['batch'], cfg['seq'], cfg['context'], cfg['over'])
eng = SAVPEngine(model.hparams, cfg['shape'], cfg['batch'], mode='train', seed=4, device=str(dev))
eng.set_images(bench.synthetic_batch(cfg['batch'], 1234, dev, cfg['seq'], cfg['shape']))
eng.use_graph = False
for _ in range(3):
    eng.train_step()
torch.cuda.synchronize()

def measured_step(mode, targets=()):
    K.INSITU = {'mode': mode, 'targets': set(targets), 'events': [], 'ranked': {}}
    eng.train_step()
    torch.cuda.synchronize()
    tot = collections.defaultdict(lambda: [0, 0.0])
    for key, e0, e1 in K.INSITU['events']:
        tot[key][0] += 1
        tot[key][1] += e0.elapsed_time(e1) * 1e3
    ranked = K.INSITU['ranked']
    K.INSITU = None
    return tot, ranked

tot, _ = measured_step('all')
order = sorted(tot.items(), key=lambda kv: -kv[1][1])
print('conv launches %d, in-step conv time %.2f ms (events include the hand-over gap in front of each kernel)' % (sum(v[0] for v in tot.values()), sum(v[1] for v in tot.values()) / 1e3))
MODES = tuple(int(m) for m in os.environ.get('MODES', '0,1').split(','))        # 0 FPROP, 1 DGRAD, 2 WGRAD
targets = [k for k, v in order if k[0] in MODES and (K.AUTOTUNE['cache'].get(k, (0, 0)) != (0, 0) or k[0] == lib.CONV_WGRAD)][:TOP]
_, ranked = measured_step('rank', targets)
shipped = {k: K.AUTOTUNE['cache'][k] for k in targets}
plans = {}
for k in targets:
    c = [(tile, sk) for _, tile, sk in ranked.get(k, [])][:NC]
    if shipped[k] in c:
        c.remove(shipped[k])
    plans[k] = [shipped[k]] + c[:NC - 1]
results = {k: [] for k in targets}
for i in range(NC):
    act = [k for k in targets if i < len(plans[k])]
    if not act:
        break
    for k in targets:
        K.AUTOTUNE['cache'][k] = plans[k][i] if i < len(plans[k]) else shipped[k]
    eng.train_step()                               # settle
    t, _ = measured_step('targets', act)
    t2, _ = measured_step('targets', act)
    for k in act:
        results[k].append((min(t[k][1], t2[k][1]), plans[k][i]))
changed = 0
gain = 0.0
for k in targets:
    base = results[k][0][0]
    best_t, best_c = min(results[k])
    iso = {(tile, sk): ms for ms, tile, sk in ranked.get(k, [])}
    pick = best_c if best_t < 0.97 * base else shipped[k]       # 3 % margin: below that it is noise
    K.AUTOTUNE['cache'][k] = pick
    flag = ''
    if pick != shipped[k]:
        changed += 1; gain += base - best_t; flag = '  <-- changed'
    print('mode %d N%d %dx%d Cx%d Cy%d k%s s%s src16=%d out16=%d x%d: ' % (k[0], k[2], k[4], k[5], k[6], k[10], k[11][1:], k[12][1:], k[20], k[21], tot[k][0]) +
          ' '.join('%s/%d:%.0f(%.0f)' % (hex(c[0]), c[1], t / tot[k][0], iso.get(c, 0) * 1e3 / 4) for t, c in results[k]) + flag, flush=True)
print('changed %d problems, expected in-step gain %.2f ms' % (changed, gain / 1e3))
# final check: whole step with the picks vs the shipped choices (eager, events around the step)
def step_ms(n=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.train_step()
    e0.record()
    for _ in range(n):
        eng.train_step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
picks = {k: K.AUTOTUNE['cache'][k] for k in targets}
a = step_ms()
for k in targets: K.AUTOTUNE['cache'][k] = shipped[k]
b = step_ms()
for k in targets: K.AUTOTUNE['cache'][k] = picks[k]
a2 = step_ms()
print('eager step: picks %.2f / %.2f ms, shipped %.2f ms' % (a, a2, b))
d = json.load(open(table))
new = 0
for k, v in K.AUTOTUNE['cache'].items():          # problems the shipped table did not know (tuned live during the warm-up steps)
    if repr(k) not in d:
        d[repr(k)] = list(v); new += 1
for k in targets:
    d[repr(k)] = list(picks[k])
print('table: %d entries, %d new' % (len(d), new))
json.dump(d, open(out_path, 'w'), indent=0, sort_keys=True)
