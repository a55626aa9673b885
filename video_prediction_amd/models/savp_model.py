"""SAVP model on the MI355X engine: generator_fn / posterior_fn / discriminator_fn plug-ins and the model class.

API surface follows /root/reference/video_prediction/models/savp_model.py: the three plug-in functions keep their
signatures (``generator_fn(inputs, mode, hparams)``, ``discriminator_fn(inputs, outputs, mode, hparams)``,
``posterior_fn(inputs, hparams)``), take/return time-major NHWC tensors in ``OrderedDict``s with the reference's keys,
and ``SAVPVideoPredictionModel`` keeps the constructor/hparams handling of savp_model.py:771-846.  TensorFlow's
implicit state (variable scopes, random ops, sessions) is made explicit: variables live in a ``ParamStore`` keyed by
the TF variable names, random draws are passed in a ``noise`` dict (generated from a seeded torch Generator when not
given), and one ``train_step`` call is one ``sess.run(train_op)``.
"""
import collections
from collections import OrderedDict

import os

import numpy as np
import torch

from .. import kernels as K
from .. import variables as V
from ..engine import ParamStore, copy_view, add_views
from .base_model import VideoPredictionModel, learning_rate, kl_weight
from .hparam_defaults import savp_defaults, SAVP_DEPRECATED_KEYS
from .networks import PosteriorEncoder, SNDiscriminator
from .savp_cell import SAVPGenerator


def _image_shape(images):
    return tuple(images.shape[2:])


IDX_KEYS = ('enc_real', 'enc_fake', 'real', 'fake')


class _StepProgram(object):
    """One train step as a sequence of hipGraph segments and host actions.

    The launch sequence of a step has no host input, so on one GPU it is one hipGraph.  With data-parallel replicas the step
    also contains collectives (base_model.py:590-592,614-616: gradient sums; here chunked onto a side stream, parallel.py) that
    must stay outside a capture: RCCL calls are issued by the host, in the same order on every rank.  The program therefore
    cuts the launch sequence at every host action: [graph 0] action [graph 1] action ... [graph n].  Replaying it costs a
    handful of hipGraphLaunch + torch.distributed calls per step, so eight replica processes on one host no longer each need
    tens of milliseconds of Python per step to keep their GPU busy.  All segments share one private memory pool (they are
    always replayed in capture order)."""

    def __init__(self, device):
        self.device = device
        self.items = []
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device=device)
        self.cur = None
        self.arena_mark = 0

    def _begin(self):
        self.cur = torch.cuda.CUDAGraph()
        # thread_local: ProcessGroupNCCL's watchdog thread may query events while this thread captures
        self.cur.capture_begin(pool=self.pool, capture_error_mode='thread_local')

    def _end(self):
        g, self.cur = self.cur, None
        g.capture_end()
        self.items.append(g)

    def host_op(self, fn):
        self._end()
        self.items.append(fn)
        self._begin()

    def capture(self, engine, body):
        """Run body() once under capture (nothing executes; host actions are recorded, not performed).  Returns body's value."""
        import gc
        torch.cuda.synchronize()
        gc.collect()
        main = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(main)
        arena = K.zero_arena(self.device)
        arena.hi = 0
        try:
            with torch.cuda.stream(self.stream):
                try:
                    self._begin()
                    engine._rec = self
                    out = body()
                    self._end()
                except Exception:
                    # end the open capture ON THE CAPTURE STREAM (capture_end checks the stream it began on; outside this `with` it
                    # would raise, the exception would be swallowed and the stream would stay in capture mode: the caller's eager
                    # fallback then fails in its first synchronize)
                    if self.cur is not None:
                        try:
                            self.cur.capture_end()
                        except Exception:
                            pass
                        self.cur = None
                    self.items = []
                    raise
        finally:
            engine._rec = None
        main.wait_stream(self.stream)
        self.arena_mark = arena.hi          # the body starts with arena.reset(): see ZeroArena.replayed
        return out

    @property
    def segments(self):
        return sum(1 for it in self.items if isinstance(it, torch.cuda.CUDAGraph))

    def run(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                it()
        K.zero_arena(self.device).replayed(self.arena_mark)

    replay = run


class SAVPEngine(object):
    """All device state of one SAVP replica: variables, generator unroll (batch 2B: posterior + prior), posterior
    encoder, the two video discriminators; implements one training step and inference."""

    # loss terms of base_model.py:733-829 that need networks / outputs the SAVP generator does not have (VGG features; `gen_features` and
    # `gen_images_dec` are outputs of other model families: the reference's loss_fn raises KeyError for them on SAVP): accepting them
    # silently would train a different model than the recipe asks for.  (z_l1_weight is a declared hyper-parameter no loss reads,
    # base_model.py:398: accepted and without effect here as there.)
    UNSUPPORTED_WEIGHTS = ('vgg_cdist_weight', 'feature_l2_weight', 'ae_l2_weight')

    def __init__(self, hp, image_shape, batch_size, mode='train', values=None, seed=4, device='cuda:0', base_seed=0, rank=0,
                 cond=(0, 0)):
        """cond = (n_actions, n_states): the widths of inputs['actions'] [B, T-1, na] / inputs['states'] [B, T, ns] when the dataset
        supplies them (the action / state-conditioned model, savp_model.py:24-26,411-444,655-661; base_model.py:758-762)."""
        self.hp, self.mode, self.B = hp, mode, batch_size
        bad = [k for k in self.UNSUPPORTED_WEIGHTS if getattr(hp, k, 0)]
        if bad and mode == 'train':
            raise NotImplementedError('loss weights not covered by the HIP path: %s' % ', '.join(bad))
        self.na, self.ns = int(cond[0]), int(cond[1])
        self.cond = (self.na, self.ns)
        if getattr(hp, 'state_weight', 0) and mode == 'train' and not self.ns:
            raise KeyError('states')               # base_model.py:758-760 reads inputs['states'] whenever state_weight is set
        if getattr(hp, 'tv_weight', 0) and mode == 'train' and hp.transformation != 'flow':
            raise KeyError('gen_flows')            # base_model.py:763-764: only the flow transformation has that output (savp_model.py:668-669)
        self.base_seed, self.rank = int(base_seed), int(rank)
        self.image_shape = tuple(image_shape)
        self.device = torch.device(device)
        self.train = mode == 'train'
        specs = V.variable_specs(hp, image_shape, mode=mode, cond=self.cond)
        if values is None:
            values = V.init_variables(specs, seed=seed)
        self.store = ParamStore(specs, values, self.device)
        self.nz = hp.nz
        B = batch_size
        self.T, self.T1 = hp.sequence_length, hp.sequence_length - 1
        H, W, C = image_shape
        self.N = N = 2 * B if self.nz else B
        self.gen = SAVPGenerator(self.store, hp, image_shape, N, train=self.train, cond=self.cond)
        self.enc = PosteriorEncoder(self.store, hp, image_shape, B, train=self.train, n_actions=self.na) if self.nz else None
        # conditioning inputs, time-major; both unrolls (N = 2B: posterior half, prior half) see the same actions / states
        self.actions_tm = torch.zeros(self.T1, B, self.na, device=self.device) if self.na else None
        self.actions_n = (torch.zeros(self.T1, N, self.na, device=self.device) if self.nz else self.actions_tm) if self.na else None
        self.states_tm = torch.zeros(self.T, B, self.ns, device=self.device) if self.ns else None
        self.states_n = (torch.zeros(self.T, N, self.ns, device=self.device) if self.nz else self.states_tm) if self.ns else None
        self.images_tm = torch.empty(self.T, B, H, W, C, device=self.device)
        self.images_n = torch.empty(self.T, N, H, W, C, device=self.device) if self.nz else self.images_tm
        self.zs_all = torch.zeros(self.T1, N, self.nz, device=self.device) if self.nz else None
        self.dz_post = torch.zeros(self.T1, B, self.nz, device=self.device) if (self.nz and self.train) else None
        self.dz_prior = torch.zeros(self.T1, B, self.nz, device=self.device) if (self.nz and self.train and hp.learn_prior) else None
        self.has_d = self.train and V.uses_discriminator(hp)
        if self.has_d and self.T1 < hp.clip_length:
            # tf.random_uniform(maxval <= minval) raises in the reference (savp_model.py:97); a silent pass would gather clips
            # beyond the sequence
            raise ValueError('clip_length=%d needs sequence_length >= %d (got %d)' % (hp.clip_length, hp.clip_length + 1, self.T))
        # learned prior (prior_fn, savp_model.py:54-85,717-721): its own encoder + recurrent tail under scope generator/prior
        self.learn_prior = bool(self.nz and hp.learn_prior)
        self.prior = (PosteriorEncoder(self.store, hp, image_shape, B, train=self.train, prefix='generator/prior/', prior=True,
                                       n_actions=self.na)
                      if self.learn_prior else None)
        # (discriminator, loss weight, loss-name infix, operates on the posterior ('_enc') unroll?, clip index keys)
        self.discs = []
        if self.has_d:
            if hp.gan_loss_type not in K.GAN_TYPES:
                raise ValueError('Unknown GAN loss type %s' % hp.gan_loss_type)
            slot = 0
            for enc in ((True, False) if self.nz else (False,)):       # encoder scope first (savp_model.py:130-147)
                pre = 'discriminator/' + ('encoder/' if (enc and not hp.use_same_discriminator) else '')
                for kind, infix, w_gan, w_vae in (('image', 'image_sn', hp.image_sn_gan_weight, hp.image_sn_vae_gan_weight),
                                                  ('video', 'video_sn', hp.video_sn_gan_weight, hp.video_sn_vae_gan_weight),
                                                  ('images', 'images_sn', hp.images_sn_gan_weight, hp.images_sn_vae_gan_weight)):
                    if not (w_gan or w_vae):
                        continue                                            # network not instantiated (savp_model.py:105,112,119)
                    if enc and hp.use_same_discriminator:
                        D = [d for d in self.discs if d['kind'] == kind and not d['enc']]
                        D = D[0]['D'] if D else None
                    else:
                        D = None
                    if D is None:
                        D = SNDiscriminator(self.store, hp, image_shape, 2 * B, pre + kind + '/', kind=kind)
                    self.discs.append(dict(D=D, kind=kind, enc=enc, w=(w_vae if enc else w_gan), infix=infix,
                                           name='%s_%s' % (infix, 'vae_gan' if enc else 'gan'),
                                           kr='enc_real' if enc else 'real', kf='enc_fake' if enc else 'fake', slot=slot))
                    slot += 6
            self.loss_buf_size = max(16, slot + 8)
        self.d_gan = next((d['D'] for d in self.discs if d['kind'] == 'video' and not d['enc']), None)
        self.d_vae = next((d['D'] for d in self.discs if d['kind'] == 'video' and d['enc']), None)
        # loss scalars: FLOAT64 accumulators (every workgroup of a loss kernel adds its partial: exact in float64, so the value does not
        # depend on arrival order); rounded to fp32 once in the step's bookkeeping
        self.loss_buf = torch.zeros(getattr(self, 'loss_buf_size', 16), device=self.device, dtype=torch.float64)
        self.step = 0
        self.world = 1
        self.dp = False
        self.graph_collectives = False
        self._rec = None
        self.dist = None
        # per-step inputs drawn on the host are staged into these persistent device buffers BEFORE the kernels of the step
        # are launched, so that the launch sequence itself has no host input and can be captured into a hipGraph
        dev = self.device
        self.d_gt = torch.zeros(self.T1, N, dtype=torch.int32, device=dev)
        self.d_eps = torch.zeros(self.T1, B, self.nz, device=dev) if self.nz else None
        self.d_prior = torch.zeros(self.T - hp.context_frames, B, self.nz, device=dev) if self.nz else None
        self.d_prior_eps = torch.zeros(self.T1, B, self.nz, device=dev) if self.learn_prior else None
        self.d_idx = torch.zeros(2, len(IDX_KEYS), 2, B, dtype=torch.int32, device=dev)    # [pre/post][key][t_sample/t_start][B]
        self.d_scal = torch.zeros(4, device=dev)                                           # lr_t(D), lr_t(G), kl weight
        self.graph = None
        self.graph_info = None
        self.use_graph = self.train and os.environ.get('SAVP_GRAPH', '1') == '1'
        self.side_prep = os.environ.get('SAVP_SIDE_PREP', '0') == '1' and self.device.type == 'cuda'
        self._prep_stream = None
        self.eager_steps = 0
        self.infer_graph = os.environ.get('SAVP_INFER_GRAPH', '1') == '1' and self.device.type == 'cuda'
        self.gen_graph, self.gen_graph_out, self.gen_eager = None, None, 0

    # -- data-parallel replicas (base_model.py:517-692 / tf_utils.allreduce_grads) ---------------------------------
    def attach_process_group(self, dist_module, force=None):
        """One process per GPU; see parallel.ReplicaGroup.  force (default: SAVP_FORCE_DIST=1): keep every collective of the
        step even in a group of one rank -- RCCL / the side stream / the segmented replay exercised on a one-GPU box."""
        from ..parallel import ReplicaGroup
        if force is None:
            force = os.environ.get('SAVP_FORCE_DIST', '0') == '1'
        want_graph_coll = os.environ.get('SAVP_GRAPH_COLLECTIVES', '0') == '1'
        self.replicas = ReplicaGroup(self.store, dist_module, overlap=os.environ.get('SAVP_DP_OVERLAP', '1') == '1', force=force,
                                     own_comm=want_graph_coll)
        self.dist = dist_module
        self.world = self.replicas.world
        self.dp = self.replicas.active          # the step carries collectives (world > 1, or forced)
        # Collectives inside the step's hipGraph (SAVP_GRAPH_COLLECTIVES=1): one graph launch per step instead of 8 segments + 7 host actions.
        # The replica group then owns its RCCL communicator and issues ncclAllReduce / ncclBroadcast through the C ABI
        # (parallel.ReplicaGroup._own_communicator): captured through ProcessGroupNCCL instead, the process-group watchdog thread polls
        # end events that were recorded while capturing and aborts the process (profiles/r06_graph_collectives_watchdog_abort.log).
        # Opt-in: validated at world size 1 (tests/test_gpu_dp.py, tests/tools/ab_calls/graph_collectives_probe.py); no N > 1 lease exists
        # to show every rank's replayed graph issuing its RCCL kernels in a compatible order, and a hang there would cost the scaling run.
        self.graph_collectives = self.dp and want_graph_coll and self.replicas.comm is not None
        # Segmented replay pays when the collectives are stream-ordered (RCCL).  Under a backend whose collectives block the host
        # on device tensors (gloo) every segment boundary is a full drain; measured with two ranks time-slicing one MI355X:
        # 146 ms/step launch by launch against 213 ms/step replayed in 8 segments (profiles/r04_ab_calls.md, call 13).
        try:
            backend = str(dist_module.get_backend())
        except Exception:
            backend = None
        if self.dp and backend != 'nccl' and 'SAVP_GRAPH' not in os.environ:
            self.use_graph = False
        K.set_tuning_group(dist_module, force=self.dp)     # rank 0's tuning table, and rank 0's choice for problems tuned live

    def wait_aux(self):
        """Order the compute stream behind the side-stream broadcast of the spectral-norm u vectors (ReplicaGroup.sync_aux):
        called by every entry point that reads or writes the 'aux' arena outside the train step (discriminator_fn, restore,
        save, variable loads)."""
        if self.dp:
            self.replicas.wait_aux()

    def _host_op(self, fn):
        """A host-side action between launches of the step (a collective on the side stream, an event wait).  Eager step: done
        now.  While the step is being captured (_StepProgram): closes the current hipGraph segment, is recorded as the action
        to perform between this segment and the next one, and a new segment is opened."""
        if self._rec is not None and not self.graph_collectives:
            self._rec.host_op(fn)
        else:
            # eager step -- or SAVP_GRAPH_COLLECTIVES=1: the action runs while the step is being captured, so the RCCL call on the side
            # stream and its event fork / join become nodes of the ONE hipGraph the step is replayed as (ProcessGroupNCCL issues
            # ncclAllReduce on torch's current stream: a capturing stream records it)
            fn()

    def _begin_allreduce(self, group, prefix):
        """Start the exchange of the gradients of the variables under `prefix` (one network = one contiguous chunk)."""
        if self.dp:
            try:
                lo, hi = self.store.chunk_of(group, prefix)
            except ValueError:          # scope not contiguous in the arena: left to finish_allreduce (exchanged at the end)
                return
            self._host_op(lambda: self.replicas.begin_allreduce(group, lo, hi))

    def _allreduce(self, group):
        """Complete the exchange of the group's gradient bucket (chunks not started yet are exchanged now)."""
        if self.dp:
            self._host_op(lambda: self.replicas.finish_allreduce(group))

    # -- input staging -------------------------------------------------------------------------------------------------
    def set_conditioning(self, actions=None, states=None, time_major=False):
        """actions [B, T-1, na] / states [B, T, ns] (reference layout; [T, B, ...] if time_major), sliced to the model's sequence length
        (tf_utils.maybe_pad_or_slice, savp_model.py:690-691)."""
        for name, x, buf, buf_n, steps in (('actions', actions, self.actions_tm, self.actions_n, self.T1),
                                          ('states', states, self.states_tm, self.states_n, self.T)):
            if buf is None:
                if x is not None:
                    raise ValueError("inputs[%r] given to a model built without it (build_graph fixes the input structure)" % name)
                continue
            if x is None:
                raise KeyError(name)
            x = x if time_major else x.transpose(0, 1)
            if x.shape[0] < steps or tuple(x.shape[1:]) != tuple(buf.shape[1:]):
                raise ValueError('inputs[%r]: expected %d steps of %r, got %r' % (name, steps, tuple(buf.shape[1:]), tuple(x.shape)))
            buf.copy_(x[:steps])
            if buf_n is not buf:
                buf_n[:, :self.B].copy_(buf)
                buf_n[:, self.B:].copy_(buf)

    def set_images(self, images, time_major=False):
        """images: device fp32 [B,T,H,W,C] (reference layout) or [T,B,H,W,C] if time_major; or the dataset's inputs dict ('images' and,
        for a model built with cond, 'actions' / 'states'; 'pix_distribs' is refused, see refuse_conditioning_inputs)."""
        if isinstance(images, dict):
            refuse_conditioning_inputs(images)
            self.set_conditioning(images.get('actions'), images.get('states'), time_major=time_major)
            images = images['images']
        T = self.T
        if time_major:
            self.images_tm.copy_(images[:T])
        else:
            for t in range(T):
                copy_view(images[:, t], [self.images_tm[t]])                 # transpose_batch_time (tf_utils.py:118-122)
        if self.nz:
            B = self.B
            HW = self.image_shape[0] * self.image_shape[1]
            C = self.image_shape[2]
            src = self.images_tm.reshape(T, B * HW, C)
            copy_view(src, [self.images_n[:, :B].reshape(T, B * HW, C)])
            copy_view(src, [self.images_n[:, B:].reshape(T, B * HW, C)])

    def _noise_seed(self, step, stream=0):
        """Seed of the host-side draws of one step: every (base_seed, rank, step, stream) gets its own sequence -- the reference's
        random ops are independent per tower (base_model.py:523-560) and per run."""
        x = (self.base_seed * 0x9E3779B97F4A7C15 + self.rank * 0xBF58476D1CE4E5B9 + (step + 1) * 0x94D049BB133111EB +
             stream * 0xD6E8FEB86659FD93) & ((1 << 63) - 1)
        x ^= x >> 31
        return (x * 0x2545F4914F6CDD1D) & ((1 << 63) - 1)

    def default_noise(self, generator=None):
        """Draw the step's random tensors (eps, prior z, scheduled-sampling masks, clip indices) on the host."""
        g = generator or torch.Generator().manual_seed(self._noise_seed(self.step))
        hp, B, T1 = self.hp, self.B, self.T1
        noise = {}
        if self.nz:
            noise['eps'] = torch.randn(T1, B, self.nz, generator=g)
            if self.learn_prior:
                noise['prior_eps'] = torch.randn(T1, B, self.nz, generator=g)          # eps of the learned prior (:719-720)
            else:
                noise['prior'] = torch.randn(self.T - hp.context_frames, B, self.nz, generator=g)
        ns = T1 - hp.context_frames
        if self.train and hp.schedule_sampling != 'none':
            prob = self.schedule_sampling_prob()
            for key in ('ground_truth_sampling', 'ground_truth_sampling_enc'):
                noise[key] = (torch.rand(ns, B, generator=g) < prob) if prob >= 0.001 else torch.zeros(ns, B, dtype=torch.bool)
        L = T1
        idx = {}
        if not self.has_d:                                     # no discriminator clips to draw
            return noise
        for phase in ('pre', 'post'):
            idx[phase] = {k: (torch.randint(0, L, (B,), generator=g), torch.randint(0, L - hp.clip_length + 1, (B,), generator=g))
                          for k in ('enc_real', 'enc_fake', 'real', 'fake')}
        noise['d_indices_pre'], noise['d_indices_post'] = idx['pre'], idx['post']
        return noise

    def schedule_sampling_prob(self):
        """savp_model.py:312-322."""
        hp = self.hp
        if hp.schedule_sampling == 'inverse_sigmoid':
            k = hp.schedule_sampling_k
            start = hp.schedule_sampling_steps[0]
            it = float(self.step)
            return 1.0 if it < start else float(k / (k + np.exp((it - start) / k)))
        if hp.schedule_sampling == 'linear':
            start, end = hp.schedule_sampling_steps
            s = min(max(self.step, start), end)
            return 1.0 - float(s - start) / float(end - start)
        return 0.0

    def _gt_mask(self, noise):
        """self.ground_truth of savp_model.py:333-334 for the 2B-batched unroll -> int32 [T1, N] (host)."""
        hp, B, T1 = self.hp, self.B, self.T1
        ns = T1 - hp.context_frames
        halves = []
        keys = ('ground_truth_sampling_enc', 'ground_truth_sampling') if self.nz else ('ground_truth_sampling',)
        for key in keys:
            s = noise.get(key)
            if s is None or self.mode != 'train' or hp.schedule_sampling == 'none':
                s = torch.zeros(ns, B, dtype=torch.bool)
            m = torch.cat([torch.ones(hp.context_frames, B, dtype=torch.bool), torch.as_tensor(s, dtype=torch.bool)], dim=0)
            halves.append(m)
        return torch.cat(halves, dim=1).to(torch.int32)

    def _stage_noise(self, noise):
        """Host -> device copies of the step's random inputs (the only host inputs of the launch sequence)."""
        self.d_gt.copy_(self._gt_mask(noise))
        if self.nz:
            self.d_eps.copy_(torch.as_tensor(noise['eps'], dtype=torch.float32))
            if self.learn_prior:
                self.d_prior_eps.copy_(torch.as_tensor(noise['prior_eps'], dtype=torch.float32))
            else:
                self.d_prior.copy_(torch.as_tensor(noise['prior'], dtype=torch.float32))
        if self.train and 'd_indices_pre' in noise:
            idx = np.zeros((2, len(IDX_KEYS), 2, self.B), dtype=np.int32)
            for pi, ph in enumerate(('d_indices_pre', 'd_indices_post')):
                for ki, k in enumerate(IDX_KEYS):
                    if k in noise[ph]:
                        for which in (0, 1):
                            idx[pi, ki, which] = np.asarray(noise[ph][k][which])
            self.d_idx.copy_(torch.from_numpy(idx))

    # -- forward ---------------------------------------------------------------------------------------------------------
    def prep_generator_weights(self):
        self.gen.prep_weights()
        if self.enc:
            self.enc.prep_weights()
        if self.prior:
            self.prior.prep_weights()

    def forward_generator(self, noise, collect_masks=False):
        """generator_fn (savp_model.py:699-768).  Returns gen [T1, N, H, W, C]: [:, :B] posterior ('_enc'), [:, B:] prior."""
        hp, B, T1 = self.hp, self.B, self.T1
        if noise is not None:
            self._stage_noise(noise)
        gt = self.d_gt
        if self.nz:
            eps = self.d_eps
            z_post = self.enc.forward(self.images_tm, eps, kl=not self.learn_prior, actions=self.actions_tm)
            nzv = self.nz
            copy_view(z_post.reshape(T1, B, nzv), [self.zs_all[:, :B]])
            if self.learn_prior:
                # zs_prior = mu_p + sigma_p * eps for ALL steps (:717-721); KL(posterior || learned prior) (base_model.py:825-828)
                z_prior = self.prior.forward(self.images_tm, self.d_prior_eps, kl=False, actions=self.actions_tm)
                copy_view(z_prior, [self.zs_all[:, B:]])
                K.kl_gauss(self.enc.mu, self.enc.ls_raw, self.prior.mu, self.prior.ls_raw, kl_out=self.enc.kl)
            else:
                # prior half = [posterior z for the first context_frames-1 steps ; N(0,1)]  (:724-725)
                c1 = hp.context_frames - 1
                if c1 > 0:
                    copy_view(z_post[:c1], [self.zs_all[:c1, B:]])
                copy_view(self.d_prior, [self.zs_all[c1:, B:]])
            return self.gen.forward(self.images_n, self.zs_all, gt, collect_masks=collect_masks, actions=self.actions_n,
                                    states=self.states_n)
        return self.gen.forward(self.images_n, None, gt, collect_masks=collect_masks, actions=self.actions_n, states=self.states_n)

    # -- one sess.run(train_op) --------------------------------------------------------------------------------------------
    def _d_clips(self, D, phase, key_real, key_fake, fake_half, lo_real, lo_fake):
        """discriminator_given_video_fn's frame / clip gather (savp_model.py:93-102) into D.clip[lo_real:...] and
        D.clip[lo_fake:...].  Staged indices d_idx[phase][key] = (t_sample[B], t_start[B]); the image discriminator uses
        t_sample, the others t_start.  key_real None: only the fake clips."""
        B = self.B
        which = 0 if D.kind == 'image' else 1
        real_src = self.images_tm[1:self.T]                                       # inputs['images'][1:]
        if key_real is not None:
            K.gather_clips(real_src, D.clip[lo_real:lo_real + B], self.d_idx[phase, IDX_KEYS.index(key_real), which])
        ts_f = self.d_idx[phase, IDX_KEYS.index(key_fake), which]
        K.gather_clips(fake_half, D.clip[lo_fake:lo_fake + B], ts_f)
        return ts_f

    def train_step(self, noise=None, return_grads=False):
        """D Adam update, then G(+E) Adam update against the updated D (base_model.py:486-510).  Returns a dict of
        device scalars (losses) -- nothing is synchronised unless the caller reads them.

        The step = (a) host part: draw / stage the random inputs and the step-dependent scalars into device buffers,
        (b) the launch sequence _step_body(), which has no host input.  (b) is captured after the first eager step(s) and
        replayed afterwards (SAVP_GRAPH=0 keeps it eager): on one GPU as ONE hipGraph; with replicas as a _StepProgram of
        hipGraph segments with the collectives issued by the host between them (a replica issues ~10 host calls per step
        instead of ~2.5k launches: 39 ms of Python / ctypes per 55 ms step otherwise)."""
        hp = self.hp
        if noise is None:
            noise = self.default_noise()
        lr = learning_rate(hp, self.step)
        klw = kl_weight(hp, self.step)
        store = self.store
        self._stage_noise(noise)
        lr_d = store.groups['d'].next_lr_t(lr, hp.beta1, hp.beta2) if self.discs else 0.0
        lr_g = store.groups['g'].next_lr_t(lr, hp.beta1, hp.beta2)
        self.d_scal.copy_(torch.tensor([lr_d, lr_g, klw or 0.0, 0.0], dtype=torch.float32))
        graph_ok = self.use_graph and not return_grads
        if graph_ok and self.graph is not None:
            self.graph.run()
            info = self._fresh_info(self.graph_info, klw)
        elif graph_ok and self.eager_steps >= 1:
            # capture: every conv problem has been tuned and every kernel launched once by the eager step(s)
            prog = _StepProgram(self.device)
            try:
                info = prog.capture(self, lambda: self._step_body(klw, False))
                self.graph, self.graph_info = prog, info
            except Exception as ex:        # capture refused (e.g. an untuned conv problem wanted to time itself): stay eager
                import warnings
                warnings.warn('hipGraph capture of the train step failed (%r); continuing eagerly' % (ex,))
                self.use_graph, self.graph = False, None
                torch.cuda.synchronize()
                info = self._step_body(klw, False)
            else:
                prog.run()
        else:
            info = self._step_body(klw, return_grads)
            self.eager_steps += 1
        self.step += 1
        info = OrderedDict(info)
        info['learning_rate'] = lr
        return info

    @staticmethod
    def _fresh_info(info, klw):
        """The dict returned for a graph replay: same device scalars (they are rewritten by every replay: clone them to keep
        a value across steps), host-side weights of the CURRENT step (the annealed KL weight changes between replays)."""
        out = OrderedDict(info)
        g = OrderedDict(info['g_losses'])
        if 'gen_kl_loss' in g:
            g['gen_kl_loss'] = (g['gen_kl_loss'][0], klw)
        out['g_losses'] = g
        out['d_losses'] = OrderedDict(info['d_losses'])
        return out

    def _step_body(self, klw, return_grads):
        """The launch sequence of one train step (no host inputs: see train_step)."""
        hp, B, T1 = self.hp, self.B, self.T1
        store = self.store
        lb = self.loss_buf
        lb.zero_()
        K.zero_arena(self.device).reset()          # one memset for every reduction workspace of the step
        discs = self.discs
        # The discriminators' weight preparation for the D step (spectral-norm power iteration, packs: ~110 tiny launches, ~1.3 ms)
        # depends on nothing but the discriminator variables: it is forked onto a side stream and runs underneath the latency-bound
        # generator forward; the D step joins it with an event.  Off by default (SAVP_SIDE_PREP=1 enables): measured 74.4 vs 74.3 ms
        # per step -- the side stream's launches compete with the forward chain for the same CUs, like every other overlap tried.
        d_prep_done = None
        if discs and self.dp:
            self._host_op(self.replicas.wait_aux)  # last step's u broadcast (side stream) before anything touches u again
        if discs and self.side_prep:
            if self._prep_stream is None:
                self._prep_stream = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            fork = torch.cuda.Event()
            fork.record(main)
            self._prep_stream.wait_event(fork)
            with torch.cuda.stream(self._prep_stream):
                for D in {id(d['D']): d['D'] for d in discs}.values():
                    D.prep_weights(update_u=True)
                d_prep_done = torch.cuda.Event()
                d_prep_done.record(self._prep_stream)
        self.prep_generator_weights()
        gen = self.forward_generator(None)
        gen_enc, gen_prior = (gen[:, :B], gen[:, B:]) if self.nz else (None, gen)
        info = OrderedDict()
        for d in discs:
            d['fake'] = gen_enc if d['enc'] else gen_prior
        # ---------------- discriminator step ---------------------------------------------------------------------------------
        if discs:
            store.groups['d'].zero_grad()
            prepped = set()
            if d_prep_done is not None:
                torch.cuda.current_stream(self.device).wait_event(d_prep_done)        # join the side-stream preparation
                prepped = set(id(d['D']) for d in discs)
            last_use = {id(d['D']): i for i, d in enumerate(discs)}
            for i, d in enumerate(discs):
                D, w, slot = d['D'], d['w'], d['slot']
                if id(D) not in prepped:
                    D.prep_weights(update_u=True)
                    prepped.add(id(D))
                if w:
                    self._d_clips(D, 0, d['kr'], d['kf'], d['fake'], 0, B)
                    D.forward()
                    r0, r1 = D.rows(0, B)
                    f0, f1 = D.rows(B, 2 * B)
                    K.gan_loss(D.logits[r0:r1], 1.0, w, hp.gan_loss_type, lb[slot:slot + 1], D.dlogits[r0:r1])    # discrim_*_loss real
                    K.gan_loss(D.logits[f0:f1], 0.0, w, hp.gan_loss_type, lb[slot + 1:slot + 2], D.dlogits[f0:f1])  # ... fake
                    D.backward(0, 2 * B, weights=True, data=False)
                if last_use[id(D)] == i:
                    # this network's gradients are final: exchange them on the side stream while the next discriminator runs
                    D.finish_weight_grads()
                    store.groups['d'].fold64()             # float64 accumulators -> fp32 gradients (ParamGroup.grad64)
                    if not return_grads:
                        self._begin_allreduce('d', D.prefix)
            if return_grads:
                info['d_grads'] = {n: store.grad(n).clone() for n in store.names() if store.group_of[n] == 'd'}
            # independent of the D update: clear the generator-side gradient buffers while the last D chunk is in flight
            store.groups['g'].zero_grad()
            self.gen.gen.g.zero_()
            if not hp.joint_gan_optimization:
                self._allreduce('d')
                store.groups['d'].adam_apply(0.0, hp.beta1, hp.beta2, gscale=1.0 / self.world, lr_t_dev=self.d_scal[0:1])
            # joint_gan_optimization (base_model.py:498-505): no control dependency on the D update and no replace_read_ops, i.e. the
            # generator loss is taken against the PRE-update discriminator; D's Adam is applied after the generator step below
            # (its gradient exchange then overlaps the whole generator step)
        else:
            store.groups['g'].zero_grad()
            self.gen.gen.g.zero_()
        # ---------------- generator (+ encoder) step --------------------------------------------------------------------------
        if discs:
            prepped = set()
            for d in discs:
                D, w, slot, is_vae = d['D'], d['w'], d['slot'], d['enc']
                if id(D) not in prepped:
                    D.prep_weights(update_u=False)                 # updated W, same pre-assign u
                    prepped.add(id(D))
                if not w:
                    continue
                wf_cd = hp.vae_gan_feature_cdist_weight if is_vae else hp.gan_feature_cdist_weight
                wf_l2 = hp.vae_gan_feature_l2_weight if is_vae else hp.gan_feature_l2_weight
                if wf_cd or wf_l2:
                    ts_f = self._d_clips(D, 1, d['kr'], d['kf'], d['fake'], 0, B)
                    D.forward()
                    lo, hi = B, 2 * B
                    r0, r1 = D.rows(0, B)
                    f0, f1 = D.rows(lo, hi)
                    for L in D.layers:
                        if wf_cd:       # losses.cosine_distance over the channel axis (base_model.py:794-797,821-824)
                            K.cosine_distance(L['y'][f0:f1], L['y'][r0:r1], wf_cd, lb[slot + 3:slot + 4], L['dy'][f0:f1])
                        else:
                            L['dy'][f0:f1].zero_()
                        if wf_l2:       # losses.l2_loss between fake and real features (base_model.py:790-793,817-820)
                            K.lp_loss(L['y'][f0:f1], L['y'][r0:r1], wf_l2, lb[slot + 4:slot + 5], L['dy'][f0:f1], p2=True)
                else:
                    ts_f = self._d_clips(D, 1, None, d['kf'], d['fake'], 0, 0)
                    D.forward(n=B)
                    lo, hi = 0, B
                    f0, f1 = D.rows(lo, hi)
                K.gan_loss(D.logits[f0:f1], 1.0, w, hp.gan_loss_type, lb[slot + 2:slot + 3], D.dlogits[f0:f1])   # gen_*_gan_loss
                D.backward(lo, hi, weights=False, data=True, feature_grads=bool(wf_cd or wf_l2))
                gfake = self.gen.gen.g[:, :B] if (is_vae and self.nz) else (self.gen.gen.g[:, B:] if self.nz else self.gen.gen.g)
                K.gather_clips(gfake, D.dclip[lo:hi], ts_f, adjoint=True)
        target = self.images_tm[1:self.T]
        pred = gen_enc if self.nz else gen_prior
        dpred = self.gen.gen.g[:, :B] if self.nz else self.gen.gen.g
        if hp.l1_weight:
            K.lp_loss(pred, target, hp.l1_weight, lb[-2:-1], dpred, p2=False)
        if hp.l2_weight:
            K.lp_loss(pred, target, hp.l2_weight, lb[-1:], dpred, p2=True)
        state_w = getattr(hp, 'state_weight', 0) if self.ns else 0
        if state_w:
            # gen_state_loss = l2_loss(gen_states(_enc), inputs['states'][1:]) (base_model.py:758-762): the posterior half where there is one
            gs = self.gen.gen_states
            gs.g.zero_()
            K.lp_loss(gs.v[:, :B], self.states_tm[1:self.T], state_w, lb[-3:-2], gs.g[:, :B], p2=True)
        tv_w = getattr(hp, 'tv_weight', 0)
        # gen_tv_loss (base_model.py:763-769) is taken step by step inside BPTT, where each step's flow gradient is complete
        self.gen.tv = (tv_w, B, lb[-4:-3]) if tv_w else None
        dzs = self.gen.backward(state_grad=bool(state_w))
        store.groups['g'].fold64()                         # float64 accumulators of the cell's norms / z-LSTM -> fp32 gradients
        if self.nz and not return_grads:
            # the generator cell's gradients are final after BPTT: exchange them under the encoder's backward pass
            self._begin_allreduce('g', self.gen.prefix_root)
        if self.nz:
            c1 = hp.context_frames - 1
            copy_view(dzs[:, :B], [self.dz_post])
            if self.learn_prior:
                copy_view(dzs[:, B:], [self.dz_prior])
                self.enc.reparam_backward(self.dz_post)
                self.prior.reparam_backward(self.dz_prior)
                if hp.kl_weight:
                    K.kl_gauss(self.enc.mu, self.enc.ls_raw, self.prior.mu, self.prior.ls_raw, klw=klw or 0.0, klw_dev=self.d_scal[2:3],
                               grads=(self.enc.dmu, self.enc.dls, self.prior.dmu, self.prior.dls))
                self.enc.backward_network()
                self.prior.backward_network()
            else:
                if c1 > 0:
                    add_views([dzs[:c1, B:]], self.dz_post[:c1])
                self.enc.backward(self.dz_post, klw, kl_weight_dev=self.d_scal[2:3])
        store.groups['g'].fold64()                         # ... and the encoder's
        if return_grads:
            info['g_grads'] = {n: store.grad(n).clone() for n in store.names() if store.group_of[n] == 'g'}
        self._allreduce('g')
        store.groups['g'].adam_apply(0.0, hp.beta1, hp.beta2, gscale=1.0 / self.world, lr_t_dev=self.d_scal[1:2])
        if discs and hp.joint_gan_optimization:
            self._allreduce('d')
            store.groups['d'].adam_apply(0.0, hp.beta1, hp.beta2, gscale=1.0 / self.world, lr_t_dev=self.d_scal[0:1])
        for D in {id(d['D']): d['D'] for d in discs}.values():
            D.commit_u()
        if discs and self.dp:
            self._host_op(self.replicas.sync_aux)    # u vectors: bit-identical replicas (parallel.ReplicaGroup.sync_aux)
        # ---- loss bookkeeping (device scalars; base_model.py:733-852) ----------------------------------------------------------
        d_losses, g_losses = OrderedDict(), OrderedDict()
        lb = lb.float()                            # the float64 accumulators, rounded once
        for d in discs:
            w, slot, name = d['w'], d['slot'], d['name']
            if not w:
                continue
            d_losses['discrim_%s_loss' % name] = (lb[slot] + lb[slot + 1], w)
            g_losses['gen_%s_loss' % name] = (lb[slot + 2], w)
            wf_cd = hp.vae_gan_feature_cdist_weight if d['enc'] else hp.gan_feature_cdist_weight
            if wf_cd:
                g_losses['gen_%s_feature_cdist_loss' % name] = (lb[slot + 3], wf_cd)
            wf_l2 = hp.vae_gan_feature_l2_weight if d['enc'] else hp.gan_feature_l2_weight
            if wf_l2:
                g_losses['gen_%s_feature_l2_loss' % name] = (lb[slot + 4], wf_l2)
        if hp.l1_weight:
            g_losses['gen_l1_loss'] = (lb[-2], hp.l1_weight)
        if hp.l2_weight:
            g_losses['gen_l2_loss'] = (lb[-1], hp.l2_weight)
        if state_w:
            g_losses['gen_state_loss'] = (lb[-3], state_w)
        if tv_w:
            g_losses['gen_tv_loss'] = (lb[-4], tv_w)
        if self.nz and hp.kl_weight:
            g_losses['gen_kl_loss'] = (self.enc.kl.float()[0], klw)
        info['d_losses'], info['g_losses'] = d_losses, g_losses
        info['d_loss'] = sum(l * w for l, w in d_losses.values()) if d_losses else torch.zeros((), device=self.device)
        # the annealed KL weight is a device scalar here (the graph is replayed with the weight of the current step)
        info['g_loss'] = sum(l * (self.d_scal[2] if k == 'gen_kl_loss' else w) for k, (l, w) in g_losses.items())
        return info

    # -- inference (scripts/generate.py:166: model.outputs['gen_images']) ------------------------------------------------------
    def generate(self, noise=None, collect_masks=False):
        """One prior (and posterior) unroll of the generator on the staged images: gen [T1, N, H, W, C] (a buffer of the engine: the
        next call overwrites it).  Like the train step, the launch sequence (weight preparation + unroll, ~1 k launches) has no host
        input once the noise is staged, so from the second call on it is replayed as ONE hipGraph (SAVP_INFER_GRAPH=0: eager): the
        100 unrolls of an evaluation (eval_outputs_and_metrics) cost 100 graph launches instead of ~10^5 host calls."""
        if noise is None:
            noise = self.default_noise()
        if not (self.infer_graph and not collect_masks and K.fused_ok()):
            self.prep_generator_weights()
            return self.forward_generator(noise, collect_masks=collect_masks)
        self._stage_noise(noise)

        def body():
            K.zero_arena(self.device).reset()
            self.prep_generator_weights()
            return self.forward_generator(None)
        if self.gen_graph is not None:
            self.gen_graph.run()
            return self.gen_graph_out
        if self.gen_eager >= 1:              # every conv problem tuned and every kernel loaded by the eager call(s)
            prog = _StepProgram(self.device)
            try:
                out = prog.capture(self, body)
            except Exception as ex:
                import warnings
                warnings.warn('hipGraph capture of the generator unroll failed (%r); continuing eagerly' % (ex,))
                self.infer_graph = False
                torch.cuda.synchronize()
                return body()
            self.gen_graph, self.gen_graph_out = prog, out
            prog.run()
            return out
        self.gen_eager += 1
        return body()

    # -- evaluation: metrics_fn / eval_outputs_and_metrics_fn (base_model.py:113-227) ------------------------
    METRICS = ('psnr', 'mse', 'ssim')            # base_model.py:119-124 without lpips (external AlexNet weights)

    def _frame_metrics(self, pred, buf):
        """psnr / mse / ssim [T_future, B] of the future frames of pred [T1, B, H, W, C] against the staged images."""
        hp = self.hp
        fut = self.T - hp.context_frames
        target = self.images_tm[self.T - fut:]
        p = pred[self.T1 - fut:]
        K.frame_mse_psnr(target, p, mse=buf['mse'], psnr=buf['psnr'])
        K.frame_ssim(target, p, buf['ssim'])
        return buf

    def metrics(self, gen=None):
        """metrics_fn (base_model.py:113-130): mean psnr / mse / ssim over the future frames of the prior unroll."""
        gen = self.generate() if gen is None else gen
        fut = self.T - self.hp.context_frames
        buf = {k: torch.empty(fut, self.B, device=self.device) for k in self.METRICS}
        self._frame_metrics(gen[:, self.B:] if self.nz else gen, buf)
        return OrderedDict((k, buf[k].mean()) for k in self.METRICS)

    def eval_outputs_and_metrics(self, num_samples=100, noises=None):
        """eval_outputs_and_metrics_fn (base_model.py:132-227): draw num_samples prior unrolls; per metric keep, for every
        sequence, the sample whose time-mean is smallest / largest, and the running mean.  noises: optional list of noise dicts
        (one per sample; default = fresh draws).  Returns (eval_outputs, eval_metrics) with the reference's keys, time-major;
        the lpips / eval_diversity entries need external network weights and are not produced."""
        hp, B, dev = self.hp, self.B, self.device
        fut = self.T - hp.context_frames
        outs, mets = OrderedDict(), OrderedDict()
        outs['eval_images'] = self.images_tm
        buf = {k: torch.empty(fut, B, device=dev) for k in self.METRICS}
        if not self.nz:                                               # deterministic model (:163-168)
            gen = self.generate(noises[0] if noises else None)
            self._frame_metrics(gen, buf)
            for k in self.METRICS:
                for sfx in ('min', 'avg', 'max'):
                    mets['eval_%s/%s' % (k, sfx)] = buf[k]
            outs['eval_gen_images'] = gen
            return outs, mets
        shape = (self.T1, B) + tuple(self.image_shape)
        st = {}
        for k in self.METRICS:                                         # initializer (:201-210)
            st[k] = dict(min=torch.full((fut, B), float('inf'), device=dev), sum=torch.zeros(fut, B, device=dev),
                         max=torch.full((fut, B), float('-inf'), device=dev), gmin=torch.zeros(shape, device=dev),
                         gsum=torch.zeros(shape, device=dev), gmax=torch.zeros(shape, device=dev))
        cmin = torch.zeros(B, dtype=torch.int32, device=dev)
        cmax = torch.zeros(B, dtype=torch.int32, device=dev)
        for s_i in range(num_samples):                                 # accum_gen_images_and_metrics_fn (:176-198)
            gen = self.generate(noises[s_i] if noises else self.default_noise(
                torch.Generator().manual_seed(self._noise_seed(self.step, stream=1 + s_i))))
            prior = gen[:, B:]                                         # the prior unroll ('gen_images')
            self._frame_metrics(prior, buf)
            for k in self.METRICS:
                a = st[k]
                K.eval_accumulate(buf[k], a['min'], a['sum'], a['max'], cmin, cmax)
                K.select_batch(cmin, prior, a['gmin'])
                K.select_batch(cmax, prior, a['gmax'])
                K.select_batch(None, prior, a['gsum'], mode=1)
        inv = 1.0 / float(num_samples)
        for k in self.METRICS:                                         # (:215-221)
            a = st[k]
            K.axpby(inv, a['gsum'].reshape(-1), 0.0, a['gsum'].reshape(-1), a['gsum'].reshape(-1))
            K.axpby(inv, a['sum'].reshape(-1), 0.0, a['sum'].reshape(-1), a['sum'].reshape(-1))
            outs['eval_gen_images_%s/min' % k] = a['gmin']
            outs['eval_gen_images_%s/avg' % k] = a['gsum']
            outs['eval_gen_images_%s/max' % k] = a['gmax']
            mets['eval_%s/min' % k] = a['min']
            mets['eval_%s/avg' % k] = a['sum']
            mets['eval_%s/max' % k] = a['max']
        return outs, mets


# ---------------------------------------------------------------------------------------------------------------------
# plug-in functions with the reference's signatures
# ---------------------------------------------------------------------------------------------------------------------
_ENGINES = {}


def refuse_conditioning_inputs(inputs):
    """inputs['pix_distribs'] (savp_model.py:252-255,408-410,647-653: designated-pixel distributions pushed through the predicted
    transformations, a visual-servoing output no training loss reads) is not on the MI355X path: a batch that carries it is refused
    instead of being run as if it were not there.  'actions' / 'states' are handled (cond_of)."""
    if isinstance(inputs, dict) and inputs.get('pix_distribs') is not None:
        raise NotImplementedError("inputs['pix_distribs'] given: the pixel-distribution outputs of SAVPCell (reference savp_model.py:"
                                  "408-410,647-653) are not on the MI355X hot path; drop the key to proceed")


def cond_of(inputs):
    """(n_actions, n_states) of an inputs dict -- the structure SAVPCell reads off `inputs` (savp_model.py:256-257,291-292,416-422)."""
    if not isinstance(inputs, dict):
        return (0, 0)
    a, s_ = inputs.get('actions'), inputs.get('states')
    return (int(a.shape[-1]) if a is not None else 0, int(s_.shape[-1]) if s_ is not None else 0)


def _engine_for(inputs, mode, hparams, engine=None):
    refuse_conditioning_inputs(inputs)
    images = inputs['images']
    if engine is not None:
        return engine
    T, B = images.shape[:2]
    cond = cond_of(inputs)
    key = (id(hparams), mode, tuple(images.shape), cond)
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _ENGINES[key] = SAVPEngine(hparams, tuple(images.shape[2:]), B, mode='train' if mode == 'train' else 'test',
                                         device=images.device, cond=cond)
    return eng


def posterior_fn(inputs, hparams, engine=None, noise=None):
    """savp_model.py:21-51.  inputs['images'] time-major [T,B,H,W,C] on the device."""
    eng = _engine_for(inputs, 'test', hparams, engine)
    eng.set_images(inputs, time_major=True)
    eng.enc.prep_weights()
    nz = hparams.nz
    eps = (noise or {}).get('eps')
    if eps is None:
        eps = torch.zeros(eng.T1, eng.B, nz)
    eng.enc.forward(eng.images_tm, eps.to(eng.device, torch.float32), actions=eng.actions_tm)
    return {'zs_mu': eng.enc.mu, 'zs_log_sigma_sq': eng.enc.ls}


def prior_fn(inputs, hparams, engine=None, noise=None):
    """savp_model.py:54-85 (learn_prior=True).  inputs['images'] time-major [T,B,H,W,C] on the device."""
    if not hparams.learn_prior:
        raise ValueError('prior_fn needs hparams.learn_prior=True (the prior network is not instantiated otherwise)')
    eng = _engine_for(inputs, 'test', hparams, engine)
    eng.set_images(inputs, time_major=True)
    eng.prior.prep_weights()
    eps = (noise or {}).get('prior_eps')
    if eps is None:
        eps = torch.zeros(eng.T1, eng.B, hparams.nz)
    eng.prior.forward(eng.images_tm, eps.to(eng.device, torch.float32), kl=False, actions=eng.actions_tm)
    return {'zs_mu': eng.prior.mu, 'zs_log_sigma_sq': eng.prior.ls}


def generator_fn(inputs, mode, hparams, engine=None, noise=None, samples=False):
    """savp_model.py:699-768.  samples=True (or injected draws noise['samples_prior'] [T-context, S, B, nz] / ['samples_prior_eps']
    [T-1, S, B, nz]) also runs the visualisation unroll of :745-767: hparams.num_samples draws from the prior per sequence ->
    gen_images_samples [T-1, B, H, W, C, S] and their mean gen_images_samples_avg (one more unroll per draw: not on the train path)."""
    eng = _engine_for(inputs, mode, hparams, engine)
    eng.set_images(inputs, time_major=True)
    if noise is None:
        noise = eng.default_noise()
    eng.prep_generator_weights()
    gen = eng.forward_generator(noise, collect_masks=True)
    B, C, M = eng.B, eng.image_shape[2], eng.gen.M
    g = eng.gen
    timgs = g.maskin.v[..., hparams.ngf:hparams.ngf + M * C].reshape(g.T1, g.N, g.H, g.W, M, C).permute(0, 1, 2, 3, 5, 4)   # [..., C, M]
    masks = g.masks.reshape(g.T1, g.N, g.H, g.W, 1, M)
    outputs = OrderedDict()
    lo = B if eng.nz else 0
    outputs['gen_images'] = gen[:, lo:]
    outputs['transformed_images'] = timgs[:, lo:]
    outputs['masks'] = masks[:, lo:]
    gt = eng._gt_mask(noise)
    outputs['ground_truth_sampling_mean'] = gt[hparams.context_frames:, lo:].float().mean()
    if eng.ns:
        outputs['gen_states'] = g.gen_states.v[:, lo:]             # savp_model.py:666-667
    if eng.learn_prior:
        outputs['zs_mu_prior'] = eng.prior.mu                    # savp_model.py:735 (keys get the '_prior' suffix)
        outputs['zs_log_sigma_sq_prior'] = eng.prior.ls
    if eng.nz:
        outputs['zs_mu_enc'] = eng.enc.mu
        outputs['zs_log_sigma_sq_enc'] = eng.enc.ls
        outputs['gen_images_enc'] = gen[:, :B]
        outputs['transformed_images_enc'] = timgs[:, :B]
        outputs['masks_enc'] = masks[:, :B]
        outputs['ground_truth_sampling_mean_enc'] = gt[hparams.context_frames:, :B].float().mean()
        if eng.ns:
            outputs['gen_states_enc'] = g.gen_states.v[:, :B]
    if eng.nz and (samples or 'samples_prior' in noise or 'samples_prior_eps' in noise):
        # the prior half of one more 2B unroll per draw (the posterior half repeats the posterior unroll above: same eps, same schedule)
        key = 'samples_prior_eps' if eng.learn_prior else 'samples_prior'
        draws = noise.get(key)
        if draws is None:
            S = int(hparams.num_samples)
            gen_ = torch.Generator().manual_seed(eng._noise_seed(eng.step, stream=3))
            T0 = eng.T1 if eng.learn_prior else eng.T - hparams.context_frames
            draws = torch.randn(T0, S, B, eng.nz, generator=gen_)
        draws = torch.as_tensor(draws, dtype=torch.float32)
        S = draws.shape[1]
        gts = noise.get('samples_ground_truth_sampling')
        outs = []
        for i in range(S):
            n_i = dict(noise)
            n_i['prior_eps' if eng.learn_prior else 'prior'] = draws[:, i]
            n_i['ground_truth_sampling'] = None if gts is None else torch.as_tensor(gts)[:, i * B:(i + 1) * B]
            outs.append(eng.forward_generator(n_i)[:, B:].clone())
        outputs['gen_images_samples'] = torch.stack(outs, dim=-1)                            # savp_model.py:762-765
        outputs['gen_images_samples_avg'] = outputs['gen_images_samples'].mean(dim=-1)
        eng.forward_generator(noise, collect_masks=True)          # leave the engine's buffers as the main unrolls wrote them (the views above)
    return outputs


def discriminator_fn(inputs, outputs, mode, hparams, engine=None, noise=None):
    """savp_model.py:129-166: runs the video discriminator(s) on (real, fake) clips; keys
    discrim_video_sn_{logits,feature%d}_{real,fake,enc_real,enc_fake}."""
    eng = _engine_for(inputs, mode, hparams, engine)
    if not eng.has_d:
        return OrderedDict()
    if noise is None:
        noise = eng.default_noise()
    eng._stage_noise(noise)
    eng.wait_aux()                 # D.prep_weights reads u: order behind a u broadcast still in flight on the side stream
    B = eng.B
    out = OrderedDict()
    for d in eng.discs:
        D, sfx = d['D'], ('_enc' if d['enc'] else '')
        fake = outputs['gen_images_enc'] if d['enc'] else outputs['gen_images']
        D.prep_weights(update_u=False)
        eng._d_clips(D, 0, d['kr'], d['kf'], fake, 0, B)
        D.forward()
        for part, lo in (('real', 0), ('fake', B)):
            r0, r1 = D.rows(lo, lo + B)
            out['discrim_%s_logits%s_%s' % (d['infix'], sfx, part)] = D.logits[r0:r1].clone()
            for i, f in enumerate(D.features()):
                fv = f[r0:r1]
                out['discrim_%s_feature%d%s_%s' % (d['infix'], i, sfx, part)] = fv.transpose(0, 1) if D.kind == 'video' else fv
    return out


class SAVPVideoPredictionModel(VideoPredictionModel):
    """savp_model.py:771-855."""

    def __init__(self, *args, **kwargs):
        super(SAVPVideoPredictionModel, self).__init__(generator_fn, discriminator_fn, *args, **kwargs)
        if self.mode != 'train':
            self.discriminator_fn = None
        self.deterministic = not self.hparams.nz
        self.engine = None

    def get_default_hparams_dict(self):
        return savp_defaults()

    def parse_hparams(self, hparams_dict, hparams):
        hparams_dict = dict(hparams_dict or {})
        for k in SAVP_DEPRECATED_KEYS:                     # backwards compatibility (savp_model.py:826-845)
            hparams_dict.pop(k, None)
        return super(SAVPVideoPredictionModel, self).parse_hparams(hparams_dict, hparams)

    def build_graph(self, inputs, values=None, seed=4, device='cuda:0'):
        """inputs: {'images': [B,T,H,W,C] device tensor, optionally 'actions': [B,T-1,na], 'states': [B,T,ns]} (batch-major like the
        reference's dataset iterator).  As in the reference the input STRUCTURE is fixed here: a model built with actions / states
        expects them in every later batch."""
        refuse_conditioning_inputs(inputs)
        super(SAVPVideoPredictionModel, self).build_graph(inputs)
        images = inputs['images']
        B = images.shape[0]
        self.engine = SAVPEngine(self.hparams, tuple(images.shape[2:]), B, mode=self.mode, values=values, seed=seed,
                                 device=device, cond=cond_of(inputs))
        self.saveable_variables = self.engine.store.names()
        self.post_init_ops = []
        self.outputs = {}
        return self

    @property
    def global_step(self):
        return self.engine.step if self.engine else 0

    def train_step(self, inputs=None, noise=None):
        """One ``sess.run(model.train_op)``."""
        if inputs is not None:
            refuse_conditioning_inputs(inputs)
            self.inputs = inputs
        self.engine.set_images(self.inputs)
        info = self.engine.train_step(noise)
        self.d_losses, self.g_losses = info['d_losses'], info['g_losses']
        self.d_loss, self.g_loss = info['d_loss'], info['g_loss']
        return info

    def generate(self, inputs=None, noise=None):
        """Fills self.outputs['gen_images'] batch-major [B,T-1,H,W,C] (scripts/generate.py:166)."""
        if inputs is not None:
            refuse_conditioning_inputs(inputs)
            self.inputs = inputs
        self.engine.set_images(self.inputs)
        gen = self.engine.generate(noise)
        lo = self.engine.B if self.engine.nz else 0
        self.outputs['gen_images'] = gen[:, lo:].transpose(0, 1)
        if self.engine.nz:
            self.outputs['gen_images_enc'] = gen[:, :self.engine.B].transpose(0, 1)
        return self.outputs

    def metrics_fn(self, inputs=None, outputs=None):
        """base_model.py:113-130 on the current inputs (psnr / mse / ssim; lpips needs external weights)."""
        if inputs is not None:
            self.inputs = inputs
            self.engine.set_images(self.inputs)
        self.metrics = self.engine.metrics()
        return self.metrics

    def eval_outputs_and_metrics_fn(self, inputs=None, outputs=None, num_samples=None, num_samples_for_diversity=None,
                                    parallel_iterations=None, noises=None):
        """base_model.py:132-227: best / mean / worst of num_samples prior samples per sequence (time-major tensors)."""
        if inputs is not None:
            self.inputs = inputs
            self.engine.set_images(self.inputs)
        self.eval_outputs, self.eval_metrics = self.engine.eval_outputs_and_metrics(num_samples or self.eval_num_samples, noises)
        return self.eval_outputs, self.eval_metrics

    def restore(self, checkpoints, restore_to_checkpoint_mapping=None):
        """savp_model.py:848-855 / base_model.py:229-247: `checkpoints` is a TensorFlow V2 checkpoint directory or prefix (or a
        list of them, each holding a subset of the variables), read by video_prediction_amd.checkpoint without TensorFlow; a
        {variable name: array} dict is accepted as well.  Names fall back from `savp_cell` to `dna_cell` like the reference."""
        self.engine.wait_aux()                    # a u broadcast of the last step may still be writing the 'aux' arena
        if isinstance(checkpoints, dict):
            self.engine.store.load(checkpoints)
            return
        from .. import checkpoint as CK

        def mapping(name, names):
            name = name.split(':')[0]
            if name not in names:
                name = name.replace('savp_cell', 'dna_cell')
            return name
        log = lambda head, items: print(head + '\n' + '\n'.join('     ' + i for i in items))
        store = self.engine.store
        slot_names = self._optimizer_slot_names()
        wanted = store.names() + ['global_step'] + list(slot_names)
        vals = CK.restore_values(checkpoints, wanted, restore_to_checkpoint_mapping or mapping, log=log, optional=set(slot_names))
        step = vals.pop('global_step', None)
        slots = {k: vals.pop(k) for k in list(vals) if k in slot_names}
        store.load(vals)
        if step is not None:
            self.engine.step = int(step)
        self._load_optimizer_slots(slots, step)

    # -- optimizer state (base_model.py:229-247,512-515: saveable_variables include the Adam slots and beta powers) -----------
    def _optimizer_slot_names(self):
        """{checkpoint name: (group, kind, variable name)} in tf.train.AdamOptimizer's naming: slots '<var>/Adam' (m) and
        '<var>/Adam_1' (v); the non-slot accumulators 'beta1_power' / 'beta2_power' of the optimizer whose apply_gradients is
        built first (D when there is one, base_model.py:489-496) and 'beta1_power_1' / 'beta2_power_1' of the second (G).
        (Names inferred from TF's slot creator; unverifiable offline -- restore treats every one of them as optional.)"""
        store = self.engine.store
        out = OrderedDict()
        for n in store.names():
            grp = store.group_of[n]
            if grp == 'aux':
                continue
            out[n + '/Adam'] = (grp, 'm', n)
            out[n + '/Adam_1'] = (grp, 'v', n)
        order = (['d'] if self.engine.discs else []) + ['g']
        for i, grp in enumerate(order):
            sfx = '' if i == 0 else '_%d' % i
            out['beta1_power' + sfx] = (grp, 'b1', None)
            out['beta2_power' + sfx] = (grp, 'b2', None)
        return out

    def _load_optimizer_slots(self, slots, step):
        import math
        import warnings
        store, hp = self.engine.store, self.hparams
        names = self._optimizer_slot_names()
        seen = {'d': 0, 'g': 0}
        t_from_power = {}
        for k, val in slots.items():
            grp, kind, var = names[k]
            G = store.groups[grp]
            if kind in ('m', 'v'):
                dst = G.arena.view_of(G.m if kind == 'm' else G.v, var)
                dst.copy_(torch.as_tensor(np.asarray(val, dtype=np.float32)).reshape(dst.shape))
                seen[grp] += 1
            elif kind == 'b1' and 0.0 < float(val) < 1.0 and 0.0 < hp.beta1 < 1.0:
                t_from_power[grp] = int(round(math.log(float(val)) / math.log(hp.beta1))) - 1      # power = beta^(t+1) after t steps
        for grp in ('d', 'g'):
            G = store.groups[grp]
            ntrain = sum(1 for n in store.names() if store.group_of[n] == grp)
            if not ntrain or self.mode != 'train':
                continue
            if seen[grp] == 2 * ntrain:
                G.t = max(t_from_power.get(grp, int(step) if step is not None else 0), 0)
            else:
                warnings.warn('checkpoint holds %d of %d Adam slots of group %r: optimizer state RESET (moments zero, bias '
                              'correction restarts)' % (seen[grp], 2 * ntrain, grp))
                G.m.zero_()
                G.v.zero_()
                G.t = 0

    def save(self, prefix):
        """tf.train.Saver.save equivalent: all variables (TF names), global_step and the optimizer state (Adam slots + beta
        powers of both optimizers) as a V2 checkpoint at `prefix`."""
        from .. import checkpoint as CK
        store, hp = self.engine.store, self.hparams
        self.engine.wait_aux()                     # the u broadcast of the last step runs on the side stream
        vals = store.to_numpy()
        vals['global_step'] = np.asarray(self.engine.step, dtype=np.int64)
        if self.mode == 'train':
            for k, (grp, kind, var) in self._optimizer_slot_names().items():
                G = store.groups[grp]
                if kind in ('m', 'v'):
                    vals[k] = G.arena.view_of(G.m if kind == 'm' else G.v, var).detach().cpu().numpy().copy()
                else:
                    beta = hp.beta1 if kind == 'b1' else hp.beta2
                    vals[k] = np.asarray(beta ** (G.t + 1), dtype=np.float32)       # TF: initial value beta, multiplied per step
        CK.write_checkpoint(prefix, vals)
