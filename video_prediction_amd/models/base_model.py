"""Model-class API of the reference (models/base_model.py) without TensorFlow.

``BaseVideoPredictionModel`` / ``VideoPredictionModel`` keep the constructor signature, hparams handling
(defaults -> JSON dict -> ``k=v`` string, base_model.py:99-109), mode validation (:37-38), the required
``context_frames`` / ``sequence_length`` (:54-59), the learning-rate and KL-weight schedules (:286-319) and the
loss-weight bookkeeping (:733-852).  The TF graph/session machinery is replaced by an explicit ``train_step``.
"""
import functools
import itertools
from collections import OrderedDict

from ..hparams import HParams
from . import hparam_defaults


class BaseVideoPredictionModel(object):
    def __init__(self, mode='train', hparams_dict=None, hparams=None, num_gpus=None, eval_num_samples=100,
                 eval_num_samples_for_diversity=10, eval_parallel_iterations=1):
        if mode not in ('train', 'test'):
            raise ValueError('mode must be train or test, but %s given' % mode)
        self.mode = mode
        self.num_gpus = num_gpus
        self.eval_num_samples = eval_num_samples
        self.eval_num_samples_for_diversity = eval_num_samples_for_diversity
        self.eval_parallel_iterations = eval_parallel_iterations
        self.hparams = self.parse_hparams(hparams_dict, hparams)
        if self.hparams.context_frames == -1:
            raise ValueError('Invalid context_frames %r. It might have to be '
                             'specified.' % self.hparams.context_frames)
        if self.hparams.sequence_length == -1:
            raise ValueError('Invalid sequence_length %r. It might have to be '
                             'specified.' % self.hparams.sequence_length)
        self.deterministic = True
        self.inputs = None
        self.gen_images = None
        self.outputs = None
        self.metrics = None
        self.eval_outputs = None
        self.eval_metrics = None
        self.saveable_variables = None
        self.post_init_ops = None

    def get_default_hparams_dict(self):
        return hparam_defaults.base_defaults()

    def get_default_hparams(self):
        return HParams(**self.get_default_hparams_dict())

    def parse_hparams(self, hparams_dict, hparams):
        parsed_hparams = self.get_default_hparams().override_from_dict(hparams_dict or {})
        if hparams:
            if not isinstance(hparams, (list, tuple)):
                hparams = [hparams]
            for hparam in hparams:
                parsed_hparams.parse(hparam)
        return parsed_hparams

    def build_graph(self, inputs):
        self.inputs = inputs


def learning_rate(hp, step):
    """base_model.py:286-301."""
    if any(hp.lr_boundaries):
        vals = [hp.lr * 0.1 ** i for i in range(len(hp.lr_boundaries) + 1)]
        return vals[sum(1 for b in hp.lr_boundaries if step > b)]
    elif any(hp.decay_steps):
        start_step, end_step = hp.decay_steps
        if start_step == end_step:
            schedule = 0.0 if step < start_step else 1.0
        else:
            s = min(max(step, start_step), end_step)
            schedule = float(s - start_step) / float(end_step - start_step)
        return hp.lr + (hp.end_lr - hp.lr) * schedule
    return hp.lr


def kl_weight(hp, step):
    """base_model.py:303-319."""
    if not hp.kl_weight:
        return None
    if hp.kl_anneal == 'none':
        return hp.kl_weight
    if hp.kl_anneal == 'sigmoid':
        import math
        k = hp.kl_anneal_k
        if k == -1.0:
            raise ValueError('Invalid kl_anneal_k %d when kl_anneal is sigmoid.' % k)
        return hp.kl_weight / (1 + k * math.exp(-float(step) / k))
    if hp.kl_anneal == 'linear':
        start_step, end_step = hp.kl_anneal_steps
        s = min(max(step, start_step), end_step)
        return hp.kl_weight * float(s - start_step) / float(end_step - start_step)
    raise NotImplementedError


class VideoPredictionModel(BaseVideoPredictionModel):
    def __init__(self, generator_fn, discriminator_fn=None, generator_scope='generator',
                 discriminator_scope='discriminator', aggregate_nccl=False, mode='train', hparams_dict=None,
                 hparams=None, **kwargs):
        super(VideoPredictionModel, self).__init__(mode, hparams_dict, hparams, **kwargs)
        self.generator_fn = functools.partial(generator_fn, mode=self.mode, hparams=self.hparams)
        self.discriminator_fn = functools.partial(discriminator_fn, mode=self.mode, hparams=self.hparams) \
            if discriminator_fn else None
        self.generator_scope = generator_scope
        self.discriminator_scope = discriminator_scope
        self.aggregate_nccl = aggregate_nccl
        self.gen_images_enc = None
        self.g_losses = None
        self.d_losses = None
        self.g_loss = None
        self.d_loss = None
        self.train_op = None

    @property
    def learning_rate(self):
        return learning_rate(self.hparams, self.global_step)

    @property
    def kl_weight(self):
        return kl_weight(self.hparams, self.global_step)

    global_step = 0

    def get_default_hparams_dict(self):
        return hparam_defaults.trainable_defaults()
