"""Model-class API of the reference (models/base_model.py) without TensorFlow.

``BaseVideoPredictionModel`` / ``VideoPredictionModel`` keep the constructor signature, hparams handling
(defaults -> JSON dict -> ``k=v`` string, base_model.py:99-109), mode validation (:37-38), the required
``context_frames`` / ``sequence_length`` (:54-59), the learning-rate and KL-weight schedules (:286-319) and the
loss-weight bookkeeping (:733-852).  The TF graph/session machinery is replaced by an explicit ``train_step``.
"""
import functools

from ..hparams import HParams
from . import hparam_defaults


# attributes the reference's constructors create empty (base_model.py:60-68, 221-229); callers and subclasses read them
_BASE_SLOTS = ('inputs', 'gen_images', 'outputs', 'metrics', 'eval_outputs', 'eval_metrics', 'saveable_variables', 'post_init_ops')
_TRAINABLE_SLOTS = ('gen_images_enc', 'g_losses', 'd_losses', 'g_loss', 'd_loss', 'train_op')
_REQUIRED_HPARAMS = ('context_frames', 'sequence_length')      # -1 in the defaults = "the dataset has to say" (base_model.py:54-59)


class BaseVideoPredictionModel(object):
    def __init__(self, mode='train', hparams_dict=None, hparams=None, num_gpus=None, eval_num_samples=100,
                 eval_num_samples_for_diversity=10, eval_parallel_iterations=1):
        if mode != 'train' and mode != 'test':
            raise ValueError('mode must be train or test, but %s given' % mode)
        self.mode, self.num_gpus = mode, num_gpus
        self.eval_num_samples, self.eval_num_samples_for_diversity = eval_num_samples, eval_num_samples_for_diversity
        self.eval_parallel_iterations = eval_parallel_iterations
        self.hparams = self.parse_hparams(hparams_dict, hparams)
        for name in _REQUIRED_HPARAMS:
            value = getattr(self.hparams, name)
            if value == -1:
                raise ValueError('Invalid %s %r. It might have to be specified.' % (name, value))
        self.deterministic = True
        for slot in _BASE_SLOTS:
            setattr(self, slot, None)

    def get_default_hparams_dict(self):
        return hparam_defaults.base_defaults()

    def get_default_hparams(self):
        return HParams(**self.get_default_hparams_dict())

    def parse_hparams(self, hparams_dict, hparams):
        """defaults <- JSON dict <- one or several ``k=v,...`` strings, in that order (base_model.py:99-109)."""
        hp = self.get_default_hparams().override_from_dict(hparams_dict or {})
        strings = [] if not hparams else (list(hparams) if isinstance(hparams, (list, tuple)) else [hparams])
        for text in strings:
            hp.parse(text)
        return hp

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
        BaseVideoPredictionModel.__init__(self, mode, hparams_dict, hparams, **kwargs)

        def bound(fn):                                          # the plug-ins see the model's mode and hparams (base_model.py:212-216)
            return functools.partial(fn, mode=self.mode, hparams=self.hparams) if fn else None
        self.generator_fn, self.discriminator_fn = bound(generator_fn), bound(discriminator_fn)
        self.generator_scope, self.discriminator_scope = generator_scope, discriminator_scope
        self.aggregate_nccl = aggregate_nccl
        for slot in _TRAINABLE_SLOTS:
            setattr(self, slot, None)

    @property
    def learning_rate(self):
        return learning_rate(self.hparams, self.global_step)

    @property
    def kl_weight(self):
        return kl_weight(self.hparams, self.global_step)

    global_step = 0

    def get_default_hparams_dict(self):
        return hparam_defaults.trainable_defaults()
