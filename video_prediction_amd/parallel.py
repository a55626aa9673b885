"""Data-parallel replicas: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference replicates the model in-graph per GPU and sums every gradient tensor separately with
tf.contrib.nccl.all_sum, then scales by 1/K (/root/reference/video_prediction/utils/tf_utils.py:450-480, called at
models/base_model.py:590-592 and :614-616); replicas are initialised by copying tower 0's variables (:640-646).
Here each optimiser group (discriminator; generator+encoder) is ONE flat fp32 bucket (ParamGroup.g), so a step issues
exactly two all-reduces (41.2 MB and 29.4 MB for BAIR SAVP) and the 1/K scale is folded into the Adam kernel.
The time axis is a serial recurrence and is never sharded; the batch is.
"""


class ReplicaGroup(object):
    def __init__(self, store, dist_module=None):
        self.store = store
        self.dist = dist_module
        self.world = dist_module.get_world_size() if dist_module is not None else 1
        self.rank = dist_module.get_rank() if dist_module is not None else 0
        if self.world > 1:
            for g in store.groups.values():            # post_init_ops: every replica starts from rank 0's variables
                dist_module.broadcast(g.p, src=0)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def allreduce_grads(self, group, async_op=False):
        """Sum the flat gradient bucket of one optimiser group over all replicas (average = grad_scale in Adam)."""
        if self.world > 1:
            return self.dist.all_reduce(self.store.groups[group].g, async_op=async_op)
        return None

    def shard(self, global_batch_tensor, dim=0):
        """tf.split(input, num_gpus) along the batch axis (base_model.py:523-527)."""
        n = global_batch_tensor.shape[dim]
        if n % self.world:
            raise ValueError('batch %d not divisible by %d replicas' % (n, self.world))
        per = n // self.world
        return global_batch_tensor.narrow(dim, self.rank * per, per)

    def checksum_identical(self):
        """True iff every replica holds bit-identical variables (they must: identical averaged grads, identical Adam)."""
        if self.world == 1:
            return True
        import torch
        ok = True
        for g in self.store.groups.values():
            s = g.p.double().sum().reshape(1).clone()
            a = g.p.double().abs().sum().reshape(1).clone()
            mine = torch.cat([s, a])
            gathered = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(gathered, mine)
            ok = ok and all(bool((t == gathered[0]).all()) for t in gathered)
        return ok
