"""Data-parallel replicas: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference replicates the model in-graph per GPU and sums every gradient tensor separately with
tf.contrib.nccl.all_sum, then scales by 1/K (/root/reference/video_prediction/utils/tf_utils.py:450-480, called at
models/base_model.py:590-592 and :614-616); replicas are initialised by copying tower 0's variables (:640-646).

Here each optimiser group (discriminator; generator+encoder) is ONE flat fp32 arena (ParamGroup.g) whose variables are laid
out network by network, so the gradients of one network are one contiguous chunk.  A chunk is all-reduced as soon as the
backward pass that produces it has been ISSUED -- on a side HIP stream, chained to the compute stream with events:

    compute stream : ... backward of network A | record(eA) | forward/backward of network B ...... | wait(dA) wait(dB) | Adam
    comm stream    :                            wait(eA) | RCCL all-reduce(chunk A) | record(dA) ...

so the reduction of chunk A overlaps the compute of network B (D step: the posterior-side discriminator's 20.6 MB under the
prior-side discriminator's forward/backward; G step: the generator cell's 26 MB under the encoder's backward; with
joint_gan_optimization the whole D bucket under the generator step).  The 1/K scale is folded into the Adam kernel.  The time
axis is a serial recurrence and is never sharded; the batch is.

Why torch.distributed is the boundary (SURVEY.md 8(b) lists `savp_allreduce_bucket(ncclComm_t, hipStream_t, void*, size_t)`):
the communicator bootstrap (rendezvous, unique-id exchange, process-group lifetime) is exactly what torch.distributed provides
around RCCL, bench.py is launched by torch.distributed.run, and ProcessGroupNCCL issues ncclAllReduce on the flat bucket
pointer with no copy -- a C-ABI shim would call the same RCCL entry point with the same arguments.  See INTEGRATION.md.
"""
import torch


class ReplicaGroup(object):
    def __init__(self, store, dist_module=None, overlap=True, force=False, own_comm=False):
        """force: run every collective even in a group of ONE rank (SAVP_FORCE_DIST=1).  A sum over one replica is the identity,
        so the step's numbers do not change, but process-group creation, the rank-0 broadcast, the side-stream chunked
        all-reduce, the u broadcast and the event chaining all execute on the real transport -- the way to exercise RCCL on a
        one-GPU box before a multi-GPU lease does (tests/test_gpu_dp.py)."""
        self.store = store
        self.dist = dist_module
        self.world = dist_module.get_world_size() if dist_module is not None else 1
        self.rank = dist_module.get_rank() if dist_module is not None else 0
        self.active = dist_module is not None and (self.world > 1 or bool(force))
        dev = torch.device(store.device)
        self.on_gpu = dev.type == 'cuda'
        # side stream of the gradient exchange (None on the CPU: gloo reduces host tensors synchronously)
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.on_gpu and overlap and self.active) else None
        self.pending = {}          # group -> list of (lo, hi, done event | None)
        self.stats = {'chunks': 0, 'elements': 0, 'aux_broadcasts': 0}
        self.aux_done = None
        self.comm = None           # own_comm: an ncclComm_t this object owns (see _own_communicator)
        self._rccl = None
        if own_comm and self.active and self.on_gpu:
            self._own_communicator()
        if self.active:
            for g in store.groups.values():            # post_init_ops: every replica starts from rank 0's variables
                dist_module.broadcast(g.p, src=0)

    # -- a communicator of our own (SAVP_GRAPH_COLLECTIVES=1) ----------------------------------------------------------------------
    def _own_communicator(self):
        """RCCL communicator owned by this object: ncclGetUniqueId on rank 0, the id handed round through the existing process group,
        ncclCommInitRank everywhere.  The step's collectives then go through the C ABI (`savp_allreduce_bucket`: ncclAllReduce on the given
        stream; the u vectors' broadcast: ncclBroadcast) instead of ProcessGroupNCCL -- the same RCCL kernels, but no Work objects: when the
        step is CAPTURED with its collectives inside, ProcessGroupNCCL's watchdog thread polls the end event of every collective it issued,
        and polling an event that was recorded while capturing is an error that aborts the process
        (profiles/r06_graph_collectives_watchdog_abort.log).  librccl.so is the copy torch already holds."""
        import ctypes

        class UID(ctypes.Structure):
            _fields_ = [('internal', ctypes.c_char * 128)]
        rccl = ctypes.CDLL('librccl.so')
        rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UID)]
        rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UID, ctypes.c_int]
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclBroadcast.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        uid = UID()
        if self.rank == 0 and rccl.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            raise RuntimeError('ncclGetUniqueId failed')
        dev = torch.device(self.store.device)
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
        self.dist.broadcast(t, src=0)
        ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        comm = ctypes.c_void_p()
        if rccl.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank) != 0 or not comm.value:
            raise RuntimeError('ncclCommInitRank failed')
        self.comm, self._rccl = comm, rccl

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            self._rccl.ncclCommDestroy(self.comm)
            self.comm = None

    def _all_reduce(self, chunk):
        """Sum `chunk` (a contiguous fp32 slice of a gradient arena) over the replicas on the CURRENT stream."""
        if self.comm is None:
            self.dist.all_reduce(chunk)
            return
        import ctypes
        from . import lib
        lib.check(lib.get().savp_allreduce_bucket(self.comm, ctypes.c_void_p(torch.cuda.current_stream(chunk.device).cuda_stream),
                                                  ctypes.c_void_p(chunk.data_ptr()), chunk.numel()), 'savp_allreduce_bucket')

    def _broadcast(self, buf):
        if self.comm is None:
            self.dist.broadcast(buf, src=0)
            return
        import ctypes
        NCCL_FLOAT32 = 7
        assert buf.dtype == torch.float32 and buf.is_contiguous()
        rc = self._rccl.ncclBroadcast(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(buf.data_ptr()), buf.numel(), NCCL_FLOAT32, 0, self.comm,
                                      ctypes.c_void_p(torch.cuda.current_stream(buf.device).cuda_stream))
        if rc != 0:
            raise RuntimeError('ncclBroadcast failed with code %d' % rc)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    # -- chunked, stream-overlapped exchange ---------------------------------------------------------------------------------
    def begin_allreduce(self, group, lo=0, hi=None):
        """Start summing elements lo:hi of the group's flat gradient arena over all replicas.  Everything launched on the
        current stream so far (the backward pass that produced the chunk) is waited for by the side stream; later launches on
        the current stream run concurrently with the exchange until finish_allreduce(group)."""
        if not self.active:
            return
        g = self.store.groups[group].g
        hi = g.numel() if hi is None else hi
        if hi <= lo:
            return
        chunk = g[lo:hi]
        done = None
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream(g.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            self.comm_stream.wait_event(ready)
            with torch.cuda.stream(self.comm_stream):
                self._all_reduce(chunk)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
        else:
            self._all_reduce(chunk)
        self.pending.setdefault(group, []).append((lo, hi, done))
        self.stats['chunks'] += 1
        self.stats['elements'] += hi - lo

    def finish_allreduce(self, group):
        """Exchange whatever part of the group's arena no begin_allreduce covered, then make the current stream wait for every
        chunk: after this call the arena holds the sum over replicas (divide by world in Adam)."""
        if not self.active:
            return
        n = self.store.groups[group].g.numel()
        covered = sorted((lo, hi) for lo, hi, _ in self.pending.get(group, []))
        pos = 0
        gaps = []
        for lo, hi in covered:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < n:
            gaps.append((pos, n))
        for lo, hi in gaps:
            self.begin_allreduce(group, lo, hi)
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream(self.store.groups[group].g.device)
            for _, _, done in self.pending.get(group, []):
                cur.wait_event(done)
        self.pending[group] = []

    def sync_aux(self):
        """Non-trainable state that every replica recomputes from identical weights (the spectral-norm power-iteration vectors
        u): the GEMVs behind it sum in a hardware-dependent order, so the replicas' copies drift apart in the last bit.  One
        small broadcast of the 'aux' arena per step (a few KB, on the side stream) keeps the replicas bit-identical.  The compute
        stream does NOT wait here: the next reader of u is the discriminators' weight preparation of the NEXT step, a whole
        generator forward later -- wait_aux() in front of it finds the broadcast long finished, so the collective is off the
        critical path of every step."""
        if not self.active:
            return
        aux = self.store.groups.get('aux')
        if aux is None or aux.p.numel() == 0:
            return
        self.stats['aux_broadcasts'] += 1
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream(aux.p.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            self.comm_stream.wait_event(ready)
            with torch.cuda.stream(self.comm_stream):
                self._broadcast(aux.p)
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            self.aux_done = done
        else:
            self._broadcast(aux.p)

    def wait_aux(self):
        """Order the current stream behind the last sync_aux() broadcast (call before anything reads or writes the 'aux' arena)."""
        done = self.aux_done
        if done is not None:
            aux = self.store.groups['aux']
            torch.cuda.current_stream(aux.p.device).wait_event(done)
            self.aux_done = None

    def allreduce_grads(self, group, async_op=False):
        """Sum the whole flat gradient bucket of one optimiser group (blocking with respect to the current stream)."""
        if self.active:
            self.finish_allreduce(group)
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
        if not self.active:
            return True
        self.wait_aux()
        ok = True
        for g in self.store.groups.values():
            s = g.p.double().sum().reshape(1).clone()
            a = g.p.double().abs().sum().reshape(1).clone()
            mine = torch.cat([s, a])
            gathered = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(gathered, mine)
            ok = ok and all(bool((t == gathered[0]).all()) for t in gathered)
        return ok
