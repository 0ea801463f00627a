"""Mini-batch data parallelism: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code at all; this module is the one
parallelism the north star asks for.  Each global mini-batch is drawn identically on
every rank (same seed -> same Keras index stream) and rank r keeps the utterances
``r::world`` of the SORTED index array (HDF5 reads stay monotonic,
datasets/dataset_generator.py:200).  Every rank pads to the GLOBAL T_max so the
backward LSTM sees the same zero tail as a single-GPU run.  Gradients are summed
with ONE fp32 all-reduce over the model's flat gradient buffer (27.7 MB at cfg2,
110.6 MB at cfg3) and were already scaled by 1/N_global in the CTC kernel, so the
sum IS the gradient of the global batch mean; clip + Adam then run replicated.
The collective itself is issued through the library's own C ABI (``asr_comm_*``: RCCL
resolved with dlopen); ``torch.distributed`` only ferries the communicator's 128-byte id
(and serves the CPU tests over gloo).  Fallback: ``ASR_COMM=torch``, or a failed probe of the
library's RCCL entry points on ANY rank (agreed through the process group before the
communicator is created), routes the same calls through ``torch.distributed``.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE (torchrun).
    Returns (rank, world).  No-op for single-process runs."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world <= 1:
        return 0, 1
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {}
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device('cuda', local)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_indices(index_array, rank, world):
    """Rank r's share of a batch: r::world of the sorted indices."""
    return np.sort(np.asarray(index_array))[rank::world]


def allreduce_sum_(flat):
    """In-place sum over ranks of a flat tensor (the model's gradient buffer) through the
    process' gradient communicator (grad_comm): the library's RCCL entry points for a tensor in
    HBM, the host process group for a CPU tensor (the gloo tests)."""
    if world_size() > 1:
        comm = grad_comm(flat.device)
        comm.allreduce_after(flat, None)
        comm.join(None)
    return flat


def broadcast_parameters(model, src=0):
    """Every rank starts from rank `src`'s weights.  A STAGING copy is summed through the
    gradient communicator (rank `src` contributes its weights, the others zeros: exact, x + 0 +
    ...) and copied over the parameters only once the collective has been enqueued without an
    error -- a failing all-reduce leaves every rank's own weights untouched."""
    if world_size() > 1:
        stage = model.params.clone() if dist.get_rank() == src else torch.zeros_like(model.params)
        allreduce_sum_(stage)
        model.params.copy_(stage)


_COUNT_LIMB = 4096.0


def encode_metrics(vals, count):
    """float64 metric sums + integer sample count -> the float32 vector that rides the gradient
    communicator: [hi parts, lo parts, count % 4096, count // 4096]."""
    vals = torch.as_tensor(vals, dtype=torch.float64)
    hi = vals.float()
    c = float(int(count))
    limbs = torch.tensor([c % _COUNT_LIMB, float(int(c) // int(_COUNT_LIMB))], dtype=torch.float32)
    return torch.cat([hi, (vals - hi.double()).float(), limbs])


def decode_metrics(pair, k):
    """The summed vector of encode_metrics -> (float64 sums (k,), exact sample count)."""
    pair = pair.double()
    return pair[:k] + pair[k:2 * k], float(pair[2 * k] + _COUNT_LIMB * pair[2 * k + 1])


def reduce_metrics(sums, count):
    """Sum (metric_sums, sample_count) over ranks -> global batch-weighted means.  On the GPU
    the values travel through the gradient communicator (float32 only): each metric sum as a
    float32 (hi, lo) pair -- RCCL adds the hi parts across ranks in float32, so the global sum
    carries float32 accuracy (2^-24 relative per addition), not 48 bits; the SAMPLE COUNT as two
    integer limbs (count % 4096, count // 4096) whose cross-rank sums stay below 2^24 and are
    therefore exact for counts up to 2^36 / world.  torch.distributed carries nothing but the
    communicator's 128-byte id."""
    vals = torch.tensor(list(sums), dtype=torch.float64)
    if world_size() > 1:
        if torch.cuda.is_available() and dist.get_backend() == 'nccl':
            dev = torch.device('cuda', torch.cuda.current_device())
            pair = encode_metrics(vals, count).to(dev)
            comm = grad_comm(dev)
            comm.allreduce_after(pair, torch.cuda.current_stream(dev))
            comm.join(torch.cuda.current_stream(dev))
            vals, count = decode_metrics(pair.cpu(), vals.numel())
        else:
            t = torch.cat([vals, torch.tensor([float(count)], dtype=torch.float64)])
            dist.all_reduce(t)
            vals, count = t[:-1], float(t[-1])
    return (vals / max(float(count), 1.0)).tolist()


class ShardedBatch(list):
    """``[x, labels, lens]`` of one rank's shard plus what the step needs to know about the
    GLOBAL batch it came from: ``n_global`` (samples in the global batch: the gradient is
    scaled by 1/n_global on every rank so that the all-reduced sum is the gradient of the
    global batch mean, and an epoch counts global samples) and ``n_local`` (this rank's real
    samples; 0 when the global batch was smaller than the world and this rank only carries a
    zero-weight dummy so that it still takes part in the collectives)."""

    def __init__(self, items, n_global, n_local):
        super(ShardedBatch, self).__init__(items)
        self.n_global, self.n_local = int(n_global), int(n_local)


class ShardedFlow(object):
    """Wraps a DatasetIterator: every rank advances the same index stream and keeps
    its ``rank::world`` shard; inputs are padded to the global batch's T_max."""

    def __init__(self, flow, rank, world):
        self.flow, self.rank, self.world = flow, rank, world

    @property
    def len(self):
        return self.flow.len

    def __iter__(self):
        return self

    def __next__(self):
        (x, labels, lens), (zeros, _) = next(self.flow)
        n = len(np.asarray(lens).reshape(-1))
        keep = np.arange(n)[self.rank::self.world]
        n_local = len(keep)
        if n_local == 0:                    # zero-weight dummy: sample 0 with gradient scale 0
            keep = np.arange(n)[:1]
        csr = labels.tocsr()
        lab = [csr.data[csr.indptr[i]:csr.indptr[i + 1]] for i in keep]
        if isinstance(x, tuple) and x[0] == 'slab':
            raise NotImplementedError('on-device features + sharding: shard before extraction')
        x = np.asarray(x)[keep]                  # still padded to the GLOBAL T_max
        return (ShardedBatch([x, lab, np.asarray(lens).reshape(-1)[keep]], n, n_local),
                [zeros[keep], lab])

    next = __next__


_gpu_comm_kind = [None]         # 'capi' | 'torch', decided once per process, the same on every rank


def _decide_gpu_comm(device):
    """ASR_COMM=torch forces the torch.distributed path; otherwise every rank probes the
    library's RCCL entry points (dlopen of librccl + ncclGetUniqueId) and the ranks AGREE on the
    outcome through the process group (MIN of the flags) before any of them enters the collective
    ncclCommInitRank -- a rank whose dlopen failed would otherwise leave the others hanging in
    it.  Any failure -> every rank uses TorchDistComm (a warning says so)."""
    if os.environ.get('ASR_COMM', 'capi') == 'torch':
        return 'torch'
    ok = 1
    try:
        import ctypes as C
        from . import _lib
        buf = (C.c_char * 128)()
        ok = 1 if _lib.load().asr_comm_unique_id(buf) == 0 else 0
    except Exception:
        ok = 0
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([ok], dtype=torch.int32,
                         device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        import warnings
        warnings.warn('asr_comm_* (RCCL behind the C ABI) is unavailable on at least one rank: '
                      'the gradient all-reduce falls back to torch.distributed on every rank')
    return 'capi' if ok else 'torch'


def grad_comm(device):
    """The communicator the gradient all-reduce goes through: CapiComm (asr_comm_*: RCCL behind
    the C ABI, the product's path) for tensors in HBM -- TorchDistComm (the process group's own
    RCCL communicator) if ASR_COMM=torch or the library's entry points fail their probe on any
    rank -- and HostGroupComm (the host process group, gloo) for CPU tensors, which only the CPU
    tests have."""
    device = torch.device(device)
    if device.type != 'cuda':
        return HostGroupComm.get()
    if _gpu_comm_kind[0] is None:
        _gpu_comm_kind[0] = _decide_gpu_comm(device)
    return CapiComm.get() if _gpu_comm_kind[0] == 'capi' else TorchDistComm.get()


class TorchDistComm(object):
    """Fallback (ASR_COMM=torch, or asr_comm_* unavailable): the same interface on
    torch.distributed's RCCL communicator, collectives on this object's own stream."""

    _instance = None

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def __init__(self):
        self.world = world_size()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.calls = 0

    def allreduce_after(self, flat, after=None):
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        after = after or torch.cuda.current_stream(flat.device)
        self.stream.wait_stream(after)
        flat.record_stream(self.stream)
        if dist.is_available() and dist.is_initialized():       # (also at world 1: an identity)
            with torch.cuda.stream(self.stream):
                dist.all_reduce(flat)
            self.calls += 1                 # collectives actually issued
        return flat

    def join(self, stream=None):
        (stream or torch.cuda.current_stream(self.device)).wait_stream(self.stream)

    def allreduce_sum_(self, flat):
        self.allreduce_after(flat, None)
        self.join(None)
        return flat

    def close(self):
        TorchDistComm._instance = None


class HostGroupComm(object):
    """CPU tensors (the world-2 gloo tests of the bookkeeping): the process group itself."""

    _instance = None

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def allreduce_after(self, flat, after):
        dist.all_reduce(flat)
        return flat

    def join(self, stream):
        pass


class CapiComm(object):
    """The gradient all-reduce through the library's own RCCL entry points
    (include/asr_hip.h C1): rank 0 draws the RCCL unique id, the process group only ferries
    those 128 bytes, and every collective is enqueued by ``asr_comm_allreduce_sum`` on this
    communicator's OWN stream: ``allreduce_after(t, s)`` orders it behind the work already
    enqueued on stream ``s`` (None = the current one), ``join(s)`` makes ``s`` wait for every
    collective issued so far.  One stream per communicator: the collectives execute in the
    order they were issued, identically on every rank."""

    _instance = None

    @classmethod
    def get(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def __init__(self):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib.load()
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        box = [None]
        if rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(self._lib.asr_comm_unique_id(buf), 'asr_comm_unique_id')
            box[0] = bytes(buf)
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        self._comm = C.c_void_p()
        _lib.check(self._lib.asr_comm_init(box[0], rank, world, C.byref(self._comm)),
                   'asr_comm_init')
        self.world = world
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.calls = 0

    def allreduce_after(self, flat, after=None):
        from . import _lib
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        after = after or torch.cuda.current_stream(flat.device)
        self.stream.wait_stream(after)
        flat.record_stream(self.stream)
        _lib.check(self._lib.asr_comm_allreduce_sum(
            self._comm, self._C.c_void_p(flat.data_ptr()), flat.numel(),
            self._C.c_void_p(self.stream.cuda_stream)), 'asr_comm_allreduce_sum')
        self.calls += 1
        return flat

    def join(self, stream=None):
        (stream or torch.cuda.current_stream(self.device)).wait_stream(self.stream)

    def allreduce_sum_(self, flat):
        """Blocking form (in stream order): reduce, then the current stream waits."""
        self.allreduce_after(flat, None)
        self.join(None)
        return flat

    def close(self):
        if self._comm:
            torch.cuda.synchronize(self.device)
            self._lib.asr_comm_destroy(self._comm)
            self._comm = self._C.c_void_p()
        CapiComm._instance = None
