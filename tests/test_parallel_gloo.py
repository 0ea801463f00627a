"""world_size-2 gloo test (CPU) of the data-parallel path: rank shards + one
all-reduce of the flat gradient == the single-process gradient of the whole batch.
The per-rank gradient comes from the float64 oracle (stand-in for the HIP engine,
which needs a GPU); what is under test is the sharding, the 1/N_global scaling, the
global-T_max padding rule and the collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from oracle import lstm as OL
    rs = np.random.RandomState(0)
    N, T, F, H, C = 6, 17, 5, 4, 6
    params = OL.init_model(seed=2, num_features=F, num_hiddens=H, num_layers=2, num_classes=C,
                           dtype=np.float64)
    for _, a in OL.flatten(params):
        a += rs.randn(*a.shape) * 0.2
    x = rs.randn(T, N, F)
    lens = np.array([T, 9, T, 12, T, 5])
    for n in range(N):
        x[lens[n]:, n] = 0
    labels = [rs.randint(0, C - 1, size=rs.randint(1, 4)).tolist() for _ in range(N)]
    return params, x, labels, lens


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from asr_study_amd import parallel
    from oracle import lstm as OL
    from oracle import ctc as OC
    r, w = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    params, x, labels, lens = _problem()
    N = x.shape[1]
    keep = parallel.shard_indices(np.arange(N), rank, world)
    xs = x[:, keep]                                   # padded to the GLOBAL T_max
    logits, caches = OL.model_forward(params, xs)
    ctc, dlog = OC.ctc_loss_grad(logits, [labels[i] for i in keep], lens[keep])
    grads = OL.model_backward(params, caches, dlog / N)          # 1 / N_global
    flat = torch.from_numpy(np.concatenate([g.ravel() for _, g in OL.flatten(grads)]))
    parallel.allreduce_sum_(flat)
    means = parallel.reduce_metrics([float(np.sum(ctc))], len(keep))
    if rank == 0:
        np.save(out, np.concatenate([flat.numpy(), [means[0]]]))
    parallel.finalize()


def test_two_rank_gradient_equals_single_process(tmp_path):
    from oracle import lstm as OL
    out = str(tmp_path / 'g.npy')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    params, x, labels, lens = _problem()
    want = OL.loss_and_grads(params, x, labels, lens)
    ref = np.concatenate([g.ravel() for _, g in OL.flatten(want['grads'])])
    np.testing.assert_allclose(got[:-1], ref, atol=1e-12)
    assert abs(got[-1] - float(np.mean(want['ctc']))) < 1e-12


def test_shard_indices_monotonic_and_disjoint():
    from asr_study_amd import parallel
    idx = np.array([9, 2, 7, 4, 1, 8, 3])
    parts = [parallel.shard_indices(idx, r, 3) for r in range(3)]
    assert sorted(np.concatenate(parts).tolist()) == sorted(idx.tolist())
    for p in parts:
        assert np.all(np.diff(p) > 0)


def test_sharded_flow_counts_global_samples_and_weights_uneven_shards():
    """ShardedFlow (host logic, no process group): every rank sees the same number of steps
    per epoch = ceil(len / batch); the shards of a batch add up to the global batch;
    n_global is the GLOBAL batch size on every rank (the 1/N_global gradient scale), also
    for the short last batch; a rank left without a sample carries a zero-weight dummy."""
    import scipy.sparse as sp
    from asr_study_amd import parallel

    class Flow(object):
        def __init__(self, total, batch):
            self.len, self.batch, self.pos = total, batch, 0

        def __next__(self):
            n = min(self.batch, self.len - self.pos)
            self.pos = (self.pos + n) % self.len
            x = np.zeros((n, 4, 3), np.float32) + np.arange(n)[:, None, None]
            rows = np.repeat(np.arange(n), 2)
            cols = np.tile(np.arange(2), n)
            lab = sp.coo_matrix((np.arange(2 * n, dtype=np.int32) % 5, (rows, cols)))
            return [x, lab, np.full(n, 4)], [np.zeros(n), lab]
        next = __next__

    world, total, batch = 3, 9, 4            # batches of 4, 4, 1 -> the last one < world
    per_rank = []
    for rank in range(world):
        flow = parallel.ShardedFlow(Flow(total, batch), rank, world)
        seen, steps, shards = 0, 0, []
        while seen < flow.len:               # engine.fit_generator's epoch loop
            inputs, _ = next(flow)
            assert isinstance(inputs, parallel.ShardedBatch)
            seen += inputs.n_global
            steps += 1
            shards.append((inputs.n_global, inputs.n_local, np.asarray(inputs[0])[:, 0, 0].tolist()))
        per_rank.append((steps, shards))
    assert [s for s, _ in per_rank] == [3, 3, 3]
    for step in range(3):
        ng = {per_rank[r][1][step][0] for r in range(world)}
        assert ng == {(4, 4, 1)[step]}
        assert sum(per_rank[r][1][step][1] for r in range(world)) == ng.pop()
    # the 1-sample batch: rank 0 owns it, ranks 1 and 2 hold a zero-weight dummy (sample 0)
    assert [per_rank[r][1][2][1] for r in range(world)] == [1, 0, 0]
    assert per_rank[1][1][2][2] == [0.0]
    # an uneven 4-sample batch over 3 ranks: 2 + 1 + 1 samples, all scaled by 1/4
    assert [per_rank[r][1][0][1] for r in range(world)] == [2, 1, 1]


def _ar_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    from asr_study_amd import parallel
    from asr_study_amd.core import engine
    parallel.init_from_env(backend='gloo')
    n = 1000
    gbuf = torch.zeros(n + 4)
    gbuf[:n] = torch.arange(n, dtype=torch.float32) * (rank + 1)

    def collect():          # rank 1's BPTT kernel "timed out": flag slot 1 of ITS buffer
        gbuf[n:] = torch.tensor([0.0, 1.0 if rank == 1 else 0.0, 0.0, 0.0])
    fake = types.SimpleNamespace(n_params=n, _gbuf=gbuf, grads=gbuf[:n], _collect_flags=collect,
                                 _dist_active=lambda: True)
    # what backward() does for the layers whose gradients finish early: all-reduces of their
    # slices through the gradient communicator, in the order the layers finish (top layer first)
    comm = parallel.grad_comm(gbuf.device)
    assert isinstance(comm, parallel.HostGroupComm)       # CPU tensors: the host process group
    fake._ar_covered = []
    for lo, hi in ((700, 900), (300, 700), (120, 300)):
        comm.allreduce_after(fake.grads[lo:hi], None)
        fake._ar_covered.append((lo, hi))
    w = engine.Model._allreduce(fake)      # reduces [0,120) and [900,1000) + flags, joins
    # the veto flags arrive on EVERY rank (the guard of rank 0 must see rank 1's timeout)
    assert engine.Model.veto_flags(fake).tolist() == [0.0, 1.0]
    if rank == 0:
        np.save(out, np.concatenate([fake.grads.numpy(), [w]]))
    parallel.finalize()


def test_engine_allreduce_covers_every_gradient_exactly_once(tmp_path):
    """engine.Model._allreduce (host bookkeeping of the overlapped per-layer all-reduce): slices
    reduced asynchronously during BPTT are waited for, the remaining slices of the flat buffer
    (first layer, Dense head) are reduced in place, nothing twice; the two timeout-flag slots
    behind the gradients ride on the last collective, so a timeout on one rank vetoes the update
    on all of them -- world 2 over gloo."""
    out = str(tmp_path / 'ar.npy')
    port = _free_port()
    mp.spawn(_ar_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    np.testing.assert_array_equal(got[:-1], np.arange(1000, dtype=np.float32) * 3.0)
    assert got[-1] == 2


def _w8_problem(N):
    from oracle import lstm as OL
    rs = np.random.RandomState(40 + N)
    T, F, H, C = 9, 4, 3, 5
    params = OL.init_model(seed=3, num_features=F, num_hiddens=H, num_layers=1, num_classes=C,
                           dtype=np.float64)
    for _, a in OL.flatten(params):
        a += rs.randn(*a.shape) * 0.2
    x = rs.randn(T, N, F)
    lens = rs.randint(4, T + 1, size=N)
    lens[0] = T
    for n in range(N):
        x[lens[n]:, n] = 0
    labels = [rs.randint(0, C - 1, size=rs.randint(1, 3)).tolist() for _ in range(N)]
    return params, x, labels, lens


def _w8_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import types
    from asr_study_amd import parallel
    from asr_study_amd.core import engine
    from oracle import lstm as OL
    from oracle import ctc as OC
    torch.set_num_threads(1)
    parallel.init_from_env(backend='gloo')
    result = {}
    # (1) parameters: every rank starts elsewhere, rank 0's weights win -- through a staging copy
    mdl = types.SimpleNamespace(params=torch.arange(37, dtype=torch.float32) * (rank + 1) + rank)
    parallel.broadcast_parameters(mdl, src=0)
    assert torch.equal(mdl.params, torch.arange(37, dtype=torch.float32))
    for N, bad_rank in ((11, None), (5, 5)):        # 11: shards of 2,2,2,1,..; 5 < world: dummies
        params, x, labels, lens = _w8_problem(N)
        keep = np.arange(N)[rank::world]            # ShardedFlow's rule on a sorted batch
        n_local = len(keep)
        if n_local == 0:
            keep = np.arange(N)[:1]                 # zero-weight dummy (sample 0)
        xs = x[:, keep]
        logits, caches = OL.model_forward(params, xs)
        ctc, dlog = OC.ctc_loss_grad(logits, [labels[i] for i in keep], lens[keep])
        scale = (1.0 / N) if n_local else 0.0       # engine: grad_scale = 1/n_global, 0 for a dummy
        grads = OL.model_backward(params, caches, dlog * scale)
        flat = np.concatenate([g.ravel() for _, g in OL.flatten(grads)])
        n = flat.size
        gbuf = torch.zeros(n + 4, dtype=torch.float64)
        gbuf[:n] = torch.from_numpy(flat)

        def collect(gbuf=gbuf, n=n, bad=bad_rank):      # rank `bad`'s forward kernel timed out
            gbuf[n:] = torch.tensor([1.0 if rank == bad else 0.0, 0.0, 0.0, 0.0],
                                    dtype=torch.float64)
        fake = types.SimpleNamespace(n_params=n, _gbuf=gbuf, grads=gbuf[:n],
                                     _collect_flags=collect, _dist_active=lambda: True,
                                     _ar_covered=[])
        w = engine.Model._allreduce(fake)
        assert w == world
        flags = engine.Model.veto_flags(fake).tolist()
        assert flags == ([1.0, 0.0] if bad_rank is not None else [0.0, 0.0]), flags
        # (2) metrics: the float32 (hi, lo) + count-limb vector the GPU path sends, summed by the
        # process group in float32 exactly as RCCL would, then the plain float64 path
        loss_sum = float(np.sum(ctc)) if n_local else 0.0
        big = 3000000 + rank                        # a sample count beyond 2^24 / world in total
        vec = parallel.encode_metrics([loss_sum, 1e6 * (rank + 1) + 0.123], big)
        assert vec.dtype == torch.float32
        dist.all_reduce(vec)
        vals, cnt = parallel.decode_metrics(vec, 2)
        means = parallel.reduce_metrics([loss_sum], n_local)
        result[N] = dict(grad=gbuf[:n].numpy().copy(), mean=means[0], cnt=cnt,
                         loss_pair=float(vals[0]), big_pair=float(vals[1]))
    if rank == 0:
        np.save(out, result, allow_pickle=True)
    parallel.finalize()


@pytest.mark.timeout(300)
def test_world_8_bookkeeping_uneven_shards_dummy_ranks_veto_and_metric_pairs(tmp_path):
    """SURVEY 8e at the world size the driver will launch (8 ranks, gloo / HostGroupComm on the
    CPU): a global batch of 11 (shards of 2,2,2,1,1,1,1,1) and one of 5 (< world: three ranks
    carry a zero-weight dummy and still join every collective); sum of the 1/N_global-scaled
    shard gradients == the whole-batch gradient; a timeout flag raised on ONE rank reaches all;
    parameters broadcast through a staging copy; metric sums as float32 (hi, lo) pairs to
    float32 accuracy and the sample count EXACT through its limbs beyond 2^24."""
    from oracle import lstm as OL
    out = str(tmp_path / 'w8.npy')
    port = _free_port()
    mp.spawn(_w8_worker, args=(8, port, out), nprocs=8, join=True)
    got = np.load(out, allow_pickle=True).item()
    for N in (11, 5):
        params, x, labels, lens = _w8_problem(N)
        want = OL.loss_and_grads(params, x, labels, lens)
        ref = np.concatenate([g.ravel() for _, g in OL.flatten(want['grads'])])
        np.testing.assert_allclose(got[N]['grad'], ref, atol=1e-12)
        assert abs(got[N]['mean'] - float(np.mean(want['ctc']))) < 1e-12
        assert got[N]['cnt'] == sum(3000000 + r for r in range(8))          # exact, > 2^24
        assert abs(got[N]['loss_pair'] - float(np.sum(want['ctc']))) < 1e-5 * float(np.sum(want['ctc']))
        tot = sum(1e6 * (r + 1) + 0.123 for r in range(8))
        assert abs(got[N]['big_pair'] - tot) < 4e-7 * tot
