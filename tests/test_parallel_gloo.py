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
