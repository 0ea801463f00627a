"""Worker of tests/test_gpu_multi.py, one process per GPU under torch.distributed.run: the
data-parallel step at world size W on real GPUs.  Prints ONE line ``RESULT {json}`` on rank 0.

  1. cfg2 width (2 x BiLSTM(256), 39 features), T = 200, a global batch of 16 W + 5 utterances
     (uneven shards): every rank computes the gradient of ITS shard (r::W of the batch, scaled
     by 1 / N_global), the flat buffer is summed by the step's own all-reduce (asr_comm_*: RCCL
     through the C ABI) -- and must equal the gradient of the WHOLE batch computed by the same
     process without any collective (1e-6 of the maximum).
  2. cfg3 width (3 x BiLSTM(512), 64 utterances per rank: the recurrence fills the chip, the
     backward pass runs the compact schedule): the buckets of the two layers above the bottom
     one are all-reduced beside the compact BPTTs below them (r6), no recurrent-kernel timeout,
     no fallback, on any rank.
Runs at W = 1 too (ASR_FORCE_ALLREDUCE=1: the collective is an identity, the code path is the
same), which is what a single-GPU box exercises."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from asr_study_amd import ops, parallel
    from asr_study_amd.core import models, optimizers
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    out = {'world': world}

    # ---- 1. sharded gradient + all-reduce == whole-batch gradient
    rs = np.random.RandomState(0)                       # the same global batch on every rank
    F, H, C, T = 39, 256, 28, 200
    n_global = 16 * world + 5
    x = rs.randn(n_global, T, F).astype(np.float32)
    lens = rs.randint(T // 2, T + 1, size=n_global)
    for n in range(n_global):
        x[n, lens[n]:] = 0
    labels = [rs.randint(0, C - 1, size=rs.randint(2, 20)).tolist() for _ in range(n_global)]
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=2,
                          dropout=0.0, weight_decay=1e-4, seed=3, device=dev)
    model.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    parallel.broadcast_parameters(model)
    keep = parallel.shard_indices(np.arange(n_global), rank, world)
    slab = model.to_slab(x[keep])                       # padded to the GLOBAL T_max = T
    model.loss_and_grads(slab, [labels[i] for i in keep], lens[keep], training=False,
                         n_global=n_global, n_ref=n_global)
    model._allreduce()
    torch.cuda.synchronize()
    got = model.grads.clone()
    active = model._dist_active
    model._dist_active = lambda: False                  # the same process, no collective
    model.loss_and_grads(model.to_slab(x), labels, lens, training=False, n_global=n_global)
    model._dist_active = active
    want = model.grads.clone()
    scale = float(want.abs().max().item())
    out['grad_max_err_rel'] = float((got - want).abs().max().item()) / scale
    out['shard_sizes'] = [int(len(parallel.shard_indices(np.arange(n_global), r, world)))
                          for r in range(world)]

    # ---- 2. chip-filling width: one collective behind BPTT, no timeout on any rank
    big = models.brsmv1(num_features=16, num_classes=7, num_hiddens=512, num_layers=3,
                        dropout=0.0, seed=1, device=dev)
    big.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    rb = np.random.RandomState(10 + rank)
    xb = rb.randn(64, 40, 16).astype(np.float32)
    lb = [rb.randint(0, 6, size=3).tolist() for _ in range(64)]
    covered = []
    orig = big._allreduce
    def counting():
        covered.append(len(big._ar_covered))
        return orig()
    big._allreduce = counting
    for _ in range(3):
        m = big.train_on_batch(parallel.ShardedBatch([xb, lb, [40] * 64], 64 * world, 64))
    flags = float(ops.lstm_timeout_flags(dev).ne(0).sum().item()) + float(big.fallbacks)
    t = torch.tensor([flags, 1.0, float(covered[-1])], dtype=torch.float32, device=dev)
    parallel.grad_comm(dev).allreduce_sum_(t)
    out['timeouts_or_fallbacks_any_rank'] = float(t[0].item())
    out['ranks_seen_by_rccl'] = int(t[1].item())
    out['collectives_during_bptt_chipfill'] = float(t[2].item())
    out['chipfill_loss_finite'] = bool(np.isfinite(m[0]))
    comm = parallel.grad_comm(dev)
    out['collectives_through_capi'] = comm.calls
    out['comm_kind'] = type(comm).__name__
    if rank == 0:
        print('RESULT ' + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
