"""The N > 1 gate of the data-parallel path on real GPUs: armed on every box,
exercised at world size 2 wherever a second GPU is visible, at world size 1 otherwise (same
worker, same code path, the collective an identity).  The day a multi-GPU node exists the first
SCALE run is then a measurement, not a debugging session."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_worker(world, port, **env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', ASR_FORCE_ALLREDUCE='1', **env_extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(world), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', 'dp_worker.py')]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=560, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('RESULT ')][0]
    return json.loads(line[7:])


def _check(res, world):
    assert res['world'] == world and res['ranks_seen_by_rccl'] == world
    assert sum(res['shard_sizes']) == 16 * world + 5
    # sharded gradients summed by the step's all-reduce == the whole-batch gradient
    assert res['grad_max_err_rel'] < 1e-6, res
    # chip-filling recurrences, compact backward schedule: the two upper layers' buckets go out
    # beside the compact BPTTs (every rank: the sum over ranks is 2 W), no timeout / fallback
    assert res['collectives_during_bptt_chipfill'] == 2 * world
    assert res['timeouts_or_fallbacks_any_rank'] == 0 and res['chipfill_loss_finite']
    assert res['collectives_through_capi'] >= 4


@pytest.mark.timeout(600)
def test_data_parallel_worker_at_world_one():
    _check(_run_worker(1, 29581), 1)


@pytest.mark.timeout(600)
def test_data_parallel_worker_on_the_torch_distributed_fallback():
    """ASR_COMM=torch (parallel.TorchDistComm: what every rank falls back to if the library's
    RCCL entry points fail their probe on any rank): the same worker, the collectives through
    torch.distributed's own RCCL communicator on the fallback's stream."""
    res = _run_worker(1, 29585, ASR_COMM='torch')
    assert res['comm_kind'] == 'TorchDistComm'
    _check(res, 1)


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs a second GPU')
def test_two_rank_gradients_equal_the_single_rank_gradient_of_the_whole_batch():
    _check(_run_worker(2, 29583), 2)


@pytest.mark.timeout(600)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs a second GPU')
def test_bench_at_two_gpus_reports_both_ranks_and_a_measured_allreduce():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
           '--warmup', '1', '--no-extras', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=560, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['ranks_seen_by_rccl'] == 2 and line['fallbacks'] == 0
    assert line['config']['global_batch'] == 128 and line['scaling'] == 'weak'
    assert line['allreduce']['bus_GBps'] > 1.0 and line['allreduce']['bytes'] > 1e8
