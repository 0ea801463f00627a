"""The RCCL leg of the data-parallel step on real hardware.  A GPU box has ONE GPU, so
the collective runs at world size 1 (forced): this covers process-group creation with
the 'nccl' backend, the all-reduce of the flat gradient buffer ordered against the HIP
kernels on torch's current stream, and bench.py's launch contract under torchrun.  The
world-size-2 arithmetic (sharding, 1/N_global, sum) is covered by test_parallel_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', ASR_BENCH_CONFIG='cfg2', **extra_env)
    cmd = launcher + [os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup',
                      '1', '--no-cpu-baseline']
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=280, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_contract_under_torchrun_with_rccl_allreduce():
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
                '1', '--master-addr', '127.0.0.1', '--master-port', '29533']
    line = _bench({'ASR_FORCE_ALLREDUCE': '1'}, launcher)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                'roofline'):
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['scaling'] == 'weak'
    assert line['value'] > 0 and line['roofline']['frac'] > 0
    assert line['config']['workload'].startswith('cfg2')
