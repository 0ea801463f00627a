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


_SCRIPT = r'''
import os, sys, json
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["ASR_ROOT"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from asr_study_amd.core import models, optimizers
out = {}
for mode in ("1", "0", "capi", "noside", "noside0"):
    os.environ["ASR_AR_OVERLAP"] = {"capi": "0", "noside": "1", "noside0": "0"}.get(mode, mode)
    os.environ["ASR_COMM"] = "capi" if mode == "capi" else "torch"
    os.environ["ASR_OVERLAP"] = "0" if mode.startswith("noside") else "auto"
    model = models.brsmv1(num_features=9, num_classes=7, num_hiddens=16, num_layers=3,
                          dropout=0.0, weight_decay=1e-4, seed=1)
    model.compile(optimizer=optimizers.Adam(lr=1e-2, clipnorm=1.0))
    rs = np.random.RandomState(0)
    x = rs.randn(6, 40, 9).astype(np.float32)
    labels = [rs.randint(0, 6, size=5).tolist() for _ in range(6)]
    for _ in range(3):
        m = model.train_on_batch([x, labels, [40] * 6])
    out[mode] = [float(np.abs(w).sum()) for w in model.get_weights()] + [m[0]]
    model.loss_and_grads(model.to_slab(x), labels, [40] * 6, training=False)
    out["layers_reduced_during_bptt_" + mode] = len(model._ar_covered)
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.timeout(300)
def test_layerwise_allreduce_overlap_equals_single_allreduce():
    """The per-layer asynchronous all-reduces issued during BPTT (ASR_AR_OVERLAP=1, default;
    from the side stream, or from the main stream when there is none) and one all-reduce of the
    whole buffer after it give the same training trajectory (world size 1 on the box's single GPU: the collective is an
    identity, what is tested is stream ordering and buffer coverage)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', ASR_FORCE_ALLREDUCE='1',
               ASR_ROOT=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    out = subprocess.run([sys.executable, '-c', _SCRIPT], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280,
                         stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('RESULT ')][0]
    res = json.loads(line[7:])
    assert res['1'] == res['0'] == res['capi']      # capi: asr_comm_* instead of torch
    # no side stream (cfg3's schedule): per-layer all-reduces from the main stream == one at the
    # end; against the side-stream schedule only the frame-range pipelining differs (the
    # gradient GEMMs' power-of-two pre-scale is taken per slice there): last-bit differences
    assert res['noside'] == res['noside0']
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(res['noside'], res['1']))
    assert res['layers_reduced_during_bptt_1'] == 2 and res['layers_reduced_during_bptt_0'] == 0
    assert res['layers_reduced_during_bptt_noside'] == 2
    assert res['layers_reduced_during_bptt_noside0'] == 0
