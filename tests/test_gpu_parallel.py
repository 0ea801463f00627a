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
for mode in ("1", "0", "noside", "noside0"):
    os.environ["ASR_AR_OVERLAP"] = {"noside": "1", "noside0": "0"}.get(mode, mode)
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
# ASR_AR_OVERLAP=auto (the default): per-layer asynchronous all-reduces only while a recurrence
# leaves CUs free; 2 x 4 batch tiles x 32 workgroups (BiLSTM(512), batch 64) fill the chip
os.environ.pop("ASR_AR_OVERLAP"); os.environ["ASR_OVERLAP"] = "auto"
out["layers_reduced_during_bptt_auto_small"] = None
model.loss_and_grads(model.to_slab(x), labels, [40] * 6, training=False)
out["layers_reduced_during_bptt_auto_small"] = len(model._ar_covered)
big = models.brsmv1(num_features=16, num_classes=7, num_hiddens=512, num_layers=3, dropout=0.0,
                    seed=1)
xb = rs.randn(64, 6, 16).astype(np.float32)
lb = [rs.randint(0, 6, size=2).tolist() for _ in range(64)]
big.loss_and_grads(big.to_slab(xb), lb, [6] * 64, training=False)
out["layers_reduced_during_bptt_auto_chipfill"] = len(big._ar_covered)
out["compact_launches_chipfill"] = big._compact_launches
g_layerwise = big.grads.clone()
# ... and none without the compact schedule (every BPTT fills the chip: one collective behind them)
big._compact_mode = "0"
big.loss_and_grads(big.to_slab(xb), lb, [6] * 64, training=False)
out["layers_reduced_during_bptt_auto_chipfill_serial"] = len(big._ar_covered)
big._allreduce()
torch.cuda.synchronize()
g_single = big.grads.clone()
big._compact_mode = "auto"
big.loss_and_grads(big.to_slab(xb), lb, [6] * 64, training=False)
big._allreduce()
torch.cuda.synchronize()
out["chipfill_layerwise_equals_single_collective"] = bool(torch.equal(big.grads, g_single))
big._compact_mode = "0"
# the decision is taken on the rank-invariant reference shard pad16(ceil(n_global / world)),
# not on this rank's own n_pad: a 48-row local shard of a global batch whose reference shard
# pads to 64 rows (fills the chip; compact schedule off) must NOT start per-layer collectives
# (ADVICE r3)
x48 = rs.randn(48, 6, 16).astype(np.float32)
l48 = [rs.randint(0, 6, size=2).tolist() for _ in range(48)]
big.loss_and_grads(big.to_slab(x48), l48, [6] * 48, training=False, n_global=49, n_ref=49)
out["layers_reduced_ragged_ref64_local48"] = len(big._ar_covered)
big.loss_and_grads(big.to_slab(x48), l48, [6] * 48, training=False, n_global=48, n_ref=48)
out["layers_reduced_ragged_ref48_local48"] = len(big._ar_covered)
big._compact_mode = "auto"
big.loss_and_grads(big.to_slab(x48), l48, [6] * 48, training=False, n_global=49, n_ref=49)
out["layers_reduced_ragged_ref64_local48_compact"] = len(big._ar_covered)
from asr_study_amd import parallel
out["collectives_through_capi"] = parallel.CapiComm.get().calls
# the veto of an update is collective: the flags travel behind the gradients through the
# all-reduce and the guard reads the reduced slots
from asr_study_amd import ops
model.train_on_batch([x, labels, [40] * 6])
before = model.params.clone()
it0 = model.optimizer.iterations
ops.WS.get("lstm_bwd", 0, model.device)[:4].view(torch.int32)[0] = 1
r = model.train_on_batch([x, labels, [40] * 6], sync=False)
torch.cuda.synchronize()
out["veto_slots"] = model._gbuf[model.n_params:].cpu().tolist()
out["veto_params_unchanged"] = bool(torch.equal(model.params, before))
out["veto_metrics_none"] = model._lagged((r, labels, 6)) is None
out["veto_state"] = [model.fallbacks, model.lstm_mode, model.optimizer.iterations - it0]
m2 = model.train_on_batch([x, labels, [40] * 6])
out["veto_recovered"] = bool(np.isfinite(m2[0]) and not torch.equal(model.params, before))
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.timeout(300)
def test_layerwise_allreduce_overlap_equals_single_allreduce():
    """The per-layer asynchronous all-reduces issued during BPTT (ASR_AR_OVERLAP=1; the default
    `auto` issues them only where a recurrence leaves CUs free; behind the side stream, or behind
    the main stream when there is none; all through asr_comm_* on the communicator's stream) and
    one all-reduce of the
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
    assert res['1'] == res['0']
    # every collective of the run went through the library's own RCCL entry points
    assert res['collectives_through_capi'] > 10
    assert res['layers_reduced_ragged_ref64_local48'] == 0
    assert res['layers_reduced_ragged_ref48_local48'] == 2
    # no side stream (cfg3's schedule): per-layer all-reduces from the main stream == one at the
    # end; against the side-stream schedule only the frame-range pipelining differs (the
    # gradient GEMMs' power-of-two pre-scale is taken per slice there): last-bit differences
    assert res['noside'] == res['noside0']
    assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(res['noside'], res['1']))
    assert res['layers_reduced_during_bptt_1'] == 2 and res['layers_reduced_during_bptt_0'] == 0
    assert res['layers_reduced_during_bptt_noside'] == 2
    assert res['layers_reduced_during_bptt_noside0'] == 0
    assert res['layers_reduced_during_bptt_auto_small'] == 2
    # chip-filling recurrences (3 x BiLSTM(512), batch 64): with the compact backward schedule
    # (r6) the buckets of the two layers above the bottom one go out beside the compact BPTTs
    # below them; with ASR_BPTT_COMPACT=0 every BPTT fills the chip and there is ONE collective
    # behind them; the reduced gradients are bit-equal either way
    assert res['compact_launches_chipfill'] == 2
    assert res['layers_reduced_during_bptt_auto_chipfill'] == 2
    assert res['layers_reduced_during_bptt_auto_chipfill_serial'] == 0
    assert res['chipfill_layerwise_equals_single_collective']
    # (the reference shard decides, not the local one: a 48-row shard of a batch whose reference
    # shard fills the chip follows the chip-filling schedule of the other ranks)
    assert res['layers_reduced_ragged_ref64_local48_compact'] == 2
    assert res['veto_slots'] == [0.0, 1.0, 0.0, 0.0] and res['veto_params_unchanged']
    assert res['veto_metrics_none'] and res['veto_state'] == [1, 1, 0] and res['veto_recovered']


def _cfg3_layer(T, seed=0):
    import numpy as np
    import torch
    rs = np.random.RandomState(seed)
    H, n_pad = 512, 64
    dev = torch.device('cuda:0')

    def t(*shape, scale=1.0):
        return torch.from_numpy((rs.randn(*shape) * scale).astype(np.float32)).to(dev)
    U = t(2, H, 4 * H, scale=0.04)
    zx = t(T, n_pad, 2, 4 * H)
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    dy = t(T, n_pad, 2 * H, scale=1e-3)
    return dict(T=T, n_pad=n_pad, H=H, U=U, zx=zx, y=y, cell=cell, gates=gates, dy=dy)


@pytest.mark.timeout(120)
def test_foreign_kernel_holding_cus_delays_a_chip_filling_recurrence_without_a_timeout():
    """The hazard of a collective beside a cfg3 recurrence: 2 x 4 x 32 spin-waiting workgroups
    need all 256 CUs.  A foreign kernel (asr_debug_occupy: the stand-in for an RCCL kernel)
    that holds 64 CUs for 30 ms on another stream when the recurrence starts only DELAYS it
    (measured on MI355X: the dispatcher does not start the 256-workgroup grid piecemeal beside
    the foreign kernel; the recurrence ends when that kernel does): no timeout flag,
    bit-identical activations and gate gradients."""
    import torch
    from asr_study_amd import ops
    L = _cfg3_layer(T=120)
    dev = L['zx'].device
    side = torch.cuda.Stream(device=dev)
    res = []
    for hog in (False, True):
        for k in ('y', 'cell', 'gates'):
            L[k].zero_()
        dz = torch.zeros(L['T'], L['n_pad'], 2, 4 * L['H'], device=dev)
        torch.cuda.synchronize()
        if hog:
            with torch.cuda.stream(side):
                ops.debug_occupy(64, 96 * 1024, 0.03)
        ops.lstm_seq_fwd(L['zx'], L['U'], L['y'], L['cell'], L['gates'], L['T'], L['n_pad'], L['H'])
        if hog:
            with torch.cuda.stream(side):
                ops.debug_occupy(64, 96 * 1024, 0.03)
        ops.lstm_seq_bwd(L['dy'], L['U'], L['cell'], L['gates'], dz, L['T'], L['n_pad'], L['H'])
        torch.cuda.synchronize()
        assert not ops.lstm_timeout_flags(dev).any().item()
        res.append((L['y'].clone(), dz))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.timeout(240)
def test_cu_starved_recurrence_times_out_is_vetoed_and_the_step_is_recovered(monkeypatch):
    """A chip-filling recurrence that does NOT get all its workgroups resident: the whole step
    runs on a stream confined to 64 of the 256 CUs (asr_stream_create_cu_mask), so only a
    quarter of the 2 x 4 x 32 spin-waiting workgroups are resident at a time and their bounded
    spins (shortened to 30 ms for the test) give up -- a REAL timeout, not a flag set by hand.
    The step's update is vetoed on the device, the engine runs the batch again on the stepwise
    kernels (no co-residency needed) and ends up with the weights of an undisturbed run; the
    persistent kernels come back after the retry gap."""
    import numpy as np
    import torch
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    monkeypatch.setenv('ASR_LSTM_SPIN_MS', '30')
    rs = np.random.RandomState(3)
    N, T, F, C = 64, 24, 16, 7
    x = rs.randn(N, T, F).astype(np.float32)
    labels = [rs.randint(0, C - 1, size=3).tolist() for _ in range(N)]
    batch = [x, labels, [T] * N]

    def fresh():
        m = models.brsmv1(num_features=F, num_classes=C, num_hiddens=512, num_layers=1,
                          dropout=0.0, weight_decay=1e-4, seed=4)
        m.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=1.0))
        return m
    model, ref = fresh(), fresh()
    model._retry_gap = 2
    model.train_on_batch(batch)
    ref.train_on_batch(batch)
    assert model._recurrence_fills_chip(64)
    dev = model.device
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    masked = ops.cu_masked_stream(dev, 64, total)
    torch.cuda.synchronize()
    with torch.cuda.stream(masked):
        out = model.train_on_batch(batch)
    torch.cuda.synchronize()
    want = ref.train_on_batch(batch)
    assert model.fallbacks == 1 and model.vetoed_steps == 1 and model.lstm_mode == 1
    assert model.optimizer.iterations == ref.optimizer.iterations == 2
    assert np.allclose(out, want, rtol=1e-4, atol=1e-5)
    assert (model.params - ref.params).abs().max().item() < 1e-5
    for _ in range(3):
        model.train_on_batch(batch)
        ref.train_on_batch(batch)
    assert model.lstm_mode == 0 and model.fallbacks == 1
    assert (model.params - ref.params).abs().max().item() < 1e-5
    assert not ops.lstm_timeout_flags(dev).any().item()
