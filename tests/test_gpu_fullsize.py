"""-m gpu: properties of the hot path at BASELINE.json's FULL cfg2 size (brsmv1
5xBiLSTM(256), 32 utterances x 10 s = 999 frames), where the float64 oracle would need
minutes (and the same at cfg3: 5xBiLSTM(512), 64 utterances, 80 log-mel features):
size-independent invariants instead of element-wise comparison.

* utterances are independent: the logits / CTC losses of a 32-utterance batch equal, bit
  for bit, those of its two 16-utterance halves (no arithmetic crosses a batch row);
* gradients are additive over utterances: grad(batch of 32, scale 1/32) equals the sum of
  the two half-batch gradients computed with the same 1/32 scale, to fp32 round-off
  (the split-fp16 GEMM pre-scale depends on the batch maximum, so not bit-exact);
* one Adam step with global-norm clipping moves every weight by at most lr (|m|/sqrt(v)
  <= 1 at step 1) and leaves the persistent kernels' status words clean;
* the front-end computes an utterance identically alone and inside a batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T_SAMPLES = 160000


def _inputs(n, seed):
    rs = np.random.RandomState(seed)
    sigs = [rs.randn(T_SAMPLES).astype(np.float32) for _ in range(n)]
    labels = [rs.randint(0, 25, size=rs.randint(2, 50)).tolist() for _ in range(n)]
    return sigs, labels


CASES = {
    # BASELINE.json configs[1] / configs[2] at full size
    'cfg2': dict(F=39, H=256, N=32, feat='mfcc'),
    'cfg3': dict(F=80, H=512, N=64, feat='logfbank80'),
}


@pytest.mark.timeout(900)
@pytest.mark.parametrize('name', ['cfg2', 'cfg3'])
def test_fullsize_batch_rows_are_independent_and_gradients_additive(name):
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    from asr_study_amd.preprocessing import audio
    cfg = CASES[name]
    N, half = cfg['N'], cfg['N'] // 2
    dev = torch.device('cuda:0')
    model = models.brsmv1(num_features=cfg['F'], num_classes=28, num_hiddens=cfg['H'],
                          num_layers=5, dropout=0.0, weight_decay=1e-4, seed=0, device=dev)
    model.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    feat = audio.MFCC(device=dev) if cfg['feat'] == 'mfcc' else \
        audio.LogFbank(num_filt=80, device=dev)
    sigs, labels = _inputs(N, 5)
    slab, frames = feat.batch(sigs)
    assert slab.shape == (999, N, cfg['F']) and int(frames.min()) == 999
    lens = [999] * N

    def run(rows):
        sub = slab[:, rows].contiguous()
        ctc, logits, _ = model.loss_and_grads(sub, [labels[i] for i in rows],
                                              [lens[i] for i in rows], training=False,
                                              n_global=N)
        torch.cuda.synchronize()
        return (ctc.cpu().numpy().copy(), logits[:, :len(rows)].cpu().numpy().copy(),
                model.grads.cpu().numpy().copy())
    full_ctc, full_logits, full_grad = run(list(range(N)))
    a_ctc, a_logits, a_grad = run(list(range(half)))
    b_ctc, b_logits, b_grad = run(list(range(half, N)))
    for ws in ('lstm_fwd', 'lstm_bwd'):
        ops.lstm_status(ops.WS.get(ws, 0, dev))
    assert np.all(np.isfinite(full_ctc)) and np.all(full_ctc > 0)
    # forward: bit-exact independence of batch rows
    assert np.array_equal(full_logits[:, :half], a_logits)
    assert np.array_equal(full_logits[:, half:], b_logits)
    assert np.array_equal(full_ctc, np.concatenate([a_ctc, b_ctc]))
    # backward: additivity to fp32 round-off of a 32k-64k-term reduction
    gsum = a_grad + b_grad
    scale = np.abs(full_grad).max()
    assert scale > 0
    err = np.abs(full_grad - gsum).max()
    print('%s gradient additivity: max|g|=%.3e max err=%.3e' % (name, scale, err))
    assert err < 2e-5 * scale
    # one clipped Adam step: bounded update, finite weights
    before = model.params.clone()
    model._step += 1
    model.optimizer.step(model)
    torch.cuda.synchronize()
    delta = (model.params - before).abs().max().item()
    assert 0 < delta <= 1e-3 * 1.001
    assert torch.isfinite(model.params).all()


def test_frontend_alone_equals_in_batch():
    from asr_study_amd.preprocessing import audio
    dev = torch.device('cuda:0')
    for feat in (audio.MFCC(device=dev), audio.LogFbank(num_filt=80, device=dev)):
        sigs, _ = _inputs(5, 11)
        sigs[1] = sigs[1][:40000]
        sigs[3] = sigs[3][:401]
        slab, frames = feat.batch(sigs)
        for i, s in enumerate(sigs):
            alone, fr = feat.batch([s])
            t = int(fr[0])
            assert t == int(frames[i])
            assert torch.equal(alone[:t, 0], slab[:t, i])
            assert torch.count_nonzero(slab[t:, i]) == 0        # pad_sequences 'post'


@pytest.mark.timeout(600)
def test_compact_bptt_under_uneven_load_at_full_size_is_bit_identical():
    """The compact BPTT geometry (asr_lstm_args.compact) at cfg3's FULL size (H = 512, 64 rows,
    T = 999: 128 workgroups, one per CU) run (a) alone and (b) five times beside a stream of
    K-major weight-gradient GEMMs and packs on the other CUs -- the load it meets in
    engine.backward, uneven and at the package power cap -- against the default geometry alone:
    every word of dz, max|dz| and the bias partials bit for bit, no timeout flag.  (The hand-off
    protocol is the default kernel's; this is its test under the load the guide asks for:
    uneven, L1-warm consumers, every word checked.)"""
    from asr_study_amd import ops
    T, N, H = 999, 64, 512
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(3)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    U = rnd(2, H, 4 * H, scale=1.0 / np.sqrt(H))
    zx = rnd(T, N, 2, 4 * H)
    y = torch.empty(T, N, 2 * H, device=dev)
    cell = torch.empty(T, N, 2, H, device=dev)
    gates = torch.empty(T, N, 2, 4 * H, device=dev)
    dy = rnd(T, N, 2 * H, scale=0.01)
    mask = ((torch.rand(2, N, H, generator=g) > 0.2).float() / 0.8).to(dev)
    ops.lstm_status(ops.lstm_seq_fwd(zx, U, y, cell, gates, T, N, H, mask_u=mask))
    rows = T * N
    one = torch.ones(1, device=dev)
    pdz = ops.HlPlanes(rows, 8 * H, dev)
    px = ops.HlPlanes(rows, 2 * H, dev)
    gz = rnd(rows, 8 * H, scale=0.01)
    ops.pack_hl(gz, rows, 8 * H, absmax=ops.absmax(gz), r=pdz)
    ops.pack_hl(rnd(rows, 2 * H, scale=0.5), rows, 2 * H, absmax=one, r=px)
    yu = ops.HlPlanes(rows, H, dev)
    gW = torch.zeros(2 * H * 8 * H, device=dev)

    def load():                     # what the side stream runs beside a BPTT in the engine
        ops.pack_hl(y, rows, H, ld=2 * H, absmax=one, r=yu)
        ops.gemm_hl(yu, pdz, gW, H, 4 * H, (T - 1) * N, b_row=N, split_k='auto',
                    ws_name='gemm_side', k_major=True)
        ops.gemm_hl(px, pdz, gW, 2 * H, 8 * H, rows, split_k='auto', ws_name='gemm_side',
                    k_major=True)

    def bptt(compact):
        dz = torch.full((T, N, 2, 4 * H), 7.0, device=dev)
        dbp = torch.full((N // 16, 2, 4 * H), 9.0, device=dev)
        amax = torch.zeros(1, device=dev)
        ws = ops.lstm_seq_bwd(dy, U, cell, gates, dz, T, N, H, mask_u=mask, dz_absmax=amax,
                              db_part=dbp, compact=compact)
        return dz, dbp, amax, ws
    want = bptt(False)
    ops.lstm_status(want[3])
    alone = bptt(True)
    ops.lstm_status(alone[3])
    for a, b in zip(alone[:3], want[:3]):
        assert torch.equal(a, b)
    side = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    for rep in range(5):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(1 + rep % 3):        # a different amount of work beside it every time
                load()
        got = bptt(True)
        main.wait_stream(side)
        ops.lstm_status(got[3])
        for a, b in zip(got[:3], want[:3]):
            assert torch.equal(a, b), rep


@pytest.mark.timeout(300)
def test_configs2_as_written_runs_its_first_steps_without_a_vetoed_step():
    """Regression (round 6): with BPTT writing dz as packed planes the first bound policy (8 x the
    measured maximum, followed at once) vetoed a step in EVERY run of configs[2] as written -- the
    gate gradients of the layer above the conv front-end jump ~800 x between the first two steps
    and fall ~5000 x in the third.  bench.py asserts `no timeout / fallback during the timed
    steps`: eight steps from fresh weights must pass it."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', 'cfg3_conv',
                          '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-extras'],
                         cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280,
                         stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-1500:]
    line = json.loads([ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert line['fallbacks'] == 0 and line['config']['workload'].startswith('cfg3')
