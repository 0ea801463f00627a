#!/usr/bin/env python
"""Headline benchmark: audio-seconds/second TRAINED on the acoustic hot path
(MFCC/log-mel front-end -> BiLSTM stack fwd/bwd -> CTC loss/grad/greedy decode ->
global-norm clip + Adam, + RCCL gradient all-reduce when N > 1), synthetic 16 kHz
10 s utterances, random-init weights of the named topology.

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; weak scaling (per-GPU batch fixed).  Rank 0 prints ONE JSON
line.  A "step" = one pass of the hot path over one mini-batch whose samples are
already resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1]: brsmv1 defaults, batch 32
    'cfg2': dict(model='brsmv1', F=39, H=256, L=5, C=28, N=32, feat='mfcc',
                 desc='brsmv1 5xBiLSTM(256), MFCC-39, 28-class CTC, batch 32 x 10 s @16 kHz'),
    # BASELINE.json configs[2]: 5xBiLSTM(512), 80-dim log-mel, batch 64 (the conv
    # front-end named there does not exist in the reference, SURVEY.md 8: not built)
    'cfg3': dict(model='brsmv1', F=80, H=512, L=5, C=28, N=64, feat='logfbank80',
                 desc='5xBiLSTM(512), log-mel-80, 28-class CTC, batch 64 x 10 s @16 kHz'),
}
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-in MFMA = fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
SAMPLES = 160000                  # 10 s @ 16 kHz -> T = 999 frames


def cpu_baseline(cfg):
    """The oracle (a NumPy port of the reference algorithm: per-timestep two-matmul
    LSTM loop, CPU CTC, BPTT, Adam) timed on this host on a BOUNDED sample of the
    same workload: same topology, 4 utterances of 2 s (T = 199)."""
    from oracle import frontend as OF
    from oracle import lstm as OL
    from oracle import optim as OO
    n, secs = 4, 2.0
    rs = np.random.RandomState(0)
    sigs = [rs.randn(int(16000 * secs)).astype(np.float32) for _ in range(n)]
    labels = [rs.randint(0, 25, size=rs.randint(2, 20)).tolist() for _ in range(n)]
    params = OL.init_model(seed=0, num_features=cfg['F'], num_hiddens=cfg['H'],
                           num_layers=cfg['L'], num_classes=cfg['C'], dtype=np.float32)
    opt = OO.Adam(lr=1e-3, clipnorm=400.0)
    kind, kw = ('mfcc', {}) if cfg['feat'] == 'mfcc' else ('logfbank', {'num_filt': 80})

    def step():
        feats = [OF.extract(kind, s.astype(np.float64), **kw).astype(np.float32) for s in sigs]
        x = np.stack(feats, axis=1)                       # (T, N, F)
        out = OL.loss_and_grads(params, x, labels, [x.shape[0]] * n, weight_decay=1e-4)
        opt.step([a for _, a in OL.flatten(params)], [a for _, a in OL.flatten(out['grads'])])
    step()                                                # warm caches / BLAS threads
    t0 = time.time()
    reps = 0
    while reps < 2 or time.time() - t0 < 10.0:
        step()
        reps += 1
        if time.time() - t0 > 30.0:
            break
    dt = (time.time() - t0) / reps
    try:
        import threadpoolctl
        threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {'value': round(n * secs / dt, 2), 'unit': 'audio-seconds/s', 'cores': int(threads),
            'kind': 'port',
            'sample': '%d utterances x %.0f s (T=199), same topology, %d step(s), NumPy/BLAS '
                      'float32 oracle port incl. front-end, CTC, BPTT, Adam' % (n, secs, reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default=os.environ.get('ASR_BENCH_CONFIG', 'cfg2'),
                    choices=sorted(CONFIGS))
    ap.add_argument('--dropout', type=float, default=0.2)      # brsmv1 default
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the hot path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    force_dist = os.environ.get('ASR_FORCE_ALLREDUCE') == '1'   # exercise RCCL at world 1 (tests)
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import __graft_entry__
    __graft_entry__.build()
    from asr_study_amd import ops
    from asr_study_amd.core import models, optimizers
    from asr_study_amd.preprocessing import audio

    N, F, H, L, C = cfg['N'], cfg['F'], cfg['H'], cfg['L'], cfg['C']
    model = models.brsmv1(num_features=F, num_classes=C, num_hiddens=H, num_layers=L,
                          dropout=args.dropout, weight_decay=1e-4, seed=0, device=dev)
    model.compile(optimizer=optimizers.Adam(lr=1e-3, clipnorm=400))
    feat = audio.MFCC(device=dev) if cfg['feat'] == 'mfcc' else audio.LogFbank(num_filt=80, device=dev)

    # ---- synthetic inputs, resident in HBM before the timed region
    sig = np.empty(N * SAMPLES, np.float32)
    for i in range(N):
        sig[i * SAMPLES:(i + 1) * SAMPLES] = np.random.RandomState(1000 * rank + i).randn(SAMPLES)
    audio_d = torch.from_numpy(sig).to(dev)
    offs = (torch.arange(N, dtype=torch.int32) * SAMPLES).to(dev)
    lens = torch.full((N,), SAMPLES, dtype=torch.int32, device=dev)
    host_lens = [SAMPLES] * N
    rs = np.random.RandomState(77 + rank)
    lab_len = rs.randint(2, 50, size=N)
    lab = np.zeros((N, 49), np.int32)
    for n in range(N):
        lab[n, :lab_len[n]] = rs.randint(0, 25, size=lab_len[n])
    lab_d = torch.from_numpy(lab).to(dev)
    lab_len_d = torch.from_numpy(lab_len.astype(np.int32)).to(dev)

    lstm_ev = {'lstm_seq_fwd': [], 'lstm_seq_bwd': []}

    def _timed(name):
        orig = getattr(ops, name)

        def timed(*a, **k):
            # HIP events on the stream the kernel is launched on (torch's current stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **k)
            e1.record()
            # a layer's recurrence may be launched in slices (asr_lstm_args.step_count)
            steps = k.get('steps')
            lstm_ev[name].append((e0, e1, int(steps[1]) if steps else None))
            return r
        return orig, timed

    def step(record=False):
        slab, frames = feat.batch_device(audio_d, offs, lens, host_lens)
        if record:
            saved = {}
            for name in lstm_ev:
                saved[name], wrapped = _timed(name)
                setattr(ops, name, wrapped)
            try:
                return model.train_step_device(slab, lab_d, lab_len_d, frames, N, world)
            finally:
                for name, fn in saved.items():
                    setattr(ops, name, fn)
        return model.train_step_device(slab, lab_d, lab_len_d, frames, N, world)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = step(record=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the persistent kernels must not have abandoned a spin
    for name in ('lstm_fwd', 'lstm_bwd'):
        ops.lstm_status(ops.WS.get(name, 0, dev))
    ctc = out[0].cpu().numpy()
    assert np.all(np.isfinite(ctc)), 'non-finite CTC loss in the benchmark step'

    extra = {}
    if rank == 0:
        # ---- secondary roofline figures (outside the timed region): the gate GEMM of
        # a middle layer and the CTC loss+gradient, each timed with HIP events
        T0 = 999
        n_pad0 = ops.pad16(N)
        rows = T0 * n_pad0

        def ev_time(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        xg = torch.randn(rows, 2 * H, device=dev)
        wg = torch.randn(2 * H, 8 * H, device=dev) * 0.05
        zg = torch.empty(rows, 8 * H, device=dev)
        tg = ev_time(lambda: ops.gemm(xg, wg, zg, rows, 8 * H, 2 * H))
        gf = 2.0 * rows * 8 * H * 2 * H
        # split-fp16: every fp32 product is three fp16 MFMAs, so the matrix pipes execute
        # 3x the algorithmic flops; the roofline is the dense fp16 MFMA peak
        extra['roofline_gate_gemm'] = {
            'kernel': 'gemm_f16x2_fast_kernel %dx%dx%d (x@W, one BiLSTM layer)' % (rows, 8 * H, 2 * H),
            'bound': 'mfma', 'achieved': round(3 * gf / tg / 1e9, 2), 'peak': PEAK_F16_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(3 * gf / tg / 1e9 / PEAK_F16_MFMA_TFLOPS, 4),
            'algorithmic_fp32_tflops': round(gf / tg / 1e9, 2),
            'vs_fp32_mfma_peak': round(gf / tg / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
            'avg_launch_ms': round(tg, 4),
            'note': 'achieved = executed fp16-MFMA flop/s (3 per fp32 product, fp32 accumulate); '
                    'algorithmic_fp32_tflops = 2*M*N*K / time'}
        lg = torch.randn(T0, n_pad0, C, device=dev)
        gg = torch.empty_like(lg)
        sl0 = torch.full((N,), T0, dtype=torch.int32, device=dev)
        tc = ev_time(lambda: ops.ctc_loss_grad(lg, lab_d, lab_len_d, sl0, N, grad=gg,
                                               grad_scale=1.0 / N))
        cb = 2.0 * T0 * N * C * 4
        extra['roofline_ctc'] = {
            'kernel': 'ctc_logsoftmax + ctc_alpha_beta + ctc_grad (T=%d, N=%d, C=%d)' % (T0, N, C),
            'bound': 'hbm', 'achieved': round(cb / tc / 1e6, 2), 'peak': PEAK_HBM_GBS,
            'unit': 'GB/s', 'frac': round(cb / tc / 1e6 / PEAK_HBM_GBS, 5),
            'avg_ms': round(tc, 4),
            'note': 'latency-bound: a 999-step dependent recursion per utterance (DESIGN.md 6)'}
        del xg, wg, zg, lg, gg
    if rank == 0:
        T = 999
        n_pad = ops.pad16(N)
        ms = dt / args.steps * 1e3
        value = world * N * 10.0 / (dt / args.steps)
        def lstm_times(name):
            ev = lstm_ev[name]
            if not ev:
                return None, None, None
            tot = float(sum(a.elapsed_time(b) for a, b, _ in ev))
            steps = float(sum(T if n is None else n for _, _, n in ev))
            return tot / len(ev), tot, steps        # ms per launch, total ms, total steps
        fwd_t, bwd_t = lstm_times('lstm_seq_fwd'), lstm_times('lstm_seq_bwd')
        # dominant kernels: the persistent recurrent kernels (one launch per layer and
        # pass, both directions).  Algorithmic flops per launch = 2*T*n_pad*2 dirs*H*4H
        # (h@U forward, dz@U^T in BPTT); algorithmic HBM bytes per launch:
        # fwd reads zx, writes y+cell+gates; bwd reads dy+gates+cell, writes dz.
        flops = 2.0 * T * n_pad * 2 * H * 4 * H
        slab_b = 4.0 * T * n_pad * 2 * H
        alg_bytes = {'fwd': 4 * slab_b + slab_b + slab_b + 4 * slab_b,
                     'bwd': slab_b + 4 * slab_b + slab_b + 4 * slab_b}
        # HBM traffic per LAYER from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE; per-launch
        # average x launches per layer), profiles/r1i_bench_cfg2_hbm_traffic.md; cfg2 only
        pmc = {'cfg2': {'fwd': 647.6e6, 'bwd': 654.7e6}}.get(args.config, {})

        def roof(kind, times, kernel):
            per_launch_ms, tot_ms, tot_steps = times
            if not tot_ms:
                return {'kernel': kernel, 'bound': 'mfma', 'achieved': None,
                        'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': None,
                        'traffic': None}
            # a launch covers steps_per_launch of the layer's T dependent steps: its share
            # of the layer's algorithmic flops / bytes / PMC traffic is that fraction
            share = tot_steps / T / (tot_ms / per_launch_ms)      # avg fraction of a layer
            ach = flops * (tot_steps / T) / (tot_ms * 1e-3) / 1e12
            return {'kernel': kernel, 'bound': 'mfma', 'achieved': round(ach, 3),
                    'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    'traffic': round(pmc[kind] * share, 1) if kind in pmc else None,
                    'algorithmic_bytes': round(alg_bytes[kind] * share, 1),
                    'avg_launch_ms': round(per_launch_ms, 4),
                    'steps_per_launch': round(tot_steps / (tot_ms / per_launch_ms), 1),
                    'us_per_timestep': round(tot_ms * 1e3 / tot_steps, 3),
                    'flops_per_launch': flops * share,
                    'note': 'latency-bound recurrence (T dependent steps, cross-workgroup '
                            'hand-off per step; a layer is launched in two slices when its '
                            'neighbouring GEMMs are pipelined): DESIGN.md 5'}
        line = {
            'metric': 'audio-seconds/sec trained (MFCC+BiLSTM+CTC)',
            'value': round(value, 1), 'unit': 'audio-seconds/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'arithmetic': ('fp32 storage and accumulation; matrix products as split-fp16 (hi+lo) '
                           'MFMAs, 2^-22 relative error per product (parity tolerance 1e-4); '
                           'ASR_LSTM_PREC=0 ASR_GEMM_PREC=0 selects exact fp32 MFMA')
            if os.environ.get('ASR_GEMM_PREC', '1') != '0' or
            os.environ.get('ASR_LSTM_PREC', '1') != '0' else 'exact fp32 MFMA everywhere',
            'config': {'workload': '%s: %s' % (args.config, cfg['desc']),
                       'global_batch': world * N, 'utterance_seconds': 10.0, 'frames': T,
                       'dropout': args.dropout, 'optimizer': 'adam(clipnorm=400)',
                       'parallelism': 'dp%d' % world, 'params': model.count_params()},
            'roofline': roof('bwd', bwd_t, 'lstm_bwd_kernel_h (persistent BPTT of one BiLSTM '
                                            'layer, both directions)'),
            'roofline_lstm_fwd': roof('fwd', fwd_t, 'lstm_fwd_kernel_k2 (persistent forward '
                                                     'recurrence of one BiLSTM layer)'),
        }
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(line))
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
