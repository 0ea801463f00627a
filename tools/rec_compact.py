#!/usr/bin/env python
"""The compact BPTT geometry (asr_lstm_args.compact: H/32 workgroups per chain, 128 of 256 CUs at
cfg3) against the default one, alone and with the K-major weight-gradient GEMMs of a cfg3 layer
running beside it on a second stream -- the schedule engine.backward uses.

    python tools/rec_compact.py [cfg3|cfg3c]      (cfg3c: T = 500, the stack behind the conv front-end)

Prints us per BPTT step for: default alone, compact alone, default + GEMMs one after the other
(the serial schedule: sum), compact with the GEMMs beside it (wall time of the pair), and what
the GEMMs cost alone on the whole chip."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_study_amd import ops  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
    T = 500 if name == 'cfg3c' else 999
    N, H = 64, 512
    n_pad = ops.pad16(N)
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)

    def rnd(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev)
    U = rnd(2, H, 4 * H, scale=1.0 / np.sqrt(H))
    zx = rnd(T, n_pad, 2, 4 * H)
    y = torch.empty(T, n_pad, 2 * H, device=dev)
    cell = torch.empty(T, n_pad, 2, H, device=dev)
    gates = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    dy = rnd(T, n_pad, 2 * H, scale=0.01)
    dz = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    dz2 = torch.empty(T, n_pad, 2, 4 * H, device=dev)
    ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
    rows = T * n_pad
    # the weight-gradient work of the layer above: dU (2 x) and dW, K-major on packed planes
    one = torch.ones(1, device=dev)
    pdz = ops.HlPlanes(rows, 8 * H, dev)
    px = ops.HlPlanes(rows, 2 * H, dev)
    gz = rnd(rows, 8 * H, scale=0.01)
    gx = rnd(rows, 2 * H, scale=0.5)
    ops.pack_hl(gz, rows, 8 * H, absmax=ops.absmax(gz), r=pdz)
    ops.pack_hl(gx, rows, 2 * H, absmax=one, r=px)
    yu = [ops.HlPlanes(rows, H, dev) for _ in range(2)]
    gW = torch.zeros(2 * H * 8 * H + 2 * H * 4 * H, device=dev)

    def wgrads(wsn='gemm_side'):
        kk = (T - 1) * n_pad
        for d in range(2):
            ops.pack_hl(y, rows, H, ld=2 * H, src_off=d * H, absmax=one, r=yu[d])
            ops.gemm_hl(yu[d], pdz, gW, H, 4 * H, kk, a_row=0 if d == 0 else n_pad,
                        b_k=d * 4 * H, b_row=n_pad if d == 0 else 0,
                        c_off=2 * H * 8 * H + d * H * 4 * H, split_k='auto', ws_name=wsn,
                        k_major=True)
        ops.gemm_hl(px, pdz, gW, 2 * H, 8 * H, rows, c_off=0, split_k='auto', ws_name=wsn,
                    k_major=True)

    def bptt(compact, out=dz):
        return ops.lstm_seq_bwd(dy, U, cell, gates, out, T, n_pad, H, compact=compact)
    side = torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream(dev)

    def pair(compact):
        ev = torch.cuda.Event()
        bptt(compact)
        ev.record(main_s)            # (nothing to wait for: the GEMMs' inputs are ready)
        with torch.cuda.stream(side):
            wgrads()
        main_s.wait_stream(side)

    def serial():
        bptt(False)
        wgrads('gemm')

    def timeit(fn, reps=4):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best
    def pair_detail(compact, reps=4):
        """durations of each stream's own work in the overlapped schedule (HIP events per stream)"""
        tb = tg = 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side.wait_stream(main_s)
            b0.record(main_s)
            bptt(compact)
            b1.record(main_s)
            with torch.cuda.stream(side):
                g0.record(side)
                wgrads()
                g1.record(side)
            main_s.wait_stream(side)
            torch.cuda.synchronize()
            tb += b0.elapsed_time(b1) / reps
            tg += g0.elapsed_time(g1) / reps
        return tb, tg
    for _ in range(4):
        bptt(False); bptt(True); wgrads('gemm')
    torch.cuda.synchronize()
    # bit-exactness of the two geometries on this input
    bptt(False, dz); bptt(True, dz2)
    torch.cuda.synchronize()
    print('compact == default bit for bit:', bool(torch.equal(dz, dz2)))
    t_def = timeit(lambda: bptt(False))
    t_cmp = timeit(lambda: bptt(True))
    t_gemm = timeit(lambda: wgrads('gemm'))
    t_ser = timeit(serial)
    t_pair = timeit(lambda: pair(True))
    t_pair_def = timeit(lambda: pair(False))
    ws = ops.WS.get('lstm_bwd', 0, dev)
    ops.lstm_status(ws)
    print('%s T=%d: BPTT default alone  %.3f ms = %.3f us/step' % (name, T, t_def, t_def * 1e3 / T))
    print('%s T=%d: BPTT compact alone  %.3f ms = %.3f us/step' % (name, T, t_cmp, t_cmp * 1e3 / T))
    print('%s weight-gradient GEMMs + y packs alone (whole chip) %.3f ms' % (name, t_gemm))
    print('%s serial  (default BPTT, then the GEMMs)              %.3f ms' % (name, t_ser))
    print('%s overlap (compact BPTT || GEMMs on a side stream)    %.3f ms  (%.3f us/step wall)' %
          (name, t_pair, t_pair * 1e3 / T))
    print('%s overlap with the DEFAULT geometry (GEMMs queue)      %.3f ms' % (name, t_pair_def))
    print('%s gain per layer: %.3f ms' % (name, t_ser - t_pair))
    tb, tg = pair_detail(True)
    print('%s inside the overlap: compact BPTT %.3f ms (%.3f us/step; alone %.3f), GEMM stream '
          '%.3f ms (alone on the whole chip %.3f)' % (name, tb, tb * 1e3 / T, t_cmp, tg, t_gemm))
    # shader clock of the compact BPTT alone and beside the GEMMs: the phase profiler
    # (ASR_LSTM_DBG=32) counts shader cycles per step in workgroup 0, HIP events give the wall time
    def ghz(beside):
        os.environ['ASR_LSTM_DBG'] = '32'
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if beside:
            side.wait_stream(main_s)
        e0.record(main_s)
        bptt(True)
        e1.record(main_s)
        if beside:
            with torch.cuda.stream(side):
                wgrads()
            main_s.wait_stream(side)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / T
        pr = ops.lstm_profile(ops.WS.get('lstm_bwd', 0, dev))
        os.environ.pop('ASR_LSTM_DBG', None)
        clk = sum(pr[0]) / float(T - 1)
        return us, clk, clk / us / 1e3
    for beside in (False, True, False, True):
        us, clk, g_ = ghz(beside)
        print('%s compact BPTT %s: %.3f us/step (profiled), %.0f shader clocks/step -> %.2f GHz' %
              (name, 'beside the GEMMs' if beside else 'alone           ', us, clk, g_))
    # the GEMM stream alone on a 128-CU mask (what half of the chip is worth without a neighbour)
    try:
        half = ops.cu_masked_stream(dev, 128, 256)
        def masked():
            half.wait_stream(main_s)
            with torch.cuda.stream(half):
                wgrads()
            main_s.wait_stream(half)
        print('%s GEMM stream alone on a 128-CU masked stream: %.3f ms' % (name, timeit(masked)))
    except Exception as e:
        print('cu mask unavailable:', repr(e)[:200])


if __name__ == '__main__':
    main()
