#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in
its own run, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter unit: KiB-ish "kilobytes"
as rocprofv3 reports them; the gfx950 correction from the guide's HBM section (FETCH_SIZE
tallies 128-B requests at 64 B for wide coalesced reads) is applied as a x2 column.

EVERY kernel of the run is listed (no top-N cut: the round-3 table silently dropped the CTC
and front-end kernels), with its launch count, so that each `roofline*.traffic` figure of the
bench line can be recomputed from the table alone.  With --json the families bench.py quotes
(recurrences per layer, packed GEMM per launch, the three CTC kernels per training step, the
front-end per batch, the packs per step) are written into profiles/pmc_traffic.json.

Usage: summarize_pmc.py <dir_with_pmc_FETCH_SIZE_and_pmc_WRITE_SIZE> <out.md> [config]
                        [--json profiles/pmc_traffic.json] [--layers 5] [--source <tag>]"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)      # an un-demangled anonymous-namespace kernel
    if m:
        name = name[m.end():m.end() + int(m.group(1))]
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^>]*>)?)', name)
    return (m.group(1) if m else name)[:60]


def load(d, counter):
    agg = collections.defaultdict(list)
    path = os.path.join(d, 'pmc_' + counter, 'bench_counter_collection.csv')
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return agg


def families(rd, wr, layers):
    """Bytes per unit of the kernel families bench.py quotes.  Recurrences (16-byte loads: the
    guide's x2 correction applies): sum over the family's launches of (2 x FETCH_SIZE +
    WRITE_SIZE), per training step profiled (= launches of adam_kernel) and layer.  Packed GEMM:
    per launch.  Families launched once per step and kernel (CTC loss+gradient, front-end,
    optimiser, greedy decode): the SUM OF THE PER-LAUNCH AVERAGES of their kernels -- bench.py
    launches some of them again outside the steps, so a per-step division would be wrong.  The
    CTC kernels read 4 bytes per lane, where the guide leaves FETCH_SIZE uncalibrated: `ctc` is
    the uncorrected sum (fetch + write) and `ctc_fetch_x2` the upper bound with the x2."""
    def names_of(pred):
        return [k for k in set(rd) | set(wr) if pred(k)]

    def total(pred):
        names = names_of(pred)
        t = sum(2 * sum(rd.get(k, [])) + sum(wr.get(k, [])) for k in names) * 1e3
        n = sum(max(len(rd.get(k, [])), len(wr.get(k, []))) for k in names)
        return t, n

    def per_launch_sum(pred, fetch_mult):
        t = 0.0
        for k in names_of(pred):
            r, w = rd.get(k, []), wr.get(k, [])
            t += fetch_mult * sum(r) / max(len(r), 1) + sum(w) / max(len(w), 1)
        return t * 1e3
    steps = max([len(v) for k, v in list(rd.items()) + list(wr.items()) if k.startswith('adam_kernel')] + [1])

    def per_step_sum(pred, fetch_mult):
        # kernels launched several times per step (the conv front-end's packs, band builds and
        # folds): every launch counts
        t = 0.0
        for k in names_of(pred):
            t += fetch_mult * sum(rd.get(k, [])) + sum(wr.get(k, []))
        return t * 1e3 / steps
    out = {'_steps_profiled': steps}
    # (r6) BPTT kernels that write dz as packed planes (`lstm_bwd_kernel_c<.., true>`) are preceded,
    # in the FIRST step of a run only, by one MEASURING pass per layer of the fp32-slab kernel
    # (`<.., false>`, engine.backward): not part of a steady-state step, so left out of `bwd`
    planes_on = any(k.startswith('lstm_bwd_kernel_c<') and k.rstrip().endswith('true>')
                    for k in set(rd) | set(wr))

    def is_bwd(k):
        if not k.startswith('lstm_bwd_kernel'):
            return False
        return not (planes_on and k.startswith('lstm_bwd_kernel_c<') and k.rstrip().endswith('false>'))
    for key, pred in (('fwd', lambda k: k.startswith('lstm_fwd_kernel')),
                      ('bwd', is_bwd),
                      ('pack', lambda k: k.startswith(('pack_hl', 'pack_rows')))):
        t, n = total(pred)
        if n:
            out[key] = round(t / (steps * (1 if key == 'pack' else layers)), 1)
            out[key + '_launches_per_step'] = round(n / float(steps), 2)
    is_ctc = lambda k: k.startswith('ctc_') and 'greedy' not in k and 'beam' not in k
    for key, pred, mult in (('ctc', is_ctc, 1), ('ctc_fetch_x2', is_ctc, 2),
                            ('ctc_greedy', lambda k: k.startswith('ctc_greedy'), 1),
                            ('frontend', lambda k: k.startswith('fe_'), 2),
                            ('conv', lambda k: k.startswith('conv_'), 2),
                            ('optimizer', lambda k: k.startswith('adam_kernel') or
                             k.startswith('norm_partial'), 2)):
        if names_of(pred):
            out[key] = round((per_step_sum if key == 'conv' else per_launch_sum)(pred, mult), 1)
    t, n = total(lambda k: k.startswith(('gemm_hlx_kernel', 'gemm_hlp_kernel')))
    if n:
        out['gemm'] = round(t / n, 1)               # per LAUNCH (average over both forms)
        out['gemm_launches_per_step'] = round(n / float(steps), 2)
    return out


def main():
    argv = sys.argv[1:]
    opts = {}
    for flag in ('--json', '--layers', '--source'):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    d, out = argv[:2]
    config = argv[2] if len(argv) > 2 else 'cfg2'
    rd, wr = load(d, 'FETCH_SIZE'), load(d, 'WRITE_SIZE')
    names = sorted(set(rd) | set(wr), key=lambda k: -(sum(rd.get(k, [0])) * 2 + sum(wr.get(k, [0]))))
    lines = ['# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs)',
             '', 'Command: `rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py '
             '--steps 3 --warmup 1 --no-cpu-baseline --no-extras` (%s).  MB = counter kilobytes / 1000.' % config,
             '"fetch x2" applies the gfx950 correction for wide coalesced reads; writes are '
             'uncalibrated (taken as reported).  All %d kernels of the run are listed (nothing '
             'dropped), sorted by total traffic.' % len(names), '',
             '| kernel | launches | fetch MB | fetch x2 MB | write MB | traffic MB per launch (fetch x2 + write) | total MB |',
             '|---|---:|---:|---:|---:|---:|---:|']
    for k in names:
        r, w = rd.get(k, []), wr.get(k, [])
        n = max(len(r), len(w))
        fr = sum(r) / max(len(r), 1) / 1e3
        fw = sum(w) / max(len(w), 1) / 1e3
        lines.append('| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.1f |' % (
            k, n, fr, 2 * fr, fw, 2 * fr + fw, (2 * fr + fw) * n))
    fam = families(rd, wr, int(opts.get('--layers', 5)))
    lines += ['', 'Families as `bench.py` quotes them (bytes; recurrences per layer, `gemm` per '
              'launch, the rest per training step; %d steps profiled):' % fam['_steps_profiled'], '',
              '```', json.dumps(fam, indent=1, sort_keys=True), '```']
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))
    if '--json' in opts:
        path = opts['--json']
        try:
            cur = json.load(open(path))
        except Exception:
            cur = {}
        fam['_source'] = opts.get('--source', os.path.basename(out))
        cur[config] = fam
        json.dump(cur, open(path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
