#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats (csv) directory into a short markdown
summary for profiles/.  Usage: summarize_prof.py <dir> <prefix> <out.md> [title]"""
import csv
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
    return (m.group(1) if m else name)[:70]


def overlap_table(trace_csv):
    """From the kernel TRACE (start / end time of every dispatch): which launches of the GEMM /
    pack / reduce kernels ran while a BPTT kernel was running (the compact schedule: weight
    gradients on the side stream beside `lstm_bwd_kernel_c<.., 2>`), and their durations apart
    from the launches that had the chip to themselves -- the `avg us` of the table above mixes
    the two, and only the second kind says anything about the kernel."""
    if not os.path.exists(trace_csv):
        return []
    rows = list(csv.DictReader(open(trace_csv)))
    ev = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    bptt = sorted((a, b) for n, a, b in ev if n.startswith('lstm_bwd_kernel'))
    if not bptt:
        return []
    import bisect
    starts = [a for a, _ in bptt]
    agg = {}
    for n, a, b in ev:
        if not (n.startswith('gemm_') or n.startswith(('pack_hl', 'pack_rows')) or n.startswith('colsum')):
            continue
        i = bisect.bisect_right(starts, b) - 1
        ov = 0
        while i >= 0 and bptt[i][1] > a - 10 ** 7:       # BPTT launches are ms apart: look back a few
            ov += max(0, min(b, bptt[i][1]) - max(a, bptt[i][0]))
            i -= 1
        beside = ov * 2 > (b - a)
        t = agg.setdefault(n, [0, 0.0, 0, 0.0])
        t[2 if beside else 0] += 1
        t[3 if beside else 1] += (b - a) / 1e3
    if not any(t[2] for t in agg.values()):
        return []
    out = ['', 'Launches that ran WHILE a BPTT kernel was running (more than half of their duration '
           'inside a `lstm_bwd_kernel*` dispatch: the side stream of the compact schedule, half of the '
           'CUs at best) against the launches that did not:', '',
           '| kernel | alone: calls | avg us | beside BPTT: calls | avg us |', '|---|---:|---:|---:|---:|']
    for n, t in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
        out.append('| `%s` | %d | %s | %d | %s |' % (
            n, t[0], '%.2f' % (t[1] / t[0]) if t[0] else '-', t[2],
            '%.2f' % (t[3] / t[2]) if t[2] else '-'))
    return out


def main():
    d, prefix, out = sys.argv[1:4]
    title = sys.argv[4] if len(sys.argv) > 4 else prefix
    rows = list(csv.DictReader(open(os.path.join(d, prefix + '_kernel_stats.csv'))))
    total = sum(float(r['TotalDurationNs']) for r in rows)
    lines = ['# %s' % title, '',
             'Source: `rocprofv3 --kernel-trace --stats --output-format csv` (raw csv next to '
             'this file).', '', '| kernel | calls | total ms | avg us | % of GPU time |',
             '|---|---:|---:|---:|---:|']
    for r in rows[:16]:
        lines.append('| `%s` | %s | %.3f | %.2f | %.2f |' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, 100.0 * float(r['TotalDurationNs']) / total))
    lines.append('')
    lines.append('Total kernel time: %.3f ms over the traced run.' % (total / 1e6))
    lines += overlap_table(os.path.join(d, prefix + '_kernel_trace.csv'))
    log = os.path.join(d, 'bench_stdout.log')
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith('{'):
                lines += ['', 'bench.py line of the same run:', '', '```json', ln.strip(), '```']
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:24]))


if __name__ == '__main__':
    main()
