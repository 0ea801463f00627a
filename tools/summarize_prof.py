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
    log = os.path.join(d, 'bench_stdout.log')
    if os.path.exists(log):
        for ln in open(log):
            if ln.startswith('{'):
                lines += ['', 'bench.py line of the same run:', '', '```json', ln.strip(), '```']
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[:24]))


if __name__ == '__main__':
    main()
