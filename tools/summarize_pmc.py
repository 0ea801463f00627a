#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each in
its own run, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Counter unit: KiB-ish "kilobytes"
as rocprofv3 reports them; the gfx950 correction from the guide's HBM section (FETCH_SIZE
tallies 128-B requests at 64 B for wide coalesced reads) is applied as a x2 column.
Usage: summarize_pmc.py <dir_with_pmc_FETCH_SIZE_and_pmc_WRITE_SIZE> <out.md> [config]"""
import collections
import csv
import os
import re
import sys


def short(name):
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


def main():
    d, out = sys.argv[1:3]
    config = sys.argv[3] if len(sys.argv) > 3 else 'cfg2'
    rd, wr = load(d, 'FETCH_SIZE'), load(d, 'WRITE_SIZE')
    names = sorted(set(rd) | set(wr), key=lambda k: -(sum(rd.get(k, [0])) * 2 + sum(wr.get(k, [0]))))
    lines = ['# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs)',
             '', 'Command: `rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python bench.py '
             '--steps 3 --warmup 1 --no-cpu-baseline --no-extras` (%s).  MB = counter kilobytes / 1000.' % config,
             '"fetch x2" applies the gfx950 correction for wide coalesced reads; writes are '
             'uncalibrated (taken as reported).', '',
             '| kernel | launches | fetch MB | fetch x2 MB | write MB | traffic MB (fetch x2 + write) |',
             '|---|---:|---:|---:|---:|---:|']
    for k in names[:18]:
        r, w = rd.get(k, []), wr.get(k, [])
        fr = sum(r) / max(len(r), 1) / 1e3
        fw = sum(w) / max(len(w), 1) / 1e3
        lines.append('| `%s` | %d | %.2f | %.2f | %.2f | %.2f |' % (k, max(len(r), len(w)), fr,
                                                                   2 * fr, fw, 2 * fr + fw))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
