#!/usr/bin/env python
"""Per-kernel averages of every counter in a directory of rocprofv3 --pmc passes
(tools/pmc_gemm.sh) -> markdown.  Usage: pmc_parse.py <dir> <out.md> [name filter ...]"""
import collections
import csv
import glob
import re
import sys


def short(n):
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n)      # an un-demangled anonymous-namespace kernel
    if m:
        n = n[m.end():m.end() + int(m.group(1))]
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'([A-Za-z0-9_:]+(<[^>]*>)?)', n)
    return (m.group(1) if m else n)[:60]


def main():
    d0, out = sys.argv[1:3]
    filt = sys.argv[3:] or ['gemm_f16x2', 'lstm_']
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(d0 + '/*/g_counter_collection.csv')):
        group = path.split('/')[-2].split('_')[0]
        for r in csv.DictReader(open(path)):
            k = short(r['Kernel_Name'])
            if any(f in k for f in filt):
                agg[(group, k)][r['Counter_Name']].append(float(r['Counter_Value']))
    lines = ['# PMC counters per kernel (rocprofv3 --pmc, one counter group per run; averages over the '
             'launches of each run)', '',
             'SQ_*_CYCLES and SQ_WAIT_* / SQ_ACTIVE_* are summed over waves (quad-cycle units except '
             'SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES); GRBM_GUI_ACTIVE is summed over the 8 XCDs '
             '(measured 15.7 counts per ns of kernel time = 8 x 1.96 GHz).  MFMA pipe utilisation = '
             'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS).', '']
    for (group, k) in sorted(agg):
        c = {n: sum(v) / len(v) for n, v in agg[(group, k)].items()}
        extra = []
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c.get('GRBM_GUI_ACTIVE'):
            extra.append('MFMA util %.1f %%' % (100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] /
                                                (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024)))
        if 'TCC_HIT_sum' in c:
            extra.append('L2 hit %.1f %%' % (100 * c['TCC_HIT_sum'] /
                                             max(c['TCC_HIT_sum'] + c['TCC_MISS_sum'], 1)))
        if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
            extra.append('LDS conflict cycles %.1f %% of LDS-active'
                         % (100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']))
        if 'SQ_WAVE_CYCLES' in c:
            extra.append('wave time: active %.0f %% / wait-any %.0f %% / issue-stall %.0f %%' % (
                100 * c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'],
                100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'],
                100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']))
        lines.append('* **%s** `%s` (%d launches): %s' % (
            group, k, max(len(v) for v in agg[(group, k)].values()),
            '; '.join(extra) if extra else ''))
        lines.append('  ' + ', '.join('%s=%.4g' % kv for kv in sorted(c.items())))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
