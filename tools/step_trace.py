"""Per-kernel totals, launch counts and idle gaps of ONE steady-state step out of a
`rocprofv3 --kernel-trace --output-format csv` trace of bench.py (steps are delimited by the
optimiser kernel).  usage: python tools/step_trace.py <..._kernel_trace.csv> [top]"""
import csv
import re
import sys


def short(name):
    name = name.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', name)
    if m:
        n = int(m.group(1))
        name = name[len(m.group(0)):][:n]
    return re.sub(r'\(.*', '', name)[:72]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name'] or 'sgd_kernel' in r['Kernel_Name']]
    a, b = marks[-2], marks[-1]
    t0 = int(rows[a]['End_Timestamp'])
    tot, cnt = {}, {}
    prev_end, gaps, small = t0, 0, 0
    for r in rows[a + 1:b + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        n = short(r['Kernel_Name'])
        tot[n] = tot.get(n, 0) + (e - s)
        cnt[n] = cnt.get(n, 0) + 1
        if s > prev_end:
            gaps += s - prev_end
        prev_end = max(prev_end, e)
    span = int(rows[b]['End_Timestamp']) - t0
    print('step span %.3f ms, %d launches, kernel time %.3f ms, idle gaps %.3f ms'
          % (span / 1e6, b - a, sum(tot.values()) / 1e6, gaps / 1e6))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]:
        print('%-72s %4d %9.3f ms' % (k, cnt[k], v / 1e6))


if __name__ == '__main__':
    main()
