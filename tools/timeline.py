#!/usr/bin/env python
"""Prints the kernel timeline of the last complete bench step from a rocprofv3
--kernel-trace csv (ms since the previous step's Adam kernel; one line per kernel;
q = HSA queue, i.e. main vs side stream).  Usage: timeline.py <bench_kernel_trace.csv> [min_us]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    a, b = idx[-2], idx[-1]
    t0 = int(rows[a]['End_Timestamp'])
    for r in rows[a + 1:b + 1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if (e - s) / 1e3 >= min_us:
            print('%8.3f %8.3f %8.1f q%s %s' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3,
                                               r['Queue_Id'], short(r['Kernel_Name'])))


if __name__ == '__main__':
    main()
