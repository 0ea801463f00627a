#!/usr/bin/env python
"""Scans the gfx950 ISA of the sources that use 16-byte BUFFER stores for the store-data hazard
hipcc does not pad (round 6): `buffer_store_dwordx4 v[a:b], .., sN offen` -- a
register in the soffset field -- followed within two instructions, without an s_nop between, by
a VALU instruction that writes one of v[a:b].  The store reads its data registers over several
cycles; measured on gfx950 as the second dword of lanes 12..15 of each row going out stale in
about one wave per launch.  (For stores WITHOUT a register soffset, and for global / flat
stores, the compiler inserts the wait state itself.)

    python tools/store_hazard_scan.py [source.hip ...]      # default: every source with such stores
Prints one line per kernel and exits 1 if a risky pair is found."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'asr_study_amd', 'csrc')
STORE = re.compile(r'buffer_store_dwordx[34] v\[(\d+):(\d+)\], (?:v\d+|off), s\[\d+:\d+\], (s\d+|\S+)')
VALU = re.compile(r'(v_\w+)\s+(?:v\[(\d+):(\d+)\]|v(\d+)\b)')


def sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith('.hip'):
            text = open(os.path.join(CSRC, f)).read()
            if 'raw_buffer_store_b128' in text or 'xstore<' in text:
                out.append(f)
    return out


def isa(src):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-x', 'hip',
                               '--cuda-device-only', '-S', os.path.join(CSRC, src), '-o', out])
        return open(out).read()


def scan(text):
    """-> [(kernel, stores with a register soffset, [(store line, clobbering line)])]"""
    res = []
    parts = re.split(r'\n(?=\S+:\s+; @)', text)
    for part in parts:
        m = re.match(r'(\S+):\s+; @', part)
        if not m:
            continue
        lines = [l.split(';')[0].strip() for l in part.split('\n')]
        lines = [l for l in lines if l and not l.startswith('.')]
        n_reg, risky = 0, []
        for i, l in enumerate(lines):
            mm = STORE.search(l)
            if not mm or not mm.group(3).startswith('s'):
                continue
            n_reg += 1
            lo, hi = int(mm.group(1)), int(mm.group(2))
            for k in (1, 2):
                if i + k >= len(lines):
                    break
                nl = lines[i + k]
                if nl.startswith('s_nop') or nl.startswith('s_waitcnt'):
                    break
                w = VALU.match(nl)
                if w and not w.group(1).startswith('v_cmp'):
                    a, b = (int(w.group(2)), int(w.group(3))) if w.group(2) else (int(w.group(4)),) * 2
                    if not (b < lo or a > hi):
                        risky.append((l, nl))
                        break
        if n_reg:
            res.append((m.group(1), n_reg, risky))
    return res


def main():
    srcs = sys.argv[1:] or sources()
    bad = 0
    for src in srcs:
        for kern, n, risky in scan(isa(src)):
            print('%-14s %-60s %3d stores with a register soffset, %d risky' % (src, kern[:60], n, len(risky)))
            for st, cl in risky:
                print('    %s\n      -> %s' % (st, cl))
            bad += len(risky)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
