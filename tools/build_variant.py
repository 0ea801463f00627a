#!/usr/bin/env python
"""Measurement builds: recompile ONE source of the library with extra -D flags and link it with
the other objects of the current build into variants/libasr_<name>.so (git-ignored, travels to
the GPU box); run it with ASR_LIB_PATH=variants/libasr_<name>.so for A/B runs on one box.

    python tools/build_variant.py <name> <source> -DFLAG[=V] [-DFLAG2 ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_study_amd import build as B  # noqa: E402


def main():
    name, src = sys.argv[1], sys.argv[2]
    flags = sys.argv[3:]
    B.build()
    out_dir = os.path.join(ROOT, 'variants')
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    obj = os.path.join(out_dir, '%s_%s.o' % (os.path.splitext(src)[0], name))
    subprocess.check_call([hipcc, '--offload-arch=' + B.ARCH, '-O3', '-std=c++17', '-fPIC', '-x', 'hip',
                           '-c', os.path.join(B.CSRC, src), '-o', obj] + flags)
    objs = [obj if s == src else os.path.join(B.CSRC, os.path.splitext(s)[0] + '.o') for s in B.SOURCES]
    lib = os.path.join(out_dir, 'libasr_%s.so' % name)
    subprocess.check_call([hipcc, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', lib] + objs +
                          ['-lpthread', '-ldl'])
    print(lib)


if __name__ == '__main__':
    main()
