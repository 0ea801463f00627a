"""Build libasr_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

The shared object is a plain C-ABI library (include/asr_hip.h): it links against
the HIP runtime only -- no torch, no Python.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libasr_hip.so')
SOURCES = ['capi.cpp', 'ctc.hip', 'frontend.hip', 'gemm.hip', 'lstm.hip',
           'optim.hip', 'decode_host.cpp']
ARCH = 'gfx950'


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, 'common.h'),
        os.path.join(os.path.dirname(HERE), 'include', 'asr_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libasr_hip.so."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + '.o')
        cmd = [hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
               '-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE,
                                            stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, out))
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs + ['-lpthread']
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
