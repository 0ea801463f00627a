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
SOURCES = ['capi.cpp', 'ctc.hip', 'beam.hip', 'frontend.hip', 'conv.hip', 'gemm.hip', 'lstm.hip', 'lstm_fwd.hip', 'lstm_bwd.hip', 'lstm_ln.hip',
           'optim.hip', 'random.hip', 'decode_host.cpp', 'comm.cpp', 'roles.cpp']
ARCH = 'gfx950'


STAMP = LIB + '.srchash'


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + [
        os.path.join(CSRC, 'common.h'),
        os.path.join(CSRC, 'lstm_common.h'),
        os.path.join(os.path.dirname(HERE), 'include', 'asr_hip.h')]


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    """Content-based (a snapshot copy may not preserve mtimes): the library is current
    iff the hash of every source it was built from is recorded next to it."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libasr_hip.so.  Safe to call from
    several processes at once (one rank per GPU): an exclusive file lock serialises
    them, the first one builds, the others find the library current."""
    force = force or os.environ.get('ASR_BUILD_FORCE') == '1'
    if not force and not _stale():
        print('asr_study_amd.build: libasr_hip.so is current (source hash %s...): nothing to compile; '
              'ASR_BUILD_FORCE=1 recompiles all %d sources' % (_source_hash()[:12], len(SOURCES)),
              file=sys.stderr)
        return LIB
    import fcntl
    with open(LIB + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale():
                _compile(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


def _compile(verbose):
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
    tmp = LIB + '.tmp.%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs + ['-lpthread', '-ldl']
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)                     # atomic: a loader never sees a partial file
    print('asr_study_amd.build: compiled %d sources for %s with %s -> %s'
          % (len(SOURCES), ARCH, hipcc, LIB), file=sys.stderr)
    with open(STAMP, 'w') as f:
        f.write(_source_hash() + '\n')


def toolchain_probe(workdir=None):
    """Compile TWO small sources of the library (capi.cpp + random.hip, a few seconds) with THIS
    machine's hipcc into a scratch directory and link them into libasr_probe.so; returns its
    path.  The shipped libasr_hip.so is normally reused as built (its source hash matches), so
    a GPU box never runs hipcc on its own: __graft_entry__.smoke() and one -m gpu test load this
    probe next to the shipped library and compare their Philox streams bit for bit -- a
    toolchain / runtime skew on the box shows up as a red test, not as a silently reused
    binary."""
    import tempfile
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    workdir = workdir or tempfile.mkdtemp(prefix='asr_probe_')
    objs = []
    for src in ('capi.cpp', 'random.hip'):
        obj = os.path.join(workdir, os.path.splitext(src)[0] + '.o')
        subprocess.check_call([hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
                               '-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj])
        objs.append(obj)
    lib = os.path.join(workdir, 'libasr_probe.so')
    subprocess.check_call([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', lib] + objs +
                          ['-lpthread', '-ldl'])
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
