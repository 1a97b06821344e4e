"""Builds libsrec_hip.so (gfx950 HIP kernels behind the C ABI of include/srec.h) in-tree.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the
gpurun snapshot.  Usage: python -m ... or `python sessionrec-pytorch_amd/build.py`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsrec_hip.so')
COLLATE_LIB = os.path.join(HERE, 'libsrec_collate.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def _newer(srcs, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=True):
    hip_srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    deps = hip_srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    if force or _newer(deps, LIB):
        objs = []
        for s in hip_srcs:
            o = s[:-4] + '.o'
            if force or _newer([s] + [d for d in deps if d.endswith('.h')], o):
                cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o]
                if verbose:
                    print(' '.join(cmd), flush=True)
                subprocess.check_call(cmd)
            objs.append(o)
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    cpp = os.path.join(CSRC, 'collate.cpp')
    if os.path.exists(cpp) and (force or _newer([cpp], COLLATE_LIB)):
        cmd = ['g++', '-O3', '-std=c++17', '-fPIC', '-shared', cpp, '-o', COLLATE_LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
