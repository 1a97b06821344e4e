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


def _digest(paths, extra=''):
    """content hash (mtimes do not survive the gpurun snapshot, and a stale .so segfaults)."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, digest):
    stamp = target + '.stamp'
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    return open(stamp).read().strip() != digest


def _mark(target, digest):
    with open(target + '.stamp', 'w') as f:
        f.write(digest)


def build(force=False, verbose=True):
    hip_srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    inc = os.path.join(os.path.dirname(HERE), 'include')
    hdrs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')]
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
    dig = _digest(hip_srcs + hdrs, ' '.join(flags))
    if force or _stale(LIB, dig):
        objs, todo = [], []
        for s in hip_srcs:
            o = s[:-4] + '.o'
            odig = _digest([s] + hdrs, ' '.join(flags))
            if force or _stale(o, odig):
                todo.append((s, o, odig))
            objs.append(o)

        def compile_one(job):
            s, o, odig = job
            cmd = [HIPCC] + flags + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            _mark(o, odig)
        if todo:                       # translation units are independent: compile them side by side
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
                list(ex.map(compile_one, todo))
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        _mark(LIB, dig)
    cpp = os.path.join(CSRC, 'collate.cpp')
    if os.path.exists(cpp):
        cdig = _digest([cpp], 'g++ -O3')
        if force or _stale(COLLATE_LIB, cdig):
            cmd = ['g++', '-O3', '-std=c++17', '-fPIC', '-shared', cpp, '-o', COLLATE_LIB]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            _mark(COLLATE_LIB, cdig)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
