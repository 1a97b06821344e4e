"""Development probe: the fused GRU backward's workgroup lives and phase clocks INSIDE the replayed C3 step (the probes of
tools/gruf_timing.py read after `bench.py --step-only`).  usage (GPU box): python tools/gruf_in_step.py"""
import ctypes, glob, importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not (o.endswith('gruf.o') or o.endswith('grufb.o'))]
for nm in ('gruf', 'grufb'):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_GRUF_TIMING',
                           '-c', pk + '/csrc/%s.hip' % nm, '-o', '/tmp/%s_tim.o' % nm])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_gruftim.so',
                       '/tmp/gruf_tim.o', '/tmp/grufb_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_gruftim.so'
sys.argv = ['bench.py', '--step-only', '--steps', '50', '--warmup', '10']
import bench
bench.main()
dll = L.lib.load()
tim, blk = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 2048)()
assert dll.srec_grub_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(1024, 2)
live = b[:, 1] > b[:, 0]
live &= b[:, 0] > b[live, 0].max() - 20000           # the last replay only (other batches leave entries of other workgroups: 200 us)
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('backward inside the step: %d workgroups, span %.1f us, starts within %.1f us, life: mean %.2f median %.2f max %.2f us'
      % (live.sum(), en.max(), st.max(), life.mean(), np.median(life), life.max()))
print('phase cycles (wave 0 of one order-3 workgroup): gate derivatives %d, barrier %d, d(gh) W_hh %d, d h store %d, barrier %d, '
      'd x product %d, d x store + bias sums %d' % tuple(tim[i] for i in range(7)))
print('that workgroup: %d shader clocks over %.2f us -> %.2f GHz' % (tim[8], tim[9] * 0.01, tim[8] / max(tim[9], 1) / 10.0))
