"""Development probe for hg_bwd_src (the LAST launch that writes the probe buffers; csrc/hgat.hip, -DSREC_HG_TIMING): wall-clock life of every workgroup and the phase clocks of
one, for one eager MSGIFSR forward on a bench batch.  usage (GPU box): python tools/hg_timing.py"""
import ctypes, glob, importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('hgat.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_HG_TIMING',
                       '-c', pk + '/csrc/hgat.hip', '-o', '/tmp/hgat_tim.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_hgtim.so',
                       '/tmp/hgat_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_hgtim.so'
import torch
import bench
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
sp = importlib.import_module('sessionrec-pytorch_amd')
ops.set_precision('bf16')
dev = torch.device('cuda:0')
batches, _ = bench.make_batches('MSGIFSR', 3, 2, 512, 37484, 20, 123, padded=True)
torch.manual_seed(123)
model = bench.build_model(sp, 'MSGIFSR', 37484, 256, 3, 0.1).to(dev)
model.train()
inp, lab = batches[0]
inp = [x.to(dev) for x in inp]; lab = lab.to(dev)
for _ in range(3):
    model.zero_grad()
    model.fused_loss(*inp, lab).backward()
torch.cuda.synchronize()
dll = L.lib.load()
tim, blk = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 32768)()
assert dll.srec_hg_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(16384, 2)
live = b[:, 1] > b[:, 0]
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('%d workgroups, span %.1f us, life: mean %.2f median %.2f p90 %.2f max %.2f us' % (live.sum(), en.max(), life.mean(), np.median(life), np.percentile(life, 90), life.max()))
ts = np.linspace(0, en.max(), 10)
print('alive at t:', ' '.join('%.0fus:%d' % (t, int(((st <= t) & (en > t)).sum())) for t in ts))
print('hg_bwd_src, workgroup 100, wave 0 (cycles from start, every phase drained): metadata %d, edge loop %d, der loads %d, barrier %d, dense term + stores %d' % (
    tim[1], tim[2], tim[3], tim[4], tim[5]))
short = life < 1.5
print('workgroups with life < 1.5 us (capacity padding): %d' % short.sum())
