"""Development probe for the fused read-out head forward (csrc/headf.hip, -DSREC_HEADF_TIMING): wall-clock life of every
workgroup and the phase clocks of wave 0, at the bench batch's shape.  usage (GPU box): python tools/headf_timing.py"""
import ctypes, glob, importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('headf.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_HEADF_TIMING',
                       '-c', pk + '/csrc/headf.hip', '-o', '/tmp/headf_tim.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_headftim.so',
                       '/tmp/headf_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_headftim.so'
import torch
import bench
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
ops.set_precision('bf16')
dev = torch.device('cuda:0')
B, d = 512, 256
# session sizes of the bench batch: rows of the per-session concatenation of all three orders
_, smp = bench.make_batches('MSGIFSR', 3, 1, B, 37484, 20, 123)
col = importlib.import_module('sessionrec-pytorch_amd.collate')
(mg,), _ = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 3)(smp[0])
seg = mg.cat_seg.to(dev)
NT = int(seg[-1])
lens = (seg[1:] - seg[:-1]).cpu().numpy()
segc = seg.cpu().numpy()
win = np.array([lens[(segc[:-1] >= w) & (segc[:-1] < w + 64)].sum() for w in range(0, NT, 64)])
print('rows %d, per session mean %.1f max %d; rows owned per 64-row window: mean %.1f max %d' % (NT, lens.mean(), lens.max(), win.mean(), win.max()))
torch.manual_seed(0)
allf = torch.nn.functional.normalize(torch.randn(NT, d, device=dev), dim=1)
buf = torch.zeros(B, 2 * d, device=dev)
buf[:, :d] = torch.nn.functional.normalize(torch.randn(B, d, device=dev), dim=1)
v = buf[:, :d]
v._srec_cat_left = True
sc = 1 / 16.0
per = [(v, (torch.rand(d, d, device=dev) * 2 - 1) * sc, (torch.rand(d, device=dev) * 2 - 1) * sc, (torch.rand(d, d, device=dev) * 2 - 1) * sc,
        (torch.rand(1, d, device=dev) * 2 - 1) * sc, (torch.rand(d, 2 * d, device=dev) * 2 - 1) * sc)]
dT = torch.tensor([NT], device=dev, dtype=torch.int32)
dB = torch.tensor([B], device=dev, dtype=torch.int32)
assert ops.readout_head_fused_ok(allf, per)
for _ in range(3):
    with torch.no_grad():
        ops.readout_head_fused(allf, seg, dT, dB, per)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    with torch.no_grad():
        ops.readout_head_fused(allf, seg, dT, dB, per)
e1.record()
torch.cuda.synchronize()
print('fused head forward (L2-warm, back to back): %.1f us per call' % (e0.elapsed_time(e1) / 20 * 1e3))
dll = L.lib.load()
tim, blk = (ctypes.c_ulonglong * 8192)(), (ctypes.c_ulonglong * 2048)()
assert dll.srec_headf_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(1024, 2)
t = np.array(list(tim), dtype=np.int64).reshape(1024, 8)
live = b[:, 1] > b[:, 0]
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('%d workgroups, span %.1f us, life: mean %.2f median %.2f max %.2f us' % (live.sum(), en.max(), life.mean(), np.median(life), life.max()))
names = ['prologue', 'Vq product', 'chunk staging+barriers', 'U products', 'e epilogues', 'soft-max + read-out', 'fc_sr product', 'normalise + stores']
tl = t[live]
order = np.argsort(-life)
print('phase cycles (wave 0): mean over workgroups | slowest workgroup (%d rows)' % win[np.flatnonzero(live)[order[0]]])
for i, nm in enumerate(names):
    print('  %-24s %8d | %8d' % (nm, tl[:, i].mean(), tl[order[0], i]))
print('  %-24s %8d | %8d' % ('sum', tl.sum(1).mean(), tl[order[0]].sum()))

# ---- backward (same probe arrays, overwritten by the backward kernel)
allf_g = allf.clone().requires_grad_()
ys = ops.readout_head_fused(allf_g, seg, dT, dB, per)
gy = torch.randn_like(ys[0])
for _ in range(3):
    torch.autograd.grad(ys, [allf_g], [gy], retain_graph=True)
torch.cuda.synchronize()
assert dll.srec_headf_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(1024, 2)
t = np.array(list(tim), dtype=np.int64).reshape(1024, 8)
live = b[:, 1] > b[:, 0]
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('backward: %d workgroups, span %.1f us, life: mean %.2f median %.2f max %.2f us' % (live.sum(), en.max(), life.mean(), np.median(life), life.max()))
names = ['g_s / staging', 'd cat product', 'd alpha, dX rows', 'd e', 'dU / dVq / dwp columns']
tl = t[live]
order = np.argsort(-life)
for i, nm in enumerate(names):
    print('  %-24s %8d | %8d' % (nm, tl[:, i].mean(), tl[order[0], i]))
print('  %-24s %8d | %8d' % ('sum', tl[:, :5].sum(1).mean(), tl[order[0], :5].sum()))
