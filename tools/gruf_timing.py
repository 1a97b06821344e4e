"""Development probe for the fused k-gram GRU forward (csrc/gruf.hip, -DSREC_GRUF_TIMING): wall-clock life of every workgroup
and the phase clocks of one order-3 workgroup, at the bench's node counts.  usage (GPU box): python tools/gruf_timing.py"""
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
import torch
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
ops.set_precision('bf16')
dev = torch.device('cuda:0')
d, ks, ns = 256, [2, 3], [1900, 1615]
torch.manual_seed(0)
grus = [torch.nn.GRU(d, d, 1, True, True).to(dev) for _ in ks]
DN, DR = [None, None], [None, None]
if os.environ.get('PAD'):             # the step's shapes: capacity 2560 nodes per order, live counts on the device
    live_n = ns
    ns = [2560, 2560]
    DN = [torch.tensor([n], device=dev, dtype=torch.int32) for n in live_n]
    DR = [torch.tensor([n * k], device=dev, dtype=torch.int32) for n, k in zip(live_n, ks)]
xs = [torch.randn(n * k, d, device=dev) * 0.1 for n, k in zip(ns, ks)]
for _ in range(3):
    with torch.no_grad():
        ops.gru_expand_all(xs, grus, ks, DN, DR)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    with torch.no_grad():
        ops.gru_expand_all(xs, grus, ks, DN, DR)
e1.record()
torch.cuda.synchronize()
print('forward (weights_bf16 + wfrag + fused): %.1f us per call' % (e0.elapsed_time(e1) / 20 * 1e3))
dll = L.lib.load()
tim, blk = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 2048)()
assert dll.srec_gruf_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(1024, 2)
live = b[:, 1] > b[:, 0]
if not live.any():           # (gru_fused_fwd16, the kernel the bench shapes take, carries no probes)
    b[:, 1] = b[:, 0] + 1
    live = b[:, 1] > b[:, 0]
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('%d workgroups, span %.1f us, life: mean %.2f median %.2f max %.2f us' % (live.sum(), en.max(), life.mean(), np.median(life), life.max()))
nb2 = (ns[0] + 15) // 16
print('order 2 lives: mean %.1f us; order 3 lives: mean %.1f us' % (life[:nb2].mean(), life[nb2:live.sum()].mean()))
print('order-3 workgroup, wave 0, cycles summed over the 3 time steps: stage x %d, barrier %d, k-loop %d, gates+stores %d, '
      'h tile %d, drain %d' % tuple(tim[i] for i in range(6)))

# ---- backward
xs = [x.requires_grad_() for x in xs]
gout = [torch.randn(n, d, device=dev) for n in ns]
for _ in range(3):
    outs = ops.gru_expand_all(xs, grus, ks, DN, DR)
    torch.autograd.backward(list(outs), gout)
torch.cuda.synchronize()
if os.environ.get('COLD'):            # the step's situation: everything this kernel reads was written ~0.5 GB of traffic ago
    outs = ops.gru_expand_all(xs, grus, ks, DN, DR)
    junk = torch.empty(256 << 20, device=dev, dtype=torch.float32).fill_(1.0)
    torch.cuda.synchronize()
    torch.autograd.backward(list(outs), gout)
    torch.cuda.synchronize()
assert dll.srec_grub_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(1024, 2)
live = b[:, 1] > b[:, 0]
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('backward: %d workgroups, span %.1f us, life: mean %.2f median %.2f max %.2f us' % (live.sum(), en.max(), life.mean(), np.median(life), life.max()))
print('order 2 lives: mean %.1f us; order 3 lives: mean %.1f us' % (life[:nb2].mean(), life[nb2:live.sum()].mean()))
print('order-3 workgroup, wave 0, cycles summed over the 3 time steps (gru_fused_bwd16): gate derivatives %d, barrier %d, d(gh) W_hh %d, '
      'd h store %d, barrier %d, d x = d(gi) W_ih (all steps) %d, d x store + bias sums %d' % tuple(tim[i] for i in range(7)))
print('that workgroup: %d shader clocks over %.2f us -> %.2f GHz' % (tim[8], tim[9] * 0.01, tim[8] / max(tim[9], 1) / 10.0))
