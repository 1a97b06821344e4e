"""Development probe for csrc/gemm16.hip (-DSREC_G16_TIMING): phase clocks of wave 0 of one workgroup and the wall-clock life
of every workgroup, for the forward (bf16 out) and backward-data launches at the step's GAT shapes.
usage (GPU box): python tools/g16_timing.py [rows per order, default 2560]"""
import ctypes, glob, importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('gemm16.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_G16_TIMING',
                       '-c', pk + '/csrc/gemm16.hip', '-o', '/tmp/gemm16_tim.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_g16tim.so',
                       '/tmp/gemm16_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_g16tim.so'
import torch
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = torch.device('cuda:0')
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
lives, D, HD = (int(cap * 0.95), int(cap * 0.9), int(cap * 0.85)), 256, 2048
bf = lambda *s: (torch.randn(*s, device=dev) * 0.3).bfloat16()
dyns = [torch.tensor([n], device=dev, dtype=torch.int32) for n in lives]
x16, w16, wt16 = [bf(cap, D) for _ in range(12)], [bf(HD, D) for _ in range(12)], [bf(D, HD) for _ in range(12)]
P = [torch.empty(cap, HD, device=dev, dtype=torch.bfloat16) for _ in range(12)]
dP = [bf(cap, HD) for _ in range(12)]
tg = [torch.empty(cap, D, device=dev) for _ in range(12)]
dll = L.lib.load()
tim, blk = (ctypes.c_ulonglong * 8)(), (ctypes.c_ulonglong * 16384)()
for name, fn in (('forward (bf16 out)', lambda: ops.gemm16('nt', [(cap, HD, D, [(x16[i], w16[i])], P[i], dyns[i % 3]) for i in range(12)], D, D, HD, c16=True, keep_dead=True)),
                 ('backward-data', lambda: ops.gemm16('nt', [(cap, D, HD, [(dP[i], wt16[i])], tg[i], dyns[i % 3]) for i in range(12)], HD, HD, D))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dll.srec_g16_timing_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert dll.srec_g16_timing(tim, blk) == 0
    b = np.array(list(blk), dtype=np.int64).reshape(8192, 2)
    live = b[:, 1] > b[:, 0]
    t0 = b[live, 0].min()
    st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
    print('%s: event %.1f us; %d workgroups, start 0..%.1f us, end %.1f..%.1f us, life mean %.1f max %.1f us' % (
        name, e0.elapsed_time(e1) * 1e3, live.sum(), st.max(), en.min(), en.max(), (en - st).mean(), (en - st).max()))
    print('    workgroup 17 wave 0 (cycles): prologue %d, loop: wait+barrier %d, DMA issue %d, reads+MFMA %d (k-steps %d), epilogue %d, total %d' % (
        tim[0], tim[1], tim[2], tim[3], tim[6], tim[4], tim[5]))
    # concurrency: how many workgroups are alive over time
    ts = np.linspace(0, en.max(), 12)
    print('    alive at t:', ' '.join('%.0fus:%d' % (t, int(((st <= t) & (en > t)).sum())) for t in ts))
