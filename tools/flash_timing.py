"""Development probe: per-phase clocks (s_memtime) inside the bf16 flash-CE backward - wave 0 of one item-tile
workgroup and of one session-tile workgroup.  Builds a private copy of the library with -DSREC_FLASH_TIMING and runs
the merged backward launch of bench.py --kernel-only (B 512, V 37 484, d 256).
usage (GPU box): python tools/flash_timing.py"""
import ctypes, glob, importlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('score_ce_bf16.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_FLASH_TIMING',
                       '-I', root + '/include', '-c', pk + '/csrc/score_ce_bf16.hip', '-o', '/tmp/score_ce_bf16_tim.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_tim.so',
                       '/tmp/score_ce_bf16_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_tim.so'
import torch
import bench
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
sp = importlib.import_module('sessionrec-pytorch_amd')
ops.set_precision('bf16')
dev = torch.device('cuda:0')
B, V, d = 512, 37484, 256
torch.manual_seed(123)
model = bench.build_model(sp, 'SRGNN', V, d, 1).to(dev)
table = model._table().detach()[:V]
sr = torch.randn(B, d, device=dev) * 0.1
labels = torch.randint(0, V, (B,), device=dev, dtype=torch.int32)
ws = ops.CEWorkspace(B, V, d, dev)
lse, lossvec, loss = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty((), device=dev)
dE, dsr = torch.empty_like(table), torch.empty(B, d, device=dev)
tb = ops.TableBF16(table).refresh(table)
ops._ce_fwd(sr, table, None, labels, ws, None, tb, ws.lab_logit, lse, lossvec, loss)
dll = L.lib.load()
out = (ctypes.c_ulonglong * 16)()
names = ['wait+barrier', 'stage issue', 'product 1 (S)', 'exp / P', 'product 2 (acc)', 'prologue', 'epilogue', 'lifetime']
import numpy as np
blk = (ctypes.c_ulonglong * 2048)()
for parts in (3, 3, 3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops._ce_bwd(sr, table, None, labels, lse, None, None, None, ws, None, tb, dE, dsr, parts)
    e1.record()
    torch.cuda.synchronize()
    assert dll.srec_flash_timing(out) == 0
    print('parts', parts, 'events: %.1f us' % (e0.elapsed_time(e1) * 1e3))
    for r, role in enumerate(('item tile (dE)', 'session tile (d sr)')):
        print('   ', role, {n: out[r * 8 + i] for i, n in enumerate(names)})
    dll.srec_flash_blocks(blk)
    b = np.array(list(blk), dtype=np.int64).reshape(1024, 2)
    live = b[:, 1] > b[:, 0]
    t0 = b[live, 0].min()
    st, en = (b[:, 0] - t0) * 0.01, (b[:, 1] - t0) * 0.01      # us (100 MHz)
    nde = 296
    for nm, sl in (('item tiles', slice(0, nde)), ('session tiles', slice(nde, 1024))):
        m = live[sl]
        if m.any():
            print('    %-13s n %3d  start %.1f..%.1f us  end %.1f..%.1f (mean %.1f)  life mean %.1f max %.1f' % (
                nm, m.sum(), st[sl][m].min(), st[sl][m].max(), en[sl][m].min(), en[sl][m].max(), en[sl][m].mean(),
                (en[sl][m] - st[sl][m]).mean(), (en[sl][m] - st[sl][m]).max()))
