"""Development probe for the row-window MSHGNN kernels (csrc/hgwin.hip, -DSREC_HGWIN_TIMING): wall-clock life of every workgroup
and the phase clocks of wave 0, one eager layer forward on a bench batch.  usage (GPU box): python tools/hgwin_timing.py"""
import ctypes, glob, importlib, os, subprocess, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pk = os.path.join(root, 'sessionrec-pytorch_amd')
objs = [o for o in glob.glob(pk + '/csrc/*.o') if not o.endswith('hgwin.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DSREC_HGWIN_TIMING', '-DSREC_HGWIN_KO=%s' % os.environ.get('KO', '0'),
                       '-c', pk + '/csrc/hgwin.hip', '-o', '/tmp/hgwin_tim.o'])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/libsrec_hgwintim.so',
                       '/tmp/hgwin_tim.o'] + objs)
L = importlib.import_module('sessionrec-pytorch_amd._lib')
L.LIB_PATH = '/tmp/libsrec_hgwintim.so'
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
(mg,), lab = batches[0]
mg = mg.to(dev)
layer = model.layers[0]
NT = sum(mg.meta['ncap'][k] for k in (1, 2, 3))
x = torch.nn.functional.normalize(torch.randn(NT, 256, device=dev), dim=1)
ops.HG_WIN['mode'] = 'fwd'
ops.HG_WIN['force_slow'] = int(sys.argv[1]) if len(sys.argv) > 1 else 0
with torch.no_grad():
    for _ in range(3):
        layer.forward_stacked(mg, x)
torch.cuda.synchronize()
dll = L.lib.load()
tim, blk = (ctypes.c_ulonglong * (2048 * 8))(), (ctypes.c_ulonglong * 4096)()
assert dll.srec_hgwin_timing(tim, blk) == 0
b = np.array(list(blk), dtype=np.int64).reshape(2048, 2)
tm = np.array(list(tim), dtype=np.int64).reshape(2048, 8)
live = b[:, 1] > b[:, 0]
t0 = b[live, 0].min()
st, en = (b[live, 0] - t0) * 0.01, (b[live, 1] - t0) * 0.01
life = en - st
print('%d live workgroups, span %.1f us, life: mean %.2f median %.2f p90 %.2f max %.2f us' % (live.sum(), en.max(), life.mean(), np.median(life), np.percentile(life, 90), life.max()))
names = ['degrees + scan', 'edge list + soft-max', 'phase A (all passes)', 'barrier after A', 'phase B (all passes)', 'barrier after B', 'epilogue']
m = tm[live].mean(0)
for n, v in zip(names, m):
    print('  %-24s %9.0f cycles  (%.1f us at 2.4 GHz)' % (n, v, v / 2400.))
print('  sum %.0f cycles = %.1f us' % (m[:7].sum(), m[:7].sum() / 2400.))
