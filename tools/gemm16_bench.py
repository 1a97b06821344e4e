"""times the bf16-in-HBM GEMMs at the bench's GAT shapes (12 problems: 3 node types x (intra, inter) x 2 convs), per kernel
python tools/gemm16_bench.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
cap, lives, D, HD = 2048, (1400, 1300, 1150), 256, 2048
bf = lambda *s: (torch.randn(*s, device=dev) * 0.3).bfloat16()


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


dyns = [torch.tensor([n], device=dev, dtype=torch.int32) for n in lives]
x16 = [bf(cap, D) for _ in range(12)]
w16 = [bf(HD, D) for _ in range(12)]
wt16 = [bf(D, HD) for _ in range(12)]
P = [torch.empty(cap, HD, device=dev, dtype=torch.bfloat16) for _ in range(12)]
dP = [bf(cap, HD) for _ in range(12)]
tg = [torch.empty(cap, D, device=dev) for _ in range(12)]
gW = [torch.empty(HD, D, device=dev) for _ in range(12)]
live_rows = 4 * sum(lives)
res = {}
fwd = lambda: ops.gemm16('nt', [(cap, HD, D, [(x16[i], w16[i])], P[i], dyns[i % 3]) for i in range(12)], D, D, HD,
                         c16=True, keep_dead=True)
dgr = lambda: ops.gemm16('nt', [(cap, D, HD, [(dP[i], wt16[i])], tg[i], dyns[i % 3]) for i in range(12)], HD, HD, D)
wg = lambda: ops.gemm16('tn', [(HD, D, cap, [(dP[i], x16[i])], gW[i], dyns[i % 3]) for i in range(12)], HD, D, D)
print('fwd %.1f us, dgrad %.1f us, wgrad %.1f us' % (timed(fwd), timed(dgr), timed(wg)), flush=True)
fl = 2.0 * live_rows * D * HD
print('flop per product %.2f GF; at 1 PF/s = %.1f us; P bytes %.0f MB' % (fl / 1e9, fl / 1e15 * 1e6, live_rows * HD * 2 / 1e6))
