"""forward projection GEMM at the bench's GAT shapes: weights-in-registers kernel (variant 64) against the tiled kernel (0)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(0)
D, HD = 256, 2048
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


for name, cap, lives in (('step caps (2560, ~93 % live)', 2560, (2371, 2300, 2200)), ('loose caps', 2048, (1400, 1300, 1150))):
    # the layer's forward: 3 intra modules (one type each) + 1 inter module over all 3 types, for 2 convs
    dyns = [torch.tensor([n], device=dev, dtype=torch.int32) for n in lives]
    probs = []
    for conv in range(2):
        for t in range(3):
            probs.append((cap, HD, D, [(bf(cap, D), bf(HD, D))], torch.empty(cap, HD, device=dev, dtype=torch.bfloat16), dyns[t], 0, 1, None, lives[t]))
        w = bf(HD, D)
        for t in range(3):
            probs.append((cap, HD, D, [(bf(cap, D), w)], torch.empty(cap, HD, device=dev, dtype=torch.bfloat16), dyns[t], 0, 1, None, lives[t]))
    for variant in (0, 64, 0, 64):
        t = timed(lambda: ops.gemm16('nt', probs, D, D, HD, c16=True, keep_dead=True, variant=variant))
        rows = 4 * sum(lives)
        print('%s: variant %2d %.1f us  (%.0f MB out, %.2f TB/s)' % (name, variant, t, rows * HD * 2 / 1e6, rows * HD * 2 / t / 1e6))
