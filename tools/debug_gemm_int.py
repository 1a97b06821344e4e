import sys, torch, importlib, itertools
sys.path.insert(0, '/root/repo')
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
bad = 0
ri = lambda *s: torch.randint(-4, 5, s, generator=g).float().to(dev)
for M, N, K in itertools.product((1, 7, 9, 40, 64, 65, 130, 300), (4, 32, 96, 256), (4, 12, 32, 36, 64, 100, 256, 260)):
    a, b, gy, bias = ri(M, K), ri(N, K), ri(M, N), ri(N)
    y = torch.empty(M, N, device=dev); ops.gemm_nt(a, b, y, bias)
    e1 = (y - (a @ b.t() + bias)).abs().max().item()
    c0 = ri(M, N); y = c0.clone(); ops.gemm_nt(a, b, y, None, beta=1.0)
    e1b = (y - (a.double() @ b.double().t() + c0.double()).float()).abs().max().item()
    gx = torch.empty(M, K, device=dev); ops.gemm_nn(gy, b, gx)
    e2 = (gx - (gy.double() @ b.double()).float()).abs().max().item()
    gw = torch.empty(N, K, device=dev); ops.gemm_tn(gy, a, gw)
    e3 = (gw - (gy.double().t() @ a.double()).float()).abs().max().item()
    if max(e1, e1b, e2, e3) > 0:
        bad += 1
        print('MISMATCH', (M, N, K), e1, e1b, e2, e3)
print('bad', bad)
