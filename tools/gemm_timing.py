"""micro-benchmark of the GEMM kernels (HIP events on the current stream)"""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = torch.device('cuda:0')


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K) in [(3000, 2048, 256), (3000, 256, 2048), (3000, 256, 256), (512, 256, 512), (9000, 768, 256)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    y = torch.empty(M, N, device=dev)
    g = torch.randn(M, N, device=dev)
    gx = torch.empty(M, K, device=dev)
    gw = torch.empty(N, K, device=dev)
    res = {}
    for mode in ('fp32', 'bf16'):
        ops.set_precision(mode)
        res[mode] = (timeit(lambda: ops.gemm_nt(x, w, y)), timeit(lambda: ops.gemm_nn(g, w, gx)), timeit(lambda: ops.gemm_tn(g, x, gw)))
    fl = 2.0 * M * N * K
    print('M=%d N=%d K=%d  GFLOP %.2f' % (M, N, K, fl / 1e9))
    for mode in ('fp32', 'bf16'):
        t = res[mode]
        print('   %s  NT %.1f us (%.0f TF)  NN %.1f us (%.0f TF)  TN %.1f us (%.0f TF)' % (mode, t[0], fl / t[0] / 1e6, t[1], fl / t[1] / 1e6, t[2], fl / t[2] / 1e6))
ops.set_precision('fp32')
