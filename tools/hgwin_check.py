"""MSHGNN layer at the C3 shape: the row-window kernels (csrc/hgwin.hip) against the batched formulation (csrc/hgat.hip) on a
bench batch - outputs / arg-max agreement / saved soft-max values, gradients once the window backward exists, and event-timed
forward / forward + backward of the layer alone.  usage (GPU box): python tools/hgwin_check.py [--dim 256] [--drop 0.1]"""
import argparse, importlib, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--dim', type=int, default=256)
ap.add_argument('--drop', type=float, default=0.1)
ap.add_argument('--order', type=int, default=3)
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--modes', default='off,fwd')
ap.add_argument('--slow', type=int, default=0)
args = ap.parse_args()

ops = importlib.import_module('sessionrec-pytorch_amd.ops')
sp = importlib.import_module('sessionrec-pytorch_amd')
ops.set_precision('bf16')
dev = torch.device('cuda:0')
d = args.dim
batches, _ = bench.make_batches('MSGIFSR', args.order, 2, 512, 37484, 20, 123, padded=True)
torch.manual_seed(123)
model = bench.build_model(sp, 'MSGIFSR', 37484, d, args.order, args.drop).to(dev)
model.train()
(mg,), lab = batches[0]
mg = mg.to(dev)
layer = model.layers[0]
K = args.order
ncap = mg.meta['ncap']
NT = sum(ncap[k] for k in range(1, K + 1))
g = torch.Generator(device='cpu').manual_seed(7)
x0 = torch.nn.functional.normalize(torch.randn(NT, d, generator=g), dim=1)
o = 0
for k in range(1, K + 1):
    x0[o + mg.count('N%d' % k):o + ncap[k]] = 0
    o += ncap[k]
x0 = x0.to(dev)
R = torch.randn(NT, d, generator=g).to(dev)


def run(mode, grad=True):
    ops.HG_WIN['mode'] = mode
    ops.HG_WIN['force_slow'] = args.slow
    torch.manual_seed(5)
    ops.seed_dropout()
    x = x0.clone().requires_grad_(grad)
    for p in layer.parameters():
        p.grad = None
    with torch.set_grad_enabled(grad):
        out = layer.forward_stacked(mg, x)
        if grad:
            (out * R).sum().backward()
    ops.flush_deferred()
    gs = {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None} if grad else {}
    return out.detach().clone(), (x.grad.clone() if grad else None), gs


def timeit(mode, grad):
    ops.HG_WIN['mode'] = mode
    ops.HG_WIN['force_slow'] = args.slow
    for _ in range(3):
        run(mode, grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x = x0.clone().requires_grad_(grad)
    e0.record()
    for _ in range(args.iters):
        with torch.set_grad_enabled(grad):
            out = layer.forward_stacked(mg, x)
            if grad:
                out.backward(R)
    ops.flush_deferred()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


modes = args.modes.split(',')
ref = run(modes[0])
print('reference mode %s: |out| rms %.4f' % (modes[0], ref[0].pow(2).mean().sqrt().item()))
for mode in modes[1:]:
    out, gx, gs = run(mode)
    diff = (out - ref[0]).abs()
    print('mode %s vs %s: out max abs diff %.3e, rms diff %.3e (rms %.3e)' % (
        mode, modes[0], diff.max().item(), diff.pow(2).mean().sqrt().item(), ref[0].pow(2).mean().sqrt().item()))
    if gx is not None and ref[1] is not None:
        print('   dx: rel %.3e' % ((gx - ref[1]).norm() / ref[1].norm()).item())
        for n in sorted(gs):
            if n in ref[2]:
                print('   d %-28s rel %.3e' % (n, ((gs[n] - ref[2][n]).norm() / ref[2][n].norm().clamp(min=1e-12)).item()))
for mode in modes:
    tf = timeit(mode, False)
    tb = timeit(mode, True)
    print('mode %-4s: forward (eager, no grad) %.1f us, forward + backward %.1f us per layer call' % (mode, tf, tb))
