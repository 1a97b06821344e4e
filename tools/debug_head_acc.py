import sys, torch, importlib
sys.path.insert(0, '/root/repo')
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = torch.device('cuda:0')
torch.manual_seed(11)
B, d = 9, 32
lens = torch.randint(1, 6, (B,))
seg = torch.zeros(B + 1, dtype=torch.int32); seg[1:] = lens.cumsum(0)
NT = int(seg[-1])
seg_d = seg.to(dev)
allf0 = torch.randn(NT, d, device=dev) * 0.3
P = [torch.randn(B, d, device=dev) * 0.3, torch.randn(d, d, device=dev) * 0.2, torch.randn(d, device=dev) * 0.2,
     torch.randn(d, d, device=dev) * 0.2, torch.randn(1, d, device=dev) * 0.2, torch.randn(d, 2 * d, device=dev) * 0.2]
w = torch.randn(B, d, device=dev)

def run(mode):
    dt = torch.float64 if mode == 'f64' else torch.float32
    allf = allf0.to(dt).clone().requires_grad_()
    per = [t.to(dt).clone().requires_grad_() for t in P]
    v, Wu, bu, Wv, we, Wsr = per
    if mode == 'fused':
        s = ops.readout_head(allf, seg_d, None, None, [per])[0]
    elif mode == 'unfused':
        U = ops.linear(allf, Wu, bu, None, exact=True); Vq = ops.linear(v, Wv, None, None, exact=True)
        srg = ops.seg_attn(U, Vq, we, allf, seg_d, None)
        s = ops.linear(ops.cat_cols(v, srg), Wsr, None, None, exact=True)
    else:
        U = allf @ Wu.t() + bu; Vq = v @ Wv.t()
        sid = torch.repeat_interleave(torch.arange(B, device=dev), lens.to(dev))
        e = (torch.sigmoid(U + Vq[sid]) * we).sum(-1)
        srg = torch.zeros(B, d, device=dev, dtype=dt)
        for b in range(B):
            a = torch.softmax(e[seg[b]:seg[b + 1]], 0)
            srg[b] = (a[:, None] * allf[seg[b]:seg[b + 1]]).sum(0)
        s = torch.cat([v, srg], 1) @ Wsr.t()
    s = s / s.norm(dim=1, keepdim=True)
    loss = (s * w.to(dt)).sum()
    gr = torch.autograd.grad(loss, [allf] + per)
    return [s.detach().double()] + [g.double() for g in gr]
r64 = run('f64')
for mode in ('fused', 'unfused', 'torch32'):
    r = run(mode)
    print(mode, ['%.1e' % (a - b).abs().max().item() for a, b in zip(r, r64)])
