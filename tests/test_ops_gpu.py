"""GPU parity of every C-ABI kernel against a plain PyTorch fp32 restatement of the same op.
Tolerances (fp32 HIP vs fp32 torch): forward 1e-4 abs/rel, gradients 1e-4 rel."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    return importlib.import_module('sessionrec-pytorch_amd.ops')


def close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ref = b.abs().max().item() if b.numel() else 0.0
    # element-wise rtol plus an absolute floor tied to the tensor's scale (fp32 accumulation-order noise)
    atol = max(atol, 2e-6 * ref)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), '%s: max abs err %.3e (ref max %.3e)' % (what, err, ref)


@pytest.mark.parametrize('M,N,K', [(1, 4, 4), (37, 64, 32), (130, 96, 100), (512, 256, 512), (1391, 2048, 256), (3000, 32, 96)])
def test_gemm_nt_nn_tn(dev, M, N, K):
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    gy = torch.randn(M, N, generator=g).to(dev)
    y = torch.empty(M, N, device=dev)
    ops.gemm_nt(x, w, y, b)
    close(y, x @ w.t() + b, what='nt')
    ops.gemm_nt(x, w, y, None, beta=1.0)
    close(y, 2 * (x @ w.t()) + b, what='nt beta', atol=5e-5)
    gx = torch.empty(M, K, device=dev)
    ops.gemm_nn(gy, w, gx)
    close(gx, gy @ w, what='nn', atol=5e-5)
    gw = torch.empty(N, K, device=dev)
    ops.gemm_tn(gy, x, gw)
    close(gw, gy.t() @ x, what='tn', atol=2e-4)


def test_gemm_dynamic_extent(dev):
    ops = _ops()
    M, N, K, live = 300, 64, 32, 170
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    dyn = torch.tensor([live], dtype=torch.int32, device=dev)
    y = torch.full((M, N), 7.0, device=dev)
    ops.gemm_nt(x, w, y, None, dyn, 1)
    close(y[:live], x[:live] @ w.t(), what='dyn rows')
    assert (y[live:] == 0).all()
    gy = torch.randn(M, N, device=dev)
    gw = torch.empty(N, K, device=dev)
    ops.gemm_tn(gy, x, gw, dyn)
    close(gw, gy[:live].t() @ x[:live], what='dyn K', atol=1e-4)


def test_linear_cat_autograd(dev):
    ops = _ops()
    torch.manual_seed(0)
    a = torch.randn(50, 32, device=dev, requires_grad=True)
    b = torch.randn(50, 64, device=dev, requires_grad=True)
    w = torch.randn(48, 96, device=dev, requires_grad=True)
    bias = torch.randn(48, device=dev, requires_grad=True)
    y = ops.linear_cat([a, b], w, bias)
    ref = torch.nn.functional.linear(torch.cat([a, b], 1), w, bias)
    close(y, ref, what='fwd')
    gy = torch.randn_like(ref)
    g1 = torch.autograd.grad(y, [a, b, w, bias], gy)
    g2 = torch.autograd.grad(ref, [a, b, w, bias], gy)
    for u, v, n in zip(g1, g2, 'a b w bias'.split()):
        close(u, v, what='grad ' + n, atol=1e-4)


def _ce_ref(sr, E, cs, labels):
    z = sr @ E.t()
    if cs is not None:
        z = z * cs.unsqueeze(0)
    logp = torch.log_softmax(z, dim=1)
    return torch.nn.functional.nll_loss(logp, labels), logp


@pytest.mark.parametrize('B,V,d,cosine', [(32, 3429, 32, False), (32, 3429, 32, True), (100, 1000, 96, False),
                                          (512, 5000, 256, True), (7, 70, 64, False), (64, 64, 100, True)])
def test_score_ce_fwd_bwd(dev, B, V, d, cosine):
    ops = _ops()
    torch.manual_seed(B + V)
    sr = (torch.randn(B, d, device=dev) * 0.3).requires_grad_()
    E = (torch.randn(V, d, device=dev) * 0.3).requires_grad_()
    labels = torch.randint(0, V, (B,), device=dev)
    labels[0], labels[-1] = 0, V - 1
    cs = None
    if cosine:
        cs = (12.0 / E.detach().norm(dim=1)).contiguous()
    ws = ops.CEWorkspace(B, V, d, dev)
    tg = ops.TableGrad(E.detach())
    loss, lse = ops.score_ce(sr, E.detach(), cs, labels.int(), ws, tg, None, 1.0 / 12.0)
    sr2, E2 = sr.detach().clone().requires_grad_(), E.detach().clone().requires_grad_()
    if cosine:
        z = 12.0 * (sr2 @ torch.nn.functional.normalize(E2, dim=1).t())
        ref = torch.nn.functional.cross_entropy(z, labels)
    else:
        ref, _ = _ce_ref(sr2, E2, None, labels)
    close(loss, ref, what='loss', rtol=1e-5, atol=1e-5)
    loss.backward()
    ref.backward()
    close(sr.grad, sr2.grad, what='dsr', rtol=1e-4, atol=1e-6)
    close(tg.buf, E2.grad, what='dE', rtol=1e-4, atol=1e-6)
    # materialised log-probabilities (forward() contract)
    logp = ops.score_logp(sr.detach(), E.detach(), cs, ws, 1.0 / 12.0)
    _, lref = _ce_ref(sr.detach(), E.detach(), cs, labels)
    close(logp, lref, what='logp', rtol=1e-4, atol=1e-4)


def test_score_logp_autograd(dev):
    ops = _ops()
    torch.manual_seed(3)
    B, V, d = 20, 333, 32
    sr = (torch.randn(B, d, device=dev) * 0.3).requires_grad_()
    E = (torch.randn(V, d, device=dev) * 0.3).requires_grad_()
    labels = torch.randint(0, V, (B,), device=dev)
    ws = ops.CEWorkspace(B, V, d, dev)
    logp = ops.score_logp(sr, E, None, ws)
    loss = torch.nn.functional.nll_loss(logp, labels)
    loss.backward()
    sr2, E2 = sr.detach().clone().requires_grad_(), E.detach().clone().requires_grad_()
    ref, _ = _ce_ref(sr2, E2, None, labels)
    ref.backward()
    close(sr.grad, sr2.grad, what='dsr')
    close(E.grad, E2.grad, what='dE')


def test_score_ce_dynamic_batch(dev):
    ops = _ops()
    torch.manual_seed(5)
    B, live, V, d = 128, 77, 900, 64
    sr = torch.randn(B, d, device=dev) * 0.3
    E = torch.randn(V, d, device=dev) * 0.3
    labels = torch.randint(0, V, (B,), device=dev)
    dyn = torch.tensor([live], dtype=torch.int32, device=dev)
    ws = ops.CEWorkspace(B, V, d, dev)
    tg = ops.TableGrad(E)
    srg = sr.clone().requires_grad_()
    loss, _ = ops.score_ce(srg, E, None, labels.int(), ws, tg, dyn)
    loss.backward()
    sr2, E2 = sr[:live].clone().requires_grad_(), E.clone().requires_grad_()
    ref, _ = _ce_ref(sr2, E2, None, labels[:live])
    ref.backward()
    close(loss, ref, what='loss')
    close(srg.grad[:live], sr2.grad, what='dsr')
    assert (srg.grad[live:] == 0).all()
    close(tg.buf, E2.grad, what='dE')


def test_gather_scatter_normalize(dev):
    ops = _ops()
    torch.manual_seed(1)
    V, d, n = 500, 64, 300
    W = torch.randn(V, d, device=dev, requires_grad=True)
    idx = torch.randint(0, 50, (n,), device=dev)
    items, inv = torch.unique(idx, return_inverse=True)
    pos = torch.argsort(idx, stable=True).int()
    cnt = torch.bincount(inv)
    ptr_ = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), cnt.cumsum(0)]).int()
    out = ops.embedding_lookup(W, idx.int(), (items.int(), ptr_, pos), None)
    close(out, W[idx], what='gather')
    g = torch.randn(n, d, device=dev)
    (gw,) = torch.autograd.grad(out, W, g)
    ref = torch.zeros(V, d, device=dev).index_add_(0, idx, g)
    close(gw, ref, what='scatter', atol=1e-5)
    x = torch.randn(n, d, device=dev, requires_grad=True)
    for mode in (0, 1):
        y = ops.normalize(x, mode)
        r = torch.nn.functional.normalize(x, dim=1)
        close(y, r, what='normalize')
        gy = torch.randn_like(r)
        close(torch.autograd.grad(y, x, gy)[0], torch.autograd.grad(r, x, gy)[0], what='normalize bwd', atol=1e-5)
    sel = torch.randperm(n, device=dev)[:40].int()
    y = ops.row_gather(x, sel)
    close(y, x[sel.long()], what='row_gather')
    gy = torch.randn(40, d, device=dev)
    close(torch.autograd.grad(y, x, gy)[0], torch.zeros_like(x).index_add_(0, sel.long(), gy), what='row_gather bwd')


@pytest.mark.parametrize('live,d', [(37, 96), (64, 32), (1, 256), (0, 64)])
def test_row_gather_ascending_backward_writes_every_row(dev, live, d):
    """the last-node pick (srgnn.py:140, lessr.py:177) with its one-launch backward (srec_expand_rows_sorted): ascending
    picks, capacity padding (-1 entries behind the live count, rows behind the last session), a device-side live count -
    every row of the gradient written (the buffer starts as NaN poison), equal to the scatter of the live rows"""
    ops = _ops()
    torch.manual_seed(live + d)
    cap_b, nrows = 64, 700
    lens = torch.randint(1, 12, (cap_b,))
    seg = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)])
    last = torch.stack([seg[b] + torch.randint(0, int(lens[b]), ()) for b in range(cap_b)])
    last[live:] = -1
    idx = last.to(torch.int32).to(dev)
    dyn = torch.tensor([live], dtype=torch.int32, device=dev)
    x = torch.randn(nrows, d, device=dev, requires_grad=True)
    y = ops.row_gather(x, idx, dyn, ascending=True)
    close(y[:live], x[last[:live].to(dev)], what='pick')
    assert live == cap_b or float(y[live:].abs().max()) == 0.0
    gy = torch.randn(cap_b, d, device=dev)
    poison = torch.full((nrows * d + 64,), float('nan'), device=dev)         # the allocator hands this block out again
    del poison
    (gx,) = torch.autograd.grad(y, x, gy)
    ref = torch.zeros(nrows, d, device=dev)
    if live:
        ref.index_add_(0, last[:live].to(dev), gy[:live])
    assert torch.equal(gx, ref)


def test_seg_attn(dev):
    ops = _ops()
    torch.manual_seed(2)
    lens = torch.tensor([1, 5, 3, 20, 2, 7, 1, 64, 9], device=dev)
    B, N, h, D = len(lens), int(lens.sum()), 48, 96
    seg = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), lens.cumsum(0)]).int()
    sid = torch.repeat_interleave(torch.arange(B, device=dev), lens)
    U = torch.randn(N, h, device=dev, requires_grad=True)
    Vq = torch.randn(B, h, device=dev, requires_grad=True)
    we = torch.randn(1, h, device=dev, requires_grad=True)
    X = torch.randn(N, D, device=dev, requires_grad=True)
    out = ops.seg_attn(U, Vq, we, X, seg)
    e = (torch.sigmoid(U + Vq[sid]) * we).sum(1)
    mx = torch.full((B,), -1e30, device=dev).index_reduce_(0, sid, e.detach(), 'amax')
    ex = torch.exp(e - mx[sid])
    alpha = ex / torch.zeros(B, device=dev).index_add_(0, sid, ex)[sid]
    ref = torch.zeros(B, D, device=dev).index_add_(0, sid, X * alpha.unsqueeze(1))
    close(out, ref, what='fwd')
    g = torch.randn_like(ref)
    g1 = torch.autograd.grad(out, [U, Vq, we, X], g)
    g2 = torch.autograd.grad(ref, [U, Vq, we, X], g)
    for a, b, n in zip(g1, g2, ['dU', 'dVq', 'dwe', 'dX']):
        close(a, b, what=n, atol=2e-5)
    H = torch.randn(N, D, device=dev, requires_grad=True)
    o = ops.seg_mean_add(H, X, seg, B)
    mean = torch.zeros(B, D, device=dev).index_add_(0, sid, X) / lens.unsqueeze(1)
    r = H + mean[sid]
    close(o, r, what='seg_mean_add')
    g = torch.randn_like(r)
    for a, b, n in zip(torch.autograd.grad(o, [H, X], g), torch.autograd.grad(r, [H, X], g), ['dH', 'dF']):
        close(a, b, what=n, atol=1e-5)


def test_fused_adam_matches_torch(dev):
    optim = importlib.import_module('sessionrec-pytorch_amd.optim')
    torch.manual_seed(4)
    shapes = [(301, 64), (64,), (5, 3), (1, 48)]
    p1 = [torch.randn(s, device=dev).requires_grad_() for s in shapes]
    p2 = [p.detach().clone().requires_grad_() for p in p1]
    groups = lambda ps: [{'params': ps[:2]}, {'params': ps[2:], 'weight_decay': 0}]
    o1 = optim.FusedAdam(groups(p1), lr=1e-2, weight_decay=1e-2)
    o2 = torch.optim.Adam(groups(p2), lr=1e-2, weight_decay=1e-2)
    for step in range(4):
        for a, b in zip(p1, p2):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        if step == 2:
            p1[1].grad, p2[1].grad = None, None
        o1.step()
        o2.step()
    for a, b in zip(p1, p2):
        close(a, b, what='adam', rtol=1e-5, atol=1e-6)


def test_fused_adam_multi_tensor_launch_shapes(dev):
    """srec_adam_multi: more tensors than one launch carries (48), sizes that are not multiples of 4 or of the 4096
    chunk, and a parameter whose storage is only 4-byte aligned (scalar path)."""
    optim = importlib.import_module('sessionrec-pytorch_amd.optim')
    torch.manual_seed(9)
    sizes = [1, 2, 3, 5, 7, 64, 255, 1023, 4095, 4096, 4097, 9001, 70001] + [13 + 3 * i for i in range(50)]
    base = torch.randn(1000 + 1, device=dev)
    p1 = [torch.randn(n, device=dev).requires_grad_() for n in sizes]
    p1.append(base[1:].detach().requires_grad_())            # data_ptr % 16 == 4
    assert p1[-1].data_ptr() % 16 == 4
    p2 = [p.detach().clone().requires_grad_() for p in p1]
    o1 = optim.FusedAdam(p1, lr=3e-3, weight_decay=1e-2)
    o2 = torch.optim.Adam(p2, lr=3e-3, weight_decay=1e-2)
    for step in range(3):
        for a, b in zip(p1, p2):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        o1.step()
        o2.step()
    for n, a, b in zip(sizes + [1000], p1, p2):
        close(a, b, what='adam multi n=%d' % n, rtol=1e-5, atol=1e-6)


def test_gru_step_and_gram_combine(dev):
    ops = _ops()
    torch.manual_seed(6)
    n, d, k = 37, 32, 3
    gru = torch.nn.GRU(d, d, 1, True, True).to(dev)
    x = torch.randn(n * k, d, device=dev, requires_grad=True)
    GI = ops.linear(x, gru.weight_ih_l0, gru.bias_ih_l0).view(n, k, 3 * d)
    h = None
    for t in range(k):
        if t == 0:
            h = ops.gru_step(GI[:, t, :], None, gru.bias_hh_l0, None)
        else:
            h = ops.gru_step(GI[:, t, :], ops.linear(h, gru.weight_hh_l0, gru.bias_hh_l0), None, h)
    out = ops.gram_combine(x.view(n, k, d), h, k)
    x2 = x.detach().clone().requires_grad_()
    ref = 0.5 * x2.view(n, k, d).mean(1) + 0.5 * gru(x2.view(n, k, d))[1].squeeze(0)
    close(out, ref, what='expander fwd')
    g = torch.randn_like(ref)
    ps = list(gru.parameters())
    g1 = torch.autograd.grad(out, [x] + ps, g)
    g2 = torch.autograd.grad(ref, [x2] + ps, g)
    for a, b, nm in zip(g1, g2, ['dx', 'w_ih', 'w_hh', 'b_ih', 'b_hh']):
        close(a, b, what='expander ' + nm, atol=2e-5)


def _csr(key, n):
    idx = torch.argsort(key, stable=True).int()
    ptr_ = torch.zeros(n + 1, dtype=torch.long, device=key.device)
    ptr_[1:] = torch.bincount(key, minlength=n).cumsum(0)
    return ptr_.int(), idx


def test_gat_relation_and_head_combine(dev):
    ops = _ops()
    torch.manual_seed(8)
    Ns, Nd, H, D, E = 23, 17, 8, 32, 60
    src = torch.randint(0, Ns, (E,), device=dev)
    dst = torch.randint(0, Nd - 3, (E,), device=dev)          # last 3 destinations have no in-edges
    in_ptr, in_idx = _csr(dst, Nd)
    out_ptr, out_idx = _csr(src, Ns)
    graph = (in_ptr, in_idx, out_ptr, out_idx, src.int(), dst.int())
    Fs = torch.randn(Ns, H * D, device=dev, requires_grad=True)
    Fd = torch.randn(Nd, H * D, device=dev, requires_grad=True)
    al = torch.randn(1, H, D, device=dev, requires_grad=True)
    ar = torch.randn(1, H, D, device=dev, requires_grad=True)
    rst = ops.gat_relation(Fs, Fd, al, ar, graph, H)

    def ref_fn(Fs, Fd, al, ar):
        fs, fd = Fs.view(Ns, H, D), Fd.view(Nd, H, D)
        el, er = (fs * al).sum(-1), (fd * ar).sum(-1)
        e = torch.nn.functional.leaky_relu(el[src] + er[dst], 0.2)
        mx = torch.full((Nd, H), -1e30, device=dev).index_reduce_(0, dst, e.detach(), 'amax')
        ex = torch.exp(e - mx[dst])
        a = ex / torch.zeros(Nd, H, device=dev).index_add_(0, dst, ex)[dst]
        return torch.zeros(Nd, H, D, device=dev).index_add_(0, dst, fs[src] * a.unsqueeze(-1)).view(Nd, H * D)
    ref = ref_fn(Fs, Fd, al, ar)
    close(rst, ref, what='gat fwd')
    g = torch.randn_like(ref)
    g1 = torch.autograd.grad(rst, [Fs, Fd, al, ar], g)
    g2 = torch.autograd.grad(ref, [Fs, Fd, al, ar], g)
    for a, b, nm in zip(g1, g2, ['dFs', 'dFd', 'dal', 'dar']):
        close(a, b, what='gat ' + nm, atol=5e-5)
    # head combine
    x = torch.randn(Nd, D, device=dev, requires_grad=True)
    bias = torch.randn(H * D, device=dev, requires_grad=True)
    R1 = torch.randn(Nd, H * D, device=dev, requires_grad=True)
    R2 = torch.randn(Nd, H * D, device=dev, requires_grad=True)
    out = ops.head_combine(x, bias, 2.0, H, [R1, R2])
    r = ((R1 + R2 + bias).view(Nd, H, D) + 2.0 * x.unsqueeze(1)).max(1)[0]
    close(out, r, what='head combine')
    g = torch.randn_like(r)
    for a, b, nm in zip(torch.autograd.grad(out, [x, bias, R1, R2], g), torch.autograd.grad(r, [x, bias, R1, R2], g),
                        ['dx', 'dbias', 'dR1', 'dR2']):
        close(a, b, what='combine ' + nm, atol=1e-5)


def test_srgnn_layer_matches_oracle(dev):
    """K3 (srgnn.py:11-51): weighted-mean in/out aggregation + GRUCell vs the oracle layer."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import collate_ref as oc, models_ref as om
    sp = importlib.import_module('sessionrec-pytorch_amd')
    col = importlib.import_module('sessionrec-pytorch_amd.collate')
    gnn = importlib.import_module('sessionrec-pytorch_amd.gnn')
    srg = importlib.import_module('sessionrec-pytorch_amd.srgnn')
    rng = np.random.default_rng(3)
    samples = [(rng.integers(0, 30, size=int(rng.integers(1, 12))).tolist(), 1) for _ in range(20)]
    torch.manual_seed(0)
    d = 32
    ref = om.SRGNNLayer(d, d)
    mine = srg.SRGNNLayer(d, d)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev)
    (fb,), _ = col.collate_fn_factory(col.seq_to_session_graph)(samples)
    (ob,), _ = oc.collate_fn_factory(oc.seq_to_session_graph)(samples)
    ob = om.to_torch(ob)
    N = fb.count('N')
    x = torch.randn(N, d)
    xr = x.clone().requires_grad_()
    xg = x.to(dev).requires_grad_()
    out_r = ref(ob, xr)
    out_g = gnn.srgnn_layer(mine, fb.to(dev), xg)
    close(out_g, out_r, what='srgnn layer fwd')
    g = torch.randn(N, d)
    out_r.backward(g)
    out_g.backward(g.to(dev))
    close(xg.grad, xr.grad, what='srgnn layer dx', atol=2e-5)
    for (n1, p1), (n2, p2) in zip(mine.named_parameters(), ref.named_parameters()):
        close(p1.grad, p2.grad, what='srgnn layer ' + n1, atol=5e-5)


@pytest.mark.parametrize('name', ['srgnn_layer_s32', 'srgnn_layer_edge'])
def test_srgnn_layer_matches_the_reference_layer_fixture(dev, name):
    """row a3: gnn.srgnn_layer against the fixture produced by calling the REFERENCE's SRGNNLayer.forward directly
    (srgnn.py:31-51 / niser.py:29-49; tests/golden/make_golden.py srgnn_layer_cases) - not only against the oracle."""
    import os
    col = importlib.import_module('sessionrec-pytorch_amd.collate')
    gnn = importlib.import_module('sessionrec-pytorch_amd.gnn')
    srg = importlib.import_module('sessionrec-pytorch_amd.srgnn')
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.npz'), allow_pickle=False)
    samples = [([int(x) for x in s.split(',')], int(l)) for s, l in zip(z['seqs'].tolist(), z['labels'].tolist())]
    layer = srg.SRGNNLayer(32, 32)
    layer.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('init/')})
    layer = layer.to(dev)
    (fb,), _ = col.collate_fn_factory(col.seq_to_session_graph)(samples)
    x = torch.from_numpy(z['feat']).to(dev).requires_grad_()
    out = gnn.srgnn_layer(layer, fb.to(dev), x)
    out.backward(torch.from_numpy(z['gout']).to(dev))
    close(out, torch.from_numpy(z['out']), what='layer out')
    close(x.grad, torch.from_numpy(z['dfeat']), what='layer d feat', atol=2e-5)
    for k, p in layer.named_parameters():
        close(p.grad, torch.from_numpy(z['grad/' + k]), what='layer grad ' + k, atol=5e-5)


def test_batch_norm_prelu(dev):
    ops = _ops()
    torch.manual_seed(9)
    n, D = 200, 96
    x = torch.randn(n, D, device=dev) * 2 + 1
    for training in (True, False):
        bn1 = torch.nn.BatchNorm1d(D).to(dev)
        bn2 = torch.nn.BatchNorm1d(D).to(dev)
        with torch.no_grad():
            for b in (bn1, bn2):
                b.weight.copy_(torch.linspace(0.5, 1.5, D))
                b.bias.copy_(torch.linspace(-1, 1, D))
                b.running_mean.copy_(torch.linspace(-0.2, 0.2, D))
                b.running_var.copy_(torch.linspace(0.5, 2.0, D))
        bn1.train(training)
        bn2.train(training)
        x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
        y1, y2 = ops.batch_norm(x1, bn1), bn2(x2)
        close(y1, y2, what='bn fwd %s' % training, atol=2e-5)
        g = torch.randn_like(y2)
        y1.backward(g)
        y2.backward(g)
        close(x1.grad, x2.grad, what='bn dx', atol=2e-5)
        close(bn1.weight.grad, bn2.weight.grad, what='bn dgamma', atol=1e-4)
        close(bn1.bias.grad, bn2.bias.grad, what='bn dbeta', atol=1e-4)
        close(bn1.running_mean, bn2.running_mean, what='running mean')
        close(bn1.running_var, bn2.running_var, what='running var')
    a = torch.rand(D, device=dev, requires_grad=True)
    x1 = x.clone().requires_grad_()
    y = ops.prelu(x1, a)
    a2, x2 = a.detach().clone().requires_grad_(), x.clone().requires_grad_()
    r = torch.nn.functional.prelu(x2, a2)
    close(y, r, what='prelu')
    g = torch.randn_like(r)
    y.backward(g)
    r.backward(g)
    close(x1.grad, x2.grad, what='prelu dx')
    close(a.grad, a2.grad, what='prelu da', atol=1e-4)


@pytest.mark.parametrize('n,cap,D', [(200, 200, 96), (1000, 1280, 24), (17, 64, 300), (4097, 4097, 32)])
def test_batch_norm_chunk_statistics(dev, n, cap, D):
    """two-launch BatchNorm (per-chunk sum / M2 partials, combined inside the normalising kernel) == nn.BatchNorm1d on the
    live rows: capacity-padded input with a device-side row count, fewer rows than chunks, a column mean 100 x its
    deviation (the M2 combination must not cancel), running statistics and num_batches_tracked after two steps."""
    ops = _ops()
    torch.manual_seed(n + D)
    x = torch.randn(cap, D, device=dev) * torch.linspace(0.5, 3, D, device=dev) + torch.linspace(-5, 5, D, device=dev)
    x[:, 0] = 100.0 + torch.randn(cap, device=dev)
    x[n:] = 7e5                                               # padding rows: never read
    dyn = torch.tensor([n], dtype=torch.int32, device=dev)
    bn1, bn2 = torch.nn.BatchNorm1d(D).to(dev), torch.nn.BatchNorm1d(D).to(dev).double()      # yardstick: fp64
    for step in range(2):
        x1, x2 = x.clone().requires_grad_(), x[:n].double().requires_grad_()
        y1, y2 = ops.batch_norm(x1, bn1, dyn), bn2(x2)
        close(y1[:n], y2, what='y', atol=2e-4)                # column 0: x - mean loses 7 bits to the cancellation, both sides
        assert cap == n or float(y1.detach()[n:].abs().max()) == 0.0
        g = torch.randn(cap, D, device=dev)
        y1.backward(g)
        y2.backward(g[:n].double())
        close(x1.grad[:n], x2.grad, what='dx', atol=2e-4)
        close(bn1.weight.grad, bn2.weight.grad, what='dgamma', atol=2e-3, rtol=1e-4)
        close(bn1.bias.grad, bn2.bias.grad, what='dbeta', atol=2e-4, rtol=1e-4)
        bn1.zero_grad(); bn2.zero_grad()
    close(bn1.running_mean, bn2.running_mean, what='running mean', rtol=1e-5)
    close(bn1.running_var, bn2.running_var, what='running var', rtol=1e-4)
    assert int(bn1.num_batches_tracked) == int(bn2.num_batches_tracked) == 2
    a = torch.rand(D, device=dev, requires_grad=True)
    x1 = x.clone().requires_grad_()
    y = ops.prelu(x1, a, dyn)
    a2, x2 = a.detach().clone().requires_grad_(), x[:n].clone().requires_grad_()
    r = torch.nn.functional.prelu(x2, a2)
    g = torch.randn(cap, D, device=dev)
    y.backward(g)
    r.backward(g[:n])
    close(y[:n], r, what='prelu')
    close(x1.grad[:n], x2.grad, what='prelu dx')
    assert cap == n or float(x1.grad[n:].abs().max()) == 0.0
    close(a.grad, a2.grad, what='prelu da', rtol=1e-4, atol=1e-3 * float(a2.grad.abs().max()))


@pytest.mark.parametrize('D', [16, 24, 32, 48])
def test_gru_seq_matches_a_gru_cell_loop(dev, D):
    """EOPA's per-node GRU over the in-edges in edge-id order (lessr.py:20-27,35) vs a torch loop of GRU-cell steps on the
    precomputed input projections: last hidden state per node, dGI, dW_hh, db_hh.  D <= 32 runs the register-resident
    kernels, D = 48 the streaming ones; in-degrees 0 .. 9, 40 padded nodes."""
    ops = _ops()
    g = torch.Generator().manual_seed(D)
    N, cap = 300, 340
    deg = torch.randint(0, 10, (N,), generator=g)
    deg[5] = 0
    edst = torch.repeat_interleave(torch.arange(N), deg)
    E = int(edst.numel())
    perm = torch.randperm(E, generator=g)
    edst = edst[perm]                                         # edge ids in random order over the destinations
    esrc = torch.randint(0, N, (E,), generator=g)

    def csr(key):
        order = torch.sort(key, stable=True).indices          # edges of one node in edge-id order
        ptr = torch.zeros(cap + 1, dtype=torch.int64)
        ptr[1:N + 1] = torch.cumsum(torch.bincount(key, minlength=N), 0)
        ptr[N + 1:] = ptr[N]
        return ptr.to(torch.int32).to(dev), order.to(torch.int32).to(dev)
    in_ptr, in_idx = csr(edst)
    out_ptr, out_idx = csr(esrc)
    graph = (in_ptr, in_idx, out_ptr, out_idx, esrc.to(torch.int32).to(dev), edst.to(torch.int32).to(dev))
    dyn = torch.tensor([N], dtype=torch.int32, device=dev)
    dynE = torch.tensor([E], dtype=torch.int32, device=dev)
    GI = torch.randn(cap, 3 * D, generator=g).to(dev)
    Whh = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to(dev)
    bhh = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
    gy = torch.randn(cap, D, generator=g).to(dev)

    a = [t.clone().requires_grad_() for t in (GI, Whh, bhh)]
    out = ops.gru_seq(a[0], a[1], a[2], graph, dyn, dynE)
    out.backward(gy)

    b = [t.double().clone().requires_grad_() for t in (GI, Whh, bhh)]
    rows = []
    ins = [[] for _ in range(N)]
    for e in range(E):
        ins[int(edst[e])].append(e)
    for v in range(N):
        h = torch.zeros(D, dtype=torch.float64, device=dev)
        for e in ins[v]:
            gi, gh = b[0][int(esrc[e])], b[1] @ h + b[2]
            r = torch.sigmoid(gi[:D] + gh[:D])
            z = torch.sigmoid(gi[D:2 * D] + gh[D:2 * D])
            n = torch.tanh(gi[2 * D:] + r * gh[2 * D:])
            h = (1 - z) * n + z * h
        rows.append(h)
    ref = torch.stack(rows)
    ref.backward(gy[:N].double())
    close(out[:N], ref.float(), what='neigh', atol=2e-5)
    assert float(out[N:].abs().max()) == 0.0
    close(a[0].grad[:N], b[0].grad[:N].float(), what='dGI', atol=2e-5)
    close(a[1].grad, b[1].grad.float(), what='dWhh', atol=2e-4, rtol=1e-4)
    close(a[2].grad, b[2].grad.float(), what='dbhh', atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize('M,N,K', [(1, 4, 32), (200, 96, 64), (3000, 2048, 256), (513, 256, 2048)])
def test_gemm_bf16_matches_bf16_rounded_reference(dev, M, N, K):
    """bf16-operand MFMA with fp32 accumulation == fp32 matmul of bf16-rounded operands (tolerance: fp32
    accumulation order only); vs the unrounded fp32 product the error is the bf16 input rounding (2^-9 rel/elem)."""
    ops = _ops()
    torch.manual_seed(M + N)
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    ops.set_precision('bf16')
    try:
        y = torch.empty(M, N, device=dev)
        ops.gemm_nt(x, w, y, b)
        ref = x.bfloat16().float() @ w.bfloat16().float().t() + b
        close(y, ref, what='bf16 nt', rtol=1e-4, atol=1e-4)
        exact = x @ w.t() + b
        rel = (y - exact).norm() / exact.norm()
        assert rel < 6e-3, rel
        g = torch.randn(M, N, device=dev)
        gx = torch.empty(M, K, device=dev)
        ops.gemm_nn(g, w, gx)
        if N % 32 == 0:       # otherwise the exact fp32 kernel is (correctly) used
            ref = g.bfloat16().float() @ w.bfloat16().float()
            close(gx, ref, what='bf16 nn', rtol=1e-4, atol=2e-4)
    finally:
        ops.set_precision('fp32')


@pytest.mark.parametrize('M,N,K', [(256, 4, 8), (1000, 96, 64), (3001, 2048, 256), (5000, 256, 768)])
def test_gemm_bf16_tn_weight_gradient(dev, M, N, K):
    """dW = dY^T X with bf16 operands / fp32 accumulation == fp32 product of the bf16-rounded operands; a device-side
    live row count clamps the reduction (padded batches)."""
    ops = _ops()
    torch.manual_seed(M + K)
    g, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    ops.set_precision('bf16')
    try:
        out = torch.empty(N, K, device=dev)
        ops.gemm_tn(g, x, out)
        ref = g.bfloat16().float().t() @ x.bfloat16().float()
        close(out, ref, what='bf16 tn', rtol=1e-4, atol=1e-3)
        exact = g.t() @ x
        assert (out - exact).norm() / exact.norm() < 6e-3
        live = M - 137
        dyn = torch.tensor([live], device=dev, dtype=torch.int32)
        out2 = torch.full((N, K), 7.0, device=dev)
        ops.gemm_tn(g, x, out2, dyn)
        ref2 = g[:live].bfloat16().float().t() @ x[:live].bfloat16().float()
        close(out2, ref2, what='bf16 tn dyn', rtol=1e-4, atol=1e-3)
        out3 = torch.ones(N, K, device=dev)
        ops.gemm_tn(g, x, out3, None, beta=1.0)
        close(out3, ref + 1.0, what='bf16 tn beta', rtol=1e-4, atol=1e-3)
    finally:
        ops.set_precision('fp32')


def test_score_ce_full_size_properties(dev):
    """BASELINE full size (C3: V=37 484, d=256, B=512; logits would be 77 MB) through size-independent
    properties instead of a materialised reference:
      (1) identical catalog rows  -> loss = log V exactly, d sr = 0, every dE row = (1/V - [v==label]) sr / B summed;
      (2) gradient linearity in the upstream scale;  (3) sum_v dE_v = -(1/B) sum_b (1 - 1) ... = 0 row-sum identity:
          sum_v dS[b,v] = 0  =>  sum_v dE_v = sum_b (sum_v dS[b,v]) sr_b = 0;
      (4) fused loss == -mean(log-prob[label]) from the LOGP pass on a slice of sessions."""
    ops = _ops()
    torch.manual_seed(11)
    B, V, d = 512, 37484, 256
    sr = torch.randn(B, d, device=dev) * 0.2
    labels = torch.randint(0, V, (B,), device=dev)
    ws = ops.CEWorkspace(B, V, d, dev)
    # (1) identical rows
    e = torch.randn(d, device=dev) * 0.1
    E = e.unsqueeze(0).repeat(V, 1).contiguous()
    tg = ops.TableGrad(E)
    s1 = sr.clone().requires_grad_()
    loss, _ = ops.score_ce(s1, E, None, labels.int(), ws, tg)
    loss.backward()
    assert abs(loss.item() - float(np.log(V))) < 1e-4, loss.item()
    assert s1.grad.abs().max().item() < 1e-6
    # (2)+(3) random catalog
    E = (torch.randn(V, d, device=dev) * 0.1).contiguous()
    tg = ops.TableGrad(E)
    s2 = sr.clone().requires_grad_()
    loss, _ = ops.score_ce(s2, E, None, labels.int(), ws, tg)
    (3.0 * loss).backward()
    g3, dE3 = s2.grad.clone(), tg.buf.clone()
    s3 = sr.clone().requires_grad_()
    loss2, _ = ops.score_ce(s3, E, None, labels.int(), ws, tg)
    loss2.backward()
    close(g3, 3.0 * s3.grad, what='linearity dsr', rtol=1e-5, atol=1e-9)
    close(dE3, 3.0 * tg.buf, what='linearity dE', rtol=1e-5, atol=1e-9)
    colsum = tg.buf.double().sum(0)
    assert colsum.abs().max().item() < 1e-5, colsum.abs().max().item()
    # (4) against the materialised log-probabilities of the first 64 sessions
    logp = ops.score_logp(sr[:64].contiguous(), E, None, ops.CEWorkspace(64, V, d, dev))
    ref = -logp.gather(1, labels[:64].unsqueeze(1)).mean()
    l64, _ = ops.score_ce(sr[:64].contiguous(), E, None, labels[:64].int(), ops.CEWorkspace(64, V, d, dev), ops.TableGrad(E))
    close(l64, ref, what='loss vs logp', rtol=1e-5, atol=1e-5)
    assert torch.allclose(torch.logsumexp(logp, 1), torch.zeros(64, device=dev), atol=1e-4)


@pytest.mark.parametrize('B,V,d,cosine', [(512, 5000, 256, True), (100, 1000, 96, False), (37, 700, 64, True), (64, 64, 32, False)])
def test_score_ce_bf16_within_stated_tolerance(dev, B, V, d, cosine):
    """bf16-operand scoring kernels vs the fp32 reference.  Stated bf16 tolerances: loss 5e-3 rel;
    gradients: norm-wise relative error <= 2e-2 (bf16 operands + bf16 P, fp32 accumulation)."""
    ops = _ops()
    torch.manual_seed(B + V)
    sr = (torch.randn(B, d, device=dev) * 0.3)
    E = (torch.randn(V, d, device=dev) * 0.3)
    labels = torch.randint(0, V, (B,), device=dev)
    cs = (12.0 / E.norm(dim=1)).contiguous() if cosine else None
    if cosine:
        sr = torch.nn.functional.normalize(sr, dim=1)
    ws = ops.CEWorkspace(B, V, d, dev)
    tg = ops.TableGrad(E)
    tb = ops.TableBF16(E).refresh(E)
    srg = sr.clone().requires_grad_()
    loss, lse = ops.score_ce(srg, E, cs, labels.int(), ws, tg, None, 1.0 / 12.0, tb)
    loss.backward()
    sr2, E2 = sr.clone().requires_grad_(), E.clone().requires_grad_()
    z = sr2 @ (torch.nn.functional.normalize(E2, dim=1) * 12.0 if cosine else E2).t()
    ref = torch.nn.functional.cross_entropy(z, labels)
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 5e-3 * abs(ref.item()), (loss.item(), ref.item())
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(srg.grad, sr2.grad) < 2e-2, rel(srg.grad, sr2.grad)
    assert rel(tg.buf, E2.grad) < 2e-2, rel(tg.buf, E2.grad)
    close(lse, torch.logsumexp(z, 1), what='lse', rtol=0, atol=3e-2)


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_gemm_group_bf16(dev, mode):
    """grouped / K-segmented bf16 GEMM == per-problem fp32 products of the bf16-rounded operands"""
    ops = _ops()
    torch.manual_seed(mode)
    r = lambda *s: torch.randn(*s, device=dev)
    bf = lambda t: t.bfloat16().float()
    if mode == 0:
        xs, ws = [r(300, 64), r(1000, 64)], [r(128, 64), r(128, 64)]
        Cs = [torch.empty(300, 128, device=dev), torch.full((1000, 128), 5.0, device=dev)]
        dyn = torch.tensor([777], device=dev, dtype=torch.int32)
        ops.gemm_group(0, [(300, 128, 64, [(xs[0], ws[0])], Cs[0], None), (1000, 128, 64, [(xs[1], ws[1])], Cs[1], dyn)], 64, 64, 128)
        close(Cs[0], bf(xs[0]) @ bf(ws[0]).t(), what='group nt 0', rtol=1e-4, atol=1e-3)
        ref = bf(xs[1]) @ bf(ws[1]).t()
        ref[777:] = 0
        close(Cs[1], ref, what='group nt 1 (dyn rows zeroed)', rtol=1e-4, atol=1e-3)
    elif mode == 1:
        g1, g2, g3 = r(500, 256), r(500, 256), r(200, 256)
        w1, w2 = r(256, 64), r(256, 64)
        C0, C1 = torch.ones(500, 64, device=dev), torch.ones(200, 64, device=dev)
        dyn = torch.tensor([450], device=dev, dtype=torch.int32)
        ops.gemm_group(1, [(500, 64, 256, [(g1, w1), (g2, w2)], C0, dyn), (200, 64, 256, [(g3, w2)], C1, None)], 256, 64, 64, beta=1.0)
        ref0 = 1.0 + bf(g1) @ bf(w1) + bf(g2) @ bf(w2)
        ref0[450:] = 1.0
        close(C0, ref0, what='group nn segmented', rtol=1e-4, atol=2e-3)
        close(C1, 1.0 + bf(g3) @ bf(w2), what='group nn 1', rtol=1e-4, atol=2e-3)
    else:
        g1, x1, g2, x2 = r(900, 128), r(900, 64), r(333, 128), r(333, 64)
        C0, C1 = torch.empty(128, 64, device=dev), torch.empty(128, 64, device=dev)
        dyn = torch.tensor([801], device=dev, dtype=torch.int32)
        ops.gemm_group(2, [(128, 64, 900, [(g1, x1)], C0, dyn), (128, 64, 333, [(g2, x2)], C1, None)], 128, 64, 64)
        close(C0, bf(g1[:801]).t() @ bf(x1[:801]), what='group tn dyn', rtol=1e-4, atol=2e-3)
        close(C1, bf(g2).t() @ bf(x2), what='group tn', rtol=1e-4, atol=2e-3)


def test_gemm_group_bf16_large_tiles_and_bf16_storage(dev):
    """the 128x128-tile forward variant (taken when it still fills the chip) and bf16-stored C / A operands, at the shapes
    of the MSHGNN layer: forward into a bf16 projection, weight gradient from a bf16 projection gradient"""
    ops = _ops()
    torch.manual_seed(5)
    r = lambda *s: torch.randn(*s, device=dev)
    bf = lambda t: t.bfloat16().float()
    xs = [r(1100 + 100 * i, 64) for i in range(8)]
    ws = [r(2048, 64) for _ in range(8)]
    Ps = [torch.empty(x.shape[0], 2048, device=dev, dtype=torch.bfloat16) for x in xs]
    dyn = torch.tensor([1000], device=dev, dtype=torch.int32)
    ops.gemm_group(0, [(x.shape[0], 2048, 64, [(x, w)], P, dyn if i == 3 else None) for i, (x, w, P) in enumerate(zip(xs, ws, Ps))],
                   64, 64, 2048, c16=True)
    for i, (x, w, P) in enumerate(zip(xs, ws, Ps)):
        ref = (bf(x) @ bf(w).t())
        if i == 3:
            ref[1000:] = 0
        close(P.float(), ref.bfloat16().float(), what='group nt 128 -> bf16 C (%d)' % i, rtol=1e-2, atol=2e-2)
    gs = [r(x.shape[0], 2048).bfloat16() for x in xs]
    xw = [r(x.shape[0], 256) for x in xs]
    Cs = [torch.empty(2048, 256, device=dev) for _ in xs]
    ops.gemm_group(2, [(2048, 256, x.shape[0], [(g, x2)], C, dyn if i == 5 else None)
                       for i, (x, g, x2, C) in enumerate(zip(xs, gs, xw, Cs))], 2048, 256, 256, a16=True)
    for i, (g, x2, C) in enumerate(zip(gs, xw, Cs)):
        n = 1000 if i == 5 else g.shape[0]
        close(C, g[:n].float().t() @ bf(x2[:n]), what='group tn 128, bf16 A (%d)' % i, rtol=1e-4, atol=5e-3)


@pytest.mark.parametrize('B,V,d,K', [(5, 300, 32, 20), (512, 37484, 256, 20), (33, 5000, 96, 7)])
def test_score_topk_matches_materialised_ranking(dev, B, V, d, K):
    """fused top-K == torch.topk of the materialised score matrix (values exactly comparable up to fp32 dot-product
    order; ids equal wherever neighbouring scores are separated by more than that round-off)"""
    ops = _ops()
    torch.manual_seed(B + V)
    sr = torch.randn(B, d, device=dev) * 0.3
    E = torch.randn(V, d, device=dev) * 0.2
    cs = torch.rand(V, device=dev) + 0.5
    val, idx = ops.score_topk(sr, E, cs, K)
    z = (sr.double() @ E.double().t()) * cs.double()
    rv, ri = z.topk(K, dim=1)
    close(val, rv.float(), what='top-k values', rtol=1e-4, atol=1e-4)
    # ids must agree at every rank whose score is separated from both neighbours (and from the best excluded item) by
    # more than the fp32 round-off of a d-term dot product
    nxt = z.masked_fill(torch.zeros_like(z, dtype=torch.bool).scatter_(1, ri, True), -1e30).max(1)[0]
    ext = torch.cat([torch.full_like(rv[:, :1], 1e30), rv, nxt[:, None]], 1)
    sep = ((ext[:, :-2] - rv) > 1e-4) & ((rv - ext[:, 2:]) > 1e-4)
    assert sep.float().mean() > 0.8
    assert torch.equal(idx.long()[sep], ri[sep])
    assert (val[:, :-1] >= val[:, 1:]).all()
    # the ids returned always carry the returned values
    close(z.gather(1, idx.long()).float(), val, what='values at the returned ids', rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------- the measured shapes, at full size
def _ce_reference_chunked(sr, E, cs, labels, chunk=512):
    """fp32 torch reference of mean CE(cs * sr E^T, labels) with its gradients, materialising only `chunk` sessions of
    logits at a time (a full (B, V) matrix is 20 GB at the C5 per-rank shape)"""
    B = sr.shape[0]
    En = E if cs is None else E * cs.unsqueeze(1)          # cs[v] * E_v: the scaled rows the logits are taken against
    dE = torch.zeros_like(E)
    dsr = torch.empty_like(sr)
    lse = torch.empty(B, device=sr.device)
    loss = 0.0
    for b0 in range(0, B, chunk):
        s = sr[b0:b0 + chunk]
        z = s @ En.t()
        l = torch.logsumexp(z, 1)
        lse[b0:b0 + chunk] = l
        lab = labels[b0:b0 + chunk].long()
        loss += float((l - z.gather(1, lab[:, None])[:, 0]).double().sum())
        p = torch.exp(z - l[:, None])
        p[torch.arange(p.shape[0], device=p.device), lab] -= 1.0
        p /= B
        dsr[b0:b0 + chunk] = p @ En
        dE += p.t() @ s
        del z, p
    if cs is not None:
        dE *= cs.unsqueeze(1)          # d / d(scaled row) -> d / d E_v at fixed cs (the kernels' dE before rownorm_project)
    return loss / B, lse, dsr, dE


@pytest.mark.parametrize('B,V,d,cosine,tag', [(512, 37484, 256, True, 'C3: the launch BENCH reports'),
                                              (512, 43097, 96, False, 'C2: SRGNN / Diginetica'),
                                              (512, 43097, 96, True, 'C2 shape, cosine (NISER)'),
                                              (4096, 4332, 256, True, 'C3 / C4 as rank 7 of 8 scores it (weak scaling): split 7'),
                                              (2048, 9260, 256, True, 'C3 as a rank of 4 scores it: split 4'),
                                              (1000, 18732, 256, True, 'a rank of 2, ragged session count (1000 < 1024: one piece)')])
@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_score_ce_at_benchmarked_shapes(dev, B, V, d, cosine, tag, precision):
    """The fused scoring / CE kernels at exactly the shapes the benchmark and BASELINE configs C2 / C3 name (the bf16
    backward's 293-item-tile + range grid at V = 37 484; d = 96 padded to 128 in the bf16 kernels) against a
    materialised fp32 torch reference.  fp32 mode: loss 1e-5 rel, gradients 1e-4; bf16 mode: SURVEY 8(c) - loss
    5e-3 rel, lse atol 3e-2, gradients 2e-2 norm-wise; plus the size-independent identity sum_v dS[b, v] = 0."""
    ops = _ops()
    torch.manual_seed(V + d)
    sr = torch.randn(B, d, device=dev) * 0.3
    E = torch.randn(V, d, device=dev) * 0.3
    labels = torch.randint(0, V, (B,), device=dev)
    cs = (12.0 / E.norm(dim=1)).contiguous() if cosine else None
    if cosine:
        sr = torch.nn.functional.normalize(sr, dim=1)
    ref_loss, ref_lse, ref_dsr, ref_dE = _ce_reference_chunked(sr, E, cs, labels)
    ops.set_precision(precision)
    try:
        ws = ops.CEWorkspace(B, V, d, dev)
        tg = ops.TableGrad(E)
        tb = ops.TableBF16(E).refresh(E) if ops.use_bf16_scoring(d) else None
        assert (tb is not None) == (precision == 'bf16')
        srg = sr.clone().requires_grad_()
        loss, lse = ops.ScoreCE.apply(srg, E, cs, labels.int(), ws, tg, None, 0.0, tb)
        # cs_inv_scale = 0: rownorm_project subtracts nothing -> tg.buf is d loss / d E_v at fixed cs, like the reference
        loss.backward()
        if precision == 'bf16' and 'split' in tag:
            # the session-split backward (round 6: item tiles of a row shard cut over the sessions of all ranks) ran: its
            # slab workspace exists and has the advertised number of pieces
            want = int(tag.split('split ')[1])
            (split, slabs), = ws._de.values()
            assert split == want and slabs is not None and slabs.numel() == split * V * d, (split, want)
    finally:
        ops.set_precision('fp32')
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    if precision == 'fp32':
        assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss), (loss.item(), ref_loss)
        close(lse, ref_lse, what='lse', rtol=1e-5, atol=1e-4)
        assert rel(srg.grad, ref_dsr) < 1e-4 and rel(tg.buf, ref_dE) < 1e-4, (rel(srg.grad, ref_dsr), rel(tg.buf, ref_dE))
    else:
        assert abs(loss.item() - ref_loss) <= 5e-3 * abs(ref_loss), (loss.item(), ref_loss)
        close(lse, ref_lse, what='lse', rtol=0, atol=3e-2)
        assert rel(srg.grad, ref_dsr) < 2e-2, rel(srg.grad, ref_dsr)
        assert rel(tg.buf, ref_dE) < 2e-2, rel(tg.buf, ref_dE)
    # sum_v dS[b, v] = 0  =>  sum_v dE_v / cs_v = 0 (exact algebra; fp32 accumulation noise only, both precisions)
    colsum = (tg.buf.double() / (cs.double().unsqueeze(1) if cs is not None else 1.0)).sum(0)
    scale = (tg.buf.double().abs() / (cs.double().unsqueeze(1) if cs is not None else 1.0)).sum(0).max().item()
    assert colsum.abs().max().item() < (1e-5 if precision == 'fp32' else 2e-3) * scale, (colsum.abs().max().item(), scale)


def test_score_ce_at_the_c5_per_rank_shape(dev):
    """BASELINE config C5 as ONE rank sees it: V/8 = 1.25 M rows, d = 256, B = 4096 gathered sessions (logits would be
    20 GB: never materialised by the product; the reference below walks them 256 sessions at a time).  fp32 and bf16
    kernels + the size-independent identities, then the same shard through dist.VocabParallel on a 1-rank world."""
    ops = _ops()
    torch.manual_seed(5)
    B, V, d = 4096, 1_250_000, 256
    sr = torch.nn.functional.normalize(torch.randn(B, d, device=dev), dim=1)
    E = torch.randn(V, d, device=dev) * 0.06
    labels = torch.randint(0, V, (B,), device=dev)
    cs = (12.0 / E.norm(dim=1)).contiguous()
    ref_loss, ref_lse, ref_dsr, ref_dE = _ce_reference_chunked(sr, E, cs, labels, chunk=256)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for precision in ('fp32', 'bf16'):
        ops.set_precision(precision)
        try:
            ws = ops.CEWorkspace(B, V, d, dev)
            tg = ops.TableGrad(E)
            tb = ops.TableBF16(E).refresh(E) if precision == 'bf16' else None
            srg = sr.clone().requires_grad_()
            loss, lse = ops.ScoreCE.apply(srg, E, cs, labels.int(), ws, tg, None, 0.0, tb)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            ops.set_precision('fp32')
        tol_l, tol_g = (1e-5, 1e-4) if precision == 'fp32' else (5e-3, 2e-2)
        assert abs(loss.item() - ref_loss) <= tol_l * abs(ref_loss), (precision, loss.item(), ref_loss)
        assert rel(srg.grad, ref_dsr) < tol_g and rel(tg.buf, ref_dE) < tol_g, (precision, rel(srg.grad, ref_dsr), rel(tg.buf, ref_dE))
        del ws, tg, tb
    # eval at this size: fused top-20 == ranking of materialised scores for a slice of the sessions
    val, idx = ops.score_topk(sr[:64].contiguous(), E, cs, 20)
    z = (sr[:64] @ (E * cs.unsqueeze(1)).t())
    tv, ti = z.topk(20, dim=1)
    assert torch.equal(idx.long(), ti)
    close(val, tv, what='top-20 scores', rtol=1e-5, atol=1e-5)
    # the same shard through the row-sharded path (1-rank world: lookup / scoring / merge kernels, no exchange)
    D = importlib.import_module('sessionrec-pytorch_amd.dist')

    class Holder:
        shard = None

        def __init__(self, w):
            self.w = torch.nn.Parameter(w)

        def _table(self):
            return self.w

        def _state(self, B):
            st = self.__dict__.setdefault('_srec_state', {})
            if 'tgrad' not in st:
                st['tgrad'] = ops.TableGrad(self.w)
            return st

    holder = Holder(E)
    vp = D.VocabParallel(holder)                     # the parameter now holds this rank's (tile-padded) rows
    srg = sr.clone().requires_grad_()
    loss = vp.loss(srg, holder._table(), cs, labels, 0.0)
    loss.backward()
    assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss), (loss.item(), ref_loss)
    assert rel(srg.grad, ref_dsr) < 1e-4 and rel(vp.dE[:V], ref_dE) < 1e-4


# ---------------------------------------------------------------------- bf16-in-HBM GEMMs (csrc/gemm16.hip)
def test_gemm16_operand_copies(dev):
    ops = _ops()
    torch.manual_seed(1)
    ws = [torch.randn(2048, 256, device=dev), torch.randn(768, 256, device=dev), torch.randn(100, 36, device=dev)]
    w16, wt16 = ops.weights_bf16(ws)
    for w, a, b in zip(ws, w16, wt16):
        assert torch.equal(a, w.bfloat16()) and torch.equal(b, w.bfloat16().t().contiguous())
    x = torch.randn(777, 64, device=dev)
    dyn = torch.tensor([700], device=dev, dtype=torch.int32)
    y = ops.rows_bf16(x, dyn)
    ref = x.bfloat16()
    ref[700:] = 0
    assert torch.equal(y, ref)


@pytest.mark.parametrize('c16', [True, False])
def test_gemm16_nt(dev, c16):
    """grouped NT product with bf16 operands == fp32 matmul of the same bf16 values (fp32 accumulation order only);
    K-segments are summed, dyn clamps the output rows, beta accumulates, ragged M / N tiles are masked"""
    ops = _ops()
    torch.manual_seed(2)
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
    for shapes in ([(300, 2048, 256), (1000, 2048, 256)], [(3850, 2048, 256)] * 3, [(130, 136, 64)]):
        probs, refs = [], []
        for (M, N, K) in shapes:
            A, B = bf(M, K), bf(N, K)
            dyn = torch.tensor([max(1, M - 77)], device=dev, dtype=torch.int32) if M > 200 else None
            C = torch.full((M, N), 3.0, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
            r = A.float() @ B.float().t()
            if dyn is not None:
                r[M - 77:] = 0
            probs.append((M, N, K, [(A, B)], C, dyn))
            refs.append(r)
        ops.gemm16('nt', probs, shapes[0][2], shapes[0][2], shapes[0][1], c16=c16)
        for (M, N, K, _, C, _), r in zip(probs, refs):
            if c16:
                close(C.float(), r.bfloat16().float(), what='nt16 c16', rtol=1e-2, atol=1e-2)     # one bf16 ulp
            else:
                close(C, r, what='nt16 f32', rtol=1e-4, atol=1e-3)
    if c16:
        # forward projections with capacity padding: dead rows zeroed, or left alone (keep_dead)
        for (M, N, K, live) in ((2560, 2048, 256, 2371), (777, 1024, 128, 500), (96, 256, 256, 96), (4000, 2048, 256, 31)):
            A, B = bf(M, K), bf(N, K)
            dyn = torch.tensor([live], device=dev, dtype=torch.int32)
            outs = []
            for keep in (False, True):
                C = torch.full((M, N), 3.0, device=dev, dtype=torch.bfloat16)
                ops.gemm16('nt', [(M, N, K, [(A, B)], C, dyn)], K, K, N, c16=True, keep_dead=keep)
                outs.append(C)
            r = (A.float() @ B.float().t()).bfloat16()
            r[live:] = 0
            close(outs[0].float(), r.float(), what='fwd %r' % ((M, N, K),), rtol=1e-2, atol=1e-2)
            assert torch.equal(outs[1][:live], outs[0][:live]) and bool((outs[1][live:] == 3.0).all())
    if not c16:   # segments + beta (backward-data: sum over modules, accumulated onto the residual gradient)
        M, N, K = 1500, 256, 2048
        A1, A2, B1, B2 = bf(M, K), bf(M, K), bf(N, K), bf(N, K)
        C = torch.randn(M, N, device=dev)
        ref = C + A1.float() @ B1.float().t() + A2.float() @ B2.float().t()
        dyn = torch.tensor([1400], device=dev, dtype=torch.int32)
        ref[1400:] = C[1400:]                                   # beta != 0: rows past the live count are left alone
        ops.gemm16('nt', [(M, N, K, [(A1, B1), (A2, B2)], C, dyn)], K, K, N, beta=1.0)
        close(C, ref, what='nt16 segments + beta', rtol=1e-4, atol=2e-3)


def test_gemm16_tn(dev):
    """weight-gradient product C = A^T B over the live rows, operands row-major bf16 (16-bit transposing LDS reads)"""
    ops = _ops()
    torch.manual_seed(3)
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
    probs, refs = [], []
    for (R, N1, N2, live) in ((1391, 2048, 256, None), (3850, 2048, 256, 3333), (64, 2048, 256, 1), (500, 768, 256, 0),
                              (257, 136, 72, 200)):
        A, B = bf(R, N1), bf(R, N2)
        dyn = torch.tensor([live], device=dev, dtype=torch.int32) if live is not None else None
        n = R if live is None else live
        C = torch.full((N1, N2), 9.0, device=dev)
        ops.gemm16('tn', [(N1, N2, R, [(A, B)], C, dyn)], N1, N2, N2)
        close(C, A[:n].float().t() @ B[:n].float(), what='tn16 R=%d live=%s' % (R, live), rtol=1e-4, atol=2e-3)
    # grouped: several modules in one launch
    As, Bs = [bf(1000, 2048), bf(1300, 2048)], [bf(1000, 256), bf(1300, 256)]
    Cs = [torch.empty(2048, 256, device=dev) for _ in range(2)]
    ops.gemm16('tn', [(2048, 256, 1000, [(As[0], Bs[0])], Cs[0], None), (2048, 256, 1300, [(As[1], Bs[1])], Cs[1], None)], 2048, 256, 256)
    for A, B, C in zip(As, Bs, Cs):
        close(C, A.float().t() @ B.float(), what='tn16 grouped', rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize('d,padded', [(64, False), (256, True), (128, True)])
def test_gru_expand_all_matches_per_order_fp32_path(dev, d, padded):
    """csrc/grux.hip + gemm16.hip (all orders per launch, bf16 operands) against the per-order exact-fp32 GRUExpand node
    (itself pinned by the msgifsr_K2 / K3 fixtures): outputs and every gradient at the bf16 tolerance of SURVEY 8(c)
    (norm-wise 2e-2); capacity-padded nodes produce exact zeros and receive no gradient."""
    ops = _ops()
    torch.manual_seed(d)
    ks, caps, lives = [2, 3], [300, 280], [300, 280]
    if padded:
        caps, lives = [512, 512], [401, 333]
    grus = [torch.nn.GRU(d, d, 1, True, True).to(dev) for _ in ks]
    for g in grus:
        for w in g.parameters():
            w.data.uniform_(-1 / d ** 0.5, 1 / d ** 0.5)
    G = sum(c * k for c, k in zip(caps, ks))
    rows = torch.randn(G, d, device=dev) * 0.5
    offs = [0, caps[0] * ks[0]]
    dyn_n = [torch.tensor([l], device=dev, dtype=torch.int32) if padded else None for l in lives]
    dyn_r = [torch.tensor([l * k], device=dev, dtype=torch.int32) if padded else None for l, k in zip(lives, ks)]
    if padded:
        for o, c, l, k in zip(offs, caps, lives, ks):
            rows[o + l * k:o + c * k] = 0
    gout = [torch.randn(c, d, device=dev) for c in caps]

    def run(fast):
        r = rows.clone().requires_grad_()
        xs = [r[o:o + c * k] for o, c, k in zip(offs, caps, ks)]
        for g in grus:
            g.zero_grad()
        if fast:
            outs = ops.gru_expand_all(xs, grus, ks, dyn_n, dyn_r)
        else:
            outs = [ops.gru_expand(x, g, k, dn, dr) for x, g, k, dn, dr in zip(xs, grus, ks, dyn_n, dyn_r)]
        torch.autograd.backward(list(outs), gout)
        return [o.detach().clone() for o in outs], r.grad.clone(), [[p.grad.clone() for p in g.parameters()] for g in grus]

    ref = run(False)
    ops.set_precision('bf16')
    try:
        assert ops.gru_expand_fast_ok(d, 'mean')
        got = run(True)
    finally:
        ops.set_precision('fp32')
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp(min=1e-12)).item()
    for p in range(2):
        assert rel(got[0][p], ref[0][p]) < 1e-2, ('out', p, rel(got[0][p], ref[0][p]))
        if padded:
            assert got[0][p][lives[p]:].abs().max().item() == 0.0
    assert rel(got[1], ref[1]) < 2e-2, ('dx', rel(got[1], ref[1]))
    if padded:
        for o, c, l, k in zip(offs, caps, lives, ks):
            assert got[1][o + l * k:o + c * k].abs().max().item() == 0.0
    for p in range(2):
        for a, b, nm in zip(got[2][p], ref[2][p], ('Wih', 'Whh', 'bih', 'bhh')):
            assert rel(a, b) < 2e-2, (p, nm, rel(a, b))


@pytest.mark.parametrize('d,padded,big', [(256, True, False), (128, True, False), (256, False, True), (128, False, True),
                                          (256, True, True), (128, False, False)])
def test_gru_fused_forward_equals_the_step_path(dev, d, padded, big, monkeypatch):
    """csrc/gruf.hip (whole recurrence in one launch: a workgroup owns 16 or 32 nodes, weights streamed fragment-major) against
    the step-by-step bf16 path (grux.hip + gemm16.hip) it replaces: same operands, same rounding points, only the order of
    the fp32 partial sums differs - outputs at 1e-4 and every gradient of the fused backward (csrc/grufb.hip) at 1.5e-3 / 3e-3.
    Both workgroup shapes: the launcher takes 16-node workgroups while 32-node ones would leave most of the chip idle
    (<= 192 tiles: the small cases) and 32-node ones beyond (`big`)."""
    ops = _ops()
    torch.manual_seed(d + 7)
    ks, caps, lives = [2, 3], [333, 290], [333, 290]          # not multiples of the 32-node tile
    if padded:
        caps, lives = [512, 480], [401, 37]
    if big:
        caps, lives = [c * 10 + 3 for c in caps], [l * 10 + 3 for l in lives]
    grus = [torch.nn.GRU(d, d, 1, True, True).to(dev) for _ in ks]
    for g in grus:
        for w in g.parameters():
            w.data.uniform_(-1 / d ** 0.5, 1 / d ** 0.5)
    G = sum(c * k for c, k in zip(caps, ks))
    rows = torch.randn(G, d, device=dev) * 0.5
    offs = [0, caps[0] * ks[0]]
    dyn_n = [torch.tensor([l], device=dev, dtype=torch.int32) if padded else None for l in lives]
    dyn_r = [torch.tensor([l * k], device=dev, dtype=torch.int32) if padded else None for l, k in zip(lives, ks)]
    gout = [torch.randn(c, d, device=dev) for c in caps]

    def run():
        r = rows.clone().requires_grad_()
        xs = [r[o:o + c * k] for o, c, k in zip(offs, caps, ks)]
        for g in grus:
            g.zero_grad()
        outs = ops.gru_expand_all(xs, grus, ks, dyn_n, dyn_r)
        torch.autograd.backward(list(outs), gout)
        return [o.detach().clone() for o in outs], r.grad.clone(), [[p.grad.clone() for p in g.parameters()] for g in grus]

    ops.set_precision('bf16')
    try:
        monkeypatch.setattr(ops, 'FUSED_GRU', False)
        assert not ops.gru_fused_ok(d, 2)
        ref = run()
        monkeypatch.setattr(ops, 'FUSED_GRU', True)
        assert ops.gru_fused_ok(d, 2)
        got = run()
    finally:
        ops.set_precision('fp32')
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp(min=1e-12)).item()
    for p in range(2):
        assert rel(got[0][p], ref[0][p]) < 1e-4, ('out', p, rel(got[0][p], ref[0][p]))
        assert (got[0][p] - ref[0][p]).abs().max().item() < 1e-4
        if padded:
            assert got[0][p][lives[p]:].abs().max().item() == 0.0
    # the fused path saves its gates as fp16 (2^-11 relative; the step path fp32): 4.5e-4 on d x, 1.1e-3 on the weight gradients -
    # below the bf16 rounding (2^-9) of the d(gi) / d(gh) operands both paths feed to the weight-gradient GEMM
    assert rel(got[1], ref[1]) < 1.5e-3, ('dx', rel(got[1], ref[1]))
    for p in range(2):
        for a, b, nm in zip(got[2][p], ref[2][p], ('Wih', 'Whh', 'bih', 'bhh')):
            assert rel(a, b) < 3e-3, (p, nm, rel(a, b))


def test_lookup_with_fused_dropout(dev):
    """feature dropout fused into the embedding gather and its backward (msgifsr.py:247): the output is table[idx] times a
    0 / (1 / (1 - p)) mask with keep-rate 1 - p, the backward applies the SAME mask (recomputed from the counter-based
    hash), the mask follows torch.manual_seed and changes with the device step counter"""
    ops = _ops()
    torch.manual_seed(4)
    V, d, n, p = 500, 64, 3000, 0.3
    table = (torch.rand(V, d, device=dev) + 0.5).requires_grad_()
    idx = torch.randint(0, V, (n,), device=dev, dtype=torch.int32)
    items, inv = torch.unique(idx.long(), return_inverse=True)
    pos = torch.argsort(idx.long(), stable=True).int()
    uptr = torch.zeros(items.numel() + 1, dtype=torch.int32, device=dev)
    uptr[1:] = torch.bincount(inv).cumsum(0).int()
    uniq = (items.int(), uptr, pos)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    old = ops.RNG_COUNTER.get(str(dev))
    ops.RNG_COUNTER[str(dev)] = counter
    try:
        outs = []
        for seed, cnt in ((11, 0), (11, 0), (12, 0), (11, 1)):
            torch.manual_seed(seed)
            ops.seed_dropout()                                   # (an equal seed value is not observable: explicit restart)
            counter.fill_(cnt)
            table.grad = None
            out = ops.embedding_lookup(table, idx, uniq, None, None, None, (p, 7))
            mask = (out / table[idx.long()]).detach()
            keep = (mask > 0)
            assert torch.allclose(mask[keep], torch.full_like(mask[keep], 1 / (1 - p)), rtol=1e-5)
            assert abs(keep.float().mean().item() - (1 - p)) < 0.01
            g = torch.randn(n, d, device=dev)
            out.backward(g)
            ref = torch.zeros(V, d, device=dev).index_add_(0, idx.long(), g * mask)
            close(table.grad, ref, what='dropout lookup backward', rtol=1e-5, atol=1e-5)
            outs.append(mask)
        assert torch.equal(outs[0], outs[1])                     # same seed, same step: same mask
        assert not torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[3])
        # rows and columns are not correlated: every row / column keeps about 1 - p
        k = (outs[0] > 0).float()
        assert (k.mean(0) - (1 - p)).abs().max() < 0.05 and (k.mean(1) - (1 - p)).abs().max() < 0.25
    finally:
        if old is None:
            ops.RNG_COUNTER.pop(str(dev), None)
        else:
            ops.RNG_COUNTER[str(dev)] = old


def test_gemm_f32_group_mixed_layouts(dev):
    """one grouped launch of exact-fp32 products in all three operand layouts, with bias, beta, dynamic row counts, a strided
    output view and a long-K weight gradient (k-split into slabs + grouped reduce)"""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    NT, B, d = 3000, 512, 256
    allf, Wu, bu, v, Wv = r(NT, d), r(d, d), r(d), r(B, d), r(d, d)
    dU, dVq = r(NT, d), r(B, d)
    liveT = torch.tensor([2777], device=dev, dtype=torch.int32)
    liveB = torch.tensor([500], device=dev, dtype=torch.int32)
    U, Vq = torch.full((NT, d), 7.0, device=dev), torch.empty(B, d, device=dev)
    dX0 = r(NT, d)
    dX0[2777:] = 0
    dX = dX0.clone()
    gWu = torch.empty(d, d, device=dev)
    wide0 = r(B, 2 * d)
    wide = wide0.clone()
    gWv = torch.empty(d, d, device=dev)
    small = torch.empty(36, 8, device=dev)
    sa, sb = r(36, 12), r(8, 12)
    ops.gemm_f32_group([('nt', allf, Wu, U, bu, liveT, 0.0), ('nt', v, Wv, Vq, None, liveB, 0.0),
                        ('nn', dU, Wu, dX, None, liveT, 1.0), ('tn', dU, allf, gWu, None, liveT, 0.0),
                        ('nn', dVq, Wv, wide[:, :d], None, liveB, 1.0), ('tn', dVq, v, gWv, None, liveB, 0.0),
                        ('nt', sa, sb, small, None, None, 0.0)])
    ref = allf @ Wu.t() + bu
    ref[2777:] = 0
    close(U, ref, what='U', atol=5e-5)
    ref = v @ Wv.t()
    ref[500:] = 0
    close(Vq, ref, what='Vq', atol=5e-5)
    ref = dX0 + dU @ Wu
    ref[2777:] = 0
    close(dX, ref, what='dX', atol=5e-5)
    close(gWu, dU[:2777].t() @ allf[:2777], what='gWu', atol=2e-4)
    ref = wide0.clone()
    ref[:500, :d] += (dVq @ Wv)[:500]
    close(wide, ref, what='strided accumulate', atol=5e-5)
    close(gWv, dVq[:500].t() @ v[:500], what='gWv', atol=1e-4)
    close(small, sa @ sb.t(), what='small')


def test_gemm_f32_group_three_term_split(dev):
    """split3: the grouped products on the bf16 matrix pipe as hi hi + lo hi + hi lo of operands split in registers - against
    the fp64 product: every layout, dynamic row counts, beta, a k-split weight gradient; norm-wise within 2e-5 (2^-17 per
    operand pair, the lo lo term dropped) where the plain bf16 product is at 3e-3"""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(11)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    NT, B, d = 5000, 512, 256
    allf, Wu, bu, v, Wv, dU, dVq = r(NT, d), r(d, d), r(d), r(B, d), r(d, d), r(NT, d), r(B, d)
    allf[:, 3] *= 1e3                                            # a column 1000 x the others: the split is per element
    liveT = torch.tensor([4321], device=dev, dtype=torch.int32)
    liveB = torch.tensor([500], device=dev, dtype=torch.int32)
    U, dX0, gWu, gWv = torch.empty(NT, d, device=dev), r(NT, d), torch.empty(d, d, device=dev), torch.empty(d, d, device=dev)
    dX = dX0.clone()
    ops.gemm_f32_group([('nt', allf, Wu, U, bu, liveT, 0.0), ('nn', dU, Wu, dX, None, liveT, 1.0),
                        ('tn', dU, allf, gWu, None, liveT, 0.0), ('tn', dVq, v, gWv, None, liveB, 0.0)], split3=True)
    D = lambda t: t.double()

    def rel(a, b):
        return float((D(a) - b).norm() / b.norm())
    assert rel(U[:4321], D(allf[:4321]) @ D(Wu).t() + D(bu)) < 2e-5
    assert float(U[4321:].abs().max()) == 0.0
    assert rel(dX[:4321], D(dX0[:4321]) + D(dU[:4321]) @ D(Wu)) < 2e-5
    assert rel(gWu, D(dU[:4321]).t() @ D(allf[:4321])) < 2e-5
    assert rel(gWv, D(dVq[:500]).t() @ D(v[:500])) < 2e-5


@pytest.mark.parametrize('n_orders', [1, 2])
def test_readout_head_matches_unfused_ops(dev, n_orders):
    """ops.readout_head (grouped launches) against the same head assembled from ops.linear / seg_attn / cat_cols"""
    ops = _ops()
    torch.manual_seed(11)
    B, d = 96, 64
    lens = torch.randint(1, 9, (B,))
    seg = torch.zeros(B + 1, dtype=torch.int32)
    seg[1:] = lens.cumsum(0)
    NT = int(seg[-1]) + 13                                  # padded capacity behind the live nodes
    seg_d = seg.to(dev)
    dT = torch.tensor([int(seg[-1])], device=dev, dtype=torch.int32)
    dB = torch.tensor([B], device=dev, dtype=torch.int32)
    allf0 = torch.randn(NT, d, device=dev)
    allf0[int(seg[-1]):] = 0

    def params():
        return [t.requires_grad_() for t in (torch.randn(B, d, device=dev), torch.randn(d, d, device=dev) * 0.2,
                                             torch.randn(d, device=dev) * 0.2, torch.randn(d, d, device=dev) * 0.2,
                                             torch.randn(1, d, device=dev) * 0.2, torch.randn(d, 2 * d, device=dev) * 0.2)]
    per = [params() for _ in range(n_orders)]
    wts = [torch.randn(B, d, device=dev) for _ in range(n_orders)]

    def run(fused):
        allf = allf0.clone().requires_grad_()
        if fused:
            ss = ops.readout_head(allf, seg_d, dT, dB, per)
        else:
            ss = []
            for v, Wu, bu, Wv, we, Wsr in per:
                U = ops.linear(allf, Wu, bu, dT, exact=True)
                Vq = ops.linear(v, Wv, None, dB, exact=True)
                srg = ops.seg_attn(U, Vq, we, allf, seg_d, dB)
                ss.append(ops.linear(ops.cat_cols(v, srg), Wsr, None, dB, exact=True))
        loss = sum((s * w).sum() for s, w in zip(ss, wts))
        leaves = [allf] + [t for po in per for t in po]
        grads = torch.autograd.grad(loss, leaves)
        return [s.detach() for s in ss], grads
    s1, g1 = run(True)
    s0, g0 = run(False)
    for a, b in zip(s1, s0):
        close(a, b, what='s', atol=2e-5)
    names = ['allf'] + ['%s%d' % (nm, i) for i in range(n_orders) for nm in ('v', 'Wu', 'bu', 'Wv', 'we', 'Wsr')]
    for nm, a, b in zip(names, g1, g0):
        close(a, b, what='grad ' + nm, rtol=2e-4, atol=1e-4)


@pytest.mark.parametrize('n_orders,B,d,maxlen', [(1, 512, 256, 15), (1, 509, 256, 40), (3, 40, 128, 15), (2, 21, 256, 70)])
def test_fused_readout_head_matches_the_exact_fp32_grouped_head(dev, n_orders, B, d, maxlen):
    """csrc/headf.hip (ops.ReadoutHeadFused: Vq, U, soft-max read-out, fc_sr, F.normalize of a group of 8 sessions per
    workgroup in ONE launch, every product a 3-term hi / lo bf16 split) against the exact-fp32 grouped head
    (ops.ReadoutHead) followed by ops.normalize, msgifsr.py:124-155 + :269-273: normalised session vectors, the bf16
    operand copy, the saved soft-max weights, and every gradient (the backward shares the grouped launches, fed with the
    fused forward's saved tensors).  Padded layouts: live sessions / rows below capacity, B not a multiple of the group
    size, sessions longer than one 32-row chunk, sessions of a single node."""
    ops = _ops()
    torch.manual_seed(5)
    lens = torch.randint(1, maxlen, (B,))
    lens[::7] = 1
    seg = torch.zeros(B + 1, dtype=torch.int32)
    live_B = B - 3
    seg[1:] = lens.cumsum(0)
    seg[live_B + 1:] = seg[live_B]                          # capacity padding: empty sessions behind the live ones
    n_live = int(seg[live_B])
    NT = n_live + 77
    seg_d = seg.to(dev)
    dT = torch.tensor([n_live], device=dev, dtype=torch.int32)
    dB = torch.tensor([live_B], device=dev, dtype=torch.int32)
    allf0 = torch.randn(NT, d, device=dev)
    allf0 = allf0 / allf0.norm(dim=1, keepdim=True)
    allf0[n_live:] = 0
    sc = 1.0 / d ** 0.5
    v0 = []
    for _ in range(n_orders):
        vv = torch.randn(B, d, device=dev)
        vv = vv / vv.norm(dim=1, keepdim=True)
        vv[live_B:] = 0
        v0.append(vv)
    par = [[((torch.rand(d, d, device=dev) * 2 - 1) * sc), ((torch.rand(d, device=dev) * 2 - 1) * sc),
            ((torch.rand(d, d, device=dev) * 2 - 1) * sc), ((torch.rand(1, d, device=dev) * 2 - 1) * sc),
            ((torch.rand(d, 2 * d, device=dev) * 2 - 1) * sc)] for _ in range(n_orders)]
    gws = [torch.randn(B, d, device=dev) for _ in range(n_orders)]
    for w in gws:
        w[live_B:] = 0

    class WS:
        sr16 = None
        sr_fresh = None

    def run(fused):
        allf = allf0.clone().requires_grad_()
        vs, per = [], []
        for i in range(n_orders):
            buf = torch.full((B, 2 * d), 7.0, device=dev)      # the right half is overwritten by the head
            buf[:, :d] = v0[i]
            vleaf = buf.requires_grad_()
            v = vleaf[:, :d]
            v._srec_cat_left = True
            vs.append(vleaf)
            per.append([v] + [t.clone().requires_grad_() for t in par[i]])
        ws = WS()
        ops.set_precision('bf16' if fused else 'fp32')
        try:
            if fused:
                ws.sr16 = torch.zeros(B + 5, d, device=dev, dtype=torch.bfloat16)
                assert ops.readout_head_fused_ok(allf, per)
                ys = ops.readout_head_fused(allf, seg_d, dT, dB, per, ws if n_orders == 1 else None)
            else:
                ss = ops.ReadoutHead.apply(allf, seg_d, dT, dB, *[t for po in per for t in po])
                ys = [ops.normalize(s_, 0, dB) for s_ in ss]
            loss = sum((y * w).sum() for y, w in zip(ys, gws))
            leaves = [allf] + [t for i in range(n_orders) for t in [vs[i]] + per[i][1:]]
            grads = torch.autograd.grad(loss, leaves)
        finally:
            ops.set_precision('fp32')
        return [y.detach() for y in ys], grads, ws

    y1, g1, ws1 = run(True)
    y0, g0, _ = run(False)

    def near(a, b, what, tol=1e-4, rtol=3e-5):
        a, b = a.double().cpu(), b.double().cpu()
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= tol * max(scale, 1e-30), '%s: max |err| %.3e against scale %.3e' % (what, err, scale)
        rel = float((a - b).norm() / b.norm().clamp(min=1e-30))
        assert rel < rtol, '%s: relative error %.3e' % (what, rel)
    for i, (a, b) in enumerate(zip(y1, y0)):
        near(a[:live_B], b[:live_B], 'y%d' % i)
        assert float(a[live_B:].abs().max()) == 0.0            # capacity padding: zero rows
    if n_orders == 1:
        assert ws1.sr_fresh == (y1[0].data_ptr(), B, d)
        assert torch.equal(ws1.sr16[:B].float(), y1[0].to(torch.bfloat16).float())      # the scoring operand = bf16(y)
    names = ['allf'] + ['%s%d' % (nm, i) for i in range(n_orders) for nm in ('cat', 'Wu', 'bu', 'Wv', 'we', 'Wsr')]
    for nm, a, b in zip(names, g1, g0):
        if nm.startswith('cat'):
            a, b = a[:live_B, :d], b[:live_B, :d]              # d v (the right half of the buffer is not an input)
        if nm.startswith(('bu', 'we')):                        # column sums that cancel to ~1e-3 of their summands
            near(a, b, 'grad ' + nm, tol=2e-3, rtol=5e-4)
        elif nm.startswith('W'):
            near(a, b, 'grad ' + nm, tol=5e-4, rtol=3e-4)
        else:
            near(a, b, 'grad ' + nm)


@pytest.mark.parametrize('d,Dp,max_norm', [(256, 256, 1.0), (100, 128, 1.0), (64, 64, 0.0), (516, 516, 2.0)])
def test_renorm_rows_bf16_one_pass(dev, d, Dp, max_norm):
    """Embedding(max_norm) renorm (lessr.py:126 / msgifsr.py:162) + the bf16 operand copy of the table in one pass"""
    L = importlib.import_module('sessionrec-pytorch_amd._lib')
    torch.manual_seed(d)
    V = 777
    W = torch.randn(V, d, device=dev) * (2.0 / d ** 0.5)          # norms scattered around 2: some above, some below
    W[5] *= 0.01
    ref = W.clone()
    if max_norm > 0:
        n = ref.norm(dim=1, keepdim=True)
        ref = torch.where(n > max_norm, ref * (max_norm / (n + 1e-7)), ref)
    out16 = torch.full((V + 3, Dp), 7.0, device=dev, dtype=torch.bfloat16)
    L.lib.srec_renorm_rows_bf16(L.ptr(W), W.stride(0), V, d, max_norm, L.ptr(out16), Dp, L.stream())
    close(W, ref, what='renormed rows', rtol=1e-6, atol=1e-7)
    assert torch.equal(out16[:V, :d], W.bfloat16()), 'bf16 copy is the RNE rounding of the renormed rows'
    assert float(out16[:V, d:].abs().max() if Dp > d else 0.0) == 0.0
    assert float((out16[V:] - 7.0).abs().max()) == 0.0             # rows past n untouched


@pytest.mark.parametrize('n', [1, 3, 4, 1027, 300001])
def test_copy_words_reads_pinned_host_memory(dev, n):
    """srec_copy_words: the batch intake of a replayed step - a kernel that loads page-locked HOST words (rowops.hip)"""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    src = torch.randint(-2 ** 31, 2 ** 31 - 1, (n + 8,), dtype=torch.int32, generator=g).pin_memory()
    dst = torch.full((n + 8,), 77, dtype=torch.int32, device=dev)
    ops.copy_words(src, dst, n)
    torch.cuda.synchronize()
    assert torch.equal(dst[:n].cpu(), src[:n]) and bool((dst[n:] == 77).all())
    dst2 = torch.zeros(n + 4, dtype=torch.int32, device=dev)
    ops.copy_words(dst, dst2, n)                                  # device source
    assert torch.equal(dst2[:n].cpu(), src[:n]) and bool((dst2[n:] == 0).all())
    with pytest.raises(AssertionError):
        ops.copy_words(torch.zeros(8, dtype=torch.int32), dst, 8)  # pageable host memory: refused, a kernel cannot read it


def test_gemm_f32_group_split_k_with_bias_beta_and_dynamic_rows(dev):
    """srec_gemm_f32_group_run: long-K problems are k-split into slabs and summed in slab order by the reduce launch -
    repeatable bit for bit, with bias / beta / dynamic row counts next to unsplit problems in one group"""
    ops = _ops()
    g = torch.Generator(device='cpu').manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    NT, B, d = 7000, 512, 256
    allf, dU, v, dVq, cat, Wsr, bias = r(NT, d), r(NT, d), r(B, d), r(B, d), r(B, 2 * d), r(d, 2 * d), r(d)
    liveT = torch.tensor([6543], device=dev, dtype=torch.int32)
    liveB = torch.tensor([501], device=dev, dtype=torch.int32)
    acc0 = r(B, d)

    def run():
        gWu, gWv, out = torch.empty(d, d, device=dev), torch.empty(d, d, device=dev), torch.full((B, d), 5.0, device=dev)
        acc = acc0.clone()
        ops.gemm_f32_group([('tn', dU, allf, gWu, None, liveT, 0.0), ('tn', dVq, v, gWv, None, liveB, 0.0),
                            ('nt', cat, Wsr, out, bias, liveB, 0.0), ('nt', cat, Wsr, acc, None, liveB, 1.0)])
        return gWu, gWv, out, acc
    ref = run()
    for rep in range(3):
        got = run()
        for a, b, nm in zip(got, ref, ('gWu', 'gWv', 'out', 'acc')):
            assert torch.equal(a, b), (rep, nm, (a - b).abs().max().item())
    close(ref[0], dU[:6543].t() @ allf[:6543], what='gWu', atol=5e-4)
    o = cat @ Wsr.t() + bias
    o[501:] = 0
    close(ref[2], o, what='out', atol=1e-4)


def test_deferred_sums_of_an_aborted_backward_are_forgotten(dev):
    """a backward pass that raises leaves its deferred slab sums registered; the next training forward drops them instead of
    summing freed buffers into freed gradients at the end of the next backward (and registers its own callback again)"""
    ops = _ops()
    part = torch.ones(4, 8, device=dev)
    out = torch.zeros(8, device=dev)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            ops.defer_slab_sum(part, out, True)
            raise RuntimeError('boom')

    x = torch.ones(3, device=dev, requires_grad=True)
    with pytest.raises(RuntimeError, match='boom'):
        Boom.apply(x).sum().backward()
    assert len(ops._DEFERRED) == 1
    ops.drop_stale_deferred()
    assert len(ops._DEFERRED) == 0 and float(out.sum()) == 0.0

    class Fine(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            ops.defer_slab_sum(part, out, True)
            return g

    Fine.apply(x).sum().backward()
    torch.cuda.synchronize()
    assert len(ops._DEFERRED) == 0 and float(out.sum()) == 32.0


@pytest.mark.parametrize('w,ucap,nloc,proj', [(2, 256, 700, False), (8, 512, 4332, True), (5, 300, 97, True)])
def test_add_rows_of_all_ranks_in_one_launch_equals_the_rank_by_rank_loop(dev, w, ucap, nloc, proj):
    """round 6, row-sharded lookup backward (the nn.Embedding gradient of srgnn.py:133 / msgifsr.py:247 with the table sharded over
    the ranks): srec_add_rows_ranks adds the per-item gradient rows of ALL ranks' request lists - the same item sits in several
    lists - per item in rank order.  BIT-identical to w rank-by-rank srec_scatter_add_sorted(_ex) launches, dense gradient and
    the radial side sums of the deferred row-normalisation projection alike."""
    D = importlib.import_module('sessionrec-pytorch_amd.dist')
    local = D.HipLocal()
    g = torch.Generator().manual_seed(w * 1000 + ucap)
    d, lo, V = 256, 3 * nloc, 8 * nloc
    ids = torch.full((w, ucap), -1, dtype=torch.int32)
    for r in range(w):
        # Zipf-ish: hot items land in every rank's list; ascending, -1 padding behind (collate: np.unique + caps)
        n = int(torch.randint(ucap // 2, ucap + 1, (1,), generator=g))
        hot = torch.arange(lo, lo + min(40, nloc))
        rest = torch.randperm(V, generator=g)[:n]
        u = torch.unique(torch.cat([hot, rest]))[:n]
        ids[r, :u.numel()] = u.to(torch.int32)
    ids = ids.reshape(-1).to(dev)
    rows = torch.randn(w * ucap, d, generator=g).to(dev)
    rel = local.localize(ids, lo, nloc)
    assert int((rel >= 0).sum()) > 40 * w // 2
    base = torch.randn(nloc, d, generator=g).to(dev)
    W = torch.randn(nloc, d, generator=g).to(dev) if proj else None
    rad0 = torch.randn(nloc, generator=g).to(dev) if proj else None
    # reference: the loop the launch replaces
    dst_a, rad_a = base.clone(), (rad0.clone() if proj else None)
    for r in range(w):
        local.add_rows(rows[r * ucap:(r + 1) * ucap], rel[r * ucap:(r + 1) * ucap], dst_a, (W, rad_a) if proj else None)
    dst_b, rad_b = base.clone(), (rad0.clone() if proj else None)
    assert local.add_rows_all(rows, rel, ids, w, ucap, dst_b, (W, rad_b) if proj else None)
    torch.cuda.synchronize()
    assert torch.equal(dst_a, dst_b), (dst_a - dst_b).abs().max().item()
    if proj:
        assert torch.equal(rad_a, rad_b), (rad_a - rad_b).abs().max().item()
    touched = torch.unique(rel[rel >= 0]).long()
    untouched = torch.ones(nloc, dtype=torch.bool, device=dev)
    untouched[touched] = False
    assert torch.equal(dst_b[untouched], base[untouched])


@pytest.mark.parametrize('w,B', [(1, 37), (2, 512), (8, 4096), (11, 1500)])
def test_merge_stats_of_the_shards(dev, w, B):
    """global (lse, label logit, mean loss, per-session weights) from the shards' partial statistics (dist._merge_stats ->
    srec_merge_stats32; train.py:99 nll_loss(mean) over the row-sharded catalog): against torch.logsumexp, padding sessions
    (label < 0) left out of the mean.  8 x 4 096 is what a rank of 8 merges per weak-scaling step (1 024-thread form)."""
    D = importlib.import_module('sessionrec-pytorch_amd.dist')
    local = D.HipLocal()
    g = torch.Generator().manual_seed(w * 31 + B)
    st = torch.randn(w, 2, B, generator=g) * 3.0
    lab = torch.randint(0, 1000, (B,), generator=g).to(torch.int32)
    lab[torch.rand(B, generator=g) < 0.1] = -1
    lse, lb, loss, gw = local.merge_stats(st.to(dev), lab.to(dev))
    ref_lse = torch.logsumexp(st[:, 0].double(), dim=0)
    ref_lab = st[:, 1].double().sum(0)
    live = (lab >= 0)
    n = int(live.sum())
    close(lse, ref_lse.float(), rtol=1e-6, atol=1e-5, what='lse')
    close(lb, ref_lab.float(), rtol=1e-6, atol=1e-5, what='label logit')
    ref_loss = float(((ref_lse - ref_lab) * live).sum() / max(n, 1))
    assert abs(float(loss) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (float(loss), ref_loss)
    assert torch.equal(gw.cpu() > 0, live) and abs(float(gw.sum()) - (1.0 if n else 0.0)) < 1e-5


def test_step_prep_roles_equal_the_single_launches(dev):
    """srec_step_prep (csrc/prep.hip): the bf16 / transposed copies of fc weights, both fragment layouts of GRU weights and the
    hi / lo fragment copies of the head's weights as workgroup ranges of ONE launch - bit-identical to srec_weights_bf16,
    srec_gru_wfrag_both and srec_head_wfrag run one by one (the same role code behind all of them); the bf16 copy itself
    against torch's round-to-nearest-even, the transposed copy against its transpose, odd shapes included."""
    ops = _ops()
    torch.manual_seed(3)
    w16 = [torch.randn(512, 64, device=dev), torch.randn(1024, 128, device=dev), torch.randn(70, 33, device=dev)]
    gru = [torch.randn(384, 128, device=dev) for _ in range(4)]
    head = [(torch.randn(128, 128, device=dev), 0), (torch.randn(128, 256, device=dev), 0), (torch.randn(128, 256, device=dev), 1)]
    ops.weights_changed()
    a16, t16 = ops.weights_bf16(w16)
    gf, gb = ops.gru_wfrag_both(gru)
    hf = ops.head_wfrag([w for w, _ in head], [t for _, t in head])
    for w, a, t in zip(w16, a16, t16):
        assert torch.equal(a, w.to(torch.bfloat16)) and torch.equal(t, a.t().contiguous())
    ref = [x.clone() for x in a16 + t16 + gf + gb + hf]
    ops.weights_changed()
    ops.step_prologue(w16, gru, head)
    b16, u16 = ops.weights_bf16(w16)                      # taken from the prologue's copies: no launch
    hf2 = ops.head_wfrag([w for w, _ in head], [t for _, t in head])
    gf2, gb2 = ops.gru_wfrag_both(gru)
    assert not ops._WPREP
    for x, y in zip(ref, b16 + u16 + gf2 + gb2 + hf2):
        assert x.data_ptr() != y.data_ptr() and torch.equal(x, y)
    # an in-place update of a weight between the prologue and its reader: the copy is not used
    ops.step_prologue(w16[:1])
    w16[0].mul_(2.0)
    c16, _ = ops.weights_bf16(w16[:1])
    assert torch.equal(c16[0], w16[0].to(torch.bfloat16))
    ops.weights_changed()


def test_gru_mixed_launch_plan(dev):
    """srec_gru_fused_wide: which k-gram problems of a fused GRU launch take 32-node workgroups so that the launch is one round
    of the chip (at most one workgroup per CU) - the shortest first and never the longest order, none while the 16-node tiles fit or
    when an order exceeds the 16-node kernels' 4 time steps; the benchmarked capacities (2560 nodes of order 3 and of order 2) widen order 2 only."""
    import ctypes as ct
    L = importlib.import_module('sessionrec-pytorch_amd._lib')
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def wide(ns, ks):
        m, a_n, a_k = ct.c_int(-1), (ct.c_int * len(ns))(*ns), (ct.c_int * len(ks))(*ks)      # (kept alive across the call)
        L.lib.srec_gru_fused_wide(len(ns), ct.addressof(a_n), ct.addressof(a_k), ct.addressof(m))
        return m.value

    assert wide([2560, 2560], [3, 2]) == 0b10                 # 160 + 160 tiles -> 160 + 80
    assert wide([1000, 1000], [3, 2]) == 0                    # 63 + 63 tiles fit
    assert wide([16 * cus, 16], [3, 2]) == 0                  # the short problem cannot shrink (one tile); the longest order never widens
    assert wide([4000, 4000, 4000], [4, 3, 2]) == 0b110       # 750 tiles: every problem but the longest, still more than one round
    assert wide([3072, 3072], [3, 2]) == 0b10                 # the end-to-end loop's capacities: 192 + 96 by capacity, ~200 live
    assert wide([2560, 2560], [5, 2]) == 0                    # order 5: not a launch of the 16-node kernels
