"""Parity cases the round-2 review listed as untested: sessions of up to 50 clicks (config C5's encoder side), the
kernels' static per-session budgets (a clean error, never a truncated soft-max), batches so small that relations /
GAT modules are missing, a deferred table-gradient projection that no optimizer step consumed, and dropout masks that
must renew without a FusedAdam.  Product (HIP, through the C ABI) vs the CPU oracle (oracle/models_ref.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from util import close, pkg

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _pair(kind, V, d, K=3, **kw):
    """(product model, oracle model, product collate, oracle collate) with equal weights"""
    from oracle import collate_ref as oc, models_ref as om
    sp, c = pkg(), pkg('collate')
    torch.manual_seed(7)
    if kind == 'msgifsr':
        ref = om.MSGIFSR(V, 'x', d, 1, order=K, extra=False, fusion=False, **kw)
        model = sp.MSGIFSR(V, 'x', d, 1, order=K, extra=False, fusion=False, **kw)
        fn, ofn = c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), K), oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), K)
    elif kind == 'lessr':
        ref, model = om.LESSR(V, d, 2), sp.LESSR(V, d, 2)
        fn = c.collate_fn_factory(c.seq_to_eop_multigraph, c.seq_to_shortcut_graph)
        ofn = oc.collate_fn_factory(oc.seq_to_eop_multigraph, oc.seq_to_shortcut_graph)
    else:
        cls, rcls = (sp.NISER, om.NISER) if kind == 'niser' else (sp.SRGNN, om.SRGNN)
        ref, model = rcls(V, d, 1), cls(V, d, 1, use_gnn_output=False)
        fn, ofn = c.collate_fn_factory(c.seq_to_session_graph), oc.collate_fn_factory(oc.seq_to_session_graph)
    model.load_state_dict(ref.state_dict())
    return model, ref, fn, ofn


def _step_vs_oracle(dev, model, ref, fn, ofn, samples, tag):
    from oracle import models_ref as om
    model = model.to(dev).train()
    ref.train()
    inputs, labels = fn(samples)
    inputs, labels = [x.to(dev) for x in inputs], labels.to(dev)
    oin, olab = ofn(samples)
    oin, olab = [om.to_torch(x) for x in oin], torch.from_numpy(olab)
    loss = model.fused_loss(*inputs, labels)
    loss.backward()
    rl = torch.nn.functional.nll_loss(ref(*oin), olab)
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 1e-5 * max(1.0, abs(rl.item())), (tag, loss.item(), rl.item())
    rp = dict(ref.named_parameters())
    tname = [k for k, p in model.named_parameters() if p is model._table()][0]
    close(model.table_grad.buf, rp[tname].grad, rtol=1e-4, atol=1e-7, what=tag + ': table gradient')
    for k, p in model.named_parameters():
        if k == tname or rp[k].grad is None:
            continue
        if p.grad is None:
            assert float(rp[k].grad.abs().max()) == 0.0, tag + ': %s has no gradient on the HIP path' % k
            continue
        close(p.grad, rp[k].grad, rtol=1e-4, atol=1e-7, what=tag + ': grad ' + k)
    model.eval()
    ref.eval()
    with torch.no_grad():
        close(model(*inputs)[:len(olab)], ref(*oin), rtol=1e-4, atol=1e-4, what=tag + ': log-probs')


def _long_sessions(n, V, lo, hi, seed, hub=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        # a small item pool per session: revisits, repeated 2- and 3-grams, nodes of high degree
        pool = rng.integers(0, V, size=max(3, int(L * 0.7)))
        seq = pool[rng.integers(0, len(pool), size=L)].tolist()
        out.append((seq, int(rng.integers(0, V))))
    return out


@pytest.mark.parametrize('kind', ['msgifsr', 'srgnn', 'niser', 'lessr'])
def test_sessions_of_up_to_50_clicks_match_the_oracle(dev, kind):
    """config C5 names sessions of <= 50 clicks; every fixture and bench batch stops at 20 (collate.py:87-217 builds any
    length, preprocess.py:45-50 truncates the public datasets to 20)"""
    V, d = 400, 32
    model, ref, fn, ofn = _pair(kind, V, d)
    samples = _long_sessions(12, V, 35, 50, 3) + [([5], 9), ([7, 7, 7, 7], 1)]
    inputs, _ = fn(samples)
    assert inputs[0].meta['max_nodes'] >= 20 and inputs[0].meta['max_deg'] >= 2, inputs[0].meta
    _step_vs_oracle(dev, model, ref, fn, ofn, samples, kind + ' len<=50')


@pytest.mark.parametrize('d', [36, 40, 48])
def test_msgifsr_widths_off_the_kernel_fast_paths_match_the_oracle(dev, d):
    """embedding widths that leave the fast paths of the MSHGNN layer kernels: d % 16 != 0 exercises the zero-padded k tail
    of the matrix-pipe attention-logit product (hg_dots), d % 8 != 0 (36) falls back from the wavefront-per-node kernels
    (8-column lanes) to the workgroup-per-node ones; fp32 mode, product vs the CPU oracle (msgifsr.py:70-91,
    gatconv.py:267-311)."""
    V = 300
    model, ref, fn, ofn = _pair('msgifsr', V, d)
    samples = _long_sessions(24, V, 2, 14, 11) + [([5], 9), ([7, 7, 7, 7], 1)]
    _step_vs_oracle(dev, model, ref, fn, ofn, samples, 'msgifsr d=%d' % d)


def test_oversized_sessions_are_refused_not_truncated(dev):
    """the per-session kernels keep a session's nodes / a node's edge list in fixed LDS arrays (srec_limits); a batch beyond
    them must raise on the host before any kernel runs"""
    ops = pkg('ops')
    L = ops.limits()
    assert L['nodes'] >= 147 and L['deg'] >= 50, L            # C5: 50 + 49 + 48 nodes, degree <= 50
    V, d = 2000, 32
    # (a) MSGIFSR order 3, 100 distinct clicks: 100 + 99 + 98 read-out nodes of one session
    model, ref, fn, ofn = _pair('msgifsr', V, d)
    model = model.to(dev).train()
    seq = list(range(100))
    (mg,), labels = fn([(seq, 3), ([1, 2, 3], 4)])
    assert mg.meta['max_nodes'] == 297 > L['nodes']
    with pytest.raises(ValueError, match='read-out nodes'):
        model.fused_loss(mg.to(dev), labels.to(dev))
    with pytest.raises(ValueError, match='read-out nodes'):
        model(mg.to(dev))
    # (b) a hub item with more distinct predecessors than a GAT workgroup's edge list holds
    hub, n = 1999, L['deg'] + 2
    seq = []
    for i in range(n):
        seq += [i, hub]
    model, ref, fn, ofn = _pair('niser', V, d)
    model = model.to(dev).train()
    (mg,), labels = fn([(seq, 3)])
    assert mg.meta['max_deg'] == n and mg.meta['max_nodes'] <= L['nodes']
    with pytest.raises(ValueError, match='degree'):
        model.fused_loss(mg.to(dev), labels.to(dev))
    # the same sessions inside the budgets run (and match the oracle)
    model, ref, fn, ofn = _pair('msgifsr', V, d)
    _step_vs_oracle(dev, model, ref, fn, ofn, [(list(range(80)), 3), ([1, 2, 3], 4)], 'msgifsr 80 clicks')


@pytest.mark.parametrize('lens', [(1, 1, 1), (2, 2, 1, 2), (2, 3, 1), (1, 4)])
def test_msgifsr_batches_with_missing_relations(dev, lens):
    """a rank's share of a small last batch can consist of sessions of one or two clicks: no 3-grams, relations without
    edges, fewer live GAT modules than node types (HeteroGraphConv skips edgeless relations, msgifsr.py:74-82; sessions
    shorter than the order get a dummy node, collate.py:191-211)"""
    V, d = 300, 32
    rng = np.random.default_rng(sum(lens))
    samples = [(rng.integers(0, V, size=L).tolist(), int(rng.integers(0, V))) for L in lens]
    model, ref, fn, ofn = _pair('msgifsr', V, d)
    _step_vs_oracle(dev, model, ref, fn, ofn, samples, 'msgifsr lens %r' % (lens,))


@pytest.mark.parametrize('kind', ['niser', 'msgifsr'])
def test_backward_without_a_step_leaves_nothing_behind(dev, kind):
    """FusedAdam(fuse_projection=True) defers the chain rule of the catalog-row normalisation to its row pass and the lookup
    backward records radial side sums for it.  A backward that is NOT followed by a step (a skipped / NaN-guarded step, a
    training loss evaluated for monitoring) must not leak its pending projection or radial sums into the next step."""
    import copy
    train, optim = pkg('train'), pkg('optim')
    V, d = 300, 32
    base, ref, fn, ofn = _pair(kind, V, d)
    base = base.to(dev)
    samples = _long_sessions(10, V, 3, 12, 5)
    other = _long_sessions(10, V, 3, 12, 6)
    a_in, a_lab = fn(samples)
    b_in, b_lab = fn(other)
    a_in, a_lab, b_in, b_lab = [x.to(dev) for x in a_in], a_lab.to(dev), [x.to(dev) for x in b_in], b_lab.to(dev)
    runs = []
    for stale in (False, True, 'nozero'):
        m = copy.deepcopy(base).train()
        opt = optim.FusedAdam(train.fix_weight_decay(m), lr=1e-2, weight_decay=1e-4, model=m, fuse_projection=True)
        assert m.table_grad.defer
        if stale:
            opt.zero_grad()
            m.fused_loss(*b_in, b_lab).backward()          # a backward on another batch whose step never happens
            if stale == 'nozero':
                for p in m.parameters():                    # (encoder gradients would accumulate: cleared by hand; the table
                    p.grad = None                           #  gradient buffer is overwritten by the next scoring backward)
        if stale != 'nozero':
            opt.zero_grad()
        loss = m.fused_loss(*a_in, a_lab)
        loss.backward()
        opt.step()
        runs.append((loss.item(), m._table().detach().clone()))
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        close(r[1], runs[0][1], rtol=1e-6, atol=1e-8, what='table after a step that follows an unstepped backward')


def test_dropout_masks_renew_without_a_fused_optimizer(dev):
    """the masks are hash(nonce, device step counter, call site, element): the per-call nonce (a private CPU generator seeded from torch.initial_seed())
    renews them on every eager forward - with torch.optim.Adam, with no optimizer at all, for several micro-batches per
    optimizer step - and follows torch.manual_seed; two FusedAdam-driven models in one process keep separate counters"""
    sp, ops, c, train, optim = pkg(), pkg('ops'), pkg('collate'), pkg('train'), pkg('optim')
    V, d = 300, 32
    torch.manual_seed(1)
    m = sp.MSGIFSR(V, 'x', d, 1, dropout=0.5, order=2, extra=False, fusion=False).to(dev).train()
    fn = c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), 2)
    (mg,), labels = fn(_long_sessions(16, V, 3, 9, 2))
    mg, labels = mg.to(dev), labels.to(dev)
    ops.RNG_COUNTER.clear()
    losses = [m.fused_loss(mg, labels).item() for _ in range(4)]
    assert len(set(losses)) == 4, losses                    # no optimizer anywhere: four forwards, four masks
    torch.manual_seed(99)                                   # another seed value: the nonce stream restarts from it
    a = m.fused_loss(mg, labels).item()
    torch.manual_seed(98)
    c_ = m.fused_loss(mg, labels).item()
    torch.manual_seed(99)
    b = m.fused_loss(mg, labels).item()
    assert a == b and a != c_                               # ... reproducible from torch.manual_seed
    st = torch.get_rng_state()
    m.fused_loss(mg, labels)
    assert torch.equal(st, torch.get_rng_state())           # ... without ever advancing torch's global CPU stream
    ops.seed_dropout()                                      # (an equal seed value is not observable: explicit restart)
    assert m.fused_loss(mg, labels).item() == a
    # two models, each with its own FusedAdam: a model's forward installs ITS optimizer's device counter
    m2 = sp.MSGIFSR(V, 'x', d, 1, dropout=0.5, order=2, extra=False, fusion=False).to(dev).train()
    o1 = optim.FusedAdam(train.fix_weight_decay(m), lr=1e-3, model=m)
    o2 = optim.FusedAdam(train.fix_weight_decay(m2), lr=1e-3, model=m2)
    for o, mm in ((o1, m), (o2, m2), (o1, m)):
        o.zero_grad()
        mm.fused_loss(mg, labels).backward()
        o.step()
    c1, c2 = m.__dict__['_srec_rng_counter'], m2.__dict__['_srec_rng_counter']
    assert c1.data_ptr() != c2.data_ptr() and int(c1.item()) == 2 and int(c2.item()) == 1
    m2.fused_loss(mg, labels)
    assert ops.RNG_COUNTER[str(dev)].data_ptr() == c2.data_ptr()
    m.fused_loss(mg, labels)
    assert ops.RNG_COUNTER[str(dev)].data_ptr() == c1.data_ptr()
    # a captured step needs the device counter: without one the capture is refused instead of replaying one mask for ever
    m3 = sp.MSGIFSR(V, 'x', d, 1, dropout=0.5, order=2, extra=False, fusion=False).to(dev).train()
    g = torch.cuda.CUDAGraph()
    m3.fused_loss(mg, labels)                               # warm-up (lazy allocations) outside the capture
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='device-side step counter'):
        with torch.cuda.graph(g):
            m3.fused_loss(mg, labels)


@pytest.mark.parametrize('kind', ['srgnn', 'niser'])
def test_c2_model_step_matches_the_oracle(dev, kind):
    """config C2 (BASELINE.json configs[1]) as a MODEL step: SRGNN - and NISER, the same encoder with cosine scoring - at
    d = 96, V = 43 097, 512 synthetic Diginetica-shaped sessions, fp32, product vs the CPU oracle: loss, every gradient
    incl. all 43 097 rows of the table gradient, (512, 43 097) log-probabilities.  The other tests meet the oracle at
    d = 32 / 36 / 40 / 48 / 64 / 128 / 256; 96 is the only width C2 names (srgnn.py:131-148, niser.py:130-157)."""
    from dist_gpu_worker import synth_samples
    V, d = 43097, 96
    model, ref, fn, ofn = _pair(kind, V, d)
    samples = synth_samples(512, V, 123, max_len=20, mean_len=5.0)
    _step_vs_oracle(dev, model, ref, fn, ofn, samples, kind + ' C2')


def test_two_backwards_before_one_step_accumulate_like_the_undeferred_path(dev):
    """ops.defer_slab_sum hands autograd a weight-gradient tensor whose slab sum is launched at the END of the backward
    pass - safe only when AccumulateGrad takes the buffer over (p.grad is None).  A second backward before the step (two
    micro-batches per optimizer step) must ADD complete gradients: the deferral is withdrawn for parameters that already
    hold one (ops.can_defer), and the permission is scoped to the forward that asked for it."""
    sp, ops, c = pkg(), pkg('ops'), pkg('collate')
    V, d = 400, 128
    fn = c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), 3)
    b1, b2 = (fn(_long_sessions(24, V, 4, 12, s)) for s in (11, 12))
    ops.set_precision('bf16')                      # the deferred sums belong to the bf16-mode gemm16 weight gradients
    try:
        torch.manual_seed(3)
        m = sp.MSGIFSR(V, 'x', d, 1, order=3, extra=False, fusion=False).to(dev).train()
        grads = []
        for mode in ('separate', 'accumulated'):
            per = []
            for (mg,), lab in (b1, b2):
                if mode == 'separate' or not per:
                    m.zero_grad(set_to_none=True)
                loss = m.fused_loss(mg.to(dev), lab.to(dev))
                assert not ops.DEFER['on']             # withdrawn when the forward returns
                loss.backward()
                assert not ops._DEFERRED               # nothing left pending after backward()
                per.append({k: p.grad.detach().clone() for k, p in m.named_parameters()
                            if p.grad is not None and p is not m._table()})
            grads.append(per)
        sep, acc = grads
        watched = [k for k in sep[0] if 'fc.weight' in k or 'weight_ih' in k or 'weight_hh' in k]
        assert len(watched) >= 10
        for k in sep[0]:
            want = sep[0][k] + sep[1][k]
            close(acc[1][k], want, rtol=1e-5, atol=1e-6 * float(want.abs().max()), what='accumulated gradient of ' + k)
        # a parameter with a tensor hook never gets a deferred gradient either: the hook sees the finished sum
        seen = {}
        w = m.layers[0].conv1.mods['inter'].fc.weight
        h = w.register_hook(lambda g: seen.setdefault('g', g.detach().clone()))
        m.zero_grad(set_to_none=True)
        (mg,), lab = b1
        m.fused_loss(mg.to(dev), lab.to(dev)).backward()
        h.remove()
        torch.cuda.synchronize()
        close(seen['g'], sep[0]['layers.0.conv1.mods.inter.fc.weight'], rtol=1e-6, atol=1e-9, what='gradient seen by a tensor hook')
    finally:
        ops.set_precision('fp32')


def test_a_replayed_step_refuses_an_oversized_session(dev):
    """the per-session budgets (SREC_MAX_SESSION_NODES / degree) are checked on the host for EVERY replayed batch, not only
    during warm-up and capture: the kernels clamp, so an oversized session in a later batch would otherwise give a silently
    truncated soft-max (graph.GraphedTrainStep.__call__ -> ops.check_limits)"""
    sp, ops, c, train, optim, G = pkg(), pkg('ops'), pkg('collate'), pkg('train'), pkg('optim'), pkg('graph')
    V = 400
    torch.manual_seed(2)
    m = sp.NISER(V, 32, 1).to(dev).train()
    opt = optim.FusedAdam(train.fix_weight_decay(m), lr=1e-3, weight_decay=1e-4, model=m)
    caps = dict(B=4, N=1024, E=1024, U=1024)
    fn = c.collate_fn_factory(c.seq_to_session_graph, caps=caps)
    ok = fn(_long_sessions(4, V, 5, 12, 1))
    gs = G.GraphedTrainStep(m, opt, [x.to(dev) for x in ok[0]], ok[1].to(dev))
    gs(ok[0], ok[1])
    big = fn([(list(range(ops.limits()['nodes'] + 3)), 5)] + _long_sessions(3, V, 5, 12, 2))
    assert big[0][0].meta['padded'] and big[0][0].meta['max_nodes'] > ops.limits()['nodes']
    with pytest.raises(ValueError, match='read-out nodes'):
        gs(big[0], big[1])
    gs(ok[0], ok[1])                                   # the captured step is still usable
    torch.cuda.synchronize()
