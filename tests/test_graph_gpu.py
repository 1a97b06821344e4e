"""Capacity-padded batches and whole-step hipGraph replay (GPU): same numbers as the exact eager path."""
import copy

import numpy as np
import pytest
import torch

from util import close, pkg

pytestmark = pytest.mark.gpu


def _samples(rng, n, V, max_len=12):
    out = []
    for _ in range(n):
        L = int(rng.integers(1, max_len))
        seq = rng.integers(0, V, size=L).tolist()
        if L > 1 and rng.random() < 0.3:
            seq[1] = seq[0]
        out.append((seq, int(rng.integers(0, V))))
    return out


def _setup(kind, dev, V=400, d=32):
    sp, c = pkg(), pkg('collate')
    torch.manual_seed(1)
    if kind == 'msgifsr':
        model = sp.MSGIFSR(V, 'x', d, 1, order=3, extra=False, fusion=False).to(dev)
        mk = lambda caps: c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), 3, caps=caps)
    elif kind == 'niser':
        model = sp.NISER(V, d, 1).to(dev)
        mk = lambda caps: c.collate_fn_factory(c.seq_to_session_graph, caps=caps)
    elif kind == 'lessr':
        model = sp.LESSR(V, d, 3).to(dev)
        mk = lambda caps: c.collate_fn_factory(c.seq_to_eop_multigraph, c.seq_to_shortcut_graph, caps=caps)
    else:
        model = sp.SRGNN(V, d, 1).to(dev)
        mk = lambda caps: c.collate_fn_factory(c.seq_to_session_graph, caps=caps)
    return model, mk


@pytest.mark.parametrize('kind', ['srgnn', 'niser', 'msgifsr', 'lessr'])
def test_padded_layout_equals_exact(dev, kind):
    c = pkg('collate')
    rng = np.random.default_rng(2)
    V = 400
    model, mk = _setup(kind, dev, V)
    m2 = copy.deepcopy(model)
    samples = _samples(rng, 24, V)
    caps = c.default_caps(32, 12)
    if kind == 'lessr':                       # shortcut graphs have up to L(L+1)/2 edges per session
        caps = dict(caps, E=caps['N'] * 7)
    xe, le = mk(None)(samples)
    xp, lp = mk(caps)(samples)
    assert lp.numel() == 32 and xp[0].meta['padded']
    model.train()
    m2.train()
    l1 = model.fused_loss(*[x.to(dev) for x in xe], le.to(dev))
    l1.backward()
    l2 = m2.fused_loss(*[x.to(dev) for x in xp], lp.to(dev))
    l2.backward()
    close(l2, l1, rtol=1e-6, atol=1e-6, what='loss')
    close(m2.table_grad.buf, model.table_grad.buf, rtol=1e-4, atol=1e-7, what='table grad')
    p1, p2 = dict(model.named_parameters()), dict(m2.named_parameters())
    for k, p in p1.items():
        if p.grad is not None:
            close(p2[k].grad, p.grad, rtol=1e-4, atol=1e-7, what=k)


@pytest.mark.parametrize('kind', ['niser', 'msgifsr', 'lessr'])
def test_graph_replay_matches_eager_training(dev, kind):
    c, train, optim, G = pkg('collate'), pkg('train'), pkg('optim'), pkg('graph')
    rng = np.random.default_rng(3)
    V = 400
    model, mk = _setup(kind, dev, V)
    ref = copy.deepcopy(model)
    caps = c.default_caps(32, 12)
    if kind == 'lessr':
        caps = dict(caps, E=caps['N'] * 7)
    batches = [_samples(rng, n, V) for n in (32, 32, 20, 32, 27)]     # full and partial batches
    # eager, exact layouts
    opt_r = optim.FusedAdam(train.fix_weight_decay(ref), lr=1e-2, weight_decay=1e-4, model=ref)
    ref.train()
    ref_losses = []
    for s in batches:
        xs, lab = mk(None)(s)
        opt_r.zero_grad()
        loss = ref.fused_loss(*[x.to(dev) for x in xs], lab.to(dev))
        loss.backward()
        opt_r.step()
        ref_losses.append(loss.item())
    # graph replay, padded layouts
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-2, weight_decay=1e-4, model=model)
    model.train()
    padded = [mk(caps)(s) for s in batches]
    x0, l0 = padded[0]
    step = G.GraphedTrainStep(model, opt, [x.to(dev) for x in x0], l0.to(dev))
    losses = []
    for xs, lab in padded:
        losses.append(step([x.to(dev) for x in xs], lab.to(dev)).item())
    assert np.allclose(losses, ref_losses, rtol=2e-5, atol=2e-5), (losses, ref_losses)
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        err = (p.detach() - q.detach()).abs()
        frac = (err <= 2e-5 + 1e-4 * q.detach().abs()).float().mean().item()
        assert frac >= 0.995 and err.max().item() <= 0.02 * 1e-2 * len(batches), (k, frac, err.max().item())


def test_pipelined_graph_replay_is_independent_of_host_timing(dev):
    """Regression: the Adam step scalars (bias corrections) used to be staged through a pinned buffer the host rewrote
    before every replay - a host running ahead of the GPU paired a step with a LATER step's scalars.  The step counter
    now lives on the device: 60 replays queued without any synchronisation give bit-identical parameters to the same 60
    steps with a device sync after each."""
    c, train, optim, G = pkg('collate'), pkg('train'), pkg('optim'), pkg('graph')
    rng = np.random.default_rng(5)
    V = 400
    caps = c.default_caps(32, 12)
    batches = None
    finals = []
    for sync in (True, False):
        torch.manual_seed(0)
        model, mk = _setup('niser', dev, V)
        if batches is None:
            batches = [[x.to(dev) for x in xs] + [lab.to(dev)] for xs, lab in (mk(caps)(_samples(rng, 32, V)) for _ in range(60))]
        opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-2, weight_decay=1e-4, model=model)
        model.train()
        step = G.GraphedTrainStep(model, opt, batches[0][:-1], batches[0][-1])
        for b in batches:
            step(b[:-1], b[-1])
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        finals.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


def test_sharded_single_rank_with_padded_batch(dev):
    """the bench's N>1 configuration (row-sharded table + capacity-padded batches + fixed request capacity),
    exercised with one rank: must equal the plain fused path"""
    c, D = pkg('collate'), pkg('dist')
    rng = np.random.default_rng(5)
    V = 400
    model, mk = _setup('msgifsr', dev, V)
    plain = copy.deepcopy(model)
    samples = _samples(rng, 32, V)
    caps = c.default_caps(32, 12)
    (xp,), lp = mk(caps)(samples)
    (xe,), le = mk(None)(samples)
    vp = D.VocabParallel(model, idx_cap=xp.cap('gidx'))
    l1 = plain.fused_loss(xe.to(dev), le.to(dev))
    l1.backward()
    l2 = model.fused_loss(xp.to(dev), lp.to(dev))
    l2.backward()
    close(l2, l1, rtol=1e-6, atol=1e-6, what='loss')
    close(vp.dE[:V], plain.table_grad.buf, rtol=1e-4, atol=1e-7, what='table grad')


@pytest.mark.parametrize('kind', ['niser', 'lessr'])
def test_mailbox_intake_from_device_pinned_and_pageable_batches(dev, kind):
    """Batch intake of a replayed step (graph.GraphedTrainStep, csrc/rowops.hip copy_words_mailbox_kernel): the first
    kernel of the captured step looks the batch up in a page-locked mailbox at the optimizer's device step count.  200
    replays queued WITHOUT synchronisation - more than three laps of the 64-entry mailbox - fed in turn from device
    tensors, page-locked host buffers (a loader slot, read over PCIe inside the graph) and pageable host buffers (staged)
    end on the parameters of the same 200 steps with a device sync after each; eager steps in between (a batch with
    another layout) keep the mailbox index in step with the device counter; the mismatch flag stays clear.  The losses of
    the unsynchronised run, read from the captured step's device loss ring after the last replay (TrainRunner._loss_handle:
    no clone between graph launches), equal the losses read after every step of the synchronised run."""
    c, train, optim, G = pkg('collate'), pkg('train'), pkg('optim'), pkg('graph')
    rng = np.random.default_rng(9)
    V = 400
    caps = c.default_caps(32, 12)
    if kind == 'lessr':
        caps = dict(caps, E=caps['N'] * 7)
    finals, losses = [], []
    host_batches = None
    for sync in (True, False):
        torch.manual_seed(0)
        model, mk = _setup(kind, dev, V)
        if host_batches is None:
            host_batches = [mk(caps)(_samples(rng, 32, V)) for _ in range(40)]
            odd = mk(None)(_samples(rng, 32, V))                 # exact layout: cannot replay, runs eagerly
        runner = train.TrainRunner('x', model, [], None, dev, lr=1e-2, weight_decay=1e-4)
        model.train()
        for k in range(200):
            xs, lab = host_batches[k % len(host_batches)]
            if k % 3 == 0:
                xs, lab = [x.to(dev) for x in xs], lab.to(dev)
            elif k % 3 == 1:
                xs = [type(x)(x.buf.pin_memory(), x.layout, dict(x.meta)) for x in xs]
            if k in (50, 51, 130):
                runner.train_step(*odd)                            # an eager step between replays
            loss = runner.train_step(xs, lab)
            if sync:
                torch.cuda.synchronize()
                mine = losses[0] if losses else losses.append([]) or losses[0]
                mine.append(float(loss.item()))
            else:
                mine = losses[1] if len(losses) > 1 else losses.append([]) or losses[1]
                mine.append(runner._loss_handle(loss))
        torch.cuda.synchronize()
        gs = runner._gstep
        if not sync:
            assert all(isinstance(h, int) for h in losses[1]) and gs.loss_ring is not None
            ring = gs.loss_ring.tolist()
            losses[1] = [ring[h % len(ring)] for h in losses[1]]
        assert gs is not None and gs._mb is not None and runner.graph_steps == 200 and runner.eager_steps == 3
        assert int(gs._mb['err'].item()) == 0
        finals.append({k_: v.detach().clone() for k_, v in model.state_dict().items()})
    for k_ in finals[0]:
        assert torch.equal(finals[0][k_], finals[1][k_]), k_
    assert losses[0] == losses[1] and len(losses[0]) == 200


@pytest.mark.parametrize('kind', ['niser', 'msgifsr'])
def test_mailbox_fault_is_contained(dev, kind):
    """A replay whose mailbox entry does not carry the device's step count (here: the host's entry is corrupted after it was
    posted) must not train on whatever batch is in the static buffer: the intake kernel raises the fault flag, the optimizer's
    step-scalar kernel turns the step - and every later one - into the identity (parameters AND Adam moments keep their bits),
    and the host raises at its next readback: GraphedTrainStep.check(), which TrainRunner calls at every loss flush and at
    the end of an epoch, also for a run shorter than any polling interval."""
    c, train, G = pkg('collate'), pkg('train'), pkg('graph')
    rng = np.random.default_rng(4)
    V = 400
    caps = c.default_caps(32, 12)
    torch.manual_seed(0)
    model, mk = _setup(kind, dev, V)
    batches = [[x.to(dev) for x in xs] + [lab.to(dev)] for xs, lab in (mk(caps)(_samples(rng, 32, V)) for _ in range(12))]
    runner = train.TrainRunner('x', model, [], None, dev, lr=1e-2, weight_decay=1e-4)
    model.train()
    for k in range(6):
        runner.train_step(batches[k][:-1], batches[k][-1])
    torch.cuda.synchronize()
    gs = runner._gstep
    assert gs is not None and gs._mb is not None and runner.graph_steps == 6
    gs.check()                                                   # clean so far
    before = {k_: v.detach().clone() for k_, v in model.state_dict().items()}
    mom = {id(p): (st['exp_avg'].clone(), st['exp_avg_sq'].clone()) for p, st in runner.optimizer.state.items() if 'exp_avg' in st}
    # the next replay's entry gets a wrong expected count the moment before the graph is launched
    orig_replay = gs.graph.replay
    T = runner.optimizer._T

    def corrupt_then_replay():
        gs._mb['np'][0, T % G._MAILBOX, 3] = T + 1000
        orig_replay()
    gs.graph.replay = corrupt_then_replay
    runner.train_step(batches[6][:-1], batches[6][-1])
    gs.graph.replay = orig_replay
    for k in range(7, 10):                                       # later replays find correct entries - and stay skipped
        runner.train_step(batches[k][:-1], batches[k][-1])
    torch.cuda.synchronize()
    after = model.state_dict()
    for k_ in before:
        assert torch.equal(before[k_], after[k_]), k_
    for p, st in runner.optimizer.state.items():
        if id(p) in mom:
            assert torch.equal(st['exp_avg'], mom[id(p)][0]) and torch.equal(st['exp_avg_sq'], mom[id(p)][1])
    with pytest.raises(RuntimeError, match='mailbox'):
        gs.check()
    # the training loop's own readback raises too: a 4-batch "epoch" is far below the 512-replay polling interval
    runner.train_loader = [(b[:-1], b[-1]) for b in batches[:4]]
    runner.test_loader = [(batches[0][:-1], batches[0][-1])]
    with pytest.raises(RuntimeError, match='mailbox'):
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            runner.train(1, log_interval=2)


@pytest.mark.parametrize('kind', ['msgifsr', 'niser', 'msgifsr_bf16_d128'])
def test_step_scalars_riding_in_the_slab_sum_launch_change_nothing(dev, kind, monkeypatch):
    """In a captured step the optimizer's step-scalar kernel (device step counters, Adam bias corrections, the loss tap) rides in the
    end-of-backward slab-sum launch (optim.FusedAdam.hyper_rider -> srec_sum_slabs_multi_hyper) instead of being a launch of its
    own behind it: one kernel node less, the same arithmetic - parameters, losses (read from the device ring the rider writes) and
    step counters after six replays are BIT-identical to the captured step without the rider (SREC_HYPER_RIDER=0)."""
    c, train, optim, G = pkg('collate'), pkg('train'), pkg('optim'), pkg('graph')
    V = 400
    caps = c.default_caps(32, 12)
    runs = {}
    wide = kind == 'msgifsr_bf16_d128'                      # (the configuration whose backward DOES defer slab sums: the rider is taken)
    if wide:
        monkeypatch.setitem(pkg('ops').PRECISION, 'matmul', 'bf16')        # (restored by the fixture, whatever happens below)
        pkg('ops').weights_changed()
    for rider in ('0', '1'):
        monkeypatch.setenv('SREC_HYPER_RIDER', rider)
        rng = np.random.default_rng(9)
        model, mk = _setup('msgifsr' if wide else kind, dev, V, d=128 if wide else 32)
        opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-2, weight_decay=1e-4, model=model)
        model.train()
        padded = [mk(caps)(_samples(rng, n, V)) for n in (32, 32, 20, 32, 27, 32)]
        x0, l0 = padded[0]
        step = G.GraphedTrainStep(model, opt, [x.to(dev) for x in x0], l0.to(dev))
        losses = [step([x.to(dev) for x in xs], lab.to(dev)).item() for xs, lab in padded]
        step.check()
        nodes = step.node_counts()
        runs[rider] = (losses, {k: p.detach().clone() for k, p in model.named_parameters()},
                       int(opt._hyper[(0, 0)]['counter'].item()), nodes['kernel'] if nodes else None)
    (l0_, p0, c0, n0), (l1_, p1, c1, n1) = runs['0'], runs['1']
    if wide and n0 is not None:
        assert n1 == n0 - 1, (n0, n1)
    assert l0_ == l1_ and c0 == c1 == 6                     # (six steps taken, whoever advanced the device counter)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    if n0 is not None:                                      # (a backward that defers no slab sums has no launch to ride in: d = 32 here;
        assert n1 in (n0, n0 - 1), (n0, n1)                 #  the benchmarked step goes from 35 to 34 kernels)
