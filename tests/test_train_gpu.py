"""Config C1 (BASELINE.json configs[0]): SRGNN on the reference's shipped datasets/sample split, batch 32,
driven by the TrainRunner exactly like start.sh does - the HIP path on the GPU must reproduce the loss trace
and the (MRR@20, HR@20) of the CPU oracle on the same batches."""
import os

import numpy as np
import pytest
import torch

from util import ROOT, pkg

pytestmark = pytest.mark.gpu


class _Wrap:
    def __init__(self, x):
        self.x = x

    def to(self, device):
        return self.x


@pytest.mark.parametrize('model_name', ['SRGNN', 'LESSR', 'MSGIFSR'])
def test_sample_dataset_training_matches_oracle(dev, model_name):
    from oracle import collate_ref as oc, models_ref as om
    sp, ds, col, train = pkg(), pkg('dataset'), pkg('collate'), pkg('train')
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    train_set, test_set = ds.AugmentedDataset(tr), ds.AugmentedDataset(te)
    B, n_train, n_test = 32, 12, 6
    torch.manual_seed(123)
    if model_name == 'SRGNN':
        ref, model = om.SRGNN(V, 32, 1), sp.SRGNN(V, 32, 1)
        cf, of = col.collate_fn_factory(col.seq_to_session_graph), oc.collate_fn_factory(oc.seq_to_session_graph)
    elif model_name == 'LESSR':
        ref, model = om.LESSR(V, 32, 2), sp.LESSR(V, 32, 2)
        cf = col.collate_fn_factory(col.seq_to_eop_multigraph, col.seq_to_shortcut_graph)
        of = oc.collate_fn_factory(oc.seq_to_eop_multigraph, oc.seq_to_shortcut_graph)
    else:
        ref = om.MSGIFSR(V, 'sample', 32, 1, order=2, extra=False, fusion=False)
        model = sp.MSGIFSR(V, 'sample', 32, 1, order=2, extra=False, fusion=False)
        cf = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 2)
        of = oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), 2)
    model.load_state_dict(ref.state_dict())
    model = model.to(dev)

    def batches(data, n, fn, wrap):
        out = []
        for b in range(n):
            inp, lab = fn([data[i] for i in range(b * B, (b + 1) * B)])
            if wrap:
                out.append(([_Wrap(om.to_torch(x)) for x in inp], torch.from_numpy(lab)))
            else:
                out.append((inp, lab))
        return out
    r_ref = train.TrainRunner('sample', ref, batches(train_set, n_train, of, True), batches(test_set, n_test, of, True),
                              torch.device('cpu'), lr=1e-3, weight_decay=1e-4, patience=2)
    r_gpu = train.TrainRunner('sample', model, batches(train_set, n_train, cf, False), batches(test_set, n_test, cf, False),
                              dev, lr=1e-3, weight_decay=1e-4, patience=2)
    assert r_gpu.fused
    m_ref = r_ref.train(2, log_interval=100)
    m_gpu = r_gpu.train(2, log_interval=100)
    a, b = np.array(r_gpu.loss_trace), np.array(r_ref.loss_trace)
    assert a.shape == b.shape == (2 * n_train,)
    assert np.allclose(a, b, rtol=2e-4, atol=2e-4), np.abs(a - b).max()
    assert abs(m_gpu[0] - m_ref[0]) <= 0.02 and abs(m_gpu[1] - m_ref[1]) <= 0.02, (m_gpu, m_ref)   # (MRR@20, HR@20)


def test_checkpoint_resume_on_the_fused_path(dev, tmp_path):
    """TrainRunner(checkpoint=...) on the HIP path: FusedAdam's moments AND its device-side step counters come back, so a
    run resumed after epoch 2 repeats the uninterrupted run's losses bit for bit."""
    import copy
    sp, ds, col, train = pkg(), pkg('dataset'), pkg('collate'), pkg('train')
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    train_set = ds.AugmentedDataset(tr)
    B, n = 32, 6
    torch.manual_seed(5)
    m0 = sp.NISER(V, 32, 1).to(dev)
    cf = col.collate_fn_factory(col.seq_to_session_graph)
    loader = [cf([train_set[i] for i in range(b * B, (b + 1) * B)]) for b in range(n)]
    kw = dict(lr=1e-3, weight_decay=1e-4, patience=9)
    full = train.TrainRunner('sample', copy.deepcopy(m0), loader, loader, dev, **kw)
    full.train(4, log_interval=100)
    ck = str(tmp_path / 'run.pt')
    first = train.TrainRunner('sample', copy.deepcopy(m0), loader, loader, dev, checkpoint=ck, **kw)
    first.train(2, log_interval=100)
    second = train.TrainRunner('sample', copy.deepcopy(m0), loader, loader, dev, checkpoint=ck, **kw)
    assert second.fused
    second.train(4, log_interval=100)
    assert second.epoch == 4 and second.loss_trace == full.loss_trace[2 * n:]
    for (k, a), (_, b) in zip(full.model.state_dict().items(), second.model.state_dict().items()):
        assert torch.equal(a, b), k


def test_checkpoint_carries_the_private_generators(dev, tmp_path):
    """A run resumed in a NEW process continues the shuffle order and the dropout masks of the uninterrupted run: the
    sampler's generator (src/scripts/common.py: RandomSampler(generator=...)) and the nonce stream of the HIP path's dropout
    masks (ops._nonce) are private generators - not torch's global RNG state - and ride in the checkpoint."""
    from torch.utils.data import DataLoader, RandomSampler
    sp, ds, col, train, ops = pkg(), pkg('dataset'), pkg('collate'), pkg('train'), pkg('ops')
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    train_set = ds.AugmentedDataset(tr)
    torch.manual_seed(5)
    ops.seed_dropout()
    model = sp.NISER(V, 32, 1, feat_drop=0.2).to(dev)
    gen = torch.Generator().manual_seed(123)
    loader = DataLoader(train_set, batch_size=32, sampler=RandomSampler(train_set, generator=gen),
                        collate_fn=col.collate_fn_factory(col.seq_to_session_graph))
    runner = train.TrainRunner('sample', model, loader, loader, dev, lr=1e-3, weight_decay=1e-4)
    assert runner._loader_generators() == [gen]
    [ops._nonce() for _ in range(3)]
    next(iter(loader))                                            # one epoch's permutation drawn
    ck = str(tmp_path / 'rng.pt')
    runner.save_checkpoint(ck)
    want_perm = list(iter(RandomSampler(train_set, generator=gen)))[:16]
    want_nonce = [ops._nonce() for _ in range(4)]
    # "new process": both generators somewhere else entirely
    gen.manual_seed(999)
    ops.seed_dropout(777)
    runner.load_checkpoint(ck)
    assert list(iter(RandomSampler(train_set, generator=gen)))[:16] == want_perm
    assert [ops._nonce() for _ in range(4)] == want_nonce


@pytest.mark.parametrize('model_name,dim', [('MSGIFSR', 256), ('NISER', 128), ('LESSR', 32)])
def test_bf16_training_metrics_within_0p3pt_of_fp32(dev, model_name, dim):
    """Recall@20 / MRR@20 after training in bf16 mode within +-0.3 pt (absolute) of the same
    model (same init, same batches) trained in fp32 on the same split (datasets/sample, 3 epochs, batch 512)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bf16_metric_check
    r = bf16_metric_check.run(model_name, epochs=3, dim=dim)
    assert r['fp32']['hit'] > 20.0                                  # the model did learn something
    assert abs(r['d_mrr_pt']) <= 0.3 and abs(r['d_hit_pt']) <= 0.3, r


@pytest.mark.parametrize('model_name', ['MSGIFSR', 'NISER', 'LESSR'])
def test_train_runner_replays_the_captured_step(dev, model_name, tmp_path):
    """TrainRunner(graph='auto'): capacity-padded batches replay ONE captured hipGraph of the whole step, batches that do
    not fit the capacities (collated unpadded) run eagerly in between - and the run equals the all-eager run on the same
    batches bit for bit,
    also across a checkpoint resume (the optimizer's moments survive the capture warm-up)."""
    import copy
    sp, ds, col, train = pkg(), pkg('dataset'), pkg('collate'), pkg('train')
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    train_set = ds.AugmentedDataset(tr)
    B, n = 64, 10
    caps = col.estimate_caps(train_set, B)
    torch.manual_seed(11)
    if model_name == 'MSGIFSR':
        m0 = sp.MSGIFSR(V, 'sample', 32, 1, dropout=0.0, order=2, extra=False, fusion=True).to(dev)
        mk = lambda c: col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 2, caps=c)
    elif model_name == 'LESSR':
        caps = dict(caps, E=caps['N'] * 7)
        m0 = sp.LESSR(V, 32, 2).to(dev)
        mk = lambda c: col.collate_fn_factory(col.seq_to_eop_multigraph, col.seq_to_shortcut_graph, caps=c)
    else:
        m0 = sp.NISER(V, 32, 1).to(dev)
        mk = lambda c: col.collate_fn_factory(col.seq_to_session_graph, caps=c)
    samples = lambda b: [train_set[i] for i in range(b * B, (b + 1) * B)]
    tight = dict(caps, N=64, E=64, U=64)                         # too small for a 64-session batch: falls back to exact
    padded = [mk(caps)(samples(b)) for b in range(n)]
    padded[4] = mk(tight)(samples(4))
    assert padded[0][0][0].meta['padded'] and not padded[4][0][0].meta['padded']
    exact = [mk(None)(samples(b)) for b in range(n)]
    kw = dict(lr=1e-3, weight_decay=1e-4, patience=9)
    # same batches, same layouts, eager launches (exact layouts pick other split-K factors: equal only to round-off)
    eager = train.TrainRunner('sample', copy.deepcopy(m0), padded, exact[:2], dev, graph=False, **kw)
    eager.train(2, log_interval=100)
    assert eager.graph_steps == 0
    g = train.TrainRunner('sample', copy.deepcopy(m0), padded, exact[:2], dev, **kw)
    g.train(2, log_interval=100)
    assert g.graph_steps == 2 * (n - 1) and g.eager_steps == 2
    assert g.loss_trace == eager.loss_trace
    for (k, a), (_, b) in zip(eager.model.state_dict().items(), g.model.state_dict().items()):
        assert torch.equal(a, b), k
    # resume after epoch 1 into a fresh runner: capture happens with loaded Adam moments
    ck = str(tmp_path / 'g.pt')
    first = train.TrainRunner('sample', copy.deepcopy(m0), padded, exact[:2], dev, checkpoint=ck, **kw)
    first.train(1, log_interval=100)
    second = train.TrainRunner('sample', copy.deepcopy(m0), padded, exact[:2], dev, checkpoint=ck, **kw)
    second.train(2, log_interval=100)
    assert second.graph_steps == n - 1 and second.loss_trace == eager.loss_trace[n:]


@pytest.mark.parametrize('model_name', ['MSGIFSR', 'NISER'])
def test_training_from_the_pinned_ring_loader_matches_the_dataloader(dev, model_name):
    """loader.PinnedRingLoader feeding TrainRunner.train_step (captured steps staged from the pinned ring slots, a ring of
    only 4 slots so every slot is rewritten many times while steps are in flight) against the same steps fed by the torch
    DataLoader: identical loss traces and parameters - no slot is overwritten before the copy that reads it."""
    from torch.utils.data import BatchSampler, DataLoader, SequentialSampler
    sp, ds, col, train, L = pkg(), pkg('dataset'), pkg('collate'), pkg('train'), pkg('loader')
    rng = np.random.RandomState(11)
    V, B = 500, 64
    sessions = [rng.randint(1, V, size=rng.randint(2, 14)).tolist() for _ in range(700)]
    data = ds.AugmentedDataset(sessions)
    data.index = data.index[:60 * B]
    kind, order = ('ccs', 2) if model_name == 'MSGIFSR' else ('session', 1)
    caps = col.measure_caps(data, B, kind, order)
    sampler = BatchSampler(SequentialSampler(data), B, drop_last=False)
    fn = (col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), order, caps=caps) if kind == 'ccs'
          else col.collate_fn_factory(col.seq_to_session_graph, caps=caps))

    def run(loader):
        torch.manual_seed(7)
        model = (sp.MSGIFSR(V, 'x', 32, 1, dropout=0.0, order=2, extra=False, fusion=False) if kind == 'ccs'
                 else sp.NISER(V, 32, 1, feat_drop=0.0)).to(dev).train()
        runner = train.TrainRunner('x', model, loader, None, dev, lr=1e-3, weight_decay=1e-4)
        losses = [runner.train_step(inp, lab).detach().clone() for inp, lab in loader]
        torch.cuda.synchronize()
        return torch.stack(losses).cpu(), {k: v.detach().cpu() for k, v in model.state_dict().items()}, runner
    ring = L.PinnedRingLoader(data, sampler, kind, order=order, caps=caps, num_workers=2, slots=4)
    try:
        assert ring.pinned and ring.ring_t[:16].is_pinned()
        l_ring, p_ring, r_ring = run(ring)
    finally:
        ring.close()
    l_ref, p_ref, r_ref = run(DataLoader(data, batch_sampler=sampler, collate_fn=fn, num_workers=0, pin_memory=True))
    assert r_ring.graph_steps >= 50 and r_ring.graph_steps == r_ref.graph_steps
    assert torch.allclose(l_ring, l_ref, rtol=1e-5, atol=1e-6), (l_ring - l_ref).abs().max()
    for k in p_ref:
        assert torch.allclose(p_ring[k].float(), p_ref[k].float(), rtol=1e-4, atol=2e-6), k
