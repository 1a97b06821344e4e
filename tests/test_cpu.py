"""CPU suite (`-m "not gpu"`): oracle vs the golden vectors generated from the reference, host
logic (collate / dataset / train-loop plumbing) and the C-ABI surface (library loads and exports
every symbol include/srec.h declares; no compute calls without a GPU)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, ROOT, close, load_golden, pkg

from oracle import collate_ref as oc
from oracle import models_ref as om

ALL_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith('.npz') and f != 'srgnn_evaluate.npz'
                   and not f.startswith('srgnn_layer_'))       # whole-model fixtures (the layer fixtures have their own test)


def _oracle(name, V):
    d = 32
    if name.startswith('srgnn'):
        return om.SRGNN(V, d, 1), oc.collate_fn_factory(oc.seq_to_session_graph)
    if name.startswith('niser'):
        return om.NISER(V, d, 1), oc.collate_fn_factory(oc.seq_to_session_graph)
    if name.startswith('lessr'):
        L = int(name.split('_')[1][1:])
        fns = (oc.seq_to_eop_multigraph, oc.seq_to_shortcut_graph) if L > 1 else (oc.seq_to_eop_multigraph,)
        return om.LESSR(V, d, L), oc.collate_fn_factory(*fns)
    K = int(name.split('_')[1][1:])
    return (om.MSGIFSR(V, 'sample', d, 1, order=K, extra='_ext' in name, fusion='_fus' in name,
                       reducer='max' if '_max' in name else 'concat' if '_concat' in name else 'mean'),
            oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), K))


@pytest.mark.parametrize('name', ALL_CASES)
def test_oracle_reproduces_reference_fixture(name):
    """The oracle, re-run from the fixture's inputs and seeded weights, reproduces what the
    reference itself produced (log-probs, loss trace over 3 Adam steps, gradients)."""
    train = pkg('train')
    z, samples, init = load_golden(name)
    V = [v for k, v in init.items() if k.startswith('embedding')][0].shape[0]
    model, fn = _oracle(name, V)
    model.load_state_dict(init, strict=True)
    inputs, labels = fn(samples)
    inputs = [om.to_torch(x) for x in inputs]
    labels = torch.from_numpy(labels)
    opt = torch.optim.Adam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4)
    model.train()
    losses = []
    for step in range(3):
        opt.zero_grad()
        lp = model(*inputs)
        loss = torch.nn.functional.nll_loss(lp, labels)
        loss.backward()
        if step == 0:
            ref = torch.from_numpy(z['logprobs'])
            close(lp[:ref.shape[0]], ref, rtol=2e-5, atol=2e-6, what='logprobs')
            params = dict(model.named_parameters())
            for k in z.files:
                if k.startswith('grad/'):
                    close(params[k[5:]].grad, z[k], rtol=1e-4, atol=1e-7, what=k)
                if k.startswith('nograd/'):
                    assert params[k[7:]].grad is None
        opt.step()
        losses.append(loss.item())
    assert np.allclose(losses, z['losses'], rtol=1e-6, atol=1e-6), (losses, z['losses'])


def test_evaluate_tuple_order_and_values():
    """train.evaluate returns (MRR@20, HR@20) like the reference's evaluate (train.py:36-55)."""
    train = pkg('train')
    z = np.load(os.path.join(GOLDEN, 'srgnn_evaluate.npz'))
    init = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('init/')}
    model = om.SRGNN(3429, 32, 1)
    model.load_state_dict(init)
    ds = pkg('dataset')
    test = ds.read_sessions(os.path.join(ROOT, 'tests', 'golden', 'sample_test.txt'))
    data = ds.AugmentedDataset(test)
    fn = oc.collate_fn_factory(oc.seq_to_session_graph)

    class Wrap:
        def __init__(self, x):
            self.x = x

        def to(self, device):
            return self.x
    batches = []
    for b in range(10):
        inp, lab = fn([data[i] for i in range(b * 32, b * 32 + 32)])
        batches.append(([Wrap(om.to_torch(x)) for x in inp], torch.from_numpy(lab)))
    mrr, hit = train.evaluate(model, batches, torch.device('cpu'))
    assert abs(mrr - float(z['mrr'])) < 1e-7 and abs(hit - float(z['hit'])) < 1e-9, (mrr, hit, z['mrr'], z['hit'])


# ---------------------------------------------------------------------------------------- host logic
def _rand_samples(rng, n, V=60, max_len=12):
    out = []
    for _ in range(n):
        L = int(rng.integers(1, max_len))
        seq = rng.integers(0, V, size=L).tolist()
        if rng.random() < 0.3 and L > 1:
            seq[1] = seq[0]
        out.append((seq, int(rng.integers(0, V))))
    return out


EDGE = [([7], 3), ([5, 5], 9), ([4, 9], 1), ([3, 1, 3, 6, 2, 5, 1, 2, 4, 1, 2], 8), ([2, 2, 2], 2),
        ([250, 250, 250, 250, 3, 1, 2, 4, 1], 2), ([11, 12, 13], 14)]


@pytest.mark.parametrize('kind', ['session', 'eop', 'shortcut'])
def test_collate_homogeneous_matches_oracle(kind):
    c = pkg('collate')
    rng = np.random.default_rng(7)
    samples = EDGE + _rand_samples(rng, 40)
    pf = dict(session=c.seq_to_session_graph, eop=c.seq_to_eop_multigraph, shortcut=c.seq_to_shortcut_graph)[kind]
    of = dict(session=oc.seq_to_session_graph, eop=oc.seq_to_eop_multigraph, shortcut=oc.seq_to_shortcut_graph)[kind]
    (fb,), labels = c.collate_fn_factory(pf)(samples)
    (ob,), olab = oc.collate_fn_factory(of)(samples)
    assert np.array_equal(labels.numpy(), olab)
    assert np.array_equal(fb.esrc.numpy(), ob['src']) and np.array_equal(fb.edst.numpy(), ob['dst'])
    assert np.array_equal(np.diff(fb.seg.numpy()), ob['num_nodes'])
    assert np.array_equal(np.diff(fb.eseg.numpy()), ob['num_edges'])
    if kind != 'shortcut':
        assert np.array_equal(fb.iid.numpy(), ob['iid']) and np.array_equal(fb.last.numpy(), ob['last'])
        # item -> positions CSR really inverts the lookup
        it, pt, ps = fb.uniq_items.numpy(), fb.uniq_ptr.numpy(), fb.uniq_pos.numpy()
        assert np.all(np.diff(it) > 0)
        for u in range(len(it)):
            assert np.all(fb.iid.numpy()[ps[pt[u]:pt[u + 1]]] == it[u])
        assert pt[-1] == len(fb.iid)
        # ... and the per-position inverse (what the row-sharded lookup gathers the exchanged rows through) names each
        # position's item
        assert np.array_equal(it[fb.uniq_inv.numpy()], fb.iid.numpy())
    if kind == 'session':
        assert np.array_equal(fb.ew.numpy(), ob['w'])
    # in-edge CSR: grouped by destination, edge ids ascending inside a group (EOPA's time order)
    ip, ii = fb.in_ptr.numpy(), fb.in_idx.numpy()
    for v in range(len(ip) - 1):
        e = ii[ip[v]:ip[v + 1]]
        assert np.all(ob['dst'][e] == v) and np.all(np.diff(e) > 0)
    op, oi = fb.out_ptr.numpy(), fb.out_idx.numpy()
    for v in range(len(op) - 1):
        e = oi[op[v]:op[v + 1]]
        assert np.all(ob['src'][e] == v) and np.all(np.diff(e) > 0)


@pytest.mark.parametrize('K', [1, 2, 3, 4])
def test_collate_ccs_matches_oracle(K):
    c = pkg('collate')
    rng = np.random.default_rng(11)
    samples = EDGE + _rand_samples(rng, 40)
    (fb,), labels = c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), K)(samples)
    (ob,), olab = oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), K)(samples)
    B = len(samples)
    for k in range(1, K + 1):
        assert np.array_equal(np.diff(fb.field('seg%d' % k).numpy()), ob['num_nodes'][k])
        assert np.array_equal(fb.field('iid%d' % k).numpy().reshape(ob['iid'][k].shape), ob['iid'][k])
        assert np.array_equal(fb.field('last%d' % k).numpy(), ob['last'][k])
    for key, name in fb.meta['rels']:
        r = ob['rel'][key]
        assert np.array_equal(fb.field(name + '_src').numpy(), r['src']), key
        assert np.array_equal(fb.field(name + '_dst').numpy(), r['dst']), key
    # per-session concatenation permutation of the multi-order readout
    perm, inv, cseg = fb.cat_perm.numpy(), fb.cat_inv.numpy(), fb.cat_seg.numpy()
    assert np.array_equal(perm[inv], np.arange(len(perm)))
    offs = np.concatenate([[0], np.cumsum([fb.count('N%d' % k) for k in range(1, K + 1)])])
    for i in range(B):
        want = np.concatenate([np.arange(int(fb.field("seg%d" % k)[i]), int(fb.field("seg%d" % k)[i + 1])) + offs[k - 1]
                               for k in range(1, K + 1)])
        assert np.array_equal(perm[cseg[i]:cseg[i + 1]], want)
    # the flat lookup list, its distinct items and the per-position inverse (-1: no item at that position)
    g, it, ui = fb.gidx.numpy(), fb.uniq_items.numpy(), fb.uniq_inv.numpy()
    assert np.array_equal(ui >= 0, g >= 0) and np.array_equal(it[ui[g >= 0]], g[g >= 0])


def test_reference_collate_smoke_answer():
    """collate.py:258-266 (tuple argument fixed): batch_num_nodes('s2') == [9, 6]."""
    c = pkg('collate')
    seq = [3, 1, 3, 6, 2, 5, 1, 2, 4, 1, 2]
    seq0 = [250, 250, 250, 250, 3, 1, 2, 4, 1]
    (fb,), _ = c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), 2)([[seq, 1], [seq0, 2]])
    assert fb.batch_num_nodes(2).tolist() == [9, 6]


def test_dataset_index_and_files(tmp_path):
    ds = pkg('dataset')
    sess = [[1, 2, 3], [4, 5], [6], [7, 8, 9, 10]]
    idx = ds.create_index(sess)
    assert idx.tolist() == [[0, 1], [0, 2], [1, 1], [3, 1], [3, 2], [3, 3]]
    (tmp_path / 'train.txt').write_text('0,1,2\n3,4\n')
    (tmp_path / 'test.txt').write_text('5,6,5\n')
    (tmp_path / 'num_items.txt').write_text('7\n')
    tr, te, n = ds.read_dataset(tmp_path)
    assert n == 7 and list(tr[0]) == [0, 1, 2] and list(te[0]) == [5, 6, 5]
    a = ds.AugmentedDataset(tr)
    assert len(a) == 3 and a[1] == ([0, 1], 2)


def test_fix_weight_decay_groups():
    train = pkg('train')
    m = om.LESSR(50, 8, 2)
    decay, no_decay = train.fix_weight_decay(m)
    names = {id(p): n for n, p in m.named_parameters()}
    nd = {names[id(p)] for p in no_decay['params']}
    assert all(('bias' in n) or ('batch_norm' in n) or ('activation' in n) for n in nd)
    assert 'indices' not in {names[id(p)] for p in decay['params']}     # frozen Parameter excluded
    assert no_decay['weight_decay'] == 0


def test_train_runner_cpu_plumbing():
    """Config C1 plumbing: the runner drives a CPU model end to end (oracle SRGNN, batch 32) with the
    reference's log strings, scheduler and (mrr, hit) contract."""
    train = pkg('train')
    rng = np.random.default_rng(0)
    V = 80
    samples = _rand_samples(rng, 96, V=V, max_len=8)
    fn = oc.collate_fn_factory(oc.seq_to_session_graph)

    class G:
        def __init__(self, x):
            self.x = om.to_torch(x)

        def to(self, device):
            return self.x
    loader = []
    for b in range(3):
        inp, lab = fn(samples[b * 32:(b + 1) * 32])
        loader.append(([G(x) for x in inp], torch.from_numpy(lab)))
    torch.manual_seed(0)
    model = om.SRGNN(V, 16, 1)
    runner = train.TrainRunner('x', model, loader, loader, torch.device('cpu'), lr=1e-2, weight_decay=1e-4, patience=2)
    mrr, hit = runner.train(2, log_interval=2)
    assert 0 <= mrr <= hit <= 1
    assert len(runner.loss_trace) == 6 and runner.loss_trace[-1] < runner.loss_trace[0]


def test_train_runner_checkpoint_resume_and_hooks(tmp_path):
    """checkpoint / resume / metric hooks (SURVEY 8(f) rank 4 - absent in the reference): a run interrupted after
    epoch 2 and resumed from its checkpoint continues exactly like the uninterrupted 4-epoch run."""
    import copy
    train = pkg('train')
    rng = np.random.default_rng(1)
    V = 80
    samples = _rand_samples(rng, 96, V=V, max_len=8)
    fn = oc.collate_fn_factory(oc.seq_to_session_graph)

    class G:
        def __init__(self, x):
            self.x = om.to_torch(x)

        def to(self, device):
            return self.x
    loader = []
    for b in range(3):
        inp, lab = fn(samples[b * 32:(b + 1) * 32])
        loader.append(([G(x) for x in inp], torch.from_numpy(lab)))
    torch.manual_seed(0)
    m0 = om.SRGNN(V, 16, 1)
    cpu = torch.device('cpu')
    kw = dict(lr=1e-2, weight_decay=1e-4, patience=9)
    events = []
    full = train.TrainRunner('x', copy.deepcopy(m0), loader, loader, cpu, hooks=[events.append], **kw)
    full.train(4, log_interval=2)
    assert [e['epoch'] for e in events if e['kind'] == 'epoch'] == [0, 1, 2, 3]
    assert all(e['loss'] == e['loss'] for e in events if e['kind'] == 'interval') and any(e['kind'] == 'interval' for e in events)
    ck = str(tmp_path / 'run.pt')
    first = train.TrainRunner('x', copy.deepcopy(m0), loader, loader, cpu, checkpoint=ck, **kw)
    first.train(2, log_interval=2)
    assert os.path.exists(ck) and not os.path.exists(ck + '.tmp')
    second = train.TrainRunner('x', copy.deepcopy(m0), loader, loader, cpu, checkpoint=ck, **kw)
    second.train(4, log_interval=2)
    assert second.epoch == 4 and second.batch == 12
    assert second.loss_trace == full.loss_trace[6:]                   # same batches, same lr schedule, same moments
    for (k, a), (_, b) in zip(full.model.state_dict().items(), second.model.state_dict().items()):
        assert torch.equal(a, b), k


def test_estimate_caps_and_overflow_fallback():
    """estimate_caps bounds every sequential batch of the dataset; a batch that does not fit given capacities is collated
    with the exact layout instead (TrainRunner then runs it eagerly)"""
    ds, col = pkg('dataset'), pkg('collate')
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    data = ds.AugmentedDataset(tr)
    B = 64
    caps = col.estimate_caps(data, B)
    assert caps['B'] == B and caps['N'] % 256 == 0 and caps['N'] <= B * int(data.index[:, 1].max())
    for fac in (lambda c: col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 3, caps=c),
                lambda c: col.collate_fn_factory(col.seq_to_session_graph, caps=c)):
        ref_layout = None
        for b in range(0, len(data) // B, 7):
            (fb,), lab = fac(caps)([data[i] for i in range(b * B, (b + 1) * B)])
            assert fb.meta['padded'] and lab.numel() == B
            ref_layout = ref_layout or fb.layout
            assert fb.layout == ref_layout                      # batch-independent offsets: one device buffer, one graph
        (fb,), _ = fac(dict(caps, N=32, E=32, U=32))([data[i] for i in range(B)])
        assert not fb.meta['padded']
        (ex,), _ = fac(None)([data[i] for i in range(B)])
        assert torch.equal(fb.buf, ex.buf)


# ---------------------------------------------------------------------------------------- C ABI
def test_c_abi_exports_every_declared_symbol():
    L = pkg('_lib')
    protos = L.parse_header()
    assert len(protos) >= 15
    dll = L.lib.load()
    for name in protos:
        assert hasattr(dll, name), name


def test_collate_abi_header_matches_the_library():
    """include/srec_collate.h declares the native collate entry point: the symbol is exported and behaves as documented
    (returns the int32 count; 0 for an empty session; -(required) when the output buffer is too small)"""
    import ctypes
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'srec_collate.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = re.findall(r'\b(?:long|int)\s+(srec_\w+)\s*\(', hdr)
    assert names == ['srec_collate'], names
    c = pkg('collate')
    dll = c._native()
    if dll is None:
        pytest.skip('libsrec_collate.so not built')
    for n in names:
        assert hasattr(dll, n), n
    seqs = np.array([3, 5, 3, 7, 7], dtype=np.int64)
    offs = np.array([0, 3, 5], dtype=np.int64)
    info = np.zeros(3 * 40, dtype=np.int64)
    nf = ctypes.c_int(0)
    out = np.empty(4096, dtype=np.int32)
    n = dll.srec_collate(0, seqs.ctypes.data, offs.ctypes.data, 2, 1, None, out.ctypes.data, out.size, info.ctypes.data, 40,
                         ctypes.addressof(nf))
    assert n > 32 and nf.value > 0 and out[0] == 2                      # header slot 0 = B
    small = dll.srec_collate(0, seqs.ctypes.data, offs.ctypes.data, 2, 1, None, out.ctypes.data, 8, info.ctypes.data, 40,
                             ctypes.addressof(nf))
    assert small == -n
    bad = np.array([0, 3, 3], dtype=np.int64)                           # second session empty
    assert dll.srec_collate(0, seqs.ctypes.data, bad.ctypes.data, 2, 1, None, out.ctypes.data, out.size, info.ctypes.data,
                            40, ctypes.addressof(nf)) == 0


def test_product_ops_refuse_cpu_tensors():
    ops = pkg('ops')
    x = torch.randn(4, 8)
    w = torch.randn(8, 8)
    with pytest.raises(RuntimeError):
        ops.linear(x, w)


# ---------------------------------------------------------------------------------------- native collate
@pytest.mark.parametrize('kind,order,padded', [('session', 1, False), ('eop', 1, False), ('shortcut', 1, False),
                                               ('ccs', 1, False), ('ccs', 2, False), ('ccs', 3, False), ('ccs', 4, False),
                                               ('session', 1, True), ('ccs', 3, True)])
def test_native_collate_is_bit_identical_to_python_builder(kind, order, padded):
    """csrc/collate.cpp must emit exactly the buffer, layout and header of the python builders"""
    c = pkg('collate')
    if c._native() is None:
        pytest.skip('libsrec_collate.so not built')
    rng = np.random.default_rng(17)
    samples = EDGE + _rand_samples(rng, 57, V=40, max_len=15)
    seqs = [s for s, _ in samples]
    caps = c.default_caps(64, 15) if padded else None
    if kind == 'ccs':
        ref = c.batch_ccs([c.seq_to_ccs_graph(s, order) for s in seqs], caps)
    else:
        fn = dict(session=c.seq_to_session_graph, eop=c.seq_to_eop_multigraph, shortcut=c.seq_to_shortcut_graph)[kind]
        ref = c.batch_homogeneous([fn(s) for s in seqs], caps)
    nat = c.collate_native(kind, seqs, order, caps)
    assert nat.layout == ref.layout, [(k, nat.layout.get(k), v) for k, v in ref.layout.items() if nat.layout.get(k) != v][:5]
    assert nat.meta['counts'] == ref.meta['counts'] and nat.meta['slots'] == ref.meta['slots']
    for key in ('kind', 'B', 'padded', 'max_nodes', 'order', 'rels', 'ncap'):
        assert nat.meta.get(key) == ref.meta.get(key), (key, nat.meta.get(key), ref.meta.get(key))
    assert nat.buf.numel() == ref.buf.numel()
    diff = torch.nonzero(nat.buf != ref.buf).reshape(-1)
    assert diff.numel() == 0, ('first differing word', int(diff[0]), [k for k, v in ref.layout.items() if v[0] <= int(diff[0]) < v[0] + v[1]])


def test_session_store_equals_text_path(tmp_path):
    """binary CSR session store: same sessions, same prefix-sample order, same collated batch as
    the text reader; survives a save / memory-mapped load round trip and is reused by read_dataset(cache=True)."""
    import shutil
    ds, col = pkg('dataset'), pkg('collate')
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'datasets', 'sample')
    for f in ('train.txt', 'test.txt', 'num_items.txt'):
        shutil.copy(os.path.join(src, f), tmp_path / f)
    tr, te, n = ds.read_dataset(tmp_path)
    trs, tes, n2 = ds.read_dataset(tmp_path, cache=True)
    assert n == n2 and isinstance(trs, ds.SessionStore) and (tmp_path / 'train.sstore.npz').exists()
    trs2, _, _ = ds.read_dataset(tmp_path, cache=True)          # second call: memory-mapped reload
    for a, b in ((tr, trs), (te, tes), (tr, trs2)):
        assert len(a) == len(b)
        assert all(list(x) == list(y) for x, y in zip(a, b))
    A, S = ds.AugmentedDataset(tr), ds.AugmentedDataset(trs2)
    assert np.array_equal(A.index, S.index)
    ia, ib = [A[i] for i in range(64)], [S[i] for i in range(64)]
    assert all(list(x[0]) == list(y[0]) and int(x[1]) == int(y[1]) for x, y in zip(ia, ib))
    fn = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 3)
    (ba,), la = fn([(list(s), int(l)) for s, l in ia])
    (bb,), lb = fn([(s, l) for s, l in ib])
    assert torch.equal(ba.buf, bb.buf) and torch.equal(la, lb)


# ---------------------------------------------------------------------------------------- SRGNNLayer, reference-pinned
@pytest.mark.parametrize('name', ['srgnn_layer_s32', 'srgnn_layer_edge'])
def test_oracle_srgnn_layer_reproduces_the_reference_layer(name):
    """fixture = the reference's SRGNNLayer.forward called directly (srgnn.py:11-51; make_golden.py
    srgnn_layer_cases): the model-level fixtures cannot see this layer (its output is dead in SRGNN.forward)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    samples = [([int(x) for x in s.split(',')], int(l)) for s, l in zip(z['seqs'].tolist(), z['labels'].tolist())]
    layer = om.SRGNNLayer(32, 32)
    layer.load_state_dict({k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('init/')})
    (g,), _ = oc.collate_fn_factory(oc.seq_to_session_graph)(samples)
    x = torch.from_numpy(z['feat']).requires_grad_()
    out = layer(om.to_torch(g), x)
    out.backward(torch.from_numpy(z['gout']))
    close(out, z['out'], what='layer out')
    close(x.grad, z['dfeat'], what='layer d feat', atol=1e-6)
    for k, p in layer.named_parameters():
        close(p.grad, z['grad/' + k], what='layer grad ' + k, atol=2e-6)


@pytest.mark.parametrize('sliced', [False, True])
def test_pinned_ring_loader_yields_the_dataloader_batches(sliced):
    """loader.PinnedRingLoader (workers write native-collated batches into a shared ring) against
    DataLoader(batch_sampler, collate_fn_factory_ccs(caps)): same batches, same order, same buffers, labels included;
    two epochs over a ring smaller than the epoch (slot reuse), one oversized batch (exact-layout fallback)"""
    from torch.utils.data import BatchSampler, DataLoader, SequentialSampler
    C, L, DS = pkg('collate'), pkg('loader'), pkg('dataset')
    if C._native() is None:
        pytest.skip('libsrec_collate.so not built')
    rng = np.random.RandomState(5)
    sessions = [rng.randint(1, 300, size=rng.randint(2, 12)).tolist() for _ in range(150)]
    sessions[40] = rng.randint(1, 300, size=60).tolist()               # long prefixes: some batches overflow the capacities
    ds = DS.AugmentedDataset(sessions)
    bs = 16
    caps = C.default_caps(bs, 12)
    sampler = BatchSampler(SequentialSampler(ds), bs, drop_last=False)
    if sliced:          # rank 0 of 6 of a multi-rank job: its slice of the 3-sample last batch is empty -> a filler sample
        ds.index = ds.index[:len(ds) - (len(ds) - 3) % bs]
        assert len(ds) % bs == 3
        sampler = DS.RankSliceBatchSampler(SequentialSampler(ds), bs, 0, 6)
        caps = C.default_caps(bs // 6 + 1, 61)
    ref = list(DataLoader(ds, batch_sampler=sampler, collate_fn=C.collate_fn_factory_ccs((C.seq_to_ccs_graph,), 3, caps)))
    ld = L.PinnedRingLoader(ds, sampler, 'ccs', order=3, caps=caps, num_workers=2, slots=4)
    try:
        assert len(ld) == len(ref)
        exact = 0
        for epoch in range(2):
            n = 0
            for (inp, lab), (rinp, rlab) in zip(ld, ref):
                fb, rfb = inp[0], rinp[0]
                assert torch.equal(lab, rlab)
                assert fb.layout == rfb.layout and torch.equal(fb.buf, rfb.buf)
                assert fb.meta['counts'] == rfb.meta['counts'] and fb.meta.get('padded') == rfb.meta.get('padded')
                assert fb.meta['max_deg'] == rfb.meta['max_deg'] and fb.meta['max_nodes'] == rfb.meta['max_nodes']
                exact += not fb.meta.get('padded')
                n += 1
            assert n == len(ref)
        assert sliced or exact >= 2                                    # the overflowing batches came through, unpadded
        assert not sliced or int(ref[-1][1][0]) == -1                  # (the filler sample of the last slice)
    finally:
        ld.close()


def test_pinned_ring_loader_survives_an_abandoned_iteration():
    """an iteration left early (break in the consumer) leaves tasks with the workers; the next iteration must start from
    batch 0 with its OWN results (generation tag + drain), not with leftovers of the abandoned one"""
    from torch.utils.data import BatchSampler, DataLoader, SequentialSampler
    C, L, DS = pkg('collate'), pkg('loader'), pkg('dataset')
    if C._native() is None:
        pytest.skip('libsrec_collate.so not built')
    rng = np.random.RandomState(9)
    ds = DS.AugmentedDataset([rng.randint(1, 300, size=rng.randint(2, 12)).tolist() for _ in range(120)])
    bs = 16
    caps = C.default_caps(bs, 12)
    sampler = BatchSampler(SequentialSampler(ds), bs, drop_last=False)
    ref = list(DataLoader(ds, batch_sampler=sampler, collate_fn=C.collate_fn_factory_ccs((C.seq_to_ccs_graph,), 2, caps)))
    ld = L.PinnedRingLoader(ds, sampler, 'ccs', order=2, caps=caps, num_workers=2, slots=6)
    try:
        for stop in (3, 1, 0):
            for k, (inp, lab) in enumerate(ld):
                assert torch.equal(lab, ref[k][1]) and torch.equal(inp[0].buf, ref[k][0][0].buf), (stop, k)
                if k == stop:
                    break                                              # tasks for batches k+1.. are already with the workers
        n = 0
        for (inp, lab), (rinp, rlab) in zip(ld, ref):
            assert torch.equal(lab, rlab) and torch.equal(inp[0].buf, rinp[0].buf), n
            n += 1
        assert n == len(ref) and ld._owed == 0
    finally:
        ld.close()


def test_seeded_init_equals_the_oracle():
    """the product's constructors draw from torch's RNG in the oracle's order: the same seed gives the same initial weights
    (what tests/test_trained_metrics_gpu.py and the launcher comparison with tests/golden/trained_metrics.json rely on)"""
    from oracle import models_ref as om
    sp = pkg()
    V = 500
    for mk_o, mk_p in [(lambda: om.MSGIFSR(V, 'x', 64, 1, order=2, extra=False, fusion=False),
                        lambda: sp.MSGIFSR(V, 'x', 64, 1, order=2, extra=False, fusion=False)),
                       (lambda: om.MSGIFSR(V, 'x', 32, 1, order=3, extra=True, fusion=True),
                        lambda: sp.MSGIFSR(V, 'x', 32, 1, order=3, extra=True, fusion=True)),
                       (lambda: om.SRGNN(V, 64, 2), lambda: sp.SRGNN(V, 64, 2)),
                       (lambda: om.NISER(V, 64, 2), lambda: sp.NISER(V, 64, 2)),
                       (lambda: om.LESSR(V, 32, 3), lambda: sp.LESSR(V, 32, 3))]:
        torch.manual_seed(123)
        a = mk_o().state_dict()
        torch.manual_seed(123)
        b = mk_p().state_dict()
        assert list(a) == list(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_rank_slice_sampler_counts_batches_without_a_long_session():
    """dataset.RankSliceBatchSampler(prefix_len, need_len): global batches in which no session reaches order + 1 clicks (the
    multi-rank path's "every relation is live" assumption, msgifsr.MSHGNN.plan) are counted for the launcher's note"""
    DS = pkg('dataset')
    from torch.utils.data import SequentialSampler
    sessions = [[1, 2, 3, 4, 5]] * 3 + [[7, 8]] * 9
    ds = DS.AugmentedDataset(sessions)
    s = DS.RankSliceBatchSampler(SequentialSampler(ds), 4, 0, 2, prefix_len=ds.index[:, 1], need_len=4)
    out = list(s)
    assert len(out) == len(s) == (len(ds) + 3) // 4
    # samples 0..11 are the prefixes of the three long sessions (one of 4 clicks in every batch of four), the rest 1-click
    assert s.short_batches == len(out) - 3


def test_c4_sized_ingest_index_caps_and_rank_slices(tmp_path):
    """Config C4 (MSGIFSR on Yoochoose-1/4, the table row-sharded over 8 GPUs) on the HOST side at its real size: ~1.3 M
    sessions -> ~6 M prefix samples through dataset.read_dataset(cache=True) (text -> binary SessionStore, then the cached
    store), the vectorised prefix index, collate.measure_caps on the rank slices, dataset.RankSliceBatchSampler (rank 3 of 8
    of every 512-sample batch) and 200 batches of the ring loader - bounded in time and memory (the reference's own path -
    pandas + Python lists + per-sample Python in DataLoader workers - takes minutes and several GB here)."""
    import resource
    import time
    from torch.utils.data import SequentialSampler
    C, L, DS = pkg('collate'), pkg('loader'), pkg('dataset')
    if C._native() is None:
        pytest.skip('libsrec_collate.so not built')
    rng = np.random.default_rng(123)
    n_sess, V = 1_300_000, 37_484
    lens = np.clip(rng.geometric(1 / 4.6, n_sess) + 1, 2, 20)
    offs = np.concatenate([[0], np.cumsum(lens)])
    items = np.minimum((V ** rng.random(int(offs[-1]))).astype(np.int64), V - 1)      # log-uniform ~ Zipf(1) popularity
    d = tmp_path / 'yc14'
    d.mkdir()
    t0 = time.time()
    with open(d / 'train.txt', 'w') as f:
        txt = ','.join(map(str, items.tolist()))
        # (one join over all clicks, then the commas at session boundaries become newlines: fast enough for 6 M clicks)
        digits = np.char.str_len(items.astype(str)) + 1
        ends = np.cumsum(digits)[offs[1:] - 1] - 1
        b = bytearray(txt, 'ascii')
        for e in ends[:-1].tolist():
            b[e] = 10
        f.write(b.decode('ascii') + '\n')
    (d / 'test.txt').write_text('1,2,3\n4,5\n')
    (d / 'num_items.txt').write_text(str(V))
    t_write = time.time() - t0
    t0 = time.time()
    tr, te, nv = DS.read_dataset(d, cache=True)
    t_first = time.time() - t0
    t0 = time.time()
    tr2, _, _ = DS.read_dataset(d, cache=True)
    t_cached = time.time() - t0
    assert nv == V and len(tr) == n_sess == len(tr2) and list(tr[7]) == items[offs[7]:offs[8]].tolist() == list(tr2[7])
    t0 = time.time()
    ds = DS.AugmentedDataset(tr2)
    t_index = time.time() - t0
    n = len(ds)
    assert n == int((lens - 1).sum()) and n > 4_000_000
    seq, lab = ds[n - 1]
    assert lab == int(items[offs[-1] - 1]) and len(seq) == lens[-1] - 1
    world, rank, gb = 8, 3, 512
    bsamp = DS.RankSliceBatchSampler(SequentialSampler(ds), gb, rank, world, prefix_len=ds.index[:, 1], need_len=4)
    t0 = time.time()
    caps = C.measure_caps(ds, gb // world, 'ccs', 3)
    t_caps = time.time() - t0
    assert caps['B'] == gb // world and caps['N'] % 256 == 0
    ld = L.PinnedRingLoader(ds, bsamp, 'ccs', order=3, caps=caps, num_workers=4, slots=8, pin=False)
    t0 = time.time()
    try:
        seen = exact = 0
        for k, (inp, labels) in enumerate(ld):
            mine = bsamp.slice_of(list(range(k * gb, min(n, (k + 1) * gb))))
            assert labels[:len(mine)].tolist() == [int(ds[i][1]) for i in mine]
            assert inp[0].count('B') == len(mine)
            exact += not inp[0].meta.get('padded')
            seen += 1
            if seen == 200:
                break
    finally:
        ld.close()
    t_loader = time.time() - t0
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    print('C4-sized host path: %d sessions, %d samples; text written %.1f s, read_dataset first %.1f s / cached %.2f s, index %.1f s, '
          'measure_caps %.2f s, 200 ring-loader batches %.2f s (%d unpadded), peak RSS %.2f GB' % (
              n_sess, n, t_write, t_first, t_cached, t_index, t_caps, t_loader, exact, rss))
    assert seen == 200 and exact <= 20
    assert t_first < 60 and t_cached < 5 and t_index < 30 and t_caps < 10 and t_loader < 30 and rss < 4.0


def test_padded_batch_ends_with_the_item_list_and_the_labels():
    """capacity-padded batches carry (distinct items | labels) back to back at the end of their buffer - both builders, both
    batch kinds: the row-sharded lookup sends that stretch as its request list without a copy (dist.ShardedLookup)"""
    c = pkg('collate')
    rng = np.random.default_rng(5)
    samples = EDGE + _rand_samples(rng, 30)
    caps = c.default_caps(64, 15)
    for fac in (c.collate_fn_factory_ccs((c.seq_to_ccs_graph,), 3, caps), c.collate_fn_factory(c.seq_to_session_graph, caps=caps)):
        (fb,), lab = fac(samples)
        (o, cap, _), (lo, n, _) = fb.layout['uniq_items'], fb.layout['labels']
        assert cap == caps['U'] and o + cap == lo and lo + ((n + 3) & ~3) == fb.buf.numel() and n == caps['B']
        both = fb.buf[o:lo + n].numpy()
        U = fb.count('U')
        assert np.all(both[:U] >= 0) and np.all(np.diff(both[:U]) > 0) and np.all(both[U:cap] == -1)
        assert np.array_equal(both[cap:], lab.numpy())
        dist = pkg('dist')
        assert dist._follows(fb.uniq_items, fb.field('labels'))
