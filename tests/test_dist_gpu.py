"""The row-sharded (vocab-parallel) path with MORE THAN ONE shard, on the one GPU a test box has: W rank processes on
cuda:0 over gloo, the real HIP kernels on every rank (tests/dist_gpu_worker.py), against the plain single-GPU path on
the concatenated global batch - the single-device step of /root/reference/src/utils/train.py:94-101.

Checked per rank: the global mean loss, this rank's rows of the table gradient, the all-reduced gradients of the
replicated encoder, every parameter after the fused Adam step, the data-parallel top-20 and (fixtures) the
log-probabilities.  Then one rank of the same job is re-run alone from the recorded collective results
(dist.ReplayGroup) - eagerly and as a captured + replayed hipGraph - and must land on the parameters it reached inside
the job: the kernels of a rank with shard offset > 0 under graph capture."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from dist_gpu_worker import live_samples, make_case, rank_slice, run_rank
from test_models_gpu import adam_close
from util import close, pkg

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, case, tmp_path):
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, case, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    assert codes == [0] * world, 'rank exit codes %r' % (codes,)
    return [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r), weights_only=False) for r in range(world)]


def _plain(case, world, dev, steps):
    """the single-device path on the global batch (live sessions only)"""
    ops, train, optim = pkg('ops'), pkg('train'), pkg('optim')
    ops.set_precision(case.get('precision', 'fp32'))
    build, collate, samples, V = make_case(case)
    live = live_samples(samples, world, bool(case.get('partial')))
    inputs, labels = collate(None)(live)
    inputs, labels = [x.to(dev) for x in inputs], labels.to(dev)
    model = build().to(dev)
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model, fuse_projection=True)
    model.train()
    out = []
    for step in range(steps):
        opt.zero_grad()
        loss = model.fused_loss(*inputs, labels)
        loss.backward()
        rec = dict(loss=float(loss.item()))
        if step == 0:
            rec['dE'] = model.table_grad.buf.detach().cpu().clone()
            rec['grads'] = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()
                            if p.grad is not None and p is not model._table()}
        opt.step()
        rec['table'] = model._table().detach().cpu().clone()
        rec['params'] = {k: p.detach().cpu().clone() for k, p in model.named_parameters() if p is not model._table()}
        out.append(rec)
    model.eval()
    extra = {}
    if not (getattr(model, 'extra', False) or (getattr(model, 'fusion', False) and getattr(model, 'order', 1) > 1)):
        v, i = model.topk(*inputs, k=20)
        extra['topk'] = (v.cpu(), i.cpu())
    if case['kind'] == 'fixture':
        with torch.no_grad():
            extra['logp'] = model(*inputs).cpu()
    return out, extra, live


def _compare(case, world, res, ref, extra, live, gtol=1e-4, ltol=1e-5, bf16=False):
    samples = make_case(case)[2]
    off = 0
    for r, out in enumerate(res):
        lo, hi = out['lo'], out['hi']
        assert out['n_live'] == hi - lo and (r == 0) == (lo == 0)
        s0, r0 = out['steps'][0], ref[0]
        assert abs(s0['loss'] - r0['loss']) <= ltol * max(1.0, abs(r0['loss'])), (r, s0['loss'], r0['loss'])
        if bf16:
            # different item tilings per shard: bf16 products are the same, fp32 accumulation order differs
            rel = (s0['dE'] - r0['dE'][lo:hi]).norm() / r0['dE'][lo:hi].norm().clamp(min=1e-20)
            assert rel < 2e-3, ('table grad', r, float(rel))
        else:
            close(s0['dE'], r0['dE'][lo:hi], rtol=gtol, atol=1e-7, what='rank %d: table gradient rows [%d, %d)' % (r, lo, hi))
        for k, g in r0['grads'].items():
            assert k in s0['grads'], 'rank %d: replicated gradient %s missing from the bucket' % (r, k)
            if bf16:
                rel = (s0['grads'][k] - g).norm() / g.norm().clamp(min=1e-20)
                assert rel < 5e-3, (k, r, float(rel))
            else:
                close(s0['grads'][k], g, rtol=gtol, atol=1e-7, what='rank %d: all-reduced gradient of %s' % (r, k))
        for si, (s, rr) in enumerate(zip(out['steps'], ref)):
            assert abs(s['loss'] - rr['loss']) <= (5e-4 if bf16 else ltol * 4) * max(1.0, abs(rr['loss'])), (r, si, s['loss'], rr['loss'])
            if not bf16:
                adam_close(s['table'], rr['table'][lo:hi], steps=si + 1, what='rank %d step %d: table rows' % (r, si))
                for k, p in rr['params'].items():
                    adam_close(s['params'][k], p, steps=si + 1, what='rank %d step %d: %s' % (r, si, k))
        n_live = len(rank_slice(samples, world, r, bool(case.get('partial')))[0])
        if 'topk' in extra and 'topk' in out:
            v, i = out['topk']
            rv, ri = extra['topk']
            if not bf16:
                same = (i[:n_live] == ri[off:off + n_live]).float().mean().item()
                assert same > 0.995, 'rank %d: top-20 ids agree on %.4f' % (r, same)     # (an fp32 near-tie may swap two neighbours)
                close(v[:n_live], rv[off:off + n_live], rtol=1e-5, atol=1e-5, what='rank %d: top-20 scores' % r)
        if 'logp' in extra and 'logp' in out and not bf16:
            close(out['logp'][:n_live], extra['logp'][off:off + n_live], rtol=1e-4, atol=1e-4, what='rank %d: log-probs' % r)
        off += n_live
    assert off == len(live)


FIXTURE_CASES = [
    ('msgifsr_K3_s32', 2, {}),
    ('msgifsr_K3_s32', 8, {}),
    ('msgifsr_K3_fus_s32', 2, {}),
    ('msgifsr_K3_ext_s32', 2, {}),
    ('msgifsr_K3_edge', 2, {}),
    ('niser_s32', 2, {}),
    ('msgifsr_K3_s32', 2, dict(partial=True)),                       # exact layouts: filler sessions with label -1
    ('msgifsr_K3_s32', 8, dict(partial=True, padded=True)),          # capacity-padded layouts: fewer live sessions
    ('msgifsr_K3_s32', 2, dict(dead=True)),                          # relations without edges on one rank
]


@pytest.mark.parametrize('name,world,opts', FIXTURE_CASES)
def test_sharded_ranks_on_one_gpu_match_the_single_device_step(dev, tmp_path, name, world, opts):
    case = dict(kind='fixture', name=name, steps=2, **opts)
    res = _launch(world, case, tmp_path)
    ref, extra, live = _plain(case, world, dev, 2)
    _compare(case, world, res, ref, extra, live)
    # the exchange really happened: > 0 collectives per step, the same count on every rank
    counts = {out['steps'][1]['collectives']['count'] for out in res}
    assert len(counts) == 1 and min(counts) >= 5, counts
    if opts.get('dead'):
        # the short-session rank has no edges of the higher-order relations; in a multi-rank job every relation of the schema
        # counts as live (it is, in the global batch: msgifsr.MSHGNN.plan), so its nodes still get those relations' residual
        # and bias terms and the rank still lands on the single-device parameters (checked by _compare)
        assert any('intra3' in k for k in ref[0]['grads'])


@pytest.mark.parametrize('precision,world', [('fp32', 2), ('bf16', 2), ('bf16', 8)])
def test_sharded_ranks_at_the_benchmarked_shape(dev, tmp_path, precision, world):
    """config C3's shape (V = 37 484, d = 256, order 3, 512 sessions in total) split over W row shards"""
    case = dict(kind='synth', V=37484, d=256, order=3, B=512, precision=precision, steps=1, padded=True)
    try:
        res = _launch(world, case, tmp_path)
        ref, extra, live = _plain(case, world, dev, 1)
        _compare(case, world, res, ref, extra, live, bf16=(precision == 'bf16'))
    finally:
        pkg('ops').set_precision('fp32')


@pytest.mark.parametrize('name,world,rank', [('msgifsr_K3_s32', 2, 1), ('msgifsr_K3_s32', 8, 5), ('niser_s32', 2, 1)])
def test_one_rank_of_the_job_replayed_under_hipgraph_capture(dev, tmp_path, name, world, rank):
    """rank `rank` of a W-rank job, re-run alone from the recorded collective results: step 1 eagerly (every collective
    input is checked against what the rank handed in inside the job), step 2 as a captured and replayed hipGraph -
    the launch mode of bench.py / TrainRunner - and the rank must land where it landed inside the job."""
    D, G, train, optim = pkg('dist'), pkg('graph'), pkg('train'), pkg('optim')
    case = dict(kind='fixture', name=name, steps=2, padded=True, record=True)
    res = _launch(world, case, tmp_path)
    job = res[rank]
    assert job['lo'] > 0
    build, collate, samples, V = make_case(case)
    mine, n = rank_slice(samples, world, rank, False)
    caps = pkg('collate').default_caps(n, 20)
    inputs, labels = collate(caps)(mine)
    inputs, labels = [x.to(dev) for x in inputs], labels.to(dev)
    model = build().to(dev)
    group = D.ReplayGroup(world, rank, dev).load(job['steps'][0]['tape'])
    vp = D.VocabParallel(model, group=group, idx_cap=inputs[0].cap('uniq_items'))
    assert (vp.lo, vp.hi) == (job['lo'], job['hi'])
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model, fuse_projection=True)
    replicated = [p for p in model.parameters() if p is not model._table() and p.requires_grad]
    model.train()
    # step 1, eager
    opt.zero_grad()
    loss = model.fused_loss(*inputs, labels)
    loss.backward()
    vp.sync_replicated_grads(replicated, opt)
    opt.step()
    assert group.pos == len(group.kinds) and group.checked == len(group.kinds)
    assert abs(loss.item() - job['steps'][0]['loss']) <= 1e-6 * max(1.0, abs(job['steps'][0]['loss']))
    del loss          # (a live loss of an eager default-stream step must not outlive into the capture: graph.GraphedTrainStep)
    close(model._table().detach()[:vp.n_live], job['steps'][0]['table'], rtol=1e-6, atol=1e-7, what='table rows after the eager step')
    # step 2, captured (one eager warm-up lap with checks, one capture lap) and replayed
    group.load(job['steps'][1]['tape'])
    gs = G.GraphedTrainStep(model, opt, inputs, labels, after_backward=lambda: vp.sync_replicated_grads(replicated, opt),
                            warmup=1)
    assert group.pos == 2 * len(group.kinds) and group.checked == len(group.kinds)
    loss2 = gs(inputs, labels)
    torch.cuda.synchronize()
    assert abs(loss2.item() - job['steps'][1]['loss']) <= 1e-6 * max(1.0, abs(job['steps'][1]['loss'])), \
        (loss2.item(), job['steps'][1]['loss'])
    close(model._table().detach()[:vp.n_live], job['steps'][1]['table'], rtol=1e-6, atol=1e-7, what='table rows after the replayed step')
    for k, p in model.named_parameters():
        if p is not model._table():
            close(p, job['steps'][1]['params'][k], rtol=1e-6, atol=1e-7, what='replayed step: ' + k)
    nodes = gs.node_counts()
    if nodes is not None:
        assert nodes['kernel'] > 10
