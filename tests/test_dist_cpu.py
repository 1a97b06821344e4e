"""world_size-2 gloo test of the row-sharded (vocab-parallel) path on CPU: the collectives of
sessionrec-pytorch_amd/dist.py must reassemble exactly what a single device computes on the
concatenated global batch.  The per-rank compute is a plain-torch stand-in for the HIP kernels
(the kernels themselves are covered by the -m gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import pkg


class TorchLocal:
    """plain-torch restatement of dist.HipLocal's interface (test-only)"""

    def localize(self, idx_all, lo, n_loc):
        rel = idx_all - lo
        return torch.where((idx_all >= 0) & (rel >= 0) & (rel < n_loc), rel, torch.full_like(rel, -1)).to(torch.int32)

    def inverse_index(self, uptr, upos, U, n):
        inv = torch.full((n,), -1, dtype=torch.int32)
        for u in range(U):
            for e in range(int(uptr[u]), int(uptr[u + 1])):
                inv[int(upos[e])] = u
        return inv

    def gather_masked(self, table, idx):
        idx = idx.long()
        out = table[idx.clamp(min=0)]
        return out * (idx >= 0).unsqueeze(1)

    def segment_rows(self, g, uniq):
        items, uptr, upos = uniq[:3]
        U = items.numel()
        out = torch.zeros(U, g.shape[1])
        for u in range(U):
            for e in range(int(uptr[u]), int(uptr[u + 1])):
                out[u] += g[int(upos[e])]
        return out

    def add_rows(self, rows, items_local, dst):
        m = items_local >= 0
        dst.index_add_(0, items_local[m].long(), rows[m])

    def _z(self, sr, table, cs):
        z = sr @ table.t()
        return z if cs is None else z * cs.unsqueeze(0)

    def ce_fwd(self, sr, table, cs, labels_local, ws):
        z = self._z(sr, table, cs)
        lse = torch.logsumexp(z, dim=1)
        lab = torch.zeros(sr.shape[0])
        m = labels_local >= 0
        lab[m] = z[m, labels_local[m].long()]
        return lse, lab

    def ce_bwd(self, sr, table, cs, labels_local, lse, gscale, dE, ws, cs_inv_scale):
        z = self._z(sr, table, cs)
        p = torch.exp(z - lse.unsqueeze(1))
        m = labels_local >= 0
        p[m, labels_local[m].long()] -= 1.0
        p = p * (gscale / sr.shape[0])
        if cs is not None:
            p = p * cs.unsqueeze(0)
        g = p.t() @ sr
        if cs is not None:                      # chain rule of the row normalisation (srec_rownorm_project)
            e = table * (cs * cs_inv_scale).unsqueeze(1)
            g = g - e * (e * g).sum(1, keepdim=True)
        dE.copy_(g)
        return p @ table

    def stats_bwd(self, sr, table, cs, labels_local, lse, ga, gc, dE, ws, cs_inv_scale, accumulate):
        z = self._z(sr, table, cs)
        dz = torch.exp(z - lse.unsqueeze(1)) * ga.unsqueeze(1)
        m = labels_local >= 0
        dz[m, labels_local[m].long()] -= gc[m]
        if cs is not None:
            dz = dz * cs.unsqueeze(0)
        g = dz.t() @ sr
        g = dE + g if accumulate else g
        if cs is not None:
            e = table * (cs * cs_inv_scale).unsqueeze(1)
            g = g - e * (e * g).sum(1, keepdim=True)
        dE.copy_(g)
        return dz @ table

    def logp_cols(self, sr, table, cs, lse):
        return self._z(sr, table, cs) - lse.unsqueeze(1)

    def topk(self, sr, table, cs, k):
        z = self._z(sr, table, cs)
        o = torch.argsort(z, dim=1, descending=True, stable=True)[:, :k]      # stable: ties -> lower id
        return z.gather(1, o), o.int()

    def workspace(self, B, V, d, device):
        return None


class FakeModel:
    def __init__(self, table):
        self.w = torch.nn.Parameter(table.clone())
        self.shard = None

    def _table(self):
        return self.w

    def _state(self, B):
        ops = pkg('ops')
        st = self.__dict__.setdefault('_srec_state', {})
        if 'tgrad' not in st:
            st['tgrad'] = ops.TableGrad(self.w)
        return st


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(world, V=150, d=16, B=6):
    g = torch.Generator().manual_seed(5)
    table = torch.randn(V, d, generator=g) * 0.3
    table[130] = table[7]                                 # an exact score tie across the two shards
    table[9] = table[8]                                   # ... and inside one shard
    per_rank = []
    for r in range(world):
        n = 9 + r * 2
        idx = torch.randint(0, V, (n,), generator=g)
        idx[0] = V - 1                                   # a row of the last shard, from every rank
        sr_w = torch.randn(d, d, generator=g) * 0.3
        labels = torch.randint(0, V, (B,), generator=g)
        pick = torch.randint(0, n, (B,), generator=g)
        gout = torch.randn(n, d, generator=g)
        per_rank.append(dict(idx=idx, sr_w=sr_w, labels=labels, pick=pick, gout=gout))
    sr_eval = torch.randn(8, d, generator=g)
    sr_eval[0] = table[7] * 5                             # session 0 ranks the tied pair first
    sr_eval[1] = table[8] * 5
    per_rank[0]['sr_eval'] = sr_eval
    return table, per_rank


def _uniq(idx):
    items, inv = torch.unique(idx, return_inverse=True)
    pos = torch.argsort(idx, stable=True).int()
    ptr_ = torch.zeros(items.numel() + 1, dtype=torch.int32)
    ptr_[1:] = torch.bincount(inv).cumsum(0).int()
    return items.int(), ptr_, pos


def _reference(table, per_rank, cosine):
    """single device, global batch = concatenation of the ranks' batches"""
    W = table.clone().requires_grad_()
    srs, labs, extra = [], [], 0.0
    for b in per_rank:
        rows = W[b['idx']]
        srs.append(rows[b['pick']] @ b['sr_w'])
        labs.append(b['labels'])
        extra = extra + (rows * b['gout']).sum()
    sr = torch.cat(srs)
    lab = torch.cat(labs)
    Wn = torch.nn.functional.normalize(W, dim=1) * 12.0 if cosine else W
    loss = torch.nn.functional.cross_entropy(sr @ Wn.t(), lab)
    (loss + 1e-3 * extra).backward()
    return loss.item(), W.grad


def _worker(rank, world, port, cosine, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        D = pkg('dist')
        table, per_rank = _make(world)
        b = per_rank[rank]
        model = FakeModel(table)
        vp = D.VocabParallel(model, local=TorchLocal())
        shard = model._table()
        cs = None
        if cosine:
            cs = 12.0 / shard.detach().norm(dim=1).clamp(min=1e-12)
        vp.labels_hint = b['labels']                       # labels ride along with the lookup's integer exchange
        rows = vp.lookup(shard, b['idx'].int(), _uniq(b['idx']))
        assert vp.labels_hint is None and vp.lab_all is not None and vp.lab_all.numel() == world * b['labels'].numel()
        sr = rows[b['pick']] @ b['sr_w']
        loss = vp.loss(sr, shard, cs, b['labels'], 1.0 / 12.0)
        assert vp.lab_all is None
        (loss + 1e-3 * (rows * b['gout']).sum()).backward()
        dE = vp.dE[:vp.n_live].clone()
        # evaluation: local top-k per shard -> all-gather -> merge, both feeding conventions
        with torch.no_grad():
            sr_same = per_rank[0]['sr_eval']
            v_rep, i_rep = vp.topk(sr_same, shard, cs, 5)
            n = sr_same.shape[0] // world
            v_dp, i_dp = vp.topk(sr_same[rank * n:(rank + 1) * n], shard, cs, 5, data_parallel=True)
        q.put((rank, loss.item(), vp.lo, vp.hi, dE.numpy().tolist(), v_rep.tolist(), i_rep.tolist(), v_dp.tolist(),
               i_dp.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cosine', [False, True])
def test_vocab_parallel_two_ranks_match_single_device(cosine):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cosine, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table, per_rank = _make(world)
    ref_loss, ref_grad = _reference(table, per_rank, cosine)
    sr_eval = per_rank[0]['sr_eval']
    Wn = torch.nn.functional.normalize(table, dim=1) * 12.0 if cosine else table
    z = sr_eval @ Wn.t()
    o = torch.argsort(z, dim=1, descending=True, stable=True)[:, :5]
    n = sr_eval.shape[0] // world
    for rank, loss, lo, hi, dE, v_rep, i_rep, v_dp, i_dp in res:
        if not cosine:                           # exact ties survive only without the per-row scale round-off
            assert i_rep[0][:2] == [7, 130] and i_rep[1][:2] == [8, 9]
            assert torch.equal(torch.tensor(i_rep), o.int())
            assert torch.equal(torch.tensor(i_dp), o[rank * n:(rank + 1) * n].int())
        assert torch.allclose(torch.tensor(v_rep), z.gather(1, o), rtol=1e-5, atol=1e-5)
        assert torch.allclose(torch.tensor(v_dp), z.gather(1, o)[rank * n:(rank + 1) * n], rtol=1e-5, atol=1e-5)
        dE = torch.tensor(dE)
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
        assert torch.allclose(dE, ref_grad[lo:hi], rtol=1e-4, atol=1e-6), (rank, (dE - ref_grad[lo:hi]).abs().max())


def test_shard_bounds_cover_the_catalog():
    D = pkg('dist')
    for V in (1, 63, 64, 65, 3429, 37484, 10_000_000):
        for world in (1, 2, 4, 8):
            seen = 0
            for r in range(world):
                lo, hi, per = D.shard_bounds(V, world, r)
                assert lo == min(V, r * per) and hi - lo <= per and per % 64 == 0
                seen += hi - lo
            assert seen == V


# ---------------------------------------------------------------------- mixtures of soft-maxes over the sharded table
def _mix_reference(table, per_rank, cosine, alpha):
    W = table.clone().requires_grad_()
    Wn = torch.nn.functional.normalize(W, dim=1) * 12.0 if cosine else W
    la = torch.log_softmax(alpha, 0)
    terms, labs = [], torch.cat([b['labels'] for b in per_rank])
    for h in range(2):
        sr = torch.cat([W[b['idx']][b['pick']] @ (b['sr_w'] * (1.0 + h)) for b in per_rank])
        terms.append(torch.log_softmax(sr @ Wn.t(), 1).gather(1, labs[:, None])[:, 0] + la[h])
    loss = -torch.logsumexp(torch.stack(terms, 1), 1).mean()
    loss.backward()
    sr_eval = per_rank[0]['sr_eval']
    logp = torch.log_softmax(sr_eval @ Wn.detach().t(), 1)
    return loss.item(), W.grad, logp


def _mix_worker(rank, world, port, cosine, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        D = pkg('dist')
        table, per_rank = _make(world)
        b = per_rank[rank]
        model = FakeModel(table)
        vp = D.VocabParallel(model, local=TorchLocal())
        shard = model._table()
        cs = 12.0 / shard.detach().norm(dim=1).clamp(min=1e-12) if cosine else None
        rows = vp.lookup(shard, b['idx'].int(), _uniq(b['idx']))
        la = torch.log_softmax(torch.tensor([0.3, -0.2]), 0)
        vp.tgrad.fresh = False
        terms = []
        for h in range(2):
            sr = rows[b['pick']] @ (b['sr_w'] * (1.0 + h))
            lse, lab = vp.stats(sr, shard, cs, b['labels'], 1.0 / 12.0)
            terms.append(lab - lse + la[h])
        B = b['labels'].numel()
        loss = -torch.logsumexp(torch.stack(terms, 1), 1).sum() / (B * world)
        loss.backward()
        total = loss.detach().clone()
        dist.all_reduce(total)                                # each rank holds its share of the global mean
        logp = vp.log_probs(per_rank[0]['sr_eval'], shard, cs)
        n = 8 // world
        logp_dp = vp.log_probs(per_rank[0]['sr_eval'][rank * n:(rank + 1) * n], shard, cs, data_parallel=True)
        q.put((rank, total.item(), vp.lo, vp.hi, vp.dE[:vp.n_live].clone().numpy().tolist(), logp.numpy().tolist(),
               logp_dp.numpy().tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('cosine', [False, True])
def test_sharded_softmax_mixture_and_logprobs_match_single_device(cosine):
    """ShardedScoreStats (two heads sharing the table, second backward accumulates) and VocabParallel.log_probs"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mix_worker, args=(r, world, port, cosine, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table, per_rank = _make(world)
    ref_loss, ref_grad, ref_logp = _mix_reference(table, per_rank, cosine, torch.tensor([0.3, -0.2]))
    n = 8 // world
    for rank, loss, lo, hi, dE, logp, logp_dp in res:
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
        dE = torch.tensor(dE)
        assert torch.allclose(dE, ref_grad[lo:hi], rtol=1e-4, atol=1e-6), (rank, (dE - ref_grad[lo:hi]).abs().max())
        logp = torch.tensor(logp)
        assert logp.shape == ref_logp.shape and torch.allclose(logp, ref_logp, rtol=1e-5, atol=1e-5)
        assert torch.allclose(torch.tensor(logp_dp), ref_logp[rank * n:(rank + 1) * n], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------- partial batches, gradient bucket, launcher
def _partial_worker(rank, world, port, q):
    """rank 1's batch has dead (capacity-padding) sessions: label -1, left out of the global mean"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        D = pkg('dist')
        table, per_rank = _make(world)
        b = per_rank[rank]
        labels = b['labels'].clone()
        if rank == 1:
            labels[-2:] = -1
        model = FakeModel(table)
        vp = D.VocabParallel(model, local=TorchLocal())
        shard = model._table()
        vp.labels_hint = labels
        rows = vp.lookup(shard, b['idx'].int(), _uniq(b['idx']))
        sr = rows[b['pick']] @ b['sr_w']
        loss = vp.loss(sr, shard, None, labels, 1.0)
        loss.backward()
        q.put((rank, loss.item(), vp.lo, vp.hi, vp.dE[:vp.n_live].clone().numpy().tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_loss_ignores_capacity_padding_of_a_partial_batch():
    """the tail batch of an epoch: ranks hold different live counts inside equal padded layouts; the loss is the mean over
    the live sessions of the GLOBAL batch and padded sessions contribute no gradient (train.py:94-101 on one device)"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partial_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    table, per_rank = _make(world)
    W = table.clone().requires_grad_()
    srs, labs = [], []
    for r, b in enumerate(per_rank):
        keep = slice(None) if r == 0 else slice(0, b['labels'].numel() - 2)
        srs.append((W[b['idx']][b['pick']] @ b['sr_w'])[keep])
        labs.append(b['labels'][keep])
    ref = torch.nn.functional.cross_entropy(torch.cat(srs) @ W.t(), torch.cat(labs))
    ref.backward()
    for rank, loss, lo, hi, dE in res:
        assert abs(loss - ref.item()) < 1e-5 * max(1.0, abs(ref.item())), (loss, ref.item())
        dE = torch.tensor(dE)
        assert torch.allclose(dE, W.grad[lo:hi], rtol=1e-4, atol=1e-6), (rank, (dE - W.grad[lo:hi]).abs().max())


def _bucket_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        D = pkg('dist')
        table, _ = _make(world)
        vp = D.VocabParallel(FakeModel(table), local=TorchLocal())
        ps = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, 2)), torch.nn.Parameter(torch.zeros(4)),
              torch.nn.Parameter(torch.zeros(5))]
        # p0: gradient on both ranks; p1: only on rank 0; p2: only on rank 1; p3: on neither (never enters the bucket)
        ps[0].grad = torch.full((3,), 1.0 + rank)
        if rank == 0:
            ps[1].grad = torch.full((2, 2), 10.0)
        else:
            ps[2].grad = torch.full((4,), 100.0)

        class Opt:
            grad_override = None
        opt = Opt()
        vp.sync_replicated_grads(ps, opt)
        got = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps) if id(p) in opt.grad_override}
        # second step: the pattern flips (rank 0 now lacks p0) - same bucket, still consistent
        ps[0].grad = None if rank == 0 else torch.full((3,), 7.0)
        vp.sync_replicated_grads(ps, opt)
        got2 = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps) if id(p) in opt.grad_override}
        # a parameter outside the agreed layout that receives a gradient later, on ONE rank only: every rank sees the
        # all-reduced "late" count in the bucket's flag tail, re-agrees on the layout and repeats the exchange (no hang,
        # nothing dropped)
        if rank == 1:
            ps[3].grad = torch.ones(5)
        vp.sync_replicated_grads(ps, opt)
        got3 = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps) if id(p) in opt.grad_override}
        # a bucketed parameter nobody has a gradient for this step is skipped (one device: grad None -> no update)
        ps[1].grad = None
        vp.sync_replicated_grads(ps, opt)
        got4 = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps) if id(p) in opt.grad_override}
        q.put((rank, {k: v.tolist() for k, v in got.items()}, {k: v.tolist() for k, v in got2.items()},
               {k: v.tolist() for k, v in got3.items()}, sorted(got4)))
    finally:
        dist.destroy_process_group()


def test_replicated_gradient_bucket_is_rank_independent():
    """ranks hold gradients for different parameter subsets (MSHGNN instantiates only the GAT modules of relations with
    live edges in ITS batch): the flat bucket must have one layout everywhere, missing gradients contributing zeros"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, got2, got3, keys4 in res:
        assert sorted(got) == [0, 1, 2], got
        assert got[0] == [3.0] * 3 and got[1] == [[10.0, 10.0]] * 2 and got[2] == [100.0] * 4
        assert got2[0] == [7.0] * 3 and got2[1] == [[10.0, 10.0]] * 2
        assert sorted(got3) == [0, 1, 2, 3] and got3[3] == [1.0] * 5 and got3[0] == [7.0] * 3, got3
        assert keys4 == [0, 2, 3], keys4


def _accum_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        D = pkg('dist')
        ops = pkg('ops')
        table, _ = _make(world)
        vp = D.VocabParallel(FakeModel(table), local=TorchLocal())
        ps = [torch.nn.Parameter(torch.zeros(n)) for n in (3, 4, 5, 6)]     # default partition: p3, p2 -> bucket 0; p1 -> 1; p0 -> 2

        class Opt:
            grad_override = None
        opt = Opt()

        def backward(scale, marks=True):
            """what a backward pass does: every node asks ops.grad_buf for its destination, autograd accumulates, the model's
            markers fire when a bucket is complete"""
            for i, p in enumerate(ps):
                g = ops.grad_buf(p)
                g.copy_(torch.full_like(p, scale * (i + 1) * (rank + 1)))
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad += g
            if marks:
                vp.bucket_ready(0)
                vp.bucket_ready(1)

        def zero():
            for p in ps:
                p.grad = None
        # step 1: agrees on the layout (no arena yet: every gradient is copied in)
        vp.begin_step()
        backward(1.0)
        vp.sync_replicated_grads(ps, opt)
        one = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps)}
        zero()
        # step 2: the gradients are written INTO the arena slots (nothing to copy but the flag tail), buckets 0 and 1 leave early
        vp.begin_step()
        backward(1.0)
        resident = [ps[i].grad.data_ptr() == vp._arena[vp._slot[id(ps[i])][0]][vp._slot[id(ps[i])][1]:].data_ptr() for i in range(4)]
        early = sorted(vp._bucket_early)
        vp.sync_replicated_grads(ps, opt)
        two = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps)}
        copies = dict(vp.copy_tasks_last)
        zero()
        # step 3: two micro-batches, the first under no_sync(): ONE exchange of the accumulated sums
        with vp.no_sync():
            vp.begin_step()
            backward(1.0)
            none_early = dict(vp._bucket_early)
        vp.begin_step()
        backward(10.0)
        vp.sync_replicated_grads(ps, opt)
        acc = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps)}
        zero()
        # step 4: misuse - a second backward on top of early-reduced gradients, no no_sync(): loud on every rank
        vp.begin_step()
        backward(1.0)
        try:
            vp.begin_step()
            raised = False
        except RuntimeError as e:
            raised = 'no_sync' in str(e)
        # ... and an aborted step (gradients dropped) is forgotten quietly; the early all-reduces it issued were issued by
        # every rank, so the next step's collectives still pair up
        zero()
        vp.begin_step()
        backward(2.0)
        vp.sync_replicated_grads(ps, opt)
        rec = {i: opt.grad_override[id(p)].clone() for i, p in enumerate(ps)}
        q.put((rank, {k: v.tolist() for k, v in one.items()}, {k: v.tolist() for k, v in two.items()}, resident, early, copies,
               len(none_early), {k: v.tolist() for k, v in acc.items()}, raised, {k: v.tolist() for k, v in rec.items()}))
    finally:
        dist.destroy_process_group()


def test_gradient_arena_accumulation_and_early_launch_protocol():
    """the early bucket launches depend on rank-agreed state only, micro-batch accumulation goes through
    no_sync(), a backward on top of all-reduced gradients raises instead of mixing them, an aborted step leaves nothing
    behind; and the gradients land in the bucket arenas without a concatenation pass"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tot = sum(r + 1 for r in range(world))
    for rank, one, two, resident, early, copies, n_none, acc, raised, rec in res:
        for i, n in enumerate((3, 4, 5, 6)):
            assert one[i] == [float((i + 1) * tot)] * n, (rank, one)
            assert two[i] == one[i], (rank, two)
            assert acc[i] == [float(11 * (i + 1) * tot)] * n, (rank, acc)
            assert rec[i] == [float(2 * (i + 1) * tot)] * n, (rank, rec)
        assert resident == [True] * 4 and early == [0, 1], (resident, early)
        assert copies == {0: 0, 1: 0, 2: 1}, copies           # only the flag tail of the last bucket is copied (eager step)
        assert n_none == 0 and raised


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must spawn two ranks itself (the flag used to
    be parsed and ignored).  --launch-only stops after the process group is up (gloo on a box without GPUs)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--launch-only'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == 2 and rec['world_size'] == 2, rec


def test_rank_slice_sampler_partitions_the_reference_batches():
    ds = pkg('dataset')
    from torch.utils.data import SequentialSampler
    n, bs = 23, 8
    for w in (1, 2, 3, 8):
        per_rank = [list(ds.RankSliceBatchSampler(SequentialSampler(range(n)), bs, r, w)) for r in range(w)]
        nb = (n + bs - 1) // bs
        assert all(len(x) == nb for x in per_rank)
        for b in range(nb):
            ref = list(range(b * bs, min(n, (b + 1) * bs)))
            live = [i for r in range(w) for i in per_rank[r][b] if i >= 0]
            assert live == ref, (w, b, live, ref)               # contiguous slices in rank order = the reference batch
            for r in range(w):
                assert len(per_rank[r][b]) >= 1                  # every rank takes part in every step
                assert len(per_rank[r][b]) <= (bs + w - 1) // w
                assert all((-i - 1) in ref for i in per_rank[r][b] if i < 0)
    # a filler index returns the same prefix with label -1
    import numpy as np
    sessions = np.empty(2, dtype=object)
    sessions[:] = [[1, 2, 3], [4, 5]]
    data = ds.AugmentedDataset(sessions)
    seq, lab = data[-2 - 1]
    seq0, _ = data[2]
    assert lab == -1 and list(seq) == list(seq0)


def test_estimate_caps_for_rank_slices_bounds_every_slice():
    ds, col = pkg('dataset'), pkg('collate')
    import numpy as np
    rng = np.random.default_rng(3)
    sessions = np.empty(40, dtype=object)
    sessions[:] = [rng.integers(0, 50, size=int(rng.integers(2, 15))).tolist() for _ in range(40)]
    data = ds.AugmentedDataset(sessions)
    gb, w = 16, 4
    caps = col.estimate_caps(data, (gb + w - 1) // w, headroom=1.0, slices=(gb, w))
    lens = data.index[:, 1]
    from torch.utils.data import SequentialSampler
    for r in range(w):
        for sl in ds.RankSliceBatchSampler(SequentialSampler(data), gb, r, w):
            clicks = sum(int(lens[i if i >= 0 else -i - 1]) for i in sl)
            assert clicks <= caps['N'] and len(sl) <= caps['B']
