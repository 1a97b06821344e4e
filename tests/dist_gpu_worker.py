"""Worker side of tests/test_dist_gpu.py: W processes - one per rank, as in a real job - that all sit on cuda:0 and
talk over gloo (device tensors staged through the host by sessionrec-pytorch_amd/dist.py).  Every rank runs the REAL
per-rank compute (dist.HipLocal: the HIP kernels behind the C ABI) on its row shard of the item table, so the
kernels see a shard offset > 0, item ids owned by other ranks (srec_localize_idx -> -1, masked gathers), labels
that live on another shard and w > 1 in srec_merge_stats - none of which a 1-rank world exercises.

Reference for what is being reproduced: the single-device training step of /root/reference/src/utils/train.py:94-101
on the GLOBAL batch (the concatenation of the ranks' batches); the parent test computes that with the plain
single-GPU path and compares."""
import os

import numpy as np
import torch
import torch.distributed as dist

from util import load_golden, pkg


def synth_samples(n, V, seed, max_len=20, mean_len=6.2):
    """prefix samples of Yoochoose-like synthetic sessions (bench.py's generator, smaller)"""
    rng = np.random.default_rng(seed)
    p = 1.0 / (mean_len - 1.0)
    w = 1.0 / np.arange(1, V + 1)
    cdf = np.cumsum(w / w.sum())
    out = []
    while len(out) < n:
        L = int(np.clip(rng.geometric(p) + 1, 2, max_len))
        ids = np.searchsorted(cdf, rng.random(L)).clip(0, V - 1)
        rev = rng.random(L) < 0.2
        for i in range(1, L):
            if rev[i]:
                ids[i] = ids[i - 1]
        ids = ids.tolist()
        for e in range(1, L):
            out.append((ids[:e], int(ids[e])))
    return out[:n]


def make_case(case):
    """-> (model factory, collate factory(caps), global sample list, V).  Shared by the workers and the parent."""
    sp, col = pkg(), pkg('collate')
    kind = case['kind']
    if kind == 'fixture':
        name = case['name']
        z, samples, init = load_golden(name)
        V = init[[k for k in init if k.startswith('embedding')][0]].shape[0]
        K = int(name.split('_')[1][1:]) if name.startswith('msgifsr') else 1

        def build():
            if name.startswith('niser'):
                m = sp.NISER(V, 32, 1)
            elif name.startswith('srgnn'):
                m = sp.SRGNN(V, 32, 1)
            else:
                m = sp.MSGIFSR(V, 'sample', 32, 1, order=K, extra='_ext' in name, fusion='_fus' in name)
            m.load_state_dict(init)
            return m
        if case.get('dead'):
            # rank 0's half: long sessions (every relation of order 3 has edges); the other half: sessions of <= 2 clicks
            # (no 3-grams, no intra2 / intra3 edges) -> on those ranks the GAT modules of the missing relations get no gradient
            rng = np.random.default_rng(7)
            long_ = [(rng.integers(0, V, size=int(rng.integers(5, 12))).tolist(), int(rng.integers(0, V))) for _ in range(8)]
            short = [(rng.integers(0, V, size=int(rng.integers(1, 3))).tolist(), int(rng.integers(0, V))) for _ in range(8)]
            samples = long_ + short
    else:                                              # 'synth': MSGIFSR at a synthetic shape (C3: V 37484, d 256, order 3)
        V, d, K = case['V'], case['d'], case['order']
        samples = synth_samples(case['B'], V, 123, max_len=case.get('max_len', 20), mean_len=case.get('mean_len', 6.2))
        name = 'msgifsr'

        def build():
            torch.manual_seed(123)
            return sp.MSGIFSR(V, 'synthetic', d, 1, dropout=0.0, order=K, extra=False, fusion=False)

    def collate(caps=None):
        if name.startswith(('niser', 'srgnn')):
            return col.collate_fn_factory(col.seq_to_session_graph, caps=caps)
        return col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), K, caps=caps)
    return build, collate, samples, V


def touched_items(samples):
    """every item id the global batch reads or predicts (table rows whose gradient has a lookup / label part)"""
    t = set()
    for seq, lab in samples:
        t.update(int(i) for i in seq)
        if lab >= 0:
            t.add(int(lab))
    return torch.tensor(sorted(t), dtype=torch.int64)


def digest_rows(n, lo, touched):
    """rows of a shard [lo, lo + n) whose full contents are compared in the big (C5) cases: a stride sample of ~4096 rows
    plus every touched row of the shard (local indices)"""
    tl = touched[(touched >= lo) & (touched < lo + n)] - lo
    return torch.unique(torch.cat([torch.arange(0, n, max(1, n // 4096)), tl]))


def digest(mat, idx):
    """what leaves the process for a [n, d] matrix too large to be saved whole (C5: 1.25 M x 256 per rank, 10 M x 256 on
    the single device): float64 row sums and row norms of EVERY row + the full rows `idx`"""
    n = mat.shape[0]
    rs = torch.empty(n, dtype=torch.float64, device=mat.device)
    rn = torch.empty(n, dtype=torch.float64, device=mat.device)
    for c in range(0, n, 1 << 20):
        blk = mat[c:c + (1 << 20)].double()
        rs[c:c + (1 << 20)] = blk.sum(1)
        rn[c:c + (1 << 20)] = blk.norm(dim=1)
    return dict(rowsum=rs.cpu(), rownorm=rn.cpu(), idx=idx.cpu(), rows=mat[idx.to(mat.device)].detach().cpu().clone())


def build_on(build, dev, big):
    """big: parameters are created ON the device (a 10 M x 256 table is 10 GB: no host copy per rank process); the seeded
    draws then come from the device generator - the same in every process on the same GPU"""
    if big:
        with torch.device(dev):
            model = build()
        return model.to(dev)
    return build().to(dev)


def rank_slice(samples, world, rank, partial):
    """contiguous slice of the global batch for `rank`; partial: the last rank holds 2 live sessions fewer"""
    n = (len(samples) + world - 1) // world
    mine = samples[rank * n:(rank + 1) * n]
    if partial and rank == world - 1:
        mine = mine[:max(len(mine) - 2, 1)]
    return mine, n


def live_samples(samples, world, partial):
    out = []
    for r in range(world):
        out += rank_slice(samples, world, r, partial)[0]
    return out


def run_rank(rank, world, port, case, outdir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device('cuda:0')
        D, ops, train, optim = pkg('dist'), pkg('ops'), pkg('train'), pkg('optim')
        ops.set_precision(case.get('precision', 'fp32'))
        build, collate, samples, V = make_case(case)
        partial = bool(case.get('partial'))
        mine, n = rank_slice(samples, world, rank, partial)
        caps = None
        big = bool(case.get('big'))
        touched = touched_items(samples) if big else None
        if case.get('padded'):
            caps = pkg('collate').default_caps(n, case.get('max_len', 20))
        elif len(mine) < n:                            # exact layouts: filler sessions with label -1 (RankSliceBatchSampler)
            mine = mine + [(mine[0][0], -1)] * (n - len(mine))
        inputs, labels = collate(caps)(mine)
        inputs, labels = [x.to(dev) for x in inputs], labels.to(dev)
        model = build_on(build, dev, big)
        group = D.RecordingGroup() if case.get('record') else None
        idx_cap = inputs[0].cap('uniq_items') if caps is not None else None
        vp = D.VocabParallel(model, group=group, idx_cap=idx_cap)
        if big:
            torch.cuda.empty_cache()                   # (the full table this rank was cut from: 10 GB per process)
            rows = digest_rows(vp.n_live, vp.lo, touched)
        keep = (lambda m: digest(m, rows)) if big else (lambda m: m.detach().cpu().clone())
        opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model, fuse_projection=True)
        replicated = [p for p in model.parameters() if p is not model._table() and p.requires_grad]
        names = {id(p): k for k, p in model.named_parameters()}
        model.train()
        out = dict(rank=rank, lo=vp.lo, hi=vp.hi, n_live=vp.n_live, steps=[])
        for step in range(case.get('steps', 1)):
            if group is not None:
                group.tape = []
            D.STATS['count'] = D.STATS['bytes'] = 0
            opt.zero_grad()
            loss = model.fused_loss(*inputs, labels)
            loss.backward()
            local_none = sorted(names[id(p)] for p in replicated if p.grad is None)
            vp.sync_replicated_grads(replicated, opt)
            lval = float(loss.item())
            if getattr(model, 'extra', False) or (getattr(model, 'fusion', False) and getattr(model, 'order', 1) > 1):
                t = torch.tensor([lval], dtype=torch.float64)          # mixture losses: each rank holds its share of the
                dist.all_reduce(t)                                      # global mean (msgifsr.MSGIFSR.fused_loss)
                lval = float(t.item())
            rec = dict(loss=lval, local_none=local_none, collectives=dict(D.STATS))
            if step == 0:
                tg = model.table_grad                  # (materialises a projection left to the optimizer)
                rec['dE'] = keep(tg.buf[:vp.n_live])
                rec['grads'] = {names[i]: g.detach().cpu().clone() for i, g in (opt.grad_override or {}).items()}
            opt.step()
            rec['buckets'] = list(D.BUCKETS['bytes'])
            rec['early'] = vp.early_launches
            rec['table'] = keep(model._table().detach()[:vp.n_live])
            rec['params'] = {k: p.detach().cpu().clone() for k, p in model.named_parameters() if p is not model._table()}
            if group is not None:
                rec['tape'] = list(group.tape)
            out['steps'].append(rec)
        # evaluation over the sharded table: every rank feeds its own sessions (data-parallel evaluation)
        model.eval()
        vp.eval_data_parallel = True
        if not (getattr(model, 'extra', False) or (getattr(model, 'fusion', False) and getattr(model, 'order', 1) > 1)):
            v, i = model.topk(*inputs, k=20)
            out['topk'] = (v.cpu(), i.cpu())
        with torch.no_grad():
            if case['kind'] == 'fixture':
                out['logp'] = model(*inputs).cpu()
        torch.cuda.synchronize()
        torch.save(out, os.path.join(outdir, 'rank%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()
