"""Row-sharded (vocab-parallel) item table over the GPUs of one node - RCCL over xGMI.

Reference: none (the reference is single-device); the partitioning is the one
BASELINE.json's north star names: the item-embedding table and its use as output projection are
sharded row-wise, the small session encoder is replicated, and each rank feeds its OWN batch of B
sessions (weak scaling: global batch = world * B, one optimiser step on the mean loss of all of them).

Per step and rank r (rows [lo, hi) of the table, Adam state for those rows only):
  lookup   all-gather the (padded) item-id lists -> every rank gathers the rows it owns for every
           request (others zero) -> reduce-scatter(sum): each rank receives the rows of its own nodes.
           backward: local segmented sum to one row per distinct item -> all-gather (ids, rows) ->
           each rank adds the rows it owns into its dense shard gradient, rank by rank (deterministic).
  scoring  all-gather sr (world*B, d) and labels -> fused flash-CE forward on the local shard for ALL
           sessions -> all-gather the per-shard log-sum-exp, all-reduce the label logit -> global lse
           and loss.  backward: fused kernels with the GLOBAL lse: dE for the shard is complete
           locally; d sr partials are reduce-scattered to the owners of the sessions.
  encoder  replicated parameters: gradients all-reduced (sum of per-rank contributions to the global
           mean loss) in three flat buckets in backward completion order (sync_replicated_grads), then the same fused
           Adam everywhere.
All exchanges are small (<= a few MB): latency-bound on xGMI, so they are few and flat (one
collective per exchange, no ring of tiny messages).  The local compute goes through `local`, an
object with the HIP kernels (HipLocal); tests drive the same algebra on CPU/gloo with a plain-torch
stand-in to prove the collectives reassemble the single-device result.
"""
import os

import torch
import torch.distributed as dist

# SREC_FORCE_COLLECTIVES=1: issue the RCCL calls even on a 1-rank communicator (they are then copies) - lets the
# collective sequence and its hipGraph capture be exercised on a single-GPU box (bench.py --shard)
FORCE = bool(os.environ.get('SREC_FORCE_COLLECTIVES'))


# ------------------------------------------------------------------------------- collectives
# `group` is a torch.distributed process group (None = the default one) OR an object with `srec_loopback = True`
# (RecordingGroup / ReplayGroup below): the single-GPU rehearsal of one rank of an N-rank job.
BUCKETS = {'bytes': []}                # payload of the replicated-gradient buckets of the last step (sync_replicated_grads)
STATS = {'count': 0, 'bytes': 0}      # collectives issued through this module and their payload (bytes of the full exchanged
#                                       buffer as this rank sees it): bench.py reports them per step


def _loop(group):
    return group is not None and getattr(group, 'srec_loopback', False)


def _world(group=None):
    if _loop(group):
        return group.world
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group=None):
    if _loop(group):
        return group.rank
    return dist.get_rank(group) if dist.is_initialized() else 0


def _active(group=None):
    """collectives are issued: more than one rank, or forced on a 1-rank communicator"""
    return _world(group) > 1 or (FORCE and dist.is_initialized())


def _count(nbytes):
    STATS['count'] += 1
    STATS['bytes'] += int(nbytes)


TIMING = None                         # a list: every collective issued through this module is bracketed by HIP events on the
#                                       stream it is issued from and appended as (kind, payload bytes, start, end) - eager steps
#                                       only (bench.py's `collectives.timed`: what an exchange costs on the rank's timeline,
#                                       waiting for the slowest rank included)


def _timed(kind, nbytes, fn, t):
    if TIMING is None or not t.is_cuda or torch.cuda.is_current_stream_capturing():
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    TIMING.append((kind, int(nbytes), e0, e1))
    return out


def timing_summary(entries):
    """[(kind, bytes, start, end)] of whole steps -> [dict(kind, bytes, us)] (device synchronised by the caller)"""
    return [dict(kind=k, bytes=b, us=round(e0.elapsed_time(e1) * 1e3, 2)) for k, b, e0, e1 in entries]


def _host_staged(t, group):
    """gloo moves host memory: device tensors are staged through the host (the W-ranks-on-one-GPU tests; production = RCCL)"""
    return t.is_cuda and dist.get_backend(group) != 'nccl'


def all_gather_cat(t, group=None):
    """[n, ...] per rank -> [world*n, ...] (same n everywhere)"""
    if not _active(group):
        return t
    t = t.contiguous()
    nb = _world(group) * t.numel() * t.element_size()
    _count(nb)
    return _timed('all_gather', nb, lambda: group.collective('all_gather', t) if _loop(group) else all_gather_cat_pg(t, group), t)


def reduce_scatter_sum(t, group=None):
    """[world*n, ...] per rank -> [n, ...]: rank r receives the sum over ranks of block r"""
    if not _active(group):
        return t
    t = t.contiguous()
    _count(t.numel() * t.element_size())
    return _timed('reduce_scatter', t.numel() * t.element_size(),
                  lambda: group.collective('reduce_scatter', t) if _loop(group) else reduce_scatter_pg(t, group), t)


def all_reduce_(t, op=None, group=None, always=False):
    """in-place all-reduce (SUM unless `op`) through the job's backend.  always: also on a 1-rank world without FORCE"""
    if not (_active(group) or (always and (dist.is_initialized() or _loop(group)))):
        return t
    op = dist.ReduceOp.SUM if op is None else op
    _count(t.numel() * t.element_size())

    def run():
        if _loop(group):
            t.copy_(group.collective('all_reduce:' + {dist.ReduceOp.MAX: 'MAX', dist.ReduceOp.MIN: 'MIN'}.get(op, 'SUM'), t))
        elif _host_staged(t, group):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=group)
        return t
    return _timed('all_reduce', t.numel() * t.element_size(), run, t)


def all_reduce_sum(t, group=None):
    return all_reduce_(t, None, group)


# ---- single-GPU rehearsal of ONE rank of an N-rank job: record what the collectives returned on that rank in a real
#      multi-process run (RecordingGroup wraps the process group), then replay that rank's step alone - eagerly or under
#      hipGraph capture - with the recorded results standing in for the other ranks (ReplayGroup).  What a rank computes
#      between two collectives depends only on its own inputs and on what the collectives handed it, so the replayed step
#      runs the very kernels, with the very operands (shard offset > 0, foreign item ids, other ranks' session vectors),
#      that the rank ran inside the job.
class RecordingGroup:
    srec_loopback = True

    def __init__(self, group=None):
        self.pg, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        self.tape = []

    def collective(self, kind, t):
        if kind == 'all_gather':
            out = all_gather_cat_pg(t, self.pg)
        elif kind == 'reduce_scatter':
            out = reduce_scatter_pg(t, self.pg)
        else:
            out = t.clone()
            h = out.cpu() if _host_staged(out, self.pg) else out
            dist.all_reduce(h, op=getattr(dist.ReduceOp, kind.split(':')[1]), group=self.pg)
            out.copy_(h)
        self.tape.append((kind, t.detach().clone().cpu(), out.detach().clone().cpu()))
        return out


class ReplayGroup:
    """answers the collectives of rank `rank` from a RecordingGroup tape (load(tape); a tape holds one step and is cycled when
    the step is run again).  Outside stream capture every call checks that the rank hands in what it handed in when the
    tape was made; under capture the recorded result is copied out of a static device tensor (a memcpy node in the graph
    where the job has its RCCL call)."""
    srec_loopback = True

    def __init__(self, world, rank, device, rtol=1e-4, atol=1e-6):
        self.world, self.rank, self.device, self.rtol, self.atol = world, rank, device, rtol, atol
        self.kinds, self.inputs, self.outputs = [], [], []
        self.pos = self.checked = 0

    def load(self, tape):
        self.kinds = [k for k, _, _ in tape]
        self.inputs = [i for _, i, _ in tape]
        self.outputs = [o.to(self.device) for _, _, o in tape]   # static device tensors: a captured copy reads them
        self.pos = self.checked = 0
        return self

    def collective(self, kind, t):
        i = self.pos % len(self.kinds)
        self.pos += 1
        assert kind == self.kinds[i], 'collective %d is %s, the recorded step issued %s' % (i, kind, self.kinds[i])
        out = self.outputs[i]
        assert t.dtype == out.dtype, (i, kind, t.dtype, out.dtype)
        if not (t.is_cuda and torch.cuda.is_current_stream_capturing()):
            a, b = t.detach().cpu(), self.inputs[i]
            assert a.shape == b.shape, (i, kind, a.shape, b.shape)
            if a.dtype.is_floating_point:
                err = (a - b).abs().max().item() if a.numel() else 0.0
                tol = max(self.atol, self.rtol * (float(b.abs().max()) if b.numel() else 0.0))
                assert torch.allclose(a, b, rtol=self.rtol, atol=tol), \
                    'collective %d (%s): this rank hands in something else than in the recorded run (max |diff| %.3e)' % (i, kind, err)
            else:
                assert torch.equal(a, b), 'collective %d (%s): integer payload differs from the recorded run' % (i, kind)
            self.checked += 1
        return out.clone()


def all_gather_cat_pg(t, pg):
    w = dist.get_world_size(pg)
    shape = (w * t.shape[0],) + tuple(t.shape[1:])
    src = t.cpu() if _host_staged(t, pg) else t
    out = torch.empty(shape, dtype=t.dtype, device=src.device)
    if dist.get_backend(pg) == 'nccl':
        dist.all_gather_into_tensor(out, src, group=pg)
    else:
        dist.all_gather(list(out.chunk(w, 0)), src, group=pg)
    return out.to(t.device)


def reduce_scatter_pg(t, pg):
    w, r = dist.get_world_size(pg), dist.get_rank(pg)
    n = t.shape[0] // w
    if dist.get_backend(pg) == 'nccl':
        out = torch.empty((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=pg)
        return out
    h = t.cpu() if t.is_cuda else t.clone()
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=pg)
    return h[r * n:(r + 1) * n].clone().to(t.device)


def shard_bounds(V, world, rank):
    per = (V + world - 1) // world
    per = (per + 63) // 64 * 64                 # whole 64-row scoring tiles per shard
    lo = min(V, rank * per)
    return lo, min(V, lo + per), per


# ------------------------------------------------------------------------------- local compute (HIP)
class HipLocal:
    """The per-rank compute of the sharded path, on the HIP kernels."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def localize(self, idx_all, lo, n_loc, zero=None, zero_from=0):
        """global ids (int64 / int32, -1 padding) -> int32 rows of this shard, -1 elsewhere.  zero: a float array as long as the
        list from entry zero_from on, cleared by the same launch (int32 ids) - else by a fill"""
        from ._lib import lib, ptr, stream
        idx_all = idx_all.contiguous()
        out = torch.empty(idx_all.numel(), device=idx_all.device, dtype=torch.int32)
        if idx_all.dtype == torch.int32:
            assert zero is None or (zero.numel() == idx_all.numel() - zero_from and zero.is_contiguous())
            lib.srec_localize_idx32(ptr(idx_all), idx_all.numel(), int(lo), int(n_loc), ptr(out), ptr(zero), int(zero_from), stream())
        else:
            lib.srec_localize_idx(ptr(idx_all), idx_all.numel(), int(lo), int(n_loc), ptr(out), stream())
            if zero is not None:
                zero.zero_()
        return out

    def merge_stats(self, st, lab_all=None):
        """st [w, 2, B] gathered per-shard statistics (+ the gathered global labels, < 0 = capacity padding) ->
        (lse [B], label logit [B], mean loss over the live sessions, gw [B] = d loss / d (lse_b - lab_b))"""
        from ._lib import lib, ptr, stream
        st = st.contiguous()
        w, _, B = st.shape
        lse = torch.empty(B, device=st.device, dtype=torch.float32)
        lab = torch.empty(B, device=st.device, dtype=torch.float32)
        gw = torch.empty(B, device=st.device, dtype=torch.float32)
        loss = torch.empty((), device=st.device, dtype=torch.float32)
        fn = lib.srec_merge_stats32 if (lab_all is not None and lab_all.dtype == torch.int32) else lib.srec_merge_stats
        fn(ptr(st), w, B, ptr(lab_all), ptr(lse), ptr(lab), ptr(loss), ptr(gw), stream())
        return lse, lab, loss, gw

    def inverse_index(self, uptr, upos, U, n):
        """inv[p] = u for every position p of item u (positions listed in upos[uptr[u]:uptr[u+1]]); -1 where no item claims p"""
        from ._lib import lib, ptr, stream
        inv = torch.empty(n, device=upos.device, dtype=torch.int32)
        lib.srec_inverse_index(ptr(uptr), ptr(upos), U, n, ptr(inv), stream())
        return inv

    takes_gl = True                # stats_bwd(gl=): the upstream loss gradient as a device scalar
    fused_dropout = True           # gather_masked / segment_rows take drop = (p, seed, counter, salt): the lookup's feature
    #                                dropout rides in the last gather and in the first level of its backward (ops.EmbeddingLookup)

    def gather_masked(self, table, idx, drop=None):
        from ._lib import lib, ptr, stream
        n, d = idx.numel(), table.shape[1]
        out = torch.empty(n, d, device=table.device, dtype=torch.float32)
        if drop is not None:
            lib.srec_gather_rows_drop(ptr(table), table.stride(0), ptr(idx), ptr(out), d, n, None, d, drop[0], drop[1],
                                      drop[2], drop[3], stream())
        else:
            lib.srec_gather_rows(ptr(table), table.stride(0), ptr(idx), ptr(out), d, n, None, d, stream())
        return out

    def segment_rows(self, g, uniq, drop=None):
        """one summed row per distinct item of the local batch: [U, d].  With the (cptr, chunk_ptr) pair of the FlatBatch
        the sum runs in two balanced levels (<= 16 positions per wavefront, then the pieces of each item): a
        Zipf-popular item would otherwise serialise ~1000 row reads in one wavefront."""
        from ._lib import lib, ptr, stream
        items, uptr, upos = uniq[:3]
        U, d = items.numel(), g.shape[1]
        g = g.contiguous()
        out = torch.empty(U, d, device=g.device, dtype=torch.float32)
        if len(uniq) == 5:
            cptr, chunk_ptr = uniq[3], uniq[4]
            C = chunk_ptr.numel() - 1
            part = torch.empty(max(C, 1), d, device=g.device, dtype=torch.float32)
            ar = self.ops._arange(max(C, U) + 1, g.device)
            if drop is not None:
                lib.srec_scatter_add_sorted_drop(ptr(g), d, ptr(ar), ptr(chunk_ptr), ptr(upos), ptr(part), d, C, None, d, 0,
                                                 drop[0], drop[1], drop[2], drop[3], stream())
            else:
                lib.srec_scatter_add_sorted(ptr(g), d, ptr(ar), ptr(chunk_ptr), ptr(upos), ptr(part), d, C, None, d, 0, stream())
            lib.srec_scatter_add_sorted(ptr(part), d, ptr(ar), ptr(cptr), ptr(ar), ptr(out), d, U, None, d, 0, stream())
        else:
            ar = self.ops._arange(U + 1, g.device)
            if drop is not None:
                lib.srec_scatter_add_sorted_drop(ptr(g), d, ptr(ar), ptr(uptr), ptr(upos), ptr(out), d, U, None, d, 0,
                                                 drop[0], drop[1], drop[2], drop[3], stream())
            else:
                lib.srec_scatter_add_sorted(ptr(g), d, ptr(ar), ptr(uptr), ptr(upos), ptr(out), d, U, None, d, 0, stream())
        return out

    def add_rows(self, rows, items_local, dst, proj=None):
        """dst[items_local[u]] += rows[u] for items_local[u] >= 0 (distinct within the call).  proj = (table, radial): the
        deferred row-normalisation projection is pending on dst (ops.TableGrad) - the rows' radial parts are recorded"""
        from ._lib import lib, ptr, stream
        U, d = rows.shape
        ar = self.ops._arange(U + 1, rows.device)
        if proj is not None:
            lib.srec_scatter_add_sorted_ex(ptr(rows), d, ptr(items_local), ptr(ar), ptr(ar), ptr(dst), dst.stride(0), U, None,
                                           d, 1, 0.0, 0, None, 0, ptr(proj[0]), proj[0].stride(0), ptr(proj[1]), stream())
        else:
            lib.srec_scatter_add_sorted(ptr(rows), d, ptr(items_local), ptr(ar), ptr(ar), ptr(dst), dst.stride(0), U, None,
                                        d, 1, stream())

    def add_rows_all(self, rows_all, rel, ids, w, ucap, dst, proj=None):
        """add_rows for the lists of all w ranks in ONE launch, per item in rank order (csrc/rowops.hip add_rows_ranks_kernel);
        False when the shapes are outside the kernel's contract (the caller then loops over the ranks)"""
        from ._lib import lib, ptr, stream
        if w > 16 or ids is None or ids.dtype != torch.int32 or rel.dtype != torch.int32 or not rows_all.is_contiguous():
            return False
        d = rows_all.shape[1]
        lib.srec_add_rows_ranks(ptr(rows_all), d, ptr(rel), ptr(ids), w, ucap, ptr(dst), dst.stride(0),
                                ptr(proj[0]) if proj is not None else None, proj[0].stride(0) if proj is not None else 0,
                                ptr(proj[1]) if proj is not None else None, stream())
        return True

    def _tb(self, table, refresh):
        """bf16 operand copies of this rank's shard when set_precision('bf16') is on (refreshed once per step)"""
        ops = self.ops
        if not ops.use_bf16_scoring(table.shape[1]):
            return None
        key = (table.data_ptr(), tuple(table.shape))
        tb = self.__dict__.setdefault('_tb16', {}).get(key)
        if tb is None:
            tb = self._tb16[key] = ops.TableBF16(table)
            refresh = True
        elif refresh and self.__dict__.pop('_tb_written', None) == (key, table._version):
            refresh = False                    # FusedAdam's row pass of the previous step wrote the copy (optim.py)
        return tb.refresh(table) if refresh else tb

    def ce_fwd(self, sr, table, cs, labels_local, ws, pair=None):
        B, d = sr.shape
        dev = sr.device
        # (lse, label logit) as the two rows of ONE buffer: it is what _merge_stats all-gathers (no stack / clone launches)
        if pair is None:                           # (pair given: its label-logit row has been cleared by the caller's localize)
            pair = torch.empty(2, B, device=dev, dtype=torch.float32)
            pair[1].zero_()
        lossvec = torch.empty(B, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        self.ops._ce_fwd(sr, table, cs, labels_local, ws, None, self._tb(table, True), pair[1], pair[0], lossvec, loss)
        return pair[0], pair[1]

    def ce_bwd(self, sr, table, cs, labels_local, lse, gscale, dE, ws, cs_inv_scale, defer_tg=None):
        from ._lib import lib, ptr, stream
        B, d = sr.shape
        dsr = torch.empty(B, d, device=sr.device, dtype=torch.float32)
        self.ops._ce_bwd(sr, table, cs, labels_local, lse, gscale, None, None, ws, None, self._tb(table, False), dE, dsr, 3)
        if cs is not None and defer_tg is not None:
            defer_tg.pending = (table, cs, cs_inv_scale)      # applied by the optimizer's row pass (ops.TableGrad)
        elif cs is not None:
            lib.srec_rownorm_project(ptr(table), table.stride(0), ptr(cs), cs_inv_scale, ptr(dE), dE.stride(0),
                                     table.shape[0], d, stream())
        return dsr

    def stats_bwd(self, sr, table, cs, labels_local, lse, ga, gc, dE, ws, cs_inv_scale, accumulate, defer_tg=None, gl=None):
        """d z[b, v] = ga[b] * softmax(z_b)[v] - gc[b] * [v == label_b] through this rank's rows: dE (+)= dz^T sr,
        returns the partial d sr = dz E_local"""
        from ._lib import lib, ptr, stream
        B, d = sr.shape
        dsr = torch.empty(B, d, device=sr.device, dtype=torch.float32)
        if defer_tg is not None and not accumulate:
            defer_tg.overwritten()                     # stale pending / radial of a backward no optimizer step consumed
        self.ops._ce_bwd(sr, table, cs, labels_local, lse, gl, ga, gc, ws, None, self._tb(table, False), dE, dsr,
                         3 | (4 if accumulate else 0))
        if cs is not None and defer_tg is not None:
            defer_tg.pending = (table, cs, cs_inv_scale)      # linear: once, over the sum of the heads' contributions
        elif cs is not None:
            lib.srec_rownorm_project(ptr(table), table.stride(0), ptr(cs), cs_inv_scale, ptr(dE), dE.stride(0),
                                     table.shape[0], d, stream())
        return dsr

    def logp_cols(self, sr, table, cs, lse):
        """this rank's columns of the (B, V) log-probabilities: z[b, v] - lse[b] for the local rows v (fp32)"""
        from ._lib import lib, ptr, stream
        B, d = sr.shape
        n = table.shape[0]
        ld = (n + 3) & ~3
        out = torch.empty(B, ld, device=sr.device, dtype=torch.float32)
        sr = sr.contiguous()
        lib.srec_score_logp(ptr(sr), sr.stride(0), ptr(table), table.stride(0), ptr(cs), ptr(lse), B, n, d, None,
                            ptr(out), ld, stream())
        return out[:, :n]

    def topk(self, sr, table, cs, k):
        from . import ops
        return ops.score_topk(sr, table, cs, k)

    def workspace(self, B, V, d, device):
        return self.ops.CEWorkspace(B, V, d, device)


# ------------------------------------------------------------------------------- autograd functions
def _merge_stats(lse_r, lab_logit_r, group, local=None, lab_all=None):
    """per-shard (log-sum-exp, label logit) of every session -> global (lse, label logit, mean loss, gw): ONE all-gather of
    the [2, B] pair (the label logit is non-zero on exactly one shard, so its sum over the gathered copies is the
    all-reduce it replaces) and one merge kernel.  lab_all (optional): the gathered global labels; sessions with a
    label < 0 are the capacity padding of a rank's (partial) batch and are left out of the mean.  gw [B] =
    d loss / d (lse_b - lab_b): 1 / n_live on live sessions, 0 on padding."""
    def torch_merge(lse, lab):
        if lab_all is None:
            live = torch.ones_like(lse)
        else:
            live = (lab_all >= 0).to(lse.dtype)
        gw = live / live.sum().clamp(min=1.0)
        return lse, lab, ((lse - lab) * gw).sum(), gw
    if not _active(group):
        return torch_merge(lse_r, lab_logit_r)
    B = lse_r.numel()
    if lse_r.is_contiguous() and lab_logit_r.is_contiguous() and lab_logit_r.data_ptr() == lse_r.data_ptr() + 4 * B \
            and lse_r.dtype == torch.float32:
        pair = lse_r.as_strided((1, 2, B), (2 * B, B, 1))                      # HipLocal.ce_fwd: already adjacent rows
    else:
        pair = torch.stack([lse_r, lab_logit_r]).unsqueeze(0)
    st = all_gather_cat(pair, group)                                           # [w, 2, B]
    if local is not None and hasattr(local, 'merge_stats'):
        return local.merge_stats(st, lab_all)
    return torch_merge(torch.logsumexp(st[:, 0], dim=0).contiguous(), st[:, 1].sum(0))


def _follows(a, b):
    """b starts where a ends, in the same storage (two fields of one batch buffer): a | b is then one contiguous list"""
    return (a.dtype == b.dtype and a.dim() == 1 and b.dim() == 1 and a.is_contiguous() and b.is_contiguous() and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel())


class ShardedLookup(torch.autograd.Function):
    """rows = E[idx] with E row-sharded.  Only the DISTINCT items of a rank's batch travel: their ids are all-gathered
    (padded with -1 to a capacity that is equal on all ranks), every rank contributes the rows it owns (zeros elsewhere),
    a reduce-scatter hands each rank the rows of its own items and the positions of the batch are filled from those by a
    local gather (inv: position -> slot of its item).  The exchange volume is (distinct items) x d per rank and direction,
    not (positions) x d - ~4x less at the MSGIFSR shapes, where every click is looked up once per n-gram order."""

    @staticmethod
    def forward(ctx, shard, items_pad, inv, uniq, dE, lo, local, group, vp=None, drop=None):
        n_loc = shard.shape[0]
        # ONE integer exchange per step: the padded distinct-item list (used again by the backward) and, when the caller has
        # announced them (VocabParallel.labels_hint), the labels of the loss - a few KB each, i.e. latencies on xGMI
        ucap = items_pad.numel()
        parts = [items_pad]
        lab = vp.labels_hint if vp is not None else None
        if lab is not None:
            parts.append(lab if lab.dtype == items_pad.dtype else lab.to(items_pad.dtype))
        if len(parts) > 1 and _follows(items_pad, parts[1]):
            # the batch buffer already holds (distinct items | labels) back to back (collate._items_last): the request list is a view
            req = items_pad.as_strided((ucap + parts[1].numel(),), (1,), items_pad.storage_offset())
        elif len(parts) > 1 and vp is not None and items_pad.is_cuda and items_pad.dtype == torch.int32:
            # (ids | labels) into a request buffer by the library's multi-copy launch (no aten concatenation in the rank step)
            req = torch.empty(ucap + parts[1].numel(), device=items_pad.device, dtype=torch.int32)
            vp._copy_tasks([(items_pad.view(torch.float32), req[:ucap].view(torch.float32)),
                            (parts[1].contiguous().view(torch.float32), req[ucap:].view(torch.float32))])
        else:
            req = torch.cat(parts) if len(parts) > 1 else parts[0]
        packed = all_gather_cat(req.unsqueeze(0), group)                                      # [w, ucap (+ B)]
        w_ = packed.shape[0]
        if lab is not None and w_ > 1 and w_ <= 16 and vp is not None and packed.is_cuda and packed.dtype == torch.int32:
            # the two column blocks as contiguous lists, one multi-copy launch (two strided aten copies before)
            nb = packed.shape[1] - ucap
            both = torch.empty(w_ * (ucap + nb), device=packed.device, dtype=torch.int32)
            ctx.items_all, vp.lab_all = both[:w_ * ucap], both[w_ * ucap:]
            pf, bf = packed.view(torch.float32), both.view(torch.float32)
            vp._copy_tasks([(pf[r, :ucap], bf[r * ucap:(r + 1) * ucap]) for r in range(w_)] +
                           [(pf[r, ucap:], bf[w_ * ucap + r * nb:w_ * ucap + (r + 1) * nb]) for r in range(w_)])
        else:
            both = None
            ctx.items_all = packed[:, :ucap].reshape(-1)
            if lab is not None:
                vp.lab_all = packed[:, ucap:].reshape(-1)
        if both is not None and getattr(local, 'takes_gl', False):
            # requests and labels localised by ONE launch, which also clears the label-logit row of the scoring forward
            # (ShardedScoreCE takes both from vp.lab_pre: one small launch less in front of it)
            pair = torch.empty(2, w_ * nb, device=both.device, dtype=torch.float32)
            rel_both = local.localize(both, lo, n_loc, zero=pair[1], zero_from=w_ * ucap)
            ctx.rel, vp.lab_pre = rel_both[:w_ * ucap], (rel_both[w_ * ucap:], pair, n_loc, lo)
        else:
            ctx.rel = local.localize(ctx.items_all, lo, n_loc)     # local row of every requested item (-1: another rank's); reused by the backward
        rows_all = local.gather_masked(shard, ctx.rel)                                        # [w * ucap, d]
        mine = reduce_scatter_sum(rows_all, group)                                            # [ucap, d]: my items' rows
        ctx.drop = None
        if drop is not None and drop[0] > 0:       # feature dropout of the looked-up rows (msgifsr.py:247) fused into the gather
            seed, cnt = local.ops.rng_args(shard.device)
            ctx.drop = (float(drop[0]), seed, cnt, int(drop[1]))
            out = local.gather_masked(mine, inv, ctx.drop)                                    # [n, d]: my positions
        else:
            out = local.gather_masked(mine, inv)
        ctx.uniq, ctx.ucap, ctx.dE, ctx.lo, ctx.local, ctx.group, ctx.n_loc = uniq, ucap, dE, lo, local, group, n_loc
        ctx.vp, ctx.shard = vp, shard
        return out

    @staticmethod
    def backward(ctx, g):
        # [U, d]: one summed row per distinct item of my batch (the dropout mask re-derived while summing)
        rows = ctx.local.segment_rows(g, ctx.uniq, ctx.drop) if ctx.drop is not None else ctx.local.segment_rows(g, ctx.uniq)
        U, ucap = rows.shape[0], ctx.ucap
        if U < ucap:
            rows = torch.cat([rows, rows.new_zeros(ucap - U, rows.shape[1])])
        rows_all = all_gather_cat(rows, ctx.group)
        rel = ctx.rel
        tg = ctx.vp.tgrad if ctx.vp is not None else None
        proj = (ctx.shard, tg.radial) if (tg is not None and tg.pending is not None) else None
        if proj is not None:
            tg.radial_dirty = True
        w = _world(ctx.group)
        if w > 1 and hasattr(ctx.local, 'add_rows_all') and \
                ctx.local.add_rows_all(rows_all, rel, ctx.items_all, w, ucap, ctx.dE, proj):
            return (None,) * 10
        for r in range(w):                                   # rank by rank: distinct items within each call
            if proj is not None:
                ctx.local.add_rows(rows_all[r * ucap:(r + 1) * ucap], rel[r * ucap:(r + 1) * ucap], ctx.dE, proj)
            else:
                ctx.local.add_rows(rows_all[r * ucap:(r + 1) * ucap], rel[r * ucap:(r + 1) * ucap], ctx.dE)
        return (None,) * 10


class ShardedScoreCE(torch.autograd.Function):
    """mean CE over the GLOBAL batch (world*B sessions) against the row-sharded catalog."""

    @staticmethod
    def forward(ctx, sr, shard, cs, labels, dE, lo, ws, cs_inv_scale, local, group, lab_all=None, tgrad=None, lab_pre=None):
        n_loc = shard.shape[0]
        sr_all = all_gather_cat(sr.contiguous(), group)
        if lab_all is None:                                            # not exchanged with the lookup's request lists
            lab_all = all_gather_cat(labels if labels.dtype == torch.int32 else labels.to(torch.int64), group)
            lab_pre = None
        if lab_pre is not None and lab_pre[3] == lo and lab_pre[2] >= n_loc and lab_pre[0].numel() == sr_all.shape[0]:
            # (the lookup localised against ALL rows of the shard, capacity padding included: no label names a padding row)
            lab_loc, pair = lab_pre[:2]            # localised (and the label-logit row cleared) by the lookup's launch
            lse_r, lab_logit = local.ce_fwd(sr_all, shard, cs, lab_loc, ws, pair=pair)
        elif getattr(local, 'takes_gl', False):      # HipLocal: the label-logit row is cleared by the launch that localises the labels
            pair = torch.empty(2, sr_all.shape[0], device=sr_all.device, dtype=torch.float32)
            lab_loc = local.localize(lab_all, lo, n_loc, zero=pair[1])
            lse_r, lab_logit = local.ce_fwd(sr_all, shard, cs, lab_loc, ws, pair=pair)
        else:
            lab_loc = local.localize(lab_all, lo, n_loc)
            lse_r, lab_logit = local.ce_fwd(sr_all, shard, cs, lab_loc, ws)
        lse, lab_logit, loss, gw = _merge_stats(lse_r, lab_logit, group, local, lab_all)
        ctx.save_for_backward(sr_all, shard, cs, lab_loc, lse, gw)
        ctx.misc = (dE, ws, cs_inv_scale, local, group, tgrad)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        sr_all, shard, cs, lab_loc, lse, gw = ctx.saved_tensors
        dE, ws, cs_inv_scale, local, group, tgrad = ctx.misc
        # d z[b, v] = g_b (softmax_b[v] - [v == label_b]) with g_b = gloss / n_live on live sessions, 0 on the capacity
        # padding of a rank's partial batch (the mean runs over the live sessions of the GLOBAL batch)
        kw = {}
        if getattr(local, 'takes_gl', False):      # the HIP kernels take the upstream scalar as a pointer and multiply the
            g = gw                                 # per-session coefficients by it themselves: no scaling launch
            kw['gl'] = gloss.reshape(()).to(torch.float32)
        else:
            g = (gw * gloss.reshape(()).to(torch.float32)).contiguous()
        if tgrad is not None and tgrad.defer:
            dsr_part = local.stats_bwd(sr_all, shard, cs, lab_loc, lse, g, g, dE, ws, cs_inv_scale, False, tgrad, **kw)
        else:
            dsr_part = local.stats_bwd(sr_all, shard, cs, lab_loc, lse, g, g, dE, ws, cs_inv_scale, False, **kw)
        dsr = reduce_scatter_sum(dsr_part, group)
        return (dsr,) + (None,) * 12


class ShardedScoreStats(torch.autograd.Function):
    """(lse_b, z[b, label_b]) of this rank's sessions against the row-sharded catalog, differentiable in both outputs:
    the sharded counterpart of ops.ScoreStats, for losses that mix several soft-maxes (MSGIFSR order fusion / extra,
    msgifsr.py:281-317).  Several heads share one table: the first backward of a step overwrites dE, later ones
    accumulate (tgrad.fresh)."""

    @staticmethod
    def forward(ctx, sr, shard, cs, labels, dE, lo, ws, cs_inv_scale, local, group, tgrad):
        n_loc, n = shard.shape[0], sr.shape[0]
        sr_all = all_gather_cat(sr.contiguous(), group)
        lab_all = all_gather_cat(labels.to(torch.int64), group)
        lab_loc = local.localize(lab_all, lo, n_loc)
        lse_r, lab_logit = local.ce_fwd(sr_all, shard, cs, lab_loc, ws)
        lse, lab_logit, _, _ = _merge_stats(lse_r, lab_logit, group, local)
        r = _rank(group)
        ctx.save_for_backward(sr_all, shard, cs, lab_loc, lse)
        ctx.misc = (dE, ws, cs_inv_scale, local, group, tgrad)
        return lse[r * n:(r + 1) * n].clone(), lab_logit[r * n:(r + 1) * n].clone()

    @staticmethod
    def backward(ctx, dlse, dlab):
        sr_all, shard, cs, lab_loc, lse = ctx.saved_tensors
        dE, ws, cs_inv_scale, local, group, tgrad = ctx.misc
        ga = all_gather_cat(dlse.contiguous().float(), group)
        gc = all_gather_cat((-dlab).contiguous().float(), group)
        if tgrad.defer:
            dsr_part = local.stats_bwd(sr_all, shard, cs, lab_loc, lse, ga, gc, dE, ws, cs_inv_scale, tgrad.fresh, tgrad)
        else:
            dsr_part = local.stats_bwd(sr_all, shard, cs, lab_loc, lse, ga, gc, dE, ws, cs_inv_scale, tgrad.fresh)
        tgrad.fresh = True
        dsr = reduce_scatter_sum(dsr_part, group)
        return (dsr,) + (None,) * 10


# ------------------------------------------------------------------------------- model wiring
class VocabParallel:
    """Attach to a model (`model.shard = VocabParallel(model)`): the model's table parameter becomes this
    rank's row shard and the lookup / fused loss route through the collectives above."""

    def __init__(self, model, group=None, local=None, idx_cap=None):
        self.group = group
        self.world, self.rank = _world(group), _rank(group)
        self.local = local if local is not None else HipLocal()
        table = model._table()
        self.V, self.d = table.shape
        self.lo, self.hi, self.per = shard_bounds(self.V, self.world, self.rank)
        with torch.no_grad():
            shard = torch.zeros(self.per, self.d, device=table.device, dtype=table.dtype)
            shard[:self.hi - self.lo] = table[self.lo:self.hi]
            table.data = shard                      # parameter now holds only this rank's rows
        self.n_live = self.hi - self.lo
        model.__dict__.pop('_srec_state', None)     # rebuild the scoring state for the sharded table
        self.tgrad = model._state(1)['tgrad']
        self.dE = self.tgrad.buf
        self.idx_cap = idx_cap
        self.labels_hint = None             # set by the model before the lookup of a training step: labels ride along
        self.lab_all = None                 # ... and come back gathered for loss()
        self.eval_data_parallel = False     # evaluate(): True when every rank feeds its own equal-sized slice of a batch
        self._ws = {}
        self.model = model
        self._names = {id(p): n for n, p in model.named_parameters()} if hasattr(model, 'named_parameters') else {}
        self._bucket_early, self._bucket_zero = {}, {}
        self._side, self._side_used, self._no_sync = None, False, False
        self._arena, self._slot = [], {}
        self.early_launches = 0                # buckets whose all-reduce was issued from inside a backward pass (bucket_ready)
        # side stream for the early buckets: on by default where the collective is RCCL between real ranks (it then runs beside
        # the backward); a 1-rank / loopback rehearsal keeps one stream unless asked (graph branches cost there: DESIGN 7)
        self.side_stream = bool(self.world > 1 and not _loop(group) and dist.is_initialized()
                                and dist.get_backend(group) == 'nccl') or bool(os.environ.get('SREC_BUCKET_SIDE_STREAM'))
        model.shard = self

    def _pad(self, t, cap, fill=-1):
        out = torch.full((cap,), fill, device=t.device, dtype=t.dtype)
        out[:t.numel()] = t
        return out

    def capacity(self, n):
        """a common padded length for the distinct-item lists: `idx_cap` (capacity-padded batches: no exchange, no host
        sync, graph-capturable) or the max over ranks, rounded up (one tiny all-reduce per batch)"""
        if self.idx_cap is not None:
            assert n <= self.idx_cap
            return self.idx_cap
        t = torch.tensor([n], dtype=torch.int64, device=self.dE.device)
        if self.world > 1:
            all_reduce_(t, dist.ReduceOp.MAX, self.group)
        return (int(t.item()) + 255) // 256 * 256

    def lookup(self, table, idx, uniq, drop=None, inv=None):
        """inv: position -> slot of its item in `items` (-1: padding) when the batch carries it (FlatBatch.uniq_inv, built by
        the collate workers) - else derived from the item CSR here (two small launches)"""
        if torch.is_grad_enabled() and table.requires_grad:
            self.begin_step()                  # (a training forward: every model's first use of the sharded table is its lookup)
        items, uptr, upos = uniq[:3]
        n, U = idx.numel(), items.numel()
        ucap = self.capacity(U)
        # capacity-padded batches: the distinct-item field of the batch buffer IS the padded request list (int32, -1 in the
        # unused slots) - no conversion, no copy; exact layouts pad a copy
        items_pad = items if U == ucap else self._pad(items if items.dtype == torch.int32 else items.to(torch.int64), ucap)
        if inv is None or inv.numel() != n or inv.dtype != torch.int32:
            inv = self.local.inverse_index(uptr, upos, U, n)      # position -> slot of its item in `items`
        uq = (items, uptr[:U + 1], upos) + tuple(uniq[3:])
        self.lab_all = self.lab_pre = None
        rows = ShardedLookup.apply(table, items_pad, inv, uq, self.dE, self.lo, self.local, self.group, self, drop)
        self.labels_hint = None
        return rows

    def loss(self, sr, table, cs, labels, cs_inv_scale):
        B = sr.shape[0] * self.world
        key = B
        if key not in self._ws:
            self._ws[key] = self.local.workspace(B, table.shape[0], self.d, sr.device)
        live = table[:self.n_live] if self.n_live < table.shape[0] else table
        dE = self.dE[:self.n_live] if self.n_live < table.shape[0] else self.dE
        csl = None if cs is None else cs[:self.n_live]
        lab_all, self.lab_all = self.lab_all, None
        if lab_all is not None and lab_all.numel() != B:
            lab_all = None
        lab_pre, self.lab_pre = self.__dict__.get('lab_pre'), None
        out = ShardedScoreCE.apply(sr, live, csl, labels, dE, self.lo, self._ws[key], cs_inv_scale, self.local, self.group,
                                   lab_all, self.tgrad, lab_pre if lab_all is not None else None)
        self.tgrad.fresh = True                      # the backward of `out` overwrites every live row of dE
        return out

    def _live(self, table, cs):
        if self.n_live < table.shape[0]:
            return table[:self.n_live], self.dE[:self.n_live], (None if cs is None else cs[:self.n_live])
        return table, self.dE, cs

    def _workspace(self, B, table, device):
        if B not in self._ws:
            self._ws[B] = self.local.workspace(B, table.shape[0], self.d, device)
        return self._ws[B]

    def stats(self, sr, table, cs, labels, cs_inv_scale):
        """(lse, label logit) of this rank's sessions over the whole catalog (see ShardedScoreStats).  The caller
        clears `self.tgrad.fresh` once per step before the first head."""
        live, dE, csl = self._live(table, cs)
        ws = self._workspace(sr.shape[0] * self.world, table, sr.device)
        return ShardedScoreStats.apply(sr, live, csl, labels, dE, self.lo, ws, cs_inv_scale, self.local, self.group,
                                       self.tgrad)

    def log_probs(self, sr, table, cs, data_parallel=False):
        """(B, V) log-probabilities (the reference models' forward() output; compat / evaluation path, no gradient):
        local log-sum-exp per shard -> global lse -> each rank's columns -> all-gather of the column blocks."""
        with torch.no_grad():
            n_loc = sr.shape[0]
            sr_all = all_gather_cat(sr.contiguous(), self.group) if data_parallel else sr.contiguous()
            B = sr_all.shape[0]
            live, _, csl = self._live(table, cs)
            ws = self._workspace(B, table, sr.device)
            none = torch.full((B,), -1, dtype=torch.int32, device=sr.device)
            lse_r, _ = self.local.ce_fwd(sr_all, live, csl, none, ws)
            gathered = _active(self.group)
            lse = torch.logsumexp(all_gather_cat(lse_r.unsqueeze(0), self.group), dim=0).contiguous() if gathered else lse_r
            cols = self.local.logp_cols(sr_all, live, csl, lse)                    # [B, n_live]
            if gathered:
                pad = cols.new_full((B, self.per), float('-inf'))
                pad[:, :self.n_live] = cols
                allc = all_gather_cat(pad.t().contiguous(), self.group)            # [world*per, B]
                cols = allc[:self.V].t().contiguous()
            if data_parallel:
                cols = cols[self.rank * n_loc:(self.rank + 1) * n_loc]
            return cols

    def topk(self, sr, table, cs, k, data_parallel=False):
        """evaluation over the sharded table (SURVEY 8(e) "Eval"): every rank ranks its own rows with the fused
        top-k kernel, the (B, k) lists are all-gathered and merged (ties -> lower item id, like one device).
        data_parallel=False: every rank passes the SAME sessions (replicated evaluation loader) and gets the full
        answer.  data_parallel=True: each rank passes its own B/world sessions (same count everywhere); the session
        vectors are all-gathered first and each rank gets the answer for its own sessions."""
        n_loc = sr.shape[0]
        sr_all = all_gather_cat(sr.contiguous(), self.group) if data_parallel else sr
        live = table[:self.n_live]
        csl = None if cs is None else cs[:self.n_live]
        kk = min(k, self.n_live)
        B = sr_all.shape[0]
        if kk > 0:
            val, idx = self.local.topk(sr_all, live, csl, kk)
            idx = idx.to(torch.int64) + self.lo
        else:
            val = sr_all.new_empty(B, 0)
            idx = torch.empty(B, 0, dtype=torch.int64, device=sr.device)
        if kk < k:                                   # a shard with fewer than k live rows: pad with -inf
            val = torch.cat([val, val.new_full((B, k - kk), float('-inf'))], 1)
            idx = torch.cat([idx, idx.new_full((B, k - kk), 2 ** 62)], 1)
        if _active(self.group):
            w = self.world
            val = all_gather_cat(val, self.group).view(w, B, k).permute(1, 0, 2).reshape(B, w * k)
            idx = all_gather_cat(idx, self.group).view(w, B, k).permute(1, 0, 2).reshape(B, w * k)
        # merge: descending value, ties towards the lower item id (two stable sorts)
        o = torch.argsort(idx, dim=1, stable=True)
        val, idx = val.gather(1, o), idx.gather(1, o)
        o = torch.argsort(val, dim=1, descending=True, stable=True)[:, :k]
        val, idx = val.gather(1, o), idx.gather(1, o)
        if data_parallel:
            val, idx = val[self.rank * n_loc:(self.rank + 1) * n_loc], idx[self.rank * n_loc:(self.rank + 1) * n_loc]
        return val, idx.to(torch.int32)

    # ---- replicated-parameter gradients: bucketed all-reduce, launched in backward order ----------------------------------
    N_BUCKETS = 3

    def _bucket_index(self, p, order):
        """bucket of a replicated parameter: the model's own grouping by backward completion (model.grad_bucket(name): 0 = the
        gradients that are complete first), else thirds of the parameter list from its end (the last layers' gradients come
        first) - a function of the parameter list only, identical on every rank"""
        fn = getattr(self.model, 'grad_bucket', None)
        if fn is not None:
            return max(0, min(self.N_BUCKETS - 1, int(fn(self._names.get(id(p), '')))))
        i, n = order
        return min(self.N_BUCKETS - 1, (n - 1 - i) * self.N_BUCKETS // max(n, 1))

    def _arena_setup(self, parts, n_flags):
        """one flat fp32 ARENA per bucket, laid out in parameter order; the last one ends in the flag tail.  Every bucketed
        parameter gets its slot (`p._srec_gslot`): the backward nodes that allocate a gradient write it straight into the
        slot (ops.grad_buf), the all-reduce runs in place on the arena, the optimizer reads views of it - no concatenation
        pass, no copy back (round 5: four torch.cat kernels, 23 us of the rank step)."""
        for p in getattr(self, '_bucket_live', ()):
            p.__dict__.pop('_srec_gslot', None)
        self._arena, self._slot = [], {}
        last = len(parts) - 1
        for k, bp in enumerate(parts):
            n = sum((int(p.numel()) + 3) // 4 * 4 for p in bp) + (n_flags if k == last else 0)
            dev = bp[0].device if bp else self.dE.device
            buf = torch.zeros(n, device=dev, dtype=torch.float32) if n else None
            off = 0
            for p in bp:
                assert p.dtype == torch.float32, 'replicated parameters are fp32'
                self._slot[id(p)] = (k, off, int(p.numel()))
                p._srec_gslot = [buf, off, -1]
                off += (int(p.numel()) + 3) // 4 * 4       # slots start on 16-byte boundaries (the kernels' float4 accesses)
            self._arena.append(buf)

    def release_slots(self):
        """detach the parameters from the arenas (the model goes on without this VocabParallel)"""
        for p in getattr(self, '_bucket_live', ()):
            p.__dict__.pop('_srec_gslot', None)

    def _zero_of(self, p):
        z = self._bucket_zero.get(id(p))
        if z is None:
            z = self._bucket_zero[id(p)] = torch.zeros(p.numel(), device=p.device, dtype=torch.float32)
        return z

    @staticmethod
    def _copy_tasks(tasks):
        """[(src flat, dst flat)]: ONE launch for all of them on the GPU (the multi-slab sum with one slab per task)"""
        if not tasks:
            return
        if tasks[0][1].is_cuda:
            from . import ops
            # (16-byte column threads where sizes and addresses allow, the scalar row-lane form otherwise)
            odd = lambda a, b: bool(a.numel() & 3 or (a.data_ptr() | b.data_ptr()) & 15)
            ops._launch_slab_sums([(src.reshape(1, -1), dst, odd(src, dst)) for src, dst in tasks])
        else:
            for src, dst in tasks:
                dst.copy_(src)

    def _bucket_fill(self, k, bp, tail=None):
        """make arena k hold this rank's contribution: gradients already written into their slots stay, the others are copied
        in (a gradient autograd summed from two nodes, a slot taken twice, an aten-made gradient), parameters without a
        gradient on this rank contribute zeros; `tail` = the flag tail of the last bucket.  One copy launch for all of it."""
        buf = self._arena[k]
        tasks = []
        esz = 4
        for p in bp:
            _, off, n = self._slot[id(p)]
            dst = buf[off:off + n]
            g = p.grad
            if g is None:
                tasks.append((self._zero_of(p), dst))
            elif g.data_ptr() != buf.data_ptr() + off * esz or not g.is_contiguous():
                tasks.append((g.contiguous().reshape(-1), dst))
        if tail is not None:
            tasks.append((tail, buf[buf.numel() - tail.numel():]))
        self._copy_tasks(tasks)
        self.copy_tasks_last = getattr(self, 'copy_tasks_last', {})
        self.copy_tasks_last[k] = len(tasks)
        return buf

    def _bucket_reduce(self, k, bp, side, tail=None):
        """fill arena k (see _bucket_fill) and all-reduce it IN PLACE - on the side stream when `side` (the collective then runs
        beside whatever the compute stream does next; sync_replicated_grads joins before the optimizer)"""
        if side:
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=self._arena[k].device)
            self._side.wait_stream(cur)                # the gradients are complete; the buffer's last reader (Adam) is behind us
            with torch.cuda.stream(self._side):
                flat = self._bucket_fill(k, bp, tail)
                all_reduce_(flat, None, self.group)
            self._side_used = True
        else:
            flat = self._bucket_fill(k, bp, tail)
            all_reduce_(flat, None, self.group)
        return flat

    def begin_step(self):
        """called at the head of every training forward (lookup): a new hand-out round of the arena slots, and the state of a
        step that never reached sync_replicated_grads is dealt with - an aborted backward / a refused capture (no gradients
        held any more: forget its early launches) or, loudly, a second backward on top of all-reduced gradients."""
        from . import ops
        ops.grad_epoch()
        if self._bucket_early or self._side_used:
            held = any(p.grad is not None for p in getattr(self, '_bucket_live', ()))
            if held and not self._no_sync:
                raise RuntimeError('a backward pass launched the all-reduce of its gradient buckets and no '
                                   'sync_replicated_grads() / optimizer step followed: the gradients held are partly summed '
                                   'over the ranks.  Accumulating micro-batches: run all but the last under `with '
                                   'shard.no_sync():`, or zero_grad() first.')
            if self._side_used and self._side is not None and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().wait_stream(self._side)
            self._bucket_early, self._side_used = {}, False

    def abort_step(self):
        """a capture of the step was refused / a step was abandoned: nothing of it ran or will be used - forget its early
        bucket launches (GraphedTrainStep's failure path; the eager fallback starts from a clean slate)"""
        self._bucket_early, self._side_used = {}, False

    def no_sync(self):
        """context for gradient accumulation (like DDP's): backward passes inside it launch no bucket all-reduce; the LAST
        micro-batch runs outside it and sync_replicated_grads() then exchanges the accumulated sums once."""
        vp = self

        class _NoSync:
            def __enter__(self_):
                self_.prev, vp._no_sync = vp._no_sync, True

            def __exit__(self_, *exc):
                vp._no_sync = self_.prev
                return False
        return _NoSync()

    def bucket_ready(self, k):
        """called from inside the backward pass (ops.grad_mark placed by the model) when every gradient of bucket k has been
        accumulated: its all-reduce is issued NOW, under the rest of the backward, instead of after it.  Never for the last
        bucket (it carries the presence flags).  The decision depends on rank-AGREED state only (the bucket partition fixed by
        agree(), the marker the model places in every training forward): every rank issues the same collectives in the same
        order; a parameter without a gradient on this rank contributes zeros."""
        parts = getattr(self, '_bucket_parts', None)
        if not _active(self.group) or parts is None or self._no_sync or not (0 <= k < len(parts) - 1) \
                or k in self._bucket_early or not parts[k]:
            return
        self.ops_flush()                               # deferred slab sums of the layers behind us: their gradients are written
        self._bucket_early[k] = self._bucket_reduce(k, parts[k], self.side_stream)
        self.early_launches += 1

    def ops_flush(self):
        from . import ops
        ops.flush_deferred()

    def sync_replicated_grads(self, params, optimizer=None):
        """sum the replicated-parameter gradients over ranks in N_BUCKETS flat buckets whose layout is the same on every
        rank, in the order the backward pass completes them (bucket 0 first).  A bucket whose gradients were all complete
        when the backward passed the model's marker (bucket_ready) is already on its way - on a side stream when
        `side_stream` - ; the rest, always including the last one, are issued here, and the compute stream waits for the
        side stream before the optimizer reads the result.
        `params` is the model's replicated parameter list (same order everywhere).  Which of them carry a gradient can
        differ between ranks and between steps (MSHGNN only instantiates the GAT modules of relations with live edges in
        THIS rank's batch).  The buckets hold the parameters that had a gradient on ANY rank when the layout was agreed
        on (first call: all-reduce(MAX) of a presence mask, outside any graph capture), a rank without a gradient for one
        contributing zeros; the LAST bucket ends in a tail of flags that rides in its all-reduce: one presence count per
        bucketed parameter and one count of ranks holding a gradient for a parameter OUTSIDE the layout.  Eager steps
        read the tail back (one host sync): a parameter nobody had a gradient for this step is skipped like on one
        device (no weight decay on a zero gradient), and a late parameter makes every rank re-agree on the layout and
        repeat the exchange - the same decision everywhere, because it is taken on all-reduced values.  A captured
        step replays the layout and the skip pattern of its capture.  With `optimizer` (FusedAdam) the reduced buckets are
        handed over as the gradient source (views), so nothing is copied back per parameter."""
        if not _active(self.group):
            return
        if self._no_sync:
            raise RuntimeError('sync_replicated_grads() inside no_sync(): run the last micro-batch outside the context')
        params = list(params)
        key = tuple(id(p) for p in params)
        capturing = bool(params) and params[0].is_cuda and torch.cuda.is_current_stream_capturing()
        dev = params[0].device if params else self.dE.device

        def agree():
            if capturing:
                raise RuntimeError('the gradient bucket layout must be agreed on by an eager step before graph capture')
            present = torch.tensor([1.0 if p.grad is not None else 0.0 for p in params], device=dev)
            if self.world > 1:
                all_reduce_(present, dist.ReduceOp.MAX, self.group)
            old = getattr(self, '_bucket_ids', set()) if getattr(self, '_bucket_key', None) == key else set()
            live = [p for p, f in zip(params, present.tolist()) if f > 0 or id(p) in old]
            n = len(params)
            idx = {id(p): i for i, p in enumerate(params)}
            parts = [[] for _ in range(self.N_BUCKETS)]
            for p in live:
                parts[self._bucket_index(p, (idx[id(p)], n))].append(p)
            self._arena_setup(parts, len(live) + 1)        # (detaches the old layout's parameters from their slots first)
            self._bucket_live = live
            self._bucket_ids = {id(p) for p in live}
            self._bucket_key = key
            self._bucket_zero = {}
            self._bucket_flags = None
            self._bucket_parts = parts
            self._bucket_early = {}                        # (early results were laid out for the old partition)
            self._bucket_seen = None
            return old
        if getattr(self, '_bucket_key', None) != key:
            agree()
        ps, parts, arenas, slots = self._bucket_live, self._bucket_parts, self._arena, self._slot
        n_late = sum(1 for p in params if p.grad is not None and id(p) not in self._bucket_ids)
        if n_late and capturing:
            raise RuntimeError('%d replicated parameters carry a gradient that the captured bucket layout has no room for' % n_late)
        pat = tuple(p.grad is not None for p in ps) + (n_late,)
        fl = self._bucket_flags
        if fl is None or fl[0] != pat:                     # the tail changes only when this rank's presence pattern does
            if capturing and fl is not None:
                raise RuntimeError('gradient presence pattern changed between the warm-up and the capture')
            fl = self._bucket_flags = (pat, torch.tensor([1.0 if f else 0.0 for f in pat[:-1]] + [float(n_late)],
                                                        device=dev, dtype=torch.float32))
        flats = []
        last = len(parts) - 1
        for k, bp in enumerate(parts):
            if k in self._bucket_early:
                flats.append(self._bucket_early[k])
                continue
            # the flag tail is read by eager steps only: a captured step replays the skip pattern of its capture (the tail's
            # slots still travel - every rank all-reduces the same length - but nobody refreshes or reads them)
            tail = fl[1] if (k == last and not capturing) else None
            # (under capture also the late buckets leave on the side stream: the table's Adam pass runs beside them)
            flats.append(self._bucket_reduce(k, bp, self.side_stream and capturing, tail) if arenas[k] is not None else None)
        self._bucket_early = {}
        if self._side_used:
            # join: the early buckets become visible to the compute stream.  FusedAdam takes the join over and places it
            # behind its row pass over the table, which needs none of these gradients (optim.FusedAdam.grad_join) - unless
            # this (eager) step is about to read the flag tail back, which synchronises anyway
            self._side_used = False
            if capturing and optimizer is not None and hasattr(optimizer, '_join_grads'):
                side = self._side
                optimizer.grad_join = lambda: torch.cuda.current_stream().wait_stream(side)
            else:
                torch.cuda.current_stream().wait_stream(self._side)
        BUCKETS['bytes'] = [0 if f is None else int(f.numel()) * 4 for f in flats]
        nfl = len(ps) + 1
        pos = {id(p): i for i, p in enumerate(ps)}
        extra = {}
        if not capturing:
            seen = flats[last][-nfl:].tolist()
            self._bucket_seen = seen
            if seen[-1] > 0:
                # somebody holds a gradient the layout has no slot for - everybody sees the same count.  The buckets above are
                # exchanged and stay as they are (the all-reduce ran in place: the local values are gone); the ranks agree on a
                # layout with room for the newcomers (from the next step on) and exchange THEIR gradients in one more
                # all-reduce now - nothing dropped, no rank raises alone
                old_ids = agree()
                late = [p for p in self._bucket_live if id(p) not in old_ids]
                pieces = [(p.grad.reshape(-1).to(torch.float32) if p.grad is not None else self._zero_of(p)) for p in late]
                pieces.append(torch.tensor([1.0 if p.grad is not None else 0.0 for p in late], device=dev))
                buf = torch.cat(pieces)
                all_reduce_(buf, None, self.group)
                seen_late = buf[-len(late):].tolist()
                off = 0
                for p, f in zip(late, seen_late):
                    if f > 0:
                        extra[id(p)] = buf[off:off + p.numel()].view_as(p)
                    off += p.numel()
        else:
            seen = getattr(self, '_bucket_seen', None)
        views = dict(extra)
        for p in ps:
            k, off, n = slots[id(p)]
            if seen is None or seen[pos[id(p)]] > 0:       # nobody had a gradient this step: skipped, as on one device
                views[id(p)] = arenas[k][off:off + n].view_as(p)
        if optimizer is not None and hasattr(optimizer, 'grad_override'):
            optimizer.grad_override = views
        else:
            for p in params:
                v = views.get(id(p))
                if v is None:
                    continue
                if p.grad is None:
                    p.grad = v.clone()
                elif p.grad.data_ptr() != v.data_ptr():
                    p.grad.copy_(v)
