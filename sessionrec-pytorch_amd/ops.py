"""torch.autograd.Function wrappers over the C ABI (include/srec.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the
autograd tape; every forward AND backward below is a hand-written gfx950 kernel
launched through libsrec_hip.so.  CPU tensors raise (no fallback).

`dyn` arguments are optional 1-element int32 device tensors holding the live
extent of a capacity-padded dimension (see batch.FlatBatch.dyn).
"""
import ctypes as _ct
import os

import torch

from ._lib import lib, ptr, stream


def _rows(t):
    """2-D fp32 tensor with unit inner stride (row-strided views allowed) -> tensor usable as (ptr, ld)."""
    assert t.dim() == 2 and t.dtype == torch.float32, (t.shape, t.dtype)
    if t.stride(1) != 1 or (t.stride(0) & 3) or (t.data_ptr() & 15) or (t.shape[0] > 1 and t.stride(0) < t.shape[1]):
        t = t.contiguous()
    return t


_WS = {}


def _ws(n, device):
    """grow-only fp32 scratch per device (column-sum partials, split-K slabs)"""
    t = _WS.get(str(device))
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 22), device=device, dtype=torch.float32)
        _WS[str(device)] = t
    return t


_GEMM_WS = {}




def _gemm_ws(device):
    t = _GEMM_WS.get(str(device))
    if t is None:
        t = _GEMM_WS[str(device)] = torch.empty(1 << 24, device=device, dtype=torch.float32)   # 64 MiB of split-K slabs
    return t.data_ptr(), t.numel()


def col_sum(X, n, ncol, out, dyn=None, wgt=None, H=1, D=1, accumulate=0):
    ws = _ws(128 * ncol, X.device)            # NCHUNK_V partial rows of the vector path (32 for the weighted one)
    lib.srec_col_sum(ptr(X), _ld(X), ptr(wgt), H, D, n, ptr(dyn), ncol, ptr(out), accumulate, ptr(ws), stream())


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


_LIMITS = None


def limits():
    """static per-session budgets of the kernels (include/srec.h srec_limits)"""
    global _LIMITS
    if _LIMITS is None:
        a, b, c = _ct.c_int(), _ct.c_int(), _ct.c_int()
        lib.srec_limits(_ct.addressof(a), _ct.addressof(b), _ct.addressof(c))
        _LIMITS = dict(nodes=a.value, deg=b.value, sgat_deg=c.value)
    return _LIMITS


def copy_words(src, dst, n):
    """dst[:n] (device int32) = src[:n] (page-locked host or device int32) by a kernel on the current stream (srec_copy_words)"""
    assert dst.is_cuda and src.dtype == dst.dtype == torch.int32 and (src.is_cuda or src.is_pinned())
    lib.srec_copy_words(src.data_ptr(), dst.data_ptr(), int(n), stream())


def check_limits(mg, deg=None):
    """a batch whose longest session / largest node degree exceeds what the per-session kernels hold in LDS is refused here
    (collate records both in FlatBatch.meta): a clean error, never a silently truncated soft-max.  Host-only (two meta
    values): called by the models' forward AND by graph.GraphedTrainStep for every replayed batch - a replay never runs
    the models' Python again.  deg: which degree budget applies (default: by the batch kind - LESSR's shortcut graphs
    go through the SGAT kernels)"""
    L, m = limits(), mg.meta
    if deg is None:
        deg = 'sgat_deg' if m.get('kind') == 'shortcut' else 'deg'
    if m.get('max_nodes', 0) > L['nodes']:
        raise ValueError('a session of this batch has %d read-out nodes; the per-session kernels hold at most %d '
                         '(SREC_MAX_SESSION_NODES, csrc/common.h): truncate sessions (the reference preprocessing keeps the '
                         'last 20 clicks, preprocess.py:45-50) or lower the n-gram order' % (m['max_nodes'], L['nodes']))
    if m.get('max_deg', 0) > L[deg]:
        raise ValueError('a node of this batch has degree %d in one relation; the graph-attention kernels hold at most %d '
                         '(csrc/common.h): truncate sessions' % (m['max_deg'], L[deg]))


FUSED_GRU = True        # tests flip these two to compare the fused kernels (gruf.hip / grufb.hip, headf.hip) with the
FUSED_HEAD = True       # step-by-step / grouped launch sequences they replace; the product path never does
PRECISION = {'matmul': 'fp32'}      # 'fp32': exact fp32 MFMA everywhere; 'bf16': bf16-operand MFMA for the
                                    # forward / backward-data products and the scoring kernels (config C3)
_WT_CACHE = {}


def set_precision(mode):
    assert mode in ('fp32', 'bf16')
    PRECISION['matmul'] = mode
    weights_changed()


def weights_changed():
    """called by the optimizer after it updated parameters: cached transposed weight copies are stale"""
    _WT_CACHE.clear()
    _HEAD_WF_CACHE.clear()
    _WPREP.clear()


# bf16 operand copies of weights made by the prologue launch of the CURRENT forward pass (step_prologue) for the one node that
# reads them (weights_bf16 in HGATLayer, gru_wfrag_both in GRUExpandAll): (kind, data_ptr, shape) -> (tensor version, buffers).
# The reader TAKES its entry - nothing outlives the forward it was made for; weights_changed drops what was never read.
_WPREP = {}


def _wprep_take(kind, w):
    e = _WPREP.pop((kind, w.data_ptr(), tuple(w.shape)), None)
    return e[1] if e is not None and e[0] == w._version else None


def _wprep_put(kind, w, bufs):
    _WPREP[(kind, w.data_ptr(), tuple(w.shape))] = (w._version, bufs)


def _transposed(w):
    key = (w.data_ptr(), tuple(w.shape), w.stride(0))
    t = _WT_CACHE.get(key)
    if t is None:
        t = _WT_CACHE[key] = w.t().contiguous()
    return t


def _bf16_ok(K, *ts):
    return PRECISION['matmul'] == 'bf16' and K % 32 == 0 and all((_ld(t) & 3) == 0 and (t.data_ptr() & 15) == 0 for t in ts)


def gemm_nt(x, w, out, bias=None, dyn=None, dyn_mode=0, beta=0.0):
    """out[M,N] = x[M,K] @ w[N,K]^T (+bias) (+beta*out)"""
    M, K = x.shape
    N = w.shape[0]
    if dyn_mode in (0, 1) and _bf16_ok(K, x, w):
        lib.srec_gemm_bf16_nt(ptr(x), _ld(x), ptr(w), _ld(w), ptr(out), _ld(out), ptr(bias), M, N, K,
                              ptr(dyn) if dyn_mode == 1 else None, 1.0, beta, *_gemm_ws(x.device), stream())
        return
    lib.srec_gemm_f32(ptr(x), _ld(x), 1, ptr(w), _ld(w), 1, ptr(out), _ld(out), ptr(bias), M, N, K, ptr(dyn), dyn_mode,
                      1.0, beta, *_gemm_ws(x.device), stream())


def gemm_nn(g, w, out, dyn=None, dyn_mode=0, beta=0.0):
    """out[M,K] = g[M,N] @ w[N,K]"""
    M, N = g.shape
    K = w.shape[1]
    if dyn_mode in (0, 1) and _bf16_ok(N, g) and not (K & 3) and not (_ld(w) & 3) and not (w.data_ptr() & 15):
        # B operand reduction-major: the weight is transposed while it is staged into LDS (no transposed copy, no
        # split-K slabs: gemm_group_bf16.hip)
        gemm_group(1, [(M, K, N, [(g, w)], out, dyn if dyn_mode == 1 else None)], _ld(g), _ld(w), _ld(out), beta)
        return
    lib.srec_gemm_f32(ptr(g), _ld(g), 1, ptr(w), 1, _ld(w), ptr(out), _ld(out), None, M, K, N, ptr(dyn), dyn_mode, 1.0,
                      beta, *_gemm_ws(g.device), stream())


def gemm_tn(g, x, out, dyn=None, beta=0.0):
    """out[N,K] = g[M,N]^T @ x[M,K]   (reduction over the M rows; dyn clamps M)"""
    M, N = g.shape
    K = x.shape[1]
    if (PRECISION['matmul'] == 'bf16' and M >= 256 and not ((N | K | _ld(g) | _ld(x) | _ld(out)) & 3)
            and not ((g.data_ptr() | x.data_ptr() | out.data_ptr()) & 15)):
        lib.srec_gemm_bf16_tn(ptr(g), _ld(g), ptr(x), _ld(x), ptr(out), _ld(out), M, N, K, ptr(dyn), 1.0, beta,
                              *_gemm_ws(g.device), stream())
        return
    lib.srec_gemm_f32(ptr(g), 1, _ld(g), ptr(x), 1, _ld(x), ptr(out), _ld(out), None, N, K, M, ptr(dyn),
                      2 if dyn is not None else 0, 1.0, beta, *_gemm_ws(g.device), stream())


_SMALL_LINEAR = 128 * 256       # weight elements up to which a linear layer is launch-bound, not flop-bound


def _small_linear(*ws):
    return sum(w.numel() for w in ws) <= _SMALL_LINEAR


def _linear_backward_group(gy, xs, ws, gws, need_x, need_b, dyn, defer=False):
    """backward of y = sum_i x_i W_i^T + b as ONE grouped exact-fp32 launch (+ its slab sum): d x_i = gy W_i, d W_i =
    gy^T x_i into gws[i] (None: not needed), d b = column sums of gy as a product with a block of ones - instead of a GEMM
    per product, a split-K sum per weight gradient and two column-sum launches.  -> ([d x_i], d b)"""
    M, N = gy.shape
    probs, gxs = [], []
    for x, w, gw, nx in zip(xs, ws, gws, need_x):
        gx = None
        if nx:
            gx = torch.empty_like(x, memory_format=torch.contiguous_format)
            probs.append(('nn', gy, w, gx, None, dyn, 0.0))
        if gw is not None:
            probs.append(('tn', gy, x, gw, None, dyn, 0.0))
        gxs.append(gx)
    gb = None
    if need_b:
        sums = torch.empty(4, N, device=gy.device, dtype=torch.float32)
        probs.append(('tn', _ones4(M, gy.device)[:M], gy, sums, None, dyn, 0.0))
        gb = sums[0]
    outs = ([gw for gw in gws if gw is not None] + ([sums] if need_b else [])) if defer else None
    for c in range(0, len(probs), 16):
        gemm_f32_group(probs[c:c + 16], defer=outs)
    return gxs, gb


class LinearCat(torch.autograd.Function):
    """y = sum_i x_i @ W[:, off_i:off_i+k_i]^T + b   == nn.Linear(cat(x_i, dim=1)) without the concat.  Exact fp32 products
    when asked for (exact) and for small weights (launch-bound either way: one rounding fewer, and the backward becomes
    one grouped launch)."""

    @staticmethod
    def forward(ctx, weight, bias, dyn, exact, *xs):
        exact = exact or _small_linear(weight)
        ctx.exact = exact
        ctx.defer, ctx.wparams = defer_scope(), [weight] + ([bias] if bias is not None else [])
        if exact and PRECISION['matmul'] != 'fp32':         # session-vector head: exact fp32 MFMA even in bf16 mode
            prev, PRECISION['matmul'] = PRECISION['matmul'], 'fp32'
            try:
                return LinearCat._fwd(ctx, weight, bias, dyn, xs)
            finally:
                PRECISION['matmul'] = prev
        return LinearCat._fwd(ctx, weight, bias, dyn, xs)

    @staticmethod
    def _fwd(ctx, weight, bias, dyn, xs):
        xs = [_rows(x) for x in xs]
        M = xs[0].shape[0]
        N = weight.shape[0]
        w = _rows(weight)
        y = torch.empty(M, N, device=w.device, dtype=torch.float32)
        off = 0
        for i, x in enumerate(xs):
            k = x.shape[1]
            gemm_nt(x, w[:, off:off + k], y, bias if i == 0 else None, dyn, 1 if dyn is not None else 0,
                    0.0 if i == 0 else 1.0)
            off += k
        assert off == w.shape[1], (off, w.shape)
        ctx.save_for_backward(w, *xs)
        ctx.dyn = dyn
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.exact and PRECISION['matmul'] != 'fp32':
            prev, PRECISION['matmul'] = PRECISION['matmul'], 'fp32'
            try:
                return LinearCat._bwd(ctx, gy)
            finally:
                PRECISION['matmul'] = prev
        return LinearCat._bwd(ctx, gy)

    @staticmethod
    def _bwd(ctx, gy):
        w, *xs = ctx.saved_tensors
        dyn = ctx.dyn
        gy = _rows(gy)
        need_x = ctx.needs_input_grad[4:]
        gw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        if _small_linear(w) and gy.shape[0] > 0:
            offs = [0]
            for x in xs:
                offs.append(offs[-1] + x.shape[1])
            gxs, gb = _linear_backward_group(gy, xs, [w[:, a:b] for a, b in zip(offs, offs[1:])],
                                             [None if gw is None else (gw if len(xs) == 1 else gw[:, a:b]) for a, b in zip(offs, offs[1:])],
                                             need_x, ctx.has_bias and ctx.needs_input_grad[1], dyn,
                                             defer=can_defer(ctx.defer, ctx.wparams))
            return (gw, gb, None, None) + tuple(gxs)
        gb = None
        gxs = []
        off = 0
        for i, x in enumerate(xs):
            k = x.shape[1]
            if need_x[i]:
                gx = torch.empty_like(x, memory_format=torch.contiguous_format)
                gemm_nn(gy, w[:, off:off + k], gx, dyn, 1 if dyn is not None else 0)
                gxs.append(gx)
            else:
                gxs.append(None)
            if gw is not None:
                gemm_tn(gy, x, gw[:, off:off + k], dyn)
            off += k
        if ctx.has_bias and ctx.needs_input_grad[1]:
            gb = torch.empty(w.shape[0], device=w.device, dtype=torch.float32)
            col_sum(gy, gy.shape[0], w.shape[0], gb, dyn)
        return (gw, gb, None, None) + tuple(gxs)


class LinearSum(torch.autograd.Function):
    """y = sum_i x_i @ W_i^T (+ b): nn.Linear(cat(x_i, dim=1)) with the weight kept as separate blocks W_i [out, k_i] - no
    concatenation of weights in the forward, no slicing of a combined weight gradient in the backward (LESSR's EOPA:
    fc_self(feat) + fc_neigh(neigh), lessr.py:36-38)"""

    @staticmethod
    def forward(ctx, bias, dyn, n, *args):
        xs, ws = [_rows(x) for x in args[:n]], [_rows(w) for w in args[n:]]
        M, N = xs[0].shape[0], ws[0].shape[0]
        y = torch.empty(M, N, device=xs[0].device, dtype=torch.float32)
        ctx.small = _small_linear(*ws)
        prev = PRECISION['matmul']
        if ctx.small:
            PRECISION['matmul'] = 'fp32'
        try:
            for i, (x, w) in enumerate(zip(xs, ws)):
                gemm_nt(x, w, y, bias if i == 0 else None, dyn, 1 if dyn is not None else 0, 0.0 if i == 0 else 1.0)
        finally:
            PRECISION['matmul'] = prev
        ctx.save_for_backward(*xs, *ws)
        ctx.dyn, ctx.n, ctx.has_bias = dyn, n, bias is not None
        ctx.defer, ctx.wparams = defer_scope(), list(args[n:]) + ([bias] if bias is not None else [])
        return y

    @staticmethod
    def backward(ctx, gy):
        n, dyn = ctx.n, ctx.dyn
        xs, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        gy = _rows(gy)
        if ctx.small and gy.shape[0] > 0:
            gws = [torch.empty_like(w) if ctx.needs_input_grad[3 + n + i] else None for i, w in enumerate(ws)]
            gxs, gb = _linear_backward_group(gy, xs, ws, gws, ctx.needs_input_grad[3:3 + n],
                                             ctx.has_bias and ctx.needs_input_grad[0], dyn, defer=can_defer(ctx.defer, ctx.wparams))
            return (gb, None, None) + tuple(gxs) + tuple(gws)
        gxs, gws = [], []
        for i, (x, w) in enumerate(zip(xs, ws)):
            gx = None
            if ctx.needs_input_grad[3 + i]:
                gx = torch.empty_like(x, memory_format=torch.contiguous_format)
                gemm_nn(gy, w, gx, dyn, 1 if dyn is not None else 0)
            gw = None
            if ctx.needs_input_grad[3 + n + i]:
                gw = torch.empty_like(w)
                gemm_tn(gy, x, gw, dyn)
            gxs.append(gx)
            gws.append(gw)
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[0]:
            gb = torch.empty(ws[0].shape[0], device=gy.device, dtype=torch.float32)
            col_sum(gy, gy.shape[0], ws[0].shape[0], gb, dyn)
        return (gb, None, None) + tuple(gxs) + tuple(gws)


def linear_sum(xs, ws, bias=None, dyn=None):
    return LinearSum.apply(bias, dyn, len(xs), *xs, *ws)


def linear(x, weight, bias=None, dyn=None, exact=False):
    """exact=True: fp32 MFMA regardless of set_precision (the readout / session-vector head, whose output is scaled
    by 12 before the soft-max in NISER / MSGIFSR)"""
    return LinearCat.apply(weight, bias, dyn, exact, x)


def linear_cat(xs, weight, bias=None, dyn=None, exact=False):
    return LinearCat.apply(weight, bias, dyn, exact, *xs)


class CatCols(torch.autograd.Function):
    """[a | b] along the feature axis; the backward hands out the two column halves of the incoming gradient as views"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _rows(a), _rows(b)
        n, da, db = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty(n, da + db, device=a.device, dtype=torch.float32)
        lib.srec_cat_cols(ptr(a), _ld(a), da, ptr(b), _ld(b), db, n, ptr(out), stream())
        ctx.da = da
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.da], g[:, ctx.da:]


def cat_cols(a, b):
    return CatCols.apply(a, b)


# ------------------------------------------------------------------------------------------ dropout masks
RNG_COUNTER = {}     # str(device) -> int32[1] device tensor: the device-side step counter in effect.  A model installs the
#                      counter of ITS FusedAdam at the start of every forward (srgnn._ScoringMixin._lookup); models trained
#                      with another optimizer have none.


_NONCE = {'seed': None, 'gen': None}


def seed_dropout(seed=None):
    """(re)start the nonce stream of the dropout masks: from `seed`, or from torch.initial_seed().  Called implicitly when
    torch.initial_seed() has CHANGED since the last draw (torch.manual_seed with another value); a program that re-seeds
    with the SAME value to replay a run's masks calls it explicitly (re-seeding with an equal value is not observable)."""
    base = torch.initial_seed()
    _NONCE['seed'] = base
    # (seeded with the value itself: the first draws equal what torch's global generator would hand out right after
    #  torch.manual_seed(value) - the masks of a seeded run are the ones it had when the nonces still came from there)
    _NONCE['gen'] = torch.Generator().manual_seed(base if seed is None else int(seed))


def _nonce():
    """a fresh 31-bit draw from a PRIVATE generator seeded from torch.initial_seed(): the nonces follow torch.manual_seed
    but never advance torch's global CPU stream - the stream RandomSampler (NISER / SRGNN shuffling) and user code draw
    from - so the batch order of a seeded run does not depend on how many dropout call sites a step has (eager vs
    replayed steps, model configurations)."""
    if _NONCE['gen'] is None or _NONCE['seed'] != torch.initial_seed():
        seed_dropout()
    return int(torch.randint(0, 0x7fffffff, (1,), generator=_NONCE['gen']).item())


def rng_args(device):
    """(nonce, counter pointer) of the counter-based dropout masks (csrc/common.h srec_rng).  The nonce is a fresh draw
    from a private CPU generator per call (_nonce): it follows torch.manual_seed and renews the masks on every eager forward -
    with any optimizer, and for each of several micro-batches per optimizer step.  The callers keep it for their
    backward, which recomputes the mask.  Under hipGraph replay the kernel arguments (the nonce) are frozen; there the
    device counter, advanced by the captured optimizer step itself, renews the masks - a capture without one would
    replay ONE mask for ever and is refused."""
    c = RNG_COUNTER.get(str(device))
    dev = torch.device(device)
    if c is None and dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('dropout inside a captured step needs a device-side step counter: construct FusedAdam(..., '
                           'model=model) before capturing (a frozen nonce alone would replay the same mask every step)')
    return _nonce(), (c.data_ptr() if c is not None else None)


class TableGrad:
    """Dense gradient buffer of the item table, shared by the scoring backward (writes every row)
    and the embedding-lookup backward (adds its rows in place): no dense+dense autograd sum."""

    def __init__(self, weight, defer=False):
        self.buf = torch.zeros_like(weight)
        self.fresh = False       # True once the scoring backward has overwritten it this step
        # Deferred row-normalisation projection (cosine-scored models, FusedAdam(fuse_projection=True)): the scoring backward
        # leaves `buf` unprojected and records pending = (table, cs, inv_scale); the lookup backward adds its rows AND their
        # radial sums radial[v] += <W_v, l_v>; the optimizer's row pass applies g = G - W (<W, G> - radial) inv^2 while it reads
        # the gradient (srec_adam_rows_proj) - one pass over the table gradient less per step.  Any other reader of the
        # gradient calls materialize() first.
        self.defer = defer
        self.pending = None
        self.radial = torch.zeros(weight.shape[0], device=weight.device, dtype=torch.float32) if defer else None
        self.radial_dirty = False    # a lookup backward has added radial sums that no row pass has consumed yet

    def overwritten(self):
        """the scoring backward has just overwritten every row of `buf` (non-accumulating): whatever an earlier backward
        left behind - a pending projection that no optimizer step consumed, radial sums of lookup gradients that went
        into the old contents (a skipped / NaN-guarded step, two backwards before a step, a training loss evaluated
        without stepping) - belongs to the old contents and goes with them"""
        self.pending = None
        if self.radial is not None and self.radial_dirty:     # (never in the regular backward -> step sequence: no fill
            self.radial.zero_()                               #  kernel in a captured step)
        self.radial_dirty = False

    def reset(self):
        """forget the current contents (optimizer.zero_grad, the undo of a graph-capture warm-up)"""
        self.fresh = False
        self.overwritten()

    def materialize(self):
        """apply a pending projection to `buf` (after it, buf is the true table gradient)"""
        if self.pending is not None:
            table, cs, inv_scale = self.pending
            lib.srec_rownorm_project_radial(ptr(table), table.stride(0), ptr(cs), float(inv_scale), ptr(self.buf),
                                            self.buf.stride(0), table.shape[0], table.shape[1], ptr(self.radial), stream())
            self.pending = None
            self.radial_dirty = False
        return self.buf


class EmbeddingLookup(torch.autograd.Function):
    """rows = table[idx].  Backward = deterministic segmented row sums (no atomics): added in place
    into TableGrad when the fused loss owns the table gradient, else returned as a dense grad."""

    @staticmethod
    def forward(ctx, table, idx, uniq, tgrad, dyn_n, dyn_u, drop=None):
        n = idx.numel()
        d = table.shape[1]
        out = torch.empty(n, d, device=table.device, dtype=torch.float32)
        ctx.drop = None
        if drop is not None and drop[0] > 0:
            # feature dropout fused into the lookup (msgifsr.py:247): mask recomputed by the backward, nothing stored
            seed, cnt = rng_args(table.device)
            ctx.drop = (float(drop[0]), seed, cnt, int(drop[1]))
            lib.srec_gather_rows_drop(ptr(table), table.stride(0), ptr(idx), ptr(out), d, n, ptr(dyn_n), d, ctx.drop[0],
                                      seed, cnt, ctx.drop[3], stream())
        else:
            lib.srec_gather_rows(ptr(table), table.stride(0), ptr(idx), ptr(out), d, n, ptr(dyn_n), d, stream())
        ctx.uniq, ctx.tgrad, ctx.dyn_u, ctx.shape = uniq, tgrad, dyn_u, tuple(table.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        items, uptr, upos = ctx.uniq[:3]
        tg = ctx.tgrad
        g = _rows(g)
        V, d = ctx.shape
        if ctx.drop is not None:
            g = g.contiguous()
            pdrop, seed, cnt, salt = ctx.drop

            def first_level(gp, ldg, it, pt, ps, dst, ldd, ucap, dyn, acc):
                lib.srec_scatter_add_sorted_drop(gp, ldg, it, pt, ps, dst, ldd, ucap, dyn, d, acc, pdrop, seed, cnt, salt, stream())
        else:
            def first_level(gp, ldg, it, pt, ps, dst, ldd, ucap, dyn, acc):
                lib.srec_scatter_add_sorted(gp, ldg, it, pt, ps, dst, ldd, ucap, dyn, d, acc, stream())
        radial = ptable = None
        if tg is not None and tg.pending is not None:     # deferred projection: this gradient's radial part is recorded
            radial, ptable = tg.radial, tg.pending[0]
            tg.radial_dirty = True
        if tg is not None:
            dst, acc, ret = tg.buf, 1, None
        else:
            dst = torch.zeros(V, d, device=g.device, dtype=torch.float32)
            acc, ret = 0, dst
        if len(ctx.uniq) == 5:
            # two balanced levels: <=16 positions per wavefront, then the pieces of each item
            cptr, chunk_ptr = ctx.uniq[3], ctx.uniq[4]
            C = chunk_ptr.numel() - 1
            part = torch.empty(max(C, 1), d, device=g.device, dtype=torch.float32)
            ar = _arange(C + 1, g.device)
            first_level(ptr(g), _ld(g), ptr(ar), ptr(chunk_ptr), ptr(upos), ptr(part), d, C, None, 0)   # padded chunks are
            if radial is not None:                                                                          # empty segments
                lib.srec_scatter_add_sorted_ex(ptr(part), d, ptr(items), ptr(cptr), ptr(ar), ptr(dst), dst.stride(0),
                                               items.numel(), ptr(ctx.dyn_u), d, acc, 0.0, 0, None, 0, ptr(ptable),
                                               ptable.stride(0), ptr(radial), stream())
            else:
                lib.srec_scatter_add_sorted(ptr(part), d, ptr(items), ptr(cptr), ptr(ar), ptr(dst), dst.stride(0),
                                            items.numel(), ptr(ctx.dyn_u), d, acc, stream())
        elif radial is not None:
            pdrop, seed, cnt, salt = ctx.drop if ctx.drop is not None else (0.0, 0, None, 0)
            lib.srec_scatter_add_sorted_ex(ptr(g), _ld(g), ptr(items), ptr(uptr), ptr(upos), ptr(dst), dst.stride(0),
                                           items.numel(), ptr(ctx.dyn_u), d, acc, pdrop, seed, cnt, salt, ptr(ptable),
                                           ptable.stride(0), ptr(radial), stream())
        else:
            first_level(ptr(g), _ld(g), ptr(items), ptr(uptr), ptr(upos), ptr(dst), dst.stride(0), items.numel(),
                        ptr(ctx.dyn_u), acc)
        return ret, None, None, None, None, None, None


def embedding_lookup(table, idx, uniq, tgrad=None, dyn_n=None, dyn_u=None, drop=None):
    """drop = (p, salt): feature dropout of the looked-up rows fused into the gather (and its backward)"""
    return EmbeddingLookup.apply(table, idx, uniq, tgrad, dyn_n, dyn_u, drop)


class RowGather(torch.autograd.Function):
    """out = x[idx] for DISTINCT idx (last-node pick, permutations); backward scatters rows back.  ascending: idx is
    non-negative and strictly ascending (one last node per session, sessions stacked in order) - the backward is then ONE
    launch that writes every row (srec_expand_rows_sorted) instead of a zero fill + a scatter."""

    @staticmethod
    def forward(ctx, x, idx, dyn, ascending=False):
        x = _rows(x)
        n, d = idx.numel(), x.shape[1]
        out = torch.empty(n, d, device=x.device, dtype=torch.float32)
        lib.srec_gather_rows(ptr(x), _ld(x), ptr(idx), ptr(out), d, n, ptr(dyn), d, stream())
        ctx.save_for_backward(idx)
        ctx.nrows, ctx.dyn, ctx.ascending = x.shape[0], dyn, ascending
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _rows(g)
        n, d = g.shape
        if ctx.ascending and d % 4 == 0:
            gx = torch.empty(ctx.nrows, d, device=g.device, dtype=torch.float32)
            lib.srec_expand_rows_sorted(ptr(g), _ld(g), ptr(idx), n, ptr(ctx.dyn), ctx.nrows, d, ptr(gx), d, stream())
            return gx, None, None, None
        gx = torch.zeros(ctx.nrows, d, device=g.device, dtype=torch.float32)
        ar = _arange(n + 1, g.device)
        lib.srec_scatter_add_sorted(ptr(g), _ld(g), ptr(idx), ptr(ar), ptr(ar), ptr(gx), d, n, ptr(ctx.dyn), d, 0,
                                    stream())
        return gx, None, None, None


_ARANGE = {}


def _arange(n, device):
    key = (str(device),)
    t = _ARANGE.get(key)
    if t is None or t.numel() < n:
        t = torch.arange(max(n, 4096), device=device, dtype=torch.int32)
        _ARANGE[key] = t
    return t


def row_gather(x, idx, dyn=None, ascending=False):
    return RowGather.apply(x, idx, dyn, ascending)


class PermuteAndPick(torch.autograd.Function):
    """(x[perm], x[pick_0], x[pick_1], ...) for a PERMUTATION perm of the live rows (inv = its inverse, -1 on padded
    rows) and small distinct pick lists: MSGIFSR's per-session concatenation and last-node picks (msgifsr.py:131-147).
    Backward = one inverse-permutation gather (writes every row, padded rows zero) + one in-place scatter per pick list,
    instead of a zero-fill + scatter per output and autograd's adds."""

    @staticmethod
    def forward(ctx, x, perm, inv, dyn_n, dyn_b, *picks):
        x = _rows(x)
        d = x.shape[1]
        outs = []
        for k, (idx, dyn) in enumerate([(perm, dyn_n)] + [(p, dyn_b) for p in picks]):
            if k == 0:
                o = torch.empty(idx.numel(), d, device=x.device, dtype=torch.float32)
            else:
                # a pick is the left half of a [n, 2 d] buffer: the read-out head (ReadoutHead) puts the session's attention
                # read-out into the right half and has its fc_sr input [x_last | sr_g] without a concatenation kernel
                o = torch.empty(idx.numel(), 2 * d, device=x.device, dtype=torch.float32)[:, :d]
            lib.srec_gather_rows(ptr(x), _ld(x), ptr(idx), ptr(o), _ld(o), idx.numel(), ptr(dyn), d, stream())
            outs.append(o)
        ctx.save_for_backward(inv, *picks)
        ctx.nrows, ctx.dyn_b = x.shape[0], dyn_b
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_perm, *g_picks):
        inv, *picks = ctx.saved_tensors
        g_perm = _rows(g_perm)
        d = g_perm.shape[1]
        gx = torch.empty(ctx.nrows, d, device=g_perm.device, dtype=torch.float32)
        lib.srec_gather_rows(ptr(g_perm), _ld(g_perm), ptr(inv), ptr(gx), d, ctx.nrows, None, d, stream())
        for idx, g in zip(picks, g_picks):
            if g is None:
                continue
            g = _rows(g)
            ar = _arange(idx.numel() + 1, g.device)
            lib.srec_scatter_add_sorted(ptr(g), _ld(g), ptr(idx), ptr(ar), ptr(ar), ptr(gx), d, idx.numel(),
                                        ptr(ctx.dyn_b), d, 1, stream())
        return (gx, None, None, None, None) + (None,) * len(picks)


def permute_and_pick(x, perm, inv, picks, dyn_n=None, dyn_b=None):
    outs = PermuteAndPick.apply(x, perm, inv, dyn_n, dyn_b, *picks)
    for o in outs[1:]:
        o._srec_cat_left = True          # the left half of a private [n, 2 d] buffer (ReadoutHead writes the right half)
    return outs[0], list(outs[1:])


# ------------------------------------------------------------------------------------------ deferred slab sums
# Several backward nodes end in "out = sum over R partial slabs" kernels whose results only the optimizer reads (weight
# gradients of the row-split GEMMs).  Each is a ~5 us kernel node of the captured step whatever its size.  A node may
# register (slabs, out) here instead of launching, and ONE launch at the end of the backward pass (autograd's
# queue_callback) sums them all: the gradients are complete when backward() returns, as before.
# Deferring hands autograd a gradient tensor that is not written yet, which is only safe when AccumulateGrad takes the
# buffer over as it is: the target parameter has no gradient yet (otherwise `p.grad += out` would read the unwritten
# buffer: a second backward before a step, zero_grad(set_to_none=False)) and no hook that reads it.  So the permission is
# scoped twice: a model switches DEFER['on'] for the duration of ITS training forward (every parameter of it feeds exactly
# one backward node per layer / order), the forward of a node snapshots it (ctx.defer = defer_scope()), and the backward
# asks can_defer(ctx.defer, params) when it gets there - anything else launches the sum on the spot.
DEFER = {'on': False}
_DEFERRED = []


def defer_scope():
    """snapshot for an autograd node's forward: may its backward defer its slab sums at all?"""
    return bool(DEFER['on'])


def drop_stale_deferred():
    """a backward pass that raised leaves its waiting sums behind (the engine drops its callbacks): forget them before the next
    forward - their buffers belong to a graph that is gone, and a non-empty list would keep the next pass from registering its
    own end-of-backward callback"""
    del _DEFERRED[:]


def can_defer(flag, params):
    """backward-time check: every target parameter still without a gradient and without tensor hooks"""
    if not flag:
        return False
    for q in params:
        if not q.is_leaf:                           # a derived weight (folded norm, slice): its gradient is READ by the next node
            return False
        if q.grad is not None or getattr(q, '_backward_hooks', None) or getattr(q, '_post_accumulate_grad_hooks', None):
            return False
    return True


def defer_slab_sum(part, out, ok=True, tall=False):
    """out [n] (any shape, contiguous) = sum over the leading dimension of part [R, n...]: now, or (ok) deferred to the
    end of the running backward pass.  tall: few columns, hundreds of rows (bias partials)"""
    if not ok:
        _launch_slab_sums([(part, out, tall)])
        return
    if not _DEFERRED:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush_deferred)
        except RuntimeError:                        # not inside a backward pass: nothing to wait for
            _launch_slab_sums([(part, out, tall)])
            return
    # (an ALIAS of out: autograd takes a returned gradient as it is only when nothing else refers to the tensor object -
    #  with a second reference AccumulateGrad would clone it, before the sum has been written)
    _DEFERRED.append((part, out.detach(), tall))


# the optimizer's step-scalar launch as a rider of the END-of-backward slab-sum launch of a captured step (optim.FusedAdam.hyper_rider
# puts the arguments of srec_adam_hyper_multi here before the backward pass; flush_deferred takes them along and notes which slots
# it advanced in HYPER_DONE; FusedAdam.launch then skips its own launch)
PENDING_HYPER = []
HYPER_DONE = []


def _launch_slab_sums(tasks, rider=None):
    """out: contiguous, or a column block [r, w] of a row-major matrix (stride(0) = ld > w: a slice of a weight gradient).
    rider: arguments of srec_adam_hyper_multi (without the stream) that leave with the LAST launch"""
    for i in range(0, len(tasks), 32):
        chunk = tasks[i:i + 32]
        m = len(chunk)
        arr = _ct.c_void_p * m
        a_p, a_o = arr(*[t[0].data_ptr() for t in chunk]), arr(*[t[1].data_ptr() for t in chunk])
        a_r, a_n = (_ct.c_int * m)(*[t[0].shape[0] for t in chunk]), (_ct.c_long * m)(*[t[1].numel() for t in chunk])
        a_t = (_ct.c_int * m)(*[int(len(t) > 2 and bool(t[2])) for t in chunk])
        wl = [(0, 0) if t[1].is_contiguous() else (t[1].shape[1], t[1].stride(0)) for t in chunk]
        for t, (w, ld) in zip(chunk, wl):
            assert w == 0 or (t[1].dim() == 2 and t[1].stride(1) == 1), (t[1].shape, t[1].stride())
        a_w, a_l = (_ct.c_int * m)(*[w for w, _ in wl]), (_ct.c_int * m)(*[ld for _, ld in wl])
        if rider is not None and i + 32 >= len(tasks):
            lib.srec_sum_slabs_multi_hyper(m, _ct.addressof(a_p), _ct.addressof(a_r), _ct.addressof(a_n), _ct.addressof(a_o),
                                           _ct.addressof(a_t), _ct.addressof(a_w), _ct.addressof(a_l), *rider, stream())
        else:
            lib.srec_sum_slabs_multi_ld(m, _ct.addressof(a_p), _ct.addressof(a_r), _ct.addressof(a_n), _ct.addressof(a_o),
                                        _ct.addressof(a_t), _ct.addressof(a_w), _ct.addressof(a_l), stream())


def flush_deferred():
    if _DEFERRED:
        tasks = list(_DEFERRED)
        _DEFERRED.clear()
        pend = PENDING_HYPER.pop() if PENDING_HYPER else None
        del PENDING_HYPER[:]
        _launch_slab_sums(tasks, pend[0] if pend is not None else None)
        if pend is not None:
            HYPER_DONE[:] = [pend[1]]


# ------------------------------------------------------------------------------------------ gradient arenas
# Row-sharded training all-reduces the replicated parameters' gradients in a few flat buckets (dist.VocabParallel).  A
# parameter with a slot in such a bucket carries `_srec_gslot = [arena, offset, epoch]`; a backward node that ALLOCATES the
# gradient it returns asks grad_buf(p) for the destination and gets the slot itself - the kernel then writes the gradient where
# the all-reduce reads it, and autograd's AccumulateGrad takes the view over as p.grad (no concatenation pass).  A slot is
# handed out once per training forward (`grad_epoch()`): a parameter that feeds two backward nodes gets a private buffer for
# the second one (the engine sums the two before AccumulateGrad runs), and one that already holds a gradient (accumulation
# over micro-batches) always does - whatever does not end up in its slot is copied there by the bucket code.
_GRAD_EPOCH = [0]


def grad_epoch():
    """a new training forward begins: every arena slot may be handed out once more"""
    _GRAD_EPOCH[0] += 1


def grad_slot(p, numel=None):
    """the arena view for parameter p's gradient (flat, `numel` elements: p's own slot, or a run of adjacent slots starting
    at p's), or None when p has no slot / the slot is taken / p already holds a gradient or carries hooks"""
    slot = getattr(p, '_srec_gslot', None)
    if slot is None or slot[2] == _GRAD_EPOCH[0] or p.grad is not None or not p.is_leaf:
        return None
    if getattr(p, '_backward_hooks', None) or getattr(p, '_post_accumulate_grad_hooks', None):
        return None
    n = p.numel() if numel is None else numel
    if slot[1] + n > slot[0].numel():
        return None
    slot[2] = _GRAD_EPOCH[0]
    return slot[0][slot[1]:slot[1] + n]


def grad_buf(p, like=None):
    """where a backward node writes the gradient of parameter p (shape of `like`, default p): its bucket slot or a new tensor"""
    like = p if like is None else like
    v = grad_slot(p) if like.numel() == p.numel() else None
    if v is not None:
        return v.view(like.shape)
    return torch.empty(like.shape, device=like.device, dtype=torch.float32)


def grad_buf_pair(p, q):
    """ONE flat buffer [p.numel() + q.numel()] for two gradients a kernel writes back to back (the GRU's two bias vectors):
    the two slots when they are neighbours in the arena (nn.GRU lists bias_ih, bias_hh one after the other)"""
    sp, sq = getattr(p, '_srec_gslot', None), getattr(q, '_srec_gslot', None)
    if sp is not None and sq is not None and sp[0] is sq[0] and sq[1] == sp[1] + p.numel() and sq[2] != _GRAD_EPOCH[0] \
            and q.grad is None and q.is_leaf and not getattr(q, '_backward_hooks', None) \
            and not getattr(q, '_post_accumulate_grad_hooks', None):
        v = grad_slot(p, p.numel() + q.numel())
        if v is not None:
            sq[2] = _GRAD_EPOCH[0]
            return v
    return torch.empty(p.numel() + q.numel(), device=p.device, dtype=torch.float32)


class GradMark(torch.autograd.Function):
    """identity; its backward calls fn() when the gradient of x has arrived - i.e. when every node downstream of x, and the
    AccumulateGrad nodes of their parameters (the engine runs those first), are done.  dist.VocabParallel.bucket_ready hangs
    the early launch of a gradient bucket's all-reduce here."""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def grad_mark(x, fn):
    return GradMark.apply(x, fn) if (x.requires_grad and torch.is_grad_enabled()) else x


class NormPermutePick(torch.autograd.Function):
    """(allf, v_0, v_1, ...) = (normalize(x)[perm], normalize(x)[pick_0], ...): MSGIFSR between its last MSHGNN layer and the
    read-out (msgifsr.py:260-264 F.normalize, :131-147 per-session concatenation and last-node picks) in ONE launch, and its
    backward (inverse permutation, pick scatter-add, normalisation backward) in one."""

    @staticmethod
    def forward(ctx, x, perm, cat_seg, dyn_t, dyn_b, types, eps_mode, *picks):
        x = _rows(x)
        NTs, D = x.shape
        n_cap, B, P = perm.numel(), picks[0].numel(), len(picks)
        dev = x.device
        allf = torch.empty(n_cap, D, device=dev, dtype=torch.float32)
        invr = torch.empty(n_cap, device=dev, dtype=torch.float32)
        bufs = [torch.empty(B, 2 * D, device=dev, dtype=torch.float32) for _ in range(P)]   # v = left half of [v | read-out]
        arr, ints = _ct.c_void_p * P, _ct.c_int * P
        a_pick, a_out = arr(*[p.data_ptr() for p in picks]), arr(*[b.data_ptr() for b in bufs])
        a_ld = ints(*([2 * D] * P))                         # (ctypes arrays must outlive the call: no temporaries in addressof)
        lib.srec_norm_perm_pick_fwd(ptr(x), _ld(x), ptr(perm), n_cap, ptr(dyn_t), D, eps_mode, 1e-12, ptr(allf), ptr(invr),
                                    P, B, ptr(dyn_b), _ct.addressof(a_pick), _ct.addressof(a_out), _ct.addressof(a_ld), stream())
        ctx.save_for_backward(perm, cat_seg, allf, invr, *picks)
        ctx.meta = (NTs, D, B, dyn_b, types)
        return (allf,) + tuple(b[:, :D] for b in bufs)

    @staticmethod
    def backward(ctx, g_allf, *g_picks):
        perm, cat_seg, allf, invr, *picks = ctx.saved_tensors
        NTs, D, B, dyn_b, types = ctx.meta
        P, nt = len(picks), len(types)
        g_allf = _rows(g_allf)
        gps = [(_rows(g) if g is not None else None) for g in g_picks]
        dx = torch.empty(NTs, D, device=allf.device, dtype=torch.float32)
        arr, ints = _ct.c_void_p * P, _ct.c_int * P
        a_pick = arr(*[p.data_ptr() for p in picks])
        a_g = arr(*[(g.data_ptr() if g is not None else None) for g in gps])
        a_ld = ints(*[(_ld(g) if g is not None else D) for g in gps])
        tarr = _ct.c_void_p * nt
        a_dyn = tarr(*[ptr(t[2]) for t in types])
        a_r0, a_nc = (_ct.c_int * nt)(*[t[0] for t in types]), (_ct.c_int * nt)(*[t[1] for t in types])
        lib.srec_norm_perm_pick_bwd(ptr(allf), ptr(invr), ptr(g_allf), _ld(g_allf), ptr(perm), perm.numel(), ptr(cat_seg), B,
                                    ptr(dyn_b), D, P,
                                    _ct.addressof(a_pick), _ct.addressof(a_g), _ct.addressof(a_ld), ptr(dx), D, nt,
                                    _ct.addressof(a_r0), _ct.addressof(a_nc), _ct.addressof(a_dyn), NTs, stream())
        return (dx,) + (None,) * (6 + P)


def norm_permute_pick(x, perm, cat_seg, picks, types, dyn_t=None, dyn_b=None, eps_mode=0):
    """types: [(first stacked row, capacity, dyn live count)] of the node types stacked in x.  -> (allf, [v_k]); every v_k is
    the tagged left half of a private [B, 2 D] buffer whose right half the read-out head fills"""
    outs = NormPermutePick.apply(x, perm, cat_seg, dyn_t, dyn_b, tuple(types), eps_mode, *picks)
    for o in outs[1:]:
        o._srec_cat_left = True
    return outs[0], list(outs[1:])


class _GradArena:
    """gradient buffer of a row-split tensor, shared through the pieces: a consumer whose backward produces the gradient of a
    piece (normalize_stack, gru_expand_all) writes it straight into that piece's rows of ONE buffer, and SplitRows.backward
    hands the buffer on instead of concatenating (the concatenation was the last aten kernel of the captured step)."""

    def __init__(self, shape, device):
        self.shape, self.device, self.buf = tuple(shape), device, None

    def rows(self, off, n):
        if self.buf is None:
            self.buf = torch.empty(self.shape, device=self.device, dtype=torch.float32)
        return self.buf[off:off + n]


def _arena_tag(x):
    """(arena, first row) of a tensor that SplitRows handed out, else None"""
    return getattr(x, '_srec_arena', None)


def _arena_rows(tag, n, d, device):
    """gradient rows for a piece: inside its arena when it has one"""
    if tag is not None and len(tag[0].shape) == 2 and tag[0].shape[1] == d:
        return tag[0].rows(tag[1], n)
    return torch.empty(n, d, device=device, dtype=torch.float32)


class SplitRows(torch.autograd.Function):
    """row-range views x[o_i : o_i + n_i]; the backward is ONE buffer (filled in place by the consumers, see _GradArena)
    or one concatenation, instead of a zero-fill, a slice copy and an add per piece (autograd's SliceBackward)."""

    @staticmethod
    def forward(ctx, x, sizes):
        ctx.sizes, ctx.shape = sizes, x.shape
        ctx.arena = _GradArena(x.shape, x.device)
        outs, off = [], 0
        for n in sizes:
            outs.append(x[off:off + n])
            off += n
        assert off == x.shape[0]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        buf = ctx.arena.buf
        if buf is not None and all(g is not None for g in gs):
            off, esz, ok = 0, buf.element_size() * (buf.stride(0) if buf.dim() > 1 else 1), True
            for g, n in zip(gs, ctx.sizes):
                ok = ok and g.shape[0] == n and g.is_contiguous() and g.data_ptr() == buf.data_ptr() + off * esz
                off += n
            if ok:
                return buf, None
        dev = next(t for t in gs if t is not None).device
        gs = [g if g is not None else torch.zeros((n,) + tuple(ctx.shape[1:]), device=dev) for g, n in zip(gs, ctx.sizes)]
        return torch.cat(gs, 0), None


def split_rows(x, sizes):
    outs = SplitRows.apply(x, tuple(sizes))
    fn = outs[0].grad_fn if len(outs) else None
    arena = getattr(fn, 'arena', None)               # the node's ctx attributes are visible on grad_fn
    if arena is not None:
        off = 0
        for o, n in zip(outs, sizes):
            o._srec_arena = (arena, off)
            off += n
    return outs


class Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps_mode, dyn, ws=None):
        x = _rows(x)
        n, d = x.shape
        y = torch.empty(n, d, device=x.device, dtype=torch.float32)
        inv = torch.empty(n, device=x.device, dtype=torch.float32)
        sr16 = getattr(ws, 'sr16', None)
        if sr16 is not None and n <= sr16.shape[0] and d <= sr16.shape[1]:
            # ws: the scoring workspace this session vector is headed for - its bf16 operand copy is written here
            lib.srec_normalize_fwd_bf16(ptr(x), _ld(x), ptr(y), d, ptr(inv), n, ptr(dyn), d, eps_mode, 1e-12, ptr(sr16),
                                        sr16.shape[1], stream())
            ws.sr_fresh = (y.data_ptr(), n, d)
        else:
            lib.srec_normalize_fwd(ptr(x), _ld(x), ptr(y), d, ptr(inv), n, ptr(dyn), d, eps_mode, 1e-12, stream())
        ctx.save_for_backward(y, inv)
        ctx.dyn = dyn
        return y

    @staticmethod
    def backward(ctx, gy):
        y, inv = ctx.saved_tensors
        gy = _rows(gy)
        n, d = y.shape
        gx = torch.empty_like(y)
        lib.srec_normalize_bwd(ptr(y), d, ptr(gy), _ld(gy), ptr(inv), ptr(gx), d, n, ptr(ctx.dyn), d, stream())
        return gx, None, None, None


def normalize(x, eps_mode=0, dyn=None, ws=None):
    """ws (a CEWorkspace in bf16 scoring mode): also write the bf16 operand copy of the rows into ws.sr16"""
    return Normalize.apply(x, eps_mode, dyn, ws)


class NormalizeStack(torch.autograd.Function):
    """stacked [sum n_p, d] = concatenation of the row-normalised x_p (np <= 4 tensors, each with its own live count): one
    launch forward, one backward, no concatenation kernel (the per-order features MSGIFSR feeds its batched MSHGNN layer)"""

    @staticmethod
    def forward(ctx, eps_mode, dyns, *xs):
        ctx.tags = [_arena_tag(x) for x in xs]              # pieces of a split tensor: their gradients go into its buffer
        xs = [_rows(x) for x in xs]
        P, d, dev = len(xs), xs[0].shape[1], xs[0].device
        ns = [x.shape[0] for x in xs]
        y = torch.empty(sum(ns), d, device=dev, dtype=torch.float32)
        inv = torch.empty(sum(ns), device=dev, dtype=torch.float32)
        arr = _ct.c_void_p * P
        a_x, a_d = arr(*[x.data_ptr() for x in xs]), arr(*[ptr(t) for t in dyns])
        a_l, a_n = (_ct.c_int * P)(*[_ld(x) for x in xs]), (_ct.c_int * P)(*ns)
        lib.srec_normalize_group_fwd(P, _ct.addressof(a_x), _ct.addressof(a_l), _ct.addressof(a_n), _ct.addressof(a_d), ptr(y),
                                     d, ptr(inv), d, eps_mode, 1e-12, stream())
        ctx.save_for_backward(y, inv)
        ctx.meta = (ns, dyns)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, inv = ctx.saved_tensors
        ns, dyns = ctx.meta
        gy = _rows(gy)
        P, d = len(ns), y.shape[1]
        dxall = torch.empty_like(y)
        dxs, o = [], 0
        for n, tag in zip(ns, ctx.tags):
            dxs.append(_arena_rows(tag, n, d, y.device) if tag is not None else dxall[o:o + n])
            o += n
        arr = _ct.c_void_p * P
        a_x, a_d = arr(*[t.data_ptr() for t in dxs]), arr(*[ptr(t) for t in dyns])
        a_l, a_n = (_ct.c_int * P)(*([d] * P)), (_ct.c_int * P)(*ns)
        lib.srec_normalize_group_bwd(P, _ct.addressof(a_x), _ct.addressof(a_l), _ct.addressof(a_n), _ct.addressof(a_d), ptr(y), d,
                                     ptr(gy), _ld(gy), ptr(inv), d, stream())
        return (None, None) + tuple(dxs)


def normalize_stack(xs, eps_mode=0, dyns=None):
    dyns = tuple(dyns) if dyns is not None else (None,) * len(xs)
    return NormalizeStack.apply(eps_mode, dyns, *xs)


class SegAttn(torch.autograd.Function):
    """alpha = softmax_session(fc_e(sigmoid(U + Vq[b])));  out_b = sum_i alpha_i x_i"""

    @staticmethod
    def forward(ctx, U, Vq, we, X, seg, dynB):
        U, Vq, X = _rows(U), _rows(Vq), _rows(X)
        ctx.defer, ctx.wparams = defer_scope(), [we]
        we = we.reshape(-1).contiguous()
        B, h = Vq.shape
        N, D = X.shape
        alpha = torch.empty(N, device=X.device, dtype=torch.float32)      # live nodes written, padded never read
        out = torch.empty(B, D, device=X.device, dtype=torch.float32)
        lib.srec_seg_attn_fwd(ptr(U), _ld(U), ptr(Vq), _ld(Vq), ptr(we), ptr(X), _ld(X), ptr(seg), B, ptr(dynB), h, D,
                              ptr(alpha), ptr(out), D, stream())
        ctx.save_for_backward(U, Vq, we, X, alpha, seg)
        ctx.dynB = dynB
        return out

    @staticmethod
    def backward(ctx, gout):
        U, Vq, we, X, alpha, seg = ctx.saved_tensors
        gout = _rows(gout)
        B, h = Vq.shape
        N, D = X.shape
        dX = torch.empty(N, D, device=X.device, dtype=torch.float32)      # rows behind the live nodes zeroed in-kernel
        dU = torch.empty(N, h, device=X.device, dtype=torch.float32)
        dVq = torch.empty(B, h, device=X.device, dtype=torch.float32)
        dwp = torch.empty(B, h, device=X.device, dtype=torch.float32)
        lib.srec_seg_attn_bwd(ptr(gout), _ld(gout), ptr(X), _ld(X), ptr(alpha), ptr(U), _ld(U), ptr(Vq), _ld(Vq),
                              ptr(we), ptr(seg), B, ptr(ctx.dynB), h, D, N, ptr(dX), D, ptr(dU), h, ptr(dVq), h, ptr(dwp),
                              h, stream())
        dwe = torch.empty(h, device=X.device, dtype=torch.float32)
        if h % 4 == 0 and can_defer(ctx.defer, ctx.wparams):
            # d fc_e = the column sums of the per-session rows (dead sessions' rows are zeros): a 'tall' task of the ONE
            # end-of-backward slab-sum launch instead of two launches here
            defer_slab_sum(dwp, dwe, True, tall=True)
        else:
            col_sum(dwp, B, h, dwe, ctx.dynB)
        return dU, dVq, dwe.view(1, h), dX, None, None


def seg_attn(U, Vq, we, X, seg, dynB=None):
    return SegAttn.apply(U, Vq, we, X, seg, dynB)


class GemmF32Group(_ct.Structure):
    """host mirror of srec_gemm_f32_group (include/srec_hg.h)"""
    _fields_ = [('np', _ct.c_int),
                ('A', _ct.c_void_p * 16), ('a_rs', _ct.c_int * 16), ('a_cs', _ct.c_int * 16),
                ('B', _ct.c_void_p * 16), ('b_rs', _ct.c_int * 16), ('b_cs', _ct.c_int * 16),
                ('C', _ct.c_void_p * 16), ('ldc', _ct.c_int * 16), ('bias', _ct.c_void_p * 16),
                ('M', _ct.c_int * 16), ('N', _ct.c_int * 16), ('K', _ct.c_int * 16),
                ('dyn', _ct.c_void_p * 16), ('dyn_mode', _ct.c_int * 16),
                ('alpha', _ct.c_float * 16), ('beta', _ct.c_float * 16), ('nsplit', _ct.c_int * 16), ('ws', _ct.c_void_p),
                ('split3', _ct.c_int)]


def gemm_f32_group(probs, split3=False, defer=None):
    """defer: a list of output tensors - the split-K slab sums of the group's weight-gradient ('tn', whole contiguous output
    tensor) problems that write one of them join the end-of-backward slab-sum launch (defer_slab_sum) instead of a reduce
    launch behind this one; the caller has checked can_defer() for the parameters those outputs become the gradients of.
    ONE launch of up to 16 independent exact-fp32 products (csrc/gemm.hip, srec_gemm_f32_group_run; split3: as 3-term
    hi / lo bf16 splits on the bf16 matrix pipe, ~2^-17 relative).  A problem is
    (kind, a, b, out, bias, dyn, beta):  'nt' out[M,N] = a[M,K] b[N,K]^T + bias (dyn clamps M);  'nn' out[M,K] = a[M,N] b[N,K]
    (dyn clamps M);  'tn' out[N,K] = a[M,N]^T b[M,K] (dyn clamps the reduction rows M).  (+ beta * out)"""
    assert 0 < len(probs) <= 16
    g = GemmF32Group()
    g.np, g.split3 = len(probs), int(split3)
    for p, (kind, a, b, out, bias, dyn, beta) in enumerate(probs):
        if kind == 'nt':
            M, K = a.shape
            N = b.shape[0]
            g.a_rs[p], g.a_cs[p], g.b_rs[p], g.b_cs[p], mode = _ld(a), 1, _ld(b), 1, 1
        elif kind == 'nn':
            M, K = a.shape
            N = b.shape[1]
            g.a_rs[p], g.a_cs[p], g.b_rs[p], g.b_cs[p], mode = _ld(a), 1, 1, _ld(b), 1
        else:
            K, M = a.shape
            N = b.shape[1]
            g.a_rs[p], g.a_cs[p], g.b_rs[p], g.b_cs[p], mode = 1, _ld(a), 1, _ld(b), 2
        g.A[p], g.B[p], g.C[p], g.ldc[p], g.bias[p] = ptr(a), ptr(b), ptr(out), _ld(out), ptr(bias)
        g.M[p], g.N[p], g.K[p] = M, N, K
        g.dyn[p], g.dyn_mode[p] = ptr(dyn), (mode if dyn is not None else 0)
        g.alpha[p], g.beta[p] = 1.0, beta
    dev = probs[0][1].device
    if defer:
        ok = [kind == 'tn' and out.dim() == 2 and out.stride(1) == 1 and beta == 0.0 and any(out is t for t in defer)
              for (kind, a, b, out, bias, dyn, beta) in probs]
        # (the launcher only splits problems of < 128 output tiles, into <= 32 slabs)
        need = sum(32 * int(g.M[p]) * int(g.N[p]) for p in range(len(probs))
                   if ((int(g.M[p]) + 63) // 64) * ((int(g.N[p]) + 63) // 64) < 128)
        if any(ok) and need <= (1 << 24):
            # a PRIVATE slab buffer: it has to survive until the end of the backward pass (the shared one is rewritten by the
            # next group); problems that may not wait (cnt[p] = 0 on entry: accumulating / clamped / strided outputs, outputs
            # somebody reads during the backward pass) keep their reduce launch
            ws = torch.empty(need, device=dev, dtype=torch.float32)
            off, cnt = (_ct.c_long * 16)(), (_ct.c_int * 16)(*[int(o) for o in ok])
            lib.srec_gemm_f32_group_run_defer(_ct.addressof(g), ptr(ws), need, _ct.addressof(off), _ct.addressof(cnt), stream())
            for p, (kind, a, b, out, bias, dyn, beta) in enumerate(probs):
                if off[p] >= 0 and cnt[p] > 1:
                    n = out.numel()
                    defer_slab_sum(ws[off[p]:off[p] + cnt[p] * n].view(cnt[p], n), out)
            if _GROUP_DEBUG:
                print('gemm_f32_group', [(kind, int(g.M[p]), int(g.N[p]), int(g.K[p]), bool(ok[p]), off[p], cnt[p])
                                         for p, (kind, *_r) in enumerate(probs)], flush=True)
            return
    lib.srec_gemm_f32_group_run(_ct.addressof(g), *_gemm_ws(dev), stream())


_ONES4 = {}
_GROUP_DEBUG = bool(os.environ.get('SREC_GROUP_DEBUG'))


def _ones4(n, device):
    t = _ONES4.get(str(device))
    if t is None or t.shape[0] < n:
        t = _ONES4[str(device)] = torch.ones(n, 4, device=device, dtype=torch.float32)
    return t


class ReadoutHead(torch.autograd.Function):
    """Attention read-out + session-vector projection of MSGIFSR (msgifsr.py:127-146 AttnReadout, :272-279 fc_sr) for
    every live order as grouped launches - exact fp32 in fp32 mode, 3-term hi / lo bf16 splits (fp32-grade, ~2^-17) in
    bf16 mode, like the fused head of d = 128 / 256:
        U_i = allf Wu_i^T + bu_i;  Vq_i = v_i Wv_i^T;  alpha_i = softmax_session(we_i . sigmoid(U_i + Vq_i[b]));
        g_i = sum alpha_i allf;  s_i = [v_i | g_i] Wsr_i^T
    forward: 1 grouped GEMM (all U, Vq) + per order (read-out kernel, concat) + 1 grouped GEMM (all s);
    backward: 1 grouped GEMM (d cat, d Wsr) + per order read-out backward + 1 grouped GEMM (+ 1 slab reduce) for
    d allf += dU Wu, d Wu, d v += dVq Wv, d Wv + the two bias / fc_e column sums.  The same products launched one by one
    were 9 GEMM + 7 split-K reduce + 2 accumulate nodes of the captured step."""

    @staticmethod
    def forward(ctx, allf, seg, dT, dB, *flat):
        n = len(flat) // 6
        allf = _rows(allf)
        NT, D = allf.shape
        dev = allf.device
        per = []
        for i in range(n):
            v, Wu, bu, Wv, we, Wsr = flat[6 * i:6 * i + 6]
            per.append((_rows(v), _rows(Wu), bu, _rows(Wv), we.reshape(-1).contiguous(), _rows(Wsr)))
        B = per[0][0].shape[0]
        h = per[0][1].shape[0]
        Us = [torch.empty(NT, h, device=dev, dtype=torch.float32) for _ in range(n)]
        Vqs = [torch.empty(B, h, device=dev, dtype=torch.float32) for _ in range(n)]
        probs = []
        for i, (v, Wu, bu, Wv, we, Wsr) in enumerate(per):
            probs.append(('nt', allf, Wu, Us[i], bu, dT, 0.0))
            probs.append(('nt', v, Wv, Vqs[i], None, dB, 0.0))
        gemm_f32_group(probs, split3=PRECISION['matmul'] == 'bf16')
        alphas, cats, outs, probs = [], [], [], []
        for i, (v, Wu, bu, Wv, we, Wsr) in enumerate(per):
            alpha = torch.empty(NT, device=dev, dtype=torch.float32)
            dv = v.shape[1]
            cat = None
            # only a tensor its producer tagged as the left half of a PRIVATE [B, dv + D] buffer (permute_and_pick /
            # norm_permute_pick): any other column slice of a live [B, 2 d] tensor must not have its right half overwritten
            if B > 1 and getattr(flat[6 * i], '_srec_cat_left', False) and v.stride(0) == dv + D and v.stride(1) == 1:
                try:                                       # v already is the left half of a [B, dv + D] buffer (permute_and_pick)
                    cat = v.as_strided((B, dv + D), (dv + D, 1))
                except RuntimeError:
                    cat = None
            if cat is not None:
                lib.srec_seg_attn_fwd(ptr(Us[i]), h, ptr(Vqs[i]), h, ptr(we), ptr(allf), _ld(allf), ptr(seg), B, ptr(dB), h, D,
                                      ptr(alpha), cat[:, dv:].data_ptr(), dv + D, stream())
            else:
                srg = torch.empty(B, D, device=dev, dtype=torch.float32)
                lib.srec_seg_attn_fwd(ptr(Us[i]), h, ptr(Vqs[i]), h, ptr(we), ptr(allf), _ld(allf), ptr(seg), B, ptr(dB), h, D,
                                      ptr(alpha), ptr(srg), D, stream())
                cat = torch.empty(B, dv + D, device=dev, dtype=torch.float32)
                lib.srec_cat_cols(ptr(v), _ld(v), dv, ptr(srg), D, D, B, ptr(cat), stream())
            out = torch.empty(B, Wsr.shape[0], device=dev, dtype=torch.float32)
            probs.append(('nt', cat, Wsr, out, None, dB, 0.0))
            alphas.append(alpha)
            cats.append(cat)
            outs.append(out)
        gemm_f32_group(probs, split3=PRECISION['matmul'] == 'bf16')
        ctx.save_for_backward(allf, seg, *[t for i in range(n) for t in (per[i][0], per[i][1], per[i][3], per[i][4], per[i][5],
                                                                        Us[i], Vqs[i], alphas[i], cats[i])])
        ctx.n, ctx.dT, ctx.dB = n, dT, dB
        ctx.has_bu = [per[i][2] is not None for i in range(n)]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        allf, seg, *rest = ctx.saved_tensors
        return _readout_head_backward(allf, seg, [rest[9 * i:9 * i + 9] for i in range(ctx.n)], ctx.dT, ctx.dB, ctx.has_bu, gs)


def _readout_head_backward(allf, seg, per, dT, dB, has_bu, gs):
    """the grouped backward of the read-out head (ReadoutHead, ReadoutHeadFused): per[i] = (v, Wu, Wv, we, Wsr, U, Vq, alpha,
    cat) of live order i, gs[i] = gradient of s_i -> (d allf, None, None, None, per order: d v, d Wu, d bu, d Wv, d we, d Wsr)"""
    n = len(per)
    NT, D = allf.shape
    dev = allf.device
    B = per[0][0].shape[0]
    h = per[0][1].shape[0]
    gcats, gWsrs, probs = [], [], []
    for i, (v, Wu, Wv, we, Wsr, U, Vq, alpha, cat) in enumerate(per):
        g = _rows(gs[i])
        gcat = torch.empty_like(cat)
        gWsr = torch.empty_like(Wsr)
        probs.append(('nn', g, Wsr, gcat, None, dB, 0.0))
        probs.append(('tn', g, cat, gWsr, None, dB, 0.0))
        gcats.append(gcat)
        gWsrs.append(gWsr)
    gemm_f32_group(probs, split3=PRECISION['matmul'] == 'bf16')
    g_allf = None
    probs, grads, extra = [], [], []
    for i, (v, Wu, Wv, we, Wsr, U, Vq, alpha, cat) in enumerate(per):
        dv = v.shape[1]
        gsrg = gcats[i][:, dv:]
        dX = torch.empty(NT, D, device=dev, dtype=torch.float32)      # rows behind the live nodes zeroed in-kernel
        dU = torch.empty(NT, h, device=dev, dtype=torch.float32)
        dVq = torch.empty(B, h, device=dev, dtype=torch.float32)
        dwp = torch.empty(B, h, device=dev, dtype=torch.float32)
        lib.srec_seg_attn_bwd(ptr(gsrg), _ld(gsrg), ptr(allf), _ld(allf), ptr(alpha), ptr(U), h, ptr(Vq), h, ptr(we),
                              ptr(seg), B, ptr(dB), h, D, NT, ptr(dX), D, ptr(dU), h, ptr(dVq), h, ptr(dwp), h, stream())
        gWu, gWv = torch.empty_like(Wu), torch.empty_like(Wv)
        gv = gcats[i][:, :dv]                                         # d v: the concat half, + dVq Wv in place
        probs.append(('nn', dU, Wu, dX, None, dT, 1.0))               # d allf (this order) = read-out term + dU Wu
        probs.append(('tn', dU, allf, gWu, None, dT, 0.0))
        probs.append(('nn', dVq, Wv, gv, None, dB, 1.0))
        probs.append(('tn', dVq, v, gWv, None, dB, 0.0))
        # column sums (d bu = sum_n dU, d we = sum_b dwp) as products with a block of ones, inside the same launch:
        # as col_sum calls they were two kernel nodes each
        ones = _ones4(max(NT, B), dev)
        sums = torch.empty(8, h, device=dev, dtype=torch.float32)
        gbu = None
        if has_bu[i]:
            probs.append(('tn', ones[:NT], dU, sums[:4], None, dT, 0.0))
            gbu = sums[0]
        probs.append(('tn', ones[:B], dwp, sums[4:], None, dB, 0.0))
        grads.append((gv, gWu, gbu, gWv, sums[4:5], gWsrs[i]))
        extra.append(dX)
    per_launch = 12 if len(probs) > 16 else 16          # whole orders per launch (6 problems each)
    for c in range(0, len(probs), per_launch):
        gemm_f32_group(probs[c:c + per_launch], split3=PRECISION['matmul'] == 'bf16')
    g_allf = extra[0]
    for dX in extra[1:]:
        g_allf = g_allf + dX
    return (g_allf, None, None, None) + tuple(t for gr in grads for t in gr)


# ---- fused read-out head (csrc/headf.hip) ----------------------------------------------------------------------------------
_HEAD_WF_CACHE = {}    # (data_ptr, shape, trans) -> hi / lo fragment-major bf16 copy of this step (dropped by weights_changed)


def _head_wfrag_args(ws, trans):
    """output buffers + the HOST argument arrays of srec_head_wfrag for <= 16 weights (kept alive by the caller across the call)"""
    m = len(ws)
    assert 0 < m <= 16
    for w in ws:
        assert w.is_contiguous() and w.dtype == torch.float32
    bufs = [torch.empty(2 * w.numel(), device=w.device, dtype=torch.bfloat16) for w in ws]
    arr, ints = _ct.c_void_p * m, _ct.c_int * m
    return bufs, (arr(*[w.data_ptr() for w in ws]), arr(*[b.data_ptr() for b in bufs]), ints(*[w.shape[0] for w in ws]),
                  ints(*[w.shape[1] for w in ws]), ints(*[int(t) for t in trans]))


def head_wfrag(ws, trans):
    """hi / lo fragment-major bf16 operand copies of fp32 weights (W_i or W_i^T), computed once per optimizer step, all
    missing ones in ONE launch (srec_head_wfrag) - normally none: the step's prologue launch made them (step_prologue)"""
    out, todo = [None] * len(ws), []
    for i, (w, t) in enumerate(zip(ws, trans)):
        c = _HEAD_WF_CACHE.get((w.data_ptr(), tuple(w.shape), int(t)))
        if c is None:
            todo.append(i)
        else:
            out[i] = c
    for c0 in range(0, len(todo), 16):
        ch = todo[c0:c0 + 16]
        bufs, args = _head_wfrag_args([ws[i] for i in ch], [trans[i] for i in ch])
        lib.srec_head_wfrag(len(ch), *[_ct.addressof(a) for a in args], stream())
        for i, b in zip(ch, bufs):
            out[i] = _HEAD_WF_CACHE[(ws[i].data_ptr(), tuple(ws[i].shape), int(trans[i]))] = b
    return out


class HeadDesc(_ct.Structure):
    """host mirror of srec_head_desc (include/srec_hg.h)"""
    _fields_ = ([(nm, _ct.c_int) for nm in ('nh', 'd', 'B', 'NT', 'ld_x', 'ld16', 'eps_mode')] + [('eps', _ct.c_float)] +
                [(nm, _ct.c_void_p) for nm in ('X', 'seg', 'dynB')] +
                [(nm, _ct.c_void_p * 4) for nm in ('cat', 'Wu_f', 'Wv_f', 'Wsr_f', 'bu', 'we', 'alpha', 'U', 'Vq', 'y', 'inv', 'y16')])


class HeadBwdDesc(_ct.Structure):
    """host mirror of srec_head_bwd_desc (include/srec_hg.h)"""
    _fields_ = ([(nm, _ct.c_int) for nm in ('nh', 'd', 'B', 'NT', 'ld_x', 'ld_gy')] + [(nm, _ct.c_void_p) for nm in ('X', 'seg', 'dynB')] +
                [(nm, _ct.c_void_p * 4) for nm in ('gy', 'y', 'inv', 'WsrT_f', 'alpha', 'U', 'Vq', 'we', 'gs', 'gcat', 'dX', 'dU',
                                                   'dVq', 'dwp')])


class ReadoutHeadFused(torch.autograd.Function):
    """ReadoutHead + the normalisation of its outputs (msgifsr.py:124-155, :269-273) with the FORWARD as one launch: a
    workgroup owns 8 sessions and runs Vq, U, the soft-max read-out, fc_sr and F.normalize on them, every product as a
    3-term hi / lo bf16 split on the matrix pipe (csrc/headf.hip; results to ~2^-16 of the exact-fp32 grouped head).
    Returns the NORMALISED session vectors; `ws` (CEWorkspace, optional) receives the bf16 operand copy of head 0.
    Backward: the normalisation backward + the grouped launches of ReadoutHead."""

    @staticmethod
    def forward(ctx, allf, seg, dT, dB, ws, eps_mode, *flat):
        n = len(flat) // 6
        allf = _rows(allf)
        NT, D = allf.shape
        dev = allf.device
        per = [(flat[6 * i], _rows(flat[6 * i + 1]), flat[6 * i + 2], _rows(flat[6 * i + 3]),
                flat[6 * i + 4].reshape(-1).contiguous(), _rows(flat[6 * i + 5])) for i in range(n)]
        B = per[0][0].shape[0]
        # fragment-major hi / lo copies of fc_u, fc_v, fc_sr - and of fc_sr^T for the backward - in ONE launch per step
        need_bwd = any(ctx.needs_input_grad)
        wf = head_wfrag([w for po in per for w in (po[1], po[3], po[5])] + ([po[5] for po in per] if need_bwd else []),
                        [0] * (3 * n) + ([1] * n if need_bwd else []))
        ctx.wft = wf[3 * n:] if need_bwd else None
        q = HeadDesc()
        q.nh, q.d, q.B, q.NT, q.ld_x, q.eps_mode, q.eps = n, D, B, NT, _ld(allf), int(eps_mode), 1e-12
        q.X, q.seg, q.dynB = ptr(allf), ptr(seg), ptr(dB)
        sr16 = getattr(ws, 'sr16', None) if n == 1 else None
        if sr16 is not None and not (B <= sr16.shape[0] and D <= sr16.shape[1]):
            sr16 = None
        q.ld16 = sr16.shape[1] if sr16 is not None else 0
        keep, ys = [], []
        for i, (v, Wu, bu, Wv, we, Wsr) in enumerate(per):
            cat = v.as_strided((B, 2 * D), (2 * D, 1))             # v is the left half of a private [B, 2 D] buffer (checked)
            alpha = torch.empty(NT, device=dev, dtype=torch.float32)
            U = torch.empty(NT, D, device=dev, dtype=torch.float32)
            Vq = torch.empty(B, D, device=dev, dtype=torch.float32)
            y = torch.empty(B, D, device=dev, dtype=torch.float32)
            inv = torch.empty(B, device=dev, dtype=torch.float32)
            q.cat[i], q.Wu_f[i], q.Wv_f[i], q.Wsr_f[i] = ptr(cat), ptr(wf[3 * i]), ptr(wf[3 * i + 1]), ptr(wf[3 * i + 2])
            q.bu[i], q.we[i], q.alpha[i], q.U[i], q.Vq[i] = ptr(bu), ptr(we), ptr(alpha), ptr(U), ptr(Vq)
            q.y[i], q.inv[i] = ptr(y), ptr(inv)
            q.y16[i] = ptr(sr16) if (sr16 is not None and i == 0) else None
            keep += [v, Wu, Wv, we, Wsr, U, Vq, alpha, cat, y, inv]
            ys.append(y)
        lib.srec_head_fwd(_ct.addressof(q), stream())
        if sr16 is not None:
            ws.sr_fresh = (ys[0].data_ptr(), B, D)
        ctx.save_for_backward(allf, seg, *keep)
        ctx.n, ctx.dT, ctx.dB = n, dT, dB
        ctx.has_bu = [po[2] is not None for po in per]
        ctx.defer = defer_scope()
        ctx.wparams = [flat[6 * i + j] for i in range(n) for j in (1, 2, 3, 4, 5) if flat[6 * i + j] is not None]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        """one launch for everything per session (normalise-backward, d cat = g_s Wsr, the attention read-out backward:
        srec_head_bwd), then the batch-wide products as ONE grouped exact-fp32 launch + its split-K sum:
        d allf += dU Wu, d v += dVq Wv, d Wu, d Wv, d Wsr and the two column sums (d bu, d we)"""
        allf, seg, *rest = ctx.saved_tensors
        n, dT, dB = ctx.n, ctx.dT, ctx.dB
        NT, D = allf.shape
        dev = allf.device
        per = [rest[11 * i:11 * i + 11] for i in range(n)]
        B = per[0][0].shape[0]
        wft = ctx.wft if ctx.wft is not None else head_wfrag([po[4] for po in per], [1] * n)   # fc_sr^T [2 D, D], fragment-major
        q = HeadBwdDesc()
        gy0 = _rows(gys[0])
        q.nh, q.d, q.B, q.NT, q.ld_x, q.ld_gy = n, D, B, NT, _ld(allf), _ld(gy0)
        q.X, q.seg, q.dynB = ptr(allf), ptr(seg), ptr(dB)
        outs = []
        for i, (v, Wu, Wv, we, Wsr, U, Vq, alpha, cat, y, inv) in enumerate(per):
            gy = _rows(gys[i])
            if _ld(gy) != q.ld_gy:
                gy = gy.contiguous() if q.ld_gy == D else gy0.new_empty(B, q.ld_gy)[:, :D].copy_(gy)
            gs, gcat = torch.empty(B, D, device=dev, dtype=torch.float32), torch.empty(B, 2 * D, device=dev, dtype=torch.float32)
            dX, dU = torch.empty(NT, D, device=dev, dtype=torch.float32), torch.empty(NT, D, device=dev, dtype=torch.float32)
            dVq, dwp = torch.empty(B, D, device=dev, dtype=torch.float32), torch.empty(B, D, device=dev, dtype=torch.float32)
            q.gy[i], q.y[i], q.inv[i], q.WsrT_f[i], q.alpha[i] = ptr(gy), ptr(y), ptr(inv), ptr(wft[i]), ptr(alpha)
            q.U[i], q.Vq[i], q.we[i] = ptr(U), ptr(Vq), ptr(we)
            q.gs[i], q.gcat[i], q.dX[i], q.dU[i], q.dVq[i], q.dwp[i] = ptr(gs), ptr(gcat), ptr(dX), ptr(dU), ptr(dVq), ptr(dwp)
            outs.append((gy, gs, gcat, dX, dU, dVq, dwp))
        lib.srec_head_bwd(_ct.addressof(q), stream())
        probs, grads, blocks = [], [], []
        for i, (v, Wu, Wv, we, Wsr, U, Vq, alpha, cat, y, inv) in enumerate(per):
            gy, gs, gcat, dX, dU, dVq, dwp = outs[i]
            gWu, gWv, gWsr = grad_buf(Wu), grad_buf(Wv), grad_buf(Wsr)       # (bucket slots when the table is row-sharded)
            gv = gcat[:, :D]                                              # d v: the concat half, + dVq Wv in place
            probs.append(('nn', dU, Wu, dX, None, dT, 1.0))               # d allf (this order) = read-out term + dU Wu
            probs.append(('tn', dU, allf, gWu, None, dT, 0.0))
            probs.append(('nn', dVq, Wv, gv, None, dB, 1.0))
            probs.append(('tn', dVq, v, gWv, None, dB, 0.0))
            probs.append(('tn', gs, cat, gWsr, None, dB, 0.0))
            # column sums as products with a block of ones inside the same launch: d bu = sum_n dU = sum_b dVq, d we = sum_b dwp
            ones = _ones4(B, dev)
            sums = torch.empty(8, D, device=dev, dtype=torch.float32)
            s_bu, s_we = sums[:4], sums[4:]
            gbu = None
            if ctx.has_bu[i]:
                probs.append(('tn', ones[:B], dVq, s_bu, None, dB, 0.0))
                gbu = sums[0]
                blocks.append(s_bu)
            probs.append(('tn', ones[:B], dwp, s_we, None, dB, 0.0))
            blocks.append(s_we)
            grads.append((gv, gWu, gbu, gWv, sums[4:5], gWsr))
        per_launch = 14 if len(probs) > 16 else 16          # whole orders per launch (7 problems each)
        later = ([t for gr in grads for t in (gr[1], gr[3], gr[5])] + blocks) if can_defer(ctx.defer, ctx.wparams) else None
        for c in range(0, len(probs), per_launch):
            gemm_f32_group(probs[c:c + per_launch], split3=True,       # (the forward's products are 3-term splits too)
                           defer=later)
        g_allf = outs[0][3]
        for o in outs[1:]:
            g_allf = g_allf + o[3]
        return (g_allf, None, None, None, None, None) + tuple(t for gr in grads for t in gr)


HEAD_FUSED_MAX_B = 8192      # session capacity of the fused read-out head kernels (csrc/headf.hip: srec_head_fwd / srec_head_bwd)


def readout_head_fused_ok(allf, per_order):
    """the fused forward applies: bf16 mode, d = hidden = output in (128, 256), <= 4 heads, contiguous weights, every query
    tensor the tagged left half of a private [B, 2 d] buffer (norm_permute_pick / permute_and_pick)"""
    if not (allf.is_cuda and PRECISION['matmul'] == 'bf16' and 1 <= len(per_order) <= 4 and FUSED_HEAD):
        return False
    D = allf.shape[1]
    if D not in (128, 256) or allf.stride(1) != 1 or allf.stride(0) % 4:
        return False
    B = per_order[0][0].shape[0]
    if B > HEAD_FUSED_MAX_B:              # srec_head_fwd / _bwd keep a copy of seg[] in LDS: larger batches take the grouped launches
        return False
    for v, Wu, bu, Wv, we, Wsr in per_order:
        if not (getattr(v, '_srec_cat_left', False) and tuple(v.shape) == (B, D) and v.stride(0) == 2 * D and v.stride(1) == 1):
            return False
        if tuple(Wu.shape) != (D, D) or tuple(Wv.shape) != (D, D) or tuple(Wsr.shape) != (D, 2 * D) or we.numel() != D:
            return False
        if not (Wu.is_contiguous() and Wv.is_contiguous() and Wsr.is_contiguous()):
            return False
    return B > 1


def readout_head_fused(allf, seg, dT, dB, per_order, ws=None, eps_mode=0):
    """-> tuple of NORMALISED session vectors y_i [B, d] (ReadoutHeadFused)"""
    flat = [t for po in per_order for t in po]
    return ReadoutHeadFused.apply(allf, seg, dT, dB, ws, eps_mode, *flat)


def readout_head(allf, seg, dT, dB, per_order):
    """per_order: [(v_i, Wu_i, bu_i, Wv_i, we_i, Wsr_i)] -> tuple of s_i [B, d] (before the optional normalisation): the grouped
    exact-fp32 launches (ReadoutHead).  bf16 mode at d = 128 / 256 takes readout_head_fused instead."""
    flat = [t for po in per_order for t in po]
    return ReadoutHead.apply(allf, seg, dT, dB, *flat)


class SegMeanAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, F, seg, B, dynB):
        H, F = _rows(H), _rows(F)
        N, D = F.shape
        out = torch.zeros(N, D, device=F.device, dtype=torch.float32)
        lib.srec_seg_mean_add_fwd(ptr(H), _ld(H), ptr(F), _ld(F), ptr(seg), B, ptr(dynB), D, ptr(out), D, stream())
        ctx.save_for_backward(seg)
        ctx.B, ctx.dynB = B, dynB
        return out

    @staticmethod
    def backward(ctx, g):
        (seg,) = ctx.saved_tensors
        g = _rows(g)
        N, D = g.shape
        dF = torch.zeros(N, D, device=g.device, dtype=torch.float32)
        lib.srec_seg_mean_add_bwd(ptr(g), _ld(g), ptr(seg), ctx.B, ptr(ctx.dynB), D, ptr(dF), D, stream())
        return g, dF, None, None, None


def seg_mean_add(H, F, seg, B, dynB=None):
    return SegMeanAdd.apply(H, F, seg, B, dynB)


# ------------------------------------------------------------------------------------------ scoring
def _bf16_dim_ok(d):
    return d <= 256 and d % 4 == 0


class CEWorkspace:
    """Reusable scratch of the fused scoring/CE kernels for one (B, V, d) (sized for the fp32 and bf16 plans)."""

    def __init__(self, B, V, d, device):
        import ctypes
        nt, nr, dp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.srec_ce_plan(B, V, d, ctypes.addressof(nt), ctypes.addressof(nr))
        nrange, nstat = nr.value, nt.value
        if _bf16_dim_ok(d):
            lib.srec_ce_plan_bf16(B, V, d, ctypes.addressof(nt), ctypes.addressof(nr), ctypes.addressof(dp))
            nrange, nstat = max(nrange, nr.value), max(nstat, nt.value)
        self.B, self.V, self.d = B, V, d
        self.stats = torch.empty(2 * nstat * B, device=device, dtype=torch.float32)
        self.dsr_part = torch.empty(nrange * B * d, device=device, dtype=torch.float32)
        self.lab_logit = torch.zeros(B, device=device, dtype=torch.float32)
        # bf16 operand copies of the session vectors (row-major + transposed), zero padded to 128 rows / d_pad columns
        self.Bp = (B + 127) // 128 * 128
        self.sr16 = self.srT16 = None
        self.sr_key = None
        self._de = {}
        if _bf16_dim_ok(d):
            self.sr16 = torch.zeros(self.Bp, dp.value, device=device, dtype=torch.bfloat16)


def _ce_de_slabs(self, B, V, d):
    """(split, workspace) of the session-split scoring backward at this shape (allocated by an eager step: a captured step
    finds it in the cache)"""
    ent = self._de.get((B, V))
    if ent is None:
        sp = _ct.c_int(1)
        lib.srec_ce_de_split(B, V, d, _ct.addressof(sp))
        split = int(sp.value)
        if split > 1 and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('the scoring workspace must be sized by an eager warm-up step before graph capture')
        buf = torch.empty(split * V * d, device=self.stats.device, dtype=torch.float32) if split > 1 else None
        ent = self._de[(B, V)] = (split, buf)
    return ent


CEWorkspace.de_slabs = _ce_de_slabs


class TableBF16:
    """bf16 copy of the item table for the bf16 scoring kernels: E16 [Vp, d_pad] (row-major only: the backward takes its
    transposed fragments with transposing LDS reads), refreshed once per step (one pass over the table)."""

    def __init__(self, table):
        import ctypes
        V, d = table.shape
        nt, nr, dp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib.srec_ce_plan_bf16(1, V, d, ctypes.addressof(nt), ctypes.addressof(nr), ctypes.addressof(dp))
        self.Vp = (V + 127) // 128 * 128
        self.E16 = torch.zeros(self.Vp, dp.value, device=table.device, dtype=torch.bfloat16)

    def refresh(self, table, max_norm=0.0):
        """one streaming pass: bf16 copy of every row; max_norm > 0: Embedding(max_norm)'s in-place renorm of the fp32 rows
        in the same pass (msgifsr.py:162 / lessr.py:126)"""
        V, d = table.shape
        if d <= 1024:
            with torch.no_grad():
                lib.srec_renorm_rows_bf16(ptr(table), table.stride(0), V, d, float(max_norm), ptr(self.E16), self.E16.shape[1],
                                          stream())
        else:
            assert max_norm <= 0
            lib.srec_bf16_prepare(ptr(table), table.stride(0), V, None, d, ptr(self.E16), None, self.Vp, stream())
        return self


def use_bf16_scoring(d):
    return PRECISION['matmul'] == 'bf16' and _bf16_dim_ok(d)


def _prepare_sr(sr, ws, dynB):
    """bf16 copies of the session vectors; skipped when the workspace still holds exactly this tensor (the backward
    of the head whose forward ran last)."""
    key = (sr.data_ptr(), sr._version, tuple(sr.shape))
    if ws.sr_key != key:
        B, d = sr.shape
        lib.srec_bf16_prepare(ptr(sr), _ld(sr), B, ptr(dynB), d, ptr(ws.sr16), None, ws.Bp, stream())
        ws.sr_key = key


def _ce_fwd(sr, table, cs, labels, ws, dynB, tb, lab, lse, lossvec, loss):
    B, d = sr.shape
    V = table.shape[0]
    if tb is not None:
        if getattr(ws, 'sr_fresh', None) == (sr.data_ptr(), sr.shape[0], sr.shape[1]):
            ws.sr_key = (sr.data_ptr(), sr._version, tuple(sr.shape))     # written by the normalisation that produced sr
        else:
            ws.sr_key = None
            _prepare_sr(sr, ws, dynB)
        ws.sr_fresh = None
        lib.srec_score_ce_fwd_bf16(ptr(ws.sr16), ws.Bp, ptr(tb.E16), tb.Vp, ptr(cs), ptr(labels), B, V, d, ptr(dynB),
                                   ptr(ws.stats), ptr(lab), ptr(lse), ptr(lossvec), ptr(loss), stream())
    else:
        lib.srec_score_ce_fwd(ptr(sr), _ld(sr), ptr(table), table.stride(0), ptr(cs), ptr(labels), B, V, d, ptr(dynB),
                              ptr(ws.stats), ptr(lab), ptr(lse), ptr(lossvec), ptr(loss), stream())


def _ce_bwd(sr, table, cs, labels, lse, gl, ga, gc, ws, dynB, tb, dE, dsr, parts):
    B, d = sr.shape
    V = table.shape[0]
    if tb is not None:
        _prepare_sr(sr, ws, dynB)
        # many sessions against few table rows (a rank's shard scored for the sessions of ALL ranks): the item tiles of the
        # backward are split over the sessions, slabs in a workspace the split decides the size of (srec_ce_de_split)
        split, slabs = ws.de_slabs(B, V, d) if (parts & 1) and dE.stride(0) == d else (1, None)
        lib.srec_score_ce_bwd_bf16(ptr(ws.sr16), ptr(slabs), ws.Bp, ptr(tb.E16), None, tb.Vp, ptr(cs),
                                   ptr(labels), ptr(lse), ptr(gl), ptr(ga), ptr(gc), B, V, d, ptr(dynB), ptr(dE),
                                   dE.stride(0), ptr(ws.dsr_part), ptr(dsr), parts | (split << 8), stream())
    else:
        lib.srec_score_ce_bwd(ptr(sr), _ld(sr), ptr(table), table.stride(0), ptr(cs), ptr(labels), ptr(lse), ptr(gl),
                              ptr(ga), ptr(gc), B, V, d, ptr(dynB), ptr(dE), dE.stride(0), ptr(ws.dsr_part), ptr(dsr),
                              parts, stream())


class ScoreCE(torch.autograd.Function):
    """loss = mean_b CE(cs * sr_b E^T, label_b), logits never materialised.  Writes the dense
    table gradient into `tgrad.buf` (all rows) instead of returning it."""

    @staticmethod
    def forward(ctx, sr, table, cs, labels, ws, tgrad, dynB, cs_inv_scale, tb=None):
        sr = _rows(sr)
        B, d = sr.shape
        V = table.shape[0]
        lse = torch.empty(B, device=sr.device, dtype=torch.float32)
        lossvec = torch.empty(B, device=sr.device, dtype=torch.float32)
        loss = torch.empty((), device=sr.device, dtype=torch.float32)
        _ce_fwd(sr, table, cs, labels, ws, dynB, tb, ws.lab_logit, lse, lossvec, loss)
        ctx.save_for_backward(sr, table, cs, labels, lse)
        ctx.ws, ctx.tgrad, ctx.dynB, ctx.cs_inv_scale, ctx.tb = ws, tgrad, dynB, cs_inv_scale, tb
        ctx.mark_non_differentiable(lse)
        ctx.set_materialize_grads(False)       # no zero-filled [B] gradient for the unused lse output (a fill kernel per step)
        return loss, lse

    @staticmethod
    def backward(ctx, gloss, _glse):
        sr, table, cs, labels, lse = ctx.saved_tensors
        B, d = sr.shape
        V = table.shape[0]
        tg, ws = ctx.tgrad, ctx.ws
        gl = gloss.reshape(1).to(torch.float32).contiguous()
        dsr = torch.empty(B, d, device=sr.device, dtype=torch.float32)
        tg.overwritten()
        _ce_bwd(sr, table, cs, labels, lse, gl, None, None, ws, ctx.dynB, ctx.tb, tg.buf, dsr, 3)
        if cs is not None and tg.defer:
            tg.pending = (table, cs, ctx.cs_inv_scale)     # applied by the optimizer's row pass (or TableGrad.materialize)
        elif cs is not None:    # rows were L2-normalised before scoring: project out the radial part
            lib.srec_rownorm_project(ptr(table), table.stride(0), ptr(cs), ctx.cs_inv_scale, ptr(tg.buf),
                                     tg.buf.stride(0), V, d, stream())
        tg.fresh = True
        return dsr, None, None, None, None, None, None, None, None


class ScoreStats(torch.autograd.Function):
    """(lse_b, z[b,label_b]) of the full-catalog logits, logits never materialised; differentiable in both
    outputs, so any loss built from them (mixtures of soft-maxes: msgifsr.py:311-317) trains through the
    fused kernels.  Several heads may share one table: the first backward of a step overwrites the dense
    table gradient, later ones accumulate."""

    @staticmethod
    def forward(ctx, sr, table, cs, labels, ws, tgrad, dynB, cs_inv_scale, tb=None):
        sr = _rows(sr)
        B, d = sr.shape
        V = table.shape[0]
        dev = sr.device
        lse = torch.empty(B, device=dev, dtype=torch.float32)
        lossvec = torch.empty(B, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        lab = torch.zeros(B, device=dev, dtype=torch.float32)
        _ce_fwd(sr, table, cs, labels, ws, dynB, tb, lab, lse, lossvec, loss)
        ctx.save_for_backward(sr, table, cs, labels, lse)
        ctx.ws, ctx.tgrad, ctx.dynB, ctx.cs_inv_scale, ctx.tb = ws, tgrad, dynB, cs_inv_scale, tb
        return lse, lab

    @staticmethod
    def backward(ctx, dlse, dlab):
        sr, table, cs, labels, lse = ctx.saved_tensors
        B, d = sr.shape
        V = table.shape[0]
        tg, ws = ctx.tgrad, ctx.ws
        ga = dlse.contiguous().float()
        gc = (-dlab).contiguous().float()
        dsr = torch.empty(B, d, device=sr.device, dtype=torch.float32)
        parts = 3 | (4 if tg.fresh else 0)
        if not tg.fresh:
            tg.overwritten()
        _ce_bwd(sr, table, cs, labels, lse, None, ga, gc, ws, ctx.dynB, ctx.tb, tg.buf, dsr, parts)
        if cs is not None and tg.defer:
            tg.pending = (table, cs, ctx.cs_inv_scale)     # linear: once, over the sum of the heads' contributions
        elif cs is not None:    # projection is linear and idempotent: safe after every accumulation
            lib.srec_rownorm_project(ptr(table), table.stride(0), ptr(cs), ctx.cs_inv_scale, ptr(tg.buf),
                                     tg.buf.stride(0), V, d, stream())
        tg.fresh = True
        return dsr, None, None, None, None, None, None, None, None


def score_stats(sr, table, cs, labels, ws, tgrad, dynB=None, cs_inv_scale=1.0, tb=None):
    return ScoreStats.apply(sr, table, cs, labels, ws, tgrad, dynB, cs_inv_scale, tb)


def score_ce(sr, table, cs, labels, ws, tgrad, dynB=None, cs_inv_scale=1.0, tb=None):
    return ScoreCE.apply(sr, table, cs, labels, ws, tgrad, dynB, cs_inv_scale, tb)


_TOPK_WS = {}


def score_topk(sr, table, cs, k):
    """(values [B,k], item ids [B,k]) of the k largest z[b,v] = cs[v] <sr_b, E_v> - no (B, V) tensor (evaluation)"""
    sr = _rows(sr.detach())
    B, d = sr.shape
    V = table.shape[0]
    n = _ct.c_long()
    lib.srec_score_topk_ws(B, V, k, _ct.addressof(n))
    key = (sr.device.index, n.value)
    ws = _TOPK_WS.get(key)
    if ws is None:
        ws = _TOPK_WS[key] = torch.empty(n.value, device=sr.device, dtype=torch.uint8)
    val = torch.empty(B, k, device=sr.device, dtype=torch.float32)
    idx = torch.empty(B, k, device=sr.device, dtype=torch.int32)
    lib.srec_score_topk(ptr(sr), _ld(sr), ptr(table), table.stride(0), ptr(cs), B, V, d, k, ptr(val), ptr(idx), ptr(ws),
                        stream())
    return val, idx


class ScoreLogProb(torch.autograd.Function):
    """(B,V) log-probabilities - the tensor the reference models' forward() returns (compat /
    evaluation path).  Backward materialises d z (B,V) and runs two MFMA GEMMs."""

    @staticmethod
    def forward(ctx, sr, table, cs, ws, cs_inv_scale):
        sr = _rows(sr)
        B, d = sr.shape
        V = table.shape[0]
        dev = sr.device
        lse = torch.empty(B, device=dev, dtype=torch.float32)
        lossvec = torch.empty(B, device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        zeros = torch.zeros(B, device=dev, dtype=torch.int32)
        lib.srec_score_ce_fwd(ptr(sr), _ld(sr), ptr(table), table.stride(0), ptr(cs), ptr(zeros), B, V, d, None,
                              ptr(ws.stats), ptr(ws.lab_logit), ptr(lse), ptr(lossvec), ptr(loss), stream())
        ldp = (V + 3) & ~3
        logp = torch.empty(B, ldp, device=dev, dtype=torch.float32)[:, :V]
        lib.srec_score_logp(ptr(sr), _ld(sr), ptr(table), table.stride(0), ptr(cs), ptr(lse), B, V, d, None, ptr(logp),
                            ldp, stream())
        ctx.save_for_backward(sr, table, cs, logp)
        ctx.cs_inv_scale = cs_inv_scale
        return logp

    @staticmethod
    def backward(ctx, g):
        sr, table, cs, logp = ctx.saved_tensors
        dz = g - torch.exp(logp) * g.sum(dim=1, keepdim=True)
        if cs is not None:
            dz = dz * cs.unsqueeze(0)
        V, d = table.shape
        ldp = (V + 3) & ~3
        dzp = torch.zeros(dz.shape[0], ldp, device=dz.device, dtype=torch.float32)
        dzp[:, :V] = dz
        tablep = table if ldp == V else _pad_rows(table, ldp)
        dsr = torch.empty_like(sr)
        gemm_nn(dzp, tablep, dsr)
        dEp = torch.empty(ldp, d, device=dz.device, dtype=torch.float32)
        gemm_tn(dzp, sr, dEp)
        dE = dEp[:V]
        if cs is not None:
            lib.srec_rownorm_project(ptr(table), table.stride(0), ptr(cs), ctx.cs_inv_scale, ptr(dE), dE.stride(0), V,
                                     d, stream())
        return dsr, dE, None, None, None


def _pad_rows(t, n):
    out = torch.zeros(n, t.shape[1], device=t.device, dtype=t.dtype)
    out[:t.shape[0]] = t
    return out


def score_logp(sr, table, cs, ws, cs_inv_scale=1.0):
    return ScoreLogProb.apply(sr, table, cs, ws, cs_inv_scale)


# ------------------------------------------------------------------------------------------ GRU
class GRUPointwise(torch.autograd.Function):
    """One GRU time step given the two projections (GI, GH); GH=None <=> h_prev = 0 (gh = b_hh)."""

    @staticmethod
    def forward(ctx, GI, GH, bhh, Hp, dyn):
        GI = _rows(GI)
        n, d3 = GI.shape
        d = d3 // 3
        dev = GI.device
        Hn = torch.empty(n, d, device=dev, dtype=torch.float32)
        gates = torch.empty(n, d3, device=dev, dtype=torch.float32)
        if GH is not None:
            GH, Hp = _rows(GH), _rows(Hp)
            lib.srec_gru_pointwise_fwd(ptr(GI), _ld(GI), ptr(GH), _ld(GH), None, ptr(Hp), _ld(Hp), n, ptr(dyn), d,
                                       ptr(Hn), d, ptr(gates), stream())
        else:
            bhh = bhh.contiguous()
            lib.srec_gru_pointwise_fwd(ptr(GI), _ld(GI), None, 0, ptr(bhh), None, 0, n, ptr(dyn), d, ptr(Hn), d,
                                       ptr(gates), stream())
        ctx.save_for_backward(gates, GH, bhh, Hp)
        ctx.dyn = dyn
        return Hn

    @staticmethod
    def backward(ctx, dHn):
        gates, GH, bhh, Hp = ctx.saved_tensors
        dHn = _rows(dHn)
        n, d3 = gates.shape
        d = d3 // 3
        dev = gates.device
        dGI = torch.empty(n, d3, device=dev, dtype=torch.float32)
        dGH = torch.empty(n, d3, device=dev, dtype=torch.float32)
        if GH is not None:
            dHp = torch.empty(n, d, device=dev, dtype=torch.float32)
            lib.srec_gru_pointwise_bwd(ptr(dHn), _ld(dHn), ptr(gates), ptr(GH), _ld(GH), None, ptr(Hp), _ld(Hp), n,
                                       ptr(ctx.dyn), d, ptr(dGI), d3, ptr(dGH), d3, ptr(dHp), d, stream())
            return dGI, dGH, None, dHp, None
        lib.srec_gru_pointwise_bwd(ptr(dHn), _ld(dHn), ptr(gates), None, 0, ptr(bhh), None, 0, n, ptr(ctx.dyn), d,
                                   ptr(dGI), d3, ptr(dGH), d3, None, 0, stream())
        db = torch.empty(d3, device=dev, dtype=torch.float32)
        col_sum(dGH, n, d3, db, ctx.dyn)
        return dGI, None, db, None, None


class UnbindMid(torch.autograd.Function):
    """x [n, k, m] -> k row-strided views x[:, t, :]; the backward is ONE stack instead of autograd's k zero-fills,
    k slice copies and k-1 adds (SelectBackward)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return tuple(x[:, t, :] for t in range(x.shape[1]))

    @staticmethod
    def backward(ctx, *gs):
        n, k, m = ctx.shape
        gs = [g if g is not None else torch.zeros(n, m, device=gs[0].device if gs[0] is not None else None) for g in gs]
        return torch.stack(gs, 1)


def unbind_mid(x):
    return UnbindMid.apply(x)


def gru_step(GI, GH, bhh, Hp, dyn=None):
    return GRUPointwise.apply(GI, GH, bhh, Hp, dyn)


class GramCombine(torch.autograd.Function):
    """0.5 * mean_t x[n,t,:] + 0.5 * h_last[n,:]"""

    @staticmethod
    def forward(ctx, X, Hl, k, dyn):
        X = X.contiguous()
        Hl = _rows(Hl)
        n, d = Hl.shape
        out = torch.empty(n, d, device=Hl.device, dtype=torch.float32)
        lib.srec_gram_combine_fwd(ptr(X), ptr(Hl), _ld(Hl), n, ptr(dyn), k, d, ptr(out), d, stream())
        ctx.k, ctx.dyn, ctx.xshape = k, dyn, tuple(X.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _rows(g)
        n, d = g.shape
        dX = torch.empty(ctx.xshape, device=g.device, dtype=torch.float32)
        dH = torch.empty(n, d, device=g.device, dtype=torch.float32)
        lib.srec_gram_combine_bwd(ptr(g), _ld(g), n, ptr(ctx.dyn), ctx.k, d, ptr(dX), ptr(dH), d, stream())
        return dX, dH, None, None


class GRUExpand(torch.autograd.Function):
    """MSGIFSR SemanticExpander for one order k (msgifsr.py:32-45): out = 0.5 * mean_t x[n,t,:] + 0.5 * GRU(x).h_last,
    as ONE autograd node on stacked buffers: the time steps share [k, n, *] tensors, so the backward needs one
    weight-gradient GEMM and one bias column-sum for W_hh / b_hh over all steps, the hidden-state gradient is
    accumulated by the backward-data GEMM itself (beta = 1) and nothing goes through autograd's select / add kernels."""

    @staticmethod
    def forward(ctx, x, Wih, bih, Whh, bhh, k, dyn_n, dyn_rows, combine=True):
        x = x.contiguous()
        nk, d = x.shape
        n, d3 = nk // k, 3 * d
        dev = x.device
        Wih, Whh, bhh = _rows(Wih), _rows(Whh), bhh.contiguous()
        GI = torch.empty(nk, d3, device=dev, dtype=torch.float32)
        gemm_nt(x, Wih, GI, bih, dyn_rows, 1 if dyn_rows is not None else 0)
        H = torch.empty(k, n, d, device=dev, dtype=torch.float32)
        gates = torch.empty(k, n, d3, device=dev, dtype=torch.float32)
        GH = torch.empty(max(k - 1, 1), n, d3, device=dev, dtype=torch.float32)
        st = stream()
        for t in range(k):
            gi = GI.data_ptr() + 4 * t * d3                                   # GI[:, t, :], row stride k * 3d
            if t == 0:
                lib.srec_gru_pointwise_fwd(gi, k * d3, None, 0, ptr(bhh), None, 0, n, ptr(dyn_n), d, ptr(H[0]), d,
                                           ptr(gates[0]), st)
            else:
                gemm_nt(H[t - 1], Whh, GH[t - 1], bhh, dyn_n, 1 if dyn_n is not None else 0)
                lib.srec_gru_pointwise_fwd(gi, k * d3, ptr(GH[t - 1]), d3, None, ptr(H[t - 1]), d, n, ptr(dyn_n), d,
                                           ptr(H[t]), d, ptr(gates[t]), st)
        ctx.save_for_backward(x, Wih, Whh, bhh, H, gates, GH)
        ctx.k, ctx.dyn_n, ctx.dyn_rows, ctx.combine = k, dyn_n, dyn_rows, combine
        if not combine:                                   # 'max' / 'concat' reducers: only the GRU's last hidden state
            return H[k - 1].clone()
        out = torch.empty(n, d, device=dev, dtype=torch.float32)
        lib.srec_gram_combine_fwd(ptr(x), ptr(H[k - 1]), d, n, ptr(dyn_n), k, d, ptr(out), d, st)
        return out

    @staticmethod
    def backward(ctx, g):
        x, Wih, Whh, bhh, H, gates, GH = ctx.saved_tensors
        k, dyn_n, dyn_rows = ctx.k, ctx.dyn_n, ctx.dyn_rows
        g = _rows(g)
        n, d = g.shape
        d3, dev, st = 3 * d, g.device, stream()
        if ctx.combine:
            dX = torch.empty(n * k, d, device=dev, dtype=torch.float32)
            dh = torch.empty(n, d, device=dev, dtype=torch.float32)
            lib.srec_gram_combine_bwd(ptr(g), _ld(g), n, ptr(dyn_n), k, d, ptr(dX), ptr(dh), d, st)
        else:
            dX = torch.zeros(n * k, d, device=dev, dtype=torch.float32)
            dh = g.contiguous()
        # d(gi) and d(gh) side by side in ONE [n k, 2 d3] buffer: both bias gradients are then one column-sum launch.
        # d(gi) rows are (node, t) = node k + t; d(gh) slot t = rows [t n, (t+1) n).  The gate kernel writes every row of
        # both (zeros for nodes past the live count).
        dG = torch.empty(n * k, 2 * d3, device=dev, dtype=torch.float32)
        dGI = dG[:, :d3]
        dGH = [dG[t * n:(t + 1) * n, d3:] for t in range(k)]
        for t in range(k - 1, -1, -1):
            dgi = dG.data_ptr() + 4 * t * 2 * d3
            if t > 0:
                dhp = torch.empty(n, d, device=dev, dtype=torch.float32)
                lib.srec_gru_pointwise_bwd(ptr(dh), d, ptr(gates[t]), ptr(GH[t - 1]), d3, None, ptr(H[t - 1]), d, n,
                                           ptr(dyn_n), d, dgi, k * 2 * d3, ptr(dGH[t]), 2 * d3, ptr(dhp), d, st)
                gemm_nn(dGH[t], Whh, dhp, dyn_n, 1 if dyn_n is not None else 0, beta=1.0)      # dh_{t-1} += dgh_t W_hh
                dh = dhp
            else:
                lib.srec_gru_pointwise_bwd(ptr(dh), d, ptr(gates[0]), None, 0, ptr(bhh), None, 0, n, ptr(dyn_n), d,
                                           dgi, k * 2 * d3, ptr(dGH[0]), 2 * d3, None, 0, st)
        gWhh = torch.zeros_like(Whh) if k == 1 else torch.empty_like(Whh)
        if k > 1:
            gemm_tn(dG[n:, d3:], H[:k - 1].reshape((k - 1) * n, d), gWhh, None)
        gb = torch.empty(2 * d3, device=dev, dtype=torch.float32)
        col_sum(dG, n * k, 2 * d3, gb, None)
        gbih, gbhh = gb[:d3], gb[d3:]
        gemm_nn(dGI, Wih, dX, dyn_rows, 1 if dyn_rows is not None else 0, beta=1.0)             # + the mean term
        gWih = torch.empty_like(Wih)
        gemm_tn(dGI, x, gWih, dyn_rows)
        return dX, gWih, gbih, gWhh, gbhh, None, None, None, None


def gru_expand(x, gru, k, dyn_n=None, dyn_rows=None, combine=True):
    return GRUExpand.apply(x, gru.weight_ih_l0, gru.bias_ih_l0, gru.weight_hh_l0, gru.bias_hh_l0, k, dyn_n, dyn_rows,
                           combine)


class GruStepDesc(_ct.Structure):
    """host mirror of srec_gru_step_desc (include/srec_hg.h)"""
    _fields_ = ([('np', _ct.c_int), ('d', _ct.c_int), ('n', _ct.c_int * 4), ('k', _ct.c_int * 4), ('t', _ct.c_int * 4),
                 ('dyn', _ct.c_void_p * 4)] +
                [(nm, _ct.c_void_p * 4) for nm in ('GI', 'GH', 'bih', 'bhh', 'Hp', 'Hn', 'Hn16', 'gates', 'X', 'out',
                                                   'dH', 'dout', 'dGI16', 'dGH16', 'dHp', 'dX', 'bias_part')] +
                [('part_row0', _ct.c_int * 4)])


class GruFusedDesc(_ct.Structure):
    """host mirror of srec_gru_fused_desc (include/srec_hg.h)"""
    _fields_ = ([('np', _ct.c_int), ('d', _ct.c_int), ('n', _ct.c_int * 4), ('k', _ct.c_int * 4), ('dyn', _ct.c_void_p * 4)] +
                [(nm, _ct.c_void_p * 4) for nm in ('X', 'X16', 'Wih_f', 'Whh_f', 'bih', 'bhh', 'H', 'H16', 'gates', 'out')])


class GruFusedBwdDesc(_ct.Structure):
    """host mirror of srec_gru_fused_bwd_desc (include/srec_hg.h)"""
    _fields_ = ([('np', _ct.c_int), ('d', _ct.c_int), ('n', _ct.c_int * 4), ('k', _ct.c_int * 4), ('dyn', _ct.c_void_p * 4)] +
                [(nm, _ct.c_void_p * 4) for nm in ('gates', 'H', 'dout', 'Wih_f', 'Whh_f', 'dGI16', 'dGH16', 'dX', 'bias_part')] +
                [('part_row0', _ct.c_int * 4)])


def gru_wfrag_t(ws):
    """fragment-major bf16 copies of GRU weights [3 d, d] for the backward-data products (csrc/grufb.hip), one launch"""
    n, d = len(ws), ws[0].shape[1]
    outs = [torch.empty(w.numel(), device=w.device, dtype=torch.bfloat16) for w in ws]
    arr = _ct.c_void_p * n
    a_w, a_o = arr(*[w.data_ptr() for w in ws]), arr(*[o.data_ptr() for o in outs])
    lib.srec_gru_wfrag_t(n, _ct.addressof(a_w), _ct.addressof(a_o), d, stream())
    return outs


def _gru_wfrag_args(ws):
    n = len(ws)
    of = [torch.empty(w.numel(), device=w.device, dtype=torch.bfloat16) for w in ws]
    ob = [torch.empty(w.numel(), device=w.device, dtype=torch.bfloat16) for w in ws]
    arr = _ct.c_void_p * n
    return of, ob, (arr(*[w.data_ptr() for w in ws]), arr(*[o.data_ptr() for o in of]), arr(*[o.data_ptr() for o in ob]))


def gru_wfrag_both(ws):
    """gru_wfrag and gru_wfrag_t of the same weights in one launch -> (forward copies, backward copies) - unless the prologue
    launch of this forward pass made them (step_prologue)"""
    hit = [_wprep_take('gru', w) for w in ws]
    if all(h is not None for h in hit):
        return [h[0] for h in hit], [h[1] for h in hit]
    of, ob, args = _gru_wfrag_args(ws)
    lib.srec_gru_wfrag_both(len(ws), *[_ct.addressof(a) for a in args], ws[0].shape[1], stream())
    return of, ob


def gru_wfrag(ws):
    """fragment-major bf16 copies of GRU weights [3 d, d] (the B operands of the fused forward, csrc/gruf.hip), one launch"""
    hit = [_wprep_take('gru', w) for w in ws]
    if all(h is not None for h in hit):
        return [h[0] for h in hit]
    n, d = len(ws), ws[0].shape[1]
    outs = [torch.empty(w.numel(), device=w.device, dtype=torch.bfloat16) for w in ws]
    arr = _ct.c_void_p * n
    a_w, a_o = arr(*[w.data_ptr() for w in ws]), arr(*[o.data_ptr() for o in outs])
    lib.srec_gru_wfrag(n, _ct.addressof(a_w), _ct.addressof(a_o), d, stream())
    return outs


def gru_fused_ok(d, P):
    return d in (128, 256) and P <= 4 and FUSED_GRU


def gru_expand_fast_ok(d, reducer):
    return PRECISION['matmul'] == 'bf16' and reducer == 'mean' and d % 64 == 0 and d <= 1024 and 256 % (d // 4) == 0


class GRUExpandAll(torch.autograd.Function):
    """SemanticExpander (msgifsr.py:32-45, reducer 'mean') for ALL orders k >= 2 of a batch as one autograd node on the bf16
    path (csrc/grux.hip + csrc/gemm16.hip): one grouped GEMM and one fused gate kernel per time step serve every order;
    in the backward the hidden-state gradient is accumulated by the backward-data GEMM itself (beta = 1), the bias
    gradients come from per-block partial sums of the gate kernels, the weight gradients from row-split products."""

    @staticmethod
    def _rows16(xs, d, dev, st):
        """bf16 copy of the gathered rows: one pass when the orders' rows are adjacent pieces of one buffer"""
        P = len(xs)
        esz = xs[0].element_size()
        adjacent = all(xs[p + 1].data_ptr() == xs[p].data_ptr() + xs[p].numel() * esz for p in range(P - 1))
        tot = sum(x.shape[0] for x in xs)
        x16all = torch.empty(tot, d, device=dev, dtype=torch.bfloat16)
        offs, o = [], 0
        for x in xs:
            offs.append(o)
            o += x.shape[0]
        if adjacent:
            lib.srec_rows_bf16(ptr(xs[0]), d, tot, None, d, ptr(x16all), st)
        else:
            for x, o in zip(xs, offs):
                lib.srec_rows_bf16(ptr(x), d, x.shape[0], None, d, x16all[o:].data_ptr(), st)
        x16 = [x16all[o:o + x.shape[0]] for x, o in zip(xs, offs)]
        return x16all, x16

    @staticmethod
    def forward(ctx, ks, dyn_ns, dyn_rows, *args):
        P = len(ks)
        ctx.tags = [_arena_tag(a) for a in args[:P]]        # pieces of a split tensor: their gradients go into its buffer
        xs = [a.contiguous() for a in args[:P]]
        params = args[P:]
        ctx.defer, ctx.wparams = defer_scope(), [params[4 * p + j] for p in range(P) for j in (0, 2)]
        ctx.bparams = [params[4 * p + j] for p in range(P) for j in (1, 3)]
        Wih, bih, Whh, bhh = ([params[4 * p + j].contiguous() for p in range(P)] for j in range(4))
        d = xs[0].shape[1]
        d3, dev, st = 3 * d, xs[0].device, stream()
        ns = [x.shape[0] // k for x, k in zip(xs, ks)]
        ctx.fused = gru_fused_ok(d, P)
        if not ctx.fused:
            w16, wt16 = weights_bf16([w for p in range(P) for w in (Wih[p], Whh[p])])
            Wih16, Whh16 = w16[0::2], w16[1::2]
        H = [torch.empty(ks[p], ns[p], d, device=dev, dtype=torch.float32) for p in range(P)]
        H16 = [torch.empty(max(ks[p] - 1, 1), ns[p], d, device=dev, dtype=torch.bfloat16) for p in range(P)]
        # saved gates: fp16 on the fused path (values in [-1, 1] and gh_n; half the bytes of the expander's largest tensor)
        gates = [torch.empty(ks[p], ns[p], 4 * d, device=dev, dtype=torch.float16 if ctx.fused else torch.float32) for p in range(P)]
        outs = [torch.empty(ns[p], d, device=dev, dtype=torch.float32) for p in range(P)]
        if ctx.fused:
            # the whole recurrence in one launch (csrc/gruf.hip): a workgroup owns 32 nodes, the weights stream from L2
            ws = [w for p in range(P) for w in (Wih[p], Whh[p])]
            if any(ctx.needs_input_grad):
                wf, wft = gru_wfrag_both(ws)         # wft: the backward's copies (csrc/grufb.hip)
            else:
                wf, wft = gru_wfrag(ws), []
            x16all = torch.empty(sum(x.shape[0] for x in xs), d, device=dev, dtype=torch.bfloat16)
            x16, o = [], 0
            for x in xs:
                x16.append(x16all[o:o + x.shape[0]])
                o += x.shape[0]
            q = GruFusedDesc()
            q.np, q.d = P, d
            for p in range(P):
                q.n[p], q.k[p], q.dyn[p] = ns[p], ks[p], ptr(dyn_ns[p])
                q.X[p], q.X16[p], q.Wih_f[p], q.Whh_f[p] = ptr(xs[p]), ptr(x16[p]), ptr(wf[2 * p]), ptr(wf[2 * p + 1])
                q.bih[p], q.bhh[p], q.H[p], q.H16[p] = ptr(bih[p]), ptr(bhh[p]), ptr(H[p]), ptr(H16[p])
                q.gates[p], q.out[p] = ptr(gates[p]), ptr(outs[p])
            lib.srec_gru_fused_fwd(_ct.addressof(q), st)
            ctx.save_for_backward(*x16, *H, *H16, *gates, *wft)
            ctx.meta = (ks, dyn_ns, dyn_rows, ns, d, [tuple(w.shape) for w in Wih])
            return tuple(outs)
        x16all, x16 = GRUExpandAll._rows16(xs, d, dev, st)
        GI = [torch.empty(x.shape[0], d3, device=dev, dtype=torch.float32) for x in xs]
        gemm16('nt', [(xs[p].shape[0], d3, d, [(x16[p], Wih16[p])], GI[p], dyn_rows[p]) for p in range(P)], d, d, d3,
               keep_dead=True)
        GH = [torch.empty(ns[p], d3, device=dev, dtype=torch.float32) for p in range(P)]
        for t in range(max(ks)):
            act = [p for p in range(P) if t < ks[p]]
            if t > 0:
                gemm16('nt', [(ns[p], d3, d, [(H16[p][t - 1], Whh16[p])], GH[p], dyn_ns[p]) for p in act], d, d, d3,
                       keep_dead=True)
            q = GruStepDesc()
            q.np, q.d = len(act), d
            for i, p in enumerate(act):
                q.n[i], q.k[i], q.t[i], q.dyn[i] = ns[p], ks[p], t, ptr(dyn_ns[p])
                q.GI[i], q.bih[i], q.bhh[i] = ptr(GI[p]), ptr(bih[p]), ptr(bhh[p])
                if t > 0:
                    q.GH[i], q.Hp[i] = ptr(GH[p]), ptr(H[p][t - 1])
                q.Hn[i], q.gates[i] = ptr(H[p][t]), ptr(gates[p][t])
                if t < ks[p] - 1:
                    q.Hn16[i] = ptr(H16[p][t])
                else:
                    q.X[i], q.out[i] = ptr(xs[p]), ptr(outs[p])
            lib.srec_gru_step_fwd(_ct.addressof(q), st)
        ctx.save_for_backward(*x16, *H, *H16, *gates, *wt16)
        ctx.meta = (ks, dyn_ns, dyn_rows, ns, d, [tuple(w.shape) for w in Wih])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        ks, dyn_ns, dyn_rows, ns, d, _ = ctx.meta
        P = len(ks)
        sv = ctx.saved_tensors
        x16, H, H16, gates = sv[:P], sv[P:2 * P], sv[2 * P:3 * P], sv[3 * P:4 * P]
        wt16 = sv[4 * P:]
        WihT16, WhhT16 = wt16[0::2], wt16[1::2]
        d3, dev, st = 3 * d, H[0].device, stream()
        gs = [g.contiguous() if g is not None else torch.zeros(ns[p], d, device=dev) for p, g in enumerate(gs)]
        rows = [ns[p] * ks[p] for p in range(P)]
        if all(t is not None for t in ctx.tags):
            dX = [_arena_rows(t, r, d, dev) for t, r in zip(ctx.tags, rows)]
        else:
            dXall = torch.empty(sum(rows), d, device=dev, dtype=torch.float32)
            offs, o = [], 0
            for r in rows:
                offs.append(o)
                o += r
            dX = [dXall[o:o + r] for o, r in zip(offs, rows)]
        dGI16 = [torch.empty(rows[p], d3, device=dev, dtype=torch.bfloat16) for p in range(P)]
        dGH16 = [torch.empty(max(ks[p] - 1, 1), ns[p], d3, device=dev, dtype=torch.bfloat16) for p in range(P)]   # slot t - 1
        if ctx.fused:
            # every time step of every order in one launch (csrc/grufb.hip); wt16 = the fragment-major weights here
            nr_c, ns_c = _ct.c_int(0), (_ct.c_int * P)(*ns)
            lib.srec_gru_fused_nodes(P, _ct.addressof(ns_c), d, _ct.addressof(nr_c))   # nodes per workgroup: one partial bias row each
            nr = nr_c.value
            part = [torch.empty((ns[p] + nr - 1) // nr, 6 * d, device=dev, dtype=torch.float32) for p in range(P)]
            q = GruFusedBwdDesc()
            q.np, q.d = P, d
            for p in range(P):
                q.n[p], q.k[p], q.dyn[p] = ns[p], ks[p], ptr(dyn_ns[p])
                q.gates[p], q.H[p], q.dout[p] = ptr(gates[p]), ptr(H[p]), ptr(gs[p])
                q.Wih_f[p], q.Whh_f[p] = ptr(wt16[2 * p]), ptr(wt16[2 * p + 1])
                q.dGI16[p], q.dGH16[p], q.dX[p], q.bias_part[p], q.part_row0[p] = ptr(dGI16[p]), ptr(dGH16[p]), ptr(dX[p]), ptr(part[p]), 0
            lib.srec_gru_fused_bwd(_ct.addressof(q), st)
            return GRUExpandAll._weight_grads(ctx, x16, H16, dGI16, dGH16, part, dX)
        rb = max(8, 1024 // d)                           # nodes per block of the step kernel
        nblk = [(ns[p] + rb - 1) // rb for p in range(P)]
        part = [torch.empty(ks[p] * nblk[p], 6 * d, device=dev, dtype=torch.float32) for p in range(P)]
        dHcur = [None] * P
        for t in range(max(ks) - 1, -1, -1):
            act = [p for p in range(P) if t < ks[p]]
            q = GruStepDesc()
            q.np, q.d = len(act), d
            nxt = {}
            for i, p in enumerate(act):
                q.n[i], q.k[i], q.t[i], q.dyn[i] = ns[p], ks[p], t, ptr(dyn_ns[p])
                q.gates[i] = ptr(gates[p][t])
                if t == ks[p] - 1:
                    q.dout[i], q.dX[i] = ptr(gs[p]), ptr(dX[p])
                else:
                    q.dH[i] = ptr(dHcur[p])
                q.dGI16[i] = ptr(dGI16[p])
                if t > 0:
                    q.Hp[i] = ptr(H[p][t - 1])
                    q.dGH16[i] = ptr(dGH16[p][t - 1])
                    nxt[p] = torch.empty(ns[p], d, device=dev, dtype=torch.float32)
                    q.dHp[i] = ptr(nxt[p])
                q.bias_part[i], q.part_row0[i] = ptr(part[p]), t * nblk[p]
            lib.srec_gru_step_bwd(_ct.addressof(q), st)
            if t > 0:          # d h_{t-1} += d(gh_t) W_hh
                gemm16('nt', [(ns[p], d, d3, [(dGH16[p][t - 1], WhhT16[p])], nxt[p], dyn_ns[p]) for p in act], d3, d3, d,
                       beta=1.0)
                for p in act:
                    dHcur[p] = nxt[p]
        # d x += d(gi) W_ih  (onto the mean term the last-step kernels wrote)
        gemm16('nt', [(rows[p], d, d3, [(dGI16[p], WihT16[p])], dX[p], dyn_rows[p]) for p in range(P)], d3, d3, d, beta=1.0)
        return GRUExpandAll._weight_grads(ctx, x16, H16, dGI16, dGH16, part, dX)

    @staticmethod
    def _weight_grads(ctx, x16, H16, dGI16, dGH16, part, dX):
        ks, dyn_ns, dyn_rows, ns, d, _ = ctx.meta
        P = len(ks)
        d3, dev, st = 3 * d, dX[0].device, stream()
        rows = [ns[p] * ks[p] for p in range(P)]
        # weight gradients: the reduction runs over rows - split in-kernel into ~512-row pieces (hundreds of short workgroups
        # instead of a dozen long ones), each writing its own slab; the slabs are summed in fixed order
        probs, slabs = [], []
        # (ctx.wparams = [W_ih, W_hh] per order, ctx.bparams = [b_ih, b_hh] per order: bucket slots when row-sharded)
        gWih = [grad_buf(ctx.wparams[2 * p]) for p in range(P)]
        gWhh = [grad_buf(ctx.wparams[2 * p + 1]) for p in range(P)]
        for p in range(P):
            nsp = max(1, (rows[p] + 511) // 512)
            sl = torch.empty(nsp, d3, d, device=dev, dtype=torch.float32) if nsp > 1 else gWih[p].unsqueeze(0)
            probs.append((d3, d, rows[p], [(dGI16[p], x16[p])], sl, dyn_rows[p], 0, nsp))
            if nsp > 1:
                slabs.append((sl, gWih[p]))
            nsp = max(1, (ns[p] + 511) // 512)
            sl = torch.empty(nsp, d3, d, device=dev, dtype=torch.float32) if nsp > 1 else gWhh[p].unsqueeze(0)
            segs = [(dGH16[p][t - 1], H16[p][t - 1]) for t in range(1, ks[p])]
            probs.append((d3, d, ns[p], segs, sl, dyn_ns[p], 0, nsp))
            if nsp > 1:
                slabs.append((sl, gWhh[p]))
        for i in range(0, len(probs), 16):
            gemm16('tn', probs[i:i + 16], d3, d, d)
        # bias gradients from the partial rows
        gb = [grad_buf_pair(ctx.bparams[2 * p], ctx.bparams[2 * p + 1]) for p in range(P)]
        # the weight-gradient slab sums join the ONE end-of-backward launch (defer_slab_sum); the bias partials keep their own
        # kernel: hundreds of partial rows of only 6 d columns - as a task of the generic slab sum (one thread per 4 columns
        # walking all rows) they made that launch 44 us (profiles/r03d), gru_bias_final splits the rows over 16 lanes: 5 us
        ok = can_defer(ctx.defer, ctx.wparams)
        for sl, o_ in slabs:
            defer_slab_sum(sl, o_, ok)
        if ok and can_defer(ctx.defer, ctx.bparams):
            # ... and so do the bias partials (hundreds of rows of 6 d columns: the "tall" tasks of srec_sum_slabs_multi)
            for p in range(P):
                defer_slab_sum(part[p], gb[p], True, tall=True)
        else:
            arr = _ct.c_void_p * P
            a_p, a_o = arr(*[t_.data_ptr() for t_ in part]), arr(*[t_.data_ptr() for t_ in gb])
            a_r = (_ct.c_int * P)(*[t_.shape[0] for t_ in part])
            lib.srec_gru_bias_final(P, _ct.addressof(a_p), _ct.addressof(a_r), 6 * d, _ct.addressof(a_o), st)
        grads = []
        for p in range(P):
            grads += [gWih[p], gb[p][:d3], gWhh[p], gb[p][d3:]]
        return (None, None, None) + tuple(dX) + tuple(grads)


def gru_expand_all(xs, grus, ks, dyn_ns, dyn_rows):
    """xs[i]: [N_k k, d] gathered rows of order ks[i] (>= 2) -> [N_k, d] expander outputs, all orders in one node"""
    params = []
    for g in grus:
        params += [g.weight_ih_l0, g.bias_ih_l0, g.weight_hh_l0, g.bias_hh_l0]
    return GRUExpandAll.apply(tuple(ks), tuple(dyn_ns), tuple(dyn_rows), *xs, *params)


def gram_combine(X, Hl, k, dyn=None):
    return GramCombine.apply(X, Hl, k, dyn)


# ------------------------------------------------------------------------------------------ GAT
class GATRelation(torch.autograd.Function):
    """rst[v,h,:] = sum_{u->v} softmax_v(LeakyReLU(el_u + er_v)) * feat_src[u,h,:] for one relation.
    graph = (in_ptr, in_idx, out_ptr, out_idx, esrc, edst); feats are [N, H*D]."""

    @staticmethod
    def forward(ctx, Fs, Fd, attn_l, attn_r, graph, H, dyn_ns, dyn_nd, slope):
        Fs, Fd = _rows(Fs), _rows(Fd)
        in_ptr, in_idx, out_ptr, out_idx, esrc, edst = graph
        Ns, HD = Fs.shape
        Nd = Fd.shape[0]
        D = HD // H
        E = esrc.numel()
        dev = Fs.device
        al, ar = attn_l.reshape(-1).contiguous(), attn_r.reshape(-1).contiguous()
        el = torch.empty(Ns, H, device=dev, dtype=torch.float32)
        er = torch.empty(Nd, H, device=dev, dtype=torch.float32)
        lib.srec_head_dot(ptr(Fs), _ld(Fs), ptr(al), Ns, ptr(dyn_ns), H, D, ptr(el), stream())
        lib.srec_head_dot(ptr(Fd), _ld(Fd), ptr(ar), Nd, ptr(dyn_nd), H, D, ptr(er), stream())
        A = torch.empty(max(E, 1), H, device=dev, dtype=torch.float32)    # every live edge written by the kernel
        rst = torch.empty(Nd, HD, device=dev, dtype=torch.float32)
        lib.srec_gat_agg_fwd(ptr(Fs), _ld(Fs), ptr(el), ptr(er), ptr(in_ptr), ptr(in_idx), ptr(esrc), Nd, ptr(dyn_nd),
                             H, D, slope, ptr(A), ptr(rst), HD, stream())
        ctx.save_for_backward(Fs, Fd, al, ar, el, er, A)
        ctx.same = Fs.data_ptr() == Fd.data_ptr() and Fs.shape == Fd.shape
        ctx.graph, ctx.H, ctx.dyn, ctx.slope = graph, H, (dyn_ns, dyn_nd), slope
        ctx.shapes = (attn_l.shape, attn_r.shape)
        return rst

    @staticmethod
    def backward(ctx, dR):
        Fs, Fd, al, ar, el, er, A = ctx.saved_tensors
        in_ptr, in_idx, out_ptr, out_idx, esrc, edst = ctx.graph
        dyn_ns, dyn_nd = ctx.dyn
        H = ctx.H
        dR = _rows(dR)
        Ns, HD = Fs.shape
        Nd = Fd.shape[0]
        D = HD // H
        dev = Fs.device
        DP = torch.empty_like(A)
        der = torch.empty(Nd, H, device=dev, dtype=torch.float32)
        lib.srec_gat_bwd_dst(ptr(dR), _ld(dR), ptr(Fs), _ld(Fs), ptr(el), ptr(er), ptr(A), ptr(in_ptr), ptr(in_idx),
                             ptr(esrc), Nd, ptr(dyn_nd), H, D, ctx.slope, ptr(DP), ptr(der), stream())
        dFs = torch.empty(Ns, HD, device=dev, dtype=torch.float32)
        del_ = torch.empty(Ns, H, device=dev, dtype=torch.float32)
        same = ctx.same
        lib.srec_gat_bwd_src(ptr(dR), _ld(dR), ptr(A), ptr(DP), ptr(al), ptr(out_ptr), ptr(out_idx), ptr(edst), Ns,
                             ptr(dyn_ns), H, D, ptr(dFs), HD, ptr(del_), ptr(der) if same else None,
                             ptr(ar) if same else None, stream())
        dFd = None                                    # same tensor as Fs: its er-side term is already in dFs
        if not same:
            dFd = torch.empty(Nd, HD, device=dev, dtype=torch.float32)
            lib.srec_head_outer(ptr(der), ptr(ar), Nd, ptr(dyn_nd), H, D, ptr(dFd), HD, stream())
        dal = torch.empty(HD, device=dev, dtype=torch.float32)
        dar = torch.empty(HD, device=dev, dtype=torch.float32)
        col_sum(Fs, Ns, HD, dal, dyn_ns, del_, H, D)
        col_sum(Fd, Nd, HD, dar, dyn_nd, der, H, D)
        return dFs, dFd, dal.view(ctx.shapes[0]), dar.view(ctx.shapes[1]), None, None, None, None, None


def gat_relation(Fs, Fd, attn_l, attn_r, graph, H, dyn_ns=None, dyn_nd=None, slope=0.2):
    return GATRelation.apply(Fs, Fd, attn_l, attn_r, graph, H, dyn_ns, dyn_nd, slope)


class HeadCombine(torch.autograd.Function):
    """out[v,:] = max_head( sum_i R_i[v,h,:] + bias[h,:] + nres * x[v,:] )"""

    @staticmethod
    def forward(ctx, x, bias, nres, H, dyn, *rsts):
        import ctypes
        x = _rows(x)
        N, D = x.shape
        dev = x.device
        rsts = [_rows(r) for r in rsts]
        bias = bias.reshape(-1).contiguous()
        out = torch.empty(N, D, device=dev, dtype=torch.float32)
        arg = torch.empty(N, D, device=dev, dtype=torch.uint8)
        arr = (ctypes.c_void_p * max(len(rsts), 1))(*[r.data_ptr() for r in rsts])
        lib.srec_head_combine_fwd(ctypes.addressof(arr), len(rsts), H * D, ptr(x), _ld(x), ptr(bias), float(nres), N,
                                  ptr(dyn), H, D, ptr(out), D, ptr(arg), stream())
        ctx.save_for_backward(arg)
        ctx.meta = (H, D, N, nres, dyn, len(rsts))
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        H, D, N, nres, dyn, nr = ctx.meta
        g = _rows(g)
        dR = torch.empty(N, H * D, device=g.device, dtype=torch.float32)
        lib.srec_head_combine_bwd(ptr(g), _ld(g), ptr(arg), N, ptr(dyn), H, D, ptr(dR), H * D, stream())
        dbias = torch.empty(H * D, device=g.device, dtype=torch.float32)
        col_sum(dR, N, H * D, dbias, dyn)
        dx = g * nres if nres != 0 else None
        return (dx, dbias, None, None, None) + tuple(dR for _ in range(nr))


def head_combine(x, bias, nres, H, rsts, dyn=None):
    return HeadCombine.apply(x, bias, nres, H, dyn, *rsts)


# ------------------------------------------------------------------------------------------ LESSR
class BatchNorm(torch.autograd.Function):
    """nn.BatchNorm1d over the live rows (training: batch statistics + running-stat update)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, training, momentum, eps, dyn):
        x = _rows(x)
        n, D = x.shape
        dev = x.device
        y = torch.empty(n, D, device=dev, dtype=torch.float32)
        if training:
            mean = torch.empty(D, device=dev, dtype=torch.float32)
            var = torch.empty(D, device=dev, dtype=torch.float32)
            assert nbt is None or nbt.dtype == torch.int64
            lib.srec_bn_fwd_train(ptr(x), _ld(x), n, ptr(dyn), D, ptr(gamma), ptr(beta), float(eps), float(momentum),
                                  ptr(rmean), ptr(rvar), ptr(nbt), ptr(mean), ptr(var), ptr(y), D, ptr(_ws(64 * D, dev)),
                                  stream())
        else:
            mean, var = rmean, rvar
            lib.srec_bn_apply_fwd(ptr(x), _ld(x), ptr(mean), ptr(var), float(eps), ptr(gamma), ptr(beta), n, ptr(dyn), D,
                                  ptr(y), D, stream())
        ctx.save_for_backward(x, mean, var, gamma)
        ctx.meta = (training, eps, dyn)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, mean, var, gamma = ctx.saved_tensors
        training, eps, dyn = ctx.meta
        gy = _rows(gy)
        n, D = x.shape
        dev = x.device
        dx = torch.empty(n, D, device=dev, dtype=torch.float32)
        dg = torch.empty(D, device=dev, dtype=torch.float32)
        db = torch.empty(D, device=dev, dtype=torch.float32)
        lib.srec_bn_bwd(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(mean), ptr(var), float(eps), ptr(gamma),
                        1 if training else 0, n, ptr(dyn), D, ptr(dx), D, ptr(dg), ptr(db), ptr(_ws(64 * D, dev)),
                        stream())
        return dx, dg, db, None, None, None, None, None, None, None


def batch_norm(x, bn, dyn=None):
    """bn: an nn.BatchNorm1d used as parameter / running-stat container"""
    track = bn.training and bn.track_running_stats
    return BatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked if track else None,
                           bn.training, bn.momentum, bn.eps, dyn)


class PReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a, dyn):
        x = _rows(x)
        n, D = x.shape
        y = torch.empty(n, D, device=x.device, dtype=torch.float32)
        lib.srec_prelu_fwd(ptr(x), _ld(x), ptr(a), n, ptr(dyn), D, ptr(y), D, stream())
        ctx.save_for_backward(x, a)
        ctx.dyn = dyn
        ctx.defer, ctx.wparams = defer_scope(), [a]
        return y

    @staticmethod
    def backward(ctx, gy):
        x, a = ctx.saved_tensors
        gy = _rows(gy)
        n, D = x.shape
        dx = torch.empty(n, D, device=x.device, dtype=torch.float32)
        da = torch.empty(D, device=x.device, dtype=torch.float32)
        if D % 4 == 0 and n > 0 and can_defer(ctx.defer, ctx.wparams):
            # the 32 chunk partials of d a wait (private buffer) for the end-of-backward slab-sum launch: da = NULL skips the
            # kernel's own final sum
            part = torch.empty(32, D, device=x.device, dtype=torch.float32)
            lib.srec_prelu_bwd(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(a), n, ptr(ctx.dyn), D, ptr(dx), D, None, ptr(part), stream())
            defer_slab_sum(part, da, True, tall=True)
        else:
            lib.srec_prelu_bwd(ptr(gy), _ld(gy), ptr(x), _ld(x), ptr(a), n, ptr(ctx.dyn), D, ptr(dx), D, ptr(da),
                               ptr(_ws(32 * D, x.device)), stream())
        return dx, da, None


def prelu(x, a, dyn=None):
    return PReLU.apply(x, a, dyn)


class GRUSeq(torch.autograd.Function):
    """EOPA: last hidden state of a GRU run over each node's in-neighbours in edge-id order.
    GI = ft W_ih^T + b_ih (per source node); graph = (in_ptr, in_idx, out_ptr, out_idx, esrc, edst)."""

    @staticmethod
    def forward(ctx, GI, Whh, bhh, graph, dyn, dynE=None):
        GI = _rows(GI)
        in_ptr, in_idx, out_ptr, out_idx, esrc, edst = graph
        N, D3 = GI.shape
        D = D3 // 3
        E = esrc.numel()
        dev = GI.device
        Whh = Whh.contiguous()
        WhhT = Whh.t().contiguous() if D > 32 else None      # D <= 32: the kernel keeps W_hh's rows in registers
        neigh = torch.empty(N, D, device=dev, dtype=torch.float32)
        # edge records: every live edge is an in-edge of exactly one live node and is written by the kernel; every reader
        # (the backward kernel, the dyn-clamped weight-gradient products) stops at the live edges - no zero fill
        gates = torch.empty(max(E, 1), D3, device=dev, dtype=torch.float32)
        Hprev = torch.empty(max(E, 1), D, device=dev, dtype=torch.float32)
        ghn = torch.empty(max(E, 1), D, device=dev, dtype=torch.float32)
        lib.srec_gru_seq_fwd(ptr(GI), _ld(GI), ptr(Whh), ptr(WhhT), ptr(bhh), ptr(in_ptr), ptr(in_idx), ptr(esrc), N, ptr(dyn), D,
                             ptr(neigh), D, ptr(gates), ptr(Hprev), ptr(ghn), stream())
        ctx.save_for_backward(Whh, gates, Hprev, ghn)
        ctx.graph, ctx.dyn, ctx.dynE, ctx.shape = graph, dyn, dynE, (N, D, E)
        ctx.defer, ctx.wparams = defer_scope(), [Whh, bhh]
        return neigh

    @staticmethod
    def backward(ctx, dneigh):
        Whh, gates, Hprev, ghn = ctx.saved_tensors
        in_ptr, in_idx, out_ptr, out_idx, esrc, edst = ctx.graph
        N, D, E = ctx.shape
        dev = Whh.device
        dneigh = _rows(dneigh)
        dGIe = torch.empty(max(E, 1), 3 * D, device=dev, dtype=torch.float32)
        dGHe = torch.empty(max(E, 1), 3 * D, device=dev, dtype=torch.float32)
        lib.srec_gru_seq_bwd(ptr(dneigh), _ld(dneigh), ptr(Whh), ptr(gates), ptr(Hprev), ptr(ghn), ptr(in_ptr),
                             ptr(in_idx), N, ptr(ctx.dyn), D, ptr(dGIe), ptr(dGHe), stream())
        # per-source sum of the edge records (out-edge CSR), then weight gradients as GEMMs over the E records
        dGI = torch.empty(N, 3 * D, device=dev, dtype=torch.float32)      # every live row written (rows past dyn: never read)
        ar = _arange(N + 1, dev)
        lib.srec_scatter_add_sorted(ptr(dGIe), 3 * D, ptr(ar), ptr(out_ptr), ptr(out_idx), ptr(dGI), 3 * D, N,
                                    ptr(ctx.dyn), 3 * D, 0, stream())
        dWhh = (torch.empty if E > 0 else torch.zeros)(3 * D, D, device=dev, dtype=torch.float32)
        dbhh = (torch.empty if E > 0 else torch.zeros)(3 * D, device=dev, dtype=torch.float32)
        if E > 0 and _small_linear(Whh):                       # one grouped launch: d W_hh and d b_hh (product with ones)
            sums = torch.empty(4, 3 * D, device=dev, dtype=torch.float32)
            gemm_f32_group([('tn', dGHe[:E], Hprev[:E], dWhh, None, ctx.dynE, 0.0),
                            ('tn', _ones4(E, dev)[:E], dGHe[:E], sums, None, ctx.dynE, 0.0)],
                           defer=[dWhh, sums] if can_defer(ctx.defer, ctx.wparams) else None)
            dbhh = sums[0]
        elif E > 0:
            gemm_tn(dGHe[:E], Hprev[:E], dWhh, ctx.dynE)       # padded edge records are zero rows
            col_sum(dGHe, E, 3 * D, dbhh, ctx.dynE)
        return dGI, dWhh, dbhh, None, None, None


def gru_seq(GI, Whh, bhh, graph, dyn=None, dynE=None):
    return GRUSeq.apply(GI, Whh, bhh, graph, dyn, dynE)


class SGATAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, K, we, Vf, graph, dyn):
        Q, K, Vf = _rows(Q), _rows(K), _rows(Vf)
        in_ptr, in_idx, out_ptr, out_idx, esrc, edst = graph
        N, Hh = Q.shape
        Do = Vf.shape[1]
        E = esrc.numel()
        dev = Q.device
        we = we.reshape(-1).contiguous()
        A = torch.zeros(max(E, 1), device=dev, dtype=torch.float32)
        out = torch.empty(N, Do, device=dev, dtype=torch.float32)
        lib.srec_sgat_fwd(ptr(Q), _ld(Q), ptr(K), _ld(K), ptr(we), ptr(Vf), _ld(Vf), ptr(in_ptr), ptr(in_idx), ptr(esrc),
                          N, ptr(dyn), Hh, Do, ptr(A), ptr(out), Do, stream())
        ctx.save_for_backward(Q, K, we, Vf, A)
        ctx.graph, ctx.dyn = graph, dyn
        return out

    @staticmethod
    def backward(ctx, g):
        Q, K, we, Vf, A = ctx.saved_tensors
        in_ptr, in_idx, out_ptr, out_idx, esrc, edst = ctx.graph
        g = _rows(g)
        N, Hh = Q.shape
        Do = Vf.shape[1]
        E = esrc.numel()
        dev = Q.device
        dQe = torch.zeros(max(E, 1), Hh, device=dev, dtype=torch.float32)
        dK = torch.empty(N, Hh, device=dev, dtype=torch.float32)
        dwp = torch.empty(N, Hh, device=dev, dtype=torch.float32)
        lib.srec_sgat_bwd_dst(ptr(g), _ld(g), ptr(Q), _ld(Q), ptr(K), _ld(K), ptr(we), ptr(Vf), _ld(Vf), ptr(A),
                              ptr(in_ptr), ptr(in_idx), ptr(esrc), N, ptr(ctx.dyn), Hh, Do, ptr(dQe), ptr(dK), Hh,
                              ptr(dwp), Hh, stream())
        dQ = torch.empty(N, Hh, device=dev, dtype=torch.float32)
        dV = torch.empty(N, Do, device=dev, dtype=torch.float32)
        lib.srec_sgat_bwd_src(ptr(g), _ld(g), ptr(A), ptr(dQe), ptr(out_ptr), ptr(out_idx), ptr(edst), N, ptr(ctx.dyn),
                              Hh, Do, ptr(dQ), Hh, ptr(dV), Do, stream())
        dwe = torch.empty(Hh, device=dev, dtype=torch.float32)
        col_sum(dwp, N, Hh, dwe, ctx.dyn)
        return dQ, dK, dwe.view(1, Hh), dV, None, None


def sgat_attn(Q, K, we, Vf, graph, dyn=None):
    return SGATAttn.apply(Q, K, we, Vf, graph, dyn)


class EdgeAgg(torch.autograd.Function):
    """out[v] = sum_{e into v} coef[e] x[src(e)] (coef constant); backward = the same kernel on the out-edge CSR."""

    @staticmethod
    def forward(ctx, x, coef, fwd_csr, bwd_csr, dyn):
        x = _rows(x)
        N, D = x.shape
        out = torch.empty(N, D, device=x.device, dtype=torch.float32)
        p, i, o = fwd_csr
        lib.srec_edge_agg(ptr(x), _ld(x), ptr(p), ptr(i), ptr(o), ptr(coef), N, ptr(dyn), D, ptr(out), D, stream())
        ctx.save_for_backward(coef)
        ctx.bwd_csr, ctx.dyn = bwd_csr, dyn
        return out

    @staticmethod
    def backward(ctx, g):
        (coef,) = ctx.saved_tensors
        g = _rows(g)
        N, D = g.shape
        dx = torch.empty(N, D, device=g.device, dtype=torch.float32)
        p, i, o = ctx.bwd_csr
        lib.srec_edge_agg(ptr(g), _ld(g), ptr(p), ptr(i), ptr(o), ptr(coef), N, ptr(ctx.dyn), D, ptr(dx), D, stream())
        return dx, None, None, None, None


def edge_coef(ptr_, idx, ew, n, dyn=None):
    coef = torch.zeros(max(ew.numel(), 1), device=ew.device, dtype=torch.float32)
    lib.srec_edge_coef(ptr(ptr_), ptr(idx), ptr(ew), n, ptr(dyn), ptr(coef), stream())
    return coef


def edge_agg(x, coef, fwd_csr, bwd_csr, dyn=None):
    return EdgeAgg.apply(x, coef, fwd_csr, bwd_csr, dyn)


# ------------------------------------------------------------------------------------------ MSHGNN layer (batched)


class HgDesc(_ct.Structure):
    """host mirror of srec_hg_desc (include/srec_hg.h)"""
    _T, _M, _B, _I = 4, 8, 16, 16
    _fields_ = ([(n, _ct.c_int) for n in ('H', 'D', 'n_types', 'n_mods', 'n_blocks', 'n_inst', 'B')] +
                [('slope', _ct.c_float), ('p16', _ct.c_int), ('dynB', _ct.c_void_p),
                 ('row0', _ct.c_int * 4), ('ncap', _ct.c_int * 4), ('dyn_n', _ct.c_void_p * 4), ('seg', _ct.c_void_p * 4)] +
                [(n, _ct.c_void_p * 8) for n in ('P', 'dP', 'W', 'V', 'Z', 'attn_l', 'attn_r', 'bias', 'd_attn_l', 'd_attn_r', 'd_bias')] +
                [('xin', _ct.c_void_p * 8), ('xres', _ct.c_void_p), ('rm', _ct.c_void_p)] +
                [(n, _ct.c_int * 16) for n in ('blk_mod', 'blk_type', 'blk_row')] +
                [(n, _ct.c_void_p * 16) for n in ('eL', 'eR', 'wL', 'wR')] +
                [(n, _ct.c_int * 16) for n in ('inst_mod', 'inst_sblk', 'inst_dblk')] +
                [(n, _ct.c_void_p * 16) for n in ('in_ptr', 'in_idx', 'esrc', 'out_ptr', 'out_idx', 'edst', 'A', 'DP', 'der', 'Mk')] +
                [('smean', _ct.c_void_p * 4), ('sess', _ct.c_void_p)] +
                [('rm_cnt', _ct.c_void_p), ('rm_counter', _ct.c_void_p), ('rm_p', _ct.c_float), ('rm_seed', _ct.c_int),
                 ('rm_salt', _ct.c_int)])


class GemmGroup(_ct.Structure):
    """host mirror of srec_gemm_group (include/srec_hg.h)"""
    _fields_ = [('np', _ct.c_int), ('lda', _ct.c_int), ('ldb', _ct.c_int), ('ldc', _ct.c_int), ('beta', _ct.c_float),
                ('a16', _ct.c_int), ('c16', _ct.c_int),
                ('M', _ct.c_int * 8), ('N', _ct.c_int * 8), ('K', _ct.c_int * 8), ('nseg', _ct.c_int * 8),
                ('A', (_ct.c_void_p * 4) * 8), ('B', (_ct.c_void_p * 4) * 8), ('C', _ct.c_void_p * 8),
                ('dyn', _ct.c_void_p * 8)]


def gemm_group(mode, probs, lda, ldb, ldc, beta=0.0, a16=False, c16=False):
    """probs: [(M, N, K, [(A, B), ...] segments, C, dyn)] tensors -> one srec_gemm_group_bf16 launch"""
    g = GemmGroup()
    g.np, g.lda, g.ldb, g.ldc, g.beta, g.a16, g.c16 = len(probs), lda, ldb, ldc, beta, int(a16), int(c16)
    for p, (M, N, K, segs, C, dyn) in enumerate(probs):
        g.M[p], g.N[p], g.K[p], g.nseg[p], g.C[p], g.dyn[p] = M, N, K, len(segs), ptr(C), ptr(dyn)
        for si, (A, B) in enumerate(segs):
            g.A[p][si], g.B[p][si] = ptr(A), ptr(B)
    lib.srec_gemm_group_bf16(_ct.addressof(g), mode, stream())


class GemmGroup16(_ct.Structure):
    """host mirror of srec_gemm16_group (include/srec_hg.h)"""
    _fields_ = [('np', _ct.c_int), ('lda', _ct.c_int), ('ldb', _ct.c_int), ('ldc', _ct.c_int), ('beta', _ct.c_float),
                ('c16', _ct.c_int),
                ('M', _ct.c_int * 16), ('N', _ct.c_int * 16), ('K', _ct.c_int * 16), ('nseg', _ct.c_int * 16),
                ('A', (_ct.c_void_p * 4) * 16), ('B', (_ct.c_void_p * 4) * 16), ('C', _ct.c_void_p * 16),
                ('dyn', _ct.c_void_p * 16), ('koff', _ct.c_int * 16), ('nsplit', _ct.c_int * 16),
                ('lda_p', _ct.c_int * 16), ('ldb_p', _ct.c_int * 16), ('ldc_p', _ct.c_int * 16), ('mhint', _ct.c_int * 16)]


def gemm16(kind, probs, lda, ldb, ldc, beta=0.0, c16=False, keep_dead=False):
    """one grouped launch of the bf16-in-HBM GEMMs (csrc/gemm16.hip).  probs: [(M, N, K, [(A16, B16), ...], C, dyn)].
    kind 'nt': C [M, N] (+)= sum_s A_s [M, K] B_s [N, K]^T (c16: bf16 output);  'tn': C [M, N] = sum_s A_s [K, M]^T B_s [K, N]
    (reduction over the K rows, clamped by dyn)."""
    assert 0 < len(probs) <= 16
    g = GemmGroup16()
    g.np, g.lda, g.ldb, g.ldc, g.beta, g.c16 = len(probs), lda, ldb, ldc, beta, int(c16) | (2 if keep_dead else 0)
    for p, pr in enumerate(probs):
        M, N, K, segs, C, dyn = pr[:6]
        g.M[p], g.N[p], g.K[p], g.nseg[p], g.C[p], g.dyn[p] = M, N, K, len(segs), ptr(C), ptr(dyn)
        g.koff[p] = pr[6] if len(pr) > 6 else 0          # tn: first reduction row of this piece of a row-split product
        g.nsplit[p] = pr[7] if len(pr) > 7 else 1        # tn: in-kernel row split, C = [nsplit, M, N] slabs
        if len(pr) > 8 and pr[8] is not None:            # per-problem (lda, ldb, ldc); 0 = the group's
            g.lda_p[p], g.ldb_p[p], g.ldc_p[p] = pr[8]
        if len(pr) > 9 and pr[9]:                        # nt: expected live rows of a capacity-padded problem
            g.mhint[p] = int(pr[9])
        for si, (A, B) in enumerate(segs):
            g.A[p][si], g.B[p][si] = ptr(A), ptr(B)
    (lib.srec_gemm16_nt if kind == 'nt' else lib.srec_gemm16_tn)(_ct.addressof(g), stream())


def rows_bf16(x, dyn=None):
    """bf16 copy of fp32 rows [n, d] (zero rows past the live count)"""
    x = _rows(x)
    n, d = x.shape
    out = torch.empty(n, d, device=x.device, dtype=torch.bfloat16)
    lib.srec_rows_bf16(ptr(x), _ld(x), n, ptr(dyn), d, ptr(out), stream())
    return out


def _weights_bf16_args(ws, transposed=True):
    n = len(ws)
    assert 0 < n <= 8
    dev = ws[0].device
    w16 = [torch.empty(w.shape, device=dev, dtype=torch.bfloat16) for w in ws]
    wt16 = [torch.empty(w.shape[1], w.shape[0], device=dev, dtype=torch.bfloat16) if transposed else None for w in ws]
    arr = _ct.c_void_p * n
    return w16, wt16, (arr(*[w.data_ptr() for w in ws]), arr(*[w.data_ptr() for w in w16]),
                       arr(*[(w.data_ptr() if w is not None else None) for w in wt16]),
                       (_ct.c_int * n)(*[w.shape[0] for w in ws]), (_ct.c_int * n)(*[w.shape[1] for w in ws]))


def weights_bf16(ws, transposed=True):
    """[(W16, WT16)] bf16 copies (and transposed copies) of up to 8 contiguous fp32 matrices, one launch - unless the prologue
    launch of this forward pass made them (step_prologue)"""
    hit = [_wprep_take('w16', w) for w in ws]
    if all(h is not None and (h[1] is not None or not transposed) for h in hit):
        return [h[0] for h in hit], [h[1] for h in hit]
    w16, wt16, args = _weights_bf16_args(ws, transposed)          # (args: kept alive across the call)
    lib.srec_weights_bf16(len(ws), *[_ct.addressof(a) for a in args], stream())
    return w16, wt16


class HgPlan:
    """Static topology of one MSHGNN layer call (built by msgifsr.MSHGNN from the FlatBatch).
    types:   [(row0, ncap, dyn_n, seg)]                          node types, stacked rows
    modules: [(row_start, n_rows, dyn)]                          rows of x each GAT module projects (dyn or None)
    blocks:  [(module, type)]                                    projection blocks
    insts:   [(module, src_block, dst_block, (in_ptr, in_idx, out_ptr, out_idx, esrc, edst))]"""

    def __init__(self, H, D, slope, B, dynB, types, modules, blocks, insts, mod_conv=None):
        self.mod_conv = mod_conv if mod_conv is not None else [0] * len(modules)      # 0: conv1, 1: conv2 (reversed graph)
        self.layer_id = 0                                 # dropout-mask salt: distinct per MSHGNN layer of a model
        self.H, self.D, self.slope, self.B, self.dynB = H, D, slope, B, dynB
        self.types, self.modules, self.blocks, self.insts = types, modules, blocks, insts
        assert len(types) <= 4 and len(modules) <= 8 and len(blocks) <= 16 and len(insts) <= 16

    def pieces(self, m):
        """[(row offset inside module m's projection, first stacked row, rows, dyn)] - one piece per node type the module
        covers: GEMM problems are cut at type boundaries so that every piece carries its own live row count (the shared
        'inter' module spans all types; as ONE problem it would multiply the capacity padding between them too)"""
        r0, nr, dyn = self.modules[m]
        out = [(t0 - r0, t0, nc, dyn_t) for (t0, nc, dyn_t, _) in self.types if r0 <= t0 and t0 + nc <= r0 + nr]
        assert sum(p[2] for p in out) == nr, 'a module projects whole node types'
        return out

    def scratch_layout(self):
        """-> (n_floats, offsets) of the small per-call scratch: eL,eR,wL,wR per block; A,DP,der per instance"""
        H, off, lay = self.H, 0, {}
        for b, (m, t) in enumerate(self.blocks):
            n = self.types[t][1] * H
            for nm in ('eL', 'eR', 'wL', 'wR'):
                lay[(nm, b)] = off
                off += n
        # (Z slot t < n_types also holds the summed bias row of node type t during the forward: slots for max(modules, types))
        for m in range(max(len(self.modules), len(self.types))):
            for nm in ('V', 'Z'):
                lay[(nm, m)] = off
                off += 2 * self.D * H
        for i, (m, sb, db, gr) in enumerate(self.insts):
            E = max(gr[4].numel(), 1) * H
            nd = self.types[self.blocks[db][1]][1] * H
            for nm, n in (('A', E), ('DP', E), ('der', nd)):
                lay[(nm, i)] = off
                off += n
        for t in range(len(self.types)):                   # session means of the input rows per node type [B, D]
            lay[('smean', t)] = off
            off += self.B * self.D
        lay[('sess', 0)] = off                             # session of every stacked row (int32)
        off += sum(tp[1] for tp in self.types)
        return off, lay

    def fill(self, desc, small, lay, P, dP, params, grads, drop=None):
        d = desc
        if drop is not None:
            xc, xres, rm, mk = drop[:4]
            for m in range(len(self.modules)):
                d.xin[m] = ptr(xc[self.mod_conv[m]])
            d.xres, d.rm = ptr(xres), ptr(rm)
            if rm is None and len(drop) > 6:               # rm recomputed in the backward from the masks' hash
                pf_, seed_, rc_, salt_ = drop[5]
                d.rm_cnt, d.rm_counter, d.rm_p, d.rm_seed, d.rm_salt = ptr(drop[6]), rc_, pf_, seed_, salt_
            for i in range(len(self.insts)):
                d.Mk[i] = ptr(mk[i]) if mk is not None else None
        d.H, d.D, d.slope, d.B = self.H, self.D, self.slope, self.B
        d.p16 = int(P[0].dtype == torch.bfloat16) if P else 0          # (P = None: the weight-only view srec_hg_fold reads)
        d.n_types, d.n_mods, d.n_blocks, d.n_inst = len(self.types), len(self.modules), len(self.blocks), len(self.insts)
        d.dynB = ptr(self.dynB)
        for t, (r0, nc, dyn, seg) in enumerate(self.types):
            d.row0[t], d.ncap[t], d.dyn_n[t], d.seg[t] = r0, nc, ptr(dyn), ptr(seg)
        base = small.data_ptr()
        for m in range(len(self.modules)):
            W, al, ar, bias = params[4 * m:4 * m + 4]
            d.P[m] = ptr(P[m]) if P else None
            d.W[m] = ptr(W)
            d.V[m], d.Z[m] = base + 4 * lay[('V', m)], base + 4 * lay[('Z', m)]
            d.attn_l[m], d.attn_r[m], d.bias[m] = ptr(al), ptr(ar), ptr(bias)
            if dP is not None:
                d.dP[m] = ptr(dP[m])
                d.d_attn_l[m], d.d_attn_r[m], d.d_bias[m] = (grads[m][j].data_ptr() for j in range(3))
        for m in range(len(self.modules), len(self.types)):
            d.Z[m] = base + 4 * lay[('Z', m)]
        for t in range(len(self.types)):
            d.smean[t] = base + 4 * lay[('smean', t)]
        d.sess = base + 4 * lay[('sess', 0)]
        for b, (m, t) in enumerate(self.blocks):
            d.blk_mod[b], d.blk_type[b] = m, t
            d.blk_row[b] = self.types[t][0] - self.modules[m][0]
            for nm in ('eL', 'eR', 'wL', 'wR'):
                getattr(d, nm)[b] = base + 4 * lay[(nm, b)]
        for i, (m, sb, db, gr) in enumerate(self.insts):
            d.inst_mod[i], d.inst_sblk[i], d.inst_dblk[i] = m, sb, db
            for nm, g in zip(('in_ptr', 'in_idx', 'out_ptr', 'out_idx', 'esrc', 'edst'), gr):
                getattr(d, nm)[i] = ptr(g)
            for nm in ('A', 'DP', 'der'):
                getattr(d, nm)[i] = base + 4 * lay[(nm, i)]
        return d


_HG_WS = {}


# ---- the step's prologue launch (csrc/prep.hip) ----------------------------------------------------------------------------
class StepPrepDesc(_ct.Structure):
    """host mirror of srec_step_prep_desc (include/srec_hg.h)"""
    _fields_ = [('hg', _ct.c_void_p), ('n_w16', _ct.c_int), ('w16_W', _ct.c_void_p), ('w16_out', _ct.c_void_p), ('w16_T', _ct.c_void_p),
                ('w16_R', _ct.c_void_p), ('w16_C', _ct.c_void_p), ('n_gru', _ct.c_int), ('gru_d', _ct.c_int), ('gru_W', _ct.c_void_p),
                ('gru_fwd', _ct.c_void_p), ('gru_bwd', _ct.c_void_p), ('n_head', _ct.c_int), ('head_W', _ct.c_void_p),
                ('head_out', _ct.c_void_p), ('head_rows', _ct.c_void_p), ('head_cols', _ct.c_void_p), ('head_trans', _ct.c_void_p),
                ('mailbox', _ct.c_void_p), ('M', _ct.c_int), ('counter', _ct.c_void_p), ('box_dst', _ct.c_void_p),
                ('box_cap', _ct.c_long), ('box_err', _ct.c_void_p)]


# graph.GraphedTrainStep: the mailbox intake of the step being captured, (mailbox, M, counter, dst, cap, err) as srec_copy_words_mailbox
# takes them, waiting to leave with the model's prologue launch (or on its own, ahead of the first read of the batch: flush_intake)
PENDING_INTAKE = []
STEP_PROLOGUE = os.environ.get('SREC_STEP_PROLOGUE', '1') != '0'     # (tests / A-B runs: 0 = the one-launch-per-reader sequence)


def flush_intake():
    """launch a pending batch intake on its own: called where a model first reads its batch on the device (the lookup) - a model
    with a prologue launch (MSGIFSR) has taken it along before"""
    while PENDING_INTAKE:
        box, M, counter, dst, cap, err = PENDING_INTAKE.pop(0)[:6]
        lib.srec_copy_words_mailbox(box, M, counter, dst, cap, err, stream())


def step_prologue(w16=(), gru=(), head=(), fold=None):
    """Everything a forward pass does to its WEIGHTS before it reads the first row of its batch, plus a pending batch intake, as
    ONE launch (srec_step_prep, csrc/prep.hip) instead of one small launch in front of each reader:
      w16:  fp32 matrices (<= 8) -> bf16 + transposed bf16 copies, taken by weights_bf16 (HGATLayer);
      gru:  GRU weights [3 d, d] (<= 8) -> both fragment-major layouts, taken by gru_wfrag_both / gru_wfrag (GRUExpandAll);
      head: [(W, trans)] (<= 16) -> hi / lo fragment-major copies, found by head_wfrag (ReadoutHeadFused) in its per-step cache;
      fold: (plan, params) of an MSHGNN layer call -> plan.pre = its scratch with V / the bias sums written (HGATLayer).
    Purely an optimisation: a reader that does not find its copies makes them itself."""
    q, keep = StepPrepDesc(), []
    w16 = [w for i, w in enumerate(w16) if all(w is not v for v in w16[:i])][:8]
    if w16:
        a16, t16, args = _weights_bf16_args(w16)
        q.n_w16 = len(w16)
        q.w16_W, q.w16_out, q.w16_T, q.w16_R, q.w16_C = (_ct.addressof(a) for a in args)
        keep.append(args)
    gru = list(gru)[:8]
    if gru:
        of, ob, args = _gru_wfrag_args(gru)
        q.n_gru, q.gru_d = len(gru), gru[0].shape[1]
        q.gru_W, q.gru_fwd, q.gru_bwd = (_ct.addressof(a) for a in args)
        keep.append(args)
    head = [(w, int(t)) for w, t in head if (w.data_ptr(), tuple(w.shape), int(t)) not in _HEAD_WF_CACHE][:16]
    if head:
        hb, args = _head_wfrag_args([w for w, _ in head], [t for _, t in head])
        q.n_head = len(head)
        q.head_W, q.head_out, q.head_rows, q.head_cols, q.head_trans = (_ct.addressof(a) for a in args)
        keep.append(args)
    if fold is not None:
        plan, params = fold
        nfl, lay = plan.scratch_layout()
        small = torch.empty(max(nfl, 1), device=params[0].device, dtype=torch.float32)
        flat = [p_.reshape(-1) if i % 4 else p_ for i, p_ in enumerate(params)]
        hd = plan.fill(HgDesc(), small, lay, None, None, flat, None, None)
        q.hg = _ct.addressof(hd)
        keep.append((hd, flat))
    box = PENDING_INTAKE.pop() if PENDING_INTAKE else None           # (one intake rides along; others leave ahead of it)
    flush_intake()
    if box is not None:
        q.mailbox, q.M, q.counter, q.box_dst, q.box_cap, q.box_err = box[:6]
    lib.srec_step_prep(_ct.addressof(q), stream())
    for w, a, t in zip(w16, a16 if w16 else (), t16 if w16 else ()):
        _wprep_put('w16', w, (a, t))
    for w, f, b_ in zip(gru, of if gru else (), ob if gru else ()):
        _wprep_put('gru', w, (f, b_))
    for (w, t), b_ in zip(head, hb if head else ()):
        _HEAD_WF_CACHE[(w.data_ptr(), tuple(w.shape), t)] = b_
    if fold is not None:
        plan.pre = (small, lay)


_CNT_CACHE = {}


def _inst_counts(plan, NT, dev):
    """[2, NT, 1]: how many relation instances of conv1 / conv2 use each stacked row as a DESTINATION (identity residual
    count, gatconv.py:306-308) - a function of the plan's static topology"""
    key = (str(dev), NT, tuple(plan.mod_conv), tuple((m, tuple(plan.types[plan.blocks[db][1]][:2])) for (m, sb, db, gr) in plan.insts))
    cnt = _CNT_CACHE.get(key)
    if cnt is None:
        host = torch.zeros(2, NT, 1)
        for (m, sb, db, gr) in plan.insts:
            t0, nc = plan.types[plan.blocks[db][1]][:2]
            host[plan.mod_conv[m], t0:t0 + nc] += 1.0
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('instance counts must be cached by an eager warm-up step before graph capture')
        cnt = _CNT_CACHE[key] = host.to(dev)
    return cnt


DROP_TAP = None      # tests: set to a list to receive {'ms': [2, NT, D], 'mk': [per instance (E*H,)]} of every dropout layer call


class HGATLayer(torch.autograd.Function):
    """out = MSHGNN(x): all relation instances of conv1 / conv2 in one batched pass (csrc/hgat.hip) around the fc
    GEMMs.  params = (fc.weight, attn_l, attn_r, bias) per module, in plan.modules order.
    drop = (p_feat, p_attn) in training: feature dropout with ONE mask per (conv, node type) on the inputs of that
    conv's GATConv modules (projection, logits and identity residual all see the dropped rows, gatconv.py:268-308;
    the reference draws one mask per (relation, role) - documented deviation) and attention dropout on the edge
    soft-max (gatconv.py:300)."""

    @staticmethod
    def forward(ctx, x, plan, drop, *params):
        x = _rows(x)
        NT, D = x.shape
        H = plan.H
        HD = H * D
        dev = x.device
        nm = len(plan.modules)
        dstate = None
        if drop is not None and (drop[0] > 0 or drop[1] > 0):
            pf, pa = drop
            # both convs' feature masks and every instance's attention mask in ONE launch (counter-based hash: no generator
            # state, replay-safe); the per-row instance counts depend on the plan only (cached)
            cnt = _inst_counts(plan, NT, dev)
            # (the mask tensor itself is written only for the tests' tap: the backward recomputes the masks from the hash)
            ms = torch.empty(2, NT, D, device=dev, dtype=torch.float32) if DROP_TAP is not None else None
            xcs = torch.empty(2, NT, D, device=dev, dtype=torch.float32)
            rm, xres = None, torch.empty(NT, D, device=dev)     # (rm = cnt0 m0 + cnt1 m1 is recomputed by the backward)
            xcont = x.contiguous()
            mk, allm, na = None, None, 0
            if pa > 0:
                sizes = [max(gr[4].numel(), 1) * H for (_, _, _, gr) in plan.insts]
                na = sum(sizes)
                allm = torch.empty(na, device=dev, dtype=torch.float32)
                mk = list(torch.split(allm, sizes))
            seed, rc = rng_args(dev)
            # the bf16 GEMM path reads bf16(xcs): written by the same pass
            x16_pre = None
            if PRECISION['matmul'] == 'bf16' and D % 64 == 0 and nm <= 8 and all(params[4 * m].is_contiguous() for m in range(nm)):
                x16_pre = torch.empty(2, NT, D, device=dev, dtype=torch.bfloat16)
                lib.srec_hg_drop_prep16(ptr(xcont), ptr(cnt), NT, D, float(pf), seed, rc, 101 + 2 * plan.layer_id, ptr(ms),
                                        ptr(xcs), ptr(rm), ptr(xres), float(pa), na, ptr(allm), ptr(x16_pre), stream())
            else:
                lib.srec_hg_drop_prep(ptr(xcont), ptr(cnt), NT, D, float(pf), seed, rc, 101 + 2 * plan.layer_id, ptr(ms),
                                      ptr(xcs), ptr(rm), ptr(xres), float(pa), na, ptr(allm), stream())
            xc = [xcs[0], xcs[1]]
            dstate = (xc, xres, rm, mk, ms, (float(pf), seed, rc, 101 + 2 * plan.layer_id), cnt)
            if DROP_TAP is not None:
                DROP_TAP.append(dict(ms=ms.clone(), mk=[m.clone() for m in mk] if mk is not None else None))
        xin = (lambda m: dstate[0][plan.mod_conv[m]]) if dstate is not None else (lambda m: x)
        grouped = PRECISION['matmul'] == 'bf16' and D % 8 == 0 and _ld(x) == D and nm <= 8
        # bf16 GEMM path: the projections (and their gradients) are STORED as bf16 too - every pass over them is HBM bound
        P = [torch.empty(nr, HD, device=dev, dtype=torch.bfloat16 if grouped else torch.float32)
             for (r0, nr, dyn) in plan.modules]
        g16 = None
        if grouped and D % 64 == 0 and all(params[4 * m].is_contiguous() for m in range(nm)):
            # every GEMM operand as bf16 in HBM (csrc/gemm16.hip): the small weights once per call (+ transposed copies for
            # the backward-data product), the module inputs in one pass
            w16, wt16 = weights_bf16([params[4 * m] for m in range(nm)])
            if dstate is not None:
                x16 = x16_pre if x16_pre is not None else rows_bf16(xcs.view(2 * NT, D)).view(2, NT, D)
                xin16 = lambda m: x16[plan.mod_conv[m]]
            else:
                x16 = rows_bf16(x)
                xin16 = lambda m: x16
            live = getattr(plan, 'live', {})             # host-side live row counts per type start (tile heuristics only)
            probs = [(nc, HD, D, [(xin16(m)[t0:t0 + nc], w16[m])], P[m][o:o + nc], dyn_t, 0, 1, None, live.get(t0, 0))
                     for m in range(nm) for (o, t0, nc, dyn_t) in plan.pieces(m)]
            for i in range(0, len(probs), 16):
                # rows past a type's live count are never read (every hgat.hip kernel walks the live prefix only)
                gemm16('nt', probs[i:i + 16], D, D, HD, c16=True, keep_dead=True)
            g16 = (x16, wt16)
        elif grouped:
            gemm_group(0, [(nr, HD, D, [(xin(m)[r0:r0 + nr], params[4 * m])], P[m], dyn)
                           for m, (r0, nr, dyn) in enumerate(plan.modules)], D, D, HD, c16=True)
        else:
            for m, (r0, nr, dyn) in enumerate(plan.modules):
                gemm_nt(xin(m)[r0:r0 + nr], _rows(params[4 * m]), P[m], None, dyn, 1 if dyn is not None else 0)
        pre = plan.__dict__.pop('pre', None)         # the prologue launch of this forward already folded the weights (step_prologue)
        if pre is not None:
            small, lay = pre
        else:
            nfl, lay = plan.scratch_layout()
            small = torch.empty(max(nfl, 1), device=dev, dtype=torch.float32)
        out = torch.empty(NT, D, device=dev, dtype=torch.float32)
        arg = torch.empty(NT, D, device=dev, dtype=torch.uint8)
        flat = [p.reshape(-1) if i % 4 else p for i, p in enumerate(params)]
        desc = plan.fill(HgDesc(), small, lay, P, None, flat, None, dstate)
        if pre is not None:
            desc.p16 |= 8
        lib.srec_hg_fwd(_ct.addressof(desc), ptr(x), _ld(x), ptr(out), D, ptr(arg), stream())
        ctx.save_for_backward(x, small, arg, *P, *params)
        ctx.plan, ctx.lay, ctx.grouped, ctx.dstate, ctx.g16 = plan, lay, grouped, dstate, g16
        ctx.defer, ctx.wparams = defer_scope(), [params[4 * m] for m in range(len(plan.modules))]
        return out

    @staticmethod
    def backward(ctx, g):
        plan, lay, dstate = ctx.plan, ctx.lay, ctx.dstate
        nm = len(plan.modules)
        x, small, arg = ctx.saved_tensors[:3]
        P = ctx.saved_tensors[3:3 + nm]
        params = ctx.saved_tensors[3 + nm:]
        g = _rows(g)
        NT, D = x.shape
        HD = plan.H * D
        dev = x.device
        # rows of a module's projection that no relation instance touches (a type without live 'inter' edges) get no
        # gradient from the kernels: those buffers start from zero
        cov = [sum(plan.types[bt][1] for bm, bt in plan.blocks if bm == m) for m in range(nm)]
        dP = [torch.empty_like(p) if cov[m] == p.shape[0] else torch.zeros_like(p) for m, p in enumerate(P)]
        # (attn_l, attn_r, bias gradients: [HD] each - the parameters' bucket slots when the table is row-sharded, ops.grad_buf)
        grads = [[grad_buf(params[4 * m + 1 + j]).view(-1) for j in range(3)] for m in range(nm)]
        dx = torch.empty(NT, D, device=dev, dtype=torch.float32)
        flat = [p.reshape(-1) if i % 4 else p for i, p in enumerate(params)]
        desc = plan.fill(HgDesc(), small, lay, P, dP, flat, grads, dstate)
        if ctx.g16 is not None and desc.p16:
            desc.p16 |= 2           # both gemm16 consumers of dP stop at the live rows: srec_hg_bwd leaves capacity-padding rows unwritten
        # feature dropout with recomputed masks: d x is written once, after the backward-data GEMMs (srec_hg_pre_merge)
        late_dx = dstate is not None and dstate[2] is None and len(dstate) > 6 and dstate[4] is None and _ld(g) % 4 == 0
        if late_dx:
            desc.p16 |= 4
        n = _ct.c_long()
        lib.srec_hg_ws_floats(_ct.addressof(desc), _ct.addressof(n))
        key = (dev.index, n.value)
        ws = _HG_WS.get(key)
        if ws is None:
            ws = _HG_WS[key] = torch.empty(max(n.value, 1), device=dev, dtype=torch.float32)
        lib.srec_hg_bwd(_ct.addressof(desc), ptr(x), _ld(x), ptr(g), _ld(g), ptr(arg), ptr(dx), D, ptr(ws), stream())
        xin = (lambda m: dstate[0][plan.mod_conv[m]]) if dstate is not None else (lambda m: x)
        gWs = [grad_buf(params[4 * m]) for m in range(nm)]
        convs = (0, 1) if dstate is not None else (None,)
        tgts = None
        S = 1
        if dstate is not None:
            # every node type is projected by some module of each conv in the usual plans: the grouped GEMM then writes all
            # rows (beta = 0 zeroes rows past the live count) and the buffers need no fill
            full = ctx.grouped and all(any(plan.mod_conv[bm] == cv and bt == t for bm, bt in plan.blocks)
                                       for cv in (0, 1) for t in range(len(plan.types)))
            # gemm16: when every (conv, type) is projected by the same number S of modules (intra_k + the shared 'inter'),
            # each module's product goes to its own partial buffer (S x more, S x shorter reduction loops in flight) and
            # the merge kernel sums them
            nsegs = {sum(1 for m in range(nm) if plan.mod_conv[m] == cv and any(bm == m and bt == t for bm, bt in plan.blocks))
                     for cv in (0, 1) for t in range(len(plan.types))}
            if ctx.g16 is not None and full and len(nsegs) == 1 and 1 < min(nsegs) <= 4:
                S = min(nsegs)
            tgts = (torch.empty if full else torch.zeros)(2, S, NT, D, device=dev, dtype=torch.float32)
        pend = []                                    # gemm16: the two convs' backward-data problems share ONE launch
        for cv in convs:
            # d x of one node type = sum over the modules that project it: the module sum is the K loop (segments).
            # With feature dropout the two convs see differently masked inputs: one masked contribution per conv.
            tgt = dx if cv is None else tgts[cv, 0]
            mods = [m for m in range(nm) if cv is None or plan.mod_conv[m] == cv]
            if ctx.grouped:
                probs = []
                for t, (t0, nc, dyn_t, _) in enumerate(plan.types):
                    segs = [(dP[m][t0 - plan.modules[m][0]:t0 - plan.modules[m][0] + nc], m) for m in mods
                            if plan.modules[m][0] <= t0 and t0 + nc <= plan.modules[m][0] + plan.modules[m][1]
                            and any(bm == m and bt == t for bm, bt in plan.blocks)]
                    if segs:
                        probs.append((nc, D, HD, segs, tgt[t0:t0 + nc], dyn_t, t0))
                beta = 0.0 if (cv is not None and full) else 1.0
                if probs and ctx.g16 is not None and S > 1:
                    wt16 = ctx.g16[1]
                    live = getattr(plan, 'live', {})
                    for (M_, N_, K_, segs_, C_, dyn_, t0) in probs:
                        for j, (A_, m_) in enumerate(segs_):
                            pend.append(((M_, N_, K_, [(A_, wt16[m_])], tgts[cv, j, t0:t0 + M_], dyn_, 0, 1, None, live.get(t0, 0)), 0.0))
                elif probs and ctx.g16 is not None:
                    wt16 = ctx.g16[1]
                    live = getattr(plan, 'live', {})
                    pend += [((M_, N_, K_, [(A_, wt16[m_]) for A_, m_ in segs_], C_, dyn_, 0, 1, None, live.get(_t0, 0)), beta)
                             for (M_, N_, K_, segs_, C_, dyn_, _t0) in probs]
                elif probs:
                    probs = [(M_, N_, K_, [(A_, params[4 * m_]) for A_, m_ in segs_], C_, dyn_)
                             for (M_, N_, K_, segs_, C_, dyn_, _t0) in probs]
                    gemm_group(1, probs, HD, D, D, beta=beta, a16=True)
            else:
                for m in mods:
                    r0, nr, dyn = plan.modules[m]
                    gemm_nn(dP[m], _rows(params[4 * m]), tgt[r0:r0 + nr], dyn, 1 if dyn is not None else 0, beta=1.0)
        while pend:
            beta = pend[0][1]
            batch = [pr for pr, b in pend if b == beta][:16]
            pend = [(pr, b) for pr, b in pend if not any(pr is q for q in batch)]
            gemm16('nt', batch, HD, HD, D, beta=beta)
        if late_dx:
            lib.srec_hg_pre_merge(_ct.addressof(desc), ptr(g), _ld(g), ptr(tgts), S, ptr(dx), D, stream())
        elif dstate is not None:
            pf_, seed_, rc_, salt_ = dstate[5]
            lib.srec_hg_drop_merge(ptr(tgts), S, ptr(dstate[4]), NT * D, ptr(dx), pf_, seed_, rc_, salt_, stream())
        if ctx.g16 is not None:
            # weight gradients, one balanced problem per (module, node type): a module that spans several types (the shared
            # 'inter' one) writes one slab per type, summed in fixed order afterwards
            x16 = ctx.g16[0]
            xin16 = (lambda m: x16[plan.mod_conv[m]]) if dstate is not None else (lambda m: x16)
            pcs = [plan.pieces(m) for m in range(nm)]
            multi = [m for m in range(nm) if len(pcs[m]) > 1]
            slabs = {}
            if multi:
                gWm = [gWs[m] for m in multi]
                for i, m in enumerate(multi):
                    slabs[m] = torch.empty(len(pcs[m]), HD, D, device=dev, dtype=torch.float32)
            probs = []
            for m in range(nm):
                for pi, (o, t0, nc, dyn_t) in enumerate(pcs[m]):
                    tgt = gWs[m] if m not in slabs else slabs[m][pi:pi + 1]
                    probs.append((HD, D, nc, [(dP[m][o:o + nc], xin16(m)[t0:t0 + nc])], tgt, dyn_t, 0, 1))
            for i in range(0, len(probs), 16):
                gemm16('tn', probs[i:i + 16], HD, D, D)
            if multi and can_defer(ctx.defer, [ctx.wparams[m] for m in multi]):
                for i, m in enumerate(multi):
                    defer_slab_sum(slabs[m], gWm[i])
            elif multi:
                _launch_slab_sums([(slabs[m], gWm[i]) for i, m in enumerate(multi)])
        elif ctx.grouped:
            gemm_group(2, [(HD, D, nr, [(dP[m], xin(m)[r0:r0 + nr])], gWs[m], dyn)
                           for m, (r0, nr, dyn) in enumerate(plan.modules)], HD, D, D, a16=True)
        outs = []
        for m, (r0, nr, dyn) in enumerate(plan.modules):
            gW = gWs[m]
            if not ctx.grouped:
                gemm_tn(dP[m], xin(m)[r0:r0 + nr], gW, dyn)
            outs += [gW, grads[m][0].view(params[4 * m + 1].shape), grads[m][1].view(params[4 * m + 2].shape),
                     grads[m][2].view(params[4 * m + 3].shape)]
        return (dx, None, None) + tuple(outs)


def hgat_layer(x, plan, params, drop=None):
    return HGATLayer.apply(x, plan, drop, *params)
