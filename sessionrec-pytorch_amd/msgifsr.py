"""MSGIFSR on the HIP path - host-side mirror of /root/reference/src/models/msgifsr.py:157-323
(SemanticExpander :14-45, MSHGNN :47-91, AttnReadout :94-155) and gnn_models/gatconv.py:136-319.

Same constructor signature and parameter names / shapes as the reference (state_dicts
interchange), `forward(mg) -> (B, num_items)` log-probabilities, plus `fused_loss`.
All arithmetic runs in hand-written gfx950 kernels (sessionrec-pytorch_amd.ops); the nn.*
children only hold parameters.

Documented deviations:
  * GATConv is run with zero-in-degree destinations zero-filled (the shipped module would raise);
  * one fc projection per (GAT module, node type) is shared by every relation that uses it as source or destination -
    identical to the reference whenever feat_drop == 0; with dropout (the scripts' default 0.1) the reference draws an
    independent feature mask per (relation, role), here one mask per (conv, node type) - projection, attention logits
    and the identity residual of that conv's modules see the same dropped rows; attention dropout is applied to the
    edge soft-max inside the batched kernels (mask per (relation instance, edge, head));
  * `fusion` (order mixture, msgifsr.py:311-317) is: K
    fused scoring passes, the mixture on the (lse, label-logit) pairs; `extra` (repeat / explore mixture,
    msgifsr.py:281-305) never materialises the two masked (B, V) soft-maxes: the in-session log-sum-exp is a
    [B, <=L] dot product against the session's own (already gathered) item rows and the out-of-session one follows
    from the fused full-catalog statistics, lse_ex = lse_all + log(1 - exp(lse_in - lse_all)).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .srgnn import _ScoringMixin


class GATConv(nn.Module):
    """Parameter container for gatconv.py:136-197 (in_feats int, residual=True -> Identity, bias)."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0., negative_slope=0.2):
        super().__init__()
        assert in_feats == out_feats
        self._num_heads, self._out_feats = num_heads, out_feats
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.bias = nn.Parameter(torch.empty(num_heads * out_feats))
        self.negative_slope = negative_slope
        self.feat_drop, self.attn_drop = feat_drop, attn_drop


class HeteroConv(nn.Module):
    """dglnn.HeteroGraphConv container: modules keyed by edge-type name."""

    def __init__(self, mods):
        super().__init__()
        self.mods = nn.ModuleDict(mods)


class SemanticExpander(nn.Module):
    def __init__(self, input_dim, reducer, order):
        super().__init__()
        self.input_dim, self.order, self.reducer = input_dim, order, reducer
        self.GRUs = nn.ModuleList([nn.GRU(input_dim, input_dim, 1, True, True) for _ in range(order)])
        if reducer == 'concat':
            self.Ws = nn.ModuleList([nn.Linear(input_dim * (i + 1), input_dim) for i in range(1, order)])

    def forward(self, x, k, dyn=None, dyn_rows=None):
        """x: [N_k * k, d] gathered gram rows (contiguous) -> [N_k, d]"""
        if self.reducer == 'mean':
            return ops.gru_expand(x, self.GRUs[k - 2], k, dyn, dyn_rows)          # 0.5 mean_t x + 0.5 h_last, one node
        var = ops.gru_expand(x, self.GRUs[k - 2], k, dyn, dyn_rows, combine=False)
        n, d = x.shape[0] // k, x.shape[1]
        if self.reducer == 'max':
            invar = x.view(n, k, d).max(dim=1)[0]                                  # padded rows: max of zeros = 0
        elif self.reducer == 'concat':
            W = self.Ws[k - 2]
            invar = ops.linear(x.view(n, k * d), W.weight, W.bias, dyn)
        else:
            raise ValueError('unknown reducer %r' % (self.reducer,))
        return 0.5 * invar + 0.5 * var


class MSHGNN(nn.Module):
    def __init__(self, input_dim, output_dim, dropout=0.0, activation=None, order=1, reducer='mean'):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.output_dim, self.activation, self.order = output_dim, activation, order
        mk = lambda: GATConv(input_dim, output_dim, 8, dropout, dropout)
        m1 = {'intra%d' % (i + 1): mk() for i in range(order)}
        m1['inter'] = mk()
        self.conv1 = HeteroConv(m1)
        m2 = {'intra%d' % (i + 1): mk() for i in range(order)}
        m2['inter'] = mk()
        self.conv2 = HeteroConv(m2)
        self.lint = nn.Linear(output_dim, 1, bias=False)
        self.linq = nn.Linear(output_dim, output_dim)
        self.link = nn.Linear(output_dim, output_dim, bias=False)

    def _graph(self, mg, name, reverse):
        f = mg.field
        if not reverse:
            return (f(name + '_in_ptr'), f(name + '_in_idx'), f(name + '_out_ptr'), f(name + '_out_idx'),
                    f(name + '_src'), f(name + '_dst'))
        return (f(name + '_out_ptr'), f(name + '_out_idx'), f(name + '_in_ptr'), f(name + '_in_idx'),
                f(name + '_dst'), f(name + '_src'))

    def plan(self, mg, D, all_rels=False):
        """static topology of this layer for one FlatBatch (ops.HgPlan): types, modules, projection blocks, instances.
        all_rels: treat every relation of the schema as live, also one without edges in THIS batch.  HeteroGraphConv
        skips a relation that has no edges in the (whole) batch, and a relation that is not skipped gives EVERY
        destination node its identity residual and bias, edges or not (gatconv.py:306-311) - a session's result depends
        on whether some other session of the batch has such an edge.  A rank of a multi-GPU job encodes a slice of the
        global batch, so "live" must mean live in the GLOBAL batch: any training batch has edges of every relation
        somewhere (one session of order + 1 clicks suffices), hence all relations, with no per-step exchange of flags."""
        K = self.order
        types, r = [], 0
        for k in range(1, K + 1):
            nk = mg.meta['ncap'][k]
            types.append((r, nk, mg.dynp('N%d' % k), mg.field('seg%d' % k)))
            r += nk
        NT = r
        mods, mod_id, blocks, blk_id, insts, params, mod_conv = [], {}, [], {}, [], [], []
        used = {}                                            # (conv index, etype) -> set of node types it touches
        live = [((s, et, d_), name) for (s, et, d_), name in mg.meta['rels'] if all_rels or mg.count('E_' + name) > 0]
        for ci in (0, 1):
            for (s, et, d_), name in live:
                used.setdefault((ci, et), set()).update((s, d_))
        for ci, conv in enumerate((self.conv1, self.conv2)):
            for (s, et, d_), name in live:
                key = (ci, et)
                if key not in mod_id:
                    ks = sorted(used[key])
                    if len(ks) == 1:                         # intra_k: only the rows of type k
                        r0, nr, dyn = types[ks[0] - 1][0], types[ks[0] - 1][1], types[ks[0] - 1][2]
                    else:                                    # shared 'inter' module: project every row once
                        r0, nr, dyn = 0, NT, None
                    mod_id[key] = len(mods)
                    mods.append((r0, nr, dyn))
                    mod_conv.append(ci)
                    mod = conv.mods[et]
                    params += [mod.fc.weight, mod.attn_l, mod.attn_r, mod.bias]
                m = mod_id[key]
                src_t, dst_t = (d_, s) if ci == 1 else (s, d_)
                for k in (src_t, dst_t):
                    if (m, k) not in blk_id:
                        blk_id[(m, k)] = len(blocks)
                        blocks.append((m, k - 1))
                insts.append((m, blk_id[(m, src_t)], blk_id[(m, dst_t)], self._graph(mg, name, ci == 1)))
        slope = self.conv1.mods['intra1'].negative_slope
        plan = ops.HgPlan(8, D, slope, mg.B, mg.dynp('B'), types, mods, blocks, insts, mod_conv)
        plan.layer_id = getattr(self, '_layer_id', 0)
        # host-side live node counts of THIS batch per type (first stacked row -> count): hints for the GEMM tile heuristics -
        # a captured step keeps the choice of the batch it was captured on, the kernels themselves read the device counts
        plan.live = {t[0]: mg.count('N%d' % (k + 1)) for k, t in enumerate(types)}
        return plan, params

    def forward_stacked(self, mg, x, all_rels=False):
        """x: [NT, d] node features of all orders stacked (order-1 rows first) -> [NT, d]; one batched pass"""
        pre = self.__dict__.pop('_pre', None)        # the plan the model's prologue launch already folded the weights for
        if pre is not None and pre[0] is mg and pre[1] == (x.shape[1], bool(all_rels)):
            plan, params = pre[2]
        else:
            plan, params = self.plan(mg, x.shape[1], all_rels)
        mod = self.conv1.mods['intra1']
        drop = (mod.feat_drop, mod.attn_drop) if self.training and (mod.feat_drop > 0 or mod.attn_drop > 0) else None
        return ops.hgat_layer(x, plan, params, drop)

    def forward(self, mg, feat):
        """feat: {k: [N_k, d]} -> {k: [N_k, d]}"""
        H = 8
        K = self.order
        outs = {k: [] for k in range(1, K + 1)}
        biases = {k: [] for k in range(1, K + 1)}
        dN = {k: mg.dynp('N%d' % k) for k in range(1, K + 1)}
        for conv, reverse in ((self.conv1, False), (self.conv2, True)):
            proj = {}                                   # (module name, type) -> fc(dropout(feat))
            dropped = {}

            def fc(et, k):
                key = (et, k)
                if key not in proj:
                    if k not in dropped:
                        dropped[k] = F.dropout(feat[k], conv.mods[et].feat_drop, self.training)
                    proj[key] = ops.linear(dropped[k], conv.mods[et].fc.weight, None, dN[k])
                return proj[key]
            for (s, et, d_), name in mg.meta['rels']:
                if mg.count('E_' + name) == 0:
                    continue                            # HeteroGraphConv skips relations without edges
                src_t, dst_t = (d_, s) if reverse else (s, d_)
                mod = conv.mods[et]
                rst = ops.gat_relation(fc(et, src_t), fc(et, dst_t), mod.attn_l, mod.attn_r,
                                       self._graph(mg, name, reverse), H, dN[src_t], dN[dst_t], mod.negative_slope)
                if mod.attn_drop > 0 and self.training:
                    raise NotImplementedError('attention dropout inside the fused GAT kernel')
                outs[dst_t].append(rst)
                biases[dst_t].append(mod.bias)
        h = {}
        for k in range(1, K + 1):
            nres = len(outs[k])
            if nres:
                bias = biases[k][0] if nres == 1 else torch.stack(biases[k], 0).sum(0)
            else:
                bias = torch.zeros(H * self.output_dim, device=feat[k].device)
            x = ops.head_combine(feat[k], bias, float(nres), H, outs[k], dN[k])
            h[k] = ops.seg_mean_add(x, feat[k], mg.field('seg%d' % k), mg.B, mg.dynp('B'))
        return h


class AttnReadout(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, feat_drop=0.0, activation=None, order=1,
                 device=torch.device('cpu')):
        super().__init__()
        assert output_dim == input_dim and activation is None
        self.feat_drop = nn.Dropout(feat_drop)
        self.order = order
        self.fc_u = nn.ModuleList([nn.Linear(input_dim, hidden_dim, bias=True) for _ in range(order)])
        self.fc_v = nn.ModuleList([nn.Linear(input_dim, hidden_dim, bias=False) for _ in range(order)])
        self.fc_e = nn.ModuleList([nn.Linear(hidden_dim, 1, bias=False) for _ in range(order)])
        self.fc_p = nn.ModuleList()

    def forward(self, mg, allf, feat_vs, orders):
        """allf: [NT, d] per-session concatenation of all orders' nodes; feat_vs[i]: [B, d]."""
        out = {}
        dT, dB = mg.dynp('NT'), mg.dynp('B')
        for i in orders:
            U = ops.linear(allf, self.fc_u[i].weight, self.fc_u[i].bias, dT, exact=True)
            Vq = ops.linear(feat_vs[i], self.fc_v[i].weight, None, dB, exact=True)
            out[i] = ops.seg_attn(U, Vq, self.fc_e[i].weight, allf, mg.cat_seg, dB)
        return out


class MSGIFSR(_ScoringMixin, nn.Module):
    @property
    def graph_capable(self):
        """the `extra` epilogue sizes a [B, longest session] view from a per-batch host value: eager only"""
        return not self.extra

    def __init__(self, num_items, datasets, embedding_dim, num_layers, dropout=0.0, reducer='mean', order=3,
                 norm=True, extra=True, fusion=True, device=torch.device('cpu')):
        super().__init__()
        self.embeddings = nn.Embedding(num_items, embedding_dim, max_norm=1)
        self.num_items = num_items
        self.register_buffer('indices', torch.arange(num_items, dtype=torch.long))
        self.embedding_dim, self.num_layers, self.reducer, self.order = embedding_dim, num_layers, reducer, order
        self.alpha = nn.Parameter(torch.Tensor(order))
        self.beta = nn.Parameter(torch.Tensor(1))
        self.norm = norm
        self.expander = SemanticExpander(embedding_dim, reducer, order)
        self.device = device
        self.layers = nn.ModuleList([
            MSHGNN(embedding_dim, embedding_dim, dropout=dropout, order=order, activation=nn.PReLU(embedding_dim))
            for _ in range(num_layers)])
        for li, layer in enumerate(self.layers):
            layer._layer_id = li
        self.readout = AttnReadout(embedding_dim, embedding_dim, embedding_dim, feat_drop=dropout, order=order)
        self.feat_drop = nn.Dropout(dropout)
        self.fc_sr = nn.ModuleList([nn.Linear(2 * embedding_dim, embedding_dim, bias=False) for _ in range(order)])
        self.sc_sr = nn.ModuleList([
            nn.Sequential(nn.Linear(embedding_dim, embedding_dim, bias=True), nn.ReLU(),
                          nn.Linear(embedding_dim, 2, bias=False), nn.Softmax(dim=-1)) for _ in range(order)])
        self.reset_parameters()
        self.alpha.data = torch.zeros(order)
        self.alpha.data[0] = torch.tensor(1.0)
        self.beta.data = torch.tensor(1.0)
        self.fusion, self.extra = fusion, extra
        self._max_norm = 1.0

    @staticmethod
    def grad_bucket(name):
        """bucket of a replicated parameter's gradient in backward completion order (dist.VocabParallel): 0 = read-out head and
        mixture weights (complete first), 1 = the MSHGNN layers, 2 = the k-gram expander and everything else (complete last)"""
        if name.startswith(('readout.', 'fc_sr.', 'sc_sr.', 'alpha', 'beta')):
            return 0
        return 1 if name.startswith('layers.') else 2

    def reset_parameters(self):
        stdv = 1 / math.sqrt(self.embedding_dim)
        for w in self.parameters():
            w.data.uniform_(-stdv, stdv)

    # ---- scoring head hooks
    def _table(self):
        return self.embeddings.weight

    def _cosine(self):
        return (12.0, 0) if self.norm else None

    def _col_scale(self, st):
        if self.norm:
            return super()._col_scale(st)
        W = self._table()
        if st['cs'] is None:
            st['cs'] = torch.full((W.shape[0],), 12.0, device=W.device)
        return st['cs'], 0.0

    def zero_grad_params(self):
        """Parameters the reference reaches only through `score[:, 0]` (order > 1, no fusion): autograd
        gives them ZERO (not None) gradients there, so Adam still applies weight decay to them."""
        if self.order > 1 and not self.fusion:
            ps = []
            for i in range(1, self.order):
                ps += list(self.readout.fc_u[i].parameters()) + list(self.readout.fc_v[i].parameters())
                ps += list(self.readout.fc_e[i].parameters()) + list(self.fc_sr[i].parameters())
            return ps
        return []

    def _renorm(self, mg=None):
        """Embedding(max_norm=1): rows are renormalised in place (no grad) before they are read."""
        from ._lib import lib, ptr, stream
        W = self._table()
        with torch.no_grad():
            lib.srec_renorm_rows(ptr(W), W.stride(0), None, W.shape[0], None, W.shape[1], 1.0, stream())

    def _prepare_table(self):
        # training entry: the renorm runs here, ahead of the cosine column scale (the scale must be that of the rows
        # the scoring kernels read: msgifsr.py:162 renormalises inside the lookup, :276-279 normalises the result)
        W = self._table()
        st = self._state(1) if W.is_cuda else None
        renormed, copied = self._take_prepared(st)        # done by the optimizer's row pass of the previous step?
        if self.shard is None and self.training and ops.use_bf16_scoring(W.shape[1]) and W.is_cuda:
            # bf16 scoring: the renorm and the table's bf16 operand copy in one pass over the rows
            if st.get('tb16') is None:
                st['tb16'] = ops.TableBF16(W)
                copied = False
            if not (renormed and copied):
                st['tb16'].refresh(W, 1.0)
            self._tb16_fresh = W._version
        elif not renormed:
            self._renorm()
        self._table_ready = True

    def session_repr(self, mg, tgrad=None):
        # every parameter of this model feeds exactly one backward node per layer / order: the slab sums that finish their
        # gradients may wait for ONE launch at the end of the backward pass.  The permission holds for the nodes created by
        # THIS forward only (they snapshot it: ops.defer_scope) and is withdrawn when the forward returns
        ops.DEFER['on'] = bool(mg.buf.is_cuda and self.training)
        try:
            return self._session_repr(mg, tgrad)
        finally:
            ops.DEFER['on'] = False

    def _step_prologue(self, mg):
        """the operand copies of this step's weights (k-gram GRUs, the first MSHGNN layer's fc weights and its folded attention
        vectors, the read-out head) and the batch intake of a captured step in ONE launch ahead of the lookup (ops.step_prologue,
        csrc/prep.hip) - 5 launches of 5 - 8 us each otherwise, one in front of each reader.  The conditions mirror the
        readers'; a reader that finds no copies makes its own."""
        K, d = self.order, self.embedding_dim
        bf16 = ops.PRECISION['matmul'] == 'bf16'
        grad = torch.is_grad_enabled() and self.training
        gru, w16, head, fold = [], [], [], None
        if bf16 and 1 < K <= 5 and ops.gru_expand_fast_ok(d, self.reducer) and ops.gru_fused_ok(d, K - 1):
            gru = [w for k in range(2, K + 1) for w in (self.expander.GRUs[k - 2].weight_ih_l0, self.expander.GRUs[k - 2].weight_hh_l0)]
            if not all(w.is_contiguous() for w in gru):
                gru = []
        if len(self.layers) > 0:
            layer = self.layers[0]
            multi = self.shard is not None and self.shard.world > 1
            plan, params = layer.plan(mg, d, multi)
            layer._pre = (mg, (d, bool(multi)), (plan, params))
            nm = len(plan.modules)
            fold = (plan, params) if nm > 0 else None        # (a batch without a live relation: nothing to fold)
            if bf16 and d % 64 == 0 and nm <= 8 and all(params[4 * m].is_contiguous() for m in range(nm)):
                w16 = [params[4 * m] for m in range(nm)]
        live = range(K) if (K == 1 or self.fusion) else (0,)
        if (bf16 and ops.FUSED_HEAD and self.norm and K > 1 and len(live) <= 4 and d in (128, 256) and 1 < mg.B <= ops.HEAD_FUSED_MAX_B):
            ro = self.readout
            hw = [w for i in live for w in (ro.fc_u[i].weight, ro.fc_v[i].weight, self.fc_sr[i].weight)]
            if all(w.is_contiguous() for w in hw):
                head = [(w, 0) for w in hw] + ([(self.fc_sr[i].weight, 1) for i in live] if grad else [])
        ops.step_prologue(w16, gru, head, fold)

    def _session_repr(self, mg, tgrad=None):
        K = self.order
        if mg.buf.is_cuda:
            ops.check_limits(mg)
            if ops.STEP_PROLOGUE:
                self._step_prologue(mg)
        if not self.__dict__.pop('_table_ready', False):
            self._renorm(mg)
        W = self._table()
        d = self.embedding_dim
        rows = self._lookup(mg.gidx, (mg.uniq_items, mg.uniq_ptr, mg.uniq_pos, mg.uniq_cptr, mg.chunk_ptr), tgrad,
                            None, mg.dynp('U'), drop=self.feat_drop,
                            inv=mg.uniq_inv if mg.has('uniq_inv') else None)     # padded slots of gidx are -1 -> zero rows; the
        #                                                                  feature dropout of msgifsr.py:247 rides in the gather
        feats = {}
        dB = mg.dynp('B')
        ncap = mg.meta['ncap']                             # block size of order k (capacity in padded layouts)
        pieces = ops.split_rows(rows, [ncap[k] * k for k in range(1, K + 1)])
        fast = None
        if K > 1 and K <= 5 and ops.gru_expand_fast_ok(d, self.reducer):
            # bf16 path: every order's k-gram GRU in one autograd node, one launch per time step for all orders
            fast = ops.gru_expand_all([pieces[k - 1] for k in range(2, K + 1)], [self.expander.GRUs[k - 2] for k in range(2, K + 1)],
                                      list(range(2, K + 1)), [mg.dynp('N%d' % k) for k in range(2, K + 1)],
                                      [mg.dynp('GK%d' % k) for k in range(2, K + 1)])
        raw = {}
        for k in range(1, K + 1):
            x = pieces[k - 1]
            dk = mg.dynp('N%d' % k)
            raw[k] = x if k == 1 else (fast[k - 2] if fast is not None else self.expander(x, k, dk, mg.dynp('GK%d' % k)))
        stacked0 = None
        if self.norm and 1 < K <= 4 and len(self.layers) > 0:
            # all orders normalised straight into the stacked matrix the batched layer reads: one launch, no concatenation
            stacked0 = ops.normalize_stack([raw[k] for k in range(1, K + 1)], 0, [mg.dynp('N%d' % k) for k in range(1, K + 1)])
            o = 0
            for k in range(1, K + 1):
                feats[k] = stacked0[o:o + ncap[k]]
                o += ncap[k]
        else:
            for k in range(1, K + 1):
                feats[k] = ops.normalize(raw[k], 0, mg.dynp('N%d' % k)) if self.norm else raw[k]
        # the session's own item rows (normalised): `extra` in-session logits.  Kept ONLY for `extra`: a tensor stored on the
        # module keeps the autograd graph of the last forward alive, and with it the AccumulateGrad nodes of every parameter
        # upstream (the k-gram GRUs) together with the stream they were created on - an eager step on the default stream
        # followed by a hipGraph capture then makes the engine synchronise the capturing stream with the default stream and
        # hipStreamEndCapture crashes (found by tests/test_dist_gpu.py's replayed rank)
        self._s1_feat = feats[1] if self.extra else None
        if len(self.layers) > 0:
            # all orders stacked once; every layer is one batched pass over all relations (ops.hgat_layer)
            stacked = stacked0 if stacked0 is not None else (
                feats[1] if K == 1 else torch.cat([feats[k] for k in range(1, K + 1)], 0))
            multi = self.shard is not None and self.shard.world > 1       # (see MSHGNN.plan: live = live in the GLOBAL batch)
            # row-sharded training: the replicated gradients are all-reduced in buckets as the backward completes them
            # (dist.VocabParallel.bucket_ready): the GAT layers' when it passes below them, the read-out head's above them
            early = getattr(self.shard, 'bucket_ready', None) if (self.shard is not None and self.training) else None
            if early is not None:
                stacked = ops.grad_mark(stacked, lambda: early(1))
            for layer in self.layers:
                stacked = layer.forward_stacked(mg, stacked, multi)
            if early is not None:
                stacked = ops.grad_mark(stacked, lambda: early(0))
            fuse_npp = self.norm and 1 < K <= 4 and stacked.is_cuda
            if self.norm and not fuse_npp:
                stacked = ops.normalize(stacked, 0, mg.dynp('N1') if K == 1 else None)   # padded rows are exact zeros
        else:
            h = feats
            for layer in self.layers:
                h = layer(mg, h)
            if self.norm:
                h = {k: ops.normalize(v, 0, mg.dynp('N%d' % k)) for k, v in h.items()}
            stacked = h[1] if K == 1 else torch.cat([h[k] for k in range(1, K + 1)], 0)
        live = range(K) if (K == 1 or self.fusion) else (0,)
        if K == 1:
            allf = stacked
            feat_vs = {i: ops.row_gather(stacked, mg.field('lastcat%d' % (i + 1)), dB) for i in live}
        elif len(self.layers) > 0 and fuse_npp:
            # normalisation + per-session concatenation + last-node picks in one launch (ops.NormPermutePick)
            types, r0 = [], 0
            for k in range(1, K + 1):
                types.append((r0, ncap[k], mg.dynp('N%d' % k)))
                r0 += ncap[k]
            allf, picked = ops.norm_permute_pick(stacked, mg.cat_perm, mg.cat_seg, [mg.field('lastcat%d' % (i + 1)) for i in live],
                                                 types, mg.dynp('NT'), dB, 0)
            feat_vs = dict(zip(live, picked))
        else:
            allf, picked = ops.permute_and_pick(stacked, mg.cat_perm, mg.field('cat_inv'),
                                                [mg.field('lastcat%d' % (i + 1)) for i in live], mg.dynp('NT'), dB)
            feat_vs = dict(zip(live, picked))
        srs = []
        if len(live) <= 4:
            # read-out + fc_sr(cat[x_last, sr_g]) of every live order as grouped exact-fp32 launches (ops.ReadoutHead):
            # these B-row products are launch bound one by one
            ro = self.readout
            heads = [(feat_vs[i], ro.fc_u[i].weight, ro.fc_u[i].bias, ro.fc_v[i].weight, ro.fc_e[i].weight,
                      self.fc_sr[i].weight) for i in live]
            ws = getattr(self, '_sr_ws', None) if len(heads) == 1 else None   # one head: its vector is the scoring operand
            if self.norm and ops.readout_head_fused_ok(allf, heads):
                # bf16 mode, d = 128 / 256: Vq, U, soft-max read-out, fc_sr and the normalisation as ONE launch (csrc/headf.hip)
                srs = list(ops.readout_head_fused(allf, mg.cat_seg, mg.dynp('NT'), dB, heads, ws))
            else:
                ss = ops.readout_head(allf, mg.cat_seg, mg.dynp('NT'), dB, heads)
                srs = [ops.normalize(s, 0, dB, ws) if self.norm else s for s in ss]
        else:
            sr_g = self.readout(mg, allf, feat_vs, live)
            for i in live:
                s = ops.linear(ops.cat_cols(feat_vs[i], sr_g[i]), self.fc_sr[i].weight, None, dB, exact=True)
                srs.append(ops.normalize(s, 0, dB) if self.norm else s)
        if self.fusion and K > 1:
            return srs                                     # one session vector per order (IFR mixture)
        return srs[0]

    # ---- `extra`: repeat / explore mixture (msgifsr.py:281-305)
    def _in_session(self, mg, labels=None):
        """dense [B, L] view of each session's order-1 nodes: node positions, validity, item ids, label membership"""
        seg = mg.field('seg1').long()
        f1 = self._s1_feat
        B, L = mg.B, int(mg.meta['max_nodes'])
        pos = seg[:-1, None] + torch.arange(L, device=f1.device)[None, :]
        valid = pos < seg[1:, None]
        posc = pos.clamp(max=f1.shape[0] - 1)
        items = mg.field('iid1').long()[posc]
        hit = None if labels is None else ((items == labels.long()[:, None]) & valid).any(1)
        return posc, valid, items, hit

    def _log_phi(self, sr, dynB):
        """log softmax of sc_sr[0](sr): the (repeat, explore) gate - sc_sr[0] serves every order (msgifsr.py:283)"""
        sc = self.sc_sr[0]
        hdn = torch.relu(ops.linear(sr, sc[0].weight, sc[0].bias, dynB, exact=True))
        two = (hdn.unsqueeze(1) * sc[2].weight.unsqueeze(0)).sum(-1)       # [B, 2]: too narrow for an MFMA tile
        return torch.log_softmax(two, dim=-1)

    def _extra_label_logprob(self, sr, lse_all, zlab, posc, valid, hit, dynB):
        zin = 12.0 * (self._s1_feat[posc] * sr[:, None, :]).sum(-1)
        lse_in = torch.logsumexp(zin.masked_fill(~valid, float('-inf')), dim=1)
        lse_ex = lse_all + torch.log1p(-torch.exp(lse_in - lse_all).clamp(max=1.0 - 1e-7))
        lphi = self._log_phi(sr, dynB)
        return torch.where(hit, lphi[:, 0] + zlab - lse_in, lphi[:, 1] + zlab - lse_ex)

    def _extra_log_probs(self, sr, posc, valid, items):
        """(B, V) log score of one order with the repeat / explore gate (compat / evaluation path)"""
        logp = self._log_probs(sr)                          # z - lse_all
        B, V = logp.shape
        lin = logp.gather(1, items.clamp(min=0)).masked_fill(~valid, float('-inf'))
        d_in = torch.logsumexp(lin, dim=1)                  # lse_in - lse_all
        d_ex = torch.log1p(-torch.exp(d_in).clamp(max=1.0 - 1e-7))
        mask = torch.zeros(B, V + 1, dtype=torch.bool, device=logp.device)
        mask.scatter_(1, torch.where(valid, items, torch.full_like(items, V)), True)
        lphi = self._log_phi(sr, None)
        return torch.where(mask[:, :V], lphi[:, 0:1] + logp - d_in[:, None], lphi[:, 1:2] + logp - d_ex[:, None])

    def fused_loss(self, *inputs_and_labels, dynB=None):
        if not (self.extra or (self.fusion and self.order > 1)):
            return super().fused_loss(*inputs_and_labels, dynB=dynB)
        # msgifsr.py:311-321: score = sum_k softmax(alpha)_k score_k; loss = -mean log score[label]
        mg, labels = inputs_and_labels
        B = labels.numel()
        if dynB is None:
            dynB = mg.dynp('B')
        st = self._state(B)
        self._prepare_table()
        cs, inv_scale = self._col_scale(st)
        srs = self.session_repr(mg, tgrad=st['tgrad'])
        if not isinstance(srs, (list, tuple)):
            srs = [srs]
        lab32 = labels.to(torch.int32)
        st['tgrad'].fresh = False
        sharded = self.shard is not None
        tb = None if sharded else self._table_bf16(st)
        if self.extra:
            posc, valid, _, hit = self._in_session(mg, labels)
        logits = []
        for sr in srs:
            if sharded:      # global (lse, label logit) of this rank's sessions; heads accumulate into the shard's dE
                lse, lab = self.shard.stats(sr, self._table(), cs, labels, inv_scale)
            else:
                lse, lab = ops.score_stats(sr, self._table(), cs, lab32, st['ws'][B], st['tgrad'], dynB, inv_scale, tb)
            if self.extra:
                logits.append(self._extra_label_logprob(sr, lse, lab, posc, valid, hit, dynB))
            else:
                logits.append(lab - lse)                   # log softmax_k[label]
        if len(logits) > 1:
            logp = torch.logsumexp(torch.stack(logits, 1) + torch.log_softmax(self.alpha, 0).unsqueeze(0), dim=1)
        else:
            logp = logits[0]
        if sharded:          # this rank's share of the mean over the GLOBAL batch (replicated grads are summed over ranks)
            live = labels >= 0                                  # label -1: capacity padding of a rank's partial batch
            n_live = live.sum().to(logp.dtype)
            from .dist import all_reduce_sum
            n_live = all_reduce_sum(n_live.detach().clone(), self.shard.group)
            return -torch.where(live, logp, torch.zeros_like(logp)).sum() / n_live.clamp(min=1)
        if dynB is not None:
            live = torch.arange(B, device=logp.device) < dynB
            return -torch.where(live, logp, torch.zeros_like(logp)).sum() / dynB.to(logp.dtype).clamp(min=1).sum()
        return -logp.mean()

    def forward(self, mg):
        sr = self.session_repr(mg)
        if self.extra:
            posc, valid, items, _ = self._in_session(mg)
            per_order = (lambda s: self._extra_log_probs(s, posc, valid, items))
        else:
            per_order = self._log_probs
        if self.fusion and self.order > 1:
            la = torch.log_softmax(self.alpha, 0)
            return torch.logsumexp(torch.stack([per_order(s) + la[k] for k, s in enumerate(sr)], 0), dim=0)
        return per_order(sr)
