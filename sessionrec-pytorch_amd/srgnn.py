"""SRGNN and NISER on the HIP path - host-side mirror of
/root/reference/src/models/srgnn.py:93-148 and niser.py:91-157.

Same constructor signatures, parameter names and shapes (state_dicts interchange
with the reference / the oracle), same `forward(mg, sg=None) -> (B, num_items)`
log-probabilities.  The nn.Embedding / nn.Linear / nn.GRUCell children are
parameter containers only: all arithmetic goes through sessionrec-pytorch_amd.ops
(hand-written gfx950 kernels).  `fused_loss(mg, labels)` is the training entry the
TrainRunner uses: logits are never materialised and the dense table gradient is
written once, in place, by the scoring backward.

Reference quirk kept: the SRGNNLayer outputs are not consumed by the
readout (srgnn.py:135-142), so by default the layers are not even executed;
`use_gnn_output=True` runs them and feeds their output on (a documented extension).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _ScoringMixin:
    """Shared scoring head: fused CE for training, materialised log-probs for the forward() API."""

    def _table(self):
        return self.embedding.weight

    def _cosine(self):
        return None            # (scale, eps_mode) for cosine-scored models

    def _state(self, B):
        dev = self._table().device
        st = self.__dict__.setdefault('_srec_state', {})
        if st.get('dev') != dev:
            st.clear()
            st['dev'] = dev
            st['tgrad'] = ops.TableGrad(self._table(), defer=bool(getattr(self, '_defer_projection', False)))
            st['ws'] = {}
            st['cs'] = None
            st['cs_fresh'] = False
        if B not in st['ws']:
            V, d = self._table().shape
            st['ws'][B] = ops.CEWorkspace(B, V, d, dev)
        return st

    @property
    def table_grad(self):
        tg = self._state(1)['tgrad']
        tg.materialize()               # (a projection left to the optimizer is applied before anybody else looks)
        return tg

    def _col_scale(self, st):
        cos = self._cosine()
        if cos is None:
            return None, 1.0
        scale, eps_mode = cos
        W = self._table()
        if st['cs'] is None:
            st['cs'] = torch.empty(W.shape[0], device=W.device, dtype=torch.float32)
        if not st['cs_fresh']:
            from ._lib import lib, ptr, stream
            lib.srec_row_invnorm(ptr(W), W.stride(0), W.shape[0], W.shape[1], eps_mode, 1e-12, float(scale),
                                 ptr(st['cs']), stream())
        st['cs_fresh'] = False          # the optimizer sets it again after refreshing cs in its row pass
        return st['cs'], 1.0 / float(scale)

    def _prepare_table(self):
        """in-place, no-grad mutation of the table the reference's forward performs before reading it
        (Embedding(max_norm): lessr.py:126, msgifsr.py:162); called once per step ahead of everything that reads rows"""
        self._table_ready = False

    # FusedAdam's row pass over the table also leaves the rows renormalised (models with max_norm) and writes the bf16 operand
    # copy of the bf16 scoring kernels (optim.py: st['table_prepared']); the next forward then skips its stand-alone pass.
    # One device only: the row-sharded path refreshes its shard's copy itself (dist.HipLocal._tb).
    _fold_table_prep = True

    def _table_copy(self, st, W):
        """the TableBF16 the optimizer's row pass should fill (None: no bf16 scoring on this path).  Row-sharded table: the
        copy of this rank's LIVE rows that dist.HipLocal keeps (the scoring kernels read table[:n_live])."""
        if not ops.use_bf16_scoring(W.shape[1]) or not W.is_cuda:
            return None
        if self.shard is not None:
            local, n_live = self.shard.local, self.shard.n_live
            if not hasattr(local, '_tb') or n_live <= 0:
                return None
            live = W[:n_live] if n_live < W.shape[0] else W    # (the rows the scoring kernels read: VocabParallel._live)
            tb = local._tb(live, False)
            local._tb_written = ((live.data_ptr(), tuple(live.shape)), W._version)
            return tb
        if st.get('tb16') is None:
            st['tb16'] = ops.TableBF16(W)
        return st['tb16']

    def _take_prepared(self, st, mark=True):
        """(rows renormalised, bf16 copy written) by the optimizer's last row pass - consumed once, and only while the table
        has not been written through PyTorch since (copy_ / load_state_dict bump its version counter; the HIP kernels do not).
        mark: leave the "copy is fresh" note for the _table_bf16 call of THIS forward (_prepare_table runs ahead of it); the
        note carries the table version it was taken at, so a write through PyTorch in between voids it."""
        prep = st.pop('table_prepared', None) if st is not None else None
        if prep is None or prep[2] != self._table()._version:
            return False, False
        if mark and prep[1] and self.shard is None:
            self._tb16_fresh = self._table()._version    # consumed by _table_bf16 of this forward
        return prep[0], prep[1] and self.shard is None

    def _pop_tb16_fresh(self):
        """the note _prepare_table / _take_prepared left: valid for the table version it was written at only"""
        v = self.__dict__.pop('_tb16_fresh', None)
        return v is not None and v is not False and v == self._table()._version

    def table_written(self):
        """the table (or its optimizer state) was replaced from outside the step (load_state_dict, a restored checkpoint):
        every note about prepared rows / fresh bf16 copies is void"""
        self.__dict__.pop('_tb16_fresh', None)
        self.__dict__.pop('_table_ready', None)
        st = self.__dict__.get('_srec_state')
        if st is not None:
            st.pop('table_prepared', None)
            st['cs_fresh'] = False

    shard = None               # set by dist.VocabParallel(model): row-sharded table over the node's GPUs
    graph_capable = False      # True: every kernel of the step reads its live extents from the padded batch (hipGraph replay)

    def _lookup(self, idx, uniq, tgrad, dyn_n=None, dyn_u=None, drop=None, inv=None):
        """item rows for the batch: local gather, or the collective lookup when the table is sharded.  drop: an nn.Dropout
        applied to the rows - fused into the gather kernels on the single-device path."""
        p = drop.p if isinstance(drop, nn.Dropout) and drop.training else 0.0
        # the device step counter the dropout masks of THIS model's forward are keyed by (ops.rng_args): that of its own
        # FusedAdam, or none (another optimizer: the per-call nonce alone renews the masks)
        ops.flush_intake()          # (a captured step's batch intake that no prologue launch took along: ahead of the first read)
        W = self._table()
        c = self.__dict__.get('_srec_rng_counter')
        if c is not None and c.device == W.device:
            ops.RNG_COUNTER[str(W.device)] = c
        else:
            ops.RNG_COUNTER.pop(str(W.device), None)
        if self.shard is not None:
            fused = p > 0 and getattr(self.shard.local, 'fused_dropout', False)      # HipLocal: the mask rides in the gather
            rows = self.shard.lookup(self._table(), idx, uniq, (p, 7) if fused else None, inv=inv)
            return rows if (fused or drop is None) else drop(rows)
        if drop is not None and not isinstance(drop, nn.Dropout):
            return drop(ops.embedding_lookup(self._table(), idx, uniq, tgrad, dyn_n, dyn_u))    # replaced module (tests)
        return ops.embedding_lookup(self._table(), idx, uniq, tgrad, dyn_n, dyn_u, (p, 7) if p > 0 else None)

    def fused_loss(self, *inputs_and_labels, dynB=None):
        *inputs, labels = inputs_and_labels
        B = labels.numel()
        if dynB is None and hasattr(inputs[0], 'dynp'):
            dynB = inputs[0].dynp('B')
        st = self._state(B)
        self._prepare_table()                        # Embedding(max_norm) renorm BEFORE the cosine column scale is taken
        cs, inv_scale = self._col_scale(st)
        if self.shard is not None:
            self.shard.labels_hint = labels          # gathered together with the lookup's request lists
        # bf16 scoring on one device: the model's final normalisation writes the session vectors' bf16 operand copy itself
        self._sr_ws = st['ws'][B] if (self.shard is None and ops.use_bf16_scoring(self._table().shape[1])) else None
        # every parameter of these models feeds exactly one backward node: the split-K sums of the small grouped backward
        # launches may wait for ONE launch at the end of the backward pass (ops.defer_scope; MSGIFSR switches it itself)
        prev_defer = ops.DEFER['on']
        if not prev_defer:
            ops.drop_stale_deferred()          # (left behind by a backward pass that raised)
        ops.DEFER['on'] = bool(prev_defer or (self.training and self._table().is_cuda and getattr(self, 'defer_slab_sums', True)))
        try:
            sr = self.session_repr(*inputs, tgrad=st['tgrad'])
        finally:
            self._sr_ws = None
            ops.DEFER['on'] = prev_defer
        if self.shard is not None:
            return self.shard.loss(sr, self._table(), cs, labels, inv_scale)
        loss, _ = ops.score_ce(sr, self._table(), cs, labels.to(torch.int32), st['ws'][B], st['tgrad'], dynB, inv_scale,
                               self._table_bf16(st))
        return loss

    def _table_bf16(self, st):
        """bf16 operand copies of the table for the bf16 scoring kernels (precision 'bf16'), refreshed per step"""
        W = self._table()
        if not ops.use_bf16_scoring(W.shape[1]):
            return None
        if st.get('tb16') is None:
            st['tb16'] = ops.TableBF16(W)
        if self._pop_tb16_fresh():                       # _prepare_table wrote the copy together with the renorm
            return st['tb16']
        if self._take_prepared(st, mark=False)[1]:       # the optimizer's row pass of the previous step wrote it
            return st['tb16']
        return st['tb16'].refresh(W)

    def topk(self, *inputs, k=20):
        """(scores [B,k], item ids [B,k]) of the k best items per session WITHOUT the (B, V) score matrix
        (train.evaluate: train.py:36-55).  Models whose score mixes several soft-maxes (MSGIFSR fusion / extra)
        rank through forward()."""
        with torch.no_grad():
            sr = self.session_repr(*inputs)
            mixed = isinstance(sr, (list, tuple)) or getattr(self, 'extra', False)
            if mixed:
                v, i = self(*inputs).topk(k)          # sharded table: forward() assembles (B, V) from the column blocks
                return v, i.to(torch.int32)
            st = self._state(sr.shape[0])
            cs, _ = self._col_scale(st)
            if self.shard is not None:       # local top-k per shard -> all-gather -> merge
                return self.shard.topk(sr, self._table(), cs, k, data_parallel=self.shard.eval_data_parallel)
            return ops.score_topk(sr, self._table(), cs, k)

    def _log_probs(self, sr):
        B = sr.shape[0]
        st = self._state(B)
        cs, inv_scale = self._col_scale(st)
        if self.shard is not None:           # evaluation / compat only (no gradient through the sharded (B, V) matrix)
            return self.shard.log_probs(sr, self._table(), cs, data_parallel=self.shard.eval_data_parallel)
        return ops.score_logp(sr, self._table(), cs, st['ws'][B], inv_scale)


class AttnReadout(nn.Module):
    """srgnn.py:53-91 / niser.py:51-89 (batch_norm=None there).  fc_u/fc_v run as MFMA GEMMs over all
    nodes of the batch; sigmoid / fc_e / per-session softmax / weighted sum is one wave per session."""

    def __init__(self, input_dim, hidden_dim, output_dim, batch_norm=None, feat_drop=0.0, activation=None):
        super().__init__()
        assert not batch_norm and output_dim == input_dim and activation is None
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_u = nn.Linear(input_dim, hidden_dim, bias=False)
        self.fc_v = nn.Linear(input_dim, hidden_dim, bias=True)
        self.fc_e = nn.Linear(hidden_dim, 1, bias=False)

    def forward(self, mg, feat):
        dN, dB = mg.dynp('N'), mg.dynp('B')
        feat = self.feat_drop(feat)
        U = ops.linear(feat, self.fc_u.weight, None, dN, exact=True)
        Vq = ops.linear(ops.row_gather(feat, mg.last, dB, ascending=True), self.fc_v.weight, self.fc_v.bias, dB, exact=True)
        return ops.seg_attn(U, Vq, self.fc_e.weight, feat, mg.seg, dB)


class SRGNNLayer(nn.Module):
    """Parameter container for srgnn.py:11-51 (kernel: ops.srgnn_layer, see gnn.py)."""

    def __init__(self, input_dim, output_dim, feat_drop=0.0):
        super().__init__()
        self.dropout = nn.Dropout(feat_drop)
        self.gru = nn.GRUCell(2 * input_dim, output_dim)
        self.W1 = nn.Linear(input_dim, output_dim, bias=False)
        self.W2 = nn.Linear(input_dim, output_dim, bias=False)

    def forward(self, mg, feat):
        from . import gnn
        return gnn.srgnn_layer(self, mg, feat)


class SRGNN(_ScoringMixin, nn.Module):
    graph_capable = True

    def __init__(self, num_items, embedding_dim, num_layers, feat_drop=0.0, use_gnn_output=False):
        super().__init__()
        self.embedding = nn.Embedding(num_items, embedding_dim)
        self.register_buffer('indices', torch.arange(num_items, dtype=torch.long))
        self.embedding_dim = embedding_dim
        self.num_layers = num_layers
        self.use_gnn_output = use_gnn_output
        self.layers = nn.ModuleList([SRGNNLayer(embedding_dim, embedding_dim, feat_drop) for _ in range(num_layers)])
        self.readout = AttnReadout(embedding_dim, embedding_dim, embedding_dim, feat_drop=feat_drop)
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_sr = nn.Linear(2 * embedding_dim, embedding_dim, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.embedding_dim)
        for w in self.parameters():
            w.data.uniform_(-stdv, stdv)

    def _pre(self, feat, dyn=None):
        return feat

    def _post(self, sr, dyn=None):
        return sr

    def session_repr(self, mg, sg=None, tgrad=None):
        dN, dB = mg.dynp('N'), mg.dynp('B')
        if mg.buf.is_cuda:
            ops.check_limits(mg)
        feat = self._lookup(mg.iid, (mg.uniq_items, mg.uniq_ptr, mg.uniq_pos, mg.uniq_cptr, mg.chunk_ptr), tgrad,
                            dN, mg.dynp('U'), inv=mg.uniq_inv if mg.has('uniq_inv') else None)
        feat = self._pre(self.feat_drop(feat), dN)
        if self.use_gnn_output:
            for layer in self.layers:
                feat = layer(mg, feat)
        sr_l = ops.row_gather(feat, mg.last, dB, ascending=True)
        ro = self.readout
        if feat.is_cuda and not (self.training and ro.feat_drop.p > 0):
            # read-out + fc_sr as grouped exact-fp32 launches (ops.ReadoutHead, as in MSGIFSR).  fc_v's bias rides on the U
            # product instead: sigmoid(U + (Vq + b)) = sigmoid((U + b) + Vq), and d b = sum_n dU = sum_b dVq.  (With read-out
            # dropout the fc_v input is the DROPPED last-node row while fc_sr takes the clean one: separate ops below.)
            (s,) = ops.readout_head(feat, mg.seg, dN, dB, [(sr_l, ro.fc_u.weight, ro.fc_v.bias, ro.fc_v.weight, ro.fc_e.weight,
                                                            self.fc_sr.weight)])
            return self._post(s, dB)
        sr_g = ro(mg, feat)
        return self._post(ops.linear_cat([sr_l, sr_g], self.fc_sr.weight, None, dB, exact=True), dB)

    def forward(self, mg, sg=None):
        return self._log_probs(self.session_repr(mg))


class NISER(SRGNN):
    """niser.py:91-157: L2-normalised item / session vectors, logits scaled by `scale`."""

    def __init__(self, num_items, embedding_dim, num_layers, feat_drop=0.0, norm=True, scale=12,
                 use_gnn_output=False):
        self.norm, self.scale = norm, scale
        super().__init__(num_items, embedding_dim, num_layers, feat_drop, use_gnn_output)

    def _cosine(self):
        if self.norm:
            return (self.scale if self.scale else 1.0, 1)
        return None

    def _col_scale(self, st):
        if self.norm or not self.scale:
            return super()._col_scale(st)
        W = self._table()                 # un-normalised but scaled logits
        if st['cs'] is None or st['cs'].numel() != W.shape[0]:
            st['cs'] = torch.full((W.shape[0],), float(self.scale), device=W.device)
        return st['cs'], 0.0

    def _pre(self, feat, dyn=None):
        # niser.py:135 and :142 normalise twice; the second one divides a unit vector by its norm
        # (identity up to 1 ulp, and its Jacobian is the same projection): applied once here.
        return ops.normalize(feat, 1, dyn) if self.norm else feat

    def _post(self, sr, dyn=None):
        return ops.normalize(sr, 1, dyn, getattr(self, '_sr_ws', None)) if self.norm else sr
