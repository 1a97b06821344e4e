"""LESSR on the HIP path - host-side mirror of /root/reference/src/models/lessr.py:121-183
(EOPA :8-42, SGAT :45-77, AttnReadout :80-118).  Same constructor signature, parameter names and
default (PyTorch) initialisation as the reference; `forward(mg, sg=None)` returns (B, num_items)
log-probabilities; `fused_loss` is the training entry.  nn.* children hold parameters only."""
import torch
import torch.nn as nn

from . import ops
from .srgnn import _ScoringMixin


def _graph(g):
    return (g.in_ptr, g.in_idx, g.out_ptr, g.out_idx, g.esrc, g.edst)


class EOPA(nn.Module):
    def __init__(self, input_dim, output_dim, batch_norm=True, feat_drop=0.0, activation=None):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(input_dim) if batch_norm else None
        self.feat_drop = nn.Dropout(feat_drop)
        self.gru = nn.GRU(input_dim, input_dim, batch_first=True)
        self.fc_self = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_neigh = nn.Linear(input_dim, output_dim, bias=False)
        self.activation = activation

    def forward(self, mg, feat):
        dN = mg.dynp('N')
        if self.batch_norm is not None:
            feat = ops.batch_norm(feat, self.batch_norm, dN)
        if mg.count('E') > 0:
            ft = self.feat_drop(feat)
            GI = ops.linear(ft, self.gru.weight_ih_l0, self.gru.bias_ih_l0, dN)
            neigh = ops.gru_seq(GI, self.gru.weight_hh_l0, self.gru.bias_hh_l0, _graph(mg), dN, mg.dynp('E'))
            rst = ops.linear_sum([feat, neigh], [self.fc_self.weight, self.fc_neigh.weight], None, dN)
        else:
            rst = ops.linear(feat, self.fc_self.weight, None, dN)
        if self.activation is not None:
            rst = ops.prelu(rst, self.activation.weight, dN)
        return rst


class SGAT(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, batch_norm=True, feat_drop=0.0, activation=None):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(input_dim) if batch_norm else None
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_q = nn.Linear(input_dim, hidden_dim, bias=True)
        self.fc_k = nn.Linear(input_dim, hidden_dim, bias=False)
        self.fc_v = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_e = nn.Linear(hidden_dim, 1, bias=False)
        self.activation = activation

    def forward(self, sg, feat):
        dN = sg.dynp('N')
        if self.batch_norm is not None:
            feat = ops.batch_norm(feat, self.batch_norm, dN)
        feat = self.feat_drop(feat)
        q = ops.linear(feat, self.fc_q.weight, self.fc_q.bias, dN)
        k = ops.linear(feat, self.fc_k.weight, None, dN)
        v = ops.linear(feat, self.fc_v.weight, None, dN)
        rst = ops.sgat_attn(q, k, self.fc_e.weight, v, _graph(sg), dN)
        if self.activation is not None:
            rst = ops.prelu(rst, self.activation.weight, dN)
        return rst


class AttnReadout(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, batch_norm=True, feat_drop=0.0, activation=None):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(input_dim) if batch_norm else None
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_u = nn.Linear(input_dim, hidden_dim, bias=False)
        self.fc_v = nn.Linear(input_dim, hidden_dim, bias=True)
        self.fc_e = nn.Linear(hidden_dim, 1, bias=False)
        self.fc_out = nn.Linear(input_dim, output_dim, bias=False) if output_dim != input_dim else None
        self.activation = activation

    def forward(self, mg, feat):
        dN, dB = mg.dynp('N'), mg.dynp('B')
        if self.batch_norm is not None:
            feat = ops.batch_norm(feat, self.batch_norm, dN)
        feat = self.feat_drop(feat)
        U = ops.linear(feat, self.fc_u.weight, None, dN, exact=True)
        Vq = ops.linear(ops.row_gather(feat, mg.last, dB, ascending=True), self.fc_v.weight, self.fc_v.bias, dB, exact=True)
        rst = ops.seg_attn(U, Vq, self.fc_e.weight, feat, mg.seg, dB)
        if self.fc_out is not None:
            rst = ops.linear(rst, self.fc_out.weight, None, dB, exact=True)
        if self.activation is not None:
            rst = ops.prelu(rst, self.activation.weight, dB)
        return rst


class LESSR(_ScoringMixin, nn.Module):
    graph_capable = True       # every layer (BatchNorm statistics included) reads the live extents of a padded batch

    def __init__(self, num_items, embedding_dim, num_layers, batch_norm=True, feat_drop=0.0):
        super().__init__()
        self.embedding = nn.Embedding(num_items, embedding_dim, max_norm=1)
        self.indices = nn.Parameter(torch.arange(num_items, dtype=torch.long), requires_grad=False)
        self.num_layers = num_layers
        self.layers = nn.ModuleList()
        input_dim = embedding_dim
        for i in range(num_layers):
            if i % 2 == 0:
                layer = EOPA(input_dim, embedding_dim, batch_norm, feat_drop, nn.PReLU(embedding_dim))
            else:
                layer = SGAT(input_dim, embedding_dim, embedding_dim, batch_norm, feat_drop, nn.PReLU(embedding_dim))
            input_dim += embedding_dim
            self.layers.append(layer)
        self.readout = AttnReadout(input_dim, embedding_dim, embedding_dim, batch_norm, feat_drop,
                                   nn.PReLU(embedding_dim))
        input_dim += embedding_dim
        self.batch_norm = nn.BatchNorm1d(input_dim) if batch_norm else None
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_sr = nn.Linear(input_dim, embedding_dim, bias=False)
        self._max_norm = 1.0

    def session_repr(self, mg, sg=None, tgrad=None):
        from ._lib import lib, ptr, stream
        W = self.embedding.weight
        if not self._take_prepared(self.__dict__.get('_srec_state'))[0]:   # (FusedAdam's row pass left the rows renormalised)
            with torch.no_grad():                # Embedding(max_norm=1): in-place renorm before the lookup
                lib.srec_renorm_rows(ptr(W), W.stride(0), None, W.shape[0], None, W.shape[1], 1.0, stream())
        dN, dB = mg.dynp('N'), mg.dynp('B')
        if mg.buf.is_cuda:
            ops.check_limits(mg)
            if sg is not None:
                ops.check_limits(sg, 'sgat_deg')
        feat = self._lookup(mg.iid, (mg.uniq_items, mg.uniq_ptr, mg.uniq_pos, mg.uniq_cptr, mg.chunk_ptr), tgrad,
                            dN, mg.dynp('U'), inv=mg.uniq_inv if mg.has('uniq_inv') else None)
        for i, layer in enumerate(self.layers):
            out = layer(mg, feat) if i % 2 == 0 else layer(sg, feat)
            feat = torch.cat([out, feat], dim=1)
        sr_g = self.readout(mg, feat)
        sr_l = ops.row_gather(feat, mg.last, dB, ascending=True)
        sr = torch.cat([sr_l, sr_g], dim=1)
        if self.batch_norm is not None:
            sr = ops.batch_norm(sr, self.batch_norm, dB)
        return ops.linear(self.feat_drop(sr), self.fc_sr.weight, None, dB, exact=True)

    def forward(self, mg, sg=None):
        return self._log_probs(self.session_repr(mg, sg))
