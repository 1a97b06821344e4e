"""ORACLE (test infrastructure, not product code): DGL-free pure-PyTorch CPU
restatement of the reference models' forward math, driven by the flat batch
dicts of oracle/collate_ref.py.  Autograd supplies the backward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (sessionrec-pytorch_amd/) never does.

Pinning status: the *glue* (concat orders, per-order loops, quirks) is pinned
against the unmodified reference sources imported through oracle/dgl_shim (see
tests/golden/make_golden.py).  The DGL 0.7.2 primitive semantics themselves
(degree-bucketed UDF reduce with edge-id ordered mailboxes, zero-fill of
zero-in-degree nodes, edge_softmax by destination, segment ops, HeteroGraphConv
skipping relations with no edges) are restated from DGL's documentation because
dgl is not installable here: at that boundary parity is UNPINNED.

Parameter names / shapes equal the reference's so state_dicts interchange:
  SRGNN   /root/reference/src/models/srgnn.py:93-148
  NISER   /root/reference/src/models/niser.py:91-157
  LESSR   /root/reference/src/models/lessr.py:121-183
  MSGIFSR /root/reference/src/models/msgifsr.py:157-323 (+ gnn_models/gatconv.py:219-319)
"""
import math

import torch as th
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- helpers
def to_torch(x):
    if isinstance(x, dict):
        return {k: to_torch(v) for k, v in x.items()}
    if hasattr(x, 'dtype') and not isinstance(x, th.Tensor):
        return th.from_numpy(x)
    return x


def seg_ids(num_nodes):
    return th.repeat_interleave(th.arange(len(num_nodes)), num_nodes)


def segment_softmax(num_nodes, e):
    """dgl.ops.segment.segment_softmax: max-subtracted softmax over contiguous segments."""
    sid = seg_ids(num_nodes)
    B = len(num_nodes)
    shape = (B,) + tuple(e.shape[1:])
    mx = th.full(shape, float('-inf'), dtype=e.dtype).index_reduce_(0, sid, e.detach(), 'amax', include_self=True)
    ex = th.exp(e - mx[sid])
    den = th.zeros(shape, dtype=e.dtype).index_add_(0, sid, ex)
    return ex / den[sid]


def segment_sum(num_nodes, x):
    sid = seg_ids(num_nodes)
    return th.zeros((len(num_nodes),) + tuple(x.shape[1:]), dtype=x.dtype).index_add_(0, sid, x)


def segment_mean(num_nodes, x):
    s = segment_sum(num_nodes, x)
    return s / num_nodes.clamp(min=1).to(x.dtype).view(-1, *([1] * (x.dim() - 1)))


def edge_softmax(dst, e, n_dst):
    """softmax over the in-edges of each destination node."""
    shape = (n_dst,) + tuple(e.shape[1:])
    mx = th.full(shape, float('-inf'), dtype=e.dtype).index_reduce_(0, dst, e.detach(), 'amax', include_self=True)
    ex = th.exp(e - mx[dst])
    den = th.zeros(shape, dtype=e.dtype).index_add_(0, dst, ex)
    return ex / den[dst]


# --------------------------------------------------------------------------- SRGNN / NISER
class SRGNNLayer(nn.Module):
    """srgnn.py:11-51 / niser.py:11-49."""

    def __init__(self, input_dim, output_dim, feat_drop=0.0):
        super().__init__()
        self.dropout = nn.Dropout(feat_drop)
        self.gru = nn.GRUCell(2 * input_dim, output_dim)
        self.W1 = nn.Linear(input_dim, output_dim, bias=False)
        self.W2 = nn.Linear(input_dim, output_dim, bias=False)

    @staticmethod
    def _wmean(ft, src, dst, w, n):
        wf = w.to(ft.dtype)
        num = th.zeros(n, ft.shape[1], dtype=ft.dtype).index_add_(0, dst, ft[src] * wf.unsqueeze(-1))
        den = th.zeros(n, dtype=ft.dtype).index_add_(0, dst, wf)
        has = den > 0                                   # zero in-degree -> zero fill
        return th.where(has.unsqueeze(-1), num / den.clamp(min=1e-30).unsqueeze(-1), th.zeros_like(num))

    def forward(self, g, feat):
        ft = self.dropout(feat)
        if len(g['src']) > 0:
            n = feat.shape[0]
            neigh1 = self._wmean(ft, g['src'], g['dst'], g['w'], n)
            neigh2 = self._wmean(ft, g['dst'], g['src'], g['w'], n)      # reversed graph
            hn = th.cat((self.W1(neigh1), self.W2(neigh2)), dim=1)
            return self.gru(hn, feat)
        return feat


class AttnReadout(nn.Module):
    """srgnn.py:53-91, niser.py:51-89, lessr.py:80-118."""

    def __init__(self, input_dim, hidden_dim, output_dim, batch_norm=True, feat_drop=0.0, activation=None):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(input_dim) if batch_norm else None
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_u = nn.Linear(input_dim, hidden_dim, bias=False)
        self.fc_v = nn.Linear(input_dim, hidden_dim, bias=True)
        self.fc_e = nn.Linear(hidden_dim, 1, bias=False)
        self.fc_out = nn.Linear(input_dim, output_dim, bias=False) if output_dim != input_dim else None
        self.activation = activation

    def forward(self, g, feat, last_nodes):
        if self.batch_norm is not None:
            feat = self.batch_norm(feat)
        feat = self.feat_drop(feat)
        nn_ = g['num_nodes']
        feat_u = self.fc_u(feat)
        feat_v = self.fc_v(feat[last_nodes])
        feat_v = feat_v[seg_ids(nn_)]                    # dgl.broadcast_nodes
        e = self.fc_e(th.sigmoid(feat_u + feat_v))
        alpha = segment_softmax(nn_, e)
        rst = segment_sum(nn_, feat * alpha)
        if self.fc_out is not None:
            rst = self.fc_out(rst)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst


class SRGNN(nn.Module):
    def __init__(self, num_items, embedding_dim, num_layers, feat_drop=0.0):
        super().__init__()
        self.embedding = nn.Embedding(num_items, embedding_dim)
        self.register_buffer('indices', th.arange(num_items, dtype=th.long))
        self.embedding_dim = embedding_dim
        self.layers = nn.ModuleList([SRGNNLayer(embedding_dim, embedding_dim, feat_drop) for _ in range(num_layers)])
        self.readout = AttnReadout(embedding_dim, embedding_dim, embedding_dim, batch_norm=None, feat_drop=feat_drop)
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_sr = nn.Linear(2 * embedding_dim, embedding_dim, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.embedding_dim)
        for w in self.parameters():
            w.data.uniform_(-stdv, stdv)

    def session_repr(self, mg):
        feat = self.feat_drop(self.embedding(mg['iid']))
        out = feat
        for layer in self.layers:                        # result unused (srgnn.py:135-142)
            out = layer(mg, out)
        last = mg['last']
        sr_g = self.readout(mg, feat, last)
        return self.fc_sr(th.cat([feat[last], sr_g], dim=1))

    def forward(self, mg, sg=None):
        sr = self.session_repr(mg)
        logits = sr @ self.embedding(self.indices).t()
        return th.log(F.softmax(logits, dim=-1))


class NISER(nn.Module):
    def __init__(self, num_items, embedding_dim, num_layers, feat_drop=0.0, norm=True, scale=12):
        super().__init__()
        self.embedding = nn.Embedding(num_items, embedding_dim)
        self.register_buffer('indices', th.arange(num_items, dtype=th.long))
        self.embedding_dim = embedding_dim
        self.norm, self.scale = norm, scale
        self.layers = nn.ModuleList([SRGNNLayer(embedding_dim, embedding_dim, feat_drop) for _ in range(num_layers)])
        self.readout = AttnReadout(embedding_dim, embedding_dim, embedding_dim, batch_norm=None, feat_drop=feat_drop)
        self.feat_drop = nn.Dropout(feat_drop)
        self.fc_sr = nn.Linear(2 * embedding_dim, embedding_dim, bias=False)
        self.reset_parameters()

    reset_parameters = SRGNN.reset_parameters

    def session_repr(self, mg):
        feat = self.feat_drop(self.embedding(mg['iid']))
        if self.norm:
            feat = feat.div(th.norm(feat, p=2, dim=-1, keepdim=True) + 1e-12)
        out = feat
        for layer in self.layers:
            out = layer(mg, out)
        last = mg['last']
        if self.norm:
            feat = feat.div(th.norm(feat, p=2, dim=-1, keepdim=True))
        sr_g = self.readout(mg, feat, last)
        sr = self.fc_sr(th.cat([feat[last], sr_g], dim=1))
        if self.norm:
            sr = sr.div(th.norm(sr, p=2, dim=-1, keepdim=True) + 1e-12)
        return sr

    def forward(self, mg, sg=None):
        sr = self.session_repr(mg)
        target = self.embedding(self.indices)
        if self.norm:
            target = target.div(th.norm(target, p=2, dim=-1, keepdim=True) + 1e-12)
        logits = sr @ target.t()
        if self.scale:
            logits = self.scale * logits
        return th.log(F.softmax(logits, dim=-1))


# --------------------------------------------------------------------------- LESSR
class EOPA(nn.Module):
    """lessr.py:8-42: per node, GRU over in-neighbour features in edge-id order."""

    def __init__(self, input_dim, output_dim, batch_norm=True, feat_drop=0.0, activation=None):
        super().__init__()
        self.batch_norm = nn.BatchNorm1d(input_dim) if batch_norm else None
        self.feat_drop = nn.Dropout(feat_drop)
        self.gru = nn.GRU(input_dim, input_dim, batch_first=True)
        self.fc_self = nn.Linear(input_dim, output_dim, bias=False)
        self.fc_neigh = nn.Linear(input_dim, output_dim, bias=False)
        self.activation = activation

    def forward(self, g, feat):
        if self.batch_norm is not None:
            feat = self.batch_norm(feat)
        ft = self.feat_drop(feat)
        src, dst = g['src'], g['dst']
        if len(src) > 0:
            n = feat.shape[0]
            deg = th.bincount(dst, minlength=n)
            order = th.argsort(dst, stable=True)          # in-edges of each node, by edge id
            ptr = th.cat([th.zeros(1, dtype=th.long), th.cumsum(deg, 0)])
            neigh = th.zeros(n, ft.shape[1], dtype=ft.dtype)
            for dg in th.unique(deg).tolist():            # degree bucketing
                if dg == 0:
                    continue
                nodes = th.nonzero(deg == dg).squeeze(1)
                eidx = ptr[nodes].unsqueeze(1) + th.arange(dg).unsqueeze(0)
                mail = ft[src[order[eidx]]]               # (n_deg, deg, D)
                _, hn = self.gru(mail)
                neigh = neigh.index_copy(0, nodes, hn.squeeze(0))
            rst = self.fc_self(feat) + self.fc_neigh(neigh)
        else:
            rst = self.fc_self(feat)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst


class SGAT(nn.Module):
    """lessr.py:45-77."""

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
        if self.batch_norm is not None:
            feat = self.batch_norm(feat)
        feat = self.feat_drop(feat)
        q, k, v = self.fc_q(feat), self.fc_k(feat), self.fc_v(feat)
        src, dst = sg['src'], sg['dst']
        e = self.fc_e(th.sigmoid(q[src] + k[dst]))        # u_add_v
        a = edge_softmax(dst, e, feat.shape[0])
        rst = th.zeros(feat.shape[0], v.shape[1], dtype=v.dtype).index_add_(0, dst, v[src] * a)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst


class LESSR(nn.Module):
    def __init__(self, num_items, embedding_dim, num_layers, batch_norm=True, feat_drop=0.0):
        super().__init__()
        self.embedding = nn.Embedding(num_items, embedding_dim, max_norm=1)
        self.indices = nn.Parameter(th.arange(num_items, dtype=th.long), requires_grad=False)
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

    def session_repr(self, mg, sg=None):
        feat = self.embedding(mg['iid'])
        for i, layer in enumerate(self.layers):
            out = layer(mg, feat) if i % 2 == 0 else layer(sg, feat)
            feat = th.cat([out, feat], dim=1)
        last = mg['last']
        sr_g = self.readout(mg, feat, last)
        sr = th.cat([feat[last], sr_g], dim=1)
        if self.batch_norm is not None:
            sr = self.batch_norm(sr)
        return self.fc_sr(self.feat_drop(sr))

    def forward(self, mg, sg=None):
        sr = self.session_repr(mg, sg)
        logits = sr @ self.embedding(self.indices).t()
        return th.softmax(logits, dim=1).log()


# --------------------------------------------------------------------------- MSGIFSR
class GATConv(nn.Module):
    """gatconv.py:136-319 with in_feats int, residual=True (Identity res_fc), bias,
    zero-in-degree nodes zero-filled (documented deviation: the shipped module
    would raise DGLError)."""

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0., negative_slope=0.2):
        super().__init__()
        self._num_heads, self._out_feats = num_heads, out_feats
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(th.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(th.empty(1, num_heads, out_feats))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        self.bias = nn.Parameter(th.empty(num_heads * out_feats))
        assert in_feats == out_feats                      # Identity residual only

    def forward(self, src, dst, h_src_in, h_dst_in, masks=None):
        """masks (tests only): (feature multipliers of the source rows, of the destination rows, attention multipliers
        [E, H]) applied INSTEAD of the module's own random dropout draws - lets a test replay the product's masks"""
        H, D = self._num_heads, self._out_feats
        if masks is not None:
            h_src, h_dst = h_src_in * masks[0], h_dst_in * masks[1]
        else:
            h_src = self.feat_drop(h_src_in)
            h_dst = self.feat_drop(h_dst_in)
        feat_src = self.fc(h_src).view(-1, H, D)
        feat_dst = self.fc(h_dst).view(-1, H, D)
        el = (feat_src * self.attn_l).sum(dim=-1).unsqueeze(-1)
        er = (feat_dst * self.attn_r).sum(dim=-1).unsqueeze(-1)
        e = self.leaky_relu(el[src] + er[dst])
        a = edge_softmax(dst, e, h_dst.shape[0])
        a = a * masks[2].view(-1, H, 1) if masks is not None else self.attn_drop(a)
        rst = th.zeros(h_dst.shape[0], H, D, dtype=feat_src.dtype).index_add_(0, dst, feat_src[src] * a)
        rst = rst + h_dst.view(h_dst.shape[0], -1, D)     # Identity residual broadcast over heads
        rst = rst + self.bias.view(1, -1, D)
        return rst


class HeteroConv(nn.Module):
    """dglnn.HeteroGraphConv(mods, aggregate='sum'): modules keyed by etype NAME
    (one shared 'inter'), relations with zero edges skipped, per-dst-type sum."""

    def __init__(self, mods):
        super().__init__()
        self.mods = nn.ModuleDict(mods)

    def forward(self, rels, feat, reverse=False, masks=None):
        """masks (tests only): {'feat': {node type: [N, D]}, 'attn': {relation key: [E, H]}} replayed dropout masks"""
        outs = {}
        for key in sorted(rels.keys(), key=lambda t: (('s%d' % t[0]), t[1], ('s%d' % t[2]))) if not reverse else \
                sorted(rels.keys(), key=lambda t: (('s%d' % t[2]), t[1], ('s%d' % t[0]))):
            s, et, d = key
            r = rels[key]
            if len(r['src']) == 0:
                continue
            if reverse:
                mk = None if masks is None else (masks['feat'][d], masks['feat'][s], masks['attn'][key])
                out = self.mods[et](r['dst'], r['src'], feat[d], feat[s], mk)
                outs.setdefault(s, []).append(out)
            else:
                mk = None if masks is None else (masks['feat'][s], masks['feat'][d], masks['attn'][key])
                out = self.mods[et](r['src'], r['dst'], feat[s], feat[d], mk)
                outs.setdefault(d, []).append(out)
        return {k: th.stack(v, 0).sum(0) for k, v in outs.items()}


class SemanticExpander(nn.Module):
    """msgifsr.py:14-45."""

    def __init__(self, input_dim, reducer, order):
        super().__init__()
        self.input_dim, self.order, self.reducer = input_dim, order, reducer
        self.GRUs = nn.ModuleList([nn.GRU(input_dim, input_dim, 1, True, True) for _ in range(order)])
        if reducer == 'concat':
            self.Ws = nn.ModuleList([nn.Linear(input_dim * (i + 1), input_dim) for i in range(1, order)])

    def forward(self, feat):
        if feat.dim() < 3:
            return feat
        if self.reducer == 'mean':
            invar = th.mean(feat, dim=1)
        elif self.reducer == 'max':
            invar = th.max(feat, dim=1)[0]
        else:
            invar = self.Ws[feat.size(1) - 2](feat.view(feat.size(0), -1))
        var = self.GRUs[feat.size(1) - 2](feat)[1].permute(1, 0, 2).squeeze()
        return 0.5 * invar + 0.5 * var


class MSHGNN(nn.Module):
    """msgifsr.py:47-91."""

    def __init__(self, input_dim, output_dim, dropout=0.0, activation=None, order=1):
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

    def forward(self, g, feat, masks=None):
        h1 = self.conv1(g['rel'], feat, masks=None if masks is None else masks['conv1'])
        h2 = self.conv2(g['rel'], feat, reverse=True, masks=None if masks is None else masks['conv2'])
        h = {}
        for k in range(1, self.order + 1):
            hl = h1.get(k, th.zeros(1, self.output_dim))
            hr = h2.get(k, th.zeros(1, self.output_dim))
            x = hl + hr
            if x.dim() > 2:
                x = x.max(1)[0]
            nn_ = g['num_nodes'][k]
            h_mean = segment_mean(nn_, feat[k])[seg_ids(nn_)]
            h[k] = h_mean + x
        return h


class AttnReadoutMS(nn.Module):
    """msgifsr.py:94-155."""

    def __init__(self, input_dim, hidden_dim, output_dim, feat_drop=0.0, activation=None, order=1):
        super().__init__()
        self.feat_drop = nn.Dropout(feat_drop)
        self.order = order
        self.fc_u = nn.ModuleList([nn.Linear(input_dim, hidden_dim, bias=True) for _ in range(order)])
        self.fc_v = nn.ModuleList([nn.Linear(input_dim, hidden_dim, bias=False) for _ in range(order)])
        self.fc_e = nn.ModuleList([nn.Linear(hidden_dim, 1, bias=False) for _ in range(order)])
        self.fc_p = nn.ModuleList()
        self.fc_out = nn.Linear(input_dim, output_dim, bias=False) if output_dim != input_dim else None
        self.activation = activation

    def forward(self, g, feats, last_nodess):
        K = self.order
        B = len(g['num_nodes'][1])
        splits = [th.split(feats[k], g['num_nodes'][k].tolist()) for k in range(1, K + 1)]
        feat_vs = th.cat([feats[k][last_nodess[k - 1]].unsqueeze(1) for k in range(1, K + 1)], dim=1)
        allf = th.cat([th.cat([splits[j][i] for j in range(K)], dim=0) for i in range(B)], dim=0)
        bnn = sum(g['num_nodes'][k] for k in range(1, K + 1))
        idx = seg_ids(bnn)
        rsts = []
        for i in range(K):
            feat_u = self.fc_u[i](allf)
            feat_v = self.fc_v[i](feat_vs[:, i])[idx]
            e = self.fc_e[i](th.sigmoid(feat_u + feat_v))
            alpha = segment_softmax(bnn, e)
            rsts.append(segment_sum(bnn, allf * alpha).unsqueeze(1))
        return th.cat(rsts, dim=1)


class MSGIFSR(nn.Module):
    def __init__(self, num_items, datasets, embedding_dim, num_layers, dropout=0.0, reducer='mean', order=3,
                 norm=True, extra=True, fusion=True, device=th.device('cpu')):
        super().__init__()
        self.embeddings = nn.Embedding(num_items, embedding_dim, max_norm=1)
        self.num_items = num_items
        self.register_buffer('indices', th.arange(num_items, dtype=th.long))
        self.embedding_dim, self.num_layers, self.reducer, self.order = embedding_dim, num_layers, reducer, order
        self.alpha = nn.Parameter(th.Tensor(order))
        self.beta = nn.Parameter(th.Tensor(1))
        self.norm = norm
        self.expander = SemanticExpander(embedding_dim, reducer, order)
        self.device = device
        self.layers = nn.ModuleList([
            MSHGNN(embedding_dim, embedding_dim, dropout=dropout, order=order, activation=nn.PReLU(embedding_dim))
            for _ in range(num_layers)])
        self.readout = AttnReadoutMS(embedding_dim, embedding_dim, embedding_dim, feat_drop=dropout, order=order)
        self.feat_drop = nn.Dropout(dropout)
        self.fc_sr = nn.ModuleList([nn.Linear(2 * embedding_dim, embedding_dim, bias=False) for _ in range(order)])
        self.sc_sr = nn.ModuleList([
            nn.Sequential(nn.Linear(embedding_dim, embedding_dim, bias=True), nn.ReLU(),
                          nn.Linear(embedding_dim, 2, bias=False), nn.Softmax(dim=-1)) for _ in range(order)])
        self.reset_parameters()
        self.alpha.data = th.zeros(order)
        self.alpha.data[0] = th.tensor(1.0)
        self.beta.data = th.tensor(1.0)
        self.fusion, self.extra = fusion, extra

    def reset_parameters(self):
        stdv = 1 / math.sqrt(self.embedding_dim)
        for w in self.parameters():
            w.data.uniform_(-stdv, stdv)

    def session_repr(self, mg, masks=None):
        """masks (tests only): {'rows': {k: multipliers shaped like embeddings(iid_k)}, 'layers': [MSHGNN masks]}: dropout
        multipliers replayed from the product instead of this module's own random draws"""
        K = self.order
        feats = {}
        for k in range(1, K + 1):
            rows = self.embeddings(mg['iid'][k])
            rows = rows * masks['rows'][k] if masks is not None else self.feat_drop(rows)
            feat = self.expander(rows)
            if th.isnan(feat).any():
                feat = feat.masked_fill(feat != feat, 0)
            if self.norm:
                feat = F.normalize(feat, dim=-1)
            if feat.dim() == 1:
                feat = feat.unsqueeze(0)
            feats[k] = feat
        h = feats
        for li, layer in enumerate(self.layers):
            h = layer(mg, h, None if masks is None else masks['layers'][li])
        last = []
        for k in range(1, K + 1):
            if self.norm:
                h[k] = F.normalize(h[k], dim=-1)
            last.append(mg['last'][k])
        sr_g = self.readout(mg, h, last)
        sr_l = th.cat([h[k][last[k - 1]].unsqueeze(1) for k in range(1, K + 1)], dim=1)
        sr = th.cat([sr_l, sr_g], dim=-1)
        sr = th.cat([self.fc_sr[i](s).unsqueeze(1) for i, s in enumerate(th.unbind(sr, dim=1))], dim=1)
        if self.norm:
            sr = F.normalize(sr, dim=-1)
        return sr                                          # (B, K, d)

    def forward(self, mg, masks=None):
        sr = self.session_repr(mg, masks)
        target = self.embeddings(self.indices)
        if self.norm:
            target = F.normalize(target, dim=-1)
        if self.extra:
            logits = sr @ target.t()                       # (B, K, V)
            phi = self.sc_sr[0](sr).unsqueeze(-1)          # (B, K, 2, 1)
            mask = th.zeros(phi.size(0), self.num_items)
            iids = th.split(mg['iid'][1], mg['num_nodes'][1].tolist())
            for i in range(len(mask)):
                mask[i, iids[i]] = 1
            logits_in = logits.masked_fill(~mask.bool().unsqueeze(1), float('-inf'))
            logits_ex = logits.masked_fill(mask.bool().unsqueeze(1), float('-inf'))
            score = th.softmax(12 * logits_in.squeeze(), dim=-1)
            score_ex = th.softmax(12 * logits_ex.squeeze(), dim=-1)
            if self.order == 1:
                phi = phi.squeeze(1)
                score = (th.cat((score.unsqueeze(1), score_ex.unsqueeze(1)), dim=1) * phi).sum(1)
            else:
                score = (th.cat((score.unsqueeze(2), score_ex.unsqueeze(2)), dim=2) * phi).sum(2)
        else:
            logits = sr.squeeze() @ target.t()
            score = th.softmax(12 * logits, dim=-1)
        if self.order > 1 and self.fusion:
            alpha = th.softmax(self.alpha.unsqueeze(0), dim=-1).view(1, self.alpha.size(0), 1)
            score = (score * alpha.repeat(score.size(0), 1, 1)).sum(1)
        elif self.order > 1:
            score = score[:, 0]
        return th.log(score)
