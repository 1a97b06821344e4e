"""ORACLE TOOLING (test infrastructure, never shipped to the GPU box's product path).

dgl==0.7.2 (the reference's pinned dependency, environment.yaml:14-15) cannot be
installed in the build container.  This module registers minimal pure-PyTorch
stand-ins for the ~30 `dgl` entry points the reference calls (SURVEY.md §8(c)
lists every call site), plus empty `numba` / `wandb` modules, so that the
UNMODIFIED reference sources under /root/reference can be imported and executed
on CPU to produce golden vectors (tests/golden/make_golden.py).

It pins the reference's *glue* code.  The DGL primitive semantics implemented
here are restated from DGL 0.7 documentation and are themselves unverifiable in
this container ("parity unpinned" at that boundary - see oracle/models_ref.py).
"""
import contextlib
import sys
import types

import torch as th


class DGLError(Exception):
    pass


# ----------------------------------------------------------------------------- graph
class _TypeView:
    def __init__(self, g, ntype):
        self.data = g._ndata[ntype]


class _NodesAccessor:
    def __init__(self, g):
        self._g = g

    def __getitem__(self, ntype):
        return _TypeView(self._g, ntype)


class _EdgeBatch:
    def __init__(self, src, dst, data):
        self.src, self.dst, self.data = src, dst, data


class _NodeBatch:
    def __init__(self, mailbox, data):
        self.mailbox, self.data = mailbox, data


class Graph:
    """Heterograph with per-type node data and per-relation edge lists."""

    def __init__(self, rels, num_nodes):
        self.ntypes = sorted(num_nodes.keys())
        self.canonical_etypes = sorted(rels.keys())
        self._nn = dict(num_nodes)
        self._edges = {k: (th.as_tensor(v[0], dtype=th.long).reshape(-1), th.as_tensor(v[1], dtype=th.long).reshape(-1))
                       for k, v in rels.items()}
        self._ndata = {t: {} for t in self.ntypes}
        self._edata = {k: {} for k in self.canonical_etypes}
        self._bnn = {t: th.tensor([self._nn[t]], dtype=th.long) for t in self.ntypes}
        self._bne = {k: th.tensor([len(self._edges[k][0])], dtype=th.long) for k in self.canonical_etypes}
        self.is_block = False

    # ---- single-relation conveniences
    def _one_rel(self):
        assert len(self.canonical_etypes) == 1
        return self.canonical_etypes[0]

    @property
    def ndata(self):
        assert len(self.ntypes) == 1
        return self._ndata[self.ntypes[0]]

    @property
    def edata(self):
        return self._edata[self._one_rel()]

    @property
    def srcdata(self):
        return self._ndata[self._one_rel()[0]]

    @property
    def dstdata(self):
        return self._ndata[self._one_rel()[2]]

    @property
    def nodes(self):
        return _NodesAccessor(self)

    def num_nodes(self, ntype=None):
        if ntype is None:
            return sum(self._nn.values())
        return self._nn[ntype]

    number_of_nodes = num_nodes

    def number_of_edges(self):
        return sum(len(e[0]) for e in self._edges.values())

    num_edges = number_of_edges

    def number_of_dst_nodes(self):
        return self._nn[self._one_rel()[2]]

    def add_nodes(self, num, ntype=None):
        ntype = ntype if ntype is not None else self.ntypes[0]
        old = self._nn[ntype]
        self._nn[ntype] = old + num
        self._bnn[ntype] = th.tensor([old + num], dtype=th.long)
        for k, v in list(self._ndata[ntype].items()):     # zero-extend existing features
            pad = th.zeros((num,) + tuple(v.shape[1:]), dtype=v.dtype)
            self._ndata[ntype][k] = th.cat([v, pad], 0)

    def batch_num_nodes(self, ntype=None):
        ntype = ntype if ntype is not None else self.ntypes[0]
        return self._bnn[ntype]

    def in_degrees(self):
        s, _, d = self._one_rel()
        return th.bincount(self._edges[self._one_rel()][1], minlength=self._nn[d])

    def to(self, device):
        return self

    def __getitem__(self, key):
        g = Graph({key: self._edges[key]}, {key[0]: self._nn[key[0]], key[2]: self._nn[key[2]]})
        g._ndata = {t: self._ndata[t] for t in g.ntypes}      # shared frames
        g._edata = {key: self._edata[key]}
        return g

    @contextlib.contextmanager
    def local_scope(self):
        nd = {t: dict(v) for t, v in self._ndata.items()}
        ed = {k: dict(v) for k, v in self._edata.items()}
        try:
            yield
        finally:
            for t in self._ndata:
                self._ndata[t].clear()
                self._ndata[t].update(nd[t])
            for k in self._edata:
                self._edata[k].clear()
                self._edata[k].update(ed[k])

    def reverse(self, copy_ndata=True, copy_edata=False):
        rels = {(d, e, s): (v[1], v[0]) for (s, e, d), v in self._edges.items()}
        g = Graph(rels, self._nn)
        g._ndata = {t: dict(v) for t, v in self._ndata.items()}
        if copy_edata:
            g._edata = {(d, e, s): dict(self._edata[(s, e, d)]) for (s, e, d) in self._edges}
        g._bnn = dict(self._bnn)
        g._bne = {(d, e, s): v for (s, e, d), v in self._bne.items()}
        return g

    def filter_nodes(self, predicate, ntype=None):
        ntype = ntype if ntype is not None else self.ntypes[0]
        mask = predicate(_NodeBatch(None, self._ndata[ntype]))
        return th.nonzero(mask).reshape(-1)

    def apply_edges(self, func):
        key = self._one_rel()
        src, dst = self._edges[key]
        out = func(_EdgeBatch({k: v[src] for k, v in self.srcdata.items() if th.is_tensor(v) and v.shape[0] == self._nn[key[0]]},
                              {k: v[dst] for k, v in self.dstdata.items() if th.is_tensor(v) and v.shape[0] == self._nn[key[2]]},
                              self._edata[key]))
        self._edata[key].update(out)

    def update_all(self, message_func, reduce_func):
        key = self._one_rel()
        s, _, d = key
        src, dst = self._edges[key]
        n_dst = self._nn[d]
        srcd = {k: v[src] for k, v in self._ndata[s].items() if th.is_tensor(v) and v.shape[0] == self._nn[s]}
        dstd = {k: v[dst] for k, v in self._ndata[d].items() if th.is_tensor(v) and v.shape[0] == n_dst}
        msgs = message_func(_EdgeBatch(srcd, dstd, self._edata[key]))
        if isinstance(reduce_func, _BuiltinSum):
            m = msgs[reduce_func.msg]
            out = th.zeros((n_dst,) + tuple(m.shape[1:]), dtype=m.dtype).index_add_(0, dst, m)
            self._ndata[d][reduce_func.out] = out
            return
        # UDF reduce: degree bucketing; mailbox rows ordered by edge id; zero in-degree -> zero fill
        deg = th.bincount(dst, minlength=n_dst)
        order = th.argsort(dst, stable=True)
        ptr = th.cat([th.zeros(1, dtype=th.long), th.cumsum(deg, 0)])
        results = {}
        for dg in th.unique(deg).tolist():
            if dg == 0:
                continue
            nodes = th.nonzero(deg == dg).reshape(-1)
            eidx = order[ptr[nodes].unsqueeze(1) + th.arange(dg).unsqueeze(0)]
            mailbox = {k: v[eidx] for k, v in msgs.items()}
            ndata = {k: v[nodes] for k, v in self._ndata[d].items() if th.is_tensor(v) and v.shape[0] == n_dst}
            out = reduce_func(_NodeBatch(mailbox, ndata))
            for k, v in out.items():
                if k not in results:
                    results[k] = th.zeros((n_dst,) + tuple(v.shape[1:]), dtype=v.dtype)
                results[k] = results[k].index_copy(0, nodes, v)
        self._ndata[d].update(results)


def graph(data, num_nodes=None):
    src, dst = data
    src = th.as_tensor(list(src) if not th.is_tensor(src) else src, dtype=th.long).reshape(-1)
    dst = th.as_tensor(list(dst) if not th.is_tensor(dst) else dst, dtype=th.long).reshape(-1)
    if num_nodes is None:
        num_nodes = int(max(src.max().item(), dst.max().item())) + 1 if len(src) else 0
    return Graph({('_N', '_E', '_N'): (src, dst)}, {'_N': num_nodes})


def heterograph(data_dict, num_nodes_dict=None):
    rels, nn = {}, {}
    for (s, e, d), (src, dst) in data_dict.items():
        src = th.as_tensor(list(src) if not th.is_tensor(src) else src, dtype=th.long).reshape(-1)
        dst = th.as_tensor(list(dst) if not th.is_tensor(dst) else dst, dtype=th.long).reshape(-1)
        rels[(s, e, d)] = (src, dst)
        nn[s] = max(nn.get(s, 0), int(src.max().item()) + 1 if len(src) else 0)
        nn[d] = max(nn.get(d, 0), int(dst.max().item()) + 1 if len(dst) else 0)
    if num_nodes_dict is not None:
        nn.update(num_nodes_dict)
    return Graph(rels, nn)


def batch(graphs):
    g0 = graphs[0]
    nn = {t: sum(g._nn[t] for g in graphs) for t in g0.ntypes}
    offs = {t: th.cumsum(th.tensor([0] + [g._nn[t] for g in graphs]), 0) for t in g0.ntypes}
    rels = {}
    for key in g0.canonical_etypes:
        s, _, d = key
        rels[key] = (th.cat([g._edges[key][0] + offs[s][i] for i, g in enumerate(graphs)]),
                     th.cat([g._edges[key][1] + offs[d][i] for i, g in enumerate(graphs)]))
    bg = Graph(rels, nn)
    for t in g0.ntypes:
        bg._bnn[t] = th.tensor([g._nn[t] for g in graphs], dtype=th.long)
        for k in g0._ndata[t]:
            bg._ndata[t][k] = th.cat([g._ndata[t][k] for g in graphs], 0)
    for key in g0.canonical_etypes:
        bg._bne[key] = th.tensor([len(g._edges[key][0]) for g in graphs], dtype=th.long)
        for k in g0._edata[key]:
            bg._edata[key][k] = th.cat([g._edata[key][k] for g in graphs], 0)
    return bg


def broadcast_nodes(g, feat, ntype=None):
    return th.repeat_interleave(feat, g.batch_num_nodes(ntype), dim=0)


# ----------------------------------------------------------------------------- dgl.function
class _BuiltinSum:
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


def _copy_u(u, out):
    return lambda edges: {out: edges.src[u]}


def _u_mul_e(u, e, out):
    return lambda edges: {out: edges.src[u] * edges.data[e]}


def _u_add_v(u, v, out):
    return lambda edges: {out: edges.src[u] + edges.dst[v]}


# ----------------------------------------------------------------------------- dgl.ops
def _seg_ids(lens):
    return th.repeat_interleave(th.arange(len(lens)), lens)


def segment_softmax(seglen, value):
    sid = _seg_ids(seglen)
    shape = (len(seglen),) + tuple(value.shape[1:])
    mx = th.full(shape, float('-inf'), dtype=value.dtype).index_reduce_(0, sid, value.detach(), 'amax')
    ex = th.exp(value - mx[sid])
    den = th.zeros(shape, dtype=value.dtype).index_add_(0, sid, ex)
    return ex / den[sid]


def segment_reduce(seglen, value, reducer='sum'):
    sid = _seg_ids(seglen)
    out = th.zeros((len(seglen),) + tuple(value.shape[1:]), dtype=value.dtype).index_add_(0, sid, value)
    if reducer == 'mean':
        out = out / seglen.clamp(min=1).to(value.dtype).view(-1, *([1] * (value.dim() - 1)))
    elif reducer != 'sum':
        raise NotImplementedError(reducer)
    return out


def edge_softmax(g, e):
    key = g._one_rel()
    dst = g._edges[key][1]
    n = g._nn[key[2]]
    shape = (n,) + tuple(e.shape[1:])
    mx = th.full(shape, float('-inf'), dtype=e.dtype).index_reduce_(0, dst, e.detach(), 'amax')
    ex = th.exp(e - mx[dst])
    den = th.zeros(shape, dtype=e.dtype).index_add_(0, dst, ex)
    return ex / den[dst]


def ops_u_add_v(g, x, y):
    src, dst = g._edges[g._one_rel()]
    return x[src] + y[dst]


def ops_u_mul_e_sum(g, x, e):
    key = g._one_rel()
    src, dst = g._edges[key]
    m = x[src] * e
    return th.zeros((g._nn[key[2]],) + tuple(m.shape[1:]), dtype=m.dtype).index_add_(0, dst, m)


# ----------------------------------------------------------------------------- dgl.nn
class Identity(th.nn.Module):
    def forward(self, x):
        return x


class HeteroGraphConv(th.nn.Module):
    def __init__(self, mods, aggregate='sum'):
        super().__init__()
        assert aggregate == 'sum'
        self.mods = th.nn.ModuleDict(mods)

    def forward(self, g, inputs):
        src_inputs, dst_inputs = inputs if isinstance(inputs, tuple) else (inputs, inputs)
        outputs = {nty: [] for nty in g.ntypes}
        for stype, etype, dtype in g.canonical_etypes:
            rel_graph = g[stype, etype, dtype]
            if rel_graph.number_of_edges() == 0:
                continue
            if stype not in src_inputs or dtype not in dst_inputs:
                continue
            outputs[dtype].append(self.mods[etype](rel_graph, (src_inputs[stype], dst_inputs[dtype])))
        return {nty: th.stack(alist, 0).sum(0) for nty, alist in outputs.items() if len(alist) != 0}


def expand_as_pair(in_feats):
    return in_feats if isinstance(in_feats, tuple) else (in_feats, in_feats)


def install():
    """Register the stand-in modules under their dgl names."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    seg = mod('dgl.ops.segment', segment_softmax=segment_softmax, segment_reduce=segment_reduce)
    ops = mod('dgl.ops', segment=seg, u_add_v=ops_u_add_v, edge_softmax=edge_softmax, u_mul_e_sum=ops_u_mul_e_sum,
              segment_softmax=segment_softmax, segment_reduce=segment_reduce)
    fn = mod('dgl.function', copy_u=_copy_u, u_mul_e=_u_mul_e, u_add_v=_u_add_v,
             sum=lambda msg, out: _BuiltinSum(msg, out))
    nnf = mod('dgl.nn.functional', edge_softmax=edge_softmax)
    utils_pt = mod('dgl.nn.pytorch.utils', Identity=Identity)
    nnpt = mod('dgl.nn.pytorch', HeteroGraphConv=HeteroGraphConv, utils=utils_pt)
    nnm = mod('dgl.nn', pytorch=nnpt, functional=nnf)
    base = mod('dgl.base', DGLError=DGLError)
    utils = mod('dgl.utils', expand_as_pair=expand_as_pair)
    mod('dgl', graph=graph, heterograph=heterograph, batch=batch, broadcast_nodes=broadcast_nodes,
        ops=ops, function=fn, nn=nnm, base=base, utils=utils, DGLError=DGLError)
    nb = mod('numba', jit=lambda *a, **k: (lambda f: f))
    nb.__dict__['jit'] = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    mod('wandb')
