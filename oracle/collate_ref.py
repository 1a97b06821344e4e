"""ORACLE (test infrastructure, not product code): CPU restatement of the
reference's session -> graph builders.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Pure-Python loops, written for clarity not speed.

Follows /root/reference/src/utils/data/collate.py:
  seq_to_eop_multigraph   collate.py:29-44
  seq_to_shortcut_graph   collate.py:46-59
  seq_to_session_graph    collate.py:61-85
  seq_to_ccs_graph        collate.py:87-217
  collate_fn_factory[_ccs] (dgl.batch: concatenate nodes/edges per type in
  graph order, offsetting node ids)   collate.py:219-256

Output format ("flat batch dict", all numpy):
  homogeneous graph:
    {'num_nodes': int64[B], 'num_edges': int64[B], 'iid': int64[N] | None,
     'last': int64[B] (global node id of the last-clicked node) | None,
     'src': int64[E], 'dst': int64[E] (global node ids), 'w': int64[E] | None}
  ccs heterograph (order K):
    {'order': K,
     'num_nodes': {k: int64[B]}, 'iid': {k: int64[N_k] (k=1) or int64[N_k,k]},
     'last': {k: int64[B]},
     'rel': {(stype, etype, dtype): {'src','dst','num_edges'}}}   types are 1..K
"""
import numpy as np


def _dedup_in_order(pairs):
    """Counter(...).keys()/.values(): distinct pairs in first-occurrence order
    with multiplicities (collate.py:53-56, 67-72)."""
    seen = {}
    for p in pairs:
        seen[p] = seen.get(p, 0) + 1
    return list(seen.keys()), list(seen.values())


def _nids(seq):
    items = np.unique(np.asarray(seq, dtype=np.int64))   # ascending ids; node id = rank
    iid2nid = {int(iid): i for i, iid in enumerate(items)}
    return items, iid2nid, [iid2nid[int(i)] for i in seq]


def seq_to_eop_multigraph(seq):
    items, iid2nid, seq_nid = _nids(seq)
    if len(seq) > 1:
        src, dst = seq_nid[:-1], seq_nid[1:]
    else:
        src, dst = [], []
    return dict(n=len(items), iid=items, last=iid2nid[int(seq[-1])],
                src=list(src), dst=list(dst), w=None)


def seq_to_shortcut_graph(seq):
    items, _, seq_nid = _nids(seq)
    L = len(seq)
    edges, _ = _dedup_in_order([(seq_nid[i], seq_nid[j]) for i in range(L) for j in range(i, L)])
    src, dst = zip(*edges)
    return dict(n=len(items), iid=None, last=None, src=list(src), dst=list(dst), w=None)


def seq_to_session_graph(seq):
    items, iid2nid, seq_nid = _nids(seq)
    edges, cnt = _dedup_in_order([(seq_nid[i], seq_nid[i + 1]) for i in range(len(seq) - 1)])
    if len(edges) > 0:
        src, dst = zip(*edges)
        w = cnt
    else:                       # single click: one self loop with weight 1
        src, dst, w = [0], [0], [1]
    return dict(n=len(items), iid=items, last=iid2nid[int(seq[-1])],
                src=list(src), dst=list(dst), w=list(w))


def batch_graphs(graphs):
    """dgl.batch for homogeneous graphs."""
    num_nodes = np.array([g['n'] for g in graphs], dtype=np.int64)
    num_edges = np.array([len(g['src']) for g in graphs], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(num_nodes)])
    src = np.concatenate([np.asarray(g['src'], dtype=np.int64) + off[i] for i, g in enumerate(graphs)]) \
        if num_edges.sum() else np.zeros(0, np.int64)
    dst = np.concatenate([np.asarray(g['dst'], dtype=np.int64) + off[i] for i, g in enumerate(graphs)]) \
        if num_edges.sum() else np.zeros(0, np.int64)
    out = dict(num_nodes=num_nodes, num_edges=num_edges, src=src.astype(np.int64), dst=dst.astype(np.int64),
               iid=None, last=None, w=None)
    if graphs[0]['iid'] is not None:
        out['iid'] = np.concatenate([g['iid'] for g in graphs]).astype(np.int64)
        out['last'] = np.array([g['last'] + off[i] for i, g in enumerate(graphs)], dtype=np.int64)
    if graphs[0]['w'] is not None:
        out['w'] = np.concatenate([np.asarray(g['w'], dtype=np.int64) for g in graphs])
    return out


def seq_to_ccs_graph(seq, order=1):
    """collate.py:87-217.  Node types 1..K; returns per-type node data and
    per-relation edge lists (local node ids)."""
    K = order
    seq = [int(x) for x in seq]
    L = len(seq)
    eff = min(K, L)
    items, iid2nid, seq_nid = _nids(seq)
    last = {1: iid2nid[seq[-1]]}
    gram_dicts = {1: None}
    gram_lists = {}
    for k in range(2, K + 1):                      # collate.py:119-140
        d, lst, key = {}, [], None
        for j in range(L - k + 1):
            key = tuple(seq[j:j + k])              # str(seq[j:j+k]) in the reference
            if key not in d:
                d[key] = len(d)
                lst.append(list(key))
        last[k] = d[key] if len(d) > 0 else 0      # id of the LAST gram in click order
        gram_dicts[k], gram_lists[k] = d, lst

    def gid(k, i):
        return gram_dicts[k][tuple(seq[i:i + k])]

    rel = {}
    for k in range(1, eff + 1):                    # collate.py:144-159
        if k == 1:
            pairs = [(seq_nid[i], seq_nid[i + 1]) for i in range(L - 1)]
        else:
            pairs = [(gid(k, i), gid(k, i + 1)) for i in range(L - k)]
        edges, _ = _dedup_in_order(pairs)
        rel[(k, 'intra%d' % k, k)] = edges
    for k in range(2, eff + 1):                    # collate.py:161-189
        edges, _ = _dedup_in_order([(seq_nid[i], gid(k, i + 1)) for i in range(L - k)])
        rel[(1, 'inter', k)] = edges
        edges, _ = _dedup_in_order([(gid(k, i), seq_nid[i + k]) for i in range(L - k)])
        rel[(k, 'inter', 1)] = edges
    for k in range(eff + 1, K + 1):                # collate.py:191-195
        rel[(k, 'intra%d' % k, k)] = []
        rel[(k, 'inter', 1)] = []
        rel[(1, 'inter', k)] = []

    n = {1: len(items)}
    iid = {1: items}
    for k in range(2, K + 1):
        if k <= eff:                               # collate.py:207-213
            n[k] = len(gram_lists[k])
            iid[k] = np.asarray(gram_lists[k], dtype=np.int64).reshape(n[k], k)
        else:                                      # dummy node, collate.py:203-208
            n[k] = 1
            iid[k] = np.full((1, k), items[0], dtype=np.int64)
            last[k] = 0
    return dict(order=K, n=n, iid=iid, last=last, rel=rel)


def batch_ccs(graphs):
    K = graphs[0]['order']
    B = len(graphs)
    out = dict(order=K, num_nodes={}, iid={}, last={}, rel={})
    offs = {}
    for k in range(1, K + 1):
        nn = np.array([g['n'][k] for g in graphs], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(nn)])
        offs[k] = off
        out['num_nodes'][k] = nn
        out['iid'][k] = np.concatenate([g['iid'][k] for g in graphs], axis=0).astype(np.int64)
        out['last'][k] = np.array([g['last'][k] + off[i] for i, g in enumerate(graphs)], dtype=np.int64)
    for key in graphs[0]['rel'].keys():
        s, _, d = key
        src, dst, ne = [], [], np.zeros(B, np.int64)
        for i, g in enumerate(graphs):
            e = g['rel'][key]
            ne[i] = len(e)
            for (a, b) in e:
                src.append(a + offs[s][i])
                dst.append(b + offs[d][i])
        out['rel'][key] = dict(src=np.asarray(src, dtype=np.int64), dst=np.asarray(dst, dtype=np.int64),
                               num_edges=ne)
    return out


def collate_fn_factory(*seq_to_graph_fns):
    def collate_fn(samples):
        seqs, labels = zip(*samples)
        inputs = [batch_graphs([fn(s) for s in seqs]) for fn in seq_to_graph_fns]
        return inputs, np.asarray(labels, dtype=np.int64)
    return collate_fn


def collate_fn_factory_ccs(seq_to_graph_fns, order):
    def collate_fn(samples):
        seqs, labels = zip(*samples)
        inputs = [batch_ccs([fn(s, order) for s in seqs]) for fn in seq_to_graph_fns]
        return inputs, np.asarray(labels, dtype=np.int64)
    return collate_fn
