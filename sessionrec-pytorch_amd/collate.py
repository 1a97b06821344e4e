"""Session -> flat batched graph builders: the host-side mirror of the reference's
collate surface (src/utils/data/collate.py), emitting FlatBatch objects instead of
DGL graphs.

  seq_to_eop_multigraph   collate.py:29-44   every consecutive transition, click order
  seq_to_shortcut_graph   collate.py:46-59   distinct (i<=j) position pairs, self loops included
  seq_to_session_graph    collate.py:61-85   distinct transitions + multiplicity, self loop for 1 click
  seq_to_ccs_graph        collate.py:87-217  1..K-gram heterograph
  collate_fn_factory / collate_fn_factory_ccs   collate.py:219-256

Same call surface: `collate_fn_factory(*fns)(samples) -> (inputs, labels)` with
`samples = [(seq, label), ...]`; every element of `inputs` supports `.to(device)`.
The per-sequence builders return small local graphs (python tuples); batching
offsets node ids exactly like dgl.batch and additionally emits what the HIP
kernels want: int32 everywhere, in-/out-edge CSR per relation (edge-id ordered, so
EOPA's time order is preserved), the item -> positions CSR for the deterministic
embedding backward, and the per-session concatenation permutation of the
multi-order readout.  When the native builder (csrc/collate.cpp) is present it is
used instead of the python loops; both produce identical buffers.
"""
import numpy as np
import torch

from .batch import CapacityExceeded, FlatBatch


# ------------------------------------------------------------------------------------ per-sequence
def _rank(seq):
    seq = np.asarray(seq, dtype=np.int64)
    items, nid = np.unique(seq, return_inverse=True)
    return items, nid.tolist()


def _first_occurrence(pairs):
    d = {}
    for p in pairs:
        d[p] = d.get(p, 0) + 1
    return d


def seq_to_eop_multigraph(seq):
    items, nid = _rank(seq)
    return ('eop', items, nid[-1], nid[:-1], nid[1:], None)


def seq_to_shortcut_graph(seq):
    items, nid = _rank(seq)
    L = len(nid)
    d = _first_occurrence((nid[i], nid[j]) for i in range(L) for j in range(i, L))
    src, dst = zip(*d.keys())
    return ('shortcut', items, None, list(src), list(dst), None)


def seq_to_session_graph(seq):
    items, nid = _rank(seq)
    d = _first_occurrence(zip(nid[:-1], nid[1:]))
    if d:
        src, dst = zip(*d.keys())
        w = list(d.values())
    else:
        src, dst, w = (0,), (0,), [1]
    return ('session', items, nid[-1], list(src), list(dst), w)


def seq_to_ccs_graph(seq, order=1, coaDict=None):
    K = order
    seq = [int(x) for x in seq]
    L = len(seq)
    eff = min(K, L)
    items, nid = _rank(seq)
    n = {1: len(items)}
    iid = {1: items}
    last = {1: nid[-1]}
    gid = {1: nid}                       # per order: gram id at each start position
    for k in range(2, K + 1):
        ids, table, grams = [], {}, []
        for j in range(L - k + 1):
            key = tuple(seq[j:j + k])
            g = table.get(key)
            if g is None:
                g = table[key] = len(table)
                grams.append(key)
            ids.append(g)
        gid[k] = ids
        if k <= eff:
            n[k] = len(grams)
            iid[k] = np.asarray(grams, dtype=np.int64).reshape(-1, k)
            last[k] = ids[-1]
        else:                            # session shorter than k: one dummy node (collate.py:203-208)
            n[k] = 1
            iid[k] = np.full((1, k), items[0], dtype=np.int64)
            last[k] = 0
    rel = {}
    for k in range(1, K + 1):
        if k <= eff:
            g = gid[k]
            rel[(k, 'intra%d' % k, k)] = list(_first_occurrence(zip(g[:-1], g[1:])).keys())
        else:
            rel[(k, 'intra%d' % k, k)] = []
    for k in range(2, K + 1):
        if k <= eff:
            g = gid[k]
            rel[(1, 'inter', k)] = list(_first_occurrence((nid[i], g[i + 1]) for i in range(L - k)).keys())
            rel[(k, 'inter', 1)] = list(_first_occurrence((g[i], nid[i + k]) for i in range(L - k)).keys())
        else:
            rel[(1, 'inter', k)] = []
            rel[(k, 'inter', 1)] = []
    return ('ccs', K, n, iid, last, rel)


# ------------------------------------------------------------------------------------ batching
def _csr(key, n):
    """stable grouping of edge ids by `key` (node id): ptr[n+1], idx[E]."""
    key = np.asarray(key, dtype=np.int64)
    idx = np.argsort(key, kind='stable').astype(np.int32)
    ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(key, minlength=n), out=ptr[1:])
    return ptr, idx


def _uniq_csr(gidx):
    """distinct items, per item the positions where it is looked up, and per position the slot of its item (the inverse map:
    what the row-sharded lookup gathers the exchanged rows through - dist.VocabParallel.lookup)."""
    gidx = np.asarray(gidx, dtype=np.int64)
    pos = np.argsort(gidx, kind='stable').astype(np.int32)
    items, inv, cnt = np.unique(gidx, return_inverse=True, return_counts=True)
    ptr = np.zeros(len(items) + 1, dtype=np.int32)
    np.cumsum(cnt, out=ptr[1:])
    return items.astype(np.int32), ptr, pos, np.asarray(inv).reshape(-1).astype(np.int32)


CHUNK = 16


def _chunk_csr(ptr):
    """split every item's position list into pieces of <= CHUNK positions (popular items appear hundreds
    of times per batch; one wavefront per piece keeps the segmented sum balanced)."""
    cnt = np.diff(ptr)
    nch = (cnt + CHUNK - 1) // CHUNK
    cptr = np.zeros(len(cnt) + 1, dtype=np.int32)          # item -> chunk range
    np.cumsum(nch, out=cptr[1:])
    C = int(cptr[-1])
    owner = np.repeat(np.arange(len(cnt)), nch)
    k = np.arange(C) - cptr[:-1][owner]
    beg = ptr[:-1][owner] + k * CHUNK
    end = np.minimum(beg + CHUNK, ptr[1:][owner])
    chunk_ptr = np.concatenate([beg, end[-1:]]).astype(np.int32) if C else np.zeros(1, np.int32)
    return cptr, chunk_ptr


def _cat(lst, dtype=np.int64):
    lst = [np.asarray(a, dtype=dtype).reshape(-1) for a in lst]
    return np.concatenate(lst) if lst else np.zeros(0, dtype)


def batch_homogeneous(graphs, caps=None):
    kind = graphs[0][0]
    B = len(graphs)
    nn = np.array([len(g[1]) for g in graphs], dtype=np.int64)
    seg = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(nn, out=seg[1:])
    N = int(seg[-1])
    ne = np.array([len(g[3]) for g in graphs], dtype=np.int64)
    eseg = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(ne, out=eseg[1:])
    src = _cat([np.asarray(g[3], dtype=np.int64) + seg[i] for i, g in enumerate(graphs)])
    dst = _cat([np.asarray(g[4], dtype=np.int64) + seg[i] for i, g in enumerate(graphs)])
    E = len(src)
    in_ptr, in_idx = _csr(dst, N)
    out_ptr, out_idx = _csr(src, N)
    fields = dict(seg=seg, eseg=eseg, esrc=src, edst=dst, in_ptr=in_ptr, in_idx=in_idx, out_ptr=out_ptr,
                  out_idx=out_idx)
    counts = dict(B=B, N=N, E=E)
    if kind != 'shortcut':
        iid = _cat([g[1] for g in graphs])
        fields['iid'] = iid
        fields['last'] = np.array([g[2] + seg[i] for i, g in enumerate(graphs)], dtype=np.int64)
        ui, up, upos, uinv = _uniq_csr(iid)
        cptr, chptr = _chunk_csr(up)
        fields.update(uniq_items=ui, uniq_ptr=up, uniq_pos=upos, uniq_inv=uinv, uniq_cptr=cptr, chunk_ptr=chptr)
        counts['U'] = len(ui)
        counts['C'] = len(chptr) - 1
    if kind == 'session':
        fields['ew'] = _cat([g[5] for g in graphs])
    Bc = caps['B'] if caps else B
    meta = dict(kind=kind, B=Bc, max_nodes=int(nn.max()) if B else 0, padded=caps is not None)
    fcaps = None
    if caps:
        Nc, Ec, Uc = caps['N'], caps['E'], caps['U']
        if N > Nc or E > Ec:
            raise CapacityExceeded('nodes %d / %d, edges %d / %d' % (N, Nc, E, Ec))
        fcaps = dict(seg=Bc + 1, eseg=Bc + 1, esrc=Ec, edst=Ec, in_ptr=Nc + 1, in_idx=Ec, out_ptr=Nc + 1, out_idx=Ec,
                     iid=Nc, last=Bc, uniq_items=Uc, uniq_ptr=Uc + 1, uniq_pos=Nc, uniq_inv=Nc, uniq_cptr=Uc + 1,
                     chunk_ptr=Uc + Nc // CHUNK + 2, ew=Ec)
        fcaps = {k: v for k, v in fcaps.items() if k in fields}
    return FlatBatch.build(_items_last(fields), counts, meta, fcaps)


def batch_ccs(graphs, caps=None):
    """caps (optional): {'B': sessions, 'N': nodes per order, 'E': edges per relation, 'U': distinct items}
    -> capacity-padded layout whose offsets do not depend on the batch (hipGraph replay)."""
    K = graphs[0][1]
    B = len(graphs)
    Bc = caps['B'] if caps else B
    fields, counts = {}, dict(B=B)
    segs, ncap = {}, {}
    for k in range(1, K + 1):
        nn = np.array([g[2][k] for g in graphs], dtype=np.int64)
        seg = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(nn, out=seg[1:])
        segs[k] = seg
        ncap[k] = caps['N'] if caps else int(seg[-1])
        if seg[-1] > ncap[k]:
            raise CapacityExceeded('order-%d nodes %d / %d' % (k, int(seg[-1]), ncap[k]))
        fields['seg%d' % k] = seg
        iid = np.concatenate([np.asarray(g[3][k], dtype=np.int64).reshape(-1, k) for g in graphs], axis=0)
        fields['iid%d' % k] = iid if k > 1 else iid.reshape(-1)
        fields['last%d' % k] = np.array([g[4][k] + seg[i] for i, g in enumerate(graphs)], dtype=np.int64)
        counts['N%d' % k] = int(seg[-1])
        counts['GK%d' % k] = int(seg[-1]) * k
    # one fused embedding lookup for all orders: rows = [iid1 | iid2.flat | iid3.flat ...], every order's block
    # padded to its capacity (idx -1 -> zero row) so the block offsets are static
    blocks = []
    for k in range(1, K + 1):
        blk = np.full(ncap[k] * k, -1, dtype=np.int64)
        flat = np.asarray(fields['iid%d' % k]).reshape(-1)
        blk[:flat.size] = flat
        blocks.append(blk)
    gidx = np.concatenate(blocks)
    livepos = np.nonzero(gidx >= 0)[0]
    ui, up, upos, uinv_live = _uniq_csr(gidx[livepos])
    upos = livepos[upos].astype(np.int32)
    uinv = np.full(len(gidx), -1, dtype=np.int32)          # (capacity padding inside the order blocks: no item)
    uinv[livepos] = uinv_live
    cptr, chptr = _chunk_csr(up)
    fields.update(gidx=gidx, uniq_items=ui, uniq_ptr=up, uniq_pos=upos, uniq_inv=uinv, uniq_cptr=cptr, chunk_ptr=chptr)
    counts['G'] = len(gidx)
    counts['U'] = len(ui)
    counts['C'] = len(chptr) - 1
    rel_names = []
    for key in sorted(graphs[0][5].keys()):
        s, et, d = key
        name = 'r_%d_%s_%d' % (s, et, d)
        rel_names.append((key, name))
        src = _cat([np.asarray([e[0] for e in g[5][key]], dtype=np.int64) + segs[s][i] for i, g in enumerate(graphs)])
        dst = _cat([np.asarray([e[1] for e in g[5][key]], dtype=np.int64) + segs[d][i] for i, g in enumerate(graphs)])
        in_ptr, in_idx = _csr(dst, int(segs[d][-1]))
        out_ptr, out_idx = _csr(src, int(segs[s][-1]))
        fields.update({name + '_src': src, name + '_dst': dst, name + '_in_ptr': in_ptr, name + '_in_idx': in_idx,
                       name + '_out_ptr': out_ptr, name + '_out_idx': out_idx})
        counts['E_' + name] = len(src)
    # readout: per session, nodes of all orders concatenated [s1 | s2 | ...] (msgifsr.py:135); rows of the
    # stacked [h1 ; h2 ; ...] tensor whose blocks start at the (capacity) offsets below
    offs = np.concatenate([[0], np.cumsum([ncap[k] for k in range(1, K + 1)])])
    perm = []
    cat_seg = np.zeros(B + 1, dtype=np.int64)
    for i in range(B):
        for k in range(1, K + 1):
            perm.append(np.arange(segs[k][i], segs[k][i + 1]) + offs[k - 1])
        cat_seg[i + 1] = cat_seg[i] + sum(int(segs[k][i + 1] - segs[k][i]) for k in range(1, K + 1))
    perm = _cat(perm)
    inv = np.full(int(offs[-1]), -1, dtype=np.int64)
    inv[perm] = np.arange(len(perm))
    fields.update(cat_perm=perm, cat_inv=inv, cat_seg=cat_seg)
    for k in range(1, K + 1):
        fields['lastcat%d' % k] = fields['last%d' % k] + offs[k - 1]
    counts['NT'] = len(perm)
    max_nodes = int((cat_seg[1:] - cat_seg[:-1]).max()) if B else 0
    meta = dict(kind='ccs', order=K, B=Bc, rels=rel_names, max_nodes=max_nodes, ncap=ncap, padded=caps is not None)
    fcaps = None
    if caps:
        N, E, U = caps['N'], caps['E'], caps['U']
        fcaps = dict(uniq_items=U, uniq_ptr=U + 1, uniq_pos=len(gidx), uniq_inv=len(gidx), uniq_cptr=U + 1,
                     chunk_ptr=U + len(gidx) // CHUNK + 2, cat_perm=N * K, cat_seg=Bc + 1)
        for k in range(1, K + 1):
            fcaps.update({'seg%d' % k: Bc + 1, 'iid%d' % k: N * k, 'last%d' % k: Bc, 'lastcat%d' % k: Bc})
        for _, name in rel_names:
            fcaps.update({name + '_src': E, name + '_dst': E, name + '_in_idx': E, name + '_out_idx': E,
                          name + '_in_ptr': N + 1, name + '_out_ptr': N + 1})
    return FlatBatch.build(_items_last(fields), counts, meta, fcaps)


# ------------------------------------------------------------------------------------ native builder
_NATIVE = None


def _native():
    """libsrec_collate.so (csrc/collate.cpp) or None; SREC_PY_COLLATE=1 forces the python builders"""
    global _NATIVE
    if _NATIVE is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libsrec_collate.so')
        if os.environ.get('SREC_PY_COLLATE') == '1' or not os.path.exists(path):
            _NATIVE = False
        else:
            dll = ctypes.CDLL(path)
            dll.srec_collate.restype = ctypes.c_long
            dll.srec_collate.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p]
            _NATIVE = dll
    return _NATIVE or None


def _items_last(fields):
    """the distinct-item list is the LAST field of a batch (dict of fields, or list of field names): the labels the loaders write
    behind the batch then follow it without a gap (capacities are multiples of 4 words), and (items | labels) is one contiguous
    request list for the row-sharded lookup (dist.ShardedLookup) - no copy launch in the rank step"""
    if isinstance(fields, dict):
        if 'uniq_items' in fields:
            fields['uniq_items'] = fields.pop('uniq_items')
        return fields
    return [n for n in fields if n != 'uniq_items'] + (['uniq_items'] if 'uniq_items' in fields else [])


_HOMOG_FIELDS = ['seg', 'eseg', 'esrc', 'edst', 'in_ptr', 'in_idx', 'out_ptr', 'out_idx']
_ITEM_FIELDS = ['iid', 'last', 'uniq_items', 'uniq_ptr', 'uniq_pos', 'uniq_inv', 'uniq_cptr', 'chunk_ptr']


def _ccs_schema(K):
    rel_keys = sorted([(k, 'intra%d' % k, k) for k in range(1, K + 1)] + [(1, 'inter', k) for k in range(2, K + 1)] +
                      [(k, 'inter', 1) for k in range(2, K + 1)])
    rels = [(key, 'r_%d_%s_%d' % key) for key in rel_keys]
    names, shapes, counts = [], {}, ['B']
    for k in range(1, K + 1):
        names += ['seg%d' % k, 'iid%d' % k, 'last%d' % k]
        if k > 1:
            shapes['iid%d' % k] = (k,)
        counts += ['N%d' % k, 'GK%d' % k]
    names += ['gidx', 'uniq_items', 'uniq_ptr', 'uniq_pos', 'uniq_inv', 'uniq_cptr', 'chunk_ptr']
    counts += ['G', 'U', 'C']
    for _, n in rels:
        names += [n + sfx for sfx in ('_src', '_dst', '_in_ptr', '_in_idx', '_out_ptr', '_out_idx')]
        counts.append('E_' + n)
    names += ['cat_perm', 'cat_inv', 'cat_seg'] + ['lastcat%d' % k for k in range(1, K + 1)]
    counts.append('NT')
    return _items_last(names), shapes, counts, rels


class FlatSeqs:
    """a batch of click sequences already in the builder's input form: flat int64 clicks + int64 offsets [B + 1]
    (loader.PinnedRingLoader gathers it from the flattened dataset with numpy - no per-sample Python objects)"""

    def __init__(self, flat, offs):
        self.flat, self.offs = np.ascontiguousarray(flat, dtype=np.int64), np.ascontiguousarray(offs, dtype=np.int64)

    def __len__(self):
        return len(self.offs) - 1


def collate_native(kind, seqs, order=1, caps=None, into=None):
    """kind: 'session' | 'eop' | 'shortcut' | 'ccs' -> FlatBatch built by csrc/collate.cpp (or None if unavailable).
    into (optional): a writable int32 numpy array (e.g. a slot of a pinned shared ring, loader.PinnedRingLoader) the
    builder writes the batch INTO - the FlatBatch then is a view of it, nothing is copied; too small -> ValueError."""
    import ctypes
    dll = _native()
    if dll is None:
        return None
    B = len(seqs)
    if isinstance(seqs, FlatSeqs):
        flat, offs = seqs.flat, seqs.offs
    else:
        import itertools
        lens = np.fromiter(map(len, seqs), dtype=np.int64, count=B)
        offs = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        flat = np.fromiter(itertools.chain.from_iterable(seqs), dtype=np.int64, count=int(offs[-1]))   # (C-level iteration:
        #                                                              a generator expression costs ~0.2 us per click here)
    kid = {'session': 0, 'eop': 1, 'shortcut': 2, 'ccs': 3}[kind]
    if kind == 'ccs':
        names, shapes, cnames, rels = _ccs_schema(order)
    else:
        names = _items_last(list(_HOMOG_FIELDS) + (list(_ITEM_FIELDS) if kind != 'shortcut' else []) + (['ew'] if kind == 'session' else []))
        shapes, rels = {}, None
        cnames = ['B', 'N', 'E'] + (['U', 'C'] if kind != 'shortcut' else [])
    capv = None if caps is None else np.array([caps['B'], caps['N'], caps['E'], caps['U']], dtype=np.int64)
    total = int(offs[-1])
    if caps is None:
        guess = 64 + 40 * (total + B) * (order if kind == 'ccs' else 1) + (total * 21 if kind == 'shortcut' else 0)
    else:
        guess = 64 + (12 * order + 16) * (caps['N'] * max(order, 1) + caps['E'] + caps['U'] + caps['B']) + 4096
    info = np.zeros(3 * (len(names) + 4), dtype=np.int64)
    nf = ctypes.c_int(0)
    for _ in range(2):
        out = into if into is not None else np.empty(guess, dtype=np.int32)
        n = dll.srec_collate(kid, flat.ctypes.data, offs.ctypes.data, B, order,
                             None if capv is None else capv.ctypes.data, out.ctypes.data, len(out), info.ctypes.data,
                             len(names) + 4, ctypes.addressof(nf))
        if n > 0:
            break
        if n == 0:
            if caps is not None:
                raise CapacityExceeded('native collate: the batch does not fit the capacities %r' % (caps,))
            raise ValueError('native collate failed: empty session or order > 6')
        if into is not None:
            raise ValueError('native collate: the batch needs %d int32, the given buffer holds %d' % (-n, len(out)))
        guess = -n + 16
    assert nf.value == len(names), (nf.value, len(names))
    buf = out[:n] if into is not None else out[:n].copy()
    layout = {nm: (int(info[3 * i]), int(info[3 * i + 1]), shapes.get(nm)) for i, nm in enumerate(names)}
    counts = {cn: int(buf[i]) for i, cn in enumerate(cnames)}
    meta = dict(kind=kind, B=caps['B'] if caps else B, padded=caps is not None, counts=counts,
                slots={cn: i for i, cn in enumerate(cnames)})
    if kind == 'ccs':
        cs = buf[layout['cat_seg'][0]:layout['cat_seg'][0] + B + 1]
        meta.update(order=order, rels=rels, max_nodes=int(np.diff(cs).max()),
                    ncap={k: (caps['N'] if caps else counts['N%d' % k]) for k in range(1, order + 1)})
    else:
        sg = buf[layout['seg'][0]:layout['seg'][0] + B + 1]
        meta['max_nodes'] = int(np.diff(sg).max())
    return FlatBatch(torch.from_numpy(buf), layout, meta)


def _annotate_limits(fb):
    """meta['max_deg'] = largest in- / out-degree of a node within one relation (meta['max_nodes'], the longest session's
    node count, is set by the builders): what ops.check_limits holds against the kernels' per-session LDS budgets"""
    md = 0
    buf = fb.buf.numpy()
    for name, (off, cap, _) in fb.layout.items():
        if name.endswith(('in_ptr', 'out_ptr')) and cap > 1:
            md = max(md, int(np.diff(buf[off:off + cap]).max()))
    fb.meta['max_deg'] = md
    return fb


def _attach_labels(fb, lab):
    """append the batch's labels (int32) to the FIRST input's flat buffer as field 'labels': a captured training step then
    receives graph AND labels with one host-to-device copy and reads int32 labels in place (graph.GraphedTrainStep)"""
    off = int(fb.buf.numel())
    n = int(lab.numel())
    pad = (-n) % 4
    ext = torch.cat([fb.buf, lab.to(torch.int32), torch.zeros(pad, dtype=torch.int32)]) if pad else torch.cat([fb.buf, lab.to(torch.int32)])
    layout = dict(fb.layout)
    layout['labels'] = (off, n, None)
    return FlatBatch(ext, layout, fb.meta)


def _attach_labels_inplace(fb, lab, room):
    """the same when the batch lives in a larger caller-owned buffer `room` (int32 numpy array whose head fb.buf views): the
    labels are written behind the batch inside that buffer - no concatenation"""
    off, n = int(fb.buf.numel()), int(lab.numel())
    tot = off + ((n + 3) & ~3)
    if tot > len(room):
        raise ValueError('no room for the labels behind the batch')
    room[off:off + n] = lab.numpy().astype(np.int32)
    room[off + n:tot] = 0
    layout = dict(fb.layout)
    layout['labels'] = (off, n, None)
    return FlatBatch(torch.from_numpy(room[:tot]), layout, fb.meta)


def _labels(labels, caps):
    lab = np.asarray(labels, dtype=np.int64)
    if caps and len(lab) < caps['B']:
        # capacity padding carries label -1: no kernel reads past the live count on one device, and the row-sharded loss
        # (dist.ShardedScoreCE) recognises the padding of a rank's partial batch by it
        lab = np.concatenate([lab, np.full(caps['B'] - len(lab), -1, dtype=np.int64)])
    return torch.as_tensor(lab)


def collate_fn_factory(*seq_to_graph_fns, caps=None):
    """caps=None: exact layouts.  caps={'B','N','E','U'}: capacity-padded layouts with batch-independent
    offsets (every batch then fits the same device buffer and the same captured hipGraph)."""
    kinds = {seq_to_session_graph: 'session', seq_to_eop_multigraph: 'eop', seq_to_shortcut_graph: 'shortcut'}

    def collate_fn(samples):
        seqs, labels = zip(*samples)
        inputs = []
        use = caps
        for attempt in range(2):
            try:
                inputs = []
                for fn in seq_to_graph_fns:
                    fb = collate_native(kinds[fn], seqs, 1, use) if fn in kinds else None
                    if fb is None:
                        fb = batch_homogeneous([fn(s) for s in seqs], use)
                    inputs.append(_annotate_limits(fb))
                lab = _labels(labels, use)
                if use is not None:
                    inputs[0] = _attach_labels(inputs[0], lab)
                return inputs, lab
            except CapacityExceeded:
                if use is None:
                    raise
                FALLBACKS['exact'] += 1
                use = None                       # the batch does not fit the capacities: exact layout (eager step)
    return collate_fn


FALLBACKS = {'exact': 0}       # batches (of this process) that overflowed their capacities and were collated unpadded


def collate_fn_factory_ccs(seq_to_graph_fns, order, caps=None):
    def collate_fn(samples):
        seqs, labels = zip(*samples)
        inputs = []
        use = caps
        for attempt in range(2):
            try:
                inputs = []
                for fn in seq_to_graph_fns:
                    fb = collate_native('ccs', seqs, order, use) if fn is seq_to_ccs_graph else None
                    if fb is None:
                        fb = batch_ccs([fn(s, order) for s in seqs], use)
                    inputs.append(_annotate_limits(fb))
                lab = _labels(labels, use)
                if use is not None:
                    inputs[0] = _attach_labels(inputs[0], lab)
                return inputs, lab
            except CapacityExceeded:
                if use is None:
                    raise
                FALLBACKS['exact'] += 1
                use = None                       # the batch does not fit the capacities: exact layout (eager step)
    return collate_fn


def estimate_caps(dataset, batch_size, headroom=1.15, shuffled=False, slices=None):
    """capacities for capacity-padded batches of `dataset` (an AugmentedDataset: index[:, 1] = prefix length): nodes of any
    order, edges of any relation and distinct items of a batch are all bounded by its total click count, so the cap is
    the largest click count of a run of batch_size consecutive samples (exact for the sequential loaders, typical for
    the shuffled ones) plus headroom, never above the worst case.  A batch that still does not fit is collated
    unpadded (collate_fn_factory*(..., caps) falls back) and simply runs as eager launches."""
    lens = np.asarray(dataset.index[:, 1], dtype=np.int64)
    if len(lens) == 0:
        return default_caps(batch_size)
    if slices is not None:
        # multi-rank training on sequential batches (dataset.RankSliceBatchSampler): the EXACT maximum click count of any
        # rank's slice of any batch of the epoch - slices = (global batch size, world); batch_size = per-rank capacity
        gb, w = slices
        cs = np.concatenate([[0], np.cumsum(lens)])
        best = 0
        for s0 in range(0, len(lens), gb):
            n = min(gb, len(lens) - s0)
            edges = s0 + np.arange(w + 1) * n // w
            best = max(best, int(np.diff(cs[edges]).max()))
            if n < w:                                   # filler samples of ranks with an empty share
                best = max(best, int(lens[s0:s0 + n].max()))
        sums = np.array([best])
    elif shuffled and len(lens) > batch_size:
        # shuffled loaders (NISER / SRGNN): batches are random draws, not runs - take the largest click count over a few
        # hundred random batches (a run of consecutive samples of ONE long session would over- or under-estimate)
        rng = np.random.default_rng(0)
        sums = np.array([lens[rng.choice(len(lens), batch_size, replace=False)].sum() for _ in range(256)])
    else:
        cs = np.concatenate([[0], np.cumsum(lens)])
        starts = np.arange(0, len(lens), batch_size)
        sums = cs[np.minimum(starts + batch_size, len(lens))] - cs[starts]
    n = int(sums.max() * headroom) + 64
    n = min(n, batch_size * int(lens.max()))
    n = (n + 255) // 256 * 256
    return dict(B=batch_size, N=n, E=n, U=n)


def measure_caps(dataset, batch_size, kind, order=1, headroom=1.08, probes=64, shuffled=False):
    """capacities from MEASURED batches: collate (exact layout, native builder) the batches of the epoch that can be the
    largest - the `probes // 4` with the most clicks plus an even sample of the rest - and take the largest node / edge /
    distinct-item counts, with `headroom`.  The click-count bound of estimate_caps is ~1.5 x looser (a batch of c clicks has
    ~0.55 c distinct order-1 nodes), and every kernel of the captured step whose grid follows the CAPACITY pays for it:
    0.963 -> 1.055 ms per step at the C3 shape (N 2560 -> 3840, profiles/r03b).  A batch that still overflows is collated
    in the exact layout and runs as eager launches (collate_fn_factory*), so a tight estimate costs nothing but speed on
    that batch.  kind: 'ccs' (MSGIFSR, with `order`) or 'session' (SRGNN / NISER)."""
    n = len(dataset)
    if n == 0:
        return default_caps(batch_size)
    lens = np.asarray(dataset.index[:, 1], dtype=np.int64)
    nb = (n + batch_size - 1) // batch_size
    if shuffled:
        rng = np.random.default_rng(0)
        groups = [rng.choice(n, min(batch_size, n), replace=False) for _ in range(probes)]
    else:
        cs = np.concatenate([[0], np.cumsum(lens)])
        starts = np.arange(nb) * batch_size
        clicks = cs[np.minimum(starts + batch_size, n)] - cs[starts]
        top = np.argsort(clicks)[::-1][:max(probes // 4, 1)]
        even = np.linspace(0, nb - 1, num=min(nb, probes - len(top))).astype(np.int64)
        ids = sorted(set(top.tolist()) | set(even.tolist()))
        groups = [np.arange(b * batch_size, min(n, (b + 1) * batch_size)) for b in ids]
    mxN = mxE = mxU = 1
    for g in groups:
        seqs = [dataset[int(i)][0] for i in g]
        fb = collate_native(kind, seqs, order, None)
        if fb is None:
            fb = batch_ccs([seq_to_ccs_graph(q, order) for q in seqs]) if kind == 'ccs' else \
                batch_homogeneous([seq_to_session_graph(q) for q in seqs])
        cnt = fb.meta['counts']
        mxN = max([mxN] + [v for k, v in cnt.items() if k.startswith('N') and k != 'NT'])
        mxE = max([mxE] + [v for k, v in cnt.items() if k.startswith('E')])
        mxU = max(mxU, cnt.get('U', 1))
    if shuffled:
        headroom = max(headroom, 1.15)
    up = lambda v: (int(v * headroom) + 32 + 255) // 256 * 256
    return dict(B=batch_size, N=up(mxN), E=up(mxE), U=up(mxU))


def default_caps(batch_size, max_len=20, headroom=1.0):
    """worst-case capacities for sessions of <= max_len clicks"""
    n = int(batch_size * max_len * headroom)
    n = (n + 255) // 256 * 256
    return dict(B=batch_size, N=n, E=n, U=n)
