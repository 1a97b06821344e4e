"""FlatBatch: the DGL-free batched-session carrier (replaces the dgl.DGLGraph /
heterograph objects that collate.py:219-256 of the reference hands to the models).

All index data of one batch lives in ONE int32 buffer (one pinned host allocation,
one H2D copy per step), laid out as

    [ header: 32 live counts | field 0 | field 1 | ... ]     every field 16-byte aligned

Fields are addressed by name and are views into the buffer.  Each field has a
CAPACITY (its allocated length); the live extent of every variable-size dimension
is a header slot, so kernels can read it from device memory (`dyn` pointers of
include/srec.h) and one captured hipGraph replays for batches of any size.  In the
default "exact" layout capacity == live count.

Like a DGLGraph it supports `.to(device)` (train.py:29 `x.to(device)`) and pickles
across DataLoader workers.
"""
import numpy as np
import torch

HEADER = 32


class CapacityExceeded(ValueError):
    """a batch does not fit the capacities of its padded layout (the collate factories then emit the exact layout and
    the step runs as eager launches).  An exception, not an assert: `python -O` must not write past a capacity."""


def _align4(n):
    return (n + 3) & ~3


class FlatBatch:
    def __init__(self, buf, layout, meta):
        self.buf = buf                # torch.int32 1-D
        self.layout = layout          # name -> (offset, capacity, shape or None)
        self.meta = meta              # python-side: kind, order, B, counts{...}, slots{...}
        self._views = {}

    # ---- construction -------------------------------------------------------------------
    @staticmethod
    def build(fields, counts, meta, caps=None):
        """fields: name -> 1-D/2-D int array (live data); counts: name -> int (header values);
        caps: optional name -> capacity (elements) for the padded layout."""
        layout, off = {}, HEADER
        for name, arr in fields.items():
            arr = np.asarray(arr)
            cap = int(arr.size)
            if caps is not None and name in caps:
                cap = max(cap, int(caps[name]))
            layout[name] = (off, cap, tuple(arr.shape[1:]) if arr.ndim > 1 else None)
            off += _align4(max(cap, 1))
        buf = np.zeros(off, dtype=np.int32)
        slots = {}
        for i, (name, val) in enumerate(counts.items()):
            assert i < HEADER, 'too many header counts'
            slots[name] = i
            buf[i] = val
        for name, arr in fields.items():
            o, cap, _ = layout[name]
            a = np.asarray(arr).reshape(-1)
            if a.size > cap:
                raise CapacityExceeded('field %s: %d elements, capacity %d' % (name, a.size, cap))
            buf[o:o + a.size] = a
            if a.size < cap:
                # capacity padding: offset arrays repeat their last value (empty segments), index arrays -1
                if name.endswith('ptr') or name.startswith(('seg', 'eseg', 'cat_seg')):
                    buf[o + a.size:o + cap] = a[-1] if a.size else 0
                elif name.startswith(('iid', 'gidx', 'uniq_items', 'uniq_inv', 'cat_perm', 'last')):
                    buf[o + a.size:o + cap] = -1
        m = dict(meta)
        m['counts'] = {k: int(v) for k, v in counts.items()}
        m['slots'] = slots
        return FlatBatch(torch.from_numpy(buf), layout, m)

    # ---- graph-object surface ---------------------------------------------------------------
    def to(self, device, non_blocking=False):
        if torch.device(device) == self.buf.device:
            return self
        # a pinned buffer (DataLoader(pin_memory=True)) is copied asynchronously: the host goes on to the next batch
        pinned = self.buf.is_pinned()
        out = FlatBatch(self.buf.to(device, non_blocking=non_blocking or pinned), self.layout, self.meta)
        if pinned:
            ev = torch.cuda.Event()                    # (a slot of loader.PinnedRingLoader is reused only after this copy)
            ev.record()
            self.meta['_copied'] = ev
        return out

    def pin_memory(self):
        return FlatBatch(self.buf if self.buf.is_pinned() else self.buf.pin_memory(), self.layout, self.meta)

    @property
    def device(self):
        return self.buf.device

    def __getstate__(self):
        return (self.buf, self.layout, self.meta)

    def __setstate__(self, st):
        self.buf, self.layout, self.meta = st
        self._views = {}

    # ---- access ----------------------------------------------------------------------------------
    def has(self, name):
        return name in self.layout

    def field(self, name, live=False):
        """int32 view of a field (capacity extent; live=True trims to the live count when known)."""
        key = (name, live)
        if key not in self._views:
            off, cap, shape = self.layout[name]
            v = self.buf[off:off + cap]
            if shape is not None:
                v = v.view(-1, *shape)
            self._views[key] = v
        return self._views[key]

    def __getattr__(self, name):
        if name.startswith('_') or name in ('buf', 'layout', 'meta'):
            raise AttributeError(name)
        if name in self.layout:
            return self.field(name)
        raise AttributeError(name)

    def count(self, name):
        return self.meta['counts'][name]

    def cap(self, name):
        return self.layout[name][1]

    def dyn(self, name):
        """1-element int32 device view holding the live count `name` (for `dyn*` kernel arguments)."""
        s = self.meta['slots'][name]
        return self.buf[s:s + 1]

    def dynp(self, name):
        """`dyn(name)` for capacity-padded batches, None for exact layouts (kernels then use the static extent)."""
        return self.dyn(name) if self.meta.get('padded') else None

    @property
    def B(self):
        return self.meta['B']

    def batch_num_nodes(self, k=1):
        seg = self.field('seg%d' % k if self.meta['kind'] == 'ccs' else 'seg')
        return (seg[1:self.B + 1] - seg[:self.B]).long()
