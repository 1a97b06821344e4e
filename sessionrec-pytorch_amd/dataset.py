"""Session files and prefix augmentation - host-side mirror of
/root/reference/src/utils/data/dataset.py (create_index :6-13, read_sessions :16-19,
read_dataset :22-27, AugmentedDataset :29-50).

On-disk format unchanged: train.txt / test.txt hold one session per line, comma-separated item
ids; num_items.txt one integer.  (The reference's pandas `squeeze=` call no longer exists in
pandas 2; plain parsing gives the same lists.)

SessionStore is the binary form of the same data for the large splits (Yoochoose-1/4 = 6 M
samples): all clicks in one int32 array + int64 session offsets (CSR), memory-mapped on load, with the prefix index
built by vectorised numpy instead of Python lists.  `read_dataset(dir, cache=True)` writes / reuses
`<split>.sstore.npz` next to the text files; every consumer (AugmentedDataset, collate) sees the same sequences.
"""
from pathlib import Path

import numpy as np


class SessionStore:
    """CSR container of click sessions: `items` int32 [total clicks], `offsets` int64 [n_sessions + 1].
    Behaves like the reference's object array of lists (`len`, integer indexing -> the session's item ids)."""

    def __init__(self, items, offsets):
        self.items, self.offsets = items, offsets

    @classmethod
    def from_sessions(cls, sessions):
        lens = np.fromiter((len(s) for s in sessions), dtype=np.int64, count=len(sessions))
        offsets = np.zeros(len(sessions) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        items = np.empty(int(offsets[-1]), dtype=np.int32)
        for i, s in enumerate(sessions):
            items[offsets[i]:offsets[i + 1]] = s
        return cls(items, offsets)

    @classmethod
    def from_text(cls, filepath):
        """one session per line, comma separated (dataset.py:16-19) - parsed without building Python int lists"""
        with open(filepath, 'rb') as f:
            raw = f.read()
        lines = [ln for ln in raw.split(b'\n') if ln.strip()]
        lens = np.fromiter((ln.count(b',') + 1 for ln in lines), dtype=np.int64, count=len(lines))
        offsets = np.zeros(len(lines) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        items = np.array(b','.join(lines).split(b','), dtype=np.int64).astype(np.int32) if lines else np.empty(0, np.int32)
        assert items.size == offsets[-1]
        return cls(items, offsets)

    def save(self, path):
        np.savez(path, items=self.items, offsets=self.offsets)

    @classmethod
    def load(cls, path, mmap=True):
        z = np.load(path, mmap_mode='r' if mmap else None)
        return cls(z['items'], z['offsets'])

    def lengths(self):
        return np.diff(self.offsets)

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i):
        return self.items[self.offsets[i]:self.offsets[i + 1]]

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def create_index(sessions):
    """one sample per (session, label position >= 1), session-major: columns sessionId, labelIndex"""
    if isinstance(sessions, SessionStore):
        lens = sessions.lengths()
    else:
        lens = np.fromiter((len(s) for s in sessions), dtype=np.int64, count=len(sessions))
    reps = np.maximum(lens - 1, 0)
    session_idx = np.repeat(np.arange(len(sessions)), reps)
    starts = np.cumsum(reps) - reps
    label_idx = np.arange(int(reps.sum())) - np.repeat(starts, reps) + 1
    return np.column_stack((session_idx, label_idx))


def read_sessions(filepath):
    out = []
    with open(filepath) as f:
        for line in f:
            line = line.strip()
            if line:
                out.append([int(x) for x in line.split(',')])
    arr = np.empty(len(out), dtype=object)
    arr[:] = out
    return arr


def _read_split(dataset_dir, split, cache):
    txt, binp = dataset_dir / (split + '.txt'), dataset_dir / (split + '.sstore.npz')
    if not cache:
        return read_sessions(txt)
    if binp.exists() and binp.stat().st_mtime >= txt.stat().st_mtime:
        return SessionStore.load(binp)
    store = SessionStore.from_text(txt)
    try:
        store.save(binp)
    except OSError:
        pass                                  # read-only dataset directory: keep the in-memory store
    return store


def read_dataset(dataset_dir, cache=False):
    """(train_sessions, test_sessions, num_items); cache=True -> SessionStore objects backed by `<split>.sstore.npz`"""
    dataset_dir = Path(dataset_dir)
    train_sessions = _read_split(dataset_dir, 'train', cache)
    test_sessions = _read_split(dataset_dir, 'test', cache)
    with open(dataset_dir / 'num_items.txt', 'r') as f:
        num_items = int(f.readline())
    return train_sessions, test_sessions, num_items


class AugmentedDataset:
    def __init__(self, sessions, sort_by_length=False):
        self.sessions = sessions
        index = create_index(sessions)
        if sort_by_length:
            index = index[np.argsort(index[:, 1])[::-1]]
        self.index = index

    def __getitem__(self, idx):
        if idx < 0:
            # filler sample of a multi-rank batch slice (RankSliceBatchSampler): the prefix of sample -idx-1 with label -1,
            # which the sharded loss leaves out of the mean (dist.ShardedScoreCE) - the reference never sees it
            sid, lidx = self.index[-idx - 1]
            return self.sessions[sid][:lidx], -1
        sid, lidx = self.index[idx]
        seq = self.sessions[sid]
        return seq[:lidx], seq[lidx]

    def __len__(self):
        return len(self.index)


class RankSliceBatchSampler:
    """Multi-rank training on the reference's batches (strong scaling): batches are formed from `sampler` exactly as the
    single-device DataLoader forms them (`batch_size` consecutive draws, last partial batch kept: main_lessr.py:84-93,
    main_msgifsr.py:148-157, main_niser.py:83-92) and rank r of w yields positions [r n / w, (r+1) n / w) of each
    n-sample batch - the loss is then the mean over the SAME samples as on one device.  A rank whose share of a tiny
    last batch is empty gets one filler index (negative: AugmentedDataset returns it with label -1 = left out of the
    loss) so that every rank takes part in every step's collectives.  `sampler` must produce the same order on every
    rank (SequentialSampler, or a RandomSampler with an identically seeded generator)."""

    def __init__(self, sampler, batch_size, rank, world, prefix_len=None, need_len=None):
        """prefix_len (optional): array of the click count of every sample (AugmentedDataset.index[:, 1]); need_len: the
        click count a GLOBAL batch must reach somewhere for every relation of the model's schema to have an edge in it
        (MSGIFSR: order + 1).  HeteroGraphConv skips a relation without edges in the batch (msgifsr.py:60-64) and a rank
        sees only its slice, so the multi-rank path treats every relation as live (msgifsr.MSHGNN.plan all_rels); a global
        batch that breaks the assumption - in practice a tiny last batch of 1-2-click prefixes - is COUNTED here
        (`short_batches`, reported by the launcher): on it the multi-rank step adds the residual and bias of the edgeless
        relations where the single-device step would not."""
        self.sampler, self.batch_size, self.rank, self.world = sampler, batch_size, rank, world
        self.prefix_len, self.need_len = prefix_len, need_len
        self.short_batches = 0

    def _check(self, batch):
        if self.prefix_len is not None and self.need_len is not None and \
                int(np.asarray(self.prefix_len)[np.asarray(batch)].max()) < self.need_len:
            self.short_batches += 1

    def __len__(self):
        return (len(self.sampler) + self.batch_size - 1) // self.batch_size

    def slice_of(self, batch):
        n, r, w = len(batch), self.rank, self.world
        mine = batch[r * n // w:(r + 1) * n // w]
        return mine if mine else [-batch[r % n] - 1]

    def __iter__(self):
        batch = []
        for i in self.sampler:
            batch.append(int(i))
            if len(batch) == self.batch_size:
                self._check(batch)
                yield self.slice_of(batch)
                batch = []
        if batch:
            self._check(batch)
            yield self.slice_of(batch)

    def per_rank_capacity(self):
        """sessions a rank can receive in one step"""
        return (self.batch_size + self.world - 1) // self.world
