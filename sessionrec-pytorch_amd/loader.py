"""PinnedRingLoader: the training-data path of the launchers without torch's DataLoader machinery in the main process.

The reference feeds its loop from `DataLoader(collate_fn, num_workers=4, pin_memory=True)`
(/root/reference/src/scripts/main_msgifsr.py:148-166, src/utils/train.py:92-95).  Once the training step takes ~1 ms that
loader is bound in the MAIN process: every batch is a ~1 MB tensor that a worker sends over a socket as a file descriptor,
the pin-memory thread maps, unpickles and copies into pinned memory - 1.0 - 1.2 ms per batch whatever the number of workers
(profiles/r03_notes.md).  Here ("emit the flat int32 batch directly into pinned memory"):

  * ONE shared, page-locked ring of batch slots (anonymous shared mapping created before the workers are forked, registered
    with hipHostRegister), slot = capacity of a padded FlatBatch + its labels;
  * worker processes run the native builder (csrc/collate.cpp) so that it writes a batch straight INTO its slot
    (collate.collate_native(into=...)): no pickling of the batch, no copy, no pinning pass;
  * what travels back to the main process is (batch number, slot, element count, layout, meta) - a few hundred bytes;
  * the main process yields FlatBatch views of the pinned slots; the H2D copy of the training step reads them in place
    (graph.GraphedTrainStep._post).  A slot is handed to a worker again only after the copy that read it has completed
    (the stager leaves a HIP event on the batch; anything else that consumes a batch must be done with it by the time
    `slots` further batches have been drawn - the default keeps 8 x workers slots).

Same batches, same order, same FlatBatch contents as `DataLoader(dataset, batch_sampler=..., collate_fn=collate_fn_factory*(..., caps))`
(tests/test_cpu.py::test_pinned_ring_loader_yields_the_dataloader_batches).  Batches that do not fit the capacities are built
in the exact layout (and run as eager steps), like with the collate functions.
"""
import itertools
import mmap
import multiprocessing as mp
import queue as _queue
import time

import numpy as np
import torch

from . import collate as C
from .batch import CapacityExceeded, FlatBatch


class _FlatDataset:
    """an AugmentedDataset (dataset.py: sessions + (session id, prefix length) index) flattened into three numpy arrays BEFORE
    the workers are forked: a worker then gathers a batch's prefixes and labels with a few vectorised numpy calls and never
    touches a Python session object (no per-sample interpreter work, and no copy-on-write faults from reference counts on
    the parent's objects - the first epoch of forked DataLoader workers pays those for every session)"""

    def __init__(self, dataset):
        sessions = dataset.sessions
        lens = np.fromiter(map(len, sessions), dtype=np.int64, count=len(sessions))
        self.soffs = np.zeros(len(sessions) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.soffs[1:])
        self.clicks = np.fromiter(itertools.chain.from_iterable(sessions), dtype=np.int64, count=int(self.soffs[-1]))
        self.dataset = dataset                                           # (.index is read at batch time: it may be re-cut)

    def batch(self, indices):
        idx = np.asarray(indices, dtype=np.int64)
        filler = idx < 0                                                 # (RankSliceBatchSampler: prefix of -idx-1, label -1)
        rows = np.asarray(self.dataset.index)[np.where(filler, -idx - 1, idx)]
        start, ln = self.soffs[rows[:, 0]], rows[:, 1].astype(np.int64)
        offs = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(ln, out=offs[1:])
        pos = np.arange(int(offs[-1]), dtype=np.int64) + np.repeat(start - offs[:-1], ln)
        labels = np.where(filler, -1, self.clicks[start + ln])
        return C.FlatSeqs(self.clicks[pos], offs), labels


def _build(kind, order, caps, dataset, indices, room):
    """one batch into `room` (int32 numpy view of a slot) -> (n_words, layout, meta, labels) or, when the batch does not fit
    the slot (an exact-layout fallback larger than the padded layout), (None, FlatBatch, None, labels): it then travels
    pickled like a DataLoader batch"""
    if isinstance(dataset, _FlatDataset):
        seqs, labels = dataset.batch(indices)
    else:
        seqs, labels = zip(*[dataset[int(i)] for i in indices])
    use = caps
    try:
        fb = C.collate_native(kind, seqs, order, use, into=room)
    except CapacityExceeded:
        C.FALLBACKS['exact'] += 1
        use = None
        try:
            fb = C.collate_native(kind, seqs, order, None, into=room)
        except ValueError:
            fb = C._annotate_limits(C.collate_native(kind, seqs, order, None))
            return None, fb, None, C._labels(labels, None).tolist()
    C._annotate_limits(fb)
    lab = C._labels(labels, use)
    if use is not None:
        fb = C._attach_labels_inplace(fb, lab, room)
    return int(fb.buf.numel()), fb.layout, fb.meta, lab.tolist()


def _augmented_getitem():
    from .dataset import AugmentedDataset
    return AugmentedDataset.__getitem__


def _worker(wid, kind, order, caps, dataset, ring, words, tasks, results):
    torch.set_num_threads(1)
    while True:
        t = tasks.get()
        if t is None:
            return
        gen, k, slot, indices = t
        try:
            room = ring[slot * words:(slot + 1) * words]
            results.put((gen, k, slot) + _build(kind, order, caps, dataset, indices, room))
        except BaseException as e:                           # the main process re-raises
            results.put((gen, k, slot, e))


class PinnedRingLoader:
    def __init__(self, dataset, batch_sampler, kind, order=1, caps=None, num_workers=4, slots=None, pin=True):
        """batch_sampler: a re-iterable of index lists (torch BatchSampler, dataset.RankSliceBatchSampler, ...);
        kind: 'ccs' | 'session' | 'eop' (one graph per sample - LESSR's two-graph batches keep the DataLoader);
        caps: capacities of the padded layout (collate.measure_caps) - required: the slots are sized by them."""
        assert caps is not None and C._native() is not None, 'PinnedRingLoader needs capacities and libsrec_collate.so'
        self.dataset, self.batch_sampler, self.kind, self.order, self.caps = dataset, batch_sampler, kind, order, caps
        self.num_workers = max(1, int(num_workers))
        self.slots = max(4, int(slots) if slots else 8 * self.num_workers)
        # slot size: the padded layout is the same for every batch - build one to learn it (+ labels, + slack for exact-layout
        # fallbacks of slightly larger batches)
        probe = None
        for k, idx in enumerate(batch_sampler):
            try:
                probe = C.collate_native(kind, [dataset[int(i)][0] for i in idx], order, caps)
                break
            except CapacityExceeded:
                if k >= 16:
                    break
        if probe is None:
            raise ValueError('PinnedRingLoader: no batch of the sampler fits the capacities %r' % (caps,))
        self.words = (int(probe.buf.numel()) * 9 // 8 + caps['B'] + 1024 + 1023) // 1024 * 1024
        self._mm = mmap.mmap(-1, self.slots * self.words * 4)            # anonymous + shared: the forked workers see it
        self.ring = np.frombuffer(self._mm, dtype=np.int32)
        self.ring_t = torch.from_numpy(self.ring)
        self.pinned = False
        if pin and torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.ring.ctypes.data, self.ring.nbytes, 0)
            self.pinned = int(rc) == 0
        src = dataset
        if hasattr(dataset, 'sessions') and hasattr(dataset, 'index') and type(dataset).__getitem__ is _augmented_getitem():
            src = _FlatDataset(dataset)
        self._src = src
        ctx = mp.get_context('fork')
        self._tasks, self._results = ctx.Queue(), ctx.Queue()
        self._procs = [ctx.Process(target=_worker, args=(w, kind, order, caps, src, self.ring, self.words, self._tasks,
                                                          self._results), daemon=True) for w in range(self.num_workers)]
        for p in self._procs:
            p.start()
        self.stats = dict(wait_workers_s=0.0, wait_copy_s=0.0)           # where the consumer waited (seconds, cumulative)
        self._busy = {}                                                   # slot -> the FlatBatch last yielded from it
        self._gen = 0                # iteration counter: tasks and results carry it, results of an abandoned iteration are dropped
        self._owed = 0               # tasks handed to the workers whose results have not been taken off the queue yet

    def __len__(self):
        return len(self.batch_sampler)

    def _release(self, slot):
        """the previous tenant of `slot` must have been read: wait for the copy event its consumer left on it"""
        fb = self._busy.pop(slot, None)
        if fb is not None:
            ev = fb.meta.get('_copied')
            if ev is not None and not ev.query():
                t0 = time.perf_counter()
                ev.synchronize()
                self.stats['wait_copy_s'] += time.perf_counter() - t0

    def _drain(self):
        """an iteration abandoned early (break, an exception in the consumer, a worker error re-raised here) leaves tasks with
        the workers: their results must not be taken for batches of the next iteration, and the workers must have stopped
        writing into the ring before its slots are handed out again - wait for every owed result and drop it"""
        while self._owed > 0:
            try:
                self._results.get(timeout=120)
            except _queue.Empty:
                raise RuntimeError('PinnedRingLoader: %d results of an abandoned iteration never arrived' % self._owed)
            self._owed -= 1

    def __iter__(self):
        self._drain()
        self._gen += 1
        gen = self._gen
        it = iter(self.batch_sampler)
        issued = done = 0
        ready, exhausted = {}, False

        def issue():
            nonlocal issued, exhausted
            if exhausted:
                return False
            try:
                idx = next(it)
            except StopIteration:
                exhausted = True
                return False
            slot = issued % self.slots
            self._release(slot)
            self._tasks.put((gen, issued, slot, [int(i) for i in idx]))
            self._owed += 1
            issued += 1
            return True
        for _ in range(self.slots - 2):                    # (two slots stay with the batch in hand and the one before it)
            if not issue():
                break
        while done < issued:
            while done not in ready:
                t0 = time.perf_counter()
                try:
                    r = self._results.get(timeout=120)
                except _queue.Empty:
                    raise RuntimeError('PinnedRingLoader: no batch from the workers for 120 s')
                self.stats['wait_workers_s'] += time.perf_counter() - t0
                self._owed -= 1
                if r[0] != gen:                                           # (cannot happen after _drain; belt and braces)
                    continue
                r = r[1:]
                if isinstance(r[2], BaseException):
                    raise r[2]
                ready[r[0]] = r
            k, slot, n, layout, meta, lab = ready.pop(done)
            if n is None:
                fb = layout                                               # (did not fit the slot: came pickled)
            else:
                fb = FlatBatch(self.ring_t[slot * self.words:slot * self.words + n], layout, meta)
                self._busy[slot] = fb
            done += 1
            issue()
            yield [fb], torch.tensor(lab, dtype=torch.int64)

    def close(self):
        for _ in self._procs:
            self._tasks.put(None)
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._procs = []
        if self.pinned:
            try:
                torch.cuda.cudart().cudaHostUnregister(self.ring.ctypes.data)
            except Exception:
                pass
            self.pinned = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ring_loader_or_none(dataset, batch_sampler, kind, order, caps, num_workers):
    """the launchers' switch: a PinnedRingLoader where it applies (capacity-padded single-graph batches, worker processes,
    the native builder present), None otherwise (the caller then builds the torch DataLoader it always did)"""
    if caps is None or num_workers <= 0 or kind not in ('ccs', 'session', 'eop') or C._native() is None:
        return None
    return PinnedRingLoader(dataset, batch_sampler, kind, order, caps, num_workers)
