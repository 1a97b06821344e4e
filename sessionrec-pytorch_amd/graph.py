"""Whole-step hipGraph: the ~130 kernel launches of one training step (gather -> encoder -> fused
scoring/CE forward+backward -> fused Adam) are captured ONCE on capacity-padded, statically placed
tensors and replayed per batch.  Per step the host then does: one H2D copy of the FlatBatch buffer
into the static device buffer, host bookkeeping of the step counts (the counter the kernels use lives on
the device and is advanced by the captured step itself), one hipGraphLaunch.

This is what makes the launch-bound step GPU-bound: every kernel reads its live extents from the
batch header in device memory (`dyn*` arguments of include/srec.h), so the same launch sequence is
valid for every batch that fits the capacities.  (MI355X guide: capture launch-bound inner loops in
hipGraphs; no tracing compiler involved - the graph is the recorded launch sequence of our own C ABI.)
"""
import os

import torch

from .batch import FlatBatch


# How a replayed step takes a host batch: GraphedTrainStep._post (side-stream copy into a device ring, host-side wait, mailbox).
# History of the alternatives, loop time per step: profiles/r03_notes.md ("gap legs"), profiles/r04_notes.md.
LOSS_RING = 512        # slots of the device loss ring of a captured step (GraphedTrainStep.loss_ring, .last_T)
_STAGE_SLOTS = 16      # device staging buffers per input for host-fed batches (a slot is rewritten 16 replays later)
_MAILBOX = 64          # entries of the batch mailbox of a captured step (replays the host may run ahead of the GPU)


class GraphedTrainStep:
    def __init__(self, model, optimizer, inputs, labels, after_backward=None, warmup=2):
        """inputs: list of capacity-padded FlatBatch on the GPU (their layout fixes the graph), labels (B,).

        No tensor that carries the autograd graph of an EARLIER forward of `model` may be alive when this is called (a loss
        kept in a variable, an activation stored on a module): such a graph keeps the parameters' AccumulateGrad nodes -
        and the stream they were created on, typically the default stream - alive, autograd then synchronises the
        capturing stream with that stream inside the capture, and hipStreamEndCapture crashes (PyTorch warns about
        exactly this: "may ... break CUDA graph capture if the AccumulateGrad node's stream is the default stream").
        `del loss` / `.detach()` what you keep.  The warm-up below and the capture share ONE side stream, so graphs the
        warm-up leaves behind are harmless."""
        import gc
        gc.collect()                                     # reference cycles that still hold an earlier forward's graph
        assert all(x.meta.get('padded') for x in inputs), 'graph capture needs capacity-padded batches (collate caps=...)'
        self.model, self.opt, self.after_backward = model, optimizer, after_backward
        self.static_inputs = [FlatBatch(x.buf.clone(), x.layout, dict(x.meta)) for x in inputs]
        # the kernels read int32 labels.  Padded batches carry them inside the first input's flat buffer (collate
        # _attach_labels): the static labels are then a VIEW of the static batch buffer - no second copy, no int64 -> int32
        # conversion kernel per step
        self.labels_in_batch = self.static_inputs[0].has('labels') and self.static_inputs[0].cap('labels') == labels.numel()
        self.static_labels = (self.static_inputs[0].field('labels') if self.labels_in_batch else labels.to(torch.int32))
        self._one = torch.ones((), device=labels.device, dtype=torch.float32)   # backward root: no fill kernel per replay
        self._sig = [self._signature(x) for x in inputs]
        # ---- eager warm-up on a side stream (lazy allocations, LDS opt-in attributes, Adam state), then undo
        #      its effect on parameters / optimizer state so capture does not change the training trajectory
        snap_o = optimizer.snapshot()                    # a resumed / already running optimizer keeps its moments
        snap_p = [p.detach().clone() for p in model.parameters()]
        snap_b = [b.detach().clone() for b in model.buffers()]
        T0 = getattr(optimizer, '_T', 0)
        s = self._stream = torch.cuda.Stream()
        try:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
        finally:
            # whatever happened in the warm-up (including an exception half way): parameters, buffers (BatchNorm running
            # statistics) and optimizer state go back to where they were, so an eager fallback continues the SAME trajectory
            self._restore(model, optimizer, snap_p, snap_b, snap_o)
        # ---- capture
        optimizer.zero_grad(set_to_none=True)
        # cosine-scored models: every replayed step takes its column scale 12 / ||E_v|| from the fused Adam pass of the step
        # before.  Compute the FIRST one here, outside the graph, so the captured step does not carry a row_invnorm pass it
        # would then replay (and overwrite Adam's result with the same values) for ever.
        ms = model.__dict__.get('_srec_state')
        if ms is not None and getattr(model, '_cosine', lambda: None)() is not None and hasattr(model, '_col_scale'):
            with torch.no_grad():
                if hasattr(model, '_prepare_table'):
                    model._prepare_table()               # Embedding(max_norm) renorm first: the scale is that of the rows read
                ms['cs_fresh'] = False
                model._col_scale(ms)
            ms['cs_fresh'] = ms.get('cs') is not None
        # the same for the table's renorm / bf16 operand copy: FusedAdam's row pass of step k prepares the table for step k + 1
        # (optim.py), so the captured step carries no stand-alone pass - the FIRST replay's table is prepared here, eagerly
        ms = model.__dict__.get('_srec_state')
        if ms is not None and getattr(optimizer, 'folds_table_prep', lambda: False)() and hasattr(model, '_table_copy'):
            with torch.no_grad():
                W = model._table()
                mn = float(getattr(model, '_max_norm', 0.0) or 0.0)
                tb = model._table_copy(ms, W)
                ms.pop('table_prepared', None)
                from ._lib import lib as _lib, ptr as _ptr, stream as _stream
                if tb is not None:
                    tb.refresh(W, mn)
                elif mn > 0:
                    _lib.srec_renorm_rows(_ptr(W), W.stride(0), None, W.shape[0], None, W.shape[1], mn, _stream())
                ms['table_prepared'] = (True, tb is not None, W._version)
        import os
        try:                                             # the hipGraph_t stays queryable (node_counts)
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        except TypeError:
            self.graph = torch.cuda.CUDAGraph()
        self.loss_ring, self.last_T = None, None
        self._pending_advance = None
        self._setup_mailbox(optimizer, labels.device)
        if self._mb is not None:                         # (allocated outside the capture: not part of the replayed step)
            self.loss_ring = torch.zeros(LOSS_RING, device=labels.device, dtype=torch.float32)
        work = None
        from . import dist as _dist
        self.capture_mode = self._capture_mode()
        self._quiesce_collectives()
        c0 = dict(_dist.STATS)
        err = None
        try:
            with torch.cuda.graph(self.graph, stream=self._stream, capture_error_mode=self.capture_mode):
                try:
                    self._capture_intake()
                    self.loss = self.model.fused_loss(*self.static_inputs, self.static_labels)
                    from . import ops as _ops
                    if _ops.PENDING_INTAKE:
                        raise RuntimeError('the model never launched the batch intake of its captured step (ops.flush_intake)')
                    # the tap and the intake's fault flag are arguments of THIS capture's launches only (see below); the
                    # optimizer's step-scalar kernel rides in the end-of-backward slab-sum launch when there is one
                    tap = (self.loss.detach(), self.loss_ring) if self._mb is not None else None
                    skip = self._mb['err'] if self._mb is not None else None
                    if hasattr(optimizer, 'hyper_rider') and os.environ.get('SREC_HYPER_RIDER', '1') != '0':
                        optimizer.hyper_rider(tap, skip)
                    self.loss.backward(self._one)
                    if self.after_backward is not None:
                        self.after_backward()
                    work = optimizer._work()
                    optimizer._frozen = work
                    # every replay leaves its loss in slot (step count % LOSS_RING) of a device ring (written by the
                    # optimizer's step-scalar kernel): the caller reads losses in bulk, when it wants them, instead of
                    # cloning the static loss tensor between two graph launches (train.py:99-104 reads it every step).
                    # The tap and the intake's fault flag are arguments of THIS launch only: nothing of a graph stays behind
                    # in the optimizer (an eager step after a refused capture would write into a dead graph's tensors)
                    # the step counters that matter live on the device and are advanced by the captured step itself; the
                    # host-side bookkeeping is bumped before every replay (advance()): bump once here for a consistent
                    # capture and take it back afterwards
                    optimizer.advance(work)
                    optimizer.launch(work, tap=tap, skip=skip)
                    if self._mb is not None and (0, 0) not in getattr(optimizer, '_used_keys', ()):
                        # the mailbox and the loss ring are keyed by the (group 0, offset 0) step counter: a step that does
                        # not advance it would look at entry 0 for ever
                        raise RuntimeError('captured step does not advance the optimizer step counter the batch mailbox is keyed by')
                except BaseException as e:               # leave the capture context normally and raise afterwards: the undo
                    err = e                              # below must not run while the stream is still capturing
                    if os.environ.get('SREC_DEBUG_CAPTURE'):
                        import traceback
                        traceback.print_exception(type(e), e, e.__traceback__)
            if err is not None:
                raise err
        except BaseException:
            # capture refused: nothing ran on the device, but host bookkeeping may be half advanced - undo it all
            optimizer._frozen = None
            self._restore(model, optimizer, snap_p, snap_b, snap_o)
            optimizer.zero_grad(set_to_none=True)
            from . import ops as _ops
            del _ops.PENDING_INTAKE[:]                   # (an intake the dead capture never launched)
            del _ops.PENDING_HYPER[:], _ops.HYPER_DONE[:]
            sh = getattr(model, 'shard', None)
            if sh is not None and hasattr(sh, 'abort_step'):
                sh.abort_step()                          # early gradient buckets recorded by the dead capture never ran
            if hasattr(optimizer, 'grad_join'):
                optimizer.grad_join = None
            raise
        # collectives captured inside the step (row-sharded table): count and payload bytes of ONE step
        self.collectives = {k: _dist.STATS[k] - c0[k] for k in c0}
        for _, _, items in work:
            for _, _, st in items:
                st['step'] -= 1
        optimizer._T -= 1
        assert optimizer._T == T0, (optimizer._T, T0)
        self.work = work
        if hasattr(self.graph, 'instantiate'):
            try:
                self.graph.instantiate()
            except Exception:
                pass

    @staticmethod
    def _nccl_group():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            try:
                return dist.get_backend() == 'nccl'
            except Exception:
                return False
        return False

    @classmethod
    def _capture_mode(cls):
        """hipStreamCaptureMode of the step capture.  RCCL through torch.distributed: every eager collective of the warm-up
        left a work item with the process group's WATCHDOG THREAD, which polls their completion events every ~100 ms until it
        has retired them.  Under the default 'global' mode an event query from that thread while this thread captures is an
        illegal call: it invalidated the capture or aborted the process in 1 run of ~10 of `bench.py --shard`
        (profiles/r05_notes.md).  'thread_local' confines the legality check to the capturing thread - what the watchdog thread
        does no longer invalidates the capture (the first of two failure modes; the second one: _quiesce_collectives).
        SREC_CAPTURE_MODE overrides (global | thread_local | relaxed)."""
        m = os.environ.get('SREC_CAPTURE_MODE')
        if m in ('global', 'thread_local', 'relaxed'):
            return m
        return 'thread_local' if cls._nccl_group() else 'global'

    @classmethod
    def _quiesce_collectives(cls):
        """The second failure mode of the watchdog race, which the capture mode does NOT remove: every eager collective of the
        warm-up left a work item whose completion EVENT the watchdog queries until it has retired the item - and HIP refuses
        the query of an event whose stream is capturing by now ("operation not permitted on an event last recorded in a capturing
        stream", hipErrorCapturedEvent: the exception ends the watchdog thread and aborts the process; 4 runs of 34 of
        `SREC_BUCKET_SIDE_STREAM=1 bench.py --shard` with thread_local capture and no wait, tools/shard_loop.sh).  There is no
        handle on the watchdog's list; it wakes on a 100 ms timer (and on every new work item, which it cannot retire yet).  So:
        synchronise - every work item is complete - and wait three timer periods: the list is empty when the capture begins.
        SREC_CAPTURE_SLEEP=<seconds> overrides (0: no wait)."""
        if cls._nccl_group():
            torch.cuda.synchronize()
            t = float(os.environ.get('SREC_CAPTURE_SLEEP', '0.3') or 0)
            if t > 0:
                import time
                time.sleep(t)

    def _setup_mailbox(self, optimizer, device):
        """Batch intake as the FIRST kernels of the captured step: each reads where this replay's batch lives from a mailbox
        in page-locked host memory (index = the optimizer's device-side step count mod _MAILBOX, filled by __call__ before
        the launch) and copies it into the static buffer - no copy command in front of the graph launch (a device-to-device
        copy + the gap behind it cost ~12 us per step).  Needs FusedAdam's device counter and 16-byte granular buffers."""
        self._mb = None
        ent = getattr(optimizer, '_hyper', {}).get((0, 0)) if isinstance(getattr(optimizer, '_hyper', None), dict) else None
        # (the eager warm-up has just run the step: _used_keys says which counters a step of this model advances)
        if ent is None or (0, 0) not in getattr(optimizer, '_used_keys', ()) or any(x.buf.numel() % 4 for x in self.static_inputs):
            return
        n = len(self.static_inputs)
        box = torch.zeros(n, _MAILBOX, 4, dtype=torch.int32).pin_memory()
        box[:, :, 3] = -1                                  # (no entry carries a valid counter value yet)
        self._mb = dict(box=box, np=box.numpy(), counter=ent['counter'], err=torch.zeros(1, dtype=torch.int32, device=device),
                        events=[], held=[], calls=0)

    def check(self):
        """raise if a replayed step found another step count in its mailbox entry than the host wrote (the step, and every
        step after it, then ran on a stale batch and changed NOTHING: the optimizer's step-scalar kernel turns such a step
        into the identity, csrc/adam.hip).  One 4-byte readback: called wherever the host synchronises anyway (TrainRunner's
        loss flush, end of an epoch, the end of bench.py's loops), and every 512th replay."""
        if self._mb is not None and int(self._mb['err'].item()):
            raise RuntimeError('a replayed step found another step count in its batch mailbox than the host wrote: that step and '
                               'every later one were skipped (parameters and optimizer state are those before it)')

    def _capture_intake(self):
        """the intake kernels of the captured step: handed to the model, whose first launch takes one along (ops.step_prologue:
        one node for the intake and the step's weight copies) or launches them ahead of its first read of the batch
        (ops.flush_intake in the models' lookup)"""
        if self._mb is None:
            return
        from . import ops
        ops.PENDING_INTAKE[:] = [(self._mb['box'][i].data_ptr(), _MAILBOX, self._mb['counter'].data_ptr(), st.buf.data_ptr(),
                                  st.buf.numel(), self._mb['err'].data_ptr(), self._mb) for i, st in enumerate(self.static_inputs)]

    def _post(self, i, x, T):
        """entry T % _MAILBOX of input i <- (address, words, T); the batch buffer must stay untouched until the replay has
        read it: device tensors are kept alive here.  A HOST batch goes through a ring of _STAGE_SLOTS device buffers first:
        the H2D copy runs on a side stream - i.e. under the replays the GPU is still working on, the host being ahead of it -
        and the host waits for it (tens of us) before it queues this replay, so the captured step finds its batch in HBM
        and nothing in front of the graph launch waits across streams.  (The intake kernel reading the page-locked slot over
        PCIe itself put ~35 us of transfer at the head of every step: end to end 0.887 against 0.855 ms of replay.)"""
        buf = x.buf
        if not buf.is_cuda or buf.numel() % 4 or buf.data_ptr() % 16:
            stg = self.__dict__.setdefault('_stg', dict(ring={}, marks=[], stream=None, used=False))
            if i not in stg['ring']:
                stg['ring'][i] = [torch.empty_like(self.static_inputs[i].buf) for _ in range(_STAGE_SLOTS)]
            if stg['stream'] is None:
                stg['stream'] = torch.cuda.Stream(device=self.static_inputs[i].buf.device)
            # slot T % _STAGE_SLOTS was read by replay T - _STAGE_SLOTS: a mark (an event every 4th replay) at or behind it
            marks = stg['marks']
            if marks and marks[-1][0] >= T:                  # the step counter went back (a restored checkpoint)
                marks.clear()
                torch.cuda.current_stream().synchronize()
            while marks and marks[0][0] < T - _STAGE_SLOTS:
                marks.pop(0)
            if marks:
                marks[0][1].synchronize()                    # T - 16 <= its replay < T: normally long complete
            elif T >= _STAGE_SLOTS:
                torch.cuda.current_stream().synchronize()    # (staging starts in the middle of a run: no mark yet)
            slot = stg['ring'][i][T % _STAGE_SLOTS]
            n = min(buf.numel(), slot.numel())
            if buf.is_cuda:                                  # (a misaligned device batch: its producer may still be writing it)
                stg['stream'].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stg['stream']):
                slot[:n].copy_(buf[:n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stg['stream'])
            ev.synchronize()
            x.meta['_copied'] = ev                           # the host buffer may be rewritten (it already has been read)
            stg['used'] = True
            buf = slot
        if i == 0:
            self._mailbox_lap(T)
        a = buf.data_ptr()
        lo, hi = a & 0xffffffff, (a >> 32) & 0xffffffff
        self._mb['np'][i, T % _MAILBOX] = (lo - ((lo & 0x80000000) << 1), hi - ((hi & 0x80000000) << 1),
                                           min(buf.numel(), self.static_inputs[i].buf.numel()), T)
        return buf

    def _mailbox_lap(self, T):
        """entry T % _MAILBOX was read by replay T - _MAILBOX: that replay must have run before the host rewrites the entry.
        Keyed by the STEP COUNT (eager steps between replays advance T without recording events): done_T = newest step known
        complete; otherwise the oldest recorded event at or behind step T - _MAILBOX is waited for (normally long complete)."""
        mb = self._mb
        ev, need = mb['events'], T - _MAILBOX
        if mb.get('last_T', -1) >= T:                        # the step counter went back (a restored checkpoint)
            torch.cuda.current_stream().synchronize()
            ev.clear()
            mb['done_T'] = T - 1
        elif need >= 0 and mb.get('done_T', -1) < need:
            k = next((j for j, e in enumerate(ev) if e[2] >= need), None)
            if k is not None:
                ev[k][0].synchronize()
                mb['done_T'] = ev[k][2]
                del ev[:k + 1]
            elif mb.get('last_T', -1) >= need:               # replays in reach without an event behind them
                torch.cuda.current_stream().synchronize()
                mb['done_T'] = T - 1
        mb['last_T'] = T

    def node_counts(self):
        """{'kernel': n, 'memcpy': n, 'memset': n, 'other': n} of the captured step (hipGraphGetNodes on the raw hipGraph_t)
        or None when the handle is not available: the launch count of one training step, measured, not estimated"""
        import ctypes
        try:
            raw = self.graph.raw_cuda_graph()
            hip = ctypes.CDLL('libamdhip64.so')
            n = ctypes.c_size_t(0)
            if hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n)) != 0:
                return None
            nodes = (ctypes.c_void_p * n.value)()
            if hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n)) != 0:
                return None
            out = dict(kernel=0, memcpy=0, memset=0, other=0)
            for nd in nodes:
                t = ctypes.c_int(-1)
                hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t))
                out[{0: 'kernel', 1: 'memcpy', 2: 'memset'}.get(t.value, 'other')] += 1
            return out
        except Exception:
            return None

    @staticmethod
    def _restore(model, optimizer, snap_p, snap_b, snap_o):
        with torch.no_grad():
            for p, q in zip(model.parameters(), snap_p):
                p.copy_(q)
            for b, q in zip(model.buffers(), snap_b):
                b.copy_(q)
        optimizer.restore(snap_o)                        # moments, host and device step counters
        ms = model.__dict__.get('_srec_state')
        if ms is not None:
            ms['cs_fresh'] = False
            if ms.get('tgrad') is not None:
                ms['tgrad'].reset()                      # fresh flag, pending projection, radial sums of the warm-up

    @staticmethod
    def _signature(x):
        # (the layout enters as the dict itself: comparing two dicts is one C-level pass, sorting ~60 items into a tuple for
        # every batch was ~20 us of host time per step)
        rels = tuple(x.count('E_' + n) > 0 for _, n in x.meta.get('rels', ()))
        return (x.layout, rels, x.meta.get('kind'), x.meta.get('order'))

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.model.fused_loss(*self.static_inputs, self.static_labels)
        loss.backward()
        if self.after_backward is not None:
            self.after_backward()
        self.opt.step()
        return loss

    def _stage(self, i, x):
        """host batch buffer -> the static device buffer of input i WITHOUT the mailbox (an optimizer that keeps no device
        step counter): the H2D copy runs on a side stream into a small ring of device staging buffers and the compute stream
        does a device-to-device copy; a staging slot is rewritten only after the compute stream has consumed it.  The batch
        carries an event (meta['_copied']) after which its host buffer may be rewritten."""
        st = self.static_inputs[i]
        ring = self.__dict__.setdefault('_ring', {})
        if i not in ring:
            ring[i] = dict(bufs=[torch.empty_like(st.buf) for _ in range(3)], used=[None] * 3, n=0,
                           stream=torch.cuda.Stream(device=st.buf.device))
        r = ring[i]
        j = r['n'] % 3
        r['n'] += 1
        main = torch.cuda.current_stream()
        with torch.cuda.stream(r['stream']):
            if r['used'][j] is not None:
                r['stream'].wait_event(r['used'][j])       # the replay that read this slot has taken its copy
            r['bufs'][j][:x.buf.numel()].copy_(x.buf, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(r['stream'])
        x.meta['_copied'] = ready                          # the host buffer may be rewritten once this has completed
        main.wait_event(ready)
        st.buf.copy_(r['bufs'][j], non_blocking=True)
        done = torch.cuda.Event()
        done.record(main)
        r['used'][j] = done

    def __call__(self, inputs, labels):
        """inputs: capacity-padded FlatBatches with the captured layout, on the device or still on the host (see _post)"""
        from . import ops
        mb = self._mb
        T = getattr(self.opt, '_T', 0)                   # the device step counter's value when this replay starts
        keep = []
        for i, (st, x, sig) in enumerate(zip(self.static_inputs, inputs, self._sig)):
            if self._signature(x) != sig:
                raise RuntimeError('batch layout / relation pattern differs from the captured one')
            ops.check_limits(x)          # the kernels clamp to their per-session budgets: an oversized session is an error
            if mb is not None:
                keep.append(self._post(i, x, T))
            elif x.buf.is_cuda:
                st.buf.copy_(x.buf, non_blocking=True)
            else:
                self._stage(i, x)
            st.meta['counts'] = x.meta['counts']
        if not (self.labels_in_batch and inputs[0].has('labels')):
            self.static_labels.copy_(labels, non_blocking=True)
        self.opt.advance(self.work)
        self.graph.replay()
        self.last_T = T                                  # this replay's loss: loss_ring[T % LOSS_RING] (when there is a ring)
        if mb is not None:
            # The replay has been queued.  Its batch buffers may be rewritten once it has run, and the host must not lap the
            # mailbox (entry T is rewritten _MAILBOX replays later) nor a staging slot of host-fed batches: an event marks the
            # spot every 8th replay (every 4th while batches come from the host; an event record is a command of its own
            # between two graph launches)
            mb['calls'] += 1
            mb['held'].append(keep)
            stg = self.__dict__.get('_stg')
            staged = stg is not None and stg['used']
            if staged:
                stg['used'] = False
            if mb['calls'] % 8 == 0 or (staged and T % 4 == 0):
                done = torch.cuda.Event()
                done.record()
                if staged:
                    stg['marks'].append((T, done))
                ev = mb['events']
                ev.append((done, mb['held'], T))
                mb['held'] = []
                while ev and ev[0][0].query():
                    mb['done_T'] = max(mb.get('done_T', -1), ev[0][2])
                    ev.pop(0)
            if mb['calls'] % 512 == 0:
                self.check()
        return self.loss


def capture_agreed(make, agree_min=None, retries=1, log=None):
    """GraphedTrainStep construction for a job of several ranks: `make()` builds (warms up + captures) the step or raises;
    `agree_min(x)` returns the minimum of x over the ranks (None: one rank).  The ranks decide TOGETHER after every attempt:
    if any rank's capture failed, every rank drops its graph, and - because the constructor's eager warm-up steps issue the
    step's collectives - every rank retries together (`retries` times), so the collective sequences stay paired; after the
    last attempt all ranks replay or all ranks launch eagerly.  -> (step or None, attempts made, last local error or None)"""
    err = None
    for attempt in range(retries + 1):
        gs, e = None, None
        try:
            gs = make()
        except Exception as ex:                          # capture refused on THIS rank (the constructor has undone its effects)
            e = ex
            if log is not None:
                log('hipGraph capture failed on this rank, attempt %d (%s: %s)' % (attempt + 1, type(ex).__name__, str(ex)[:300]))
        ok = 1.0 if gs is not None else 0.0
        if agree_min is not None:
            ok = float(agree_min(ok))
        if ok >= 1.0:
            return gs, attempt + 1, None
        err = e if e is not None else err
        gs = None                                        # (another rank failed: this rank's graph is dropped too)
    return None, retries + 1, err
