"""FusedAdam: torch.optim.Adam semantics (coupled L2 weight decay, bias correction, parameters
with no gradient skipped) executed by the HIP kernels of csrc/adam.hip.

Mirrors the optimizer the reference's TrainRunner builds (train.py:70-75): param groups from
fix_weight_decay, lr driven by torch's StepLR (it subclasses torch.optim.Optimizer so the
scheduler works unchanged).  The item table takes the row-structured kernel, fed from the
TableGrad buffer the fused scoring backward filled, with the cosine column scale of the NEXT
step fused into the same pass over HBM.

Graph-friendly: the step counter lives on the DEVICE and is advanced by the launch sequence itself
(srec_adam_hyper: counter += 1, lr / bias corrections in double like torch), so a captured hipGraph
of `launch()` replays correctly however far the host runs ahead; `advance()` is host bookkeeping
plus a stream-ordered config copy when the lr schedule changed a group.
"""
import ctypes as _ct

import torch

from ._lib import lib, ptr, stream


class _AdamMultiDesc(_ct.Structure):
    """srec_adam_multi_desc (include/srec.h)"""
    _fields_ = [('nt', _ct.c_int), ('use_wd', _ct.POINTER(_ct.c_int)), ('numel', _ct.POINTER(_ct.c_long)),
                ('p', _ct.POINTER(_ct.c_void_p)), ('g', _ct.POINTER(_ct.c_void_p)), ('m', _ct.POINTER(_ct.c_void_p)),
                ('v', _ct.POINTER(_ct.c_void_p))]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, model=None, fuse_projection=False):
        """fuse_projection (cosine-scored models; one device or a row-sharded table): the chain rule of the catalog-row normalisation is applied
        to the table gradient inside this optimizer's row pass instead of a separate pass after the scoring backward
        (ops.TableGrad: defer / pending / radial)."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.model = model
        if fuse_projection and model is not None and hasattr(model, '_cosine') and model._cosine() is not None \
                and model._table().is_cuda:
            model._defer_projection = True
            shard = getattr(model, 'shard', None)
            if shard is None:
                model.__dict__.pop('_srec_state', None)      # the gradient buffer is re-created with the side array
            elif getattr(shard.local, 'fused_dropout', False):   # row-sharded table (HIP kernels): its gradient buffer is
                tg = shard.tgrad                                 # shared with the collectives - switched over in place
                tg.defer = True
                if tg.radial is None:
                    tg.radial = torch.zeros(tg.buf.shape[0], device=tg.buf.device, dtype=torch.float32)
        self._hyper = {}            # (group index, step offset) -> device step state
        self._frozen = None         # parameter lists frozen by a captured graph

    def folds_table_prep(self):
        """True when launch() will leave the model's table renormalised / with its bf16 operand copy written (see launch)"""
        m = self.model
        if m is None or not getattr(m, '_fold_table_prep', False) or not hasattr(m, '_table'):
            return False
        W = m._table()
        return bool(W.is_cuda and W.dim() == 2 and (W.shape[1] & 3) == 0 and W.shape[1] <= 1024 and W.is_contiguous()
                    and any(p is W for g in self.param_groups for p in g['params']))

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        st = self.model.__dict__.get('_srec_state') if self.model is not None else None
        if st is not None and st.get('tgrad') is not None:
            st['tgrad'].reset()             # incl. a deferred projection / radial sums of a backward that was never stepped

    # ------------------------------------------------------------------ helpers
    def _table_info(self):
        model = self.model
        table, tgrad, st = None, None, None
        if model is not None and hasattr(model, '_table'):
            table = model._table()
            st = model.__dict__.get('_srec_state')
            if st is not None and st.get('tgrad') is not None and st['tgrad'].fresh:
                tgrad = st['tgrad']
        return table, tgrad, st

    grad_override = None      # {id(param): tensor}: gradients to use instead of p.grad (the all-reduced flat bucket)

    def _grad_of(self, p, table, tgrad, zero_ids):
        g = p.grad
        if self.grad_override and id(p) in self.grad_override:
            g = self.grad_override[id(p)]      # all-reduced bucket view: present even when THIS rank's batch gave no gradient
        if table is not None and p is table and tgrad is not None:
            if g is not None:
                tgrad.materialize()           # a deferred projection applies to the scoring gradient only
            g = tgrad.buf if g is None else g.add_(tgrad.buf)
        if g is None and id(p) in zero_ids:
            # zero (not None) gradient in the reference: weight decay only.  One static all-zero buffer per parameter
            # (never written), not a fill per step.
            cache = self.__dict__.setdefault('_zero_grads', {})
            g = cache.get(id(p))
            if g is None or g.shape != p.shape or g.device != p.device:
                g = cache[id(p)] = torch.zeros_like(p)
        return g

    def _buffers(self, gi, off, device):
        """device-resident step state of the parameters of group gi whose step count is (global step - off):
        (counter int32[1], cfg double[5], hyper float[8]).  The counter lives on the device and is advanced by the
        captured step itself (srec_adam_hyper), so a replayed graph or a host running ahead cannot mix steps."""
        key = (gi, off)
        if key not in self._hyper:
            from . import ops
            self._hyper[key] = dict(counter=torch.zeros(1, dtype=torch.int32, device=device),
                                    cfg=torch.zeros(5, dtype=torch.float64, device=device),
                                    hyper=torch.zeros(8, dtype=torch.float32, device=device), cfg_host=None,
                                    fresh=True)
            if key == (0, 0):
                # the dropout masks of a captured step are keyed by this optimizer's device-side step count (ops.rng_args):
                # registered on the MODEL, which installs it at the start of each of its forwards - two optimizers in one
                # process (two models) each drive their own model's masks
                if self.model is not None:
                    self.model.__dict__['_srec_rng_counter'] = self._hyper[key]['counter']
                ops.RNG_COUNTER[str(device)] = self._hyper[key]['counter']
        return self._hyper[key]

    def _cfg(self, group):
        b1, b2 = group['betas']
        return (float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']))

    def _vals(self, group, step):
        b1, b2 = group['betas']
        return [group['lr'] / (1.0 - b1 ** step), b1, b2, group['eps'], group['weight_decay'], 1.0 - b1, 1.0 - b2,
                (1.0 - b2 ** step) ** 0.5]

    def reset_steps(self):
        """forget every step taken (host and device counters) - used after a graph-capture warm-up"""
        self._T = 0
        for st in self.state.values():
            if 'step' in st:
                st['step'] = 0
        for ent in self._hyper.values():
            if isinstance(ent, dict):
                ent['counter'].zero_()
                ent['fresh'] = False

    def snapshot(self):
        """moments and step counts as they are now (GraphedTrainStep: undo of its eager warm-up steps)"""
        return dict(T=getattr(self, '_T', 0),
                    state={p: (st['step'], st['exp_avg'].clone(), st['exp_avg_sq'].clone())
                           for p, st in self.state.items() if 'exp_avg' in st})

    def restore(self, snap):
        """back to a snapshot(): parameters that had no state then start from zero moments and step 0"""
        self._T = snap['T']
        with torch.no_grad():
            for p, st in self.state.items():
                if 'exp_avg' not in st:
                    continue
                if p in snap['state']:
                    st['step'] = snap['state'][p][0]
                    st['exp_avg'].copy_(snap['state'][p][1])
                    st['exp_avg_sq'].copy_(snap['state'][p][2])
                else:
                    st['step'] = 0
                    st['exp_avg'].zero_()
                    st['exp_avg_sq'].zero_()
            for key, ent in self._hyper.items():
                ent['counter'].fill_(max(self._T - key[1], 0))       # steps taken by the parameters of this offset slot
                ent['fresh'] = False

    def load_state_dict(self, state_dict):
        """torch's loader + the device-side step state: counters are rebuilt from the restored per-parameter step
        counts the next time advance() meets each (group, offset) slot."""
        super().load_state_dict(state_dict)
        steps = [int(st['step']) for st in self.state.values() if 'step' in st]
        for st in self.state.values():
            if 'step' in st:
                st['step'] = int(st['step'])
        self._T = max(steps) if steps else 0
        self._hyper = {}
        self._frozen = None

    def _work(self):
        """[(group idx, group, [(p, g, state)])] for every parameter that has a gradient now"""
        from . import ops
        ops.flush_deferred()                  # (normally already done by the end-of-backward callback)
        table, tgrad, _ = self._table_info()
        zero_ids = {}
        if self.model is not None and hasattr(self.model, 'zero_grad_params'):
            zero_ids = {id(p): p for p in self.model.zero_grad_params()}
        work = []
        for gi, group in enumerate(self.param_groups):
            items = []
            for p in group['params']:
                g = self._grad_of(p, table, tgrad, zero_ids)
                if g is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                items.append((p, g, state))
            work.append((gi, group, items))
        return work

    # ------------------------------------------------------------------ the two halves of step()
    def advance(self, work=None):
        """host only: bump the (bookkeeping) step counters; refresh a group's device config when it changed (lr
        schedule) - a stream-ordered copy from pageable memory, never a pinned buffer the GPU reads later"""
        if work is None:
            work = self._frozen
        self._T = getattr(self, '_T', 0) + 1
        for gi, group, items in work:
            offs = set()
            for p, g, state in items:
                state['step'] += 1
                offs.add(self._T - state['step'])
            cfg = self._cfg(group)
            for off in offs:
                ent = self._buffers(gi, off, items[0][0].device)
                if ent['fresh']:
                    # a parameter that skipped steps (grad None) moves to a new offset: the device counter of that
                    # offset starts where the parameter's own count stands (this step brings it to T - off)
                    ent['counter'].fill_(self._T - off - 1)
                    ent['fresh'] = False
                if ent['cfg_host'] != cfg:
                    ent['cfg'].copy_(torch.tensor(cfg, dtype=torch.float64))
                    ent['cfg_host'] = cfg
        return work

    def hyper_rider(self, tap=None, skip=None):
        """Before the backward pass of a CAPTURED step: hand the step-scalar launch of the coming launch() to the end-of-backward
        slab-sum launch (ops.PENDING_HYPER -> srec_sum_slabs_multi_hyper: one workgroup more there, one ~6 us graph node less).
        The slots are those the eager warm-up step advanced (_used_keys); launch() checks that the rider advanced exactly the slots
        of its work list and raises otherwise (the capture is then refused, nothing has run).  Without waiting slab sums the
        rider is never taken and launch() runs its own kernel as always."""
        from . import ops
        del ops.PENDING_HYPER[:], ops.HYPER_DONE[:]
        keys = sorted(getattr(self, '_used_keys', ()) or ())
        if not keys or len(keys) > 16 or not isinstance(getattr(self, '_hyper', None), dict):
            return False
        ents = []
        for gi, off in keys:
            ent = self._hyper.get((gi, off))
            if ent is None or ent.get('cfg_host') is None:
                return False
            ents.append(ent)
        n = len(ents)
        arr = _ct.c_void_p * n
        cs, cf, hy = (arr(*[e[k].data_ptr() for e in ents]) for k in ('counter', 'cfg', 'hyper'))
        first = self._hyper.get((0, 0))
        if tap is not None and first is not None and any(e is first for e in ents):
            args = (n, _ct.addressof(cs), _ct.addressof(cf), _ct.addressof(hy), ptr(first['counter']), ptr(tap[0]), ptr(tap[1]),
                    tap[1].numel(), ptr(skip))
        else:
            args = (n, _ct.addressof(cs), _ct.addressof(cf), _ct.addressof(hy), None, None, None, 0, ptr(skip))
        ops.PENDING_HYPER[:] = [(args, frozenset(keys), (cs, cf, hy, tap, skip))]          # (last: kept alive until the launch)
        return True

    def launch(self, work, tap=None, skip=None):
        """device work only (capturable): the step-scalar kernel + the Adam kernels.
        tap = (loss scalar, ring): the step's loss goes to ring[(step count before it) % len] (graph.GraphedTrainStep's loss
        ring; given per call, by the captured launch only - never optimizer state that outlives its graph);
        skip = device int32: non-zero turns the whole step into the identity (srec_adam_hyper_multi; the batch-intake fault
        flag of a captured step)."""
        table, tgrad, st = self._table_info()
        model = self.model
        T = getattr(self, '_T', 0)
        # step scalars of every (group, step offset) slot of this step: one launch
        ents, used = [], set()
        for gi, group, items in work:
            for off in sorted({T - state['step'] for _, _, state in items}):
                ent = self._buffers(gi, off, items[0][0].device)
                if ent['cfg_host'] is None:                     # first use outside advance(): step() on a fresh group
                    ent['cfg'].copy_(torch.tensor(self._cfg(group), dtype=torch.float64))
                    ent['cfg_host'] = self._cfg(group)
                ents.append(ent)
                used.add((gi, off))
        self._used_keys = used                                  # (group, step offset) slots this step advanced
        from . import ops as _ops
        done = _ops.HYPER_DONE.pop() if _ops.HYPER_DONE else None
        del _ops.PENDING_HYPER[:], _ops.HYPER_DONE[:]             # (a rider nobody took: this launch does the work itself)
        if done is not None and done != frozenset(used):
            raise RuntimeError('the step scalars were advanced for other slots than this step uses (hyper_rider): %r vs %r'
                               % (sorted(done), sorted(used)))
        for i in range(0, len(ents) if done is None else 0, 16):
            chunk = ents[i:i + 16]
            n = len(chunk)
            arr = _ct.c_void_p * n
            cs, cf, hy = (arr(*[e[k].data_ptr() for e in chunk]) for k in ('counter', 'cfg', 'hyper'))
            first = self._hyper.get((0, 0)) if isinstance(self._hyper, dict) else None
            if tap is not None and first is not None and any(e is first for e in chunk):
                lib.srec_adam_hyper_multi(n, _ct.addressof(cs), _ct.addressof(cf), _ct.addressof(hy), ptr(first['counter']),
                                          ptr(tap[0]), ptr(tap[1]), tap[1].numel(), ptr(skip), stream())
            else:
                lib.srec_adam_hyper_multi(n, _ct.addressof(cs), _ct.addressof(cf), _ct.addressof(hy), None, None, None, 0,
                                          ptr(skip), stream())
        # small tensors of ALL groups that step together: groups which differ only in weight decay (fix_weight_decay: the
        # biases / norms are the same Adam with wd 0) share ONE multi-tensor launch - the kernel takes the decay per tensor
        merged = {}                          # (lr, betas, eps, slot) -> [gi of the hyper to use, wd of it, rows]
        for gi, group, items in work:
            if not items:
                continue
            use_wd = 1 if group['weight_decay'] != 0 else 0
            slot_of = {}
            for _, _, state in items:
                slot_of.setdefault(state['step'], T - state['step'])
            multi = {}                       # step slot -> rows of the multi-tensor descriptor
            for p, g, state in items:
                hyper = self._buffers(gi, slot_of[state['step']], p.device)['hyper']
                g = g.contiguous()
                is_table = table is not None and p is table
                if not is_table and p.is_contiguous() and p.numel() < (1 << 22):
                    multi.setdefault(slot_of[state['step']], []).append((p, g, state))
                    continue
                if is_table and p.dim() == 2 and (p.shape[1] & 3) == 0:
                    cos = model._cosine() if hasattr(model, '_cosine') else None
                    cs_out, cs_scale, eps_mode = None, 1.0, 0
                    if cos is not None and st is not None and st.get('cs') is not None:
                        cs_out, cs_scale, eps_mode = st['cs'], float(cos[0]), int(cos[1])
                    # Embedding(max_norm) renorm (lessr.py:126, msgifsr.py:162: in the reference the NEXT forward's lookups do
                    # it, before anything reads a row) and the table's bf16 operand copy for the bf16 scoring kernels are
                    # written by THIS pass - the row is in registers - when the model folds them (`_fold_table_prep`): the
                    # next forward's stand-alone renorm + copy pass (srec_renorm_rows_bf16) is then skipped.  Observable
                    # difference: between a step and the next forward the stored rows are already renormalised (a
                    # checkpoint taken there holds what the reference's next forward would have made of them).
                    mn_model = float(getattr(model, '_max_norm', 0.0) or 0.0)
                    mn = mn_model if cs_out is not None else 0.0
                    fold = st is not None and self.folds_table_prep()    # ONE predicate: graph.py pre-prepares the table by it
                    tb = model._table_copy(st, p) if (fold and hasattr(model, '_table_copy')) else None
                    renorm_write = 1 if (fold and mn_model > 0) else 0
                    if renorm_write:
                        mn = mn_model
                    d16, dp16 = (tb.E16, tb.E16.shape[1]) if tb is not None else (None, 0)
                    # a row-sharded table is padded to whole scoring tiles: the rows past the shard's live count are zero,
                    # get zero gradients and stay zero under Adam with coupled decay - the pass stops at the live rows
                    shard = getattr(model, 'shard', None)
                    n_rows = shard.n_live if (shard is not None and 0 < shard.n_live < p.shape[0]) else p.shape[0]
                    pend = tgrad.pending if (tgrad is not None and g.data_ptr() == tgrad.buf.data_ptr()) else None
                    if pend is not None:
                        lib.srec_adam_rows_proj(ptr(p), ptr(g), ptr(state['exp_avg']), ptr(state['exp_avg_sq']), n_rows,
                                                p.shape[1], p.stride(0), ptr(hyper), use_wd, mn, renorm_write, ptr(cs_out), cs_scale,
                                                eps_mode, 1e-12, ptr(pend[1]), float(pend[2]), ptr(tgrad.radial), ptr(d16), dp16,
                                                stream())
                        tgrad.pending = None
                        tgrad.radial_dirty = False
                    else:
                        if tgrad is not None:
                            tgrad.materialize()
                        lib.srec_adam_rows(ptr(p), ptr(g), ptr(state['exp_avg']), ptr(state['exp_avg_sq']), n_rows,
                                           p.shape[1], p.stride(0), ptr(hyper), use_wd, mn, renorm_write, ptr(cs_out), cs_scale,
                                           eps_mode, 1e-12, ptr(d16), dp16, stream())
                    if fold:
                        # (renormalised rows in place, bf16 copy written): consumed by the model's next _prepare_table / _table_bf16
                        st['table_prepared'] = (bool(renorm_write) or mn_model <= 0, tb is not None, p._version)
                    if cs_out is not None:
                        st['cs_fresh'] = True
                else:
                    self._join_grads()
                    lib.srec_adam_flat(ptr(p), ptr(g), ptr(state['exp_avg']), ptr(state['exp_avg_sq']), p.numel(),
                                       ptr(hyper), use_wd, stream())
            for slot, rows in multi.items():
                b1, b2 = group['betas']
                key = (float(group['lr']), float(b1), float(b2), float(group['eps']), slot, str(rows[0][0].device))
                wd = float(group['weight_decay'])
                ent = merged.get(key)
                if ent is not None and wd != 0 and ent[1] != 0 and ent[1] != wd:
                    key = key + (wd,)                # two different non-zero decays: separate launches
                    ent = merged.get(key)
                if ent is None:
                    ent = merged[key] = [gi, wd, []]
                elif wd != 0 and ent[1] == 0:
                    ent[0], ent[1] = gi, wd          # the launch reads the hyper-parameters of the group that decays
                ent[2] += [(p, g, st_, use_wd) for p, g, st_ in rows]
        self._join_grads()                           # (the table's row pass above needs none of the replicated gradients)
        for key, (gi, _, rows) in merged.items():    # every small tensor that steps together in ONE launch (by-value descriptor)
            slot = key[4]
            nt = len(rows)
            arr = (_ct.c_void_p * nt)
            a_wd, a_n = (_ct.c_int * nt)(*[r[3] for r in rows]), (_ct.c_long * nt)(*[r[0].numel() for r in rows])
            a_p, a_g = arr(*[r[0].data_ptr() for r in rows]), arr(*[r[1].data_ptr() for r in rows])
            a_m, a_v = arr(*[r[2]['exp_avg'].data_ptr() for r in rows]), arr(*[r[2]['exp_avg_sq'].data_ptr() for r in rows])
            desc = _AdamMultiDesc(nt, a_wd, a_n, a_p, a_g, a_m, a_v)
            hyper = self._buffers(gi, slot, rows[0][0].device)['hyper']
            lib.srec_adam_multi(_ct.addressof(desc), ptr(hyper), stream())
        if tgrad is not None:
            tgrad.fresh = False
        from . import ops
        ops.weights_changed()

    grad_join = None          # callable: make the all-reduced replicated gradients (grad_override) visible to the compute stream -
    #                           dist.VocabParallel leaves the join of its side stream to the optimizer, which calls it AFTER the
    #                           table's row pass (independent of every collective) and before the first kernel that reads them

    def _join_grads(self):
        j, self.grad_join = self.grad_join, None
        if j is not None:
            j()

    @torch.no_grad()
    def step(self, closure=None):
        work = self._work()
        self.advance(work)
        self.launch(work)
        return None
