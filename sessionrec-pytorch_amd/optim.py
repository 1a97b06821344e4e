"""FusedAdam: torch.optim.Adam semantics (coupled L2 weight decay, bias correction, parameters
with no gradient skipped) executed by the HIP kernels of csrc/adam.hip.

Mirrors the optimizer the reference's TrainRunner builds (train.py:70-75): param groups from
fix_weight_decay, lr driven by torch's StepLR (it subclasses torch.optim.Optimizer so the
scheduler works unchanged).  The item table takes the row-structured kernel, fed from the
TableGrad buffer the fused scoring backward filled, with the Embedding(max_norm) renorm and the
cosine column scale of the NEXT step fused into the same pass over HBM.
"""
import torch

from ._lib import lib, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, model=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.model = model
        self._hyper = {}

    def _hyper_dev(self, group, step, device):
        b1, b2 = group['betas']
        key = (id(group), step)
        vals = [group['lr'] / (1.0 - b1 ** step), b1, b2, group['eps'], group['weight_decay'], 1.0 - b1, 1.0 - b2,
                (1.0 - b2 ** step) ** 0.5]
        slot = self._hyper.get(id(group))
        if slot is None:
            slot = self._hyper[id(group)] = {}
        ent = slot.get('buf')
        if ent is None:
            host = torch.empty(8, dtype=torch.float32).pin_memory() if device.type == 'cuda' else torch.empty(8)
            slot['host'] = host
        # one tiny device tensor per distinct step value inside this call (dead params keep older steps)
        t = torch.tensor(vals, dtype=torch.float32, device=device)
        return t

    @torch.no_grad()
    def step(self, closure=None):
        model = self.model
        table = None
        tgrad = None
        if model is not None and hasattr(model, '_table'):
            table = model._table()
            st = model.__dict__.get('_srec_state')
            if st is not None and st.get('tgrad') is not None and st['tgrad'].fresh:
                tgrad = st['tgrad']
        zero_ids = {}
        if model is not None and hasattr(model, 'zero_grad_params'):
            zero_ids = {id(p): p for p in model.zero_grad_params()}
        for group in self.param_groups:
            hyper_cache = {}
            use_wd = 1 if group['weight_decay'] != 0 else 0
            for p in group['params']:
                is_table = table is not None and p is table
                g = p.grad
                if is_table and tgrad is not None:
                    g = tgrad.buf if g is None else g.add_(tgrad.buf)
                if g is None and id(p) in zero_ids:
                    g = torch.zeros_like(p)          # zero (not None) gradient in the reference: decay only
                if g is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['step'] += 1
                s = state['step']
                if s not in hyper_cache:
                    hyper_cache[s] = self._hyper_dev(group, s, p.device)
                hyper = hyper_cache[s]
                g = g.contiguous()
                if is_table and p.dim() == 2 and (p.shape[1] & 3) == 0:
                    max_norm = 0.0     # Embedding(max_norm) renorm stays in forward (reference state after step())
                    cos = model._cosine() if hasattr(model, '_cosine') else None
                    st = model.__dict__.get('_srec_state')
                    cs_out, cs_scale, eps_mode = None, 1.0, 0
                    if cos is not None and st is not None and st.get('cs') is not None:
                        cs_out, cs_scale, eps_mode = st['cs'], float(cos[0]), int(cos[1])
                    lib.srec_adam_rows(ptr(p), ptr(g), ptr(state['exp_avg']), ptr(state['exp_avg_sq']), p.shape[0],
                                       p.shape[1], p.stride(0), ptr(hyper), use_wd, float(max_norm), ptr(cs_out),
                                       cs_scale, eps_mode, 1e-12, stream())
                    if cs_out is not None:
                        st['cs_fresh'] = True
                else:
                    lib.srec_adam_flat(ptr(p), ptr(g), ptr(state['exp_avg']), ptr(state['exp_avg_sq']), p.numel(),
                                       ptr(hyper), use_wd, stream())
        if tgrad is not None:
            tgrad.fresh = False
        return None
