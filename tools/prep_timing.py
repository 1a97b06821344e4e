"""Role-by-role timing of the step's prologue launch (csrc/prep.hip, srec_step_prep) at the benchmarked shapes: subsets of roles
as launches of their own, warm (10 rounds) and behind a 512 MB fill (10 rounds), read from a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python tools/prep_timing.py; durations of step_prep_kernel by grid size"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module('sessionrec-pytorch_amd.ops')
dev = 'cuda:0'
D, H = 256, 8
torch.manual_seed(0)
w16 = [torch.randn(H * D, D, device=dev) for _ in range(8)]
gru = [torch.randn(3 * D, D, device=dev) for _ in range(4)]
head = [(torch.randn(D, D, device=dev), 0), (torch.randn(D, D, device=dev), 0), (torch.randn(D, 2 * D, device=dev), 0),
        (torch.randn(D, 2 * D, device=dev), 1)]


def prologue(**kw):
    ops.weights_changed()
    ops.step_prologue(kw.get('w16', ()), kw.get('gru', ()), kw.get('head', ()), None)


# (run under rocprofv3 --kernel-trace: the durations of the step_prep_kernel launches, by grid size, are the answer)
cold = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for rep in range(20):
    for kw in (dict(head=head), dict(w16=w16), dict(gru=gru), dict(w16=w16, gru=gru, head=head), dict(w16=w16[:4]),
               dict(w16=w16, gru=gru)):
        if rep >= 10:
            cold.zero_()            # 512 MB through the caches: the weights come from HBM
        prologue(**kw)
torch.cuda.synchronize()
