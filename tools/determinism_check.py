"""Run the same training step twice from the same state and report every gradient / table-gradient tensor that is not
bit-identical (the kernels are meant to be deterministic: ordered reductions, no result atomics)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def main(precision):
    sp = importlib.import_module('sessionrec-pytorch_amd')
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    ops.set_precision(precision)
    dev = torch.device('cuda', 0)
    batches, _ = bench.make_batches('MSGIFSR', 3, 1, 512, 37484, 20, 123, padded=True)
    inp, lab = batches[0]
    inp, lab = [x.to(dev) for x in inp], lab.to(dev)
    torch.manual_seed(123)
    importlib.import_module("sessionrec-pytorch_amd.ops").seed_dropout()   # (same seed value as the previous run: explicit restart)
    model = bench.build_model(sp, 'MSGIFSR', 37484, 256, 3).to(dev)
    model.train()
    res = []
    for rep in range(3):
        model.zero_grad(set_to_none=True)
        loss = model.fused_loss(*inp, lab)
        loss.backward()
        torch.cuda.synchronize()
        g = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        g['__table_grad__'] = model.table_grad.buf.clone()
        g['__loss__'] = loss.detach().clone()
        res.append(g)
    bad = 0
    for k in res[0]:
        for r in (1, 2):
            if not torch.equal(res[0][k], res[r][k]):
                d = (res[0][k] - res[r][k]).abs().max().item()
                print('%s: rep0 vs rep%d differ, max |d| = %.3e (|g| max %.3e)' % (k, r, d, res[0][k].abs().max().item()))
                bad += 1
                break
    print(precision, 'non-deterministic tensors:', bad, 'of', len(res[0]))

for p in sys.argv[1:] or ['fp32', 'bf16']:
    main(p)
