"""torch.empty() filled with NaN (torch deterministic mode): any kernel that reads memory it (or a predecessor) did not
write shows up as NaN in the loss / gradients / updated parameters."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True

def main(precision, model_name, padded):
    sp = importlib.import_module('sessionrec-pytorch_amd')
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    train = importlib.import_module('sessionrec-pytorch_amd.train')
    optim = importlib.import_module('sessionrec-pytorch_amd.optim')
    ops.set_precision(precision)
    dev = torch.device('cuda', 0)
    d = 256 if model_name == 'MSGIFSR' else 64
    batches, _ = bench.make_batches(model_name, 3, 3, 512, 37484, 20, 123, padded=padded)
    torch.manual_seed(123)
    model = bench.build_model(sp, model_name, 37484, d, 3).to(dev)
    model.train()
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model)
    for i, (inp, lab) in enumerate(batches):
        inp, lab = [x.to(dev) for x in inp], lab.to(dev)
        opt.zero_grad()
        loss = model.fused_loss(*inp, lab)
        loss.backward()
        bad = [k for k, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        tg = model.table_grad.buf
        opt.step()
        badp = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
        print(precision, model_name, 'padded' if padded else 'exact', 'step', i, 'loss', loss.item(), 'nan grads:', bad[:6],
              'table grad finite:', bool(torch.isfinite(tg).all()), 'nan params:', badp[:6])

for prec in ('fp32', 'bf16'):
    for m in ('MSGIFSR', 'SRGNN', 'NISER', 'LESSR'):
        for padded in (True, False):
            main(prec, m, padded)
