"""Kernel-launching torch ops (not this library's) of ONE eager MSGIFSR C3 training step, with shapes and device time."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity


def main():
    sp = importlib.import_module('sessionrec-pytorch_amd')
    ops, train, optim = (importlib.import_module('sessionrec-pytorch_amd.' + m) for m in ('ops', 'train', 'optim'))
    ops.set_precision('bf16')
    dev = torch.device('cuda', 0)
    batches, _ = bench.make_batches('MSGIFSR', 3, 6, 512, 37484, 20, 123, padded=True)
    batches = [([x.to(dev) for x in inp], lab.to(dev)) for inp, lab in batches]
    torch.manual_seed(123)
    model = bench.build_model(sp, "MSGIFSR", 37484, 256, 3, 0.1).to(dev)
    model.train()
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model)

    def step(b):
        opt.zero_grad()
        loss = model.fused_loss(*b[0], b[1], dynB=b[0][0].dynp('B'))
        loss.backward()
        opt.step()
    for b in batches[:4]:
        step(b)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(batches[4])
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    tot = 0
    for e in rows:
        tot += e.count
        print('%-44s n=%3d  %7.1f us  %s' % (e.key[:44], e.count, e.self_device_time_total, str(e.input_shapes)[:110]))
    print('launching ops:', tot)


main()
