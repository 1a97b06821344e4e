"""Where does an end-to-end training step of the launcher surface spend its wall time?  (loader wait / H2D / step / sync)
usage: python tools/e2e_phases.py [workers] [steps]   (dataset: /tmp/synth_yc from tools/e2e_launcher.sh)"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sp = importlib.import_module('sessionrec-pytorch_amd')
ds, col, train, ops = (importlib.import_module('sessionrec-pytorch_amd.' + m) for m in ('dataset', 'collate', 'train', 'ops'))
from torch.utils.data import DataLoader, SequentialSampler

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
SYNC_EVERY = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device('cuda:0')
tr, te, V = ds.read_dataset('/tmp/synth_yc')
data = ds.AugmentedDataset(tr)
caps = col.estimate_caps(data, 512)
cf = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), 3, caps=caps)
loader = DataLoader(data, batch_size=512, num_workers=W, collate_fn=cf, sampler=SequentialSampler(data))
torch.manual_seed(0)
model = sp.MSGIFSR(V, 'x', 256, 1, dropout=0.1, order=3, extra=False, fusion=False).to(dev)
runner = train.TrainRunner('x', model, loader, [], dev, lr=1e-3, weight_decay=1e-4)
model.train()
t = dict(wait=0.0, h2d=0.0, step=0.0, sync=0.0)
it = iter(loader)
for i in range(steps + 20):
    if i == 20:
        t = dict(wait=0.0, h2d=0.0, step=0.0, sync=0.0)
        t_all = time.perf_counter()
    a = time.perf_counter(); batch = next(it)
    b = time.perf_counter(); inputs, labels = train.prepare_batch(batch, dev)
    c = time.perf_counter(); loss = runner.train_step(inputs, labels)
    d = time.perf_counter()
    if SYNC_EVERY == 1 or i % SYNC_EVERY == 0:
        loss.item()
    e = time.perf_counter()
    t['wait'] += b - a; t['h2d'] += c - b; t['step'] += d - c; t['sync'] += e - d
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print('workers %d: %.2f ms/step  ' % (W, tot / steps * 1e3) + '  '.join('%s %.2f' % (k, v / steps * 1e3) for k, v in t.items()),
      ' graph steps', runner.graph_steps, ' sync every', SYNC_EVERY)
g = runner._gstep
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100):
    g(inputs, labels)                    # (through the mailbox: a raw graph.replay() would find a stale entry and skip the step)
torch.cuda.synchronize()
g.check()
print('pure replay of the captured step: %.2f ms' % ((time.perf_counter() - t0) * 10), ' caps', caps)
# collate alone, one process
import timeit
samples = [data[i] for i in range(512)]
print('collate (1 process): %.2f ms/batch;  fetch 512 samples: %.2f ms' % (
    timeit.timeit(lambda: cf(samples), number=50) / 50 * 1e3, timeit.timeit(lambda: [data[i] for i in range(512)], number=50) / 50 * 1e3))
