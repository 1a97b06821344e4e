"""Two identical training runs in one process (same batches, same init): per-step losses must be bit-identical if every
kernel is deterministic and race-free.  Reports the first diverging step."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def run(sp, train, optim, G, batches, dev, graph):
    torch.manual_seed(123)
    importlib.import_module("sessionrec-pytorch_amd.ops").seed_dropout()   # (same seed value as the previous run: explicit restart)
    model = bench.build_model(sp, 'MSGIFSR', 37484, 256, 3).to(dev)
    model.train()
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model)
    losses = []
    if graph:
        g = G.GraphedTrainStep(model, opt, batches[0][0], batches[0][1])
        for inp, lab in batches:
            losses.append(g(inp, lab).item())
    else:
        for inp, lab in batches:
            opt.zero_grad()
            loss = model.fused_loss(*inp, lab)
            loss.backward()
            opt.step()
            losses.append(loss.item())
    return losses

def main():
    sp = importlib.import_module('sessionrec-pytorch_amd')
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    train = importlib.import_module('sessionrec-pytorch_amd.train')
    optim = importlib.import_module('sessionrec-pytorch_amd.optim')
    G = importlib.import_module('sessionrec-pytorch_amd.graph')
    ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    dev = torch.device('cuda', 0)
    batches, _ = bench.make_batches('MSGIFSR', 3, n, 512, 37484, 20, 123, padded=True)
    batches = [([x.to(dev) for x in inp], lab.to(dev)) for inp, lab in batches]
    runs = {}
    for name, graph in (('eager1', False), ('eager2', False), ('graph1', True), ('graph2', True)):
        runs[name] = run(sp, train, optim, G, batches, dev, graph)
    for a, b in (('eager1', 'eager2'), ('graph1', 'graph2'), ('eager1', 'graph1')):
        first = next((i for i, (x, y) in enumerate(zip(runs[a], runs[b])) if x != y), None)
        print(a, 'vs', b, ': first differing step', first, '' if first is None else (runs[a][first], runs[b][first]),
              'final', runs[a][-1], runs[b][-1])

main()
