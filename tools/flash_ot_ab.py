"""Development A/B of the bf16 flash-CE launches with one / two owner tiles per wave (SREC_FLASH_OT): timings by HIP events
over `iters` back-to-back launches and a bit / norm comparison of the outputs between the two settings.
usage (GPU box): python tools/flash_ot_ab.py            (spawns itself once per setting)"""
import importlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)


def child(tag):
    import torch
    import bench
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    sp = importlib.import_module('sessionrec-pytorch_amd')
    ops.set_precision('bf16')
    dev = torch.device('cuda:0')
    B, V, d = 512, 37484, 256
    torch.manual_seed(123)
    model = bench.build_model(sp, 'SRGNN', V, d, 1).to(dev)
    table = model._table().detach()[:V]
    sr = torch.nn.functional.normalize(torch.randn(B, d, device=dev), dim=1)
    cs = (12.0 / table.norm(dim=1)).contiguous()
    labels = torch.randint(0, V, (B,), device=dev, dtype=torch.int32)
    ws = ops.CEWorkspace(B, V, d, dev)
    lse, lossvec, loss = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty((), device=dev)
    dE, dsr = torch.empty_like(table), torch.empty(B, d, device=dev)
    tb = ops.TableBF16(table).refresh(table)
    fwd = lambda: ops._ce_fwd(sr, table, cs, labels, ws, None, tb, ws.lab_logit, lse, lossvec, loss)
    bwd = lambda: ops._ce_bwd(sr, table, cs, labels, lse, None, None, None, ws, None, tb, dE, dsr, 3)
    fwd(); bwd(); torch.cuda.synchronize()
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print('%s %s: %.1f us per call (incl. its reduce launches)' % (tag, name, e0.elapsed_time(e1) * 20))
    torch.save({'lse': lse.cpu(), 'loss': loss.cpu(), 'dE': dE.cpu(), 'dsr': dsr.cpu()}, '/tmp/flash_ot_%s.pt' % tag)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    import torch
    for fo, bo in ((1, 1), (2, 2), (1, 1), (2, 2)):
        env = dict(os.environ, SREC_FLASH_OT=str(bo))
        subprocess.check_call([sys.executable, __file__, 'ot%d' % bo], env=env)
    a, b = torch.load('/tmp/flash_ot_ot1.pt'), torch.load('/tmp/flash_ot_ot2.pt')
    for k in a:
        x, y = a[k].double(), b[k].double()
        print(k, 'identical' if torch.equal(a[k], b[k]) else 'rel diff %.3e  max abs %.3e' % (((x - y).norm() / x.norm()).item(), (x - y).abs().max().item()))
