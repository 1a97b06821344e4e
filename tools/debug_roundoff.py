"""Development probe: step-0 gradient error of the fused fp32 training path against the CPU oracle in float64, next to the
error of the reference's own fp32 gradients (fixture).  usage: python tools/debug_roundoff.py <fixture name>"""
import sys, torch, importlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_models_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else 'msgifsr_K1_edge'
dev = torch.device('cuda:0')
train, optim = T.pkg('train'), T.pkg('optim')
z, samples, init = T.load_golden(name)
V = init[[k for k in init if k.startswith('embedding')][0]].shape[0]
model = T._build(name, init, V, dev)
inputs, labels = T._collate(name, samples)
inputs = [x.to(dev) for x in inputs]; labels = labels.to(dev)
model.train()
model.zero_grad(set_to_none=True)
loss = model.fused_loss(*inputs, labels); loss.backward()
mine = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
mine['embeddings.weight'] = model.table_grad.buf.detach().double().cpu()
# fp64 oracle gradients
from oracle import collate_ref as oc, models_ref as om
K = int(name.split('_')[1][1:])
m = om.MSGIFSR(V, 'sample', 32, 1, order=K, extra='_ext' in name, fusion='_fus' in name)
m.load_state_dict(init); m = m.double()
torch.set_default_dtype(torch.float64)
fn = oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), K)
inp, lab = fn(samples)
inp = [om.to_torch(x) for x in inp]
m.train()
torch.nn.functional.nll_loss(m(*inp), torch.from_numpy(lab)).backward()
torch.set_default_dtype(torch.float32)
for k, p in m.named_parameters():
    if p.grad is None or k not in mine:
        continue
    t = p.grad.double()
    ref = torch.as_tensor(z['grad/' + k]).double() if 'grad/' + k in z.files else None
    em = (mine[k][:t.shape[0]] - t).abs()
    er = (ref - t).abs() if ref is not None and ref.shape == t.shape else None
    print('%-45s |g| %.1e  mine mean %.2e max %.2e   ref32 mean %s max %s' % (k, t.abs().mean(), em.mean(), em.max(), '%.2e' % er.mean() if er is not None else '-', '%.2e' % er.max() if er is not None else '-'))

# ---- parameter error against the float64 trajectory after every Adam step
print('--- trajectory')
model = T._build(name, init, V, dev)
model.train()
opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model)
m = om.MSGIFSR(V, 'sample', 32, 1, order=K, extra='_ext' in name, fusion='_fus' in name)
m.load_state_dict(init); m = m.double()
torch.set_default_dtype(torch.float64)
o64 = torch.optim.Adam(train.fix_weight_decay(m), lr=1e-3, weight_decay=1e-4)
m.train()
for step in range(3):
    opt.zero_grad(); model.fused_loss(*inputs, labels).backward(); opt.step()
    o64.zero_grad(); torch.nn.functional.nll_loss(m(*inp), torch.from_numpy(lab)).backward(); o64.step()
    sd, s64 = model.state_dict(), m.state_dict()
    out = []
    for k in ('embeddings.weight', 'fc_sr.0.weight', 'readout.fc_u.0.weight', 'layers.0.conv1.mods.intra1.fc.weight'):
        e = (sd[k].double().cpu() - s64[k]).abs()
        out.append('%s mean %.2e max %.2e' % (k.split('.')[0], e.mean(), e.max()))
    print('step', step, ' | '.join(out))
torch.set_default_dtype(torch.float32)
