"""debug: one rank of a 2-rank job replayed under capture, stage by stage (see tests/test_dist_gpu.py)"""
import os, sys, tempfile, faulthandler
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
faulthandler.enable()
import torch
from util import pkg
import test_dist_gpu as T
from dist_gpu_worker import make_case, rank_slice

if __name__ == '__main__':
    name, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device('cuda:0')
    D, G, train, optim = pkg('dist'), pkg('graph'), pkg('train'), pkg('optim')
    case = dict(kind='fixture', name=name, steps=2, padded=True, record=True)
    tmp = tempfile.mkdtemp()
    res = T._launch(world, case, tmp)
    job = res[rank]
    print('job done', [k for k, _, _ in job['steps'][1]['tape']], flush=True)
    build, collate, samples, V = make_case(case)
    mine, n = rank_slice(samples, world, rank, False)
    caps = pkg('collate').default_caps(n, 20)
    inputs, labels = collate(caps)(mine)
    inputs, labels = [x.to(dev) for x in inputs], labels.to(dev)
    print('counts', {k: v for k, v in inputs[0].meta['counts'].items() if k.startswith('E_')}, flush=True)
    model = build().to(dev)
    group = D.ReplayGroup(world, rank, dev).load(job['steps'][0]['tape'])
    vp = D.VocabParallel(model, group=group, idx_cap=inputs[0].cap('uniq_items'))
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model, fuse_projection=True)
    replicated = [p for p in model.parameters() if p is not model._table() and p.requires_grad]
    model.train()

    def one():
        opt.zero_grad()
        loss = model.fused_loss(*inputs, labels)
        loss.backward()
        vp.sync_replicated_grads(replicated, opt)
        opt.step()
        return loss
    keep = one()
    print('eager step 1', keep.item(), job['steps'][0]['loss'], flush=True)
    if 'hold' not in (sys.argv[4] if len(sys.argv) > 4 else ''):
        del keep
    group.load(job['steps'][1]['tape'])
    variant = sys.argv[4] if len(sys.argv) > 4 else 'manual'
    if variant != 'gstep':
        snap_o = opt.snapshot()
        snap_p = [p.detach().clone() for p in model.parameters()]
        print('eager step 2', one().item(), job['steps'][1]['loss'], flush=True)
        opt.restore(snap_o)
        with torch.no_grad():
            for p, q in zip(model.parameters(), snap_p):
                p.copy_(q)
        ms = model.__dict__.get('_srec_state')
        if ms is not None:
            ms['cs_fresh'] = False
        torch.cuda.synchronize()
    variant = sys.argv[4] if len(sys.argv) > 4 else 'manual'
    print('variant', variant, flush=True)
    if variant == 'gstep':
        gs = G.GraphedTrainStep(model, opt, inputs, labels, after_backward=lambda: vp.sync_replicated_grads(replicated, opt), warmup=1)
        print('GraphedTrainStep built', flush=True)
        print('replayed loss', gs(inputs, labels).item(), job['steps'][1]['loss'], flush=True)
        sys.exit(0)
    g = torch.cuda.CUDAGraph(keep_graph=True) if 'keep' in variant else torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    one_t = torch.ones((), device=dev)

    stage = int(variant[-1]) if (variant[-1].isdigit() and not variant.startswith(('b', 'g'))) else (1 if variant.startswith(('b', 'g')) else 4)
    if 'gc' in variant:
        import gc
        gc.collect()
        torch.cuda.synchronize()

    stash = {}
    if variant.startswith('b'):
        _sr, _lk = model.session_repr, model._lookup

        def sr_wrap(*a, **k):
            out = _sr(*a, **k)
            stash['sr'] = out
            return out

        def lk_wrap(*a, **k):
            out = _lk(*a, **k)
            stash['rows'] = out
            return out
        model.session_repr, model._lookup = sr_wrap, lk_wrap

    def body():
        print('capture: forward', flush=True)
        loss = model.fused_loss(*inputs, labels)
        if variant.startswith('b'):
            tgt = stash['sr'] if variant == 'b1' else stash['rows']
            print('capture: partial backward to', variant, tuple(tgt.shape), flush=True)
            torch.autograd.grad(loss, [tgt], one_t)
            print('capture: body done', flush=True)
            return loss
        if variant.startswith('g2'):
            sub = variant.split(':')[1] if ':' in variant else ''
            ps = [p for k, p in model.named_parameters() if p.requires_grad and p is not model._table() and sub in k]
            print('capture: autograd.grad wrt %d params' % len(ps), flush=True)
            torch.autograd.grad(loss, ps, one_t, allow_unused=True)
            print('capture: body done', flush=True)
            return loss
        if stage >= 2:
            print('capture: backward', flush=True)
            loss.backward(one_t)
        if stage >= 3:
            print('capture: sync grads', flush=True)
            vp.sync_replicated_grads(replicated, opt)
        if stage >= 4:
            print('capture: optimizer', flush=True)
            work = opt._work()
            opt.advance(work)
            opt.launch(work)
        print('capture: body done', flush=True)
        return loss
    if 'ctx' in variant:
        with torch.cuda.graph(g):
            loss = body()
    else:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g.capture_begin()
            try:
                loss = body()
            except BaseException as e:
                import traceback
                traceback.print_exc()
                print('EXCEPTION inside capture:', type(e).__name__, e, flush=True)
            print('capture_end ...', flush=True)
            g.capture_end()
    print('captured', flush=True)
    if stage < 4:
        sys.exit(0)
    g.replay()
    torch.cuda.synchronize()
    print('replayed loss', loss.item(), job['steps'][1]['loss'], flush=True)
