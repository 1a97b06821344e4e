"""bench.py - training throughput (sessions/sec) of the HIP hot path.

    python bench.py --gpus N --steps K --warmup W [--model MSGIFSR|SRGNN|NISER|LESSR]

A "step" = one pass of the hot path over one batch: embedding gather -> session encoder ->
fused full-catalog scoring + softmax-CE forward/backward -> fused Adam (dense over the item
table).  Inputs (the collated flat batches) are resident in HBM before the timed region.
Synthetic data of the Yoochoose-1/64 shape (no datasets in the image): V = 37 484 items,
batch 512, session length <= 20, ids Zipf(1.0), 20 % immediate revisits, seed 123;
random-init weights of the named architecture.

`--gpus N` with no WORLD_SIZE in the environment re-executes this script under `python -m torch.distributed.run` with N
ranks (one per GPU, rendezvous on 127.0.0.1); under the driver's own torchrun launch the ranks are simply used.  The
item table is then row-sharded over the ranks (dist.VocabParallel, RCCL).  Default = weak scaling (batch 512 PER GPU,
global batch N * 512); `--global-batch 512` = strong scaling: the SAME 512 consecutive samples per step as one GPU,
each rank encoding its contiguous slice of them (the reference's 512-sample loss semantics, train.py:94-101).

Prints ONE JSON line (rank 0) with the driver's contract plus
  "roofline":     dominant kernel (fused scoring/CE backward launch) timed live with HIP events around a hipGraph of
                  back-to-back launches on the launch stream
  "cpu_baseline": the CPU oracle (pure-PyTorch restatement of the reference math) timed on the
                  host cores of the same box on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V_YOOCHOOSE = 37484


def _latest_profile(stem):
    """newest committed PMC summary profiles/rNN_<stem>.json (the kernel's HBM traffic per launch)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s.json' % stem)))
    return os.path.basename(c[-1]) if c else 'r01_%s.json' % stem


PMC_BF16 = _latest_profile('pmc_flash_ce_bf16')
PMC_FP32 = _latest_profile('pmc_flash_ce')


def synth_sessions(n_sessions, V, mean_len, max_len, rng):
    """clipped-geometric lengths (min 2), Zipf(1.0) item ids, 20 % immediate-revisit probability."""
    p = 1.0 / (mean_len - 1.0)
    lens = np.clip(rng.geometric(p, size=n_sessions) + 1, 2, max_len)
    w = 1.0 / np.arange(1, V + 1)
    cdf = np.cumsum(w / w.sum())
    out = []
    for L in lens:
        ids = np.searchsorted(cdf, rng.random(L)).clip(0, V - 1)
        rev = rng.random(L) < 0.2
        for i in range(1, L):
            if rev[i]:
                ids[i] = ids[i - 1]
        out.append(ids.tolist())
    return out


def make_batches(model_name, order, n_batches, B, V, max_len, seed, padded=False, part=None):
    """n_batches batches of B consecutive prefix samples.  part=(r, w): collate only the contiguous slice r of w of every
    batch (strong scaling: the ranks split ONE global batch); the returned samples stay the whole batches."""
    ds = importlib.import_module('sessionrec-pytorch_amd.dataset')
    col = importlib.import_module('sessionrec-pytorch_amd.collate')
    rng = np.random.default_rng(seed)
    sessions = synth_sessions(int(n_batches * B / 4.5) + 64, V, 6.2, max_len, rng)
    arr = np.empty(len(sessions), dtype=object)
    arr[:] = sessions
    data = ds.AugmentedDataset(arr)
    assert len(data) >= n_batches * B, (len(data), n_batches * B)
    samples = [[data[b * B + i] for i in range(B)] for b in range(n_batches)]
    mine = samples
    if part is not None:
        r, w = part
        Bl = B // w
        mine = [s[r * Bl:(r + 1) * Bl] for s in samples]
        B = Bl
    def factory(caps):
        if model_name in ('SRGNN', 'NISER'):
            return col.collate_fn_factory(col.seq_to_session_graph, caps=caps)
        if model_name == 'LESSR':
            return col.collate_fn_factory(col.seq_to_eop_multigraph, caps=caps)
        return col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), order, caps=caps)
    caps = None
    if padded:          # capacities = the maxima over the epoch's batches (a loader knows them after one pass)
        mxN = mxE = mxU = 1
        for smp in mine:
            (fb,), _ = factory(None)(smp)
            cnt = fb.meta['counts']
            mxN = max([mxN] + [v for k, v in cnt.items() if k.startswith('N') and k != 'NT'])
            mxE = max([mxE] + [v for k, v in cnt.items() if k.startswith('E')])
            mxU = max(mxU, cnt.get('U', 1))
        r256 = lambda v: (v + 255) // 256 * 256
        caps = dict(B=B, N=r256(mxN), E=r256(mxE), U=r256(mxU))
        print('bench caps', caps, file=sys.stderr)
    fn = factory(caps)
    return [fn(s) for s in mine], samples


def build_model(sp, name, V, d, order, dropout=0.0):
    if name == 'SRGNN':
        return sp.SRGNN(V, d, 1, feat_drop=0.0)
    if name == 'NISER':
        return sp.NISER(V, d, 1, feat_drop=0.0)
    if name == 'LESSR':
        return sp.LESSR(V, d, 1, feat_drop=0.0)
    return sp.MSGIFSR(V, 'synthetic', d, 1, dropout=dropout, order=order, extra=False, fusion=False)


def time_dominant_kernel(model, B, V, d, dev, iters=20, force_fp32=False):
    """HIP-event timing (on the launch stream) of the fused scoring/CE kernels alone:
    forward, backward dE pass (4*B*V*d flop per launch), backward d-sr pass."""
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    L = importlib.import_module('sessionrec-pytorch_amd._lib')
    lib, ptr, stream = L.lib, L.ptr, L.stream
    table = model._table().detach()[:V]
    sr = torch.randn(B, d, device=dev) * 0.1
    labels = torch.randint(0, V, (B,), device=dev, dtype=torch.int32)
    ws = ops.CEWorkspace(B, V, d, dev)
    lse = torch.empty(B, device=dev)
    lossvec = torch.empty(B, device=dev)
    loss = torch.empty((), device=dev)
    dE = torch.empty_like(table)
    dsr = torch.empty(B, d, device=dev)
    bf16 = ops.use_bf16_scoring(d) and not force_fp32
    tb = ops.TableBF16(table).refresh(table) if bf16 else None

    def fwd():
        ops._ce_fwd(sr, table, None, labels, ws, None, tb, ws.lab_logit, lse, lossvec, loss)

    def bwd(parts):
        ops._ce_bwd(sr, table, None, labels, lse, None, None, None, ws, None, tb, dE, dsr, parts)

    def bwd_kernel():         # the backward launch alone: session copies already prepared, slabs left unreduced
        lib.srec_score_ce_bwd_bf16(ptr(ws.sr16), None, ws.Bp, ptr(tb.E16), None, tb.Vp, None,
                                   ptr(labels), ptr(lse), None, None, None, B, V, d, None, ptr(dE), dE.stride(0),
                                   ptr(ws.dsr_part), ptr(dsr), 3 | 8, stream())
    fwd()
    out = {}

    def timed(fn):
        """`iters` back-to-back launches captured in one hipGraph (no host launch gaps), HIP events around the replay"""
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    for name, fn in (('dE', lambda: bwd(1)), ('dsr', lambda: bwd(2)), ('bwd', lambda: bwd(3)), ('fwd', fwd)):
        out[name] = timed(fn)
    if bf16:
        out['table_bf16_prepare'] = timed(lambda: tb.refresh(table))
        out['bwd_kernel'] = timed(bwd_kernel)
    out['bf16'] = bf16
    return out


def cpu_baseline(model_name, samples, V, d, order, state_dict, budget_s=12.0, dropout=0.0):
    """the CPU oracle (restatement of the reference math: materialised logits, log(softmax), nll_loss,
    autograd backward, torch Adam with L2) on this box's host cores, bounded sample."""
    from oracle import collate_ref as oc
    from oracle import models_ref as om
    train = importlib.import_module('sessionrec-pytorch_amd.train')
    avail, phys = physical_cores()
    if model_name == 'SRGNN':
        m, fn = om.SRGNN(V, d, 1), oc.collate_fn_factory(oc.seq_to_session_graph)
    elif model_name == 'NISER':
        m, fn = om.NISER(V, d, 1), oc.collate_fn_factory(oc.seq_to_session_graph)
    elif model_name == 'LESSR':
        m, fn = om.LESSR(V, d, 1), oc.collate_fn_factory(oc.seq_to_eop_multigraph)
    else:
        m = om.MSGIFSR(V, 'synthetic', d, 1, dropout=dropout, order=order, extra=False, fusion=False)
        fn = oc.collate_fn_factory_ccs((oc.seq_to_ccs_graph,), order)
    m.load_state_dict(state_dict)
    opt = torch.optim.Adam(train.fix_weight_decay(m), lr=1e-3, weight_decay=1e-4)
    batches = [fn(s) for s in samples[:4]]
    batches = [([om.to_torch(x) for x in inp], torch.from_numpy(lab)) for inp, lab in batches]
    m.train()

    def step(b):
        inp, lab = b
        opt.zero_grad()
        loss = torch.nn.functional.nll_loss(m(*inp), lab)
        loss.backward()
        opt.step()
    # pick the torch thread count that is actually fastest on this host (os.cpu_count() threads on a
    # many-core box oversubscribe the small per-session ops by orders of magnitude)
    best, cores = None, 1
    for thr in sorted({t for t in (8, 16, 32) if t <= avail} or {avail}):
        torch.set_num_threads(thr)
        step(batches[0])
        t0 = time.time()
        step(batches[1 % len(batches)])
        dt1 = time.time() - t0
        if best is None or dt1 < best:
            best, cores = dt1, thr
        if dt1 > 3.0 * best:
            break
    torch.set_num_threads(cores)
    step(batches[0])                       # warm-up at the chosen thread count
    n, t0 = 0, time.time()
    while True:
        step(batches[n % len(batches)])
        n += 1
        if time.time() - t0 > budget_s or n >= 40:
            break
    dt = time.time() - t0
    B = len(samples[0])
    return dict(value=n * B / dt, unit='sessions/s', cores=cores, host_logical_cpus=avail, host_physical_cores=phys,
                kind='port',
                sample='%d training steps (batch %d) of the CPU oracle (%s, V=%d, d=%d), collate excluded, %d torch threads'
                       % (n, B, model_name, V, d, cores))


def FlatBatchCopy(x):
    return type(x)(x.buf.clone().pin_memory(), x.layout, dict(x.meta))


def end_to_end(args, sp, state, V, d, B, dev, n_batches=600, warm=16, workers=4):
    """The metric as the reference's loop defines it (train.py:92-110): wall time of `for batch in train_loader:` -
    DataLoader worker processes building the session graphs (native collate, csrc/collate.cpp), pinned batches, the
    asynchronous H2D copy, and the hipGraph replay of the whole step inside TrainRunner.train_step - on the same synthetic
    split, the launcher's own capacities (collate.estimate_caps) and loader settings (main_msgifsr.py:148-166: sequential,
    4 workers, pin_memory).  The first `warm` batches (worker start-up, graph capture) are outside the timed region."""
    from torch.utils.data import DataLoader, SequentialSampler
    ds = importlib.import_module('sessionrec-pytorch_amd.dataset')
    col = importlib.import_module('sessionrec-pytorch_amd.collate')
    train = importlib.import_module('sessionrec-pytorch_amd.train')
    rng = np.random.default_rng(321)
    sessions = synth_sessions(int(n_batches * B / 4.5) + 64, V, 6.2, 20, rng)
    arr = np.empty(len(sessions), dtype=object)
    arr[:] = sessions
    data = ds.AugmentedDataset(arr)
    n_batches = min(n_batches, len(data) // B)
    data.index = data.index[:n_batches * B]
    caps = (col.estimate_caps(data, B) if args.model == 'LESSR' else                     # the launcher's own capacities
            col.measure_caps(data, B, 'ccs' if args.model == 'MSGIFSR' else 'session', args.order))   # (src/scripts/common.py)
    if args.model in ('SRGNN', 'NISER'):
        fn = col.collate_fn_factory(col.seq_to_session_graph, caps=caps)
    elif args.model == 'LESSR':
        fn = col.collate_fn_factory(col.seq_to_eop_multigraph, caps=caps)
    else:
        fn = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), args.order, caps=caps)
    loader, which = None, 'torch DataLoader (pin_memory)'
    if getattr(args, 'e2e_loader', 'ring') == 'ring' and args.model != 'LESSR':
        # the launchers' default (src/scripts/common.py --loader ring): workers collate straight into a shared pinned ring
        from torch.utils.data import BatchSampler
        ring = importlib.import_module('sessionrec-pytorch_amd.loader')
        loader = ring.ring_loader_or_none(data, BatchSampler(SequentialSampler(data), B, drop_last=False),
                                          'ccs' if args.model == 'MSGIFSR' else 'session', args.order, caps, workers)
        which = 'pinned ring (loader.PinnedRingLoader)'
    if loader is None:
        loader = DataLoader(data, batch_size=B, sampler=SequentialSampler(data), num_workers=workers, collate_fn=fn,
                            pin_memory=True, persistent_workers=workers > 0, prefetch_factor=4 if workers > 0 else None)
        which = 'torch DataLoader (pin_memory)'
    torch.manual_seed(123)
    model = build_model(sp, args.model, V, d, args.order, args.dropout)
    model.load_state_dict(state)
    model = model.to(dev).train()
    runner = train.TrainRunner('synthetic', model, loader, None, device=dev, lr=1e-3, weight_decay=1e-4)
    probe = None
    if getattr(args, 'e2e_probe', False):
        # where the loop's time goes: (a) the loader alone (worker collate + pinning, nothing consumed on the GPU), (b) the
        # training step fed from pre-collated pinned batches (no loader: host-side step overhead + H2D + replay)
        it0 = iter(loader)
        for _ in range(warm):
            next(it0)
        t0, nb = time.perf_counter(), 0
        held = []
        for batch in it0:
            nb += 1
            if len(held) < 64:                          # (copies: a ring loader's batches are views of slots it reuses)
                held.append(([FlatBatchCopy(x) for x in batch[0]], batch[1]))
        t_loader = (time.perf_counter() - t0) / max(nb, 1)
        del it0
        probe = dict(loader_only_ms_per_batch=t_loader * 1e3)
    it = iter(loader)
    loss = None
    for _ in range(warm):
        inputs, labels = next(it)
        loss = runner.train_step(inputs, labels)
    torch.cuda.synchronize()
    g0, e0 = runner.graph_steps, runner.eager_steps
    w0 = dict(getattr(loader, 'stats', {}))
    n, t0 = 0, time.perf_counter()
    pending = []
    for batch in it:                                      # TrainRunner.train's loop body (train.py:94-104), its loss
        inputs, labels = batch                            # bookkeeping included: values read back every 256 steps
        loss = runner.train_step(inputs, labels)
        pending.append(runner._loss_handle(loss))
        if len(pending) >= 256:
            ring = runner._gstep.loss_ring.tolist() if any(isinstance(v, int) for v in pending) else None
            if ring is not None:
                runner._gstep.check()
            vals = [ring[v % len(ring)] if isinstance(v, int) else float(v) for v in pending]
            assert all(v == v for v in vals), 'loss is NaN'
            pending.clear()
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    final = float(loss.item())
    if getattr(runner, '_gstep', None) is not None:
        runner._gstep.check()                             # no replay of the loop ran on a stale batch
    # (the loader counts cumulative seconds under keys ending in _s: reported here as milliseconds per step, keys renamed)
    waits = {(k[:-2] if k.endswith('_s') else k): round((v - w0[k]) / max(n, 1) * 1e3, 4) for k, v in getattr(loader, 'stats', {}).items()}
    if probe is not None:
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(256):
                inputs, labels = held[i % len(held)]
                loss = runner.train_step(inputs, labels)
            t_host = time.perf_counter() - t0                 # host has issued 256 steps
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
        probe.update(precollated_ms_per_step=t_all / 256 * 1e3, precollated_host_issue_ms_per_step=t_host / 256 * 1e3)
        gs = runner._gstep
        if gs is not None:
            # what separates a step fed from the host from the replay of device-resident batches at the launcher's capacities:
            # (a) batches already on the device (mailbox entry -> the batch itself), (b) page-locked host batches (side-stream
            # copy + host-side wait + replay = the precollated figure above)
            dev_held = [([x.to(dev) for x in b[0]], b[1].to(dev)) for b in held[:8]]

            def timed(fn, n=512):
                for _ in range(2):
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for i in range(n):
                        fn(i)
                    t_issue = (time.perf_counter() - t) / n * 1e3
                    torch.cuda.synchronize()
                    dt_ = (time.perf_counter() - t) / n * 1e3
                return dt_, t_issue
            on_dev = timed(lambda i: gs(*dev_held[i % len(dev_held)]))
            probe['gap_ms_per_step'] = dict(replay_device_batches=on_dev[0], host_issue_device_batches=on_dev[1])
    del it
    if hasattr(loader, 'close'):
        loader.close()
    del loader
    return dict(value=n * B / dt, unit='sessions/s', ms_per_step=dt / n * 1e3, steps=n, workers=workers, collate='native'
                if col._native() is not None else 'python', loader=which, loader_waits_ms_per_step=waits or None, pinned=True, caps=caps,
                graph_steps=runner.graph_steps - g0, eager_steps=runner.eager_steps - e0, final_loss=final, probe=probe,
                note='wall time of the DataLoader loop (train.py:92-110 equivalent): worker collate + pinned H2D copy + '
                     'hipGraph replay per batch, evaluation excluded')


def quality(sp, dev, precision):
    """Recall@20 / MRR@20 after training (the second half of BASELINE.json's metric), on the one real split the image holds:
    the reference's shipped datasets/sample.  The pinned recipe of tests/golden/trained_metrics.json (MSGIFSR order 2, d 64,
    1 layer, dropout 0, batch 512 in time order, 3 epochs through TrainRunner: train.py:56-127) trained on the HIP path in
    this run's arithmetic mode, next to the committed numbers of the CPU ORACLE trained by the reference's own loop on the
    same batches from the same seeded weights (tests/golden/make_trained_metrics.py)."""
    ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    ds = importlib.import_module('sessionrec-pytorch_amd.dataset')
    col = importlib.import_module('sessionrec-pytorch_amd.collate')
    train = importlib.import_module('sessionrec-pytorch_amd.train')
    pin = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'trained_metrics.json')))
    cfg = pin['msgifsr_o2_d64']
    tr, te, V = ds.read_dataset(os.path.join(ROOT, 'datasets', 'sample'))
    train_set, test_set = ds.AugmentedDataset(tr), ds.AugmentedDataset(te)
    cf = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), cfg['order'])
    B = cfg['batch_size']
    mk = lambda data: [cf([data[i] for i in range(b, min(len(data), b + B))]) for b in range(0, len(data), B)]
    trl, tel = mk(train_set), mk(test_set)
    ops.set_precision(precision)
    torch.manual_seed(cfg['seed'])
    model = sp.MSGIFSR(V, 'sample', cfg['embedding_dim'], cfg['num_layers'], dropout=0.0, order=cfg['order'], extra=False,
                       fusion=False).to(dev)
    runner = train.TrainRunner('sample', model, trl, tel, dev, lr=1e-3, weight_decay=1e-4, patience=99)
    t0 = time.perf_counter()
    _out, sys.stdout = sys.stdout, open(os.devnull, 'w')          # (the loop prints the reference's per-epoch lines)
    try:
        runner.train(len(cfg['epochs']) - 1, log_interval=10 ** 9)
    finally:
        sys.stdout = _out
    mrr, hit = train.evaluate(model, tel, dev)
    omrr, ohit = cfg['epochs'][-1]
    return dict(split='datasets/sample (the reference\'s shipped split: %d train / %d test samples, %d items)' % (len(train_set), len(test_set), V),
                model='MSGIFSR order %d, d %d, %d epochs, batch %d, dropout 0 (the pinned recipe of tests/golden/trained_metrics.json)'
                      % (cfg['order'], cfg['embedding_dim'], len(cfg['epochs']) - 1, B),
                precision=precision, recall_at_20=100 * hit, mrr_at_20=100 * mrr, oracle_recall_at_20=100 * ohit,
                oracle_mrr_at_20=100 * omrr, unit='percent', delta_recall_pt=100 * (hit - ohit), delta_mrr_pt=100 * (mrr - omrr),
                oracle='CPU oracle (oracle/models_ref.py) trained by the reference loop, fp32', train_seconds=time.perf_counter() - t0)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a torchrun environment: re-execute under torch.distributed.run with N ranks
    on this node (one process per GPU, rendezvous on 127.0.0.1).  Rank 0's JSON line passes straight through stdout."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL / xGMI peer mappings need it on this driver
    env.setdefault('NCCL_DEBUG_FILE', '/dev/stderr')        # RCCL's version banner (NCCL_DEBUG=VERSION) must not land on stdout
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def physical_cores():
    """(logical cpus usable by this process, physical cores of the box from /proc/cpuinfo)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    phys = set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pid = line.split(':')[1].strip()
            elif line.startswith('core id'):
                cid = line.split(':')[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return avail, (len(phys) or None)


def encoder_flop(model_name, counts, d, order, H=8):
    """algorithmic flop of the session encoder for one batch, forward + backward (3 x the forward GEMM flop: forward,
    backward-data, weight gradient), from the batch's live node counts.  MSGIFSR: k-gram GRU projections
    (msgifsr.py:25,42), the GAT fc of every module of both HeteroGraphConvs (gatconv.py:282-283: intra_k projects the
    order-k rows, the shared 'inter' module every row), the readout's fc_u / fc_v and fc_sr (msgifsr.py:139-147,272)."""
    if model_name != 'MSGIFSR':
        N = counts.get('N', 0)
        B = counts.get('B', 0)
        return 3 * (2.0 * N * d * d + 2.0 * B * d * d + 2.0 * B * 2 * d * d)
    K = order
    Nk = [counts.get('N%d' % k, 0) for k in range(1, K + 1)]
    NT, B = sum(Nk), counts.get('B', 0)
    f = 0.0
    for k in range(2, K + 1):
        n = Nk[k - 1]
        f += 2.0 * n * k * d * 3 * d + 2.0 * n * (k - 1) * d * 3 * d            # GI (all steps) + GH (steps 2..k)
    for conv in range(2):
        f += sum(2.0 * n * d * H * d for n in Nk)                               # intra_k modules
        if K > 1:
            f += 2.0 * NT * d * H * d                                           # shared 'inter' module
    f += 2.0 * NT * d * d + 2.0 * B * d * d + 2.0 * B * 2 * d * d               # live readout head: fc_u, fc_v, fc_sr
    return 3 * f


def _all_reduce(dist, t, op=None):
    """all-reduce of a small device tensor through the job's backend.  RCCL takes device memory; the gloo dry run (N rank
    processes on ONE GPU: SREC_BENCH_BACKEND=gloo, tests/test_dist_gpu.py) stages through the host"""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend() != 'nccl':
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def run_timed(step, dev_batches, warmup, steps, repeats, dist, dev):
    """W untimed steps, then `repeats` regions of exactly K steps, each bracketed by barrier + synchronize on both sides,
    max over ranks.  -> (seconds of each region, last loss)"""
    nb = len(dev_batches)
    it = 0
    for _ in range(warmup):
        step(dev_batches[it % nb])
        it += 1
    regions, loss = [], None
    for _ in range(repeats):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(dev_batches[it % nb])
            it += 1
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dt = _all_reduce(dist, t, dist.ReduceOp.MAX).item()
        regions.append(dt)
    return regions, loss


_JSON_OUT = None


def emit(obj):
    """the one JSON line of this process, on the ORIGINAL stdout"""
    f = _JSON_OUT or sys.stdout
    f.write(json.dumps(obj) + '\n')
    f.flush()


class Env:
    """what every job of this process shares"""
    pass


def make_job(env, args, precision, batches, dev_batches, strong, group=None, graph=True, log_prefix=''):
    """model + optimizer + (captured) training step for one arithmetic mode and one batch geometry, from the same initial
    weights.  Multi-rank jobs (or --shard / a replayed rank: `group`) row-shard the item table (dist.VocabParallel)."""
    sp, ops, train, optim, dist, dev = env.sp, env.ops, env.train, env.optim, env.dist, env.dev
    ops.set_precision(precision)
    torch.manual_seed(123)
    model = build_model(sp, args.model, env.V, env.d, args.order, args.dropout)
    model.load_state_dict(env.state)
    model = model.to(dev)
    shard = None
    if env.world > 1 or args.shard or group is not None:       # item table row-sharded over the node's GPUs (RCCL / xGMI)
        D = importlib.import_module('sessionrec-pytorch_amd.dist')
        cap = batches[0][0][0].cap('uniq_items')       # only the distinct items of a batch are exchanged; equal padded
        if group is None:                              # request length on every rank: no per-step size exchange
            t = torch.tensor([cap], device=dev)
            _all_reduce(dist, t, dist.ReduceOp.MAX)
            cap = int(t.item())
        elif getattr(env, 'replay_cap', None):
            cap = env.replay_cap
        shard = D.VocabParallel(model, group=group, idx_cap=cap)
    opt = optim.FusedAdam(train.fix_weight_decay(model), lr=1e-3, weight_decay=1e-4, model=model, fuse_projection=True)
    replicated = [p for p in model.parameters() if p is not model._table() and p.requires_grad]
    model.train()

    def eager(b):
        inp, lab = b
        opt.zero_grad()
        loss = model.fused_loss(*inp, lab)
        loss.backward()
        if shard is not None:
            shard.sync_replicated_grads(replicated, opt)
        opt.step()
        return loss
    job = dict(model=model, shard=shard, opt=opt, eager=eager, step=eager, graphed=False, gstep=None, attempts=0,
               replicated=replicated, strong=strong, precision=precision, captured_on_all_ranks=None)
    return job


def capture_job(env, args, job, dev_batches, warmup=2):
    """capture the job's step (graph.capture_agreed: the ranks of an N-GPU job retry and fall back TOGETHER)"""
    dist, dev = env.dist, env.dev
    G = importlib.import_module('sessionrec-pytorch_amd.graph')
    model, opt, shard, replicated = job['model'], job['opt'], job['shard'], job['replicated']
    multi = shard is not None and env.world > 1 and dist is not None
    agree = (lambda x: _all_reduce(dist, torch.tensor([x], device=dev), dist.ReduceOp.MIN).item()) if multi else None
    after = (lambda: shard.sync_replicated_grads(replicated, opt)) if shard is not None else None
    # host-staged collectives (the gloo dry run) can never be captured: one attempt; RCCL: one agreed retry
    retries = 1 if (shard is not None and dist is not None and dist.is_initialized() and dist.get_backend() == 'nccl') else 0
    log = lambda m: print(m, file=sys.stderr, flush=True)
    gstep, attempts, err = G.capture_agreed(
        lambda: G.GraphedTrainStep(model, opt, dev_batches[0][0], dev_batches[0][1], after_backward=after, warmup=warmup),
        agree, retries=retries, log=log)
    job['attempts'] = attempts
    if gstep is None:
        if shard is None:
            raise err
        print('graph capture with the collectives failed (%s); running eager'
              % (('%s: %s' % (type(err).__name__, str(err)[:300])) if err is not None else 'on another rank'),
              file=sys.stderr, flush=True)
    else:
        job.update(step=(lambda b: gstep(b[0], b[1])), graphed=True, gstep=gstep)
    job['captured_on_all_ranks'] = bool(gstep is not None)     # (agreed: the same on every rank)
    return job


def loss_check(env, args, job, dev_batch, first_samples, strong):
    """the correctness bit of an N-rank line: the loss of the first global batch through the row-sharded forward (all ranks,
    evaluation mode - dropout off - so the two paths see the same function) against the single-device forward rank 0 runs on the
    same sessions from the same weights (train.py:97-99: model(*inputs) -> nll_loss)."""
    dist, dev = env.dist, env.dev
    model = job['model']
    model.eval()
    with torch.no_grad():
        l_sh = float(model.fused_loss(*dev_batch[0], dev_batch[1]).item())
    model.train()
    # every rank's first batch travels to rank 0 as the sample lists (weak scaling: different sessions per rank)
    if strong or dist is None or env.world == 1:
        parts = [first_samples]
    else:
        parts = [None] * env.world
        dist.all_gather_object(parts, first_samples)
    out = None
    if env.rank == 0:
        env.ops.set_precision(job['precision'])
        torch.manual_seed(123)
        ref = build_model(env.sp, args.model, env.V, env.d, args.order, args.dropout)
        ref.load_state_dict(env.state)
        ref = ref.to(dev).eval()
        col = importlib.import_module('sessionrec-pytorch_amd.collate')
        if args.model in ('SRGNN', 'NISER'):
            fn = col.collate_fn_factory(col.seq_to_session_graph)
        elif args.model == 'LESSR':
            fn = col.collate_fn_factory(col.seq_to_eop_multigraph)
        else:
            fn = col.collate_fn_factory_ccs((col.seq_to_ccs_graph,), args.order)
        tot, n = 0.0, 0
        with torch.no_grad():
            for smp in parts:
                inp, lab = fn(smp)
                tot += float(ref.fused_loss(*[x.to(dev) for x in inp], lab.to(dev)).item()) * len(smp)
                n += len(smp)
        l_1 = tot / max(n, 1)
        tol = 5e-3 if job['precision'] == 'bf16' else 1e-4
        rel = abs(l_sh - l_1) / max(abs(l_1), 1e-12)
        out = dict(sharded_loss=l_sh, single_device_loss=l_1, rel_err=rel, tol=tol, ok=bool(rel <= tol), sessions=n,
                   what='loss of the first global batch, evaluation mode: row-sharded forward over all ranks vs the '
                        'single-device forward on rank 0 (same weights, same sessions)')
        del ref
    return out


def time_collectives(env, job, dev_batches, steps=3):
    """per-exchange time on this rank's timeline: HIP events around every collective of `steps` EAGER steps (dist.TIMING),
    the first one dropped; all ranks run the steps (the exchanges pair up), rank 0 reports"""
    D = importlib.import_module('sessionrec-pytorch_amd.dist')
    per_step = []
    for i in range(steps + 1):
        D.TIMING = []
        job['eager'](dev_batches[i % len(dev_batches)])
        torch.cuda.synchronize()
        if i > 0:
            per_step.append(D.timing_summary(D.TIMING))
    D.TIMING = None
    n = min(len(x) for x in per_step)
    out = []
    for j in range(n):
        e = per_step[0][j]
        out.append(dict(kind=e['kind'], bytes=e['bytes'], us=round(float(np.mean([st[j]['us'] for st in per_step])), 2)))
    return dict(per_exchange=out, sum_us=round(sum(e['us'] for e in out), 1), steps=len(per_step),
                note='eager steps, HIP events on the issuing stream around each exchange (waiting for the slowest rank included); '
                     'in the captured step the bucket all-reduces overlap the backward on a side stream')


def run_job(env, args, precision, strong, Bg, B, n_batches, warm, steps, repeats, with_timing=True):
    """one multi- or single-rank configuration end to end: batches, job, correctness bit, capture, timed regions"""
    dist, dev, rank, world = env.dist, env.dev, env.rank, env.world
    if strong:       # every rank collates ITS slice of the same global batches (same seed everywhere)
        batches, samples = make_batches(args.model, args.order, n_batches, Bg, env.V, 20, 123, padded=True, part=(rank, world))
        first = samples[0]
    else:
        batches, samples = make_batches(args.model, args.order, n_batches, B, env.V, 20, 123 + rank, padded=True)
        first = samples[0]
    dev_batches = [([x.to(dev) for x in inp], lab.to(dev)) for inp, lab in batches]
    job = make_job(env, args, precision, batches, dev_batches, strong)
    shard = job['shard']
    check = None
    if shard is not None and world > 1:
        check = loss_check(env, args, job, dev_batches[0], first, strong)
    if not args.no_graph:
        capture_job(env, args, job, dev_batches)
    coll = None
    D = importlib.import_module('sessionrec-pytorch_amd.dist') if shard is not None else None
    if shard is not None:
        if job['graphed'] and getattr(job['gstep'], 'collectives', None):
            coll = dict(job['gstep'].collectives, captured_in_graph=True)
        else:
            D.STATS['count'] = D.STATS['bytes'] = 0
            job['step'](dev_batches[0])
            coll = dict(D.STATS, captured_in_graph=False)
        coll['per'] = 'training step and rank'
        coll['buckets'] = list(D.BUCKETS['bytes'])          # replicated-gradient all-reduces, in backward completion order
        coll['backend'] = dist.get_backend() if (dist is not None and dist.is_initialized()) else 'replayed tape'
        coll['capture_attempts'] = job['attempts']
        coll['captured_on_all_ranks'] = job['captured_on_all_ranks']
        coll['capture_mode'] = getattr(job['gstep'], 'capture_mode', None)
        coll['bucket_copy_tasks'] = dict(getattr(shard, 'copy_tasks_last', {}))
    regions, loss = run_timed(job['step'], dev_batches, warm, steps, max(repeats, 1), dist if world > 1 or args.shard else None, dev)
    final_loss = loss.item()
    if coll is not None and with_timing and (world > 1 or args.shard) and not args.step_only:
        coll['timed'] = time_collectives(env, job, dev_batches)   # (eager steps next to a captured graph: also on the 1-rank rehearsal)
    dt = sorted(regions)[len(regions) // 2]              # median region
    ms = [r / steps * 1e3 for r in regions]
    nodes = job['gstep'].node_counts() if job['graphed'] and hasattr(job['gstep'], 'node_counts') else None
    return dict(job=job, batches=batches, samples=samples, dev_batches=dev_batches, coll=coll, check=check, dt=dt, ms=ms,
                nodes=nodes, final_loss=final_loss, Bg=Bg, B=B, strong=strong, steps=steps)


def record_tape(args):
    """rank side of `--replay-rank R --of W`: W rank processes on ONE GPU over gloo (device tensors staged through the host),
    every rank running the real per-rank kernels on its row shard; rank R records what the collectives of the second step
    (steady state: the bucket layout is agreed) - and of the first - handed it, and saves the tapes."""
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    env = _env(args, dist, torch.device('cuda', 0), rank, world)
    D = importlib.import_module('sessionrec-pytorch_amd.dist')
    strong = args.global_batch is not None
    Bg = args.global_batch if strong else args.batch * world
    B = Bg // world
    n_batches = min(args.steps + args.warmup, 48)
    if strong:
        batches, samples = make_batches(args.model, args.order, n_batches, Bg, env.V, 20, 123, padded=True, part=(rank, world))
    else:
        batches, samples = make_batches(args.model, args.order, n_batches, B, env.V, 20, 123 + rank, padded=True)
    dev_b = ([x.to(env.dev) for x in batches[0][0]], batches[0][1].to(env.dev))
    cap = torch.tensor([batches[0][0][0].cap('uniq_items')])
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    env.replay_cap = int(cap.item())
    group = D.RecordingGroup()
    job = make_job(env, args, args.precision, batches, [dev_b], strong, group=group)
    tapes, losses = [], []
    for _ in range(2):
        group.tape = []
        loss = job['eager'](dev_b)
        tapes.append(list(group.tape))
        losses.append(float(loss.item()))
    if rank == args.replay_rank:
        torch.save(dict(tapes=tapes, losses=losses, cap=env.replay_cap, lo=job['shard'].lo, hi=job['shard'].hi), args.record_tape)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


def replay_rank(args):
    """`bench.py --replay-rank R --of W [--global-batch 512]` on ONE GPU: the step rank R of a W-rank job runs - V/W table rows;
    weak scaling: W x 512 sessions scored, 512 encoded; strong: 512 scored, 512/W encoded - captured and timed with the
    collectives replaced by their recorded results (dist.ReplayGroup: a device-to-device copy where the job has its RCCL call).
    Transport-free prediction of the N-GPU point: measured N-GPU step - this = what RCCL / xGMI cost."""
    import subprocess
    import tempfile
    R, W = args.replay_rank, args.of
    assert 0 <= R < W
    tape_path = os.path.join(tempfile.mkdtemp(prefix='srec_tape_'), 'tape.pt')
    env_ = dict(os.environ, SREC_BENCH_BACKEND='gloo')
    env_.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env_.setdefault('OMP_NUM_THREADS', '4')
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(W), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv + ['--record-tape', tape_path]
    t0 = time.time()
    rc = subprocess.call(cmd, env=env_, stdout=sys.stderr)
    if rc != 0:
        sys.exit(rc)
    t_rec = time.time() - t0
    rec = torch.load(tape_path, weights_only=False)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    env = _env(args, None, dev, R, W)
    env.replay_cap = rec['cap']
    D = importlib.import_module('sessionrec-pytorch_amd.dist')
    strong = args.global_batch is not None
    Bg = args.global_batch if strong else args.batch * W
    B = Bg // W
    n_batches = min(args.steps + args.warmup, 48)
    if strong:
        batches, samples = make_batches(args.model, args.order, n_batches, Bg, env.V, 20, 123, padded=True, part=(R, W))
    else:
        batches, samples = make_batches(args.model, args.order, n_batches, B, env.V, 20, 123 + R, padded=True)
    dev_b = ([x.to(dev) for x in batches[0][0]], batches[0][1].to(dev))
    group = D.ReplayGroup(W, R, dev, rtol=2e-3 if args.precision == 'bf16' else 1e-4, atol=1e-5).load(rec['tapes'][0])
    job = make_job(env, args, args.precision, batches, [dev_b], strong, group=group)
    shard = job['shard']
    assert (shard.lo, shard.hi) == (rec['lo'], rec['hi'])
    loss0 = float(job['eager'](dev_b).item())                  # step 1, eager: every collective input is checked against the job's
    checked = group.checked
    group.load(rec['tapes'][1])
    G = importlib.import_module('sessionrec-pytorch_amd.graph')
    D.STATS['count'] = D.STATS['bytes'] = 0
    gstep = G.GraphedTrainStep(job['model'], job['opt'], dev_b[0], dev_b[1],
                               after_backward=lambda: shard.sync_replicated_grads(job['replicated'], job['opt']), warmup=1)
    regions, loss = run_timed(lambda b: gstep(b[0], b[1]), [dev_b], args.warmup, args.steps, max(args.repeats, 1), None, dev)
    dt = sorted(regions)[len(regions) // 2]
    ms = dt / args.steps * 1e3
    tape = rec['tapes'][1]
    exch = [dict(kind=k, bytes_in=int(i.numel() * i.element_size()), bytes_out=int(o.numel() * o.element_size())) for k, i, o in tape]
    out = dict(replay_rank=R, of=W, scaling='strong' if strong else 'weak', global_batch=Bg, sessions_encoded_per_rank=B,
               table_rows_per_rank=shard.n_live, ms_per_step_kernels=ms, repeats_ms=[r / args.steps * 1e3 for r in regions],
               predicted_value=Bg / (ms * 1e-3), unit='sessions/s (transport-free: collectives = recorded results copied on the device)',
               launches=gstep.node_counts(), collectives=dict(count=len(tape), per_exchange=exch,
                                                              bytes_total=sum(e['bytes_out'] for e in exch),
                                                              buckets=list(D.BUCKETS['bytes'])),
               eager_step_loss=loss0, job_step_loss=rec['losses'][0], collective_inputs_checked=checked,
               bucket_copy_tasks=dict(getattr(shard, 'copy_tasks_last', {})),
               precision=args.precision, model=args.model, V=env.V, d=env.d, order=args.order, dropout=args.dropout,
               batch='batch 0 of rank %d (one batch: the recorded exchange results belong to it)' % R,
               recording_seconds=round(t_rec, 1), steps=args.steps, warmup=args.warmup)
    emit(out)


def _env(args, dist, dev, rank, world):
    env = Env()
    env.sp = importlib.import_module('sessionrec-pytorch_amd')
    env.ops = importlib.import_module('sessionrec-pytorch_amd.ops')
    env.train = importlib.import_module('sessionrec-pytorch_amd.train')
    env.optim = importlib.import_module('sessionrec-pytorch_amd.optim')
    env.dist, env.dev, env.rank, env.world = dist, dev, rank, world
    env.V, env.d = args.items, args.dim
    env.ops.set_precision(args.precision)
    torch.manual_seed(123)
    env.state = {k: v.clone() for k, v in build_model(env.sp, args.model, env.V, env.d, args.order, args.dropout).state_dict().items()}
    return env


def main():
    # stdout carries exactly one JSON line.  RCCL prints its version banner with plain printf at communicator teardown
    # (NCCL_DEBUG=VERSION in this image), i.e. AFTER the line: from here on file descriptor 1 of this process is stderr, and the
    # JSON line goes to a private duplicate of the original stdout.
    global _JSON_OUT
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--repeats', type=int, default=3, help='timed regions of --steps steps each; value = the median region')
    ap.add_argument('--model', default='MSGIFSR')
    ap.add_argument('--order', type=int, default=3)
    ap.add_argument('--dim', type=int, default=256)
    ap.add_argument('--items', type=int, default=V_YOOCHOOSE)
    ap.add_argument('--batch', type=int, default=512, help='sessions per GPU and step (weak scaling)')
    ap.add_argument('--global-batch', type=int, default=None,
                    help='strong scaling ONLY: this many sessions per step over ALL ranks (512 = the reference batch, '
                         'train.py:94-101); rank r encodes the contiguous slice r of every global batch.  Without it an N > 1 run '
                         'times BOTH: weak scaling (the headline value) and strong scaling at 512 (the `strong` object)')
    ap.add_argument('--no-strong', action='store_true', help='N > 1: skip the strong-scaling (global batch 512) side run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32', action='store_true', help='skip the fp32 side run (the reference arithmetic) of the same step')
    ap.add_argument('--no-end-to-end', action='store_true', help='skip the DataLoader-inclusive run of the same workload')
    ap.add_argument('--no-quality', action='store_true', help='skip the trained Recall@20 / MRR@20 run on datasets/sample')
    ap.add_argument('--e2e-loader', default='ring', choices=['ring', 'torch'],
                    help='loader of the end-to-end run: the shared pinned ring (launcher default) or torch DataLoader')
    ap.add_argument('--e2e-workers', type=int, default=4, help='DataLoader worker processes of the end-to-end run')
    ap.add_argument('--e2e-probe', action='store_true', help='end-to-end run: also time the loader alone and the step fed from pre-collated batches')
    ap.add_argument('--dropout', type=float, default=0.1,
                    help='MSGIFSR feature / attention dropout (0.1 = --feat-drop default of the reference launcher, main_msgifsr.py:42)')
    ap.add_argument('--precision', default=os.environ.get('SREC_PRECISION', 'bf16'), choices=['fp32', 'bf16'],
                    help="MFMA operand type of the forward / backward-data GEMMs ('bf16' = BASELINE config C3)")
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of the captured whole-step hipGraph')
    ap.add_argument('--shard', action='store_true', help='debug: row-sharded path + RCCL calls on a 1-rank communicator (set SREC_FORCE_COLLECTIVES=1)')
    ap.add_argument('--kernel-only', action='store_true', help='only launch the scoring/CE kernels (PMC collection target)')
    ap.add_argument('--step-only', action='store_true',
                    help='profiling target: only the timed training steps (no dominant-kernel timing loop, no CPU baseline, no '
                         'fp32 side run), so every kernel row of a rocprofv3 --stats summary belongs to the step')
    ap.add_argument('--launch-only', action='store_true',
                    help='initialise the process group over --gpus ranks, print what was seen, exit (launcher self-test; '
                         'gloo on a box without GPUs)')
    ap.add_argument('--replay-rank', type=int, default=None,
                    help='with --of W, on ONE GPU: time the step rank R of a W-rank job runs, collectives replaced by their '
                         'recorded results (transport-free prediction of the W-GPU point; weak scaling, or strong with --global-batch)')
    ap.add_argument('--of', type=int, default=None, help='world size of the job whose rank --replay-rank is replayed')
    ap.add_argument('--record-tape', default=None, help=argparse.SUPPRESS)      # (rank side of --replay-rank)
    args = ap.parse_args()
    if args.step_only:
        args.no_cpu_baseline = args.no_fp32 = args.no_end_to_end = args.no_quality = True
        args.repeats = 1

    if args.record_tape is not None:                      # a rank process of the recording job (stdout = the parent's stderr)
        return record_tape(args)
    if args.replay_rank is not None:
        assert args.of and args.of >= 1, '--replay-rank R needs --of W'
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
        return replay_rank(args)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    sys.stdout.flush()                                    # (a rank process from here on: the launcher parent keeps its stdout)
    _JSON_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    has_gpu = torch.cuda.is_available()
    dist = None
    if world > 1 or args.shard or args.launch_only:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # SREC_BENCH_BACKEND=gloo: the N-rank dry run on a box with ONE GPU - every rank on cuda:0, collectives staged through
        # the host (sessionrec-pytorch_amd/dist.py) - so that the whole N > 1 line (ranks_seen, collectives, scaling, the
        # eager fallback when the collectives cannot be captured) has been produced before an 8-GPU node ever runs it
        backend = os.environ.get('SREC_BENCH_BACKEND') or ('nccl' if has_gpu else 'gloo')
        dist.init_process_group(backend, rank=rank, world_size=world)
        if has_gpu and backend != 'nccl':
            local = local % torch.cuda.device_count()
    if args.launch_only:
        dev = torch.device('cuda', local) if has_gpu else torch.device('cpu')
        if has_gpu:
            torch.cuda.set_device(local)
        seen = _all_reduce(dist, torch.ones(1, device=dev))   # every rank contributes 1: the sum is the number of live ranks
        if rank == 0:
            emit(dict(launch_only=True, n_gpus=world, gpus_requested=args.gpus, ranks_seen=int(seen.item()),
                                  world_size=dist.get_world_size(), backend=dist.get_backend()))
        dist.barrier()
        dist.destroy_process_group()
        return
    if world != args.gpus and rank == 0 and not args.shard:
        print('note: --gpus %d but WORLD_SIZE=%d: running %d ranks' % (args.gpus, world, world), file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    env = _env(args, dist, dev, rank, world)
    sp, ops, V, d, state = env.sp, env.ops, env.V, env.d, env.state
    only_strong = args.global_batch is not None
    if only_strong:
        assert args.global_batch % world == 0, 'global batch must divide over the ranks'
        Bg, B = args.global_batch, args.global_batch // world
    else:
        B, Bg = args.batch, args.batch * world
    if args.kernel_only:
        torch.manual_seed(123)
        model = build_model(sp, 'SRGNN', V, d, 1).to(dev)
        kt = time_dominant_kernel(model, B, V, d, dev, iters=5)
        emit(kt)
        return
    n_batches = min(args.steps + args.warmup, 48)      # distinct resident batches; longer runs cycle through them

    main_run = run_job(env, args, args.precision, only_strong, Bg, B, n_batches, args.warmup, args.steps, args.repeats)
    job = main_run['job']
    model, shard, graphed, gstep = job['model'], job['shard'], job['graphed'], job['gstep']
    batches, samples, dev_batches = main_run['batches'], main_run['samples'], main_run['dev_batches']
    coll, dt, ms, nodes, final_loss = main_run['coll'], main_run['dt'], main_run['ms'], main_run['nodes'], main_run['final_loss']
    strong = only_strong
    # every rank contributes 1 through the job's own backend (RCCL): the sum is the number of ranks the collectives reach
    ranks_seen = 1
    if dist is not None:
        ranks_seen = int(_all_reduce(dist, torch.ones(1, device=dev)).item())

    strong_run = None
    if world > 1 and not only_strong and not args.no_strong and not args.step_only and args.batch % world == 0:
        # the reference's own semantics (main_msgifsr.py:46,148-157: ONE batch of 512 consecutive samples per optimiser step):
        # the same 512 sessions as the 1-GPU run, every rank encoding its slice - timed in the same process, same launch mode
        sr = run_job(env, args, args.precision, True, args.batch, args.batch // world, n_batches, args.warmup, args.steps,
                     max(1, min(args.repeats, 2)))
        strong_run = dict(scaling='strong', global_batch=args.batch, sessions_encoded_per_rank=args.batch // world,
                          value=args.batch * args.steps / sr['dt'], unit='sessions/s', ms_per_step=sr['dt'] / args.steps * 1e3,
                          repeats_ms_per_step=sr['ms'], launch='hipGraph replay' if sr['job']['graphed'] else 'eager',
                          launches=sr['nodes'], collectives=sr['coll'], correctness=sr['check'], final_loss=sr['final_loss'],
                          note='the reference batch (train.py:94-101): the SAME 512 consecutive sessions per step as on one GPU')
        del sr
        ops.set_precision(args.precision)

    fp32 = None
    if not args.no_fp32 and args.precision != 'fp32' and world == 1 and not args.shard:      # (side run on one GPU only)
        # the reference's own arithmetic (fp32 operands everywhere) on the same batches, same launch mode, same run
        j32 = make_job(env, args, 'fp32', batches, dev_batches, strong)
        if not args.no_graph:
            capture_job(env, args, j32, dev_batches)
        r32, _ = run_timed(j32['step'], dev_batches, args.warmup, args.steps, 1, None, dev)
        fp32 = dict(ms_per_step=r32[0] / args.steps * 1e3, value=Bg * args.steps / r32[0], unit='sessions/s',
                    launch='hipGraph replay' if j32['graphed'] else 'eager')
        del j32
        ops.set_precision(args.precision)

    e2e = None
    if not args.no_end_to_end and world == 1 and not args.shard:
        ops.set_precision(args.precision)
        e2e = end_to_end(args, sp, state, V, d, B, dev, workers=args.e2e_workers)
        e2e['vs_value'] = e2e['value'] / (Bg * args.steps / dt)

    qual = None
    if not args.no_quality and world == 1 and not args.shard:
        try:
            qual = quality(sp, dev, args.precision)
        except FileNotFoundError as e:                   # (a checkout without datasets/sample or the pin file)
            qual = dict(error=str(e))
        ops.set_precision(args.precision)

    if rank == 0 and args.step_only:
        emit(dict(step_only=True, ms_per_step=dt / args.steps * 1e3, value=Bg * args.steps / dt, launches=nodes,
                  final_loss=final_loss, collectives=coll))
    elif rank == 0:
        Vk = V if shard is None else shard.n_live      # rows of the catalog this rank scores
        kt = time_dominant_kernel(model, Bg, Vk, d, dev)
        kms = {k: v * 1e3 for k, v in kt.items() if k != 'bf16'}
        if kt['bf16']:
            # one launch computes dE (all item tiles) and the d-sr slabs: 2*B*V*d algorithmic flop each
            # (the S = sr E^T recomputation inside both halves is not counted)
            name, flop, t, peak = ('flash_ce_bf16_kernel<KIND_BWD> (fused scoring/CE backward: dE item tiles + d-sr '
                                   'item ranges in one launch)', 4.0 * Bg * Vk * d, kt['bwd_kernel'], 2500.0)
            alg_bytes = Vk * d * 2.0 + Vk * d * 4.0      # read the bf16 table once, write dE fp32 once
            pmc, pkey = PMC_BF16, 'KIND_BWD'
        else:
            name, flop, t, peak = ('flash_ce_kernel<MODE_DE> (fused scoring/CE backward, dE pass)', 4.0 * Bg * Vk * d,
                                   kt['dE'], 157.3)
            alg_bytes = 2.0 * Vk * d * 4
            pmc, pkey = PMC_FP32, 'MODE_DE'
        traffic, tsrc = None, None
        try:                                  # HBM bytes per launch from the committed PMC passes (same kernel & shape)
            pm = json.load(open(os.path.join(ROOT, 'profiles', pmc)))
            w = pm['workload']
            if (w['V'], w['d'], w['B']) == (Vk, d, Bg):
                traffic = pm['kernels'][pkey]['traffic_bytes']
                tsrc = ('profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 2x FETCH '
                        'correction; read from the committed file, not measured in this run)' % pmc)
        except Exception:
            pass
        # ---- the whole step against both roofs: 40 V d table bytes + ~6 N d 4 node bytes; 6 B V d
        #      scoring flop (+ the encoder's GEMM flop, reported separately)
        cnt = [b[0][0].meta['counts'] for b in batches]
        mean_counts = {k: float(np.mean([c.get(k, 0) for c in cnt])) for k in cnt[0]}
        n_rows = mean_counts.get('NT', mean_counts.get('N', 0.0))
        t_step = dt / args.steps
        step_bytes = 40.0 * Vk * d + 6.0 * n_rows * d * 4
        score_flop = 6.0 * Bg * Vk * d
        enc_flop = encoder_flop(args.model, mean_counts, d, args.order)
        mfma_peak = 2500.0 if args.precision == 'bf16' else 157.3
        step_roof = dict(launches=nodes, kernel_time_ms=t_step * 1e3,
                         kernel_time_note='hipGraph replay: kernels run back to back, step wall time = GPU busy time; per-kernel '
                                          'split in profiles/ (rocprofv3 --kernel-trace --stats of this command)',
                         algorithmic_bytes=step_bytes, algorithmic_flop=score_flop, encoder_flop=enc_flop,
                         hbm_floor_ms=step_bytes / 8e12 * 1e3, mfma_floor_ms=(score_flop + enc_flop) / (mfma_peak * 1e12) * 1e3,
                         frac_hbm=step_bytes / t_step / 8e12, frac_mfma=(score_flop + enc_flop) / t_step / (mfma_peak * 1e12),
                         frac_mfma_scoring_only=score_flop / t_step / (mfma_peak * 1e12))
        roof = dict(bound='mfma', kernel=name, achieved=flop / t / 1e12, peak=peak, unit='TFLOP/s',
                    frac=flop / t / 1e12 / peak, traffic=traffic, traffic_unit='bytes/launch', traffic_source=tsrc,
                    algorithmic_bytes=alg_bytes, algorithmic_flop=flop, kernel_ms=kms, step=step_roof)
        cpu = None
        if not args.no_cpu_baseline and world == 1:        # the CPU baseline belongs to the N = 1 line (rank 0)
            cpu = cpu_baseline(args.model, samples, V, d, args.order, state, dropout=args.dropout)   # whole 512-session batches
        scaling = 'strong' if strong else 'weak'
        out = dict(metric='sessions/sec training, Yoochoose-1/64 batch 512', value=Bg * args.steps / dt,
                   unit='sessions/s', n_gpus=world, ranks_seen=ranks_seen, collectives=coll, correctness=main_run['check'],
                   strong=strong_run,
                   steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt / args.steps * 1e3, repeats_ms_per_step=ms, spread_ms=max(ms) - min(ms),
                   higher_is_better=True, scaling=scaling, vs_baseline=None,
                   dtype='f32' if args.precision == 'fp32' else 'bf16 (MFMA operands; fp32 accumulate, master weights, scoring)', data='synthetic', launch='hipGraph replay' if graphed else 'eager',
                   config=dict(workload='%s training step, synthetic Yoochoose-1/64 shape (V=%d items, d=%d, batch %d per GPU, '
                                        'session length<=20%s)' % (args.model, V, d, B,
                                                                   ', order %d, dropout %g' % (args.order, args.dropout) if args.model == 'MSGIFSR' else ''),
                               global_batch=Bg, parallelism=('item table row-sharded x%d (vocab-parallel scoring, RCCL all-gather/reduce-scatter), '
                                            'encoder replicated, %s' % (world, 'each rank encodes its slice of one 512-session batch' if strong else 'each rank feeds its own batch')) if world > 1 else 'single GPU',
                               final_loss=final_loss),
                   roofline=roof, fp32=fp32, end_to_end=e2e, quality=qual, cpu_baseline=cpu)
        emit(out)
    if dist is not None:
        dist.barrier()                      # the other ranks wait for rank 0's kernel timing / CPU baseline
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
