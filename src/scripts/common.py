"""Shared driver of the four launcher scripts.  Flag names, defaults and printed strings follow the
reference CLIs (main_lessr.py:7-53, main_niser.py:7-53, main_msgifsr.py:35-112); main_srgnn.py is the
script start.sh:6 expects but the reference never shipped.  GPU selection is ROCm
aware (HIP_VISIBLE_DEVICES / first device) instead of shelling out to nvidia-smi (quirk 4)."""
import argparse
import os
import random
import sys
from pathlib import Path

HERE = Path(__file__).resolve()
sys.path.insert(0, str(HERE.parents[2]))
sys.path.insert(0, str(HERE.parents[1]))

DEFAULTS = {
    'LESSR': dict(embedding_dim=32, num_layers=3, feat_drop=0.2, batch_size=512, patience=2, num_workers=0),
    'NISER': dict(embedding_dim=64, num_layers=2, feat_drop=0.5, batch_size=128, patience=2, num_workers=4),
    'SRGNN': dict(embedding_dim=64, num_layers=2, feat_drop=0.5, batch_size=128, patience=2, num_workers=4),
    'MSGIFSR': dict(embedding_dim=256, num_layers=1, feat_drop=0.1, batch_size=512, patience=3, num_workers=4),
}


def parse(model):
    d = DEFAULTS[model]
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--dataset-dir', default='../datasets/sample', help='the dataset directory')
    p.add_argument('--embedding-dim', type=int, default=d['embedding_dim'], help='the embedding size')
    p.add_argument('--num-layers', type=int, default=d['num_layers'], help='the number of layers')
    p.add_argument('--feat-drop', type=float, default=d['feat_drop'], help='the dropout ratio for features')
    p.add_argument('--lr', type=float, default=1e-3, help='the learning rate')
    p.add_argument('--batch-size', type=int, default=d['batch_size'], help='the batch size for training')
    p.add_argument('--epochs', type=int, default=30, help='the number of training epochs')
    p.add_argument('--weight-decay', type=float, default=1e-4, help='the parameter for L2 regularization')
    p.add_argument('--patience', type=int, default=d['patience'],
                   help='the number of epochs that the performance does not improves after which the training stops')
    p.add_argument('--num-workers', type=int, default=d['num_workers'],
                   help='the number of processes to load the input graphs')
    p.add_argument('--valid-split', type=float, default=None, help='the fraction for the validation set')
    p.add_argument('--log-interval', type=int, default=100, help='print the loss after this number of iterations')
    p.add_argument('--precision', default=os.environ.get('SREC_PRECISION', 'fp32'), choices=['fp32', 'bf16'],
                   help='(not in the reference) fp32: results match the reference to fp32 round-off; bf16: bf16 MFMA operands '
                        'with fp32 accumulation and master weights (BASELINE config C3), ~2x faster, metrics within 0.03 pt')
    p.add_argument('--no-graph', action='store_true',
                   help='(not in the reference) eager launches instead of replaying one captured hipGraph per training step')
    p.add_argument('--checkpoint', default=None,
                   help='(not in the reference) write a resumable checkpoint here after every epoch; resume from it if present')
    p.add_argument('--loader', default='ring', choices=['ring', 'torch'],
                   help='(not in the reference) ring: worker processes collate straight into a shared pinned ring '
                        '(sessionrec-pytorch_amd/loader.py); torch: torch.utils.data.DataLoader as in the reference')
    p.add_argument('--metrics-log', default=None, help='(not in the reference) append one JSON line per logged interval / epoch')
    p.add_argument('--gpus', type=int, default=int(os.environ.get('SREC_GPUS', '1')),
                   help='(not in the reference) train on this many GPUs of the node: item table row-sharded over them (RCCL), '
                        'encoder replicated, every rank encoding its slice of each --batch-size batch (same loss as one GPU)')
    if model == 'MSGIFSR':
        p.add_argument('--order', type=int, default=3, help='order of msg')
        p.add_argument('--reducer', type=str, default='mean', help='method for reducer')
        p.add_argument('--norm', type=bool, default=True, help='whether use l2 norm')
        p.add_argument('--extra', action='store_true', help='whether use REnorm.')
        p.add_argument('--fusion', action='store_true', help='whether use IFR.')
    args = p.parse_args()
    if int(os.environ.get('RANK', '0')) == 0:
        print(args)
    return args


def seed_all(seed=123):
    import numpy as np
    import torch
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def _jsonl(path):
    import json

    def hook(event):
        with open(path, 'a') as f:
            f.write(json.dumps(event) + '\n')
    return hook


def _launch_ranks(n):
    """`--gpus N` outside a torchrun environment: re-execute this launcher under torch.distributed.run, one rank per GPU"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    env.pop('HIP_VISIBLE_DEVICES', None)          # start.sh pins one device for the single-GPU case
    cmd = [sys.executable, '-u', '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def run(model_name):
    args = parse(model_name)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        _launch_ranks(args.gpus)
    seed_all(123)
    import torch as th
    from torch.utils.data import DataLoader, SequentialSampler
    from src.models import LESSR, MSGIFSR, NISER, SRGNN
    from src.utils.data.collate import (collate_fn_factory, collate_fn_factory_ccs, seq_to_ccs_graph,
                                        seq_to_eop_multigraph, seq_to_session_graph, seq_to_shortcut_graph)
    from src.utils.data.dataset import AugmentedDataset, read_dataset
    from src.utils.train import TrainRunner

    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    sharded = world > 1 or bool(os.environ.get('SREC_FORCE_COLLECTIVES'))
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29544')
        dist.init_process_group('nccl' if th.cuda.is_available() else 'gloo', rank=rank, world_size=world)
        if th.cuda.is_available():
            th.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        if rank != 0:
            sys.stdout = open(os.devnull, 'w')            # one log, rank 0's
    device = th.device('cuda', th.cuda.current_device()) if th.cuda.is_available() else th.device('cpu')
    if device.type == 'cuda':
        from importlib import import_module
        import_module('sessionrec-pytorch_amd.ops').set_precision(args.precision)
        if args.precision == 'fp32':
            print('precision fp32 (reference numerics); --precision bf16 runs the same step about twice as fast')
    print('reading dataset')
    train_sessions, test_sessions, num_items = read_dataset(Path(args.dataset_dir))
    if args.valid_split is not None:
        num_valid = int(len(train_sessions) * args.valid_split)
        test_sessions = train_sessions[-num_valid:]
        train_sessions = train_sessions[:-num_valid]
    train_set, test_set = AugmentedDataset(train_sessions), AugmentedDataset(test_sessions)
    caps = None
    shuffled = model_name in ('NISER', 'SRGNN')
    per_rank = (args.batch_size + world - 1) // world        # sessions one rank encodes per step
    if sharded:
        # every rank must present the SAME padded layout sizes to the collectives in every step: capacities are mandatory
        # and must never overflow - exact maxima of the epoch for the sequential loaders, worst case for the shuffled ones
        from src.utils.data.collate import default_caps, estimate_caps
        max_len = int(train_set.index[:, 1].max()) if len(train_set) else 1
        caps = (default_caps(per_rank, max_len) if shuffled else
                estimate_caps(train_set, per_rank, headroom=1.0, slices=(args.batch_size, world)))
        if model_name == 'LESSR':
            caps = dict(caps, E=caps['N'] * max(7, max_len))
    elif device.type == 'cuda' and not args.no_graph and not getattr(args, 'extra', False):
        from src.utils.data.collate import estimate_caps, measure_caps
        # capacity-padded training batches -> whole-step hipGraph replay.  MSGIFSR / SRGNN / NISER: capacities MEASURED on the
        # epoch's largest batches (tight: kernels whose grid follows the capacity pay for slack); LESSR: the click-count bound
        if model_name == 'LESSR':
            caps = estimate_caps(train_set, args.batch_size, shuffled=shuffled)
            caps = dict(caps, E=caps['N'] * 7)               # shortcut graphs: up to L(L+1)/2 edges per session
        else:
            caps = measure_caps(train_set, args.batch_size, 'ccs' if model_name == 'MSGIFSR' else 'session',
                                getattr(args, 'order', 1), shuffled=shuffled)
    print(len(train_set))
    print(len(test_set))
    if model_name == 'LESSR':
        fns = (seq_to_eop_multigraph, seq_to_shortcut_graph) if args.num_layers > 1 else (seq_to_eop_multigraph,)
        collate_fn = collate_fn_factory(*fns)
        train_collate_fn = collate_fn_factory(*fns, caps=caps)
        model = LESSR(num_items, args.embedding_dim, args.num_layers, feat_drop=args.feat_drop)
    elif model_name == 'MSGIFSR':
        collate_fn = collate_fn_factory_ccs((seq_to_ccs_graph,), order=args.order)
        train_collate_fn = collate_fn_factory_ccs((seq_to_ccs_graph,), order=args.order, caps=caps)
        model = MSGIFSR(num_items, args.dataset_dir, args.embedding_dim, args.num_layers, dropout=args.feat_drop,
                        reducer=args.reducer, order=args.order, norm=args.norm, extra=args.extra, fusion=args.fusion,
                        device=device)
    else:
        collate_fn = collate_fn_factory(seq_to_session_graph)
        train_collate_fn = collate_fn_factory(seq_to_session_graph, caps=caps)
        cls = NISER if model_name == 'NISER' else SRGNN
        model = cls(num_items, args.embedding_dim, args.num_layers, feat_drop=args.feat_drop)
    pin = device.type == 'cuda'          # pinned batches: asynchronous H2D copies

    def ring(batch_sampler):
        # single-graph capacity-padded training batches: built by the workers INSIDE a shared pinned ring (no pickling, no
        # pinning pass in this process); anything else keeps the DataLoader
        if args.loader != 'ring' or device.type != 'cuda' or (model_name == 'LESSR' and args.num_layers > 1):
            return None
        from importlib import import_module
        kind = {'MSGIFSR': 'ccs', 'LESSR': 'eop'}.get(model_name, 'session')
        return import_module('sessionrec-pytorch_amd.loader').ring_loader_or_none(
            train_set, batch_sampler, kind, getattr(args, 'order', 1), caps, args.num_workers)
    pw = args.num_workers > 0            # keep the loader processes across epochs (a respawn costs seconds per epoch)
    # reference loaders: LESSR / MSGIFSR train in time order (SequentialSampler), NISER shuffles; test shuffles
    if sharded:
        # the reference's batches (order and membership), each rank collating its contiguous slice of every batch
        from importlib import import_module
        from torch.utils.data import RandomSampler
        RankSlice = import_module('sessionrec-pytorch_amd.dataset').RankSliceBatchSampler
        base = (RandomSampler(train_set, generator=th.Generator().manual_seed(123)) if shuffled
                else SequentialSampler(train_set))
        slices = RankSlice(base, args.batch_size, rank, world, prefix_len=train_set.index[:, 1],
                           need_len=(args.order + 1) if model_name == 'MSGIFSR' else None)
        train_loader = ring(slices) or DataLoader(train_set, batch_sampler=slices, num_workers=args.num_workers,
                                                  collate_fn=train_collate_fn, pin_memory=pin, persistent_workers=pw)
        # evaluation: every rank scores the same sessions against its rows (order does not enter the metrics)
        test_loader = DataLoader(test_set, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                                 collate_fn=collate_fn, persistent_workers=pw)
    elif model_name in ('LESSR', 'MSGIFSR'):
        from torch.utils.data import BatchSampler
        train_loader = (ring(BatchSampler(SequentialSampler(train_set), args.batch_size, drop_last=False)) or
                        DataLoader(train_set, batch_size=args.batch_size, num_workers=args.num_workers, collate_fn=train_collate_fn,
                                   sampler=SequentialSampler(train_set), pin_memory=pin, persistent_workers=pw))
        test_loader = None
    else:
        from torch.utils.data import BatchSampler, RandomSampler
        # (the shuffle has its own seeded generator: the batch order of a run does not depend on what else draws from
        #  torch's global stream)
        shuffle = RandomSampler(train_set, generator=th.Generator().manual_seed(123))
        train_loader = (ring(BatchSampler(shuffle, args.batch_size, drop_last=False)) or
                        DataLoader(train_set, batch_size=args.batch_size, sampler=shuffle, num_workers=args.num_workers,
                                   collate_fn=train_collate_fn, pin_memory=pin, persistent_workers=pw))
        test_loader = None
    if test_loader is None:
        test_loader = DataLoader(test_set, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers,
                                 collate_fn=collate_fn, persistent_workers=pw)
    model = model.to(device)
    print(model)
    shard = None
    if sharded:
        from importlib import import_module
        shard = import_module('sessionrec-pytorch_amd.dist').VocabParallel(model, idx_cap=caps['U'])
        print(f'item table row-sharded over {world} ranks ({shard.per} rows each), {per_rank} sessions per rank and step')
    runner = TrainRunner(args.dataset_dir, model, train_loader, test_loader, device=device, lr=args.lr,
                         weight_decay=args.weight_decay, patience=args.patience, checkpoint=args.checkpoint,
                         hooks=[_jsonl(args.metrics_log)] if args.metrics_log and rank == 0 else (),
                         graph=False if args.no_graph else 'auto', shard=shard)
    print('start training')
    mrr, hit = runner.train(args.epochs, args.log_interval)
    if runner.graph_steps:
        print(f'training steps: {runner.graph_steps} hipGraph replays, {runner.eager_steps} eager')
    if sharded and getattr(slices, 'short_batches', 0):
        print(f'note: {slices.short_batches} global batches had no session of {slices.need_len} clicks - on those the row-sharded '
              'step keeps the residual / bias of relations without edges (single-device skips them: dataset.RankSliceBatchSampler)')
    print('MRR@20\tHR@20')
    print(f'{mrr * 100:.3f}%\t{hit * 100:.3f}%')
    if sharded:
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
