"""Training / evaluation harness - host-side mirror of /root/reference/src/utils/train.py:
fix_weight_decay :12-23, prepare_batch :26-32, evaluate :36-55 (returns (MRR@k, HR@k) in that
order), TrainRunner :56-127 (Adam + StepLR(3, 0.1), log strings, early stop when BOTH metrics drop).

On a GPU model the step goes through `model.fused_loss` (scoring GEMM + softmax-CE fused, logits
never materialised) and FusedAdam; the reference-style path `nll_loss(model(*inputs), labels)`
still works on the same modules (forward() returns (B, num_items) log-probabilities).
"""
import time

import torch as th
from torch import nn, optim


def fix_weight_decay(model):
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if any(tag in name for tag in ('bias', 'batch_norm', 'activation')):
            no_decay.append(param)
        else:
            decay.append(param)
    return [{'params': decay}, {'params': no_decay, 'weight_decay': 0}]


def prepare_batch(batch, device):
    inputs, labels = batch
    return [x.to(device) for x in inputs], labels.to(device, non_blocking=labels.is_pinned())


def evaluate(model, data_loader, device, cutoff=20):
    model.eval()
    mrr, hit, num_samples = 0.0, 0, 0
    with th.no_grad():
        for batch in data_loader:
            inputs, labels = prepare_batch(batch, device)
            if hasattr(model, 'topk') and labels.is_cuda:
                topk = model.topk(*inputs, k=cutoff)[1].long()    # fused: no (B, V) score matrix
                num_samples += topk.size(0)
            else:
                logits = model(*inputs)
                num_samples += logits.size(0)
                topk = logits.topk(k=cutoff)[1]
            hit_ranks = th.where(topk == labels.unsqueeze(-1))[1] + 1
            hit += hit_ranks.numel()
            mrr += hit_ranks.float().reciprocal().sum().item()
    return mrr / num_samples, hit / num_samples


class TrainRunner:
    def __init__(self, dataset, model, train_loader, test_loader, device, lr=1e-3, weight_decay=0, patience=3,
                 checkpoint=None, resume=True, hooks=(), graph='auto', shard=None):
        """Same positional surface as the reference (train.py:57-69).  Additions (the reference has
        neither): `checkpoint` = path written after every epoch (model, optimizer incl. Adam moments and step counts,
        scheduler, epoch / batch counters, best metrics) and, with `resume`, read back at the start of train();
        `hooks` = callables hook(event: dict) invoked after every logged interval and every epoch (the reference's
        wandb calls sit at the same two places).  `graph` ('auto' | False): on the GPU, batches that arrive
        capacity-padded (collate_fn_factory*(..., caps=...)) replay ONE captured hipGraph of the whole step (forward,
        backward, FusedAdam) instead of ~120 eager launches; a batch with another layout or relation pattern runs
        eagerly, with the same result.  `shard` (dist.VocabParallel, multi-GPU): the item table is row-sharded over the
        ranks and every rank feeds its slice of each batch (dataset.RankSliceBatchSampler); the replicated encoder
        gradients are all-reduced after backward, evaluation merges per-shard top-k lists, only rank 0 prints."""
        self.shard = shard
        self.rank = shard.rank if shard is not None else 0
        self.replicated = ([p for p in model.parameters() if p is not model._table() and p.requires_grad]
                           if shard is not None else None)
        self.graph = graph
        self._gstep = None
        self.graph_steps = self.eager_steps = 0
        self.checkpoint = checkpoint
        self.resume = resume
        self.hooks = list(hooks)
        self.best = (0, 0, 0)            # max_mrr, max_hit, bad_counter
        self.dataset = dataset
        self.model = model
        params = fix_weight_decay(model) if weight_decay > 0 else model.parameters()
        self.fused = th.device(device).type == 'cuda' and hasattr(model, 'fused_loss')
        if self.fused:
            from .optim import FusedAdam
            self.optimizer = FusedAdam(params, lr=lr, weight_decay=weight_decay, model=model, fuse_projection=True)
        else:
            self.optimizer = optim.Adam(params, lr=lr, weight_decay=weight_decay)
        self.scheduler = optim.lr_scheduler.StepLR(self.optimizer, step_size=3, gamma=0.1)
        self.train_loader = train_loader
        self.test_loader = test_loader
        self.device = device
        self.epoch = 0
        self.batch = 0
        self.patience = patience
        self.loss_trace = []

    def _graph_step(self, inputs, labels):
        """replay of the captured step, or None when this batch has to run eagerly"""
        if not (self.graph and self.fused and getattr(self.model, 'graph_capable', False)):
            return None
        if not all(getattr(x, 'meta', None) is not None and x.meta.get('padded') for x in inputs):
            return None
        from .graph import GraphedTrainStep, capture_agreed
        if self._gstep is None:
            dev_in = [x.to(self.device) for x in inputs]
            dev_lab = labels.to(self.device)
            agree = None
            if self.shard is not None and self.shard.world > 1:
                # the ranks decide TOGETHER (graph.capture_agreed): a rank whose capture failed retries with all the others
                # (the constructor's warm-up steps issue the step's collectives), and after the last attempt either all
                # replay or all launch eagerly
                import torch.distributed as dist
                from .dist import all_reduce_

                def agree(x):
                    ok = th.tensor([x], device=self.device)
                    all_reduce_(ok, dist.ReduceOp.MIN, self.shard.group)
                    return ok.item()
            # (a refused capture: the constructor has put parameters, buffers and optimizer state back - same trajectory)
            self._gstep, self.capture_attempts, err = capture_agreed(
                lambda: GraphedTrainStep(self.model, self.optimizer, dev_in, dev_lab,
                                         after_backward=self._sync_grads if self.shard is not None else None),
                agree, retries=1, log=print)
            if self._gstep is None:
                print('hipGraph capture failed (%s); eager launches' % (('%s: %s' % (type(err).__name__, err)) if err is not None
                                                                        else 'on another rank'))
                self.graph = False
            if self._gstep is None:
                return None
        try:
            return self._gstep(inputs, labels)
        except RuntimeError as e:
            if 'differs from the captured one' not in str(e):
                raise
            return None

    def _sync_grads(self):
        self.shard.sync_replicated_grads(self.replicated, self.optimizer)

    def _print(self, *a):
        if self.rank == 0:
            print(*a)

    def _loss_handle(self, loss):
        """what the training loop keeps of a step's loss until it reads the values back in bulk: a replayed step overwrites
        its static loss tensor - its value waits in the captured step's device ring (graph.LOSS_RING slots, index = the
        step's count), no copy command between two graph launches; an eager step's loss is a tensor of its own"""
        g = self._gstep
        if g is not None and loss is g.loss:
            if g.loss_ring is not None:
                return int(g.last_T)
            return loss.detach().clone()
        return loss.detach()

    def train_step(self, inputs, labels):
        """inputs / labels as the collate function returned them (host, pinned) or already on the device.  A replayed step
        takes a host batch in through a side-stream copy that overlaps the replays still in flight (GraphedTrainStep._post);
        an eager step moves it to the device here."""
        loss = self._graph_step(inputs, labels)
        if loss is not None:
            self.graph_steps += 1
            return loss
        self.eager_steps += 1
        if th.device(self.device).type == 'cuda' and any(getattr(x, 'buf', None) is not None and not x.buf.is_cuda for x in inputs):
            inputs, labels = prepare_batch((inputs, labels), self.device)
        self.optimizer.zero_grad()
        if self.fused:
            loss = self.model.fused_loss(*inputs, labels)
        else:
            scores = self.model(*inputs)
            assert not th.isnan(scores).any()
            loss = nn.functional.nll_loss(scores, labels)
        loss.backward()
        if self.shard is not None:
            self._sync_grads()
        self.optimizer.step()
        return loss

    # ------------------------------------------------------------------ checkpoint / hooks (not in the reference)
    def state_dict(self):
        """plain tensors, numbers, strings, lists and dicts only: loadable with torch.load(weights_only=True).  RNG states
        ride along so that a resumed dropout / shuffled run continues the uninterrupted trajectory."""
        rng = dict(torch=th.get_rng_state())
        if th.cuda.is_available() and th.device(self.device).type == 'cuda':
            rng['cuda'] = th.cuda.get_rng_state(self.device)
        if self.fused:
            # the private generators the run draws from: the dropout nonces of the HIP path (ops._nonce) and the samplers'
            # shuffles (src/scripts/common.py: RandomSampler(generator=...)) - neither is part of torch's global state
            from . import ops
            if ops._NONCE['gen'] is not None:
                rng['nonce'] = ops._NONCE['gen'].get_state()
                rng['nonce_seed'] = int(ops._NONCE['seed'])
        gens = self._loader_generators()
        if gens:
            rng['samplers'] = [g.get_state() for g in gens]
        return dict(model=self.model.state_dict(), optimizer=self.optimizer.state_dict(),
                    scheduler=self.scheduler.state_dict(), epoch=self.epoch, batch=self.batch, best=list(self.best),
                    rng=rng)

    def load_state_dict(self, sd):
        self.model.load_state_dict(sd['model'])
        self.optimizer.load_state_dict(sd['optimizer'])
        self.scheduler.load_state_dict(sd['scheduler'])
        self.epoch, self.batch, self.best = sd['epoch'], sd['batch'], tuple(sd['best'])
        rng = sd.get('rng') or {}
        if 'torch' in rng:
            th.set_rng_state(rng['torch'].cpu())
        if 'cuda' in rng and th.cuda.is_available() and th.device(self.device).type == 'cuda':
            th.cuda.set_rng_state(rng['cuda'].cpu(), self.device)
        if 'nonce' in rng and self.fused:
            from . import ops
            ops.seed_dropout()
            ops._NONCE['gen'].set_state(rng['nonce'].cpu())
            ops._NONCE['seed'] = th.initial_seed()          # (keyed by the seed in effect now: no implicit restart at the next draw)
        for g, st in zip(self._loader_generators(), rng.get('samplers') or []):
            g.set_state(st.cpu())
        self._gstep = None                 # optimizer state tensors were replaced: a captured step would use stale ones
        if self.fused:
            from . import ops
            ops.weights_changed()          # cached bf16 copies / column scales belong to the old weights
            self.model.__dict__.pop('_srec_state', None)
            if hasattr(self.model, 'table_written'):
                self.model.table_written() # notes about rows prepared / copies written by the old optimizer's row pass

    def _loader_generators(self):
        """the torch.Generators that shuffle the training loader (sampler / batch sampler and what they wrap), in a fixed order"""
        out, seen, todo = [], set(), [self.train_loader]
        while todo:
            o = todo.pop(0)
            if o is None or id(o) in seen:
                continue
            seen.add(id(o))
            g = getattr(o, 'generator', None)
            if isinstance(g, th.Generator) and all(g is not h for h in out):
                out.append(g)
            for name in ('sampler', 'batch_sampler', 'base', 'base_sampler'):
                todo.append(getattr(o, name, None))
        return out

    def _ckpt_path(self, path=None):
        """with a row-sharded table every rank owns different rows (and Adam moments): one file per rank"""
        path = path or self.checkpoint
        if self.shard is not None and self.shard.world > 1:
            path = '%s.rank%dof%d' % (path, self.shard.rank, self.shard.world)
        return path

    def save_checkpoint(self, path=None):
        import os
        path = self._ckpt_path(path)
        tmp = path + '.tmp'
        th.save(self.state_dict(), tmp)
        os.replace(tmp, path)              # a crash mid-write never leaves a truncated checkpoint behind

    def load_checkpoint(self, path=None):
        # weights_only=True: a checkpoint is data (tensors / numbers / containers), never pickled code
        self.load_state_dict(th.load(self._ckpt_path(path), map_location=self.device, weights_only=True))

    def _emit(self, **event):
        for h in self.hooks:
            h(event)

    def train(self, epochs, log_interval=100):
        import os
        done = 0
        if self.checkpoint and self.resume and os.path.exists(self._ckpt_path()):
            self.load_checkpoint()
            done = self.epoch                # `epochs` is the total of the interrupted run
            self._print(f'Resumed from {self._ckpt_path()}: epoch {self.epoch}, batch {self.batch}')
        max_mrr, max_hit, bad_counter = self.best
        t = time.time()
        mean_loss = 0
        evaluate(self.model, self.test_loader, self.device)
        for _ in range(epochs - done):
            self.model.train()
            pending = []                  # device-side loss values not read back yet

            def flush():
                # The reference reads loss.item() after every step (train.py:99-104), which drains the GPU queue each time.
                # Same numbers, same NaN check, same running mean - read back in one go where they are needed (the log
                # line, the end of the epoch), so the host prepares batch i+1 while the GPU runs step i.
                nonlocal mean_loss
                if pending:
                    ring = None
                    if any(isinstance(v, int) for v in pending):                # replayed steps: slots of the device loss ring
                        ring = self._gstep.loss_ring.tolist()
                        self._gstep.check()                 # (the host is synchronised here anyway: batch-intake fault flag)
                    eager = [v for v in pending if not isinstance(v, int)]
                    eager = iter(th.stack(eager).tolist()) if eager else iter(())
                    for v in pending:
                        v = ring[v % len(ring)] if isinstance(v, int) else next(eager)
                        assert v == v, 'loss is NaN'
                        self.loss_trace.append(v)
                        mean_loss += v / log_interval
                    pending.clear()
            for batch in self.train_loader:
                inputs, labels = batch                   # (host batches: train_step moves / stages them as its path needs)
                if not self.fused:
                    inputs, labels = prepare_batch(batch, self.device)
                pending.append(self._loss_handle(self.train_step(inputs, labels)))
                if (self.batch > 0 and self.batch % log_interval == 0) or len(pending) >= 256:
                    flush()
                if self.batch > 0 and self.batch % log_interval == 0:
                    self._print(f'Batch {self.batch}: Loss = {mean_loss:.4f}, Time Elapsed = {time.time() - t:.2f}s')
                    self._emit(kind='interval', batch=self.batch, loss=mean_loss, seconds=time.time() - t)
                    t = time.time()
                    mean_loss = 0
                self.batch += 1
            flush()
            if self._gstep is not None:
                self._gstep.check()            # end of the epoch: also a run / tail shorter than a flush interval is checked
            self.scheduler.step()
            mrr, hit = evaluate(self.model, self.test_loader, self.device)
            self._print(f'Epoch {self.epoch}: MRR = {mrr * 100:.3f}%, Hit = {hit * 100:.3f}%')
            self._emit(kind='epoch', epoch=self.epoch, mrr=mrr, hit=hit)
            stop = False
            if mrr < max_mrr and hit < max_hit:
                bad_counter += 1
                stop = bad_counter == self.patience
            else:
                bad_counter = 0
            if stop:
                break
            max_mrr = max(max_mrr, mrr)
            max_hit = max(max_hit, hit)
            self.epoch += 1
            self.best = (max_mrr, max_hit, bad_counter)
            if self.checkpoint:
                self.save_checkpoint()
        return max_mrr, max_hit
