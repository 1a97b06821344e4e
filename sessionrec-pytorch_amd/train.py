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
    return [x.to(device) for x in inputs], labels.to(device)


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
    def __init__(self, dataset, model, train_loader, test_loader, device, lr=1e-3, weight_decay=0, patience=3):
        self.dataset = dataset
        self.model = model
        params = fix_weight_decay(model) if weight_decay > 0 else model.parameters()
        self.fused = th.device(device).type == 'cuda' and hasattr(model, 'fused_loss')
        if self.fused:
            from .optim import FusedAdam
            self.optimizer = FusedAdam(params, lr=lr, weight_decay=weight_decay, model=model)
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

    def train_step(self, inputs, labels):
        self.optimizer.zero_grad()
        if self.fused:
            loss = self.model.fused_loss(*inputs, labels)
        else:
            scores = self.model(*inputs)
            assert not th.isnan(scores).any()
            loss = nn.functional.nll_loss(scores, labels)
        loss.backward()
        self.optimizer.step()
        return loss

    def train(self, epochs, log_interval=100):
        max_mrr, max_hit, bad_counter = 0, 0, 0
        t = time.time()
        mean_loss = 0
        evaluate(self.model, self.test_loader, self.device)
        for _ in range(epochs):
            self.model.train()
            for batch in self.train_loader:
                inputs, labels = prepare_batch(batch, self.device)
                loss = self.train_step(inputs, labels).item()
                assert loss == loss, 'loss is NaN'
                self.loss_trace.append(loss)
                mean_loss += loss / log_interval
                if self.batch > 0 and self.batch % log_interval == 0:
                    print(f'Batch {self.batch}: Loss = {mean_loss:.4f}, Time Elapsed = {time.time() - t:.2f}s')
                    t = time.time()
                    mean_loss = 0
                self.batch += 1
            self.scheduler.step()
            mrr, hit = evaluate(self.model, self.test_loader, self.device)
            print(f'Epoch {self.epoch}: MRR = {mrr * 100:.3f}%, Hit = {hit * 100:.3f}%')
            if mrr < max_mrr and hit < max_hit:
                bad_counter += 1
                if bad_counter == self.patience:
                    break
            else:
                bad_counter = 0
            max_mrr = max(max_mrr, mrr)
            max_hit = max(max_hit, hit)
            self.epoch += 1
        return max_mrr, max_hit
