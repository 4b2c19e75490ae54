"""Trainer / evaluator with the reference's interface (Static/transductive/base_model.py:10-152).

``BaseModel(args, loader)``, ``train_batch(epoch)``, ``evaluate(epoch)`` behave as the reference's;
the evaluator ranks on the device (rg_rank) instead of copying [B, n_ent] scores to the host and
sorting them twice per batch with scipy (base_model.py:106-118, utils.py:7-14).
"""
import time

import numpy as np
import torch
from torch.optim import Adam
from torch.optim.lr_scheduler import ExponentialLR

from .models import RED_GNN_induc, RED_GNN_trans
from .utils import cal_performance, cal_ranks_csr


def reference_loss(scores, tails):
    """base_model.py:58-60, literally: max_n keeps its [n,1] shape, so the sum runs over an
    [n,n] broadcast (= n x the per-query cross entropy).  Kept, because it scales every gradient."""
    pos_scores = scores[torch.arange(len(scores), device=scores.device), tails]
    max_n = torch.max(scores, 1, keepdim=True)[0]
    return torch.sum(-pos_scores + max_n + torch.log(torch.sum(torch.exp(scores - max_n), 1)))


class BaseModel(object):
    def __init__(self, args, loader):
        # the inductive loader (two graphs) pairs with RED_GNN_induc (Static/inductive/base_model.py:14)
        self.model = (RED_GNN_induc if getattr(loader, "inductive", False) else RED_GNN_trans)(args, loader)
        self.model.cuda()
        self.loader = loader
        self.n_ent, self.n_rel = loader.n_ent, loader.n_rel
        self.n_batch, self.n_tbatch = args.n_batch, args.n_tbatch
        self.n_train, self.n_valid, self.n_test = loader.n_train, loader.n_valid, loader.n_test
        self.n_layer = args.n_layer
        self.args = args
        self.optimizer = Adam(self.model.parameters(), lr=args.lr, weight_decay=args.lamb)
        self.scheduler = ExponentialLR(self.optimizer, args.decay_rate)
        self.smooth = 1e-5
        self.t_time = 0

    def train_batch(self, epoch=-1, max_batches=None):
        epoch_loss = 0
        batch_size = self.n_batch
        n_batch = self.loader.n_train // batch_size + (self.loader.n_train % batch_size > 0)
        if max_batches is not None:
            n_batch = min(n_batch, max_batches)
        t_time = time.time()
        self.model.train()
        for i in range(n_batch):
            start, end = i * batch_size, min(self.loader.n_train, (i + 1) * batch_size)
            triple = self.loader.get_batch(np.arange(start, end))
            self.model.zero_grad()
            scores = self.model(triple[:, 0], triple[:, 1])
            loss = reference_loss(scores, torch.as_tensor(triple[:, 2], dtype=torch.long, device=scores.device))
            loss.backward()
            self.optimizer.step()
            # avoid NaN (base_model.py:64-69): one numpy draw per parameter, as the reference consumes them, but as a
            # masked fill on the device (no host synchronisation per parameter)
            for p in self.model.parameters():
                p.data.masked_fill_(p.data != p.data, np.random.random())
            epoch_loss += loss.item()
        self.scheduler.step()
        self.t_time += time.time() - t_time
        self.last_epoch_loss = epoch_loss
        valid_mrr, out_str = self.evaluate(epoch=epoch)
        self.loader.shuffle_train()
        return valid_mrr, out_str

    def _rank_split(self, data, n_data):
        batch_size = self.n_tbatch
        n_batch = n_data // batch_size + (n_data % batch_size > 0)
        ranking = []
        for i in range(n_batch):
            batch_idx = np.arange(i * batch_size, min(n_data, (i + 1) * batch_size))
            subs, rels, ap, ai, fp, fi = self.loader.get_batch_csr(batch_idx, data=data)
            with torch.no_grad():
                scores = self.model(subs, rels, mode=self.loader.eval_mode(data) if hasattr(self.loader, "eval_mode") else data)
                ranking.append(cal_ranks_csr(scores, ap, ai, fp, fi))
        return torch.cat(ranking).double().cpu().numpy()

    def evaluate(self, epoch=-1):
        self.model.eval()
        i_time = time.time()
        v_mrr, v_h1, v_h10 = cal_performance(self._rank_split("valid", self.n_valid))
        t_mrr, t_h1, t_h10 = cal_performance(self._rank_split("test", self.n_test))
        i_time = time.time() - i_time
        out_str = '[VALID] MRR:%.4f H@1:%.4f H@10:%.4f\t [TEST] MRR:%.4f H@1:%.4f H@10:%.4f \t[TIME] train:%.4f inference:%.4f\n' % (
            v_mrr, v_h1, v_h10, t_mrr, t_h1, t_h10, self.t_time, i_time)
        return v_mrr, out_str
