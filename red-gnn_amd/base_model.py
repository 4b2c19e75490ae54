"""Trainer / evaluator with the reference's interface (Static/transductive/base_model.py:10-152).

``BaseModel(args, loader)``, ``.train_batch(epoch)`` and ``.evaluate(epoch)`` behave as the reference's: Adam +
ExponentialLR, the reference's (broadcast) loss, the NaN scrub after every step, filtered MRR / H@1 / H@10 over the
valid and test splits, one ``loader.shuffle_train()`` per epoch.  What differs is where the work runs: scores never
leave the device — ranking is ``rg_rank`` on CSR answer/filter lists instead of a host copy of ``[B, n_ent]`` scores and
two scipy sorts per batch (base_model.py:106-118, utils.py:7-14) — and the NVML / RSS memory monitors of the reference
(NVIDIA tooling) have no counterpart.

Multi-GPU (the reference has none): with ``torch.distributed`` initialised (one process per GPU, same
seed everywhere) every training batch is split over the ranks by query, each rank scales its loss to the global batch
(the reference's loss carries a factor n = batch size), parameter gradients are summed with one flat all-reduce before
``optimizer.step()``, and evaluation batches are dealt round-robin with four metric sums reduced at the end.  All ranks
hold identical parameters after every step; with dropout = 0 the trajectory equals the single-process one up to fp32
summation order.
"""
import time

import numpy as np
import torch
from torch.optim import Adam
from torch.optim.lr_scheduler import ExponentialLR

from . import sharding
from .models import RED_GNN_induc, RED_GNN_trans
from .utils import cal_performance, cal_ranks_csr

_OUT_FMT = ('[VALID] MRR:%.4f H@1:%.4f H@10:%.4f\t [TEST] MRR:%.4f H@1:%.4f H@10:%.4f '
            '\t[TIME] train:%.4f inference:%.4f\n')


def reference_loss(scores, tails):
    """base_model.py:58-60, literally: max_n keeps its [n,1] shape, so the sum runs over an
    [n,n] broadcast (= n x the per-query cross entropy).  Kept, because it scales every gradient."""
    rows = torch.arange(scores.shape[0], device=scores.device)
    row_max = scores.max(dim=1, keepdim=True).values                      # [n,1]
    log_norm = torch.log(torch.exp(scores - row_max).sum(dim=1))         # [n]
    return (row_max + log_norm - scores[rows, tails]).sum()              # [n,1] + [n] - [n] -> [n,n]


def _chunks(n, size):
    """Contiguous index ranges of at most `size` covering range(n)."""
    return [np.arange(lo, min(lo + size, n)) for lo in range(0, n, size)]


def _default_group():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None


class BaseModel(object):
    def __init__(self, args, loader, dist="auto"):
        """``dist``: a torch.distributed-like module (tests pass their own), None for single-process, or "auto" =
        torch.distributed when a process group with more than one rank is initialised."""
        self.dist = _default_group() if isinstance(dist, str) else dist
        self.world = self.dist.get_world_size() if self.dist is not None else 1
        self.rank = self.dist.get_rank() if self.dist is not None else 0
        # the inductive loader (two graphs) pairs with RED_GNN_induc (Static/inductive/base_model.py:14)
        net = RED_GNN_induc if getattr(loader, "inductive", False) else RED_GNN_trans
        self.model = net(args, loader).cuda()
        self.loader, self.args = loader, args
        for name in ("n_ent", "n_rel", "n_train", "n_valid", "n_test"):
            setattr(self, name, getattr(loader, name))
        for name in ("n_batch", "n_tbatch", "n_layer"):
            setattr(self, name, getattr(args, name))
        # evaluation batches fused per forward pass.  A query's scores do not depend on the other queries of its batch (nodes are
        # (query, entity) pairs and every sum is taken per node in CSR order), so k reference batches of n_tbatch evaluated as ONE pass
        # of k * n_tbatch queries give the same scores bit for bit and the same ranks; n_tbatch then only bounds memory, as in the
        # reference.  1 = one pass per reference batch (the default: what the reference does).
        self.eval_coalesce = max(1, int(getattr(args, "eval_coalesce", 1)))
        # fused=True: one multi-tensor kernel for all parameters (same update rule; the default issues a dozen launches)
        self.optimizer = Adam(self.model.parameters(), lr=args.lr, weight_decay=args.lamb, fused=True)
        self.scheduler = ExponentialLR(self.optimizer, args.decay_rate)
        self.smooth = 1e-5
        self.t_time = 0.0
        self.last_epoch_loss = float("nan")

    # ---- one epoch of training followed by evaluation (base_model.py:32-83) ------------------------------------
    def train_batch(self, epoch=-1, max_batches=None):
        batches = _chunks(self.loader.n_train, self.n_batch)[:max_batches]
        started = time.time()
        self.model.train()
        # the evaluator's captured forwards hold capacity-sized buffers per lane (WN18RR: 1.7 GB each): when they amount to a real
        # share of the device, give them back before the training epoch allocates its own state
        held = getattr(self.model, "replay_bytes_held", lambda: 0)()
        if held and torch.cuda.is_available() and held > torch.cuda.mem_get_info()[0] // 4:
            self.model.release_replay_buffers()
        losses = []
        cost_table = None
        if self.world > 1:
            # per-query subgraphs differ by far more than 2x (hub subjects): every batch is dealt over the ranks by the
            # estimated cost of its queries on this epoch's training graph (equal counts), not cut into contiguous blocks
            cost_table = sharding.entity_costs(self.loader._graph_base, self.n_ent)
        for idx in batches:
            if cost_table is None:
                part = np.arange(len(idx))
            else:
                part = sharding.split_batch(cost_table[self.loader.train_data[idx, 0]], self.world, self.rank)
            lo, hi = 0, len(part)
            self.model.zero_grad()
            if hi > lo:
                triple = self.loader.get_batch(idx[part])
                scores = self.model(triple[:, 0], triple[:, 1])
                tails = torch.as_tensor(triple[:, 2], dtype=torch.long, device=scores.device)
                # the [n,n] broadcast of the reference's loss spans the whole batch: n_global x this rank's cross entropies
                loss = reference_loss(scores, tails)
                if self.world > 1:
                    loss = loss * (len(idx) / (hi - lo))
                loss.backward()
                losses.append(loss.detach())
            if self.dist is not None:
                for p in self.model.parameters():          # a rank without queries still takes part in the all-reduce
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)
                sharding.allreduce_gradients(list(self.model.parameters()), self.dist)
            self.optimizer.step()
            self._scrub_nan()
        self.scheduler.step()
        self.t_time += time.time() - started
        device = next(self.model.parameters()).device
        epoch_loss = torch.stack(losses).sum() if losses else torch.zeros((), device=device)
        if self.dist is not None:
            epoch_loss = sharding.reduce_metrics(epoch_loss.reshape(1), self.dist)[0]
        self.last_epoch_loss = float(epoch_loss)
        valid_mrr, out_str = self.evaluate(epoch=epoch)
        self.loader.shuffle_train()
        return valid_mrr, out_str

    def _scrub_nan(self):
        """base_model.py:64-69: NaN parameters are replaced by a fresh numpy uniform draw — one draw per parameter
        and step, as the reference consumes them, but replaced on the device (one launch per parameter, no host
        synchronisation; infinities stay, as in the reference)."""
        for p in self.model.parameters():
            p.data.nan_to_num_(nan=np.random.random(), posinf=float("inf"), neginf=float("-inf"))

    # ---- filtered evaluation (base_model.py:85-152) ---------------------------------------------------------------
    # evaluation batches in flight on separate HIP streams, at most (family, n_tbatch = 50: 1 lane 119 k, 2-4 lanes 172 k, 8 lanes 263 k
    # queries/s in round 2's first version; with the split dense kernel 8 / 12 / 16 lanes: 323 k / 371 k / 430 k).  Every lane holds
    # its own captured forward with capacity-sized buffers (WN18RR: 1.7 GB per lane and batch shape), so the lane count also follows
    # the replay cache's budget (WN18RR 8 / 12 / 16 / 24 lanes: 93 k / 111 k / 105 k / 105 k queries/s under a 96 GB budget; with
    # 24 GB the lanes' buffers evicted each other and 12 lanes lost to 8)
    EVAL_LANES = 16

    def _eval_lanes(self, n_batches):
        from .models import _GraphedInference, _pad4, pad_attn
        m = self.model
        n_ent = max(int(getattr(self.loader, "n_ent", 0)), int(getattr(self.loader, "n_ent_ind", 0)), 1)
        try:
            need = _GraphedInference.bytes_needed(self.n_tbatch * self.eval_coalesce, n_ent, max(16, _pad4(m.hidden_dim)), pad_attn(m.attn_dim))
        except (AttributeError, ValueError):        # a model without the replayed forward
            need = 0
        budget = _GraphedInference.budget(next(m.parameters()).device)
        fit = self.EVAL_LANES if need <= 0 else int(budget // (3 * need // 2 + 1))
        lanes = min(self.EVAL_LANES, max(8, n_batches // 8))      # (umls, 28 batches: 8 lanes 364 k queries/s, 16 lanes 305 k)
        if need > (1 << 28):       # heavy forwards fill the chip by themselves: more lanes only compete for L2 (WN18RR, 1.7 GB per
            lanes = min(lanes, 8)  # lane: 8 / 16 / 24 / 32 lanes 116 k / 101 k / 110 k / 101 k queries/s, tools/probe_eval_lanes.py)
        while lanes > 1 and lanes > fit:
            lanes //= 2
        return max(1, min(lanes, n_batches))

    def _rank_split(self, data, n_data):
        """Filtered ranks of a split.  A 50-query batch (the reference's n_tbatch) is a chain of small dependent kernels that
        leaves most of the 256 CUs idle, so consecutive batches go to EVAL_LANES streams round-robin: each lane replays its own
        captured forward graph (own buffers) and ranks on its stream; the lanes run concurrently on the device."""
        mode = self.loader.eval_mode(data) if hasattr(self.loader, "eval_mode") else data
        device = next(self.model.parameters()).device
        batches = _chunks(n_data, self.n_tbatch * self.eval_coalesce)[self.rank::self.world]             # evaluation batches dealt round-robin
        n_lanes = self._eval_lanes(len(batches))
        main = torch.cuda.current_stream(device)
        if n_lanes > 1:
            lanes = self.__dict__.setdefault("_eval_streams", [])
            while len(lanes) < n_lanes:
                lanes.append(torch.cuda.Stream(device=device))
            for s in lanes[:n_lanes]:
                s.wait_stream(main)
        ranks = []
        with torch.no_grad():
            for i, idx in enumerate(batches):
                stream = main if n_lanes == 1 else lanes[i % n_lanes]
                with torch.cuda.stream(stream):
                    subs, rels, ans_ptr, ans_idx, filt_ptr, filt_idx = self.loader.get_batch_csr(idx, data=data, device_queries=True)
                    scores = self.model(subs, rels, mode=mode)
                    r = cal_ranks_csr(scores, ans_ptr, ans_idx, filt_ptr, filt_idx)
                    if stream is not main:
                        r.record_stream(main)
                    ranks.append(r)
        if n_lanes > 1:
            for s in lanes[:n_lanes]:
                main.wait_stream(s)
        return torch.cat(ranks).double() if ranks else torch.zeros(0, dtype=torch.float64, device=device)

    def _performance(self, ranks):
        """utils.cal_performance (utils.py:16-21); across ranks through the four sums it is made of."""
        if self.dist is None:
            return cal_performance(ranks.cpu().numpy())
        sums = torch.stack([(1.0 / ranks).sum(), (ranks <= 1).sum().double(), (ranks <= 10).sum().double(),
                            torch.tensor(float(ranks.numel()), dtype=torch.float64, device=ranks.device)])
        sums = sharding.reduce_metrics(sums, self.dist).cpu().numpy()
        return sums[0] / sums[3], sums[1] / sums[3], sums[2] / sums[3]

    def evaluate(self, epoch=-1):
        self.model.eval()
        started = time.time()
        valid = self._performance(self._rank_split("valid", self.n_valid))
        test = self._performance(self._rank_split("test", self.n_test))
        i_time = time.time() - started
        return valid[0], _OUT_FMT % (*valid, *test, self.t_time, i_time)
