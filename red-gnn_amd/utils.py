"""Metrics with the reference's names (Static/transductive/utils.py:7-21).

``cal_ranks`` takes the same dense arguments as the reference but counts on the GPU (rg_rank)
instead of sorting twice with scipy; ``cal_ranks_csr`` is the form the evaluator uses (no dense
[B, n_ent] label / filter matrices are ever built).  NVIDIA-only tooling of the reference's utils.py
(select_gpu parsing nvidia-smi, NVML monitors) has no counterpart here.
"""
import numpy as np
import torch

from . import engine


def cal_ranks_csr(scores, ans_ptr, ans_idx, filt_ptr, filt_idx):
    """Filtered ranks as an fp32 device tensor, in (query, ascending answer) order."""
    return engine.rank(scores.contiguous(), ans_ptr, ans_idx, filt_ptr, filt_idx)


def cal_ranks(scores, labels, filters, device="cuda"):
    """utils.py:7-14 — same signature and result (list of ranks in np.nonzero(labels) order)."""
    scores_t = torch.as_tensor(np.asarray(scores), dtype=torch.float32).to(device)
    labels, filters = np.asarray(labels), np.asarray(filters)

    def csr(m):
        rows, cols = np.nonzero(m)
        ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=m.shape[0]))])
        to = lambda a: torch.as_tensor(a, dtype=torch.int32).to(device)
        return to(ptr), to(cols)

    ap, ai = csr(labels)
    fp, fi = csr(filters)
    return list(cal_ranks_csr(scores_t, ap, ai, fp, fi).double().cpu().numpy())


def cal_performance(ranks):
    """utils.py:17-21."""
    ranks = np.asarray(ranks, dtype=np.float64)
    mrr = (1. / ranks).sum() / len(ranks)
    h_1 = np.sum(ranks <= 1) * 1.0 / len(ranks)
    h_10 = np.sum(ranks <= 10) * 1.0 / len(ranks)
    return mrr, h_1, h_10
