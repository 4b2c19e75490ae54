"""Two-rank (or N-rank) training against the single-process trajectory.

Launch with ``python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dist_train_check.py``.
Every rank trains the same model (same seed, dropout 0) on its shard of each batch through BaseModel's distributed path;
rank 0 then repeats the run single-process and compares parameters, epoch loss and metrics.  The backend is gloo so that
all ranks can share one GPU on a one-GPU box (RCCL wants one device per rank); on a multi-GPU node pass --backend nccl.
Prints DIST_TRAIN_CHECK_OK on success, exits non-zero otherwise.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import BaseModel          # noqa: E402
from red_gnn_amd.load_data import DataLoader          # noqa: E402
from red_gnn_amd.synthetic import make_synthetic_kg   # noqa: E402


def run(loader, group, n_batches):
    class Opt:
        lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.003, 0.99, 1e-5, 32, 5, 3, 0.0, "relu", 7, 16
        n_rel = loader.n_rel
    np.random.seed(7)
    torch.manual_seed(7)
    bm = BaseModel(Opt, loader, dist=group)
    mrr, out = bm.train_batch(epoch=0, max_batches=n_batches)
    return bm, mrr, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--batches", type=int, default=6)
    args = ap.parse_args()
    n_gpu = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(n_gpu, 1))
    dist.init_process_group(args.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    kg = make_synthetic_kg(300, 6, 3000, seed=5)
    ids = dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test)
    bm, mrr, out = run(DataLoader(ids=ids, verbose=False), dist, args.batches)
    flat = torch.cat([p.detach().reshape(-1) for p in bm.model.parameters()])
    # every rank holds the same parameters, bit for bit
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    same = torch.tensor([float(torch.equal(ref, flat))], device=flat.device)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    ok = bool(same.item() == 1.0)
    if rank == 0:
        bm1, mrr1, out1 = run(DataLoader(ids=ids, verbose=False), None, args.batches)
        flat1 = torch.cat([p.detach().reshape(-1) for p in bm1.model.parameters()])
        err = (flat - flat1).abs().max().item()
        print("world %d: params identical across ranks: %s; max |param - single-process| = %.3e; loss %.6f vs %.6f; valid MRR %.6f vs %.6f"
              % (world, ok, err, bm.last_epoch_loss, bm1.last_epoch_loss, mrr, mrr1))
        ok = ok and err < 2e-4 and abs(bm.last_epoch_loss - bm1.last_epoch_loss) <= 1e-4 * abs(bm1.last_epoch_loss) and abs(mrr - mrr1) < 2e-3
        ok = ok and out.split("[TIME]")[0][:20] == out1.split("[TIME]")[0][:20]
    flag = torch.tensor([1.0 if ok else 0.0], device=flat.device)
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    if flag.item() != 1.0:
        sys.exit(1)
    if rank == 0:
        print("DIST_TRAIN_CHECK_OK")


if __name__ == "__main__":
    main()
