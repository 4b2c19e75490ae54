"""Training entry point: the reference's loop (Static/transductive/train.py:9-131) on the MI355X path.

Same two flags, same per-dataset hyper-parameter table, same 50-epoch loop and result files; the GPU is chosen by
LOCAL_RANK / --gpu instead of parsing nvidia-smi (utils.select_gpu of the reference is NVIDIA tooling).
Under ``python -m torch.distributed.run --nproc-per-node N train.py ...`` the batches are sharded by query over N GPUs
(RCCL; see red_gnn_amd/base_model.py) and rank 0 writes the result files.
"""
import argparse
import os

import numpy as np
import torch

from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader

# (lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch) — train.py:45-111
PRESETS = {
    "family": (0.0036, 0.999, 0.000017, 48, 5, 3, 0.29, "relu", 20, 50),
    "umls": (0.0012, 0.9917, 0.000115, 48, 5, 4, 0.0024, "relu", 20, 50),
    "WN18RR": (0.0021, 0.9962, 0.000037, 48, 5, 5, 0.0067, "tanh", 100, 50),
    "fb15k-237": (0.0009, 0.9938, 0.000080, 48, 5, 4, 0.0391, "relu", 5, 1),
    "nell": (0.0011, 0.9938, 0.000089, 48, 5, 5, 0.2593, "relu", 5, 1),
    "YAGO": (0.0003, 0.997, 0.000111, 48, 5, 3, 0.2131, "relu", 3, 1),
}


class Options(object):
    pass


def main():
    parser = argparse.ArgumentParser(description="RED-GNN on MI355X")
    parser.add_argument("--data_path", type=str, default="data/family/")
    parser.add_argument("--seed", type=int, default=1234)
    parser.add_argument("--gpu", type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    parser.add_argument("--epochs", type=int, default=50)
    parser.add_argument("--ids", type=str, default=None, help="npz of pre-parsed id triples (n_ent, n_rel, facts, train, valid, test) instead of text files")
    parser.add_argument("--preset", type=str, default=None, help="hyper-parameter preset name (default: the dataset directory name)")
    args = parser.parse_args()

    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    dataset = args.preset or [p for p in args.data_path.split("/") if p][-1]
    os.makedirs("results", exist_ok=True)
    opts = Options()
    opts.perf_file = os.path.join("results", dataset + "_perf.txt")
    torch.cuda.set_device(args.gpu)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.distributed.init_process_group("nccl")        # RCCL on ROCm
    rank0 = world == 1 or torch.distributed.get_rank() == 0
    print("gpu:", args.gpu)

    loader = DataLoader(ids=dict(np.load(args.ids))) if args.ids else DataLoader(args.data_path)
    opts.n_ent, opts.n_rel = loader.n_ent, loader.n_rel
    (opts.lr, opts.decay_rate, opts.lamb, opts.hidden_dim, opts.attn_dim, opts.n_layer, opts.dropout, opts.act,
     opts.n_batch, opts.n_tbatch) = PRESETS.get(dataset, PRESETS["family"])
    config_str = "%.4f, %.4f, %.6f,  %d, %d, %d, %d, %.4f,%s\n" % (
        opts.lr, opts.decay_rate, opts.lamb, opts.hidden_dim, opts.attn_dim, opts.n_layer, opts.n_batch, opts.dropout, opts.act)
    if rank0:
        print(config_str)
        with open(opts.perf_file, "a+") as f:
            f.write(config_str)

    model = BaseModel(opts, loader)
    best_mrr, best_str = 0, ""
    for epoch in range(args.epochs):
        mrr, out_str = model.train_batch(epoch=epoch)
        if rank0:
            with open(opts.perf_file, "a+") as f:
                f.write(out_str)
        if mrr > best_mrr:
            best_mrr, best_str = mrr, out_str
            if rank0:
                print(str(epoch) + "\t" + best_str)
    if rank0:
        print(best_str)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
