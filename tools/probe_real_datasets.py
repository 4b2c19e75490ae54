"""The reference's loop on the id triples of its own datasets (fixtures under tests/golden): one short training slice and
a full filtered evaluation per dataset with the reference's presets (train.py:45-111), random-init weights."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
from train import PRESETS

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ("family", "umls", "WN18RR"):
    loader = DataLoader(ids=dict(np.load(os.path.join(root, name + "_ids.npz"))), verbose=False)

    class Opt:
        lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = PRESETS[name]
        n_rel = loader.n_rel

    np.random.seed(1234); torch.manual_seed(1234)
    bm = BaseModel(Opt, loader)
    n_b = 40
    bm.n_valid_full, bm.n_test_full = bm.n_valid, bm.n_test
    bm.n_valid, bm.n_test = 100, 100
    bm.train_batch(epoch=0, max_batches=5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bm.train_batch(epoch=1, max_batches=n_b)
    torch.cuda.synchronize(); t_train = time.perf_counter() - t0
    bm.n_valid, bm.n_test = bm.n_valid_full, bm.n_test_full
    for _ in range(3):
        bm.evaluate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mrr, out = bm.evaluate()
    torch.cuda.synchronize(); t_eval = time.perf_counter() - t0
    nq = loader.n_valid + loader.n_test
    print("%-7s preset d=%d L=%d n_batch=%d n_tbatch=%d: %.2f ms per training batch (incl. a 200-query eval / %d batches); "
          "evaluation %d queries in %.3f s = %.0f queries/s" % (name, Opt.hidden_dim, Opt.n_layer, Opt.n_batch, Opt.n_tbatch,
                                                               t_train / n_b * 1e3, n_b, nq, t_eval, nq / t_eval))
