"""Evaluation throughput on the real family graph (BASELINE configs[0] shape: n_layer=3, d=64, n_tbatch=50):
BaseModel.evaluate = forward + filtered ranking over all valid + test queries, random-init weights."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader

ids = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "family_ids.npz")))
loader = DataLoader(ids=ids, verbose=False)
for tb in (50, 500):
    class Opt:
        lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, tb
        n_rel = loader.n_rel
    torch.manual_seed(1234)
    bm = BaseModel(Opt, loader)
    bm.evaluate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mrr, out = bm.evaluate()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nq = loader.n_valid + loader.n_test
    print("family eval n_tbatch=%d: %d queries in %.3f s -> %.0f queries/s  (%s)" % (tb, nq, dt, nq / dt, out.strip()[:60]))
