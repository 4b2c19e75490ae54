"""Host-side cost of the reference's training loop on the real family graph (preset: B=20, d=48)."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
if os.environ.get('RG_BLAS'): torch.backends.cuda.preferred_blas_library(os.environ['RG_BLAS'])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader

ids = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "family_ids.npz")))
loader = DataLoader(ids=ids, verbose=False)


class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 48, 5, 3, 0.29, "relu", 20, 50
    n_rel = loader.n_rel


np.random.seed(1234); torch.manual_seed(1234)
bm = BaseModel(Opt, loader)
bm.n_valid, bm.n_test = 100, 100
bm.train_batch(epoch=0, max_batches=20)
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
bm.train_batch(epoch=1, max_batches=100)
torch.cuda.synchronize(); pr.disable()
dt = time.perf_counter() - t0
print("100 training batches of 20 + eval of 200 queries: %.3f s (%.2f ms per batch)" % (dt, dt * 10))
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
