"""Host-side cost of BaseModel.evaluate on the family graph at n_tbatch=50 (graph replay path)."""
import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
ids = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "family_ids.npz")))
loader = DataLoader(ids=ids, verbose=False)
class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, 50
    n_rel = loader.n_rel
torch.manual_seed(1234)
bm = BaseModel(Opt, loader)
bm.evaluate(); bm.evaluate()
pr = cProfile.Profile(); pr.enable()
torch.cuda.synchronize(); t0 = time.perf_counter()
bm.evaluate()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print("evaluate: %.3f s" % dt)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
