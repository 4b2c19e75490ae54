import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
loader = DataLoader(ids=dict(np.load(os.path.join(root, "tests", "golden", "family_ids.npz"))), verbose=False)
class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, 50
    n_rel = loader.n_rel
torch.manual_seed(1234)
bm = BaseModel(Opt, loader)
for _ in range(4): bm.evaluate()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): bm.evaluate()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
