"""Training batches of the WN18RR preset (100 queries, 5 hops, d=48) on the id fixture; meant to run under rocprofv3 --stats."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
from train import PRESETS
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
loader = DataLoader(ids=dict(np.load(os.path.join(root, "WN18RR_ids.npz"))), verbose=False)
class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = PRESETS["WN18RR"]
    n_rel = loader.n_rel
np.random.seed(1234); torch.manual_seed(1234)
bm = BaseModel(Opt, loader)
bm.n_valid, bm.n_test = 50, 50
bm.train_batch(epoch=0, max_batches=5)
torch.cuda.synchronize(); t0 = time.perf_counter()
bm.train_batch(epoch=1, max_batches=30)
torch.cuda.synchronize(); print("30 batches: %.1f ms each" % ((time.perf_counter() - t0) / 30 * 1e3))
