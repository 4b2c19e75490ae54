"""Kernel trace of replayed WN18RR evaluation forwards (preset d=48, 5 hops, 50 queries): run under rocprofv3 --kernel-trace."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
from train import PRESETS
loader = DataLoader(ids=dict(np.load(os.path.join(ROOT, "tests", "golden", "WN18RR_ids.npz"))), verbose=False)
class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = PRESETS["WN18RR"]
    n_rel = loader.n_rel
torch.manual_seed(1234)
BaseModel.EVAL_LANES = 1
bm = BaseModel(Opt, loader)
bm.n_valid, bm.n_test = 200, 0
bm.model.eval()
for _ in range(5):
    bm._rank_split("valid", bm.n_valid)
torch.cuda.synchronize()
