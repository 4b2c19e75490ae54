import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
ids = dict(np.load(os.path.join(ROOT, "tests", "golden", "family_ids.npz")))
loader = DataLoader(ids=ids, verbose=False)
class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, 50
    n_rel = loader.n_rel
torch.manual_seed(1234)
BaseModel.EVAL_LANES = 1
bm = BaseModel(Opt, loader)
bm.n_valid, bm.n_test = 200, 0
bm.model.eval()
for _ in range(5):
    bm._rank_split("valid", bm.n_valid)
torch.cuda.synchronize()
