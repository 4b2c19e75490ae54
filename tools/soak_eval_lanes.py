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
torch.manual_seed(7)
bm = BaseModel(Opt, loader)
bm.model.eval()
BaseModel.EVAL_LANES = 1
ref_v = bm._rank_split("valid", bm.n_valid).cpu().numpy()
ref_t = bm._rank_split("test", bm.n_test).cpu().numpy()
BaseModel.EVAL_LANES = 16
bad = 0
t0 = time.time()
for i in range(300):
    v = bm._rank_split("valid", bm.n_valid).cpu().numpy()
    t = bm._rank_split("test", bm.n_test).cpu().numpy()
    if not (np.array_equal(v, ref_v) and np.array_equal(t, ref_t)):
        bad += 1
print("300 passes at 16 lanes: %d differed from the single-lane ranks; %.1f s" % (bad, time.time() - t0))
