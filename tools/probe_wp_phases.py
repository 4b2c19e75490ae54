"""Where a word-parallel wave spends its cycles, per hop: needs a library built from the timing variant of layer_fwd_wp.hip
(python tools/make_wp_timing_variant.py, then RG_LIB=red-gnn_amd/libredgnn_wpt.so: cycle counters around item load / fill / phase 1 / phase 2, exported as rg_debug_wp_timing)."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd import engine, _lib
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.models import RED_GNN_trans
from red_gnn_amd.synthetic import SHAPES, make_shape
import bench

lib = ctypes.CDLL(os.environ["RG_LIB"])
def read(reset=1):
    buf = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    assert lib.rg_debug_wp_timing(buf, reset) == 0
    return list(buf)

names = ["ticket/other", "item load", "fill", "phase 1", "phase 2"]
for cfg, B in (("C2", 1024), ("C3", 256), ("family", 50)):
    if cfg == "family":
        ids = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "family_ids.npz")))
        loader = DataLoader(ids=ids, verbose=False)
        class P: hidden_dim, attn_dim, n_layer, dropout, act = 64, 5, 3, 0.0, "relu"
        P.n_rel = loader.n_rel
        model = RED_GNN_trans(P, loader).cuda().eval()
    else:
        kg = make_shape(cfg, seed=1234)
        loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)
        model = RED_GNN_trans(bench.Params(SHAPES[cfg], kg.n_rel), loader).cuda().eval()
    model.use_graphs = False
    subs, rels, *_ = loader.get_batch_csr(np.arange(B) % loader.n_test, data="test")
    orig = engine.layer_fwd
    hop = [0]
    def timed(*a, **k):
        read()
        out = orig(*a, **k)
        t = read()
        if t[8]:
            tot = sum(t[:5])
            print("%s B=%d hop %d: %d waves, %.0f k cycles per wave, %d flushes, %.0f edges per flush | " % (cfg, B, hop[0], t[8], tot / t[8] / 1e3, t[7], t[6] / max(t[7], 1))
                  + "  ".join("%s %.0f%%" % (n, 100.0 * v / tot) for n, v in zip(names, t[:5])), flush=True)
        hop[0] += 1
        return out
    with torch.no_grad():
        model(subs, rels, mode="test")
        engine.layer_fwd = timed
        model(subs, rels, mode="test")
        engine.layer_fwd = orig
