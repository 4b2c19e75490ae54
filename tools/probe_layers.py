"""Per-hop sizes and layer-kernel times of one eval forward on a BASELINE shape (HIP events around rg_layer_fwd)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd import engine
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.models import RED_GNN_trans
from red_gnn_amd.synthetic import SHAPES, make_shape

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
engine.FORCE_WALK = int(sys.argv[3]) if len(sys.argv) > 3 else 0
shape = SHAPES[cfg]
kg = make_shape(cfg)
loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)


class P:
    n_layer, hidden_dim, attn_dim, n_rel, act, dropout = shape["n_layer"], shape["hidden_dim"], shape["attn_dim"], kg.n_rel, "relu", 0.0


torch.manual_seed(0)
model = RED_GNN_trans(P, loader).cuda().eval()
q = np.arange(B) % loader.n_test
subs = np.array([loader.test_q[i][0] for i in q]); rels = np.array([loader.test_q[i][1] for i in q])
ev = []
engine.KERNEL_EVENTS = ev
with torch.no_grad():
    for it in range(3):
        ev.clear()
        trace = []
        torch.cuda.synchronize()
        s = model(subs, rels, mode="test", trace=trace)
        torch.cuda.synchronize()
d = shape["hidden_dim"]
print("%s B=%d n_ent=%d |KG|=%d walk forced: %s" % (cfg, B, kg.n_ent, loader.tgraph.n_fact, engine.FORCE_WALK or "auto"))
indeg = np.bincount(np.concatenate([kg.facts[:, 2], kg.train[:, 2], kg.facts[:, 0], kg.train[:, 0], np.arange(kg.n_ent)]), minlength=kg.n_ent)
fr = engine.Frontier(kg.n_ent, B, 2)
n_old = B
tot = 0.0
for i, ((e0, e1, ne, nn), t) in enumerate(zip(ev, trace)):
    ms = e0.elapsed_time(e1)
    tot += ms
    by = ne * (4 * d + 16) + nn * 4 * d
    cand = int(indeg[t["nodes"][:, 1].cpu().numpy()].sum())
    ld = max(16, (d + 3) // 4 * 4)
    fr.level = i + 1
    print("hop %d: N=%9d (%.1f%% of B*n_ent) E=%10d (%.0f%% of %d candidate in-edges)  layer_fwd %.3f ms  %.2f G edges/s  alg %.0f GB/s (%.2f of 8 TB/s)  L2-gather %.2f"
          % (i, nn, 100.0 * nn / (B * kg.n_ent), ne, 100.0 * ne / max(cand, 1), cand, ms, ne / ms / 1e6, by / ms / 1e6, by / ms / 1e6 / 8000,
             ne * 4 * d / ms / 1e6 / 18800))
    n_old = nn
by_tot = sum(ne * (4 * d + 16) + nn * 4 * d for (_, _, ne, nn) in ev)
print("layer_fwd total %.3f ms  alg frac of HBM %.2f" % (tot, by_tot / tot / 1e6 / 8000))
