"""Timing probe: one training step (forward + reference loss + HIP backward + Adam) on a BASELINE shape."""
import sys, time, os
import numpy as np, torch
if os.environ.get('RG_BLAS'): torch.backends.cuda.preferred_blas_library(os.environ['RG_BLAS'])
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import reference_loss
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.models import RED_GNN_trans
from red_gnn_amd.synthetic import SHAPES, make_shape

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shape = SHAPES[cfg]
kg = make_shape(cfg)
loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)


class P:
    n_layer, hidden_dim, attn_dim, n_rel, act, dropout = shape["n_layer"], shape["hidden_dim"], shape["attn_dim"], kg.n_rel, "relu", 0.1


torch.manual_seed(0)
model = RED_GNN_trans(P, loader).cuda().train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
trip = loader.train_data[:B]
tails = torch.as_tensor(trip[:, 2], device="cuda")
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    scores = model(trip[:, 0], trip[:, 1])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss = reference_loss(scores, tails)
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    e = sum(model.last_stats["n_edges"])
    print("%s B=%d it%d: fwd %.2f ms  bwd %.2f ms  opt %.2f ms  edges %.3g  -> %.3g edges/s fwd+bwd  loss %.4f"
          % (cfg, B, it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, e, e / (t2 - t0), loss.item()))
