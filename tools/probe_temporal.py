"""Eval forward of the temporal variant on the ICEWS14-shaped synthetic of BASELINE configs[4] (C5)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.synthetic import SHAPES
from red_gnn_amd.temporal import T_RED_GNN
from red_gnn_amd.synthetic import make_temporal_shape

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sh = SHAPES["C5"]
rng = np.random.default_rng(1234)
n_ent, n_rel_base, n_time, n_q = sh["n_ent"], 230, 365, sh["n_triples"]
perm = rng.permutation(n_ent)
w = 1.0 / np.arange(1, n_ent + 1); w /= w.sum()
def ent(n):
    return np.where(rng.random(n) < 0.5, perm[rng.choice(n_ent, n, p=w)], rng.integers(0, n_ent, n))
h, t = ent(n_q), ent(n_q)
r = rng.integers(0, n_rel_base, n_q); tau = rng.integers(0, n_time, n_q)
quads = np.stack([h, r, t, tau], 1)
inv = np.stack([t, r + n_rel_base, h, tau], 1)                       # '~' inverse relations
n_rel = 2 * n_rel_base + 1                                           # + idd
idd = np.stack([np.arange(n_ent), np.full(n_ent, n_rel - 1), np.arange(n_ent), np.full(n_ent, n_time)], 1)   # sentinel = largest time id


class P:
    pass


p = P()
p.n_rel, p.n_ent, p.n_time = n_rel, n_ent, n_time + 1
p.hidden_dim, p.attn_dim, p.n_layer, p.act, p.device = sh["hidden_dim"], sh["attn_dim"], sh["n_layer"], "relu", "cuda"
p.graph = np.concatenate([quads, inv, idd], 0)
torch.manual_seed(0)
model = T_RED_GNN(p).cuda().eval()
batch = {"head": quads[:B, 0], "relation": quads[:B, 1], "time": quads[:B, 3]}
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = model(batch, mode="test")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    e = sum(model.last_stats["n_edges"])
    print("C5 temporal B=%d it%d: %.2f ms, edges %.3g per hop %s -> %.3g edges/s" % (B, it, dt * 1e3, e, model.last_stats["n_edges"], e / dt))

# one training step of the same model (mode='train': the batch's quadruples leave the graph; main.py's loss)
import torch.nn.functional as F
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batch["example_idx"] = np.arange(B)
tails = torch.as_tensor(quads[:B, 2], device="cuda")
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = __import__("red_gnn_amd.engine", fromlist=["x"]).TemporalGraph(n_ent, n_rel + 1, n_time + 1, model.quads, exclude=batch["example_idx"])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad()
    s = model(batch, mode="train")
    loss = F.nll_loss(torch.log(F.softmax(s, dim=1) + 1e-12), tails)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("C5 temporal train B=%d it%d: graph rebuild alone %.2f ms | fwd (incl. its own rebuild) %.2f ms  bwd %.2f ms  opt %.2f ms  loss %.4f"
          % (B, it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, loss.item()))
