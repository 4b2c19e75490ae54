"""Diagnostic: frontier expansion alone (reset + 3 hops, no read-back) captured in a HIP graph and replayed for changing
queries, against the eager frontier.  Only fixed-size bitmap kernels run here: safe under any stale state."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd import engine
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.synthetic import make_synthetic_kg

kg = make_synthetic_kg(500, 9, 6000, seed=8)
loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)
graph = loader.graph_for("test")
B, L = 37, 3
dev = torch.device("cuda")
fr_g = engine.Frontier(graph.n_ent, B, 2, dev)
fr_e = engine.Frontier(graph.n_ent, B, 2, dev)
q_static = torch.zeros(B, dtype=torch.int32, device=dev)


def enqueue():
    fr_g.reset(q_static)
    for _ in range(L):
        fr_g.expand_async(graph)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    enqueue()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
mode = os.environ.get("MODE", "graph")
if mode == "graph":
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg):
        enqueue()
rng = np.random.default_rng(0)
for it in range(6):
    q = torch.as_tensor(kg.test[rng.integers(0, len(kg.test), B)][:, 0].astype(np.int32)).cuda()
    q_static.copy_(q)
    if mode == "graph":
        cg.replay()
    else:
        enqueue()
    torch.cuda.synchronize()
    off = fr_g.count_ptr().value - fr_g.workspace.data_ptr()
    ctr = fr_g.workspace[off:off + 1024].view(torch.int32).cpu().numpy()
    print("   q ok:", bool((q_static == q).all()), "counters[0:8]", ctr[0:8].tolist(), "snap", ctr[72:76].tolist(), ctr[80:84].tolist(), ctr[88:92].tolist())
    try:
        got = fr_g.level_counts()[1:]
    except Exception as e:
        got = str(e)[-60:]
    fr_e.reset(q)
    ref = []
    for _ in range(L):
        n_new, n_e, _ = fr_e.expand(graph)
        ref.append((n_new, n_e))
    print(it, mode, "OK " if got == ref else "DIFF", got, ref)
