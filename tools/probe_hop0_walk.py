"""Hop 0 (one source node per query) on the word-parallel walk (the library's pick) against the per-query walk: the bench step at C2 and
the family evaluation with the plan of level 1 forced.  python tools/probe_hop0_walk.py"""
import os, sys, time, json, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd import engine
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader

orig = engine.layer_fwd_plan
def forced(frontier, graph, level, n_old, n_new, n_edges, ld):
    return 1 if level == 1 else orig(frontier, graph, level, n_old, n_new, n_edges, ld)

ids = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "family_ids.npz")))
loader = DataLoader(ids=ids, verbose=False)
for tag, plan in (("library pick", orig), ("hop 0 per-query walk", forced), ("library pick", orig)):
    class Opt:
        lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, 50
        n_rel = loader.n_rel
    torch.manual_seed(1234)
    engine.layer_fwd_plan = plan
    bm = BaseModel(Opt, loader)
    for _ in range(4):
        bm.evaluate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        mrr, out = bm.evaluate()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
    print("family %-22s: %.0f queries/s  mrr %.6f" % (tag, (loader.n_valid + loader.n_test) / dt, mrr), flush=True)

engine.layer_fwd_plan = orig
# C2 / B = 1024: per-hop kernel times of the bench step, every hop forced to the per-query walk against the library's pick
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag, env in (("library pick", {}), ("per-query walk on every hop", {"RG_BENCH_FORCE_WALK": "1"})):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-family-eval",
                          "--no-dense-f32"], capture_output=True, text=True, env=dict(os.environ, **env)).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print("C2 B=1024 %-28s: %s" % (tag, ["hop %d %.3f ms" % (h["hop"], h["ms"]) for h in d["per_hop"]]), flush=True)
