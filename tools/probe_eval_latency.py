"""Where a 50-query evaluation batch on family spends its time: GPU time of one replayed forward graph (back-to-back replays, no
host work between them), of the rank kernel, and the host-side time of BaseModel.evaluate per batch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.base_model import BaseModel
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.utils import cal_ranks_csr
ids = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "family_ids.npz")))
loader = DataLoader(ids=ids, verbose=False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
class Opt:
    lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, B
    n_rel = loader.n_rel
torch.manual_seed(1234)
bm = BaseModel(Opt, loader)
for _ in range(3):
    bm.evaluate()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    bm.evaluate()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
nq = loader.n_valid + loader.n_test
nb = (loader.n_valid + B - 1) // B + (loader.n_test + B - 1) // B
print("evaluate: %.1f ms per pass, %d batches -> %.3f ms per batch, %.0f queries/s" % (dt * 1e3, nb, dt * 1e3 / nb, nq / dt))
model = bm.model
g = [v for v in model._graphed.values() if v.n == B][0]
subs, rels, ap, ai, fp, fi = loader.get_batch_csr(np.arange(B), data="valid", device_queries=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    g.cuda_graph.replay()
e1.record(); torch.cuda.synchronize()
print("graph replay alone: %.3f ms per forward (GPU, back to back)" % (e0.elapsed_time(e1) / 200))
t0 = time.perf_counter()
for _ in range(200):
    g.cuda_graph.replay()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host time to enqueue a replay: %.3f ms" % ((t1 - t0) / 200 * 1e3))
with torch.no_grad():
    s = model(subs, rels, mode="valid")
    e0.record()
    for _ in range(200):
        r = cal_ranks_csr(s, ap, ai, fp, fi)
    e1.record(); torch.cuda.synchronize()
    print("rank kernel: %.3f ms" % (e0.elapsed_time(e1) / 200))
    t0 = time.perf_counter()
    for _ in range(200):
        s = model(subs, rels, mode="valid")
        r = cal_ranks_csr(s, ap, ai, fp, fi)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("model() + ranks per batch, same batch repeated: %.3f ms" % ((t1 - t0) / 200 * 1e3))
    t0 = time.perf_counter()
    for _ in range(200):
        subs, rels, ap, ai, fp, fi = loader.get_batch_csr(np.arange(B), data="valid", device_queries=True)
    t1 = time.perf_counter()
    print("get_batch_csr host time: %.3f ms" % ((t1 - t0) / 200 * 1e3))
for lanes in (1, 2, 4, 8, 16):
    BaseModel.EVAL_LANES = lanes
    for _ in range(3):
        bm.evaluate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        bm.evaluate()
    th = time.perf_counter()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("EVAL_LANES=%d: %.1f ms per pass -> %.0f queries/s (host loop done after %.1f ms per pass)" % (lanes, dt * 1e3, nq / dt, (th - t0) / 5 * 1e3))
import cProfile, pstats
BaseModel.EVAL_LANES = 8
pr = cProfile.Profile(); pr.enable()
bm.evaluate()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
