"""Experiment: one eval step as K independent query sub-batches on K HIP streams (one host thread each), so that one
sub-batch's MFMA-bound dense kernel can overlap another's gather-bound message-passing kernel."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.models import RED_GNN_trans
from red_gnn_amd.synthetic import SHAPES, make_shape
from red_gnn_amd.utils import cal_ranks_csr

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
shape = SHAPES[cfg]
kg = make_shape(cfg, seed=1234)
loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)


class P:
    n_layer, hidden_dim, attn_dim, n_rel, act, dropout = shape["n_layer"], shape["hidden_dim"], shape["attn_dim"], kg.n_rel, "relu", 0.0


torch.manual_seed(1234)
base = RED_GNN_trans(P, loader).cuda().eval()
q_idx = np.arange(B) % loader.n_test


def run(K, steps=10, warmup=2):
    models = [base] + [RED_GNN_trans(P, loader).cuda().eval() for _ in range(K - 1)]
    for m in models[1:]:
        m.load_state_dict(base.state_dict())
    streams = [torch.cuda.Stream() for _ in range(K)]
    parts = [q_idx[k * B // K:(k + 1) * B // K] for k in range(K)]
    batches = [loader.get_batch_csr(p, data="test") for p in parts]
    out = [None] * K

    def work(k):
        subs, rels, a_ptr, a_idx, f_ptr, f_idx = batches[k]
        with torch.cuda.stream(streams[k]), torch.no_grad():
            scores = models[k](subs, rels, mode="test")
            out[k] = (cal_ranks_csr(scores, a_ptr, a_idx, f_ptr, f_idx), sum(models[k].last_stats["n_edges"]))

    def step():
        if K == 1:
            work(0)
        else:
            ts = [threading.Thread(target=work, args=(k,)) for k in range(K)]
            for t in ts: t.start()
            for t in ts: t.join()
    for _ in range(warmup): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    edges = sum(o[1] for o in out)
    mrr = float(torch.cat([o[0] for o in out]).reciprocal().mean())
    print("%s B=%d  K=%d streams: %.2f ms/step  %.3g edges/s  (mrr %.6f)" % (cfg, B, K, dt * 1e3, edges / dt, mrr))


for K in (1, 2, 4):
    run(K)
