"""C2: one batch of 1024 queries on one stream against the same queries as 2 x 512 / 4 x 256 on concurrent streams (does the issue-bound
dense kernel of one half hide under the L2-bound walk of the other?).  usage: python tools/probe_two_streams.py [dense grid cap]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.load_data import DataLoader
from red_gnn_amd.models import RED_GNN_trans
from red_gnn_amd.synthetic import SHAPES, make_shape

shape = SHAPES["C2"]
kg = make_shape("C2", seed=1234)
loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)


class P:
    n_layer, hidden_dim, attn_dim, n_rel, act, dropout = shape["n_layer"], shape["hidden_dim"], shape["attn_dim"], kg.n_rel, "relu", 0.0


torch.manual_seed(1234)
model = RED_GNN_trans(P, loader).cuda().eval()
B = 1024
subs, rels, *_ = loader.get_batch_csr(np.arange(B), data="test")
subs, rels = np.asarray(subs), np.asarray(rels)
for use_graphs in (False, True):
    model.use_graphs = use_graphs
    for parts in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        chunks = np.array_split(np.arange(B), parts)

        def step():
            main = torch.cuda.current_stream()
            outs = []
            for st, idx in zip(streams, chunks):
                st.wait_stream(main)
                with torch.cuda.stream(st), torch.no_grad():
                    outs.append(model(subs[idx], rels[idx], mode="test"))
            for st in streams:
                main.wait_stream(st)
            return outs

        for _ in range(5):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("graphs=%s parts=%d: %.2f ms per 1024 queries" % (use_graphs, parts, dt * 1e3), flush=True)
