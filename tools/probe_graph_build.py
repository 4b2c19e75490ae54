"""Host (rg_graph_create) vs device (rg_graph_create_device) build time of the training graph of a BASELINE shape or the WN18RR fixture."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd.engine import Graph
from red_gnn_amd.synthetic import make_shape
for cfg in sys.argv[1:] or ["C2", "C3", "C4"]:
    kg = make_shape(cfg)
    trip = np.concatenate([kg.facts, kg.train], 0)
    dev = torch.as_tensor(trip, dtype=torch.int32).cuda()
    for name, fn in (("host  ", lambda: Graph(kg.n_ent, kg.n_rel, trip)), ("device", lambda: Graph.from_device(kg.n_ent, kg.n_rel, dev))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g = fn()
        torch.cuda.synchronize()
        print("%s %s build: %.2f ms (%d fact rows)" % (cfg, name, (time.perf_counter() - t0) / 5 * 1e3, g.n_fact))
