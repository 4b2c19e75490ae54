"""Time rg_dense_fwd alone at the C2 / B=1024 hop-2 size (10.24 M rows, 89 % of them with an old state), for A/B runs of kernel builds
(RG_LIB=<variant .so>).  python tools/r3_dense_bench.py [precision] [d]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from red_gnn_amd import engine
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
torch.manual_seed(0)
dev = "cuda"
n, n_old, a, ap = 10_240_000, 9_070_000, 5, 8
agg = torch.randn(n, d, device=dev)
hprev = torch.tanh(torch.randn(n_old, d, device=dev))
prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
pos = torch.sort(torch.randperm(n, device=dev)[:n_old]).values
prev[pos] = torch.arange(n_old, dtype=torch.int32, device=dev)
gate = torch.nn.GRU(d, d).to(dev)
W_h, Ws = torch.randn(d, d, device=dev) / d ** 0.5, torch.randn(a, d, device=dev) / d ** 0.5
out = torch.empty_like(agg); a_out = torch.empty(n, ap, device=dev)
import ctypes
cnt = torch.tensor([n, 0, 0, 0], dtype=torch.int32, device=dev)
def run():
    engine.dense_fwd_dev(n, ctypes.c_void_p(cnt.data_ptr()), agg, hprev, prev, d, W_h, "relu", gate, out, Ws_next=Ws, attn_dim=a, ap=ap,
                         a_s_out=a_out, n_hint=n, precision=prec)
with torch.no_grad():
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
print("%s %s d=%d: %.3f ms per launch (min of 3 x 10; %s)" % (os.path.basename(os.environ.get("RG_LIB", "libredgnn.so")), prec, d, min(ts), ["%.3f" % t for t in ts]))
