import sys, os, numpy as np, torch
sys.path.insert(0, ".")
import red_gnn_amd  # noqa
from red_gnn_amd import _lib, engine
torch.manual_seed(0)
dev = "cuda"
d, n = 64, 4000
agg = torch.randn(n, d, device=dev)
gate = torch.nn.GRU(d, d).to(dev)
W_h = torch.randn(d, d, device=dev) / 8
hprev = torch.tanh(torch.randn(n // 2, d, device=dev))
prev = torch.randint(-1, n // 2, (n,), device=dev, dtype=torch.int32)
with torch.no_grad():
    x_ = agg @ W_h.t()
    h0 = torch.zeros(n, d, device=dev); m = prev >= 0; h0[m] = hprev[prev[m].long()]
    href = torch.gru_cell(x_, h0, gate.weight_ih_l0, gate.weight_hh_l0, gate.bias_ih_l0, gate.bias_hh_l0)
    h, _ = engine.dense_fwd(agg, hprev, prev, d, W_h, "idd", gate, precision="f16x3")
    e = (h - href).abs()
    print(os.environ.get("RG_LIB"), "max err %.3e" % float(e.max()))
