import sys, numpy as np, torch
sys.path.insert(0, ".")
import red_gnn_amd  # noqa
from red_gnn_amd import _lib, engine
torch.manual_seed(0)
dev = "cuda"
x = torch.randn(8, 64, device=dev)
back = torch.empty_like(x); parts = torch.empty(8, 64, 4, device=dev)
_lib.check(_lib.lib().rg_split3_roundtrip(_lib.ptr(x), 8, 64, _lib.ptr(back), _lib.ptr(parts), _lib.stream_ptr()))
torch.cuda.synchronize()
err = (back - x).abs()
print("roundtrip max abs err", float(err.max()), "n bad", int((back != x).sum()), "of", x.numel())
bad = (back != x).nonzero()[:6]
for r, c in bad.tolist():
    m = float(x[r].abs().max())
    print("x=%r back=%r rowmax=%g parts=%s" % (float(x[r, c]), float(back[r, c]), m, parts[r, c].tolist()))
# dense: identity-ish weights
d, n = 64, 32
agg = torch.randn(n, d, device=dev)
gate = torch.nn.GRU(d, d).to(dev)
W_h = torch.eye(d, device=dev)
for prec in ("f32", "f16x2", "f16x3"):
    h, _ = engine.dense_fwd(agg, None, None, d, W_h, "idd", gate, precision=prec)
    x_ = agg @ W_h.t()
    href = torch.gru_cell(x_, torch.zeros(n, d, device=dev), gate.weight_ih_l0, gate.weight_hh_l0, gate.bias_ih_l0, gate.bias_hh_l0)
    print(prec, "max err", float((h - href).abs().max()))
# zero gate weights: h = (1 - z) n with z = sigmoid(b), n = tanh(b_in + r b_hn): independent of x
with torch.no_grad():
    gate.weight_ih_l0.zero_(); gate.weight_hh_l0.zero_()
for prec in ("f32", "f16x3"):
    h, _ = engine.dense_fwd(agg, None, None, d, W_h, "idd", gate, precision=prec)
    href = torch.gru_cell(agg, torch.zeros(n, d, device=dev), gate.weight_ih_l0, gate.weight_hh_l0, gate.bias_ih_l0, gate.bias_hh_l0)
    print(prec, "zero gate weights: max err", float((h - href).abs().max()))
# only W_in nonzero = identity (n gate sees x): checks stage 1 + one gate product
with torch.no_grad():
    gate.weight_ih_l0[2 * d:] = torch.eye(d, device=dev)
for prec in ("f32", "f16x3"):
    h, _ = engine.dense_fwd(agg, None, None, d, W_h, "idd", gate, precision=prec)
    href = torch.gru_cell(agg, torch.zeros(n, d, device=dev), gate.weight_ih_l0, gate.weight_hh_l0, gate.bias_ih_l0, gate.bias_hh_l0)
    e = (h - href).abs()
    print(prec, "W_in = I: max err", float(e.max()), "worst cols", e.max(0).values.topk(5).indices.tolist(), "worst rows", e.max(1).values.topk(5).indices.tolist())
