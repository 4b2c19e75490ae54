"""Accuracy and time of the fused dense kernel in its two precisions against an fp64 evaluation of the same operator.
usage: python tools/probe_dense_precision.py [n_rows] [d]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from red_gnn_amd import engine


def ref64(agg, hprev, prev_idx, W_h, act, gate, Ws, W_final):
    f = lambda t: t.double()
    x = f(agg) @ f(W_h).t()
    x = {"relu": torch.relu, "tanh": torch.tanh, "idd": lambda v: v}[act](x)
    h0 = torch.zeros_like(x)
    m = prev_idx >= 0
    h0[m] = f(hprev)[prev_idx[m].long()]
    gi = x @ f(gate.weight_ih_l0).t() + f(gate.bias_ih_l0)
    gh = h0 @ f(gate.weight_hh_l0).t() + f(gate.bias_hh_l0)
    d = x.shape[1]
    r = torch.sigmoid(gi[:, :d] + gh[:, :d]); z = torch.sigmoid(gi[:, d:2 * d] + gh[:, d:2 * d])
    n = torch.tanh(gi[:, 2 * d:] + r * gh[:, 2 * d:])
    h = (1 - z) * n + z * h0
    return h, h @ f(Ws).t(), h @ f(W_final).reshape(-1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = "cuda:0"
    torch.manual_seed(0)
    a, ap = 5, 8
    for act, big in (("relu", 1.0), ("relu", 3000.0), ("tanh", 1.0), ("idd", 30.0)):
        agg = torch.randn(n, d, device=dev) * big
        agg[::7] *= 1e-3                                    # small rows beside large ones
        agg[::11, ::3] = 0
        n_old = n // 2
        hprev = torch.tanh(torch.randn(n_old, d, device=dev))
        prev_idx = torch.randint(0, n_old, (n,), device=dev, dtype=torch.int32)
        prev_idx[torch.rand(n, device=dev) < 0.4] = -1
        prev_idx[: n // 8] = -1                             # whole tiles of new nodes (the W_hh products are skipped)
        W_h = torch.nn.Linear(d, d, bias=False).to(dev).weight.detach()
        gate = torch.nn.GRU(d, d).to(dev)
        Ws = torch.nn.Linear(d, a, bias=False).to(dev).weight.detach()
        W_final = torch.nn.Linear(d, 1, bias=False).to(dev).weight.detach()
        h64, as64, sc64 = ref64(agg, hprev, prev_idx, W_h, act, gate, Ws, W_final)
        out = {}
        for prec in ("f32", "f16x2"):
            h, a_s = engine.dense_fwd(agg, hprev, prev_idx, d, W_h, act, gate, Ws_next=Ws, attn_dim=a, ap=ap, precision=prec)
            eh = (h[:, :d].double() - h64).abs().max().item()
            ea = (a_s[:, :a].double() - as64).abs().max().item()
            out[prec] = (eh, ea)
        print("act=%s |agg|~%g  n=%d d=%d   max abs error vs fp64:  hidden f32 %.2e  f16x2 %.2e   a_s f32 %.2e  f16x2 %.2e"
              % (act, big, n, d, out["f32"][0], out["f16x2"][0], out["f32"][1], out["f16x2"][1]))
    # time (all rows old: every product runs)
    prev_idx = torch.randint(0, n_old, (n,), device=dev, dtype=torch.int32)
    for prec in ("f32", "f16x2"):
        for _ in range(3):
            engine.dense_fwd(agg, hprev, prev_idx, d, W_h, "relu", gate, Ws_next=Ws, attn_dim=a, ap=ap, precision=prec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            engine.dense_fwd(agg, hprev, prev_idx, d, W_h, "relu", gate, Ws_next=Ws, attn_dim=a, ap=ap, precision=prec)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = n * (2.0 * d * d * 7 + 2.0 * d * ap)
        by = n * (3 * d * 4 + ap * 4 + 4)
        print("%-6s %.3f ms  %.1f useful TFLOP/s  %.2f TB/s of rows" % (prec, ms, fl / ms * 1e-9, by / ms * 1e-9))


main()
