"""RED_GNN_trans / GNNLayer with the reference's nn.Module API and state-dict names
(Static/transductive/models.py:5-89), running the hot path in hand-written HIP kernels.

    model = RED_GNN_trans(params, loader).cuda()
    scores = model(subs, rels, mode='train'|'valid'|'test')      # fp32 [B, n_ent], 0 at unvisited entities

What is different inside (nothing is different outside):
  * no host round trip per layer: the frontier lives on the device as bitmaps (engine.Frontier),
    one 32-byte read-back per hop returns the node/edge counts;
  * models.py:29-39 (gathers, attention, torch_scatter sum) is one fused kernel, rg_layer_fwd, and
    its adjoint rg_layer_bwd; the three attention Linear layers are hoisted to per-node /
    per-relation / per-query projections (exact re-association);
  * dense algebra that is not on the E-proportional path (W_h, GRU cell, hoisted projections, W_final): inference runs it
    in one f32-MFMA kernel per layer (rg_dense_fwd); training keeps it in torch on the device, with the weight gradients
    of the node-row GEMMs issued in row-chunked batched form (``tall_linear``: a [m,n] = G^T X product over millions of
    rows otherwise lands on a handful of workgroups).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine


def _pad4(n):
    return (n + 3) // 4 * 4


_TALL_ROWS = 1 << 15        # below this a plain GEMM is as good
_TALL_CHUNKS = 256          # ~ one row chunk per CU


def _gram_tn(g, x):
    """g^T x for g [N,m], x [N,n] with N >> m,n: chunks of rows as one batched GEMM, then a sum over the chunks
    (fills the chip and shortens every fp32 accumulation chain by the chunk count)."""
    n_rows = g.shape[0]
    if n_rows < _TALL_ROWS:
        return g.t() @ x
    c = n_rows // _TALL_CHUNKS
    body = _TALL_CHUNKS * c
    out = torch.bmm(g[:body].view(_TALL_CHUNKS, c, -1).transpose(1, 2), x[:body].view(_TALL_CHUNKS, c, -1)).sum(0)
    if body < n_rows:
        out = out + g[body:].t() @ x[body:]
    return out


class _TallLinear(torch.autograd.Function):
    """F.linear(x, weight, bias) for node-row matrices: same forward, weight gradient through _gram_tn."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = _gram_tn(g, x.contiguous()) if ctx.needs_input_grad[1] else None
        gb = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def tall_linear(x, weight, bias=None):
    return _TallLinear.apply(x, weight, bias)


def gru_step(x, h0, gate):
    """Single-step nn.GRU (models.py:83) = torch.gru_cell, with the two gate GEMMs issued through tall_linear."""
    gi = tall_linear(x, gate.weight_ih_l0)
    gh = tall_linear(h0, gate.weight_hh_l0)
    return torch.ops.aten._thnn_fused_gru_cell(gi, gh, h0, gate.bias_ih_l0, gate.bias_hh_l0)[0]


class _Aggregate(torch.autograd.Function):
    """agg = rg_layer_fwd(...);  backward = rg_layer_bwd(...)."""

    @staticmethod
    def forward(ctx, hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha, frontier, graph, level, nodes_new, nodes_old, d, attn_dim):
        hidden, rela, a_s, a_r, a_q = (t.contiguous() for t in (hidden, rela, a_s, a_r, a_q))
        w_alpha, b_alpha = w_alpha.contiguous(), b_alpha.contiguous()
        agg = engine.layer_fwd(frontier, graph, level, nodes_new, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim)
        ctx.save_for_backward(hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha, nodes_old)
        ctx.misc = (frontier, graph, level, d, attn_dim)
        return agg

    @staticmethod
    def backward(ctx, grad_agg):
        hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha, nodes_old = ctx.saved_tensors
        frontier, graph, level, d, attn_dim = ctx.misc
        g_h, g_rela, g_as, g_ar, g_aq, g_w, g_b = engine.layer_bwd(
            frontier, graph, level, nodes_old, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim, grad_agg)
        return g_h, g_rela, g_as, g_ar, g_aq, g_w.view_as(w_alpha), g_b.view_as(b_alpha), None, None, None, None, None, None, None


class GNNLayer(nn.Module):
    """Parameters exactly as models.py:14-21."""

    def __init__(self, in_dim, out_dim, attn_dim, n_rel, act=lambda x: x):
        super().__init__()
        self.n_rel, self.in_dim, self.out_dim, self.attn_dim, self.act = n_rel, in_dim, out_dim, attn_dim, act
        self.rela_embed = nn.Embedding(2 * n_rel + 1, in_dim)
        self.Ws_attn = nn.Linear(in_dim, attn_dim, bias=False)
        self.Wr_attn = nn.Linear(in_dim, attn_dim, bias=False)
        self.Wqr_attn = nn.Linear(in_dim, attn_dim)
        self.w_alpha = nn.Linear(attn_dim, 1)
        self.W_h = nn.Linear(in_dim, out_dim, bias=False)

    def aggregate_nograd(self, q_rel, hidden_p, a_s, frontier, graph, level, nodes_new):
        """Inference form of ``aggregate``: hidden_p [n_old, ld] already padded, a_s [n_old, ap] given
        (produced by the previous layer's fused dense kernel).  Returns agg [n_new, ld]."""
        d, a = self.in_dim, self.attn_dim
        ld, ap = hidden_p.shape[1], a_s.shape[1]
        rela = self.rela_embed.weight
        pad_rows = lambda w: F.pad(w, (0, 0, 0, ap - a)) if ap != a else w
        a_r = F.linear(rela, pad_rows(self.Wr_attn.weight))
        a_q = F.linear(rela[q_rel], pad_rows(self.Wqr_attn.weight), F.pad(self.Wqr_attn.bias, (0, ap - a)))
        if ld != d:
            rela = F.pad(rela, (0, ld - d))
        return engine.layer_fwd(frontier, graph, level, nodes_new, hidden_p, rela.contiguous(), d, a_s, a_r.contiguous(),
                                a_q.contiguous(), self.w_alpha.weight.reshape(-1).contiguous(), self.w_alpha.bias, a)

    def aggregate(self, q_rel, hidden, frontier, graph, level, nodes_new, nodes_old):
        """models.py:29-39 on the device; returns message_agg [n_new, in_dim]."""
        d, a = self.in_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), _pad4(a)
        rela = self.rela_embed.weight
        pad_rows = lambda w: F.pad(w, (0, 0, 0, ap - a)) if ap != a else w
        a_s = tall_linear(hidden, pad_rows(self.Ws_attn.weight))                                # [n_old, ap]
        a_r = F.linear(rela, pad_rows(self.Wr_attn.weight))                                     # [2R+1, ap]
        a_q = F.linear(rela[q_rel], pad_rows(self.Wqr_attn.weight), F.pad(self.Wqr_attn.bias, (0, ap - a)))  # [B, ap]
        if ld != d:
            hidden, rela = F.pad(hidden, (0, ld - d)), F.pad(rela, (0, ld - d))
        agg = _Aggregate.apply(hidden, rela, a_s, a_r, a_q, self.w_alpha.weight.reshape(-1), self.w_alpha.bias,
                               frontier, graph, level, nodes_new, nodes_old, d, a)
        return agg[:, :d] if ld != d else agg

    def forward(self, q_sub, q_rel, hidden, frontier, graph, level, nodes_new, nodes_old):
        agg = self.aggregate(q_rel, hidden, frontier, graph, level, nodes_new, nodes_old)
        return self.act(tall_linear(agg, self.W_h.weight))                       # models.py:41


class RED_GNN_trans(nn.Module):
    def __init__(self, params, loader):
        super().__init__()
        self.n_layer, self.hidden_dim, self.attn_dim, self.n_rel = params.n_layer, params.hidden_dim, params.attn_dim, params.n_rel
        self.loader = loader
        acts = {"relu": nn.ReLU(), "tanh": torch.tanh, "idd": lambda x: x}
        act = acts[params.act]
        self.act_name = params.act
        self.gnn_layers = nn.ModuleList(
            [GNNLayer(self.hidden_dim, self.hidden_dim, self.attn_dim, self.n_rel, act=act) for _ in range(self.n_layer)])
        self.dropout = nn.Dropout(params.dropout)
        self.W_final = nn.Linear(self.hidden_dim, 1, bias=False)
        self.gate = nn.GRU(self.hidden_dim, self.hidden_dim)     # parameters only; the single step runs as gru_cell
        self._frontiers = {}
        self.last_stats = None
        self.fused_dense = True      # inference: W_h + GRU + projections + readout in one MFMA kernel (rg_dense_fwd)

    def _frontier(self, n_ent, batch, n_levels, device):
        key = (n_ent, batch, n_levels, str(device))
        fr = self._frontiers.get(key)
        if fr is None:
            if len(self._frontiers) > 8:
                self._frontiers.clear()
            fr = self._frontiers[key] = engine.Frontier(n_ent, batch, n_levels, device)
        return fr

    def forward(self, subs, rels, mode="train", trace=None):
        device = self.W_final.weight.device
        engine._require_gpu(device)
        n = len(subs)
        graph = self.loader.graph_for(mode)
        q_sub = torch.as_tensor(np.asarray(subs), dtype=torch.int32).to(device)
        q_rel = torch.as_tensor(np.asarray(rels), dtype=torch.int64).to(device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        n_ent = graph.n_ent                     # the inductive setting switches graphs (and n_ent) with the mode
        fr = self._frontier(n_ent, n, self.n_layer + 1 if need_grad else 2, device)
        fr.reset(q_sub)
        if not need_grad and self.fused_dense and engine.dense_supported(self.hidden_dim, self.attn_dim):
            return self._forward_inference(fr, graph, q_sub, q_rel, n, device, trace)

        d = self.hidden_dim
        h0 = torch.zeros((n, d), device=device)                                  # models.py:72
        hidden = torch.zeros((n, d), device=device)                              # models.py:74
        nodes_old = torch.stack([torch.arange(n, device=device, dtype=torch.int32), q_sub], 1)   # models.py:73
        g = self.gate
        n_edges = []
        for i in range(self.n_layer):                                            # models.py:77
            n_new, n_e, _ = fr.expand(graph)                                     # models.py:78 (on the device)
            nodes, _, old_new = fr.nodes(want_prev=False)
            n_edges.append(n_e)
            engine.prefer_blas(n_new)
            hidden = self.gnn_layers[i](q_sub, q_rel, hidden, fr, graph, fr.level, nodes, nodes_old)   # models.py:80
            h0 = torch.zeros((n_new, d), device=device).index_copy(0, old_new.long(), h0)           # models.py:81
            hidden = self.dropout(hidden)                                        # models.py:82
            hidden = gru_step(hidden, h0, g)                                     # models.py:83
            h0 = hidden
            if trace is not None:
                trace.append(dict(nodes=nodes, old_nodes_new_idx=old_new, n_edges=n_e, hidden=hidden))
            nodes_old = nodes
        scores = tall_linear(hidden, self.W_final.weight).squeeze(-1)            # models.py:86
        key = nodes_old[:, 0].long() * n_ent + nodes_old[:, 1].long()
        scores_all = torch.zeros(n * n_ent, device=device).index_copy(0, key, scores)    # models.py:87-88
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes_old.shape[0]))
        return scores_all.view(n, n_ent)

    def _forward_inference(self, fr, graph, q_sub, q_rel, n, device, trace):
        """The same forward with no autograd graph: per layer one expansion, one fused message-passing
        kernel and one fused dense kernel; hidden / a_s never leave their padded device layout."""
        d, a = self.hidden_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), _pad4(a)
        n_ent = graph.n_ent
        hidden = torch.zeros((n, ld), device=device)
        a_s = torch.zeros((n, ap), device=device)                     # hidden == 0 at layer 0 (models.py:74)
        scores_all = torch.zeros(n * n_ent, device=device)           # models.py:87
        n_edges = []
        nodes = None
        for i in range(self.n_layer):
            n_new, n_e, _ = fr.expand(graph)
            nodes, prev_idx, old_new = fr.nodes(want_prev=True, want_old_new=trace is not None)
            n_edges.append(n_e)
            layer = self.gnn_layers[i]
            agg = layer.aggregate_nograd(q_rel, hidden, a_s, fr, graph, fr.level, nodes)
            last = i + 1 == self.n_layer
            hidden, a_s = engine.dense_fwd(
                agg, hidden, prev_idx, d, layer.W_h.weight, self.act_name, self.gate,
                Ws_next=None if last else self.gnn_layers[i + 1].Ws_attn.weight, attn_dim=a, ap=ap,
                W_final=self.W_final.weight if last else None, nodes=nodes, n_ent=n_ent, scores_all=scores_all)
            if trace is not None:
                trace.append(dict(nodes=nodes, old_nodes_new_idx=old_new, n_edges=n_e, hidden=hidden[:, :d]))
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes.shape[0]))
        return scores_all.view(n, n_ent)


class RED_GNN_induc(RED_GNN_trans):
    """Static/inductive/models.py:45-89: the same network; ``mode`` ('transductive' | 'inductive') selects the
    graph and with it the entity count of the score matrix.  Use with ``red_gnn_amd.inductive.DataLoader``."""

    def forward(self, subs, rels, mode="transductive", trace=None):
        return super().forward(subs, rels, mode=mode, trace=trace)
