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
    in one matrix-core kernel per layer (rg_dense_fwd: fp32 arithmetic as exact three-term f16 splits by default, see dense_precision), and from the third call of a (graph, batch size) on replays the whole
    forward as one captured HIP graph (_GraphedInference); training runs W_h + act + carry + dropout + GRU step as one kernel
    too (_DenseStep: rg_dense_train_fwd / rg_dense_train_bwd), with the weight gradients - sums over millions of node rows -
    issued as row-chunked batched GEMMs (``tall_linear`` / ``_gram_tn``: a plain [m,n] = G^T X product lands on a handful of
    workgroups).
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import ctypes

from . import _lib as _lib_mod
from . import engine


def _pad4(n):
    return (n + 3) // 4 * 4


def pad_attn(a):
    """Padded attention width the layer kernels are instantiated for: multiples of 4 up to 16, then 32 (attn_dim 17..32, e.g. the
    temporal presets' 30).  The reference accepts any attn_dim (models.py:16-19); wider than 32 has no kernel here."""
    if a > 32:
        raise ValueError("attn_dim=%d: the HIP layer kernels cover attention widths up to 32" % a)
    return _pad4(a) if a <= 16 else 32


_FUSED_BWD_ROWS = 1 << 13   # below this the fused dense adjoint's prologue (three transposed weight images per workgroup) costs
                            # more than aten's GRU-cell backward and three small GEMMs
_TALL_ROWS = 1 << 15        # below this a plain GEMM is as good
_TALL_CHUNKS = 256          # ~ one row chunk per CU


def _row_chunks(t, n_chunks, c):
    """[n_chunks, c, cols] view of the first n_chunks * c rows of a 2-D tensor whose rows are unit-stride (rows may be spaced)."""
    return t.as_strided((n_chunks, c, t.shape[1]), (c * t.stride(0), t.stride(0), 1), t.storage_offset())


def _gram_tn(g, x):
    """g^T x for g [N,m], x [N,n] with N >> m,n: chunks of rows as one batched GEMM, then a sum over the chunks
    (fills the chip and shortens every fp32 accumulation chain by the chunk count).  Rows need unit-stride columns only
    (a column block of a wider matrix works without a copy)."""
    if g.stride(1) != 1:
        g = g.contiguous()
    if x.stride(1) != 1:
        x = x.contiguous()
    n_rows = g.shape[0]
    if n_rows < _TALL_ROWS:
        return g.t() @ x
    if g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32:
        return engine.gram_tn(g, x)           # one HIP kernel: rows read once, exact fp32 MFMA products, deterministic sums
    c = n_rows // _TALL_CHUNKS
    body = _TALL_CHUNKS * c
    out = torch.bmm(_row_chunks(g, _TALL_CHUNKS, c).transpose(1, 2), _row_chunks(x, _TALL_CHUNKS, c)).sum(0)
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
        gw = _gram_tn(g, x) if ctx.needs_input_grad[1] else None
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
    def forward(ctx, hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha, lease, graph, level, nodes_new, nodes_old, d, attn_dim):
        hidden, rela, a_s, a_r, a_q = (t.contiguous() for t in (hidden, rela, a_s, a_r, a_q))
        w_alpha, b_alpha = w_alpha.contiguous(), b_alpha.contiguous()
        frontier = lease.frontier
        agg = engine.layer_fwd(frontier, graph, level, nodes_new, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim)
        ctx.save_for_backward(hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha, nodes_old)
        ctx.misc = (lease, graph, level, d, attn_dim)      # the lease keeps the frontier's level bitmaps for this graph's backward
        return agg

    @staticmethod
    def backward(ctx, grad_agg):
        hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha, nodes_old = ctx.saved_tensors
        lease, graph, level, d, attn_dim = ctx.misc
        lease.check()
        g_h, g_rela, g_as, g_ar, g_aq, g_w, g_b = engine.layer_bwd(
            lease.frontier, graph, level, nodes_old, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim, grad_agg)
        if level == 1:
            lease.release()
        return g_h, g_rela, g_as, g_ar, g_aq, g_w.view_as(w_alpha), g_b.view_as(b_alpha), None, None, None, None, None, None, None


class _DenseStep(torch.autograd.Function):
    """hidden_new = GRU(dropout(act(agg W_h^T)), h0 carried from the previous frontier)  -  models.py:41,81-83 of one layer.

    Forward: one fused f32-MFMA kernel (rg_dense_train_fwd) that also leaves the GRU input and the gate workspace in the layout of
    PyTorch's fused GRU cell.  Backward: one fused kernel for everything per node row (rg_dense_train_bwd: gate, input, carry and
    agg gradients) from _FUSED_BWD_ROWS rows up - below that, and at d = 128, aten's fused GRU-cell backward and three GEMMs -
    plus row-chunked batched GEMMs for the five weight gradients (_gram_tn); the carry's gradient goes back by a gather (every
    old node is exactly one new node)."""

    @staticmethod
    def forward(ctx, agg, hidden_prev, W_h, w_ih, w_hh, b_ih, b_hh, Ws_next, prev_idx, old_new, mask, act, gate, keep):
        """Ws_next: the NEXT layer's Ws_attn.weight [attn, d] or None.  With it the kernel also emits a_s = hidden_new Ws_next^T
        (the hoisted attention projection, as the inference kernel does) and the step returns (hidden_new, a_s [n, ap])."""
        agg, hidden_prev = agg.contiguous(), hidden_prev.contiguous()
        hidden, x, ws, a_s = engine.dense_train_fwd(agg, hidden_prev, prev_idx, W_h, act, gate, mask, Ws_next)
        none = agg.new_zeros(0)
        ctx.save_for_backward(agg, x, ws, W_h, w_ih, w_hh, old_new, mask if mask is not None else none,
                              Ws_next if Ws_next is not None else none, hidden if Ws_next is not None else none)
        ctx.act, ctx.keep, ctx.n_old, ctx.has_as = act, keep, hidden_prev.shape[0], Ws_next is not None
        ctx.prev_idx = prev_idx
        return (hidden, a_s) if Ws_next is not None else hidden

    @staticmethod
    def backward(ctx, g_h, g_as=None):
        agg, x, ws, W_h, w_ih, w_hh, old_new, mask, Ws_next, hidden = ctx.saved_tensors
        n, d = agg.shape
        has_mask = mask.numel() > 0
        g_ws = None
        if ctx.has_as and g_as is not None:
            # a_s = hidden Ws_next^T: its gradient joins the new state's (one HIP pass), Ws_next's is a sum over the rows
            a = Ws_next.shape[0]
            g_as = g_as.contiguous()
            g_ws = _gram_tn(g_as[:, :a], hidden)
            g_h = engine.rows_addmm(g_h.contiguous(), g_as, Ws_next) if n >= _TALL_ROWS else g_h + g_as[:, :a] @ Ws_next
        if engine.dense_train_bwd_supported(d) and n >= _TALL_ROWS and ctx.prev_idx is not None and ctx.n_old > 0:
            # one fused kernel for everything per node row (rg_dense_train_bwd2): the hidden-side gate gradients as their n block only
            # (the r and z blocks are dgi's) and the carried state's gradient straight into the previous frontier's rows
            dgi, dgh_n, dpre, g_agg, g_prev = engine.dense_train_bwd2(g_h, ws, x, mask if has_mask else None, ctx.keep, ctx.act, W_h, w_ih,
                                                                      w_hh, ctx.prev_idx, ctx.n_old)
            h0 = ws.view(n, 5, d)[:, 3]                          # a column block of the workspace: no copy
            g_wih, dbi = engine.gram_tn(dgi, x, colsum=True)
            (g_rz, db_rz), (g_n, db_n) = engine.gram_tn(dgi[:, :2 * d], h0, colsum=True), engine.gram_tn(dgh_n, h0, colsum=True)
            g_wh = engine.gram_tn(dpre, agg)
            return g_agg, g_prev, g_wh, g_wih, torch.cat([g_rz, g_n]), dbi, torch.cat([db_rz, db_n]), g_ws, None, None, None, None, None, None
        if engine.dense_train_bwd_supported(d) and n >= _FUSED_BWD_ROWS:
            # one fused kernel for everything per node row (rg_dense_train_bwd); the weight gradients below are sums over rows
            dgi, dgh, dpre, g_agg, dh0 = engine.dense_train_bwd(g_h, ws, x, mask if has_mask else None, ctx.keep, ctx.act, W_h, w_ih, w_hh)
            h0 = ws.view(n, 5, d)[:, 3]                          # a column block of the workspace: no copy
            if n >= _TALL_ROWS:                                  # weight and bias gradients in one pass over the rows each (rg_gram_tn)
                (g_wih, dbi), (g_whh, dbh) = engine.gram_tn(dgi, x, colsum=True), engine.gram_tn(dgh, h0, colsum=True)
                g_wh = engine.gram_tn(dpre, agg)
                g_prev = dh0[old_new.long()] if ctx.n_old else dh0.new_zeros((0, d))
                return g_agg, g_prev, g_wh, g_wih, g_whh, dbi, dbh, g_ws, None, None, None, None, None, None
            dbi, dbh = dgi.sum(0), dgh.sum(0)
        else:
            dgi, dgh, dh0, dbi, dbh = torch.ops.aten._thnn_fused_gru_cell_backward(g_h.contiguous(), ws, True)
            h0 = ws.view(n, 5, d)[:, 3]
            dx = dgi @ w_ih
            dh0 = dh0 + dgh @ w_hh
            if has_mask:
                dx = dx * mask
            if ctx.act == "relu":
                dpre = dx * (x > 0)
            elif ctx.act == "tanh":
                y = x * ctx.keep if has_mask else x              # x = tanh(pre) / keep where kept; dx is 0 where dropped
                dpre = dx * (1.0 - y * y)
            else:
                dpre = dx
            g_agg = dpre @ W_h
        g_wih, g_whh = _gram_tn(dgi, x), _gram_tn(dgh, h0)
        g_wh = _gram_tn(dpre, agg)
        g_prev = dh0[old_new.long()] if ctx.n_old else dh0.new_zeros((0, d))
        return g_agg, g_prev, g_wh, g_wih, g_whh, dbi, dbh, g_ws, None, None, None, None, None, None


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

    def aggregate_nograd(self, tables, hidden_p, a_s, frontier, graph, level, nodes_new):
        """Inference form of ``aggregate``: hidden_p [n_old, ld] already padded, a_s [n_old, ap] given (produced by the previous
        layer's fused dense kernel), ``tables`` = this layer's (a_r, a_q, rela_p) from RED_GNN_trans.inference_tables.
        Returns agg [n_new, ld]."""
        a_r, a_q, rela_p = tables
        return engine.layer_fwd(frontier, graph, level, nodes_new, hidden_p, rela_p, self.in_dim, a_s, a_r, a_q,
                                self.w_alpha.weight.reshape(-1).contiguous(), self.w_alpha.bias, self.attn_dim)

    def aggregate(self, q_rel, hidden, lease, graph, level, nodes_new, nodes_old, a_s=None):
        """models.py:29-39 on the device; returns message_agg [n_new, in_dim].  ``lease``: engine.FrontierLease of the forward.
        ``a_s`` [n_old, ap]: Ws_attn(hidden) when the previous layer's dense kernel already produced it (_DenseStep with Ws_next)."""
        d, a = self.in_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), pad_attn(a)
        rela = self.rela_embed.weight
        pad_rows = lambda w: F.pad(w, (0, 0, 0, ap - a)) if ap != a else w
        if a_s is None:
            a_s = tall_linear(hidden, pad_rows(self.Ws_attn.weight))                            # [n_old, ap]
        a_r = F.linear(rela, pad_rows(self.Wr_attn.weight))                                     # [2R+1, ap]
        a_q = F.linear(rela[q_rel], pad_rows(self.Wqr_attn.weight), F.pad(self.Wqr_attn.bias, (0, ap - a)))  # [B, ap]
        if ld != d:
            hidden, rela = F.pad(hidden, (0, ld - d)), F.pad(rela, (0, ld - d))
        agg = _Aggregate.apply(hidden, rela, a_s, a_r, a_q, self.w_alpha.weight.reshape(-1), self.w_alpha.bias,
                               lease, graph, level, nodes_new, nodes_old, d, a)
        return agg[:, :d] if ld != d else agg

    def forward(self, q_sub, q_rel, hidden, lease, graph, level, nodes_new, nodes_old):
        agg = self.aggregate(q_rel, hidden, lease, graph, level, nodes_new, nodes_old)
        return self.act(tall_linear(agg, self.W_h.weight))                       # models.py:41


class _GraphedInference:
    """One inference forward of RED_GNN_trans for a fixed (graph, batch size) as a captured HIP graph.

    The eager path reads the new frontier's size back once per hop (to size its buffers) and issues ~15 launches per layer
    from Python; at the reference's evaluation batch sizes (n_tbatch = 50) that host work is the whole step.  Here every buffer
    has the capacity of the full (query, entity) grid, the hop runs without the read-back (rg_frontier_expand_async), the
    kernels take the level's size from device memory (rg_dense_fwd_dev) or do not need it (bitmap walk), and the sequence is
    captured once and replayed: one graph launch per batch, nothing synchronises."""

    MAX_BYTES = 64 << 30      # of the 288 GB: capacity-sized replay buffers of all lanes and batch shapes together may take twice this

    @classmethod
    def budget(cls, device):
        """Bytes one captured forward may take: MAX_BYTES, or a quarter of what is free on the device right now if that is less (a
        smaller or shared device, training state resident beside the evaluator's buffers)."""
        try:
            free, _ = torch.cuda.mem_get_info(device)
        except Exception:
            return cls.MAX_BYTES
        return min(cls.MAX_BYTES, free // 4)

    def __init__(self, model, graph, n, device, hints):
        self.model, self.graph, self.n, self.hints = model, graph, n, hints
        d, a = model.hidden_dim, model.attn_dim
        self.ld, self.ap = max(16, _pad4(d)), pad_attn(a)
        n_ent = graph.n_ent
        cap = self.cap = n * n_ent
        f32 = dict(dtype=torch.float32, device=device)
        self.q_sub = torch.zeros(n, dtype=torch.int32, device=device)
        self.q_rel = torch.zeros(n, dtype=torch.int64, device=device)
        self.fr = engine.Frontier(n_ent, n, 2, device)
        self.nodes = torch.empty((cap, 2), dtype=torch.int32, device=device)
        self.prev = torch.empty(cap, dtype=torch.int32, device=device)
        self.agg = torch.empty((cap, self.ld), **f32)
        self.hid = [torch.empty((cap, self.ld), **f32) for _ in range(2)]
        self.a_s = [torch.empty((cap, self.ap), **f32) for _ in range(2)]
        self.scores = torch.empty(cap, **f32)
        nbytes = engine.layer_fwd_scratch_bytes(self.fr, graph, self.ld)
        self.scratch = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        self.dense_scratch = engine.dense_scratch(d, model.dense_precision, device)      # (static: the graph replays its address)
        self.key_ptr = model.W_final.weight.data_ptr()
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():       # a warm-up pass on the capture conditions, then the capture itself
            self._enqueue()
        torch.cuda.current_stream(device).wait_stream(side)
        self.cuda_graph = None
        if os.environ.get("RG_NO_CAPTURE") != "1":           # debugging aid: enqueue the same sequence eagerly on every call
            self.cuda_graph = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(self.cuda_graph), torch.no_grad():
                self._enqueue()
            # every fill inside the forward is a kernel (common.h zero_async): a memset node here would mean a torch op started to
            # emit one, and replays of graphs with several memset nodes zeroed correctly on their first launch only (ROCm 7.2)
            n_fill = _lib_mod.lib().rg_hipgraph_fill_nodes(ctypes.c_void_p(self.cuda_graph.raw_cuda_graph()))
            if n_fill != 0:
                raise RuntimeError("captured forward holds %d memset nodes (expected none)" % n_fill)
            self.cuda_graph.instantiate()

    @staticmethod
    def bytes_needed(n, n_ent, ld, ap):
        return n * n_ent * (4 * (3 * ld + 2 * ap + 1) + 12)

    def _enqueue(self):
        m, fr, graph, n = self.model, self.fr, self.graph, self.n
        d, a, ld, ap = m.hidden_dim, m.attn_dim, self.ld, self.ap
        fr.reset(self.q_sub)
        hidden, a_s = self.hid[1], self.a_s[1]
        # hidden == 0 at layer 0 (models.py:74), scores_all = 0 (models.py:87): one multi-tensor launch
        torch._foreach_zero_([hidden[:n], a_s[:n], self.scores])
        tables = m.inference_tables(self.q_rel, ld, ap)
        for i in range(m.n_layer):
            layer = m.gnn_layers[i]
            n_hint, walk, e_hint = (tuple(self.hints[i]) + (None,))[:3]
            fr.expand_nodes_async(graph, self.nodes, self.prev, edge_hint=e_hint)
            a_r, a_q, rela_p = tables[i]
            engine.layer_fwd_into(fr, graph, fr.level, n_hint, hidden, rela_p, d, a_s, a_r, a_q,
                                  layer.w_alpha.weight.reshape(-1).contiguous(), layer.w_alpha.bias, a, self.agg, self.scratch, walk=walk)
            last = i + 1 == m.n_layer
            out_h, out_a = self.hid[i % 2], self.a_s[i % 2]
            engine.dense_fwd_dev(self.cap, fr.count_ptr(), self.agg, hidden, self.prev, d, layer.W_h.weight, m.act_name, m.gate, out_h,
                                 Ws_next=None if last else m.gnn_layers[i + 1].Ws_attn.weight, attn_dim=a, ap=ap,
                                 a_s_out=None if last else out_a, W_final=m.W_final.weight if last else None,
                                 nodes=self.nodes, n_ent=graph.n_ent, scores_all=self.scores, n_hint=n_hint, precision=m.dense_precision,
                                 scratch=self.dense_scratch)
            hidden, a_s = out_h, out_a

    def run(self, q_sub, q_rel):
        self.q_sub.copy_(q_sub, non_blocking=True)
        self.q_rel.copy_(q_rel, non_blocking=True)
        if self.cuda_graph is not None:
            self.cuda_graph.replay()
        else:
            self._enqueue()
        return self.scores.view(self.n, self.graph.n_ent).clone()     # the static buffer is overwritten by the next replay

    def stats(self):
        counts = self.fr.level_counts()                       # one read-back, only when somebody asks
        return dict(n_edges=[e for (_, e) in counts[1:]], n_nodes=counts[-1][0])


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
        self._frontiers = engine.FrontierPool()
        self._last_stats = None
        self.fused_dense = True      # inference: W_h + GRU + projections + readout in one MFMA kernel (rg_dense_fwd)
        # matrix products of that kernel.  "f16x3" (default) = fp32 arithmetic on the f16 matrix pipe: every fp32 operand carried
        # exactly as a three-term f16 split, six partial products per product, fp32 accumulation (csrc/split3.h; d <= 64, at d = 128
        # it runs the f32 MFMA kernel); "f32" = v_mfma_f32_16x16x4_f32; "f16x2" = two-term splits (22-bit operands: narrower than the
        # reference's fp32 nn.Linear / nn.GRU, opt-in only)
        self.dense_precision = "f16x3"
        self.use_graphs = True       # inference: replay a captured HIP graph per (graph, batch size) from the third call on
        self._graphed, self._seen, self._hints, self._pending_key, self._graph_failed = {}, {}, {}, None, set()

    @property
    def last_stats(self):
        """dict(n_edges=[E per layer], n_nodes=N of the last layer) of the latest forward (read back lazily after a graph replay)."""
        if callable(self._last_stats):
            self._last_stats = self._last_stats()
        return self._last_stats

    @last_stats.setter
    def last_stats(self, value):
        self._last_stats = value

    def _frontier(self, n_ent, batch, n_levels, device):
        """A Frontier of this shape that no live autograd graph still needs (engine.FrontierPool)."""
        return self._frontiers.get(n_ent, batch, n_levels, device)

    def forward(self, subs, rels, mode="train", trace=None):
        device = self.W_final.weight.device
        engine._require_gpu(device)
        n = len(subs)
        graph = self.loader.graph_for(mode)
        if torch.is_tensor(subs):        # device-resident batches (DataLoader.get_batch_csr) skip the host round trip
            q_sub, q_rel = subs.to(device=device, dtype=torch.int32), torch.as_tensor(rels).to(device=device, dtype=torch.int64)
        else:
            subs_h, rels_h = np.asarray(subs), np.asarray(rels)
            if n and (subs_h.min() < 0 or subs_h.max() >= graph.n_ent or rels_h.min() < 0 or rels_h.max() > 2 * self.n_rel):
                raise ValueError("query subject / relation id out of range (n_ent=%d, 2*n_rel+1=%d)" % (graph.n_ent, 2 * self.n_rel + 1))
            q_sub = torch.as_tensor(subs_h, dtype=torch.int32).to(device)
            q_rel = torch.as_tensor(rels_h, dtype=torch.int64).to(device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        n_ent = graph.n_ent                     # the inductive setting switches graphs (and n_ent) with the mode
        # the fused kernels implement dropout as the identity: they serve eval mode (and training mode with p = 0) only
        no_dropout = not self.training or self.dropout.p == 0.0
        fused = not need_grad and no_dropout and self.fused_dense and engine.dense_supported(self.hidden_dim, self.attn_dim)
        if fused and self.use_graphs and trace is None and engine.KERNEL_EVENTS is None and engine.DENSE_EVENTS is None:
            out = self._forward_graphed(graph, q_sub, q_rel, n, device)
            if out is not None:
                return out
        fr = self._frontier(n_ent, n, self.n_layer + 1 if need_grad else 2, device)
        fr.reset(q_sub)
        if fused:
            return self._forward_inference(fr, graph, q_sub, q_rel, n, device, trace)
        lease = engine.FrontierLease(fr)      # lives as long as the autograd contexts of this forward

        d = self.hidden_dim
        h0 = torch.zeros((n, d), device=device)                                  # models.py:72
        hidden = torch.zeros((n, d), device=device)                              # models.py:74
        nodes_old = torch.stack([torch.arange(n, device=device, dtype=torch.int32), q_sub], 1)   # models.py:73
        g = self.gate
        n_edges = []
        fused_train = self.fused_dense and engine.dense_train_supported(d, self.act_name)
        a_s_next = None
        for i in range(self.n_layer):                                            # models.py:77
            n_new, n_e, _ = fr.expand(graph)                                     # models.py:78 (on the device)
            nodes, prev_idx, old_new = fr.nodes(want_prev=fused_train)
            n_edges.append(n_e)
            engine.prefer_blas(n_new)
            layer = self.gnn_layers[i]
            if fused_train:
                # models.py:80-83 with the dense part in one kernel: W_h + act, h0 carry (gather by prev_idx), dropout, GRU step
                agg = layer.aggregate(q_rel, hidden, lease, graph, fr.level, nodes, nodes_old, a_s=a_s_next)
                mask = None
                if self.training and self.dropout.p > 0.0:
                    keep = 1.0 - self.dropout.p
                    mask = torch.empty((n_new, d), device=device).bernoulli_(keep).div_(keep)
                # the next layer's attention projection of the new state comes out of the same kernel (attn <= 16) - and after the last
                # layer the readout W_final (models.py:86) as a one-row projection
                if i + 1 < self.n_layer:
                    Ws_next = self.gnn_layers[i + 1].Ws_attn.weight if self.attn_dim <= 16 else None
                else:
                    Ws_next = self.W_final.weight
                out = _DenseStep.apply(agg, h0, layer.W_h.weight, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0, Ws_next,
                                       prev_idx, old_new, mask, self.act_name, g, 1.0 - self.dropout.p)
                hidden, a_s_next = out if Ws_next is not None else (out, None)
            else:
                hidden = layer(q_sub, q_rel, hidden, lease, graph, fr.level, nodes, nodes_old)             # models.py:80
                h0 = torch.zeros((n_new, d), device=device).index_copy(0, old_new.long(), h0)           # models.py:81
                hidden = self.dropout(hidden)                                        # models.py:82
                hidden = gru_step(hidden, h0, g)                                     # models.py:83
            h0 = hidden
            if trace is not None:
                trace.append(dict(nodes=nodes, old_nodes_new_idx=old_new, n_edges=n_e, hidden=hidden))
            nodes_old = nodes
        if fused_train and a_s_next is not None:
            scores = a_s_next[:, 0]                                              # models.py:86, computed by the last dense step
        else:
            scores = tall_linear(hidden, self.W_final.weight).squeeze(-1)        # models.py:86
        key = nodes_old[:, 0].long() * n_ent + nodes_old[:, 1].long()
        scores_all = torch.zeros(n * n_ent, device=device).index_copy(0, key, scores)    # models.py:87-88
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes_old.shape[0]))
        return scores_all.view(n, n_ent)

    def inference_tables(self, q_rel, ld, ap):
        """Per-layer attention tables of an inference forward for ALL layers in one launch (they are tiny, so their launch count -
        not their work - is what a small batch pays for): a_r[i] = Wr_i(rela_i) [2R+1, ap], a_q[i] = Wqr_i(rela_i[q_rel]) + b_i
        [B, ap] (models.py:33,36, the three attention Linear layers hoisted), rela_i padded to ld columns."""
        return engine.attn_tables(list(self.gnn_layers), q_rel.contiguous(), self.hidden_dim, ld, self.attn_dim, ap)

    def _forward_graphed(self, graph, q_sub, q_rel, n, device):
        """Replay (or, on the third call with the same graph and batch size, capture) the forward as a HIP graph.
        Returns None when this call should run eagerly: the first two calls of a shape (the eager run also provides the
        per-hop sizes that pick the kernels' walks), or shapes whose full-grid buffers would be too large."""
        # one captured graph (with its own full-grid buffers) per stream: the evaluator's lanes replay concurrently
        key = (graph.serial, n, str(device), torch.cuda.current_stream(device).cuda_stream, self.dense_precision)
        g = self._graphed.get(key)
        if g is not None and g.key_ptr != self.W_final.weight.data_ptr():        # parameters were re-allocated (.to(), ...)
            g = None
            self._graphed.pop(key)
        if g is None:
            ld, ap = max(16, _pad4(self.hidden_dim)), pad_attn(self.attn_dim)
            seen = self._seen.get(key, 0)
            self._seen[key] = seen + 1
            budget = _GraphedInference.budget(device)
            if seen < 2 or _GraphedInference.bytes_needed(n, graph.n_ent, ld, ap) > budget:
                self._pending_key = key          # the eager run that follows records its per-hop sizes under this key
                return None
            hints = self._hints.get(key) or [(n * graph.n_ent, 1, None)] * self.n_layer      # (node count, walk, edge count) per hop
            if key in self._graph_failed:
                return None
            # a split's evaluation uses a few shapes (the batch size and the last, partial batches): keep them all, within a
            # memory budget for the capacity-sized buffers
            need = _GraphedInference.bytes_needed(n, graph.n_ent, ld, ap)
            held = sum(_GraphedInference.bytes_needed(v.n, v.graph.n_ent, v.ld, v.ap) for v in self._graphed.values())
            if len(self._graphed) >= 80 or held + need > 2 * budget:
                self.release_replay_buffers()
            try:
                g = self._graphed[key] = _GraphedInference(self, graph, n, device, hints)
            except Exception as exc:      # capture refused (memory, a runtime that cannot capture here, ...): keep the eager HIP path
                import warnings
                warnings.warn("RED_GNN_trans: HIP graph capture failed for batch size %d (%s: %s); this shape stays on the eager "
                              "path" % (n, type(exc).__name__, exc))
                self._graph_failed.add(key)
                torch.cuda.synchronize(device)
                return None
        scores = g.run(q_sub, q_rel)
        self._last_stats = g.stats
        return scores

    def replay_bytes_held(self):
        return sum(_GraphedInference.bytes_needed(v.n, v.graph.n_ent, v.ld, v.ap) for v in self._graphed.values())

    def release_replay_buffers(self):
        """Drop every captured forward with its capacity-sized buffers (and the bookkeeping keyed like them).  Other lanes' replays
        may still be in flight on their streams: the device is drained first - destroying an executing graph is not known to be safe."""
        if self._graphed:
            torch.cuda.synchronize()
        self._graphed.clear()
        self._seen.clear()
        self._hints.clear()
        self._graph_failed.clear()
        self._pending_key = None

    def _forward_inference(self, fr, graph, q_sub, q_rel, n, device, trace):
        """The same forward with no autograd graph: per layer one expansion, one fused message-passing
        kernel and one fused dense kernel; hidden / a_s never leave their padded device layout."""
        d, a = self.hidden_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), pad_attn(a)
        n_ent = graph.n_ent
        hidden = torch.zeros((n, ld), device=device)
        a_s = torch.zeros((n, ap), device=device)                     # hidden == 0 at layer 0 (models.py:74)
        scores_all = torch.zeros(n * n_ent, device=device)           # models.py:87
        n_edges, sizes = [], []
        nodes = None
        tables = self.inference_tables(q_rel, ld, ap)
        for i in range(self.n_layer):
            n_new, n_e, n_old = fr.expand(graph)
            nodes, prev_idx, old_new = fr.nodes(want_prev=True, want_old_new=trace is not None)
            n_edges.append(n_e)
            sizes.append((n_new, engine.layer_fwd_plan(fr, graph, fr.level, n_old, n_new, n_e, ld), n_e))
            layer = self.gnn_layers[i]
            agg = layer.aggregate_nograd(tables[i], hidden, a_s, fr, graph, fr.level, nodes)
            last = i + 1 == self.n_layer
            hidden, a_s = engine.dense_fwd(
                agg, hidden, prev_idx, d, layer.W_h.weight, self.act_name, self.gate,
                Ws_next=None if last else self.gnn_layers[i + 1].Ws_attn.weight, attn_dim=a, ap=ap,
                W_final=self.W_final.weight if last else None, nodes=nodes, n_ent=n_ent, scores_all=scores_all,
                precision=self.dense_precision)
            if trace is not None:
                trace.append(dict(nodes=nodes, old_nodes_new_idx=old_new, n_edges=n_e, hidden=hidden[:, :d]))
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes.shape[0]))
        if getattr(self, "_pending_key", None) is not None:
            if len(self._hints) > 16:
                self._hints.clear()
            self._hints[self._pending_key] = sizes
            self._pending_key = None
        return scores_all.view(n, n_ent)


class RED_GNN_induc(RED_GNN_trans):
    """Static/inductive/models.py:45-89: the same network; ``mode`` ('transductive' | 'inductive') selects the
    graph and with it the entity count of the score matrix.  Use with ``red_gnn_amd.inductive.DataLoader``."""

    def forward(self, subs, rels, mode="transductive", trace=None):
        return super().forward(subs, rels, mode=mode, trace=trace)
