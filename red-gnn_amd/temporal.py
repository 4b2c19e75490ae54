"""T_RED_GNN (temporal interpolation) with the reference's parameter names, on the HIP path (inference).

Mirrors Temporal/interpolation/model_cuda.py:21-213: per-layer relation tables ``rela_embed_layer.{i}``,
``attention_1_layer.{i}`` (Linear(3d -> a), no bias), ``attention_2_layer.{i}`` (Linear(a -> 1), no bias),
``past_linear / now_linear / future_linear`` (d x d, no bias), ``time_embed`` (n_time x d), ``linear_classifier``
(d -> 1 with bias); ``score_embed_layer`` and ``query_relation_linear`` exist in the reference's state dict but are
unused by its forward (model_cuda.py:210) and are kept for checkpoint compatibility.  ``shared_tables=True`` gives the
parameter layout of Temporal/interpolation/model.py (one rela_embed / attention_1 / attention_2 for all layers).

What is different inside: the per-call scipy coo build, the dense [B, n_ent] index maps and the python
attention_vis loop with .item() syncs (model_cuda.py:121-135,163-166,178-184) do not exist; frontier expansion is the
device bitmap walk, and the per-edge work of a layer is one fused kernel (rg_tlayer_fwd) with the three direction linears
hoisted per node / relation / |dt| (W(h + r + tau) = Wh + Wr + Wtau).
Training: ``mode='train'`` drops the batch's own quadruples (``batch['example_idx']`` rows of ``params.graph``,
model_cuda.py:103-104) by building a device graph for the batch, applies ``nn.Dropout(params.dropout)`` before the
activation (:196) and is differentiable: the per-edge work of the backward pass is rg_tlayer_bwd, the hoisted linears are
ordinary autograd GEMMs.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine
from .models import pad_attn, tall_linear


def _pad4(n):
    return (n + 3) // 4 * 4


class _TAggregate(torch.autograd.Function):
    """agg = rg_tlayer_fwd(...);  backward = rg_tlayer_bwd(...)."""

    @staticmethod
    def forward(ctx, hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha, b_alpha, lease, graph, level, n_new, q_time,
                d, attn_dim):
        hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha = (t.contiguous() for t in (hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha))
        agg = engine.tlayer_fwd(lease.frontier, graph, level, n_new, q_time, hidden_dir, rela_dir, time_dir, d, a_s, a_r, a_q,
                                w_alpha, b_alpha, attn_dim)
        ctx.save_for_backward(hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha, b_alpha, q_time)
        ctx.misc = (lease, graph, level, d, attn_dim)      # the lease keeps the frontier's level bitmaps for this graph's backward
        return agg

    @staticmethod
    def backward(ctx, grad_agg):
        hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha, b_alpha, q_time = ctx.saved_tensors
        lease, graph, level, d, attn_dim = ctx.misc
        lease.check()
        g_hd, g_rd, g_td, g_as, g_ar, g_aq, g_w = engine.tlayer_bwd(lease.frontier, graph, level, a_s.shape[0], q_time, hidden_dir, rela_dir,
                                                              time_dir, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim, grad_agg)
        if level == 1:
            lease.release()
        return (g_hd, g_rd, g_td, g_as, g_ar, g_aq, g_w.view_as(w_alpha)) + (None,) * 8


class T_RED_GNN(nn.Module):
    def __init__(self, params, shared_tables=False):
        super().__init__()
        self.n_rel, self.n_ent, self.n_time = params.n_rel, params.n_ent, params.n_time      # n_rel: relation ids in the graph (incl. idd)
        self.hidden_dim, self.attn_dim, self.n_layer = params.hidden_dim, params.attn_dim, params.n_layer
        self.shared_tables = shared_tables
        d, a = self.hidden_dim, self.attn_dim
        if shared_tables:       # Temporal/interpolation/model.py:19-21
            self.rela_embed = nn.Embedding(self.n_rel + 1, d)
            self.attention_1 = nn.Linear(3 * d, a, bias=False)
            self.attention_2 = nn.Linear(a, 1, bias=False)
        else:                   # Temporal/interpolation/model_cuda.py:32-35
            self.rela_embed_layer = nn.ModuleList([nn.Embedding(self.n_rel + 1, d) for _ in range(self.n_layer)])
            self.score_embed_layer = nn.Embedding(self.n_rel + 1, d)
            self.attention_1_layer = nn.ModuleList([nn.Linear(3 * d, a, bias=False) for _ in range(self.n_layer)])
            self.attention_2_layer = nn.ModuleList([nn.Linear(a, 1, bias=False) for _ in range(self.n_layer)])
            self.query_relation_linear = nn.Linear(d, 1, bias=False)
        self.linear_classifier = nn.Linear(d, 1)
        self.past_linear = nn.Linear(d, d, bias=False)
        self.now_linear = nn.Linear(d, d, bias=False)
        self.future_linear = nn.Linear(d, d, bias=False)
        self.time_embed = nn.Embedding(self.n_time, d)
        acts = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": torch.relu, "idd": lambda x: x,
                "softplus": F.softplus, "leaky_relu": F.leaky_relu}
        self.act = acts[params.act]
        self.dropout = nn.Dropout(getattr(params, "dropout", 0.0))               # model_cuda.py:58
        self.quads = np.ascontiguousarray(np.asarray(params.graph, dtype=np.int32).reshape(-1, 4))
        self.graph = engine.TemporalGraph(self.n_ent, self.n_rel + 1, self.n_time, self.quads,
                                          device=getattr(params, "device", "cuda"))
        self._frontiers = engine.FrontierPool()
        self.last_stats = None

    def _tables(self, i):
        if self.shared_tables:
            return self.rela_embed.weight, self.attention_1.weight, self.attention_2.weight
        return self.rela_embed_layer[i].weight, self.attention_1_layer[i].weight, self.attention_2_layer[i].weight

    def _frontier(self, n, n_levels, device):
        return self._frontiers.get(self.n_ent, n, n_levels, device)

    def forward(self, batch, mode="train"):
        device = self.linear_classifier.weight.device
        engine._require_gpu(device)
        heads = torch.as_tensor(batch["head"]).to(device=device, dtype=torch.int32)
        q_rel = torch.as_tensor(batch["relation"]).to(device=device, dtype=torch.int64)
        q_time = torch.as_tensor(batch["time"]).to(device=device, dtype=torch.int32)
        n = heads.numel()
        graph = self.graph
        if mode == "train":     # model_cuda.py:103-104: the batch's own facts leave the graph
            drop = np.asarray(torch.as_tensor(batch["example_idx"]).cpu()).reshape(-1)
            graph = engine.TemporalGraph(self.n_ent, self.n_rel + 1, self.n_time, self.quads, device=device, exclude=drop)
        with_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        fr = self._frontier(n, self.n_layer + 1 if with_grad else 2, device)
        fr.reset(heads)
        lease = engine.FrontierLease(fr) if with_grad else None
        d, a = self.hidden_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), pad_attn(a)
        w_dir = torch.cat([self.past_linear.weight, self.now_linear.weight, self.future_linear.weight], 0)     # [3d, d]
        padc = lambda t: F.pad(t, (0, ld - d)) if ld != d else t
        pad_rows = lambda w: F.pad(w, (0, 0, 0, ap - a)) if ap != a else w
        time_dir = padc(F.linear(self.time_embed.weight, w_dir).view(self.n_time, 3, d).transpose(0, 1).reshape(3 * self.n_time, d)).contiguous()
        hidden = torch.zeros((n, d), device=device)
        zero_b = torch.zeros(1, device=device)
        n_edges, nodes = [], None
        for i in range(self.n_layer):
            rela, w1, w2 = self._tables(i)
            n_new, n_e, n_old = fr.expand(graph)
            n_edges.append(n_e)
            if with_grad:
                engine.prefer_blas(n_new)
            a_s = tall_linear(hidden, pad_rows(w1[:, :d])).contiguous()               # [n_old, ap]
            a_r = F.linear(rela, pad_rows(w1[:, d:2 * d])).contiguous()              # [n_rel+1, ap]
            a_q = F.linear(rela[q_rel], pad_rows(w1[:, 2 * d:])).contiguous()        # [B, ap]
            hidden_dir = padc(tall_linear(hidden, w_dir).view(n_old * 3, d)).contiguous()        # row 3 s + dir
            rela_dir = padc(F.linear(rela, w_dir).view(-1, 3, d).transpose(0, 1).reshape(-1, d)).contiguous()   # row dir*(R+1) + r
            w_alpha = w2.reshape(-1).contiguous()
            if with_grad:
                agg = _TAggregate.apply(hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha, zero_b, lease, graph, fr.level, n_new,
                                        q_time, d, a)
            else:
                with torch.no_grad():
                    agg = engine.tlayer_fwd(fr, graph, fr.level, n_new, q_time, hidden_dir, rela_dir, time_dir, d, a_s, a_r, a_q,
                                            w_alpha, zero_b, a)
            hidden = self.act(self.dropout(agg[:, :d]))                               # model_cuda.py:196
        nodes, _, _ = fr.nodes(want_prev=False, want_old_new=False)
        result = tall_linear(hidden, self.linear_classifier.weight, self.linear_classifier.bias).reshape(-1)   # model_cuda.py:210
        key_idx = nodes[:, 0].long() * self.n_ent + nodes[:, 1].long()
        score_all = torch.zeros(n * self.n_ent, device=device).index_copy(0, key_idx, result)
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes.shape[0]))
        self.last_nodes = nodes
        return score_all.view(n, self.n_ent)
