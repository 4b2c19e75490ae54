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
Scope (this round): forward in eval mode (`mode != 'train'`, i.e. no per-batch deletion of the query quadruples and
dropout = identity); no backward kernel for the temporal variant yet.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine


def _pad4(n):
    return (n + 3) // 4 * 4


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
        self.graph = engine.TemporalGraph(self.n_ent, self.n_rel + 1, self.n_time, params.graph,
                                          device=getattr(params, "device", "cuda"))
        self._frontiers = {}
        self.last_stats = None

    def _tables(self, i):
        if self.shared_tables:
            return self.rela_embed.weight, self.attention_1.weight, self.attention_2.weight
        return self.rela_embed_layer[i].weight, self.attention_1_layer[i].weight, self.attention_2_layer[i].weight

    @torch.no_grad()
    def forward(self, batch, mode="test"):
        if mode == "train":
            raise NotImplementedError("the temporal variant runs inference only in this round (no per-batch fact deletion, "
                                      "no backward kernel): SURVEY.md §8(f)")
        device = self.linear_classifier.weight.device
        engine._require_gpu(device)
        heads = torch.as_tensor(batch["head"]).to(device=device, dtype=torch.int32)
        q_rel = torch.as_tensor(batch["relation"]).to(device=device, dtype=torch.int64)
        q_time = torch.as_tensor(batch["time"]).to(device=device, dtype=torch.int32)
        n = heads.numel()
        key = (n, str(device))
        fr = self._frontiers.get(key)
        if fr is None:
            fr = self._frontiers[key] = engine.Frontier(self.n_ent, n, 2, device)
        fr.reset(heads)
        d, a = self.hidden_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), _pad4(a)
        w_dir = torch.cat([self.past_linear.weight, self.now_linear.weight, self.future_linear.weight], 0)     # [3d, d]
        padc = lambda t: F.pad(t, (0, ld - d)) if ld != d else t
        time_dir = padc(F.linear(self.time_embed.weight, w_dir).view(self.n_time, 3, d).transpose(0, 1).reshape(3 * self.n_time, d)).contiguous()
        hidden = torch.zeros((n, d), device=device)
        zero_b = torch.zeros(1, device=device)
        n_edges, nodes = [], None
        for i in range(self.n_layer):
            rela, w1, w2 = self._tables(i)
            n_new, n_e, n_old = fr.expand(self.graph)
            n_edges.append(n_e)
            pad_rows = lambda w: F.pad(w, (0, 0, 0, ap - a)) if ap != a else w
            a_s = F.linear(hidden, pad_rows(w1[:, :d])).contiguous()                 # [n_old, ap]
            a_r = F.linear(rela, pad_rows(w1[:, d:2 * d])).contiguous()              # [n_rel+1, ap]
            a_q = F.linear(rela[q_rel], pad_rows(w1[:, 2 * d:])).contiguous()        # [B, ap]
            hidden_dir = padc(F.linear(hidden, w_dir).view(n_old * 3, d)).contiguous()          # row 3 s + dir
            rela_dir = padc(F.linear(rela, w_dir).view(-1, 3, d).transpose(0, 1).reshape(-1, d)).contiguous()   # row dir*(R+1) + r
            agg = engine.tlayer_fwd(fr, self.graph, fr.level, n_new, q_time, hidden_dir, rela_dir, time_dir, d, a_s, a_r, a_q,
                                    w2.reshape(-1).contiguous(), zero_b, a)
            hidden = self.act(agg[:, :d])                                             # model_cuda.py:196 (dropout = identity in eval)
        nodes, _, _ = fr.nodes(want_prev=False, want_old_new=False)
        result = self.linear_classifier(hidden).reshape(-1)                           # model_cuda.py:210
        key_idx = nodes[:, 0].long() * self.n_ent + nodes[:, 1].long()
        score_all = torch.zeros(n * self.n_ent, device=device).index_copy(0, key_idx, result)
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes.shape[0]))
        self.last_nodes = nodes
        return score_all.view(n, self.n_ent)
