"""T_RED_GNN for temporal EXTRAPOLATION (forecasting) on the HIP path - inference and training.

Mirrors Temporal/extrapolation/model_cuda_new_embedding.py:57-265 (the reference's ``T_RED_GNN`` with the periodic time embedding):
parameters ``rela_embed_layer.{i}`` [n_rel + 2, d], ``attention_1_layer.{i}`` (3d -> a, no bias), ``attention_2_layer.{i}`` (a -> 1, no
bias), ``past_linear`` / ``now_linear`` / ``future_linear`` (only past_linear is used by the forward, :205), ``linear_classifier``,
``time_embed`` (+ the two absolute-time embeddings the reference constructs but does not use, :86-87).

What is different inside: the per-query python loop that slices ``self.dataset`` and stacks self-loops (:165-176), the scipy block
adjacency and the dense [B, n_ent] index maps (:178-191,229-235) and the pickled attention statistics (:145-152,216-220,254-259) do not
exist.  The whole data array lives on the device as ONE quadruple graph (self-loops first, then the time-sorted data rows, each edge
carrying its data-row index); a query's window ``dataset[time_offset_list[begin]:time_offset_list[cur_t]]`` is a pair of row bounds,
the frontier hops under those windows (rg_frontier_set_window) and the layer is one fused kernel (rg_xlayer_fwd) with the direction
matrix hoisted by linearity.  The time embedding of the ~120 distinct relative times of a window is a table computed once per forward.

Parity: the reference's model file cannot be imported in the build container (torch_scatter, pyvis, rtdl_revisiting_models are absent),
so this path is checked against the oracle's restatement only - parity UNPINNED - except ``segment_rank_fil``, whose fixture comes
from the reference's importable ``segment.py``.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine
from .models import pad_attn, tall_linear


class _XAggregate(torch.autograd.Function):
    """agg = rg_xlayer_fwd(...);  backward = rg_xlayer_bwd(...).  The frontier keeps the batch's row windows until the backward has run
    (the lease holds it; T_RED_GNN.forward clears the windows only on the inference path)."""

    @staticmethod
    def forward(ctx, hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha, b_alpha, lease, graph, level, n_new, q_time, loop_time, row_time,
                n_data, d, attn_dim):
        hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha = (t.contiguous() for t in (hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha))
        agg = engine.xlayer_fwd(lease.frontier, graph, level, n_new, q_time, loop_time, row_time, n_data, hidden_p, rela_p, time_p, d,
                                a_s, a_r, a_q, w_alpha, b_alpha, attn_dim)
        ctx.save_for_backward(hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha, b_alpha, q_time, loop_time, row_time)
        ctx.misc = (lease, graph, level, n_data, d, attn_dim)
        return agg

    @staticmethod
    def backward(ctx, grad_agg):
        hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha, b_alpha, q_time, loop_time, row_time = ctx.saved_tensors
        lease, graph, level, n_data, d, attn_dim = ctx.misc
        lease.check()
        g = engine.xlayer_bwd(lease.frontier, graph, level, a_s.shape[0], q_time, loop_time, row_time, n_data, hidden_p, rela_p, time_p, d,
                              a_s, a_r, a_q, w_alpha, b_alpha, attn_dim, grad_agg)
        if level == 1:
            lease.release()
        return (g[0], g[1], g[2], g[3], g[4], g[5], g[6].view_as(w_alpha)) + (None,) * 11


WINDOW = 120        # model_cuda_new_embedding.py:168: begin_time = cur_t - 120


def _pad4(n):
    return (n + 3) // 4 * 4


def get_time_offset_list(data, time_granularity=24):
    """Temporal/extrapolation/utils.py:692-699, vectorised: offset_list[t + 1] = index of the LAST row whose time is t (0 where no row
    has that time - the reference's zeros stay)."""
    t = np.asarray(data)[:, 3] // time_granularity
    off = np.zeros(int(t.max()) + 2, dtype=np.int32)
    off[t + 1] = np.arange(len(t), dtype=np.int32)          # later rows overwrite earlier ones, as the reference's loop does
    return off


class PeriodicEmbeddings(nn.Module):
    """The reference's vendored (and edited) rtdl PeriodicEmbeddings for one feature, lite=False (rtdl_num_embeddings.py:69-100,126-215):
    x -> [cos(2 pi w x), sin(2 pi w x)] (n_frequencies each) -> one of two linears by the sign of x -> ReLU.  Same parameter names
    and shapes (``periodic.weight`` [1, k], ``linear_neg/linear_pos.weight`` [1, 2k, d], ``.bias`` [1, d]) and initialisation."""

    def __init__(self, d_embedding, n_frequencies=48, frequency_init_scale=0.01):
        super().__init__()
        self.periodic = nn.Module()
        self.periodic.weight = nn.Parameter(torch.empty(1, n_frequencies))
        nn.init.trunc_normal_(self.periodic.weight, 0.0, frequency_init_scale, a=-3 * frequency_init_scale, b=3 * frequency_init_scale)
        for name in ("linear_neg", "linear_pos"):
            lin = nn.Module()
            lin.weight = nn.Parameter(torch.empty(1, 2 * n_frequencies, d_embedding))
            lin.bias = nn.Parameter(torch.empty(1, d_embedding))
            r = (2 * n_frequencies) ** -0.5
            nn.init.uniform_(lin.weight, -r, r)
            nn.init.uniform_(lin.bias, -r, r)
            setattr(self, name, lin)

    def forward(self, x):
        """x [N, 1] -> [N, 1, d]."""
        is_neg = (x < 0).unsqueeze(-1)
        z = 2 * math.pi * self.periodic.weight * x[..., None]
        z = torch.cat([torch.cos(z), torch.sin(z)], -1)                                    # [N, 1, 2k]
        neg = (z[..., None, :] @ self.linear_neg.weight).squeeze(-2) + self.linear_neg.bias
        pos = (z[..., None, :] @ self.linear_pos.weight).squeeze(-2) + self.linear_pos.bias
        return torch.relu(neg * is_neg + pos * (~is_neg))


class T_RED_GNN(nn.Module):
    """``params``: n_ent, n_rel (true relations incl. reversed ones; the self-loop relation is id n_rel, as Data.num_relations),
    data int [n,4] = (subject, relation, object, time) sorted by time (contents.data), time_granularity, hidden_dim, attn_dim, n_layer,
    act, time_offset_list (optional: computed by get_time_offset_list)."""

    def __init__(self, params):
        super().__init__()
        self.n_rel_true = int(params.n_rel)
        self.n_rel = self.n_rel_true + 1                       # as the reference's self.n_rel (:61)
        self.n_ent, self.hidden_dim, self.attn_dim, self.n_layer = int(params.n_ent), params.hidden_dim, params.attn_dim, params.n_layer
        self.time_granularity = int(params.time_granularity)
        d, a = self.hidden_dim, self.attn_dim
        self.rela_embed_layer = nn.ModuleList([nn.Embedding(self.n_rel + 1, d) for _ in range(self.n_layer)])
        self.attention_1_layer = nn.ModuleList([nn.Linear(3 * d, a, bias=False) for _ in range(self.n_layer)])
        self.attention_2_layer = nn.ModuleList([nn.Linear(a, 1, bias=False) for _ in range(self.n_layer)])
        self.linear_classifier = nn.Linear(d, 1)
        self.past_linear = nn.Linear(d, d, bias=False)
        self.now_linear = nn.Linear(d, d, bias=False)
        self.future_linear = nn.Linear(d, d, bias=False)
        self.time_embed = PeriodicEmbeddings(d)
        self.time_embed_absolute_query = PeriodicEmbeddings(d)      # constructed, unused by the forward (:86-87,201)
        self.time_embed_absolute_graph = PeriodicEmbeddings(d)
        acts = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": torch.relu, "idd": lambda x: x, "softplus": F.softplus,
                "leakyrelu": F.leaky_relu}
        self.act = acts[params.act]
        for i in range(self.n_layer):
            nn.init.xavier_normal_(self.rela_embed_layer[i].weight)   # init_params (:120-122)
        data = np.ascontiguousarray(np.asarray(params.data, dtype=np.int64).reshape(-1, 4))
        self.n_data = len(data)
        off = getattr(params, "time_offset_list", None)
        self.time_offset_list = np.asarray(off if off is not None else get_time_offset_list(data, self.time_granularity), dtype=np.int64)
        device = getattr(params, "device", "cuda")
        # one graph for every window: self-loops first (:172-173 puts them in front of a query's rows), then the data rows; the time
        # field of an edge is its data-row index (self-loops: n_data), which the kernels test against the query's row window
        ent = np.arange(self.n_ent)
        loops = np.stack([ent, np.full(self.n_ent, self.n_rel_true), ent, np.full(self.n_ent, self.n_data)], 1)
        rows = np.concatenate([data[:, :3], np.arange(self.n_data)[:, None]], 1)
        self.graph = engine.TemporalGraph(self.n_ent, self.n_rel + 1, self.n_data + 1, np.concatenate([loops, rows], 0).astype(np.int32),
                                          device=device)
        self._row_time_host = data[:, 3] // self.time_granularity
        self.register_buffer("row_time", torch.as_tensor(self._row_time_host, dtype=torch.int32), persistent=False)
        self._frontiers = engine.FrontierPool()
        self.last_stats = None

    def forward(self, X):
        """X: src_idx, rel_idx, ts (numpy arrays, as the reference's batch object).  Returns (score_all [B, n_ent],
        (per-query softmax over the visited entities [N], visited (batch, entity) pairs int64 numpy [N,2])) as :245-261."""
        device = self.linear_classifier.weight.device
        engine._require_gpu(device)
        src, rel = np.asarray(X.src_idx), np.asarray(X.rel_idx)
        cur_t = np.asarray(X.ts) // self.time_granularity                          # :138
        n = len(src)
        begin = np.maximum(cur_t - WINDOW, 0)                                       # :168-170
        off = self.time_offset_list
        to32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(device)
        win_lo, win_hi = to32(off[begin]), to32(off[cur_t])                         # :171 dataset[offset[begin]:offset[cur_t]]
        q_time, loop_time = to32(cur_t), to32(begin)                                # self-loops carry time begin * granularity (:172)
        q_rel = torch.as_tensor(rel, dtype=torch.int64).to(device)
        with_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        fr = self._frontiers.get(self.n_ent, n, self.n_layer + 1 if with_grad else 2, device)
        fr.set_window(win_lo, win_hi, self.n_data)
        fr.reset(to32(src))
        lease = engine.FrontierLease(fr) if with_grad else None
        d, a = self.hidden_dim, self.attn_dim
        ld, ap = max(16, _pad4(d)), pad_attn(a)
        padc = lambda t: F.pad(t, (0, ld - d)) if ld != d else t
        pad_rows = lambda w: F.pad(w, (0, 0, 0, ap - a)) if ap != a else w
        w_past = self.past_linear.weight
        # largest relative time any edge of the batch can have: a window's oldest row is its first (the data are time-sorted; the
        # reference's offsets put the last row of the day before `begin` in front, and 0 for days without rows)
        lo_h, hi_h = off[begin], off[cur_t]
        oldest = np.where(lo_h < hi_h, self._row_time_host[np.minimum(lo_h, self.n_data - 1)], begin) if self.n_data else begin
        n_tab = int(np.maximum(cur_t - oldest, cur_t - begin).max()) + 1 if n else 1
        deltas = torch.arange(n_tab, dtype=torch.float32, device=device)
        hidden = torch.zeros((n, d), device=device)
        zero_b = torch.zeros(1, device=device)
        n_edges = []
        with torch.set_grad_enabled(with_grad):
            time_p = padc(F.linear(self.time_embed(deltas.view(-1, 1)).squeeze(1), w_past)).contiguous()     # W_past time_embed(delta)  (:201,205)
            for i in range(self.n_layer):
                rela, w1, w2 = self.rela_embed_layer[i].weight, self.attention_1_layer[i].weight, self.attention_2_layer[i].weight
                n_new, n_e, n_old = fr.expand(self.graph)
                n_edges.append(n_e)
                a_s = tall_linear(hidden, pad_rows(w1[:, :d])).contiguous()                       # attention_1 on [h_s | rel | rel_q] (:207-208)
                a_r = F.linear(rela, pad_rows(w1[:, d:2 * d])).contiguous()
                a_q = F.linear(rela[q_rel], pad_rows(w1[:, 2 * d:])).contiguous()
                hidden_p = padc(tall_linear(hidden, w_past)).contiguous()                         # W_past (h + r + tau) = W_past h + ... (:203-205)
                rela_p = padc(F.linear(rela, w_past)).contiguous()
                w_alpha = w2.reshape(-1).contiguous()
                if with_grad:
                    agg = _XAggregate.apply(hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha, zero_b, lease, self.graph, fr.level, n_new,
                                            q_time, loop_time, self.row_time, self.n_data, d, a)
                else:
                    agg = engine.xlayer_fwd(fr, self.graph, fr.level, n_new, q_time, loop_time, self.row_time, self.n_data, hidden_p, rela_p,
                                            time_p, d, a_s, a_r, a_q, w_alpha, zero_b, a)
                hidden = self.act(agg[:, :d])                                                     # :238-239
            nodes, _, _ = fr.nodes(want_prev=False, want_old_new=False)
            result = tall_linear(hidden, self.linear_classifier.weight, self.linear_classifier.bias).reshape(-1)   # :244
            b_idx = nodes[:, 0].long()
            score_all = torch.zeros(n * self.n_ent, device=device).index_copy(0, b_idx * self.n_ent + nodes[:, 1].long(), result)
            # scatter_softmax(result, cur_entity[:, 0]) (:248): per-query softmax over the visited entities
            row_max = torch.full((n,), float("-inf"), device=device).scatter_reduce(0, b_idx, result.detach(), "amax")
            ex = torch.exp(result - row_max[b_idx])
            soft = ex / torch.zeros(n, device=device).index_add(0, b_idx, ex)[b_idx]
        if not with_grad:
            fr.set_window(None, None, 0)         # (a training forward's frontier keeps its windows for the backward: every reset sets them anew)
        self.last_stats = dict(n_edges=n_edges, n_nodes=int(nodes.shape[0]))
        return score_all.view(n, self.n_ent), (soft, nodes.long().cpu().numpy())


def segment_rank_fil(t, entities, target_idx_l, sp2o, spt2o, queries_sub, queries_pre, queries_ts):
    """Temporal/extrapolation/segment.py:346-387 with the same arguments and results (rank, found_mask, rank_fil, rank_fil_t): the
    rank of every query's target among ITS visited entities by score ``t`` (ties count half), raw, filtered by the other known
    objects of (s, p) and by those of (s, p, ts).  numpy per segment instead of a Python list comprehension per entity."""
    t = np.asarray(t.detach().cpu() if torch.is_tensor(t) else t)
    entities = np.asarray(entities)
    mask = entities[1:, 0] != entities[:-1, 0]
    key_idx = np.concatenate([[0], np.arange(1, len(entities))[mask], [len(entities)]]).astype(np.int64)
    rank, rank_fil, rank_fil_t, found = [], [], [], []
    for i, (s, e) in enumerate(zip(key_idx[:-1], key_idx[1:])):
        ents, sc = entities[s:e, 1], t[s:e]
        arg = np.nonzero(ents == target_idx_l[i])[0]
        if arg.size == 0:
            found.append(False); rank.append(1e9); rank_fil.append(1e9)        # (rank_fil_t gets no entry: as the reference)
            continue
        found.append(True)
        ts = sc[arg]

        def one(keep):
            return float(np.sum(sc[keep] > ts)) + (float(np.sum(sc[keep] == ts)) - 1) / 2 + 1

        everything = np.ones(len(ents), dtype=bool)
        rank.append(one(everything))
        other = np.setdiff1d(sp2o[(queries_sub[i], queries_pre[i])], [target_idx_l[i]])
        rank_fil.append(one(~np.isin(ents, other)))
        other_t = np.setdiff1d(spt2o[(queries_sub[i], queries_pre[i], queries_ts[i])], [target_idx_l[i]])
        rank_fil_t.append(one(~np.isin(ents, other_t)))
    return np.array(rank), found, np.array(rank_fil), np.array(rank_fil_t)
