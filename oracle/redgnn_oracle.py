"""CPU restatement of RED-GNN's relational-digraph message-passing path.  TEST INFRASTRUCTURE.

This file is the *oracle*: a plain numpy / torch-CPU restatement of the reference algorithm
(LARS-research/RED-GNN, Static/transductive).  It is the checker for the HIP path and the
``cpu_baseline`` leg of bench.py.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline may import it; the product package ``red_gnn_amd`` never does.

Parity status: the reference ships no tests or golden vectors, so this oracle
is pinned by fixtures generated from the reference itself, imported in the build container
(tests/golden/make_golden.py).  ``load_data.py`` and ``utils.py`` of the reference import
unmodified, so frontier expansion (a4), graph build (a7) and ranking (a8) are pinned against
the real code.  ``models.py`` needs the third-party ``torch_scatter`` 2.0.9, which is absent
offline; the generator substitutes that one op by its definition (zeros + index_add_), so the
layer arithmetic (a2/a3) is pinned only up to that definition: parity is UNPINNED at the
torch_scatter boundary.

Every function cites the reference file:line it follows (paths relative to
Static/transductive/ of the reference).
"""
import numpy as np
import torch

try:  # scipy is on the image; the closed form below is used when it is not
    from scipy.stats import rankdata as _rankdata
except Exception:  # pragma: no cover
    _rankdata = None


# --------------------------------------------------------------------------------------
# graph build  (load_data.py:69-89)
# --------------------------------------------------------------------------------------
def double_triple(triples, n_rel):
    """load_data.py:69-74 — append the inverse triple (t, r+n_rel, h) of every triple."""
    triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    inv = np.stack([triples[:, 2], triples[:, 1] + n_rel, triples[:, 0]], 1)
    return np.concatenate([triples, inv], 0)


class OracleGraph:
    """load_data.py:76-81 — KG rows = doubled triples followed by one identity row per entity
    (relation id 2*n_rel); ``M_sub`` (one-hot of the head of every row) is held as a CSR by head."""

    def __init__(self, doubled_triples, n_ent, n_rel):
        doubled_triples = np.asarray(doubled_triples, dtype=np.int64).reshape(-1, 3)
        ent = np.arange(n_ent, dtype=np.int64)
        idd = np.stack([ent, np.full(n_ent, 2 * n_rel, dtype=np.int64), ent], 1)
        self.KG = np.concatenate([doubled_triples, idd], 0)
        self.n_fact = len(self.KG)
        self.n_ent, self.n_rel = n_ent, n_rel
        order = np.argsort(self.KG[:, 0], kind="stable")        # fact rows grouped by head
        self.rows_by_head = order
        self.head_ptr = np.zeros(n_ent + 1, dtype=np.int64)
        np.add.at(self.head_ptr, self.KG[:, 0] + 1, 1)
        self.head_ptr = np.cumsum(self.head_ptr)


def get_neighbors(graph, nodes):
    """load_data.py:106-131 — frontier expansion.

    nodes: int64 [N,2] = (batch_idx, entity).  Returns (tail_nodes [N',2] sorted
    lexicographically, sampled_edges [E,6] = (batch, head, rel, tail, head_index, tail_index),
    old_nodes_new_idx [N]).  The edge order here is fact-row ascending and, inside a fact row,
    the order of ``nodes`` (the reference leaves the order inside a row unspecified).
    """
    nodes = np.asarray(nodes, dtype=np.int64).reshape(-1, 2)
    n_ent = graph.n_ent
    deg = graph.head_ptr[nodes[:, 1] + 1] - graph.head_ptr[nodes[:, 1]]
    tot = int(deg.sum())
    node_of_edge = np.repeat(np.arange(len(nodes)), deg)
    start = np.repeat(graph.head_ptr[nodes[:, 1]], deg)
    off = np.arange(tot) - np.repeat(np.cumsum(deg) - deg, deg)
    fact_row = graph.rows_by_head[start + off]                  # :116-117 (M_sub . node_1hot, nonzero)
    o = np.argsort(fact_row, kind="stable")
    fact_row, node_of_edge = fact_row[o], node_of_edge[o]
    batch = nodes[node_of_edge, 0]
    edges = np.concatenate([batch[:, None], graph.KG[fact_row]], 1)   # :118 (batch, head, rel, tail)

    hkey = edges[:, 0] * n_ent + edges[:, 1]                     # :122 unique(dim=0, sorted)
    tkey = edges[:, 0] * n_ent + edges[:, 3]                     # :123
    hu, head_index = np.unique(hkey, return_inverse=True)
    tu, tail_index = np.unique(tkey, return_inverse=True)
    tail_nodes = np.stack([tu // n_ent, tu % n_ent], 1)
    edges = np.concatenate([edges, head_index[:, None], tail_index[:, None]], 1)   # :125

    mask = edges[:, 2] == 2 * graph.n_rel                        # :127-129
    old_idx = np.argsort(head_index[mask], kind="stable")
    old_nodes_new_idx = tail_index[mask][old_idx]
    return tail_nodes, edges, old_nodes_new_idx


# --------------------------------------------------------------------------------------
# model  (models.py)
# --------------------------------------------------------------------------------------
_ACTS = {"relu": torch.relu, "tanh": torch.tanh, "idd": lambda x: x}


def _t(x, dtype):
    return (x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))).to(dtype)


def gnn_layer_forward(p, prefix, q_rel, hidden, edges, n_node, act, dtype=torch.float32):
    """models.py:23-43 — one GNNLayer.  ``p`` maps state-dict names to tensors."""
    g = lambda k: _t(p[prefix + k], dtype)
    edges = torch.as_tensor(edges, dtype=torch.long)
    sub, rel, obj, r_idx = edges[:, 4], edges[:, 2], edges[:, 5], edges[:, 0]
    rela = g("rela_embed.weight")
    hs = hidden[sub]                                             # :29
    hr = rela[rel]                                               # :30
    h_qr = rela[torch.as_tensor(q_rel, dtype=torch.long)][r_idx]  # :32-33
    message = hs + hr                                            # :35
    pre = hs @ g("Ws_attn.weight").T + hr @ g("Wr_attn.weight").T \
        + h_qr @ g("Wqr_attn.weight").T + g("Wqr_attn.bias")
    alpha = torch.sigmoid(torch.relu(pre) @ g("w_alpha.weight").T + g("w_alpha.bias"))   # :36
    message = alpha * message                                    # :37
    agg = torch.zeros(n_node, hidden.shape[1], dtype=dtype).index_add_(0, obj, message)  # :39 scatter-sum
    return act(agg @ g("W_h.weight").T), agg, alpha              # :41


def gru_step(p, x, h, dtype=torch.float32):
    """models.py:63,83 — single-step nn.GRU (gate rows ordered [r; z; n])."""
    w_ih, w_hh = _t(p["gate.weight_ih_l0"], dtype), _t(p["gate.weight_hh_l0"], dtype)
    b_ih, b_hh = _t(p["gate.bias_ih_l0"], dtype), _t(p["gate.bias_hh_l0"], dtype)
    d = x.shape[1]
    gi = x @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    r = torch.sigmoid(gi[:, :d] + gh[:, :d])
    z = torch.sigmoid(gi[:, d:2 * d] + gh[:, d:2 * d])
    n = torch.tanh(gi[:, 2 * d:] + r * gh[:, 2 * d:])
    return (1 - z) * n + z * h


def forward(p, graph, subs, rels, n_layer, act="relu", dtype=torch.float32, trace=None):
    """models.py:65-89 — RED_GNN_trans.forward in eval mode (dropout = identity).

    Returns scores_all [B, n_ent].  If ``trace`` is a list, a dict per layer is appended
    with nodes / edges / old_nodes_new_idx / agg / hidden.
    """
    subs = np.asarray(subs, dtype=np.int64)
    rels = np.asarray(rels, dtype=np.int64)
    n = len(subs)
    d = p["W_final.weight"].shape[1]
    actf = _ACTS[act]
    h0 = torch.zeros(n, d, dtype=dtype)                          # :72
    nodes = np.stack([np.arange(n, dtype=np.int64), subs], 1)    # :73
    hidden = torch.zeros(n, d, dtype=dtype)                      # :74
    for i in range(n_layer):                                     # :77
        nodes, edges, old_new = get_neighbors(graph, nodes)      # :78
        hidden, agg, alpha = gnn_layer_forward(p, "gnn_layers.%d." % i, rels, hidden, edges,
                                               len(nodes), actf, dtype)   # :80
        h0n = torch.zeros(len(nodes), d, dtype=dtype)
        h0n[torch.as_tensor(old_new)] = h0                       # :81 index_copy_
        hidden = gru_step(p, hidden, h0n, dtype)                 # :82-84 (dropout is identity in eval)
        h0 = hidden
        if trace is not None:
            trace.append(dict(nodes=nodes, edges=edges, old_nodes_new_idx=old_new,
                              agg=agg, alpha=alpha, hidden=hidden))
    scores = hidden @ _t(p["W_final.weight"], dtype).T           # :86
    scores_all = torch.zeros(n, graph.n_ent, dtype=dtype)        # :87
    scores_all[torch.as_tensor(nodes[:, 0]), torch.as_tensor(nodes[:, 1])] = scores[:, 0]   # :88
    return scores_all


# --------------------------------------------------------------------------------------
# temporal interpolation  (Temporal/interpolation/model_cuda.py:98-213, model.py:40-105)
# --------------------------------------------------------------------------------------
def temporal_forward(p, quads, n_ent, heads, rels, times, n_layer, act, shared_tables=False, dtype=torch.float32, trace=None):
    """T_RED_GNN.forward in eval mode (no fact deletion, dropout = identity).

    quads int [n,4] = (head, rel, tail, time id) incl. the identity rows, as graph.py:34-49 builds it.
    ``shared_tables``: parameter layout of model.py (one rela_embed / attention_1 / attention_2, leaky_relu in the
    reference) instead of model_cuda.py's per-layer tables.  Only model.py can be imported in the build container
    (model_cuda.py needs tkinter.tix / pyvis / pickles), so the per-layer-table variant is pinned through the shared
    per-edge arithmetic only: parity unpinned for what differs (which table a layer reads, the activation choice)."""
    quads = np.asarray(quads, dtype=np.int64).reshape(-1, 4)
    heads, rels, times = (np.asarray(x, dtype=np.int64) for x in (heads, rels, times))
    n = len(heads)
    g = lambda k: _t(p[k], dtype)
    d = g("time_embed.weight").shape[1]
    acts = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": torch.relu, "idd": lambda x: x,
            "softplus": torch.nn.functional.softplus, "leaky_relu": torch.nn.functional.leaky_relu}
    actf = acts[act]
    order = np.argsort(quads[:, 0], kind="stable")
    ptr = np.zeros(n_ent + 1, dtype=np.int64)
    np.add.at(ptr, quads[:, 0] + 1, 1)
    ptr = np.cumsum(ptr)
    cur = np.stack([np.arange(n), heads], 1)                        # model_cuda.py:108
    hidden = torch.zeros(n, d, dtype=dtype)
    qt = torch.as_tensor(times)
    for i in range(n_layer):
        if shared_tables:
            rela, w1, w2 = g("rela_embed.weight"), g("attention_1.weight"), g("attention_2.weight")
        else:
            rela, w1, w2 = g("rela_embed_layer.%d.weight" % i), g("attention_1_layer.%d.weight" % i), g("attention_2_layer.%d.weight" % i)
        deg = ptr[cur[:, 1] + 1] - ptr[cur[:, 1]]
        node_of = np.repeat(np.arange(len(cur)), deg)               # :141-145  rows whose head is in the frontier of the same query
        rows = order[np.repeat(ptr[cur[:, 1]], deg) + (np.arange(int(deg.sum())) - np.repeat(np.cumsum(deg) - deg, deg))]
        sel = np.concatenate([cur[node_of, 0][:, None], quads[rows]], 1)        # (batch, head, rel, tail, time)
        b_idx = torch.as_tensor(sel[:, 0])
        rel_t = torch.as_tensor(sel[:, 2])
        dt = torch.as_tensor(sel[:, 4]) - qt[b_idx]                 # :149
        hs = hidden[torch.as_tensor(node_of)]
        embed = hs + rela[rel_t] + g("time_embed.weight")[dt.abs()]             # :152
        out = torch.zeros_like(embed)
        out[dt > 0] = embed[dt > 0] @ g("future_linear.weight").T               # :155-157
        out[dt == 0] = embed[dt == 0] @ g("now_linear.weight").T
        out[dt < 0] = embed[dt < 0] @ g("past_linear.weight").T
        att_in = torch.cat([hs, rela[rel_t], rela[torch.as_tensor(rels)][b_idx]], 1)      # :159
        alpha = torch.sigmoid(torch.relu(att_in @ w1.T) @ w2.T)                 # :160
        key = sel[:, 0] * n_ent + sel[:, 3]                         # :175 unique (batch, tail), sorted
        uk, inv = np.unique(key, return_inverse=True)
        agg = torch.zeros(len(uk), d, dtype=dtype).index_add_(0, torch.as_tensor(inv), alpha * out)   # :192
        hidden = actf(agg)                                          # :196
        cur = np.stack([uk // n_ent, uk % n_ent], 1)
        if trace is not None:
            trace.append(dict(nodes=cur, n_edges=len(sel), hidden=hidden))
    result = (hidden @ g("linear_classifier.weight").T + g("linear_classifier.bias")).reshape(-1)   # :210
    score_all = torch.zeros(n, n_ent, dtype=dtype)
    score_all[torch.as_tensor(cur[:, 0]), torch.as_tensor(cur[:, 1])] = result
    return score_all


# --------------------------------------------------------------------------------------
# temporal extrapolation  (Temporal/extrapolation/model_cuda_new_embedding.py:135-265, utils.py:692-699)
# PARITY UNPINNED: that model file cannot be imported in the build container (torch_scatter, pyvis, rtdl_revisiting_models are
# absent) and the reference holds no outputs of it; this is a line-by-line restatement, checked by nothing but reading.
# --------------------------------------------------------------------------------------
def get_time_offset_list(data, time_granularity=24):
    """utils.py:692-699, literally (the loop; offsets of days without rows stay 0)."""
    data = np.asarray(data)
    max_time = max(data[:, 3] // time_granularity)
    offset_list = np.zeros(max_time + 2, dtype=np.int32)
    for idx, (_, _, _, t) in enumerate(data):
        offset_list[t // time_granularity + 1] = idx
    return offset_list


def periodic_embedding(p, prefix, x, dtype=torch.float32):
    """rtdl_num_embeddings.py:92-100,199-215 (the reference's edited PeriodicEmbeddings, one feature, lite=False): x [N,1] -> [N,d]."""
    g = lambda k: _t(p[prefix + k], dtype)
    x = x.to(dtype)
    is_neg = (x < 0).unsqueeze(-1)
    z = 2 * np.pi * g("periodic.weight") * x[..., None]
    z = torch.cat([torch.cos(z), torch.sin(z)], -1)
    neg = (z[..., None, :] @ g("linear_neg.weight")).squeeze(-2) + g("linear_neg.bias")
    pos = (z[..., None, :] @ g("linear_pos.weight")).squeeze(-2) + g("linear_pos.bias")
    return torch.relu(neg * is_neg + pos * (~is_neg)).squeeze(1)


def extrap_forward(p, data, time_offset_list, time_granularity, n_ent, n_rel_true, src, rel, ts, n_layer, act, dtype=torch.float32,
                   trace=None):
    """T_RED_GNN.forward of the extrapolation setting (model_cuda_new_embedding.py:135-261) without the attention statistics.
    Returns (score_all [B, n_ent], softmax over each query's visited entities [N], visited (batch, entity) [N,2])."""
    data = np.asarray(data, dtype=np.int64).reshape(-1, 4)
    src, rel, ts = (np.asarray(x, dtype=np.int64) for x in (src, rel, ts))
    g = lambda k: _t(p[k], dtype)
    acts = {"tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": torch.relu, "idd": lambda x: x,
            "softplus": torch.nn.functional.softplus, "leakyrelu": torch.nn.functional.leaky_relu}
    B = len(src)
    d = g("linear_classifier.weight").shape[1]
    cur_t = ts // time_granularity                                                         # :138
    rows = []
    for b in range(B):                                                                     # :165-176
        begin = max(int(cur_t[b]) - 120, 0)
        batch = data[time_offset_list[begin]:time_offset_list[cur_t[b]]]
        loops = np.column_stack([np.arange(n_ent), np.full(n_ent, n_rel_true), np.arange(n_ent), np.full(n_ent, begin * time_granularity)])
        batch = np.concatenate([loops, batch], 0)
        rows.append(np.concatenate([np.full((len(batch), 1), b), batch], 1))
    rel_all = np.concatenate(rows, 0).astype(np.int64)                                      # (batch, subject, relation, object, time)
    key_all = rel_all[:, 0] * n_ent + rel_all[:, 1]
    cur = np.stack([np.arange(B), src], 1)                                                 # :139
    hidden = torch.zeros(B, d, dtype=dtype)
    for i in range(n_layer):
        cur_key = cur[:, 0] * n_ent + cur[:, 1]
        sel = rel_all[np.isin(key_all, cur_key)]                                           # :186-189 (rows whose subject is in the frontier)
        src_pos = torch.as_tensor(np.searchsorted(cur_key, sel[:, 0] * n_ent + sel[:, 1]))     # node_new_index[...] - 1 (cur is sorted)
        b_idx = torch.as_tensor(sel[:, 0])
        rela = g("rela_embed_layer.%d.weight" % i)
        rel_t = torch.as_tensor(sel[:, 2])
        delta = torch.as_tensor(cur_t)[b_idx] - torch.as_tensor(sel[:, 4] // time_granularity)   # :192
        tau = periodic_embedding(p, "time_embed.", delta.reshape(-1, 1), dtype)           # :201
        hs = hidden[src_pos]
        msg = (hs + rela[rel_t] + tau) @ g("past_linear.weight").T                        # :203-205
        att_in = torch.cat([hs, rela[rel_t], rela[torch.as_tensor(rel)][b_idx]], 1)        # :207
        alpha = torch.sigmoid(torch.relu(att_in @ g("attention_1_layer.%d.weight" % i).T) @ g("attention_2_layer.%d.weight" % i).T)   # :208
        new_key = sel[:, 0] * n_ent + sel[:, 3]                                            # :224 unique(batch, object), sorted
        uk, inv = np.unique(new_key, return_inverse=True)
        agg = torch.zeros(len(uk), d, dtype=dtype).index_add_(0, torch.as_tensor(inv), alpha * msg)     # :236 scatter sum
        hidden = acts[act](agg)                                                            # :238
        cur = np.stack([uk // n_ent, uk % n_ent], 1)
        if trace is not None:
            trace.append(dict(nodes=cur, n_edges=len(sel), hidden=hidden))
    result = (hidden @ g("linear_classifier.weight").T + g("linear_classifier.bias")).reshape(-1)     # :244
    score_all = torch.zeros(B, n_ent, dtype=dtype)
    score_all[torch.as_tensor(cur[:, 0]), torch.as_tensor(cur[:, 1])] = result
    seg = torch.as_tensor(cur[:, 0])
    mx = torch.full((B,), float("-inf"), dtype=dtype).scatter_reduce(0, seg, result, "amax")
    ex = torch.exp(result - mx[seg])
    soft = ex / torch.zeros(B, dtype=dtype).index_add_(0, seg, ex)[seg]                    # :248 scatter_softmax
    return score_all, soft, cur


def loss_fn(scores, pos_tail):
    """base_model.py:58-60 — full-softmax cross entropy, restated literally.

    NOTE the reference keeps ``max_n`` as [n,1] (keepdim) while ``pos_scores`` and the
    log-sum-exp are [n], so ``-pos + max_n + log(...)`` broadcasts to an [n,n] matrix whose
    sum is n * sum_i(-s[i,t_i] + max_i + log sum_j exp(s[i,j] - max_i)).  That factor n is
    part of the reference's behaviour (it scales every gradient) and is kept."""
    idx = torch.arange(len(scores))
    pos = scores[idx, torch.as_tensor(pos_tail, dtype=torch.long)]
    max_n = torch.max(scores, 1, keepdim=True)[0]
    return torch.sum(-pos + max_n + torch.log(torch.sum(torch.exp(scores - max_n), 1)))


# --------------------------------------------------------------------------------------
# ranking  (utils.py:7-21)
# --------------------------------------------------------------------------------------
def cal_ranks(scores, labels, filters):
    """utils.py:7-14 — filtered ranks of all answers (scipy.rankdata form, as the reference)."""
    if _rankdata is None:
        return cal_ranks_closed_form(scores, labels, filters)
    scores = scores - np.min(scores, axis=1, keepdims=True) + 1e-8
    full_rank = _rankdata(-scores, method="average", axis=1)
    filter_scores = scores * filters
    filter_rank = _rankdata(-filter_scores, method="min", axis=1)
    ranks = (full_rank - filter_rank + 1) * labels
    ranks = ranks[np.nonzero(ranks)]
    return list(ranks)


def cal_ranks_closed_form(scores, labels, filters):
    """The same ranks without sorting:
    rank(a) = #{j not in filter: s'_j > s'_a} + (#{j: s'_j == s'_a} + 1)/2
    with s' = fl32(fl32(s - rowmin) + 1e-8).  Answers are a subset of the filter set."""
    scores = np.asarray(scores)
    s = scores - np.min(scores, axis=1, keepdims=True) + 1e-8
    out = []
    for i in range(len(s)):
        row = s[i]
        nf = np.asarray(filters[i]) == 0
        for a in np.nonzero(labels[i])[0]:
            gt = np.count_nonzero((row > row[a]) & nf)
            eq = np.count_nonzero(row == row[a])
            out.append(gt + (eq + 1) / 2.0)
    return out


def cal_performance(ranks):
    """utils.py:17-21."""
    ranks = np.asarray(ranks, dtype=np.float64)
    mrr = (1.0 / ranks).sum() / len(ranks)
    h_1 = np.sum(ranks <= 1) * 1.0 / len(ranks)
    h_10 = np.sum(ranks <= 10) * 1.0 / len(ranks)
    return mrr, h_1, h_10
