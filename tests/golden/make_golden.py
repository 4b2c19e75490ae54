"""Generate the golden fixtures in this directory from the REFERENCE itself.

Runs only in the build container, where the reference is mounted read-only at
/root/reference.  It imports the reference's ``load_data`` / ``utils`` / ``models`` modules
unmodified, runs them on CPU and writes small ``.npz`` fixtures (inputs + expected outputs).
Nothing of the reference's code is copied: only numbers travel.

Two things the reference needs that this image lacks:
  * ``torch_scatter`` 2.0.9 (third party, not vendored, not installable offline).  Its single
    use on the path, ``scatter(src, index=obj, dim=0, dim_size=n, reduce='sum')``
    (Static/transductive/models.py:39), is registered here as a module object whose ``scatter``
    is the op's definition: zeros(dim_size, D).index_add_(0, index, src).  Parity of the layer
    arithmetic is therefore pinned only up to that definition ("unpinned" at that boundary).
  * a GPU: the model hard-codes ``.cuda()``; this harness makes ``.cuda()`` the identity.

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Static/transductive"

sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

# --- the one missing third-party op, by definition --------------------------------------
_scatter_log = []


def _scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    assert dim == 0 and reduce == "sum"
    if dim_size is None:                      # torch_scatter's default: index.max() + 1
        dim_size = int(index.max()) + 1
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)
    _scatter_log.append(out.detach().clone())
    return out


_ts = types.ModuleType("torch_scatter")
_ts.scatter = _scatter
sys.modules["torch_scatter"] = _ts
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

sys.path.insert(0, REF)
_cwd = os.getcwd()
os.chdir(REF)
import load_data as ref_load_data   # noqa: E402
import models as ref_models         # noqa: E402
import utils as ref_utils           # noqa: E402
os.chdir(_cwd)

from red_gnn_amd.synthetic import make_synthetic_kg, write_task_dir   # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def edge_multiset_hash(edges):
    """Order-independent hash of the (batch, head, rel, tail) multiset."""
    e = np.asarray(edges)[:, :4].astype(np.int64)
    o = np.lexsort((e[:, 3], e[:, 2], e[:, 1], e[:, 0]))
    return sha(e[o])


class Params:
    pass


def make_params(n_layer, hidden_dim, attn_dim, n_rel, act, dropout=0.0):
    p = Params()
    p.n_layer, p.hidden_dim, p.attn_dim, p.n_rel, p.act, p.dropout = n_layer, hidden_dim, attn_dim, n_rel, act, dropout
    return p


def ids_of(loader):
    return dict(n_ent=np.int64(loader.n_ent), n_rel=np.int64(loader.n_rel),
                facts=np.array(loader.fact_triple, dtype=np.int32),
                train=np.array(loader.train_triple, dtype=np.int32),
                valid=np.array(loader.valid_triple, dtype=np.int32),
                test=np.array(loader.test_triple, dtype=np.int32))


def run_forward(loader, params, subs, rels, mode, seed=1234, keep_edges=False, train_mode=False):
    """Run the unmodified reference model; capture every per-layer intermediate."""
    np.random.seed(seed)
    torch.manual_seed(seed)
    model = ref_models.RED_GNN_trans(params, loader)
    model.train() if train_mode else model.eval()
    rec = {}
    expand_log = []
    orig = loader.get_neighbors

    def spy(nodes, mode="train"):
        out = orig(nodes, mode=mode)
        expand_log.append(out)
        return out

    loader.get_neighbors = spy
    layer_out, gru_out = [], []
    hooks = [l.register_forward_hook(lambda m, i, o: layer_out.append(o.detach().clone())) for l in model.gnn_layers]
    hooks.append(model.gate.register_forward_hook(lambda m, i, o: gru_out.append(o[0].detach().clone().squeeze(0))))
    _scatter_log.clear()
    scores = model(np.asarray(subs), np.asarray(rels), mode=mode)
    for h in hooks:
        h.remove()
    loader.get_neighbors = orig
    for i, (nodes, edges, old_new) in enumerate(expand_log):
        rec["L%d_nodes" % i] = nodes.numpy().astype(np.int32)
        rec["L%d_old_nodes_new_idx" % i] = old_new.numpy().astype(np.int32)
        rec["L%d_n_edges" % i] = np.int64(edges.shape[0])
        rec["L%d_edge_hash" % i] = np.array(edge_multiset_hash(edges.numpy()))
        if keep_edges:
            rec["L%d_edges" % i] = edges.numpy().astype(np.int32)
        rec["L%d_agg" % i] = _scatter_log[i].numpy()
        rec["L%d_layer_out" % i] = layer_out[i].numpy()
        rec["L%d_hidden" % i] = gru_out[i].numpy()
    rec["scores"] = scores.detach().numpy()
    rec["subs"] = np.asarray(subs, dtype=np.int32)
    rec["rels"] = np.asarray(rels, dtype=np.int32)
    for k, v in model.state_dict().items():
        rec["param::" + k] = v.numpy()
    rec["cfg"] = np.array([params.n_layer, params.hidden_dim, params.attn_dim], dtype=np.int64)
    rec["act"] = np.array(params.act)
    rec["mode"] = np.array(mode)
    return model, scores, rec


def eval_queries(loader, which, idx):
    """(subs, rels, labels, filters) for query indices ``idx`` — what base_model.py:105-115 builds.
    (loader.get_batch(data='valid'|'test') raises on numpy>=1.24 for ragged answers, so the
    query/answer lists are read directly.)"""
    q = loader.valid_q if which == "valid" else loader.test_q
    a = loader.valid_a if which == "valid" else loader.test_a
    subs = np.array([q[i][0] for i in idx])
    rels = np.array([q[i][1] for i in idx])
    objs = np.zeros((len(idx), loader.n_ent))
    filt = np.zeros((len(idx), loader.n_ent))
    for k, i in enumerate(idx):
        objs[k][a[i]] = 1
        filt[k][np.array(loader.filters[(subs[k], rels[k])])] = 1
    return subs, rels, objs, filt


def case_tiny(out_dir):
    kg = make_synthetic_kg(50, 4, 300, seed=7)
    with tempfile.TemporaryDirectory() as td:
        write_task_dir(kg, td)
        loader = ref_load_data.DataLoader(td)
    params = make_params(3, 16, 5, loader.n_rel, "relu", dropout=0.0)
    subs, rels, objs, filt = eval_queries(loader, "test", [0, 1, 2, 3])
    _, scores, rec = run_forward(loader, params, subs, rels, "test", keep_edges=True)
    rec.update(ids_of(loader))
    rec["labels"], rec["filters"] = objs.astype(np.uint8), filt.astype(np.uint8)
    rec["ranks"] = np.array(ref_utils.cal_ranks(scores.detach().numpy(), objs, filt))
    np.savez_compressed(os.path.join(out_dir, "tiny_fwd.npz"), **rec)

    # backward: train graph, train mode, dropout = 0, loss of base_model.py:58-60
    trip = loader.train_data[:6]
    model, scores, rec = run_forward(loader, params, trip[:, 0], trip[:, 1], "train", keep_edges=True, train_mode=True)
    pos = scores[[torch.arange(len(scores)), torch.LongTensor(trip[:, 2])]]
    max_n = torch.max(scores, 1, keepdim=True)[0]
    loss = torch.sum(-pos + max_n + torch.log(torch.sum(torch.exp(scores - max_n), 1)))
    loss.backward()
    rec.update(ids_of(loader))
    rec["tails"] = trip[:, 2].astype(np.int32)
    rec["loss"] = np.float64(loss.item())
    for k, v in model.named_parameters():
        rec["grad::" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "tiny_bwd.npz"), **rec)


def case_dataset(out_dir, name, n_layer, dims, act, n_q, with_ids=True, light=False):
    cwd = os.getcwd()
    os.chdir(REF)
    loader = ref_load_data.DataLoader(os.path.join("data", name))
    os.chdir(cwd)
    if with_ids:
        np.savez_compressed(os.path.join(out_dir, "%s_ids.npz" % name), **ids_of(loader))
    subs, rels, objs, filt = eval_queries(loader, "test", list(range(n_q)))
    for d in dims:
        params = make_params(n_layer, d, 5, loader.n_rel, act, dropout=0.0)
        _, scores, rec = run_forward(loader, params, subs, rels, "test")
        if light:   # sizes + hashes only (large graphs)
            keep = {}
            for k, v in rec.items():
                if k.endswith("_nodes"):
                    keep[k[:-6] + "_n_nodes"] = np.int64(len(v))
                    keep[k + "_hash"] = np.array(sha(v.astype(np.int64)))
                elif k.endswith(("_n_edges", "_edge_hash")) or k.startswith("param::") or k in ("subs", "rels", "cfg", "act", "mode"):
                    keep[k] = v
            sc = scores.detach().numpy()
            keep["score_nnz"] = np.int64(np.count_nonzero(sc))
            vis = rec["L%d_nodes" % (n_layer - 1)]
            keep["scores_visited"] = sc[vis[:, 0], vis[:, 1]]
            rec = keep
        else:       # keep fixtures small: per-layer hidden only (agg / layer_out stay in the tiny case)
            last = "L%d_hidden" % (n_layer - 1)
            rec = {k: v for k, v in rec.items()
                   if not k.endswith(("_agg", "_layer_out")) and (d == dims[-1] or not k.endswith("_hidden") or k == last)}
        rec["ranks"] = np.array(ref_utils.cal_ranks(scores.detach().numpy(), objs, filt))
        rec["labels_idx"] = np.stack(np.nonzero(objs), 1).astype(np.int32)
        rec["filters_idx"] = np.stack(np.nonzero(filt), 1).astype(np.int32)
        np.savez_compressed(os.path.join(out_dir, "%s_d%d.npz" % (name, d)), **rec)


def case_inductive(out_dir, name="WN18RR_v1", d=32, n_layer=3, n_q=4):
    """Static/inductive: same GNNLayer, two graphs (tra / ind), n_ent switches with the mode
    (inductive/models.py:65-89, inductive/load_data.py:7-192)."""
    import importlib.util
    ind_dir = "/root/reference/Static/inductive"

    def load(mod):
        spec = importlib.util.spec_from_file_location("ref_ind_" + mod, os.path.join(ind_dir, mod + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    ild, imodels = load("load_data"), load("models")
    cwd = os.getcwd()
    os.chdir(ind_dir)
    loader = ild.DataLoader(os.path.join("data", name))
    os.chdir(cwd)
    ids = dict(n_ent=np.int64(loader.n_ent), n_rel=np.int64(loader.n_rel), n_ent_ind=np.int64(loader.n_ent_ind),
               tra_kg=loader.tra_KG[:-loader.n_ent].astype(np.int32), ind_kg=loader.ind_KG[:-loader.n_ent_ind].astype(np.int32),
               tra_valid=np.array(loader.tra_valid, dtype=np.int32), tra_test=np.array(loader.tra_test, dtype=np.int32),
               ind_valid=np.array(loader.ind_valid, dtype=np.int32), ind_test=np.array(loader.ind_test, dtype=np.int32))
    np.savez_compressed(os.path.join(out_dir, "ind_%s_ids.npz" % name), **ids)
    params = make_params(n_layer, d, 5, loader.n_rel, "relu", dropout=0.0)
    global ref_models
    keep = ref_models
    ref_models = imodels
    imodels.RED_GNN_trans = imodels.RED_GNN_induc          # run_forward instantiates .RED_GNN_trans
    try:
        for mode, data, filt_d in (("transductive", "valid", loader.val_filters), ("inductive", "test", loader.tst_filters)):
            q = loader.valid_q if data == "valid" else loader.test_q
            a = loader.valid_a if data == "valid" else loader.test_a
            n_ent = loader.n_ent if mode == "transductive" else loader.n_ent_ind
            idx = list(range(n_q))
            subs = np.array([q[i][0] for i in idx]); rels = np.array([q[i][1] for i in idx])
            objs = np.zeros((n_q, n_ent)); filt = np.zeros((n_q, n_ent))
            for k, i in enumerate(idx):
                objs[k][a[i]] = 1
                filt[k][np.array(filt_d[(subs[k], rels[k])])] = 1
            _, scores, rec = run_forward(loader, params, subs, rels, mode)
            rec = {k: v for k, v in rec.items() if not k.endswith(("_agg", "_layer_out"))}
            rec["ranks"] = np.array(ref_utils.cal_ranks(scores.detach().numpy(), objs, filt))
            rec["labels_idx"] = np.stack(np.nonzero(objs), 1).astype(np.int32)
            rec["filters_idx"] = np.stack(np.nonzero(filt), 1).astype(np.int32)
            np.savez_compressed(os.path.join(out_dir, "ind_%s_%s.npz" % (name, mode)), **rec)
    finally:
        ref_models = keep


def case_temporal(out_dir):
    """Temporal/interpolation/model.py (the importable T_RED_GNN: d=20, a=30, 3 layers, leaky_relu, shared tables)
    on a synthetic quadruple graph shaped as graph.py:34-49 builds it (identity rows with the sentinel = largest
    time id).  model_cuda.py (per-layer tables) cannot be imported here (tkinter.tix, pyvis, pickles)."""
    import importlib.util
    tdir = "/root/reference/Temporal/interpolation"
    spec = importlib.util.spec_from_file_location("ref_temporal_model", os.path.join(tdir, "model.py"))
    tm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tm)
    rng = np.random.default_rng(11)
    n_ent, n_rel, n_time, n_q = 60, 6, 365, 700          # relation ids 0..4 + idd (5): len(relation_vocab) = 6
    h, t = rng.integers(0, n_ent, n_q), rng.integers(0, n_ent, n_q)
    r, tau = rng.integers(0, n_rel - 1, n_q), rng.integers(0, n_time - 1, n_q)
    tau[:40] = tau[40:80]                                 # some exact-same-day edges for the 'now' branch
    quads = np.stack([h, r, t, tau], 1)
    idd = np.stack([np.arange(n_ent), np.full(n_ent, n_rel - 1), np.arange(n_ent), np.full(n_ent, n_time - 1)], 1)
    graph = np.concatenate([quads, idd], 0).astype(np.int64)

    class Prm:
        relation_vocab, entity_vocab, device = list(range(n_rel)), list(range(n_ent)), "cpu"
    Prm.graph = graph
    np.random.seed(1234)
    torch.manual_seed(1234)
    model = tm.T_RED_GNN(Prm).eval()
    B = 6
    batch = {"head": torch.as_tensor(quads[:B, 0]), "relation": torch.as_tensor(quads[:B, 1]), "time": torch.as_tensor(quads[40:40 + B, 3]),
             "example_idx": torch.zeros(0, dtype=torch.long)}
    _scatter_log.clear()
    scores = model(batch)
    rec = dict(quads=graph.astype(np.int32), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), n_time=np.int64(n_time),
               heads=batch["head"].numpy().astype(np.int32), rels=batch["relation"].numpy().astype(np.int32),
               times=batch["time"].numpy().astype(np.int32), scores=scores.detach().numpy(),
               cfg=np.array([3, 20, 30], dtype=np.int64), act=np.array("leaky_relu"))
    for i, a in enumerate(_scatter_log):
        rec["L%d_agg" % i] = a.numpy()
    for k, v in model.state_dict().items():
        rec["param::" + k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, "temporal_model_py.npz"), **rec)


def case_temporal_train(out_dir):
    """One training step's gradients from Temporal/interpolation/model.py + the loss of main.py:70-82: the batch's own
    quadruples (example_idx) leave the graph (model.py:45), softmax + nll on the scores, parameter gradients recorded."""
    import importlib.util
    import torch.nn.functional as F
    tdir = "/root/reference/Temporal/interpolation"
    spec = importlib.util.spec_from_file_location("ref_temporal_model", os.path.join(tdir, "model.py"))
    tm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tm)
    rng = np.random.default_rng(12)
    n_ent, n_rel, n_time, n_q = 50, 6, 365, 900
    w = 1.0 / np.arange(1, n_ent + 1); w /= w.sum()
    h, t = rng.choice(n_ent, n_q, p=w), rng.integers(0, n_ent, n_q)       # a few hub heads (rows longer than one 128-entry segment)
    r, tau = rng.integers(0, n_rel - 1, n_q), rng.integers(0, n_time - 1, n_q)
    tau[:60] = tau[60:120]
    quads = np.stack([h, r, t, tau], 1)
    idd = np.stack([np.arange(n_ent), np.full(n_ent, n_rel - 1), np.arange(n_ent), np.full(n_ent, n_time - 1)], 1)
    graph = np.concatenate([quads, idd], 0).astype(np.int64)

    class Prm:
        relation_vocab, entity_vocab, device = list(range(n_rel)), list(range(n_ent)), "cpu"
    Prm.graph = graph
    np.random.seed(4321)
    torch.manual_seed(4321)
    model = tm.T_RED_GNN(Prm).train()
    ex = np.array([3, 17, 60, 61, 200, 411, 555, 899])
    batch = {"head": torch.as_tensor(quads[ex, 0]), "relation": torch.as_tensor(quads[ex, 1]), "time": torch.as_tensor(quads[ex, 3]),
             "tail": torch.as_tensor(quads[ex, 2]), "example_idx": torch.as_tensor(ex)}
    scores = model(batch)
    loss = F.nll_loss(torch.log(F.softmax(scores, dim=1) + 1e-12), batch["tail"])
    loss.backward()
    rec = dict(quads=graph.astype(np.int32), n_ent=np.int64(n_ent), n_rel=np.int64(n_rel), n_time=np.int64(n_time),
               heads=batch["head"].numpy().astype(np.int32), rels=batch["relation"].numpy().astype(np.int32),
               times=batch["time"].numpy().astype(np.int32), tails=batch["tail"].numpy().astype(np.int64),
               example_idx=ex.astype(np.int64), scores=scores.detach().numpy(), loss=np.float64(loss.item()),
               cfg=np.array([3, 20, 30], dtype=np.int64), act=np.array("leaky_relu"))
    for k, v in model.named_parameters():
        rec["param::" + k] = v.detach().numpy()
        rec["grad::" + k] = v.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "temporal_model_py_train.npz"), **rec)


def case_ranks(out_dir):
    rng = np.random.default_rng(5)
    n, m = 12, 300
    scores = np.zeros((n, m), dtype=np.float32)
    for i in range(n):
        k = int(rng.integers(1, 120))
        cols = rng.choice(m, k, replace=False)
        vals = rng.standard_normal(k).astype(np.float32)
        if i % 3 == 0:
            vals = np.round(vals, 1)            # many exact ties among visited entities
        scores[i, cols] = vals
    labels = np.zeros((n, m))
    filters = np.zeros((n, m))
    for i in range(n):
        f = rng.choice(m, int(rng.integers(1, 25)), replace=False)
        filters[i, f] = 1
        labels[i, f[: int(rng.integers(1, len(f) + 1))]] = 1
    ranks = np.array(ref_utils.cal_ranks(scores, labels, filters))
    perf = np.array(ref_utils.cal_performance(ranks))
    np.savez_compressed(os.path.join(out_dir, "ranks.npz"), scores=scores, labels=labels.astype(np.uint8),
                        filters=filters.astype(np.uint8), ranks=ranks, perf=perf)


def case_extrap_rank(out_dir):
    """segment_rank_fil of the reference's Temporal/extrapolation/segment.py:346-387 (imports unmodified: numpy + torch only) on
    segmented scores with exact ties, targets that some queries never reached, and filter sets that hide higher-scored entities."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_segment", "/root/reference/Temporal/extrapolation/segment.py")
    seg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(seg)
    rng = np.random.default_rng(7)
    n_q, n_ent = 40, 60
    ents, scores = [], []
    for q in range(n_q):
        k = int(rng.integers(1, 30))
        e = np.sort(rng.choice(n_ent, k, replace=False))
        v = np.round(rng.random(k), 1) if q % 3 == 0 else rng.random(k)           # exact ties in a third of the segments
        ents.append(np.stack([np.full(k, q), e], 1)); scores.append(v)
    entities = np.concatenate(ents, 0)
    t = torch.tensor(np.concatenate(scores), dtype=torch.float32)
    target = np.array([int(ents[q][rng.integers(0, len(ents[q])), 1]) if q % 5 else int((ents[q][-1, 1] + 1) % n_ent) for q in range(n_q)])
    sub, pre, ts = rng.integers(0, n_ent, n_q), rng.integers(0, 5, n_q), rng.integers(0, 50, n_q) * 24
    sp2o, spt2o = {}, {}
    sp_lists, spt_lists = [], []
    for q in range(n_q):
        a = np.unique(np.concatenate([[target[q]], rng.choice(n_ent, int(rng.integers(0, 12)), replace=False)]))
        b = np.unique(np.concatenate([[target[q]], rng.choice(a, int(rng.integers(0, len(a))), replace=False)]))
        sp2o[(sub[q], pre[q])] = a; spt2o[(sub[q], pre[q], ts[q])] = b
    for q in range(n_q):                      # (the dicts may have merged duplicate keys: store what the reference actually saw)
        sp_lists.append(np.asarray(sp2o[(sub[q], pre[q])])); spt_lists.append(np.asarray(spt2o[(sub[q], pre[q], ts[q])]))
    rank, found, rank_fil, rank_fil_t = seg.segment_rank_fil(t, entities, target, sp2o, spt2o, sub, pre, ts)
    ptr = lambda ls: np.concatenate([[0], np.cumsum([len(x) for x in ls])]).astype(np.int64)
    np.savez_compressed(os.path.join(out_dir, "extrap_rank.npz"), scores=t.numpy(), entities=entities, target=target, sub=sub, pre=pre, ts=ts,
                        sp_ptr=ptr(sp_lists), sp_idx=np.concatenate(sp_lists), spt_ptr=ptr(spt_lists), spt_idx=np.concatenate(spt_lists),
                        rank=rank, found=np.array(found), rank_fil=rank_fil, rank_fil_t=rank_fil_t)


if __name__ == "__main__":
    out = HERE
    if sys.argv[1:] == ["temporal_train"]:      # regenerate only this case
        case_temporal_train(out)
        sys.exit(0)
    if sys.argv[1:] == ["extrap_rank"]:
        case_extrap_rank(out)
        sys.exit(0)
    case_tiny(out)
    case_ranks(out)
    case_dataset(out, "family", 3, [48, 64], "relu", 8)
    case_dataset(out, "umls", 4, [48], "relu", 4)
    case_dataset(out, "WN18RR", 5, [48], "tanh", 4, light=True)
    case_inductive(out)
    case_temporal(out)
    case_temporal_train(out)
    case_extrap_rank(out)
    for f in sorted(os.listdir(out)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(out, f)))
