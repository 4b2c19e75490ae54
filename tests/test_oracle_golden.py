"""The oracle (CPU restatement) against fixtures generated from the reference itself.

These pin the oracle: node sets / edge multisets / old_nodes_new_idx bit-exact, hidden and
scores within fp32 tolerance, ranks exact.  Tolerance: rtol 1e-4, atol 1e-5 on fp32 values
(SURVEY.md §7 "Determinism / tolerance")."""
import numpy as np
import pytest
import torch

from oracle import redgnn_oracle as orc
from tests import _util as U

RTOL, ATOL = 1e-4, 1e-5


def _run(fx, ids, dtype=torch.float32):
    n_layer = int(fx["cfg"][0])
    g = U.oracle_graph(ids, str(fx["mode"]))
    trace = []
    scores = orc.forward(U.params_of(fx), g, fx["subs"], fx["rels"], n_layer, act=str(fx["act"]), dtype=dtype, trace=trace)
    return g, scores, trace


@pytest.mark.parametrize("name", ["tiny_fwd.npz", "tiny_bwd.npz"])
def test_tiny_every_intermediate(name):
    fx = U.load(name)
    g, scores, trace = _run(fx, fx)
    for i, t in enumerate(trace):
        assert np.array_equal(t["nodes"], fx["L%d_nodes" % i])
        assert np.array_equal(t["old_nodes_new_idx"], fx["L%d_old_nodes_new_idx" % i])
        assert np.array_equal(U.sorted_edges(t["edges"]), U.sorted_edges(fx["L%d_edges" % i]))
        np.testing.assert_allclose(t["agg"].numpy(), fx["L%d_agg" % i], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(t["hidden"].numpy(), fx["L%d_hidden" % i], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(scores.numpy(), fx["scores"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(scores.numpy() == 0, fx["scores"] == 0)


def test_tiny_ranks():
    fx = U.load("tiny_fwd.npz")
    ranks = orc.cal_ranks(fx["scores"], fx["labels"].astype(np.float64), fx["filters"].astype(np.float64))
    assert np.array_equal(np.array(ranks), fx["ranks"])
    ranks2 = orc.cal_ranks_closed_form(fx["scores"], fx["labels"], fx["filters"])
    assert np.array_equal(np.array(ranks2), fx["ranks"])


def test_tiny_backward():
    fx = U.load("tiny_bwd.npz")
    p = {k: torch.tensor(v, requires_grad=True) for k, v in U.params_of(fx).items()}
    g = U.oracle_graph(fx, "train")
    scores = orc.forward(p, g, fx["subs"], fx["rels"], int(fx["cfg"][0]), act=str(fx["act"]))
    loss = orc.loss_fn(scores, fx["tails"])
    assert abs(loss.item() - float(fx["loss"])) < 1e-4 * max(1.0, abs(float(fx["loss"])))
    loss.backward()
    for k, gref in U.grads_of(fx).items():
        np.testing.assert_allclose(p[k].grad.numpy(), gref, rtol=1e-3, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("name,ids", [("family_d48.npz", "family_ids.npz"), ("family_d64.npz", "family_ids.npz"),
                                      ("umls_d48.npz", "umls_ids.npz")])
def test_dataset_cases(name, ids):
    fx, ids = U.load(name), U.load(ids)
    g, scores, trace = _run(fx, ids)
    for i, t in enumerate(trace):
        assert np.array_equal(t["nodes"], fx["L%d_nodes" % i])
        assert np.array_equal(t["old_nodes_new_idx"], fx["L%d_old_nodes_new_idx" % i])
        assert len(t["edges"]) == int(fx["L%d_n_edges" % i])
        assert U.edge_multiset_hash(t["edges"]) == str(fx["L%d_edge_hash" % i])
        if "L%d_hidden" % i in fx:
            np.testing.assert_allclose(t["hidden"].numpy(), fx["L%d_hidden" % i], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(scores.numpy(), fx["scores"], rtol=RTOL, atol=ATOL)
    n_ent = int(ids["n_ent"])
    labels = np.zeros((len(fx["subs"]), n_ent)); labels[fx["labels_idx"][:, 0], fx["labels_idx"][:, 1]] = 1
    filt = np.zeros((len(fx["subs"]), n_ent)); filt[fx["filters_idx"][:, 0], fx["filters_idx"][:, 1]] = 1
    assert np.array_equal(np.array(orc.cal_ranks(fx["scores"], labels, filt)), fx["ranks"])
    assert np.array_equal(np.array(orc.cal_ranks_closed_form(fx["scores"], labels, filt)), fx["ranks"])


def test_wn18rr_node_sets():
    fx, ids = U.load("WN18RR_d48.npz"), U.load("WN18RR_ids.npz")
    g, scores, trace = _run(fx, ids)
    for i, t in enumerate(trace):
        assert len(t["nodes"]) == int(fx["L%d_n_nodes" % i])
        assert U.sha(t["nodes"].astype(np.int64)) == str(fx["L%d_nodes_hash" % i])
        assert len(t["edges"]) == int(fx["L%d_n_edges" % i])
        assert U.edge_multiset_hash(t["edges"]) == str(fx["L%d_edge_hash" % i])
    vis = trace[-1]["nodes"]
    np.testing.assert_allclose(scores.numpy()[vis[:, 0], vis[:, 1]], fx["scores_visited"], rtol=RTOL, atol=ATOL)


def test_ranks_heavy_ties():
    fx = U.load("ranks.npz")
    lab, fil = fx["labels"].astype(np.float64), fx["filters"].astype(np.float64)
    assert np.array_equal(np.array(orc.cal_ranks(fx["scores"], lab, fil)), fx["ranks"])
    assert np.array_equal(np.array(orc.cal_ranks_closed_form(fx["scores"], lab, fil)), fx["ranks"])
    np.testing.assert_allclose(np.array(orc.cal_performance(fx["ranks"])), fx["perf"], rtol=1e-12)


@pytest.mark.parametrize("mode", ["transductive", "inductive"])
def test_inductive_fixture(mode):
    """Static/inductive: same layer, two graphs; the oracle on the reference's own outputs."""
    fx, ids = U.load("ind_WN18RR_v1_%s.npz" % mode), U.load("ind_WN18RR_v1_ids.npz")
    n_rel = int(ids["n_rel"])
    g = orc.OracleGraph(ids["tra_kg"], int(ids["n_ent"]), n_rel) if mode == "transductive" else \
        orc.OracleGraph(ids["ind_kg"], int(ids["n_ent_ind"]), n_rel)
    trace = []
    scores = orc.forward(U.params_of(fx), g, fx["subs"], fx["rels"], int(fx["cfg"][0]), act=str(fx["act"]), trace=trace)
    for i, t in enumerate(trace):
        assert np.array_equal(t["nodes"], fx["L%d_nodes" % i])
        assert np.array_equal(t["old_nodes_new_idx"], fx["L%d_old_nodes_new_idx" % i])
        assert U.edge_multiset_hash(t["edges"]) == str(fx["L%d_edge_hash" % i])
        np.testing.assert_allclose(t["hidden"].numpy(), fx["L%d_hidden" % i], rtol=RTOL, atol=ATOL)
    assert scores.shape == fx["scores"].shape
    np.testing.assert_allclose(scores.numpy(), fx["scores"], rtol=RTOL, atol=ATOL)


def test_temporal_fixture_model_py():
    """Temporal/interpolation/model.py run in the build container vs the oracle's restatement of the shared
    per-edge arithmetic (relative-time embedding, past/now/future linears, attention, scatter-sum)."""
    fx = U.load("temporal_model_py.npz")
    trace = []
    s = orc.temporal_forward(U.params_of(fx), fx["quads"], int(fx["n_ent"]), fx["heads"], fx["rels"], fx["times"],
                             int(fx["cfg"][0]), str(fx["act"]), shared_tables=True, trace=trace)
    for i, t in enumerate(trace):
        np.testing.assert_allclose(t["hidden"].numpy(), torch.nn.functional.leaky_relu(torch.tensor(fx["L%d_agg" % i])).numpy(),
                                   rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(s.numpy(), fx["scores"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(s.numpy() == 0, fx["scores"] == 0)


def test_temporal_train_fixture_model_py():
    """One training step of model.py (batch facts deleted, main.py:70-82 loss): the oracle's scores, loss and parameter
    gradients (autograd through the restatement) against the reference's."""
    fx = U.load("temporal_model_py_train.npz")
    p = {k: torch.tensor(v, requires_grad=True) for k, v in U.params_of(fx).items()}
    quads = np.delete(fx["quads"], fx["example_idx"], axis=0)                  # model.py:45
    s = orc.temporal_forward(p, quads, int(fx["n_ent"]), fx["heads"], fx["rels"], fx["times"], int(fx["cfg"][0]), str(fx["act"]),
                             shared_tables=True)
    np.testing.assert_allclose(s.detach().numpy(), fx["scores"], rtol=RTOL, atol=ATOL)
    loss = torch.nn.functional.nll_loss(torch.log(torch.softmax(s, 1) + 1e-12), torch.tensor(fx["tails"]))
    assert abs(loss.item() - float(fx["loss"])) < 1e-5
    loss.backward()
    for k, v in p.items():
        np.testing.assert_allclose(v.grad.numpy(), fx["grad::" + k], rtol=1e-4, atol=2e-6, err_msg=k)


def test_extrapolation_rank_and_offsets_match_reference_fixture():
    """SURVEY 8 f4 host parts: segment_rank_fil (Temporal/extrapolation/segment.py:346-387) of the product against the output of the
    reference's own function (tests/golden/extrap_rank.npz: ties, unreached targets, filters), and the vectorised
    get_time_offset_list against the oracle's literal restatement of utils.py:692-699 (days without rows keep offset 0)."""
    from red_gnn_amd import extrapolation as X
    fx = U.load("extrap_rank.npz")
    n_q = len(fx["target"])
    sp2o = {(fx["sub"][q], fx["pre"][q]): fx["sp_idx"][fx["sp_ptr"][q]:fx["sp_ptr"][q + 1]] for q in range(n_q)}
    spt2o = {(fx["sub"][q], fx["pre"][q], fx["ts"][q]): fx["spt_idx"][fx["spt_ptr"][q]:fx["spt_ptr"][q + 1]] for q in range(n_q)}
    rank, found, rank_fil, rank_fil_t = X.segment_rank_fil(fx["scores"], fx["entities"], fx["target"], sp2o, spt2o, fx["sub"], fx["pre"], fx["ts"])
    assert np.array_equal(rank, fx["rank"]) and np.array_equal(np.array(found), fx["found"])
    assert np.array_equal(rank_fil, fx["rank_fil"]) and np.array_equal(rank_fil_t, fx["rank_fil_t"])
    rng = np.random.default_rng(3)
    days = np.sort(rng.choice(np.delete(np.arange(60), [0, 7, 8, 31]), 500))            # some days have no rows
    data = np.stack([rng.integers(0, 9, 500), rng.integers(0, 3, 500), rng.integers(0, 9, 500), days * 24 + rng.integers(0, 24, 500)], 1)
    data = data[np.argsort(data[:, 3], kind="stable")]
    assert np.array_equal(X.get_time_offset_list(data, 24), orc.get_time_offset_list(data, 24))
