"""CPU-only checks (-m "not gpu"): the C-ABI library loads and exports every symbol the header declares,
the host logic of the loader mirrors the reference's parsing, the product refuses to run without a GPU,
and the query-sharding collectives work under gloo with world_size 2.  No kernel is launched here."""
import os
import re
import tempfile

import numpy as np
import pytest
import torch

from oracle import redgnn_oracle as orc
from tests import _util as U

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "redgnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from red_gnn_amd import _lib
    lib = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "libredgnn.so does not export %s" % s
    assert sorted(_lib.SYMBOLS) == syms            # the ctypes binding covers exactly the header
    assert lib.rg_version() >= 1
    assert lib.rg_frontier_workspace_bytes(100, 4, 2) > 0
    assert lib.rg_frontier_workspace_bytes(100, 4, 1) == 0      # needs >= 2 levels


def test_argument_errors_are_reported_not_fatal():
    """Error contract of the ABI: non-zero return + rg_last_error(), checked before any device work."""
    import ctypes as C
    from red_gnn_amd import _lib
    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.rg_graph_create(0, 3, None, 0, 1, C.byref(h)) != 0
    assert b"n_ent" in lib.rg_last_error()
    bad = np.array([[0, 0, 99]], dtype=np.int32)
    assert lib.rg_graph_create(10, 3, _lib.ptr(bad), 1, 1, C.byref(h)) != 0
    assert b"out of range" in lib.rg_last_error()
    assert lib.rg_frontier_create(1 << 20, 1 << 12, 2, None, 0, C.byref(h)) != 0     # B*n_ent >= 2^31
    assert b"int32" in lib.rg_last_error()
    assert lib.rg_dense_fwd_supported(64, 5) == 1 and lib.rg_dense_fwd_supported(128, 5) == 1
    assert lib.rg_dense_fwd_supported(96, 5) == 0 and lib.rg_dense_fwd_supported(64, 30) == 0
    # scratch of the dense call: only d = 128 with split products (the weights' split image: 58 blocks of 8 KB + header)
    assert lib.rg_dense_scratch_bytes(64, 0) == 0 and lib.rg_dense_scratch_bytes(64, 1) == 0 and lib.rg_dense_scratch_bytes(128, 0) == 0
    assert lib.rg_dense_scratch_bytes(128, 1) == 256 + 58 * 8192


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_fails_loudly_without_gpu():
    from red_gnn_amd import _lib
    from red_gnn_amd.load_data import DataLoader
    from red_gnn_amd.models import RED_GNN_trans
    ids = U.load("tiny_fwd.npz")
    loader = DataLoader(ids=ids, verbose=False)

    class P:
        n_layer, hidden_dim, attn_dim, n_rel, act, dropout = 2, 16, 3, loader.n_rel, "relu", 0.0

    model = RED_GNN_trans(P, loader)
    with pytest.raises(_lib.NativeError):
        model(ids["subs"], ids["rels"], mode="test")          # CPU parameters: refused, no silent fallback
    with pytest.raises(_lib.NativeError):
        loader.get_neighbors(np.array([[0, 1]]), mode="test")


def test_loader_text_and_ids_agree_and_match_reference_split():
    """Text parsing (load_data.py:11-35,58-67) vs the id arrays the reference's own parser produced."""
    from red_gnn_amd.load_data import DataLoader
    from red_gnn_amd.synthetic import make_synthetic_kg, write_task_dir
    ids = U.load("tiny_fwd.npz")       # ids as parsed by the REFERENCE loader from the same generated text
    kg = make_synthetic_kg(50, 4, 300, seed=7)
    with tempfile.TemporaryDirectory() as td:
        write_task_dir(kg, td)
        a = DataLoader(td, verbose=False)
    b = DataLoader(ids=ids, verbose=False)
    for name in ("fact_triple", "train_triple", "valid_triple", "test_triple", "fact_data", "train_data"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.array_equal(a.fact_triple, ids["facts"]) and np.array_equal(a.test_triple, ids["test"])
    assert (a.n_ent, a.n_rel, a.n_train, a.n_valid, a.n_test) == (b.n_ent, b.n_rel, b.n_train, b.n_valid, b.n_test)
    assert a.filters == b.filters and a.valid_q == b.valid_q and a.test_q == b.test_q
    assert a.n_fact == 2 * len(ids["facts"]) + a.n_ent and a.tn_fact == U.oracle_graph(ids, "test").n_fact
    # the fixture's first four test queries / labels / filters are what get_batch builds
    subs, rels, objs = b.get_batch(np.arange(4), data="test")
    assert np.array_equal(subs, ids["subs"]) and np.array_equal(rels, ids["rels"])
    assert np.array_equal(objs.astype(np.uint8), ids["labels"])
    for i in range(4):
        f = np.zeros(b.n_ent, np.uint8)
        f[b.filters[(int(subs[i]), int(rels[i]))]] = 1
        assert np.array_equal(f, ids["filters"][i])
    # inverse triples (load_data.py:69-74) and the 3:1 re-split (load_data.py:152-164)
    assert np.array_equal(b.train_data, orc.double_triple(ids["train"], b.n_rel))
    np.random.seed(0)
    n_all = len(ids["facts"]) + len(ids["train"])
    b.shuffle_train()
    assert b.n_train == 2 * (n_all - n_all * 3 // 4) and b.n_fact == 2 * (n_all * 3 // 4) + b.n_ent


def test_synthetic_generator_is_deterministic_and_shaped():
    from red_gnn_amd.synthetic import SHAPES, make_synthetic_kg
    a, b = make_synthetic_kg(2000, 11, 9000, seed=1234), make_synthetic_kg(2000, 11, 9000, seed=1234)
    for n in ("facts", "train", "valid", "test"):
        assert np.array_equal(getattr(a, n), getattr(b, n))
    assert len(a.facts) + len(a.train) == 9000 and len(a.facts) == 6750 and len(a.valid) == len(a.test) == 450
    allt = np.concatenate([a.facts, a.train, a.valid, a.test])
    assert len(np.unique(allt, axis=0)) == len(allt) and np.all(allt[:, 0] != allt[:, 2])
    indeg = np.bincount(allt[:, 2], minlength=2000)
    assert indeg.max() > 20 * indeg.mean()                      # hubs, as SURVEY.md §8(d) asks
    assert set(SHAPES) == {"C2", "C3", "C4", "C5", "X"}          # X: C5's shape for the extrapolation setting


def test_shard_helpers():
    from red_gnn_amd.sharding import shard_by_cost, shard_slice
    for n, w in ((10, 3), (8, 8), (5, 8), (1024, 4)):
        sl = [shard_slice(n, w, r) for r in range(w)]
        assert sl[0][0] == 0 and sl[-1][1] == n and all(sl[i][1] == sl[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in sl) - min(b - a for a, b in sl) <= 1
    parts = shard_by_cost([9, 1, 1, 1, 8, 2, 2, 2], 2)
    assert sorted(sum(parts, [])) == list(range(8))
    assert abs(sum([9, 1, 1, 1, 8, 2, 2, 2][i] for i in parts[0]) - 13) <= 1


def test_balanced_cost_sharding_and_costs():
    """bench.py's split for N > 1 (per-query cost varies by far more than 2x): equal counts, every query exactly
    once, the inverse permutation restores query order, hub subjects are spread over the ranks."""
    from red_gnn_amd.sharding import query_costs, shard_balanced, unshard_order
    from red_gnn_amd.synthetic import make_synthetic_kg
    kg = make_synthetic_kg(500, 5, 4000, seed=3)
    base = np.concatenate([kg.facts, kg.train], 0)
    subs = kg.test[:64, 0]
    costs = query_costs(base, kg.n_ent, subs)
    og = orc.OracleGraph(orc.double_triple(base, kg.n_rel), kg.n_ent, kg.n_rel)
    for i in (0, 5, 63):                                         # cost = edges of hop 1 + hop 2 of the oracle's expansion
        nodes = np.array([[0, subs[i]]])
        e = 0
        for _ in range(2):
            nodes, edges, _ = orc.get_neighbors(og, nodes)
            e += len(edges)
        assert costs[i] == e
    for world in (2, 3, 8):
        parts = shard_balanced(costs, world)
        assert sorted(sum(parts, [])) == list(range(64))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
        x = np.arange(64) * 7
        assert np.array_equal(np.concatenate([x[p] for p in parts])[unshard_order(parts)], x)
        loads = [costs[p].sum() for p in parts]
        contiguous = [costs[a:b].sum() for a, b in (__import__("red_gnn_amd.sharding", fromlist=["x"]).shard_slice(64, world, r) for r in range(world))]
        assert max(loads) / min(loads) <= max(contiguous) / min(contiguous) + 1e-9


def _gloo_cost_worker(rank, world, port, ret):
    import torch.distributed as dist
    from red_gnn_amd.sharding import gather_scores, query_costs, shard_balanced, unshard_order
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx = U.load("tiny_fwd.npz")
        g = U.oracle_graph(fx, "test")
        p = U.params_of(fx)
        base = np.concatenate([fx["facts"], fx["train"]], 0)
        parts = shard_balanced(query_costs(base, int(fx["n_ent"]), fx["subs"]), world)      # every rank computes the same split
        mine = np.asarray(parts[rank])
        local = orc.forward(p, g, fx["subs"][mine], fx["rels"][mine], int(fx["cfg"][0]), act=str(fx["act"]))
        full = gather_scores(local, dist)[torch.as_tensor(unshard_order(parts))]            # rank order -> query order
        if rank == 0:
            ret["full"] = full.numpy()
    finally:
        dist.destroy_process_group()


def test_cost_sharded_gather_restores_query_order_gloo():
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gloo_cost_worker, args=(2, port, ret), nprocs=2, join=True)
        fx = U.load("tiny_fwd.npz")
        np.testing.assert_allclose(ret["full"], fx["scores"], rtol=1e-4, atol=1e-5)


def _gloo_worker(rank, world, port, ret):
    import torch.distributed as dist
    from red_gnn_amd.sharding import allreduce_gradients, gather_scores, reduce_metrics, shard_slice
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx = U.load("tiny_fwd.npz")
        g = U.oracle_graph(fx, "test")
        p = U.params_of(fx)
        lo, hi = shard_slice(len(fx["subs"]), world, rank)
        # each rank runs ITS queries only (the oracle stands in for the HIP path on this CPU-only box)
        local = orc.forward(p, g, fx["subs"][lo:hi], fx["rels"][lo:hi], int(fx["cfg"][0]), act=str(fx["act"]))
        full = gather_scores(local, dist)
        ranks = np.array(orc.cal_ranks(local.numpy(), fx["labels"][lo:hi].astype(float), fx["filters"][lo:hi].astype(float)))
        sums = torch.tensor([(1.0 / ranks).sum(), (ranks <= 1).sum(), (ranks <= 10).sum(), float(len(ranks))], dtype=torch.float64)
        tot = reduce_metrics(sums, dist)
        w = torch.nn.Linear(3, 2)
        w.weight.grad = torch.full_like(w.weight, float(rank + 1))
        w.bias.grad = torch.full_like(w.bias, 10.0 * (rank + 1))
        allreduce_gradients(list(w.parameters()), dist)
        if rank == 0:
            ret["full"], ret["tot"] = full.numpy(), tot.numpy()
            ret["gw"], ret["gb"] = w.weight.grad.numpy().copy(), w.bias.grad.numpy().copy()
    finally:
        dist.destroy_process_group()


def test_query_sharding_world_size_2_gloo():
    """N>1 path: queries sharded over 2 ranks, scores all-gathered, metric sums and gradients all-reduced."""
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gloo_worker, args=(2, port, ret), nprocs=2, join=True)
        fx = U.load("tiny_fwd.npz")
        np.testing.assert_allclose(ret["full"], fx["scores"], rtol=1e-4, atol=1e-5)      # sharded == unsharded reference
        ranks = fx["ranks"]
        np.testing.assert_allclose(ret["tot"], [(1.0 / ranks).sum(), (ranks <= 1).sum(), (ranks <= 10).sum(), len(ranks)])
        assert np.all(ret["gw"] == 3.0) and np.all(ret["gb"] == 30.0)


def _gloo_train_worker(rank, world, port, ret):
    import torch.distributed as dist
    from red_gnn_amd.base_model import reference_loss
    from red_gnn_amd.sharding import allreduce_gradients, entity_costs, split_batch
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fx = U.load("tiny_fwd.npz")
        g = U.oracle_graph(fx, "test")
        p = {k: torch.tensor(np.asarray(v)).requires_grad_(v.dtype.kind == 'f') for k, v in U.params_of(fx).items()}
        base = np.concatenate([fx["facts"], fx["train"]], 0)
        subs, rels = fx["subs"], fx["rels"]
        tails = torch.as_tensor(np.asarray([int(np.flatnonzero(l)[0]) for l in fx["labels"]]))
        # the trainer's split of one batch (BaseModel.train_batch): cost-balanced, equal counts, loss scaled to the global batch
        part = split_batch(entity_costs(base, int(fx["n_ent"]))[subs], world, rank)
        sc = orc.forward(p, g, subs[part], rels[part], int(fx["cfg"][0]), act=str(fx["act"]))
        loss = reference_loss(sc, tails[part]) * (len(subs) / len(part))
        loss.backward()
        params = [v for v in p.values()]
        for v in params:
            if v.grad is None:
                v.grad = torch.zeros_like(v)
        allreduce_gradients(params, dist)
        parts = [None] * world
        dist.all_gather_object(parts, part.tolist())
        if rank == 0:
            ret["grads"] = {k: v.grad.numpy().copy() for k, v in p.items()}
            ret["parts"] = parts
    finally:
        dist.destroy_process_group()


def test_trainer_cost_balanced_split_world_size_2_gloo():
    """The trainer's N > 1 step (BaseModel.train_batch): every batch dealt over the ranks by estimated query cost, each rank's loss
    scaled to the global batch (the reference's loss carries the factor n), gradients summed by one flat all-reduce - equal to the
    single-process gradients of the whole batch (the oracle stands in for the HIP path on this CPU-only box)."""
    import torch.multiprocessing as mp
    from red_gnn_amd.base_model import reference_loss
    port = 33500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gloo_train_worker, args=(2, port, ret), nprocs=2, join=True)
        grads, parts = dict(ret["grads"]), list(ret["parts"])
    fx = U.load("tiny_fwd.npz")
    n = len(fx["subs"])
    assert sorted(parts[0] + parts[1]) == list(range(n)) and abs(len(parts[0]) - len(parts[1])) <= 1
    g = U.oracle_graph(fx, "test")
    p = {k: torch.tensor(np.asarray(v)).requires_grad_(v.dtype.kind == 'f') for k, v in U.params_of(fx).items()}
    tails = torch.as_tensor(np.asarray([int(np.flatnonzero(l)[0]) for l in fx["labels"]]))
    reference_loss(orc.forward(p, g, fx["subs"], fx["rels"], int(fx["cfg"][0]), act=str(fx["act"])), tails).backward()
    for k, v in p.items():
        if v.grad is not None:
            np.testing.assert_allclose(grads[k], v.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg=k)


def test_bench_byte_model():
    import bench
    assert bench.algorithmic_bytes(10, 3, 64) == 10 * 272 + 3 * 256
    assert bench.algorithmic_bytes(1, 1, 128) == 528 + 512


def test_bench_roofline_objects():
    """bench.layer_rooflines on round 1's measured launches (C2, B=1024: profiles/r01/final_pmc_B1024.json durations): `roofline` is
    the roof that binds - gathered rows against the L2 gather rate, frac <= 1 - with SURVEY 8d's algorithmic-HBM figure (which may
    exceed 1: its byte model counts L2-served rows as HBM) beside it; stored PMC traffic and L2 request bytes are only attached to
    the workload and kernel version they were measured on."""
    import json
    import bench
    ev = [(0.43, 1.33e6, 1.29e6), (4.03, 52.0e6, 8.2e6), (7.15, 388.5e6, 10.0e6)] * 4
    roof, hops = bench.layer_rooflines(ev, 64, 3)
    assert roof["bound"] == "l2-gather" and roof["unit"] == "GB/s" and roof["launches"] == 12 and roof["traffic"] is None
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0.0 < roof["frac"] <= 1.0
    alg = roof["hbm_algorithmic"]
    assert alg["frac"] > 1.0 and abs(alg["frac"] - alg["achieved"] / alg["peak"]) < 1e-12
    assert abs(alg["bytes_per_launch"] - sum(bench.algorithmic_bytes(e, n, 64) for _, e, n in ev) / 12) < 1.0
    assert "l2_request_bytes" not in roof
    assert [h["hop"] for h in hops] == [0, 1, 2] and all(0.0 < h["l2_gather_frac"] <= 1.0 for h in hops)
    assert hops[1]["l2_gather_frac"] < 0.5 * hops[2]["l2_gather_frac"]            # the expanding hop is the one far below its roof
    assert bench.layer_rooflines([], 64, 3) == (None, None)
    # this round's launches (C2, B=1024): the binding fraction stays below 1 at the fastest launch times measured
    ev3 = [(0.30, 269705, 214497), (1.70, 51608885, 9067818), (6.58, 389993435, 10239990)] * 2
    assert 0.5 < bench.layer_rooflines(ev3, 64, 3)[0]["frac"] <= 1.0
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "t.json")
        with open(path, "w") as f:
            json.dump({"entries": [{"config": "C2", "batch": 1024, "rg_version": 7, "hbm_bytes_per_launch": 4.5e9, "source": "x",
                                    "per_kernel": {"a": {"TCC_HIT_sum": [1.0e9, 2.0e8], "TCC_MISS_sum": [1.0e8]}}}]}, f)
        assert bench.stored_traffic("C2", 1024, 7, path)["hbm_bytes_per_launch"] == 4.5e9
        assert bench.stored_traffic("C4", 1024, 7, path) is None and bench.stored_traffic("C2", 256, 7, path) is None
        assert bench.stored_traffic("C2", 1024, 8, path) is None and bench.stored_traffic("C2", 1024, 7, path + ".missing") is None
        e = bench.stored_traffic("C2", 1024, 7, path)
    roof, _ = bench.layer_rooflines(ev, 64, 3, e)
    assert roof["traffic"] == 4.5e9 and 0.0 < roof["hbm_measured_frac"] < 1.0
    assert roof["l2_request_bytes"] == 1.3e9 * 128 / 3
    # the committed file only ever answers for the library version it was measured with
    from red_gnn_amd import _lib
    with open(bench.TRAFFIC_FILE) as f:
        for e in json.load(f)["entries"]:
            assert {"config", "batch", "rg_version", "hbm_bytes_per_launch", "source"} <= set(e)
    assert _lib.lib().rg_version() >= 2


def test_bench_backward_roofline_object():
    """bench.bwd_rooflines (the --train line): gathered rows of both backward kernels against the L2 gather rate, frac <= 1 at the
    launch times measured in round 2 (C2, B=256: 18.8 ms of backward, ~8 ms of it in rg_layer_bwd)."""
    import bench
    e_lvl = [67000, 12.9e6, 97.5e6]
    rec = [(4.9, 3, 2.27e6), (2.6, 2, 53600), (0.4, 1, 256)]
    r = bench.bwd_rooflines(rec, e_lvl, 64, 8)
    assert r["bound"] == "l2-gather" and 0.0 < r["frac"] <= 1.0 and r["launches"] == 3
    main = sum(e_lvl[l - 1] * (4 * 64 + 16) + n * (4 * 64 + 32) for _, l, n in rec)
    assert abs(r["hbm_algorithmic"]["layer_bwd_kernel_bytes_per_launch"] - main / 3) < 1.0
    assert abs(r["hbm_algorithmic"]["drel_kernel_bytes_per_launch"] - sum(e_lvl) * (4 * 64 + 8 + 32) / 3) < 1.0
    assert bench.bwd_rooflines([], e_lvl, 64, 8) is None


def test_bench_dense_roofline_per_precision():
    """bench.dense_roofline prices the f32-MFMA dense kernel against the f32 matrix pipe, the exact three-term kernel (the default)
    against the f16 matrix pipe by the six MFMAs it issues per product, and the opt-in two-term kernel against HBM by its rows (with
    the issued f16 MFMA rate beside it): launches of C2 / B=1024 as measured in rounds 2 and 3."""
    import bench
    rows = [1.3e6, 8.2e6, 10.0e6]
    f32 = bench.dense_roofline([(0.5, rows[0]), (3.2, rows[1]), (3.9, rows[2])] * 2, 64, 5, 3, 1024, "f32")
    assert f32["bound"] == "mfma" and f32["kernel"] == "dense_kernel" and f32["peak"] == 157.3 and 0.5 < f32["frac"] < 1.0
    sp = bench.dense_roofline([(0.2, rows[0]), (1.0, rows[1]), (1.22, rows[2])] * 2, 64, 5, 3, 1024, "f16x2")
    assert sp["bound"] == "hbm" and sp["kernel"] == "dense_split_kernel" and sp["unit"] == "GB/s" and 0.0 < sp["frac"] < 1.0
    per_launch = (sum(rows) * (8 * 64 + 4 + 32) + (1024 + rows[0] + rows[1]) * 256) / 3
    assert abs(sp["algorithmic_bytes_per_launch"] - per_launch) < 1e-6 * per_launch
    assert abs(sp["mfma"]["issued_f16_tflops"] - 3 * sp["useful_tflops"]) < 1e-9 and sp["mfma"]["frac"] < 1.0
    assert sp["mfma"]["useful_over_f32_mfma_peak"] > 1.0          # more useful flops per second than the f32 pipe could issue
    s3 = bench.dense_roofline([(0.3, rows[0]), (2.4, rows[1]), (2.9, rows[2])] * 2, 64, 5, 3, 1024, "f16x3")
    assert s3["bound"] == "mfma" and s3["kernel"] == "dense_split3_kernel" and s3["peak"] == 2500.0 and 0.0 < s3["frac"] < 1.0
    assert abs(s3["achieved"] - 6 * s3["useful_tflops"]) < 1e-9 and 0.0 < s3["hbm_algorithmic_frac"] < 1.0
    assert bench.dense_kernel_of(128, "f16x3") == ("dense128_split3_kernel", "f16x3") and bench.dense_kernel_of(48, "f16x3")[0] == "dense_split3_kernel"
    assert bench.DTYPE_OF["f16x3"] == "f32" and bench.DTYPE_OF["f16x2"] != "f32"


def test_loader_id_cache_roundtrip():
    """Binary id cache of the parsed text: second construction reads the cache, same loader state."""
    from red_gnn_amd.load_data import DataLoader
    from red_gnn_amd.synthetic import make_synthetic_kg, write_task_dir
    kg = make_synthetic_kg(40, 3, 200, seed=2)
    with tempfile.TemporaryDirectory() as td:
        task = os.path.join(td, "toy")
        write_task_dir(kg, task)
        a = DataLoader(task, verbose=False, cache_dir=os.path.join(td, "cache"))
        assert os.path.exists(os.path.join(td, "cache", "toy_ids.npz"))
        b = DataLoader(task, verbose=False, cache_dir=os.path.join(td, "cache"))
        c = DataLoader(task, verbose=False)
    for x in (a, b):
        assert x._filters is None                                   # binary path: no per-triple Python loop ran ...
        assert x.valid_q == c.valid_q and x.test_q == c.test_q
        assert all(np.array_equal(p, q) for p, q in zip(x.test_a, c.test_a)) and all(np.array_equal(p, q) for p, q in zip(x.valid_a, c.valid_a))
        assert np.array_equal(x.fact_triple, c.fact_triple) and np.array_equal(x.train_data, c.train_data)
        fptr, fidx = x._csr_host["test"]                            # ... the filter sets of the queries come as CSR lists
        for i, (h, r) in enumerate(c.test_q):
            assert list(fidx[fptr[i]:fptr[i + 1]]) == c.filters[(h, r)]
        assert x.filters == c.filters                               # ... and the dict is rebuilt when asked for


def test_attention_width_padding():
    """Kernel attention widths: multiples of 4 up to 16, then 32; wider is refused with a clear error (the reference accepts any)."""
    from red_gnn_amd.models import pad_attn
    assert [pad_attn(a) for a in (1, 3, 4, 5, 10, 16, 17, 28, 30, 32)] == [4, 4, 4, 8, 12, 16, 32, 32, 32, 32]
    with pytest.raises(ValueError):
        pad_attn(33)
