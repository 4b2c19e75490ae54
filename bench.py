"""bench.py — edges aggregated/sec + MRR-eval queries/sec of the RED-GNN hot path on MI355X.

A *step* is one evaluation pass of the hot path over one batch of synthetic queries:
frontier expansion + n_layer fused message-passing layers + GRU/readout + filtered ranking
(RED_GNN_trans.forward + cal_ranks of the reference).  Workload = BASELINE.json configs[1]:
synthetic KG 10k entities / 50 relations / 200k triples (seed 1234), n_layer=3, hidden_dim=64,
attn_dim=5, batch of B queries per GPU.  Queries are sharded over ranks by estimated cost (weak scaling:
B per GPU fixed); for N>1 scores are all-gathered over RCCL and the 4 metric sums all-reduced.
After the timed region (never inside it) rank 0 also runs: the second half of the metric's name —
BaseModel.evaluate on the real family graph (configs[0] shape, n_tbatch=50) — and the CPU baseline,
whose oracle scores double as an end-to-end parity check at bench size.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline] [--no-family-eval]

    python bench.py --train [--config C2] [--batch 256]        one training step (forward + loss + HIP backward + Adam) per step
    python bench.py --global-batch G --gpus N                  strong scaling: G queries dealt over the N ranks (BASELINE's C4 256 / 4)

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant
kernel (the layer_fwd kernels, HIP events on their stream: the L2-gather roof that binds them, SURVEY 8d's
algorithmic-HBM figure beside it as `hbm_algorithmic`), `per_hop`, `roofline_dense`, `cpu_baseline` (the oracle on
host cores) and `family_eval` (configs[0]: real family graph, queries/s, edges/s and its own CPU leg).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3e12 achievable)
L2_GATHER_PEAK = 18.8e12   # B/s, same guide, "Indexed rows": rows shared by every workgroup served by the XCDs' L2s, 16.8-18.8 TB/s chip-wide
MFMA_F32_PEAK = 157.3e12   # FLOP/s, same guide: f32-input MFMA = the FP32 vector rate
MFMA_F16_PEAK = 2.5e15     # FLOP/s, same guide: dense f16 / bf16 MFMA
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_layer_fwd.json")

# `dtype` of the line = the arithmetic the path computes in (storage, sums and - for f32 / f16x3 - operands and products are fp32)
DTYPE_OF = {"f32": "f32", "f16x3": "f32", "f16x2": "f32 storage and accumulation, dense products on 22-bit operands (f16x2 splits)"}
DENSE_NOTE = {
    "f32": "f32: dense products on v_mfma_f32_16x16x4_f32",
    "f16x3": "f16x3: fp32 arithmetic on the f16 matrix pipe - every fp32 operand carried EXACTLY as hi + mid + lo f16 (33 >= 24 bits; "
             "bit-exact reconstruction and product tests in tests/test_gpu_parity.py), the six partial products of order >= 2^-22 per "
             "product (error <= 2^-31), fp32 accumulation; dense_f32 = the same step on the f32 MFMA forms",
    "f16x2": "f16x2 (opt-in): fp32 operands as two-term f16 splits (22 bits), fp32 accumulation; narrower than the reference's fp32 "
             "nn.Linear / nn.GRU; dense_f32 = the same step on the f32 MFMA forms",
}


class Params:
    def __init__(self, shape, n_rel):
        self.n_layer, self.hidden_dim, self.attn_dim = shape["n_layer"], shape["hidden_dim"], shape["attn_dim"]
        self.n_rel, self.act, self.dropout = n_rel, "relu", 0.0


def algorithmic_bytes(n_edges, n_nodes, d):
    """per edge one fp32 source row (4d B) + four int32 indices (16 B); per output node one fp32 row."""
    return n_edges * (4 * d + 16) + n_nodes * 4 * d


def dense_kernel_of(d, precision):
    """(kernel name, arithmetic) rg_dense_fwd runs for this width and precision (csrc/dense.hip dispatch)."""
    if precision == "f16x2":
        return ("dense_split_kernel" if d <= 64 else "dense128_split_kernel"), "f16x2"
    if precision == "f16x3":
        return ("dense_split3_kernel" if d <= 64 else "dense128_split3_kernel"), "f16x3"
    return ("dense_kernel" if d <= 64 else "dense128_kernel"), "f32"


def dense_roofline(dense_ms, d, attn_dim, n_layer, batch, precision):
    """Second kernel of the step (W_h + act + GRU + projections + readout, rg_dense_fwd), from its launches' HIP events.
    Useful flops per node row (what the reference computes): 2 d d (1 + 3 + 3) for W_h, weight_ih, weight_hh + 2 d attn_dim for the
    hoisted Ws_attn.  Algorithmic bytes per launch: every row's agg in and new state out (8 d), prev_idx (4), a_s out (4 attn_dim
    padded to 4 floats) and one old-state row per node of the previous level (4 d).
    "f32": products on v_mfma_f32_16x16x4_f32 - the f32 matrix pipe bounds the kernel, peak 157.3 TFLOP/s.
    "f16x3" (d <= 64): fp32 arithmetic as exact three-term f16 splits - six f16 / bf8 MFMAs per product: the f16 matrix pipe is issued
    6x the useful flops and is the roof that binds (2.5 PFLOP/s dense); the HBM fraction by the kernel's rows is kept beside it.
    "f16x2" (opt-in, 22-bit operands): three MFMAs per product; priced against HBM by its rows, with the issued f16 rate beside it."""
    ms = sum(m for m, _ in dense_ms)
    rows = sum(n for _, n in dense_ms)
    flops = rows * (2.0 * d * d * 7 + 2.0 * d * attn_dim)
    ap = (attn_dim + 3) // 4 * 4
    nbytes = 0.0
    for i, (_, n) in enumerate(dense_ms):
        n_old = batch if i % n_layer == 0 else dense_ms[i - 1][1]
        nbytes += n * (8.0 * d + 4 + 4 * ap) + n_old * 4.0 * d
    kernel, arith = dense_kernel_of(d, precision)
    t = ms * 1e-3
    common = dict(launches=len(dense_ms), avg_launch_ms=ms / len(dense_ms), algorithmic_flops_per_launch=flops / len(dense_ms),
                  algorithmic_bytes_per_launch=nbytes / len(dense_ms), useful_tflops=flops / t / 1e12, precision=arith,
                  hbm_algorithmic_frac=nbytes / t / HBM_PEAK, traffic=None)
    if arith == "f32":
        return dict(bound="mfma", kernel=kernel, achieved=flops / t / 1e12, peak=MFMA_F32_PEAK / 1e12, unit="TFLOP/s",
                    frac=flops / t / MFMA_F32_PEAK, **common)
    n_mfma = 6 if arith == "f16x3" else 3
    mfma = dict(issued_f16_tflops=n_mfma * flops / t / 1e12, peak=MFMA_F16_PEAK / 1e12, frac=n_mfma * flops / t / MFMA_F16_PEAK,
                useful_over_f32_mfma_peak=flops / t / MFMA_F32_PEAK)
    if arith == "f16x3":
        return dict(bound="mfma", kernel=kernel, achieved=mfma["issued_f16_tflops"], peak=mfma["peak"], unit="TFLOP/s", frac=mfma["frac"],
                    mfma=mfma, **common)
    return dict(bound="hbm", kernel=kernel, achieved=nbytes / t / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=nbytes / t / HBM_PEAK,
                mfma=mfma, **common)


def stored_traffic(config, batch, version, path=TRAFFIC_FILE):
    """HBM bytes per layer_fwd launch from the committed PMC passes (tools/pmc_traffic.sh), or None unless an entry matches
    this exact workload AND kernel version (rg_version): a profile of another shape or of older kernels is not evidence."""
    try:
        with open(path) as f:
            entries = json.load(f).get("entries", [])
    except (OSError, ValueError):
        return None
    for e in entries:
        if e.get("config") == config and int(e.get("batch", -1)) == int(batch) and int(e.get("rg_version", -1)) == int(version):
            return e
    return None


def layer_rooflines(events_ms, d, n_layer, traffic_entry=None):
    """The roofline object of the message-passing kernel from per-launch records (ms, n_edges, n_nodes), launch i being hop
    i % n_layer.  Pure arithmetic (tested on the host).

    roofline     the roof that binds: a query's hidden slab is resident in its XCD's L2, so the gathered source rows (E * 4d bytes)
                 over the measured launch time are priced against the L2 gather rate (MI355X_MICROARCH.md, "Indexed rows": 18.8 TB/s
                 chip-wide); frac <= 1.  Beside it:
      hbm_algorithmic   SURVEY 8d's figure: ALGORITHMIC bytes (every gathered row counted as if it came from HBM) over the same time
                        against the HBM peak - above 1 whenever the rows are served on chip;
      traffic           HBM bytes per launch measured by the PMC passes (or null), hbm_measured_frac = that over time and HBM peak;
      l2_request_bytes  (TCC_HIT + TCC_MISS) x 128 B per launch from the same passes: proof that the rows are touched.
    per_hop      E, N, ms, both fractions per hop (expanding hops sit far below saturated ones)."""
    n = len(events_ms)
    if n == 0:
        return None, None
    tot_ms = sum(ms for ms, _, _ in events_ms)
    t = tot_ms * 1e-3
    k_bytes = sum(algorithmic_bytes(ne, nn, d) for _, ne, nn in events_ms)
    k_edges = sum(ne for _, ne, _ in events_ms)
    row_bytes = k_edges * 4.0 * d
    traffic = traffic_entry["hbm_bytes_per_launch"] if traffic_entry else None
    roof = dict(bound="l2-gather", kernel="layer_fwd kernels (walk + word-parallel)", achieved=row_bytes / t / 1e9, peak=L2_GATHER_PEAK / 1e9,
                unit="GB/s", frac=row_bytes / t / L2_GATHER_PEAK, traffic=traffic, launches=n, avg_launch_ms=tot_ms / n,
                bytes="gathered source rows: E * 4 * d per launch", row_bytes_per_launch=row_bytes / n,
                kernel_edges_per_s=k_edges / t,
                hbm_algorithmic=dict(achieved=k_bytes / t / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=k_bytes / t / HBM_PEAK,
                                     bytes_per_launch=k_bytes / n,
                                     note="SURVEY 8d: E (4d + 16) + N 4d, every gathered row counted as HBM; a query's rows are "
                                          "L2-resident, so this exceeds the HBM peak on saturated hops"))
    if traffic is not None:
        roof["hbm_measured_frac"] = traffic / (tot_ms / n * 1e-3) / HBM_PEAK
        roof["traffic_source"] = traffic_entry.get("source")
        req = sum(sum(k.get("TCC_HIT_sum", [])) + sum(k.get("TCC_MISS_sum", [])) for k in traffic_entry.get("per_kernel", {}).values())
        if req:
            roof["l2_request_bytes"] = req * 128.0 / n_layer
    hops = []
    for h in range(n_layer):
        sel = events_ms[h::n_layer]
        if not sel:
            continue
        ms = sum(m for m, _, _ in sel) / len(sel)
        ne = sum(e for _, e, _ in sel) / len(sel)
        nn = sum(x for _, _, x in sel) / len(sel)
        hops.append(dict(hop=h, edges=ne, nodes=nn, ms=ms, edges_per_s=ne / (ms * 1e-3),
                         hbm_algorithmic_frac=algorithmic_bytes(ne, nn, d) / (ms * 1e-3) / HBM_PEAK,
                         l2_gather_frac=ne * 4.0 * d / (ms * 1e-3) / L2_GATHER_PEAK))
    return roof, hops


def cpu_baseline(kg, shape, state, subs, rels, ans, filt, gpu_scores, gpu_ranks, ans_ptr, budget_s=20.0, bs=50):
    """The oracle (CPU restatement, validated against the reference) on the host cores: same KG, same weights, the first
    queries of the same batch, in batches of 50 as the reference evaluates (train.py:46-56 n_tbatch) until ~budget_s.
    Its scores are compared with the GPU's for the very same queries: an end-to-end parity check at bench size, for free."""
    from oracle import redgnn_oracle as orc
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(n_cpu, 64)))       # the cores this process may actually run on
    og = orc.OracleGraph(np.concatenate([orc.double_triple(kg.facts, kg.n_rel), orc.double_triple(kg.train, kg.n_rel)], 0),
                         kg.n_ent, kg.n_rel)
    p = {k: v.detach().cpu() for k, v in state.items()}
    bs = min(bs, len(subs))
    done, edges, t_tot = 0, 0, 0.0
    scores_ok, max_err, rank_kernel_ok, n_rank, n_rank_same, n_viol, max_abs, f64 = True, 0.0, True, 0, 0, 0, 0.0, None
    while done + bs <= len(subs) and (t_tot < budget_s or done == 0):
        sl = slice(done, done + bs)
        t0 = time.perf_counter()
        trace = []
        sc = orc.forward(p, og, subs[sl], rels[sl], shape["n_layer"], act="relu", trace=trace).numpy()
        labels = np.zeros((bs, kg.n_ent)); fl = np.zeros((bs, kg.n_ent))
        for i in range(bs):
            labels[i, ans[done + i]] = 1
            fl[i, filt[done + i]] = 1
        cpu_ranks = np.asarray(orc.cal_ranks(sc, labels, fl))
        t_tot += time.perf_counter() - t0
        edges += sum(len(t["edges"]) for t in trace)
        # ---- parity at bench size (outside every timed region) ----
        g = gpu_scores[sl]
        viol = np.abs(g - sc) > 2e-5 + 1e-4 * np.abs(sc)
        n_viol += int(viol.sum())
        scores_ok &= bool(np.array_equal(g == 0, sc == 0))
        max_err = max(max_err, float(np.max(np.abs(g - sc))))
        max_abs = max(max_abs, float(np.max(np.abs(sc))))
        if done == 0:
            # the reference's fp32 evaluation is itself only so close to the exact value (hub destinations sum thousands of fp32
            # terms in an unspecified order): the same batch in fp64 says how far each fp32 path is from it
            s64 = orc.forward(p, og, subs[sl], rels[sl], shape["n_layer"], act="relu", dtype=torch.float64).numpy()
            f64 = dict(gpu_max_err_vs_fp64=float(np.max(np.abs(g - s64))), cpu_fp32_max_err_vs_fp64=float(np.max(np.abs(sc - s64))),
                       gpu_violations_vs_fp64=int(np.sum(np.abs(g - s64) > 2e-5 + 1e-4 * np.abs(s64))), queries=bs)
        g_ranks = gpu_ranks[ans_ptr[done]:ans_ptr[done + bs]]
        rank_kernel_ok &= bool(np.array_equal(g_ranks, np.asarray(orc.cal_ranks(g, labels, fl))))     # rg_rank on the GPU's own scores
        n_rank += len(cpu_ranks)
        n_rank_same += int(np.sum(g_ranks == cpu_ranks))                                               # vs the CPU path end to end
        done += bs
    return dict(value=edges / t_tot, unit="edges/s", cores=torch.get_num_threads(), kind="port",
                sample="%d of the batch's queries (batches of %d, the reference's n_tbatch), same KG and weights, forward + cal_ranks, "
                       "%.1f s; %.1f queries/s" % (done, bs, t_tot, done / t_tot),
                queries_per_s=done / t_tot), \
        {"parity_at_bench_size": bool(scores_ok and rank_kernel_ok and
                                      (n_viol == 0 or f64["gpu_max_err_vs_fp64"] <= max(4.0 * f64["cpu_fp32_max_err_vs_fp64"], 2e-5))),
         "criterion": "zero pattern equal; every score within rtol 1e-4 / atol 2e-5 of the CPU path, or the GPU's max error against the "
                      "fp64 evaluation of the same batch <= 4x the CPU fp32 path's own; rg_rank on the GPU's scores == the oracle's ranking of them",
         "queries_checked": done, "zero_pattern_equal": bool(scores_ok), "rtol": 1e-4, "atol": 2e-5,
         "elements_beyond_rtol_atol_of_cpu_fp32": n_viol, "elements": done * kg.n_ent, "max_abs_score_err": max_err, "max_abs_score": max_abs,
         "fp64_check_first_batch": f64,
         "rank_kernel_equals_oracle_ranking_of_gpu_scores": bool(rank_kernel_ok),
         "ranks_identical_to_cpu_path": "%d of %d" % (n_rank_same, n_rank)}


def family_eval(dist, world, engine, cpu_leg=True, rank=0):
    """BASELINE configs[0]: BaseModel.evaluate (forward + filtered ranking, valid + test) on the real family graph with the reference's
    shape (n_layer=3, hidden_dim=64, n_tbatch=50) and random-init weights: queries per second and aggregated edges per second over all
    ranks (evaluation batches are dealt round-robin), and - on rank 0 at N = 1 - the CPU leg of the same configuration: the oracle on
    the first two test batches (forward + cal_ranks) with score and rank parity against the GPU on those 100 queries.  The id triples
    are the committed fixture of the reference's data/family."""
    path = os.path.join(ROOT, "tests", "golden", "family_ids.npz")
    if not os.path.exists(path):
        return None
    from red_gnn_amd.base_model import BaseModel, _chunks
    from red_gnn_amd.load_data import DataLoader
    saved = engine.KERNEL_EVENTS, engine.DENSE_EVENTS
    engine.KERNEL_EVENTS = engine.DENSE_EVENTS = None          # graph replay path
    try:
        ids = dict(np.load(path))
        loader = DataLoader(ids=ids, verbose=False)
        nq = loader.n_valid + loader.n_test

        def build(n_tb, coalesce=1):
            class Opt:
                lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, n_tb
                n_rel = loader.n_rel
                eval_coalesce = coalesce

            torch.manual_seed(1234)
            return BaseModel(Opt, loader, dist=dist if world > 1 else None)

        def measure(bm, reps):
            for _ in range(3):                                      # the third pass of a batch shape captures its graph
                bm.evaluate()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                mrr, _ = bm.evaluate()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dt], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t[0])
            return nq * reps / dt, dt / reps, float(mrr)

        bm = build(50)
        qps, per_pass, mrr = measure(bm, 5)
        # edges of one pass (every rank's batches; untimed: one size read-back per batch)
        edges = 0
        bm.model.eval()
        with torch.no_grad():
            for data, n_data in (("valid", loader.n_valid), ("test", loader.n_test)):
                for idx in _chunks(n_data, 50)[bm.rank::bm.world]:
                    subs, rels = loader.get_batch_csr(idx, data=data)[:2]
                    bm.model(subs, rels, mode=data)
                    edges += sum(bm.model.last_stats["n_edges"])
        if dist is not None:
            t = torch.tensor([float(edges)], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            edges = float(t[0])
        out = dict(queries_per_s=qps, edges_per_s=edges / per_pass, edges_per_pass=float(edges), queries=nq, n_tbatch=50, hidden_dim=64, n_layer=3,
                   seconds_per_pass=per_pass, valid_mrr_of_random_init=mrr,
                   path="HIP graph replay per batch, batches dealt to concurrent streams", dense_precision=bm.model.dense_precision)
        if cpu_leg and world == 1 and rank == 0:
            out["cpu_baseline"], out["parity"] = family_cpu_leg(bm, loader, ids)
            out["speedup_vs_cpu_edges_per_s"] = out["edges_per_s"] / out["cpu_baseline"]["value"]
        # the same evaluation with ten reference batches fused per forward pass (BaseModel.eval_coalesce: a query's scores do not depend
        # on the rest of its batch - same scores, same ranks; not what the reference does per pass, so it is not the headline figure)
        qps_500, _, mrr_500 = measure(build(50, coalesce=10), 5)
        out.update(queries_per_s_at_n_tbatch_500=qps_500, same_mrr_at_n_tbatch_500=abs(mrr - mrr_500) < 1e-9,
                   queries_per_s_eval_coalesce_10=qps_500)
        return out
    finally:
        engine.KERNEL_EVENTS, engine.DENSE_EVENTS = saved


def family_cpu_leg(bm, loader, ids, n_batches=2, bs=50):
    """The oracle on the family evaluation graph: the first `n_batches` test batches of 50 (forward + cal_ranks, host cores), and the
    GPU's scores / ranks of the same queries against it (rtol 1e-4 / atol 2e-5 on scores, ranks equal)."""
    from oracle import redgnn_oracle as orc
    from red_gnn_amd.utils import cal_ranks_csr
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(n_cpu, 64)))
    n_ent, n_rel = int(ids["n_ent"]), int(ids["n_rel"])
    og = orc.OracleGraph(np.concatenate([orc.double_triple(ids["facts"], n_rel), orc.double_triple(ids["train"], n_rel)], 0), n_ent, n_rel)
    p = {k: v.detach().cpu() for k, v in bm.model.state_dict().items()}
    t_tot, edges, n_rank, n_same, n_viol, max_err, zero_ok, done = 0.0, 0, 0, 0, 0, 0.0, True, 0
    bm.model.eval()
    for b in range(n_batches):
        idx = np.arange(b * bs, min((b + 1) * bs, loader.n_test))
        if len(idx) == 0:
            break
        subs, rels, a_ptr, a_idx, f_ptr, f_idx = loader.get_batch_csr(idx, data="test")
        with torch.no_grad():
            g = bm.model(subs, rels, mode="test")
            g_ranks = cal_ranks_csr(g, a_ptr, a_idx, f_ptr, f_idx).double().cpu().numpy()
        g = g.cpu().numpy()
        t0 = time.perf_counter()
        trace = []
        sc = orc.forward(p, og, np.asarray(subs), np.asarray(rels), 3, act="relu", trace=trace).numpy()
        labels = np.zeros((len(idx), n_ent)); fl = np.zeros((len(idx), n_ent))
        for i, q in enumerate(idx):
            labels[i, np.asarray(loader.test_a[q])] = 1
            fl[i, np.asarray(loader.filters[(int(subs[i]), int(rels[i]))])] = 1
        cpu_ranks = np.asarray(orc.cal_ranks(sc, labels, fl))
        t_tot += time.perf_counter() - t0
        edges += sum(len(t["edges"]) for t in trace)
        n_viol += int((np.abs(g - sc) > 2e-5 + 1e-4 * np.abs(sc)).sum())
        max_err = max(max_err, float(np.abs(g - sc).max()))
        zero_ok &= bool(np.array_equal(g == 0, sc == 0))
        n_rank += len(cpu_ranks)
        n_same += int(np.sum(g_ranks == cpu_ranks))
        done += len(idx)
    cpu = dict(value=edges / t_tot, unit="edges/s", cores=torch.get_num_threads(), kind="port", queries_per_s=done / t_tot,
               sample="family evaluation graph, first %d test queries in batches of %d, oracle forward + cal_ranks, %.1f s" % (done, bs, t_tot))
    parity = dict(queries_checked=done, zero_pattern_equal=bool(zero_ok), rtol=1e-4, atol=2e-5, elements_beyond_rtol_atol=n_viol,
                  max_abs_score_err=max_err, ranks_identical_to_cpu_path="%d of %d" % (n_same, n_rank),
                  parity=bool(zero_ok and n_viol == 0 and n_same == n_rank))
    return cpu, parity


def bench_temporal(args, dist, world, rank, engine, scaling="weak"):
    """--config C5: BASELINE configs[4], the temporal interpolation path (T_RED_GNN, model_cuda.py layout) on the ICEWS14-shaped synthetic:
    a step = forward of B (head, relation, time) queries (expansion + 5 fused temporal layers + readout), queries sharded over ranks.
    The CPU leg is the oracle's temporal_forward on a few of the same queries (scores compared at the fp64-anchored tolerance)."""
    from red_gnn_amd.synthetic import SHAPES, make_temporal_shape
    from red_gnn_amd.temporal import T_RED_GNN
    sh, tkg = SHAPES["C5"], make_temporal_shape("C5", seed=1234)
    d, n_layer = sh["hidden_dim"], sh["n_layer"]

    class TP:
        pass

    p = TP()
    p.n_rel, p.n_ent, p.n_time = tkg.n_rel, tkg.n_ent, tkg.n_time
    p.hidden_dim, p.attn_dim, p.n_layer, p.act, p.graph, p.device = d, sh["attn_dim"], n_layer, "tanh", tkg.quads, "cuda"
    torch.manual_seed(1234)
    model = T_RED_GNN(p, shared_tables=False).cuda().eval()
    B = args.batch
    rows = tkg.quads[(np.arange(B * world) % tkg.n_base)[rank::world]]
    batch = {"head": rows[:, 0], "relation": rows[:, 1], "time": rows[:, 3]}
    kernel_events = []
    if not args.no_kernel_events:
        engine.KERNEL_EVENTS = kernel_events

    def step():
        with torch.no_grad():
            return model(batch, mode="test"), model.last_stats

    for _ in range(args.warmup):
        step()
    kernel_events.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for _ in range(args.steps):
        scores, st = step()
        edges += sum(st["n_edges"])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(edges)], device="cuda", dtype=torch.float64)
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, edges = float(tmax[0]), float(tsum[1])
    if rank != 0:
        return
    roof = None
    ev_ms = [(e0.elapsed_time(e1), ne, nn) for (e0, e1, ne, nn) in kernel_events]
    if ev_ms:
        # SURVEY 8(d), temporal: per edge one fp32 row + five int32 (head, rel, tail, time, query time): E (4d + 20) + N 4d
        ms = sum(m for m, _, _ in ev_ms)
        nbytes = sum(ne * (4 * d + 20) + nn * 4 * d for _, ne, nn in ev_ms)
        roof = dict(bound="hbm", kernel="layer_fwd_kernel<TEMPORAL>", achieved=nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                    frac=nbytes / (ms * 1e-3) / HBM_PEAK, traffic=None, launches=len(ev_ms), avg_launch_ms=ms / len(ev_ms),
                    algorithmic_bytes_per_launch=nbytes / len(ev_ms),
                    kernel_edges_per_s=sum(ne for _, ne, _ in ev_ms) / (ms * 1e-3),
                    note="the kernel gathers three rows per edge (direction-projected state, relation and time rows) and a 32-wide attention "
                         "row; SURVEY's byte model counts one")
    cpu = parity = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import redgnn_oracle as orc
        n_s = 4
        pd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        hb, rb, tb = (batch[k][:n_s] for k in ("head", "relation", "time"))
        t1 = time.perf_counter()
        tr = []
        ref = orc.temporal_forward(pd, tkg.quads, tkg.n_ent, hb, rb, tb, n_layer, "tanh", trace=tr).numpy()
        t_cpu = time.perf_counter() - t1
        ref64 = orc.temporal_forward(pd, tkg.quads, tkg.n_ent, hb, rb, tb, n_layer, "tanh", dtype=torch.float64).numpy()
        got = scores[:n_s].cpu().numpy()
        e_cpu = sum(t["n_edges"] for t in tr)
        err_gpu, err_cpu = float(np.abs(got - ref64).max()), float(np.abs(ref - ref64).max())
        tol_ok = bool(np.all(np.abs(got - ref) <= 2e-5 + 1e-4 * np.abs(ref)) or err_gpu <= 4 * max(err_cpu, 1e-7))
        parity = dict(parity_at_bench_size=bool(tol_ok and np.array_equal(got == 0, ref == 0)), queries_checked=n_s,
                      gpu_max_err_vs_fp64=err_gpu, cpu_fp32_max_err_vs_fp64=err_cpu,
                      note="model_cuda.py layout: checked against the oracle only (parity unpinned, DESIGN 2)")
        cpu = dict(value=e_cpu / t_cpu, unit="edges/s", cores=torch.get_num_threads(), kind="port",
                   sample="%d of the batch's queries, oracle temporal_forward, %.1f s" % (n_s, t_cpu))
    out = {
        "metric": "edges aggregated/sec + MRR-eval queries/sec, family KG n_layer=3 at 1/2/4/8 GPU",
        "value": edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "C5 ICEWS14-shaped temporal synthetic %d entities / %d relation rows / %d time ids / %d quadruples (seed 1234), "
                               "interpolation path n_layer=%d hidden_dim=%d attn_dim=%d, step = expansion + fused temporal layers + readout, "
                               "%d queries per GPU" % (tkg.n_ent, tkg.n_rel, tkg.n_time, sh["n_triples"], n_layer, d, sh["attn_dim"], B),
                   "batch_per_gpu": B, "global_batch": B * world, "sharding": "queries strided over ranks" if world > 1 else "none"},
        "eval_queries_per_s": B * world * args.steps / dt, "edges_per_step": edges / args.steps,
        "roofline": roof, "cpu_baseline": cpu, "parity": parity, "rg_version": int(_lib_version()),
    }
    print(json.dumps(out))


def bench_extrapolation(args, dist, world, rank, engine, scaling="weak"):
    """--config X: SURVEY 8 f4, the temporal EXTRAPOLATION path (Temporal/extrapolation/model_cuda_new_embedding.py) on the ICEWS14-shaped
    synthetic: a step = forward of B (subject, relation, time) queries late in the year (full 120-day windows): windowed expansion +
    3 fused layers + classifier + per-query softmax.  With --train the step also runs the loss of main.py:303-308 and the backward
    (rg_xlayer_bwd).  CPU leg: the oracle's extrap_forward on two of the queries (parity UNPINNED for this model file)."""
    from red_gnn_amd import extrapolation as X
    from red_gnn_amd.synthetic import SHAPES, make_extrapolation_shape
    import torch.nn.functional as F
    sh = SHAPES["X"]
    data, n_ent, n_rel, gran = make_extrapolation_shape("X", seed=1234)
    d, n_layer = sh["hidden_dim"], sh["n_layer"]

    class P:
        pass

    p = P()
    p.n_ent, p.n_rel, p.data, p.time_granularity, p.hidden_dim, p.attn_dim, p.n_layer, p.act, p.device = n_ent, n_rel, data, gran, d, sh["attn_dim"], n_layer, "relu", "cuda"
    torch.manual_seed(1234)
    model = X.T_RED_GNN(p).cuda()
    model.train(args.train)
    B = args.batch
    late = np.flatnonzero(data[:, 3] // gran >= 200)
    q = data[np.sort(np.random.default_rng(1234).choice(late, B * world, replace=False))[rank::world]]

    class Q:
        src_idx, rel_idx, ts = q[:, 0], q[:, 1], q[:, 3]

    target = torch.as_tensor(q[:, 2], dtype=torch.long).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3) if args.train else None
    kernel_events = []
    if not args.no_kernel_events:
        engine.KERNEL_EVENTS = kernel_events

    def step():
        if args.train:
            opt.zero_grad(set_to_none=True)
            score, _ = model(Q)
            F.nll_loss(torch.log(F.softmax(score, dim=1) + 1e-12), target).backward()
            if dist is not None:
                from red_gnn_amd.sharding import allreduce_gradients
                allreduce_gradients([x for x in model.parameters()], dist)
            opt.step()
        else:
            with torch.no_grad():
                score, _ = model(Q)
        return score, model.last_stats

    for _ in range(max(args.warmup, 1)):
        step()
    kernel_events.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for _ in range(args.steps):
        scores, st = step()
        edges += sum(st["n_edges"])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(edges)], device="cuda", dtype=torch.float64)
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, edges = float(tmax[0]), float(tsum[1])
    engine.KERNEL_EVENTS = None
    if rank != 0:
        return
    roof = None
    ev_ms = [(e0.elapsed_time(e1), ne, nn) for (e0, e1, ne, nn) in kernel_events]
    if ev_ms:
        ms = sum(m for m, _, _ in ev_ms)
        nbytes = sum(ne * (4 * d + 20) + nn * 4 * d for _, ne, nn in ev_ms)
        rows = sum(ne for _, ne, _ in ev_ms) * 4.0 * d
        roof = dict(bound="l2-gather", kernel="layer_fwd_kernel<TEMPORAL> with row windows (rg_xlayer_fwd)", achieved=rows / (ms * 1e-3) / 1e9,
                    peak=L2_GATHER_PEAK / 1e9, unit="GB/s", frac=rows / (ms * 1e-3) / L2_GATHER_PEAK, traffic=None, launches=len(ev_ms),
                    avg_launch_ms=ms / len(ev_ms), kernel_edges_per_s=sum(ne for _, ne, _ in ev_ms) / (ms * 1e-3),
                    bytes="gathered state rows only: E * 4 * d (the kernel also gathers a relation and a time row per edge, both L2-resident tables)",
                    hbm_algorithmic=dict(achieved=nbytes / (ms * 1e-3) / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=nbytes / (ms * 1e-3) / HBM_PEAK,
                                         bytes_per_launch=nbytes / len(ev_ms), note="SURVEY 8d, temporal: E (4d + 20) + N 4d"))
    cpu = parity = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import redgnn_oracle as orc
        n_s = 2
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        off = orc.get_time_offset_list(data, gran)
        t1 = time.perf_counter()
        tr = []
        ref, _, _ = orc.extrap_forward(sd, data, off, gran, n_ent, n_rel, q[:n_s, 0], q[:n_s, 1], q[:n_s, 3], n_layer, "relu", trace=tr)
        t_cpu = time.perf_counter() - t1
        ref64 = orc.extrap_forward(sd, data, off, gran, n_ent, n_rel, q[:n_s, 0], q[:n_s, 1], q[:n_s, 3], n_layer, "relu", dtype=torch.float64)[0].numpy()
        model.eval()
        with torch.no_grad():
            class Q2:
                src_idx, rel_idx, ts = q[:n_s, 0], q[:n_s, 1], q[:n_s, 3]
            got = model(Q2)[0].cpu().numpy()
        ref = ref.detach().numpy()
        err_gpu, err_cpu = float(np.abs(got - ref64).max()), float(np.abs(ref - ref64).max())
        tol_ok = bool(np.all(np.abs(got - ref) <= 2e-5 + 1e-4 * np.abs(ref)) or err_gpu <= 4 * max(err_cpu, 1e-7))
        parity = dict(parity_at_bench_size=bool(tol_ok and np.array_equal(got == 0, ref == 0)), queries_checked=n_s,
                      gpu_max_err_vs_fp64=err_gpu, cpu_fp32_max_err_vs_fp64=err_cpu,
                      note="model_cuda_new_embedding.py: checked against the oracle only (parity unpinned, DESIGN 7)")
        cpu = dict(value=sum(t["n_edges"] for t in tr) / t_cpu, unit="edges/s", cores=torch.get_num_threads(), kind="port",
                   sample="%d of the batch's queries, oracle extrap_forward, %.1f s" % (n_s, t_cpu))
    out = {
        "metric": "edges aggregated/sec + MRR-eval queries/sec, family KG n_layer=3 at 1/2/4/8 GPU",
        "value": (2.0 if args.train else 1.0) * edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "X ICEWS14-shaped temporal EXTRAPOLATION synthetic %d entities / %d relations (+ self-loop) / %d days x %d / %d "
                               "time-sorted rows (seed 1234), n_layer=%d hidden_dim=%d attn_dim=%d, 120-day windows, step = %s, %d queries per GPU"
                               % (n_ent, n_rel, sh["n_time"], gran, len(data), n_layer, d, sh["attn_dim"],
                                  "forward + loss + backward (rg_xlayer_bwd) + Adam; value counts every edge twice" if args.train
                                  else "windowed expansion + fused layers + classifier + per-query softmax", B),
                   "batch_per_gpu": B, "global_batch": B * world, "sharding": "queries strided over ranks" if world > 1 else "none"},
        "eval_queries_per_s": B * world * args.steps / dt, "edges_per_step": edges / args.steps,
        "roofline": roof, "cpu_baseline": cpu, "parity": parity, "rg_version": int(_lib_version()),
    }
    print(json.dumps(out))


def bwd_rooflines(bwd_ms, edges_per_level, d, ap, stored=None):
    """Roofline object of the backward's message-passing launches (rg_layer_bwd = layer_bwd_kernel + bwd_combine_kernel + drel_kernel)
    from per-call records (ms, level, n_old).  Algorithmic bytes: layer_bwd_kernel E (4d + 16) + N_old (4d + 4ap) - one
    grad_agg row gathered per edge, one gradient row and one attention-projection row written per source node - and drel_kernel
    E (4d + 8 + 4ap).  Like the forward's rows, the gathered grad_agg rows of a query are L2-resident: the object that binds prices
    the gathered rows (both kernels gather one row per edge: 2 E 4d) against the L2 gather rate; hbm_algorithmic is kept beside it."""
    if not bwd_ms:
        return None
    t = sum(m for m, _, _ in bwd_ms) * 1e-3
    e_tot = sum(edges_per_level[lvl - 1] for _, lvl, _ in bwd_ms)
    b_main = sum(edges_per_level[lvl - 1] * (4.0 * d + 16) + n_old * (4.0 * d + 4 * ap) for _, lvl, n_old in bwd_ms)
    b_drel = sum(edges_per_level[lvl - 1] * (4.0 * d + 8 + 4 * ap) for _, lvl, _ in bwd_ms)
    rows = 2.0 * e_tot * 4 * d
    n = len(bwd_ms)
    out = dict(bound="l2-gather", kernel="rg_layer_bwd launches (layer_bwd_kernel + bwd_combine_kernel + drel_kernel)", achieved=rows / t / 1e9,
               peak=L2_GATHER_PEAK / 1e9, unit="GB/s", frac=rows / t / L2_GATHER_PEAK, traffic=None, launches=n, avg_launch_ms=t * 1e3 / n,
               bytes="gathered grad_agg rows of both kernels: 2 E * 4 * d", kernel_edges_per_s=e_tot / t,
               hbm_algorithmic=dict(achieved=(b_main + b_drel) / t / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=(b_main + b_drel) / t / HBM_PEAK,
                                    layer_bwd_kernel_bytes_per_launch=b_main / n, drel_kernel_bytes_per_launch=b_drel / n))
    if stored:          # per-kernel split of the same step from the committed rocprofv3 --kernel-trace --stats run
        out["per_kernel"] = stored
    return out


def bench_train(args, dist, world, rank, engine):
    """--train: a step = one training step of the reference's loop (base_model.py:45-83: forward in train mode with dropout, the
    reference's loss, backward through the HIP adjoints, Adam) on --batch train triples of the synthetic KG, queries split over the ranks
    (gradients summed by one flat all-reduce).  value = edges aggregated per second, forward + backward."""
    from red_gnn_amd.base_model import reference_loss
    from red_gnn_amd.load_data import DataLoader
    from red_gnn_amd.models import RED_GNN_trans, pad_attn
    from red_gnn_amd.sharding import allreduce_gradients
    from red_gnn_amd.synthetic import SHAPES, make_shape
    shape = SHAPES[args.config]
    kg = make_shape(args.config, seed=1234)
    loader = DataLoader(ids=dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test), verbose=False)
    p = Params(shape, kg.n_rel)
    p.dropout = 0.1
    torch.manual_seed(1234)
    model = RED_GNN_trans(p, loader).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    B = args.batch
    trip = loader.train_data[(np.arange(B * world) % len(loader.train_data))[rank::world]]
    tails = torch.as_tensor(trip[:, 2], device="cuda")
    d, n_layer = shape["hidden_dim"], shape["n_layer"]
    fwd_ev, dense_ev, bwd_ev = [], [], []
    if not args.no_kernel_events:
        engine.KERNEL_EVENTS, engine.BWD_EVENTS = fwd_ev, bwd_ev
    marks = []

    def step(timed=False):
        opt.zero_grad(set_to_none=True)
        if timed:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        scores = model(trip[:, 0], trip[:, 1])
        loss = reference_loss(scores, tails) * (B * world / len(trip))
        if timed:
            e[1].record()
        loss.backward()
        if timed:
            e[2].record()
            marks.append(e)
        if dist is not None:
            allreduce_gradients(list(model.parameters()), dist)
        opt.step()
        return model.last_stats

    for _ in range(max(args.warmup, 2)):
        step()
    fwd_ev.clear(); bwd_ev.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for _ in range(args.steps):
        st = step(timed=True)
        edges += sum(st["n_edges"])
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(edges)], device="cuda", dtype=torch.float64)
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, edges = float(tmax[0]), float(tsum[1])
    engine.KERNEL_EVENTS = engine.BWD_EVENTS = None
    if rank != 0:
        return
    e_lvl = st["n_edges"]
    fwd_ms = [(a.elapsed_time(b), ne, nn) for (a, b, ne, nn) in fwd_ev]
    bwd_ms = [(a.elapsed_time(b), lvl, n_old) for (a, b, lvl, n_old) in bwd_ev]
    roof_f, per_hop = layer_rooflines(fwd_ms, d, n_layer)
    stored = None
    try:
        with open(os.path.join(ROOT, "profiles", "train_kernels.json")) as f:
            doc = json.load(f)
        if doc.get("config") == args.config and int(doc.get("batch", -1)) == B and int(doc.get("rg_version", -1)) == int(_lib_version()):
            stored = doc.get("kernels")
    except (OSError, ValueError):
        pass
    roof_b = bwd_rooflines(bwd_ms, e_lvl, d, pad_attn(shape["attn_dim"]), stored)
    ms_f = sum(m[0].elapsed_time(m[1]) for m in marks) / len(marks)
    ms_b = sum(m[1].elapsed_time(m[2]) for m in marks) / len(marks)
    out = {
        "metric": "edges aggregated/sec + MRR-eval queries/sec, family KG n_layer=3 at 1/2/4/8 GPU",
        "value": 2.0 * edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s synthetic KG %d entities / %d relations / %d triples (seed 1234), n_layer=%d hidden_dim=%d attn_dim=%d, "
                               "TRAINING step = forward (dropout 0.1) + reference loss + HIP backward + Adam, %d train triples per GPU; value counts "
                               "every edge once in the forward and once in the backward"
                               % (args.config, kg.n_ent, kg.n_rel, shape["n_triples"], n_layer, d, shape["attn_dim"], B),
                   "batch_per_gpu": B, "global_batch": B * world,
                   "sharding": "train triples strided over ranks, gradients summed by one flat all-reduce" if world > 1 else "none"},
        "forward_ms": ms_f, "backward_ms": ms_b, "optimizer_and_host_ms": dt / args.steps * 1e3 - ms_f - ms_b,
        "edges_per_step": edges / args.steps, "roofline": roof_b, "roofline_forward": roof_f, "per_hop_forward": per_hop,
        "cpu_baseline": None, "rg_version": int(_lib_version()),
    }
    print(json.dumps(out))


def _lib_version():
    from red_gnn_amd import _lib
    return _lib.lib().rg_version()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="queries per GPU per step")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--dense-precision", default=None, choices=["f32", "f16x3", "f16x2"],
                    help="matrix products of the fused dense kernel (default: the model's default, f16x3 = fp32 arithmetic as exact three-term "
                         "f16 splits; f16x2 = 22-bit operands, opt-in)")
    ap.add_argument("--no-dense-f32", action="store_true", help="skip the secondary measurement of the step with the dense products on the f32 "
                    "MFMA forms (profiling runs: keeps the launch sequence to the timed steps)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: this many queries per step over ALL ranks (BASELINE's C4 = 256 "
                    "over 4, C5 = 256 over 8); default 0 = weak scaling with --batch queries per GPU")
    ap.add_argument("--eval-collective", default="allgather", choices=["allgather", "allreduce"],
                    help="N > 1: all-gather of the score shards (north_star) + 4-sum all-reduce, or the 4-sum all-reduce only")
    ap.add_argument("--train", action="store_true", help="a step = one training step (forward + reference loss + HIP backward + Adam) on --batch "
                    "train triples (default 256); roofline objects for the backward kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-family-eval", action="store_true", help="skip BaseModel.evaluate on the real family graph (runs after the timed region "
                    "by default); for rocprofv3 runs that should see the C2 step's kernels only")
    ap.add_argument("--family-eval", action="store_true", help=argparse.SUPPRESS)     # (the default since round 2; kept for old command lines)
    ap.add_argument("--graphs", action="store_true", help="replay the forward as a captured HIP graph (no per-kernel HIP events, so no roofline object)")
    ap.add_argument("--backend", default="nccl", help="process-group backend: nccl (= RCCL; one GPU per rank) or gloo (rehearsal of the "
                    "N > 1 path with several ranks sharing one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even at world size 1 (exercises the RCCL path on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    local_rank %= max(torch.cuda.device_count(), 1)          # (a gloo rehearsal may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            # this image's RCCL prints a version banner on STDOUT at init (NCCL_DEBUG=VERSION behaviour); the contract is ONE JSON line there
            if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION"):
                os.environ["NCCL_DEBUG"] = "WARN"
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")      # ... and whatever RCCL does print does not go to stdout
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from red_gnn_amd import _lib, engine
    from red_gnn_amd.load_data import DataLoader
    from red_gnn_amd.models import RED_GNN_trans
    from red_gnn_amd.sharding import gather_scores, query_costs, reduce_metrics, shard_balanced
    from red_gnn_amd.synthetic import SHAPES, make_shape
    from red_gnn_amd.utils import cal_ranks_csr

    if args.global_batch:            # strong scaling: the per-GPU share of a fixed global batch
        if args.global_batch % world:
            raise SystemExit("--global-batch %d is not a multiple of the %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
    scaling = "strong" if args.global_batch else "weak"
    if args.config == "X":           # temporal extrapolation (forward, or a training step with --train)
        if "--batch" not in sys.argv and not args.global_batch:
            args.batch = 64
        bench_extrapolation(args, dist, world, rank, engine, scaling)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.train:
        if "--batch" not in sys.argv and not args.global_batch:
            args.batch = 256
        bench_train(args, dist, world, rank, engine)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.config == "C5":          # the temporal interpolation path has its own model and step
        bench_temporal(args, dist, world, rank, engine, scaling)
        if dist is not None:
            dist.destroy_process_group()
        return
    shape = SHAPES[args.config]
    kg = make_shape(args.config, seed=1234)
    ids = dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test)
    loader = DataLoader(ids=ids, verbose=False)
    torch.manual_seed(1234)
    model = RED_GNN_trans(Params(shape, kg.n_rel), loader).cuda().eval()
    if args.dense_precision:
        model.dense_precision = args.dense_precision
    model.use_graphs = args.graphs            # the default run keeps the eager path: its kernels are timed one by one below
    d = shape["hidden_dim"]

    B = args.batch
    n_q = loader.n_test
    glob = np.arange(B * world) % n_q                        # the global batch: the first B * world test queries
    if world > 1:
        # per-query subgraphs differ by far more than 2x (hub subjects): deal the queries by estimated cost, equal counts per rank
        costs = query_costs(loader._tgraph_base, kg.n_ent, np.array([loader.test_q[i][0] for i in glob]))
        mine = np.asarray(shard_balanced(costs, world)[rank])
    else:
        mine = np.arange(B)
    q_idx = glob[mine]
    subs, rels, a_ptr, a_idx, f_ptr, f_idx = loader.get_batch_csr(q_idx, data="test")

    engine.FORCE_WALK = int(os.environ.get("RG_BENCH_FORCE_WALK", "0"))      # (kernel experiments: tools/pmc.sh runs with a forced walk)
    kernel_events, dense_events = [], []
    if not args.no_kernel_events and not args.graphs:
        engine.KERNEL_EVENTS = kernel_events
        engine.DENSE_EVENTS = dense_events

    pending = []     # the previous step's all-gather (RCCL stream): it overlaps the next step's kernels
    last = {}

    def step():
        with torch.no_grad():
            scores = model(subs, rels, mode="test")
            ranks = cal_ranks_csr(scores, a_ptr, a_idx, f_ptr, f_idx)
            sums = torch.stack([(1.0 / ranks).sum(), (ranks <= 1).sum(), (ranks <= 10).sum(),
                                torch.tensor(float(ranks.numel()), device=ranks.device)])
            if dist is not None:
                while pending:
                    pending.pop()[0].wait()
                if args.eval_collective == "allgather":
                    # north star: RCCL all-gather of the score shards over xGMI; asynchronous, joined one step later
                    pending.append(gather_scores(scores, dist, async_op=True) + (scores,))
                sums = reduce_metrics(sums, dist)
            last["scores"], last["ranks"] = scores, ranks
        return sums, model.last_stats

    if args.graphs:
        for _ in range(3):          # the third call of a shape captures its HIP graph: keep that out of the timed region
            step()
    for _ in range(args.warmup):
        step()
    kernel_events.clear()
    dense_events.clear()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges = 0
    for _ in range(args.steps):
        sums, st = step()
        edges += sum(st["n_edges"])
    while pending:
        pending.pop()[0].wait()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(edges)], device="cuda", dtype=torch.float64)
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, edges = float(tmax[0]), float(tsum[1])
    total_edges = float(edges)

    # per-launch records before anything else touches the event lists
    ev_ms = [(e0.elapsed_time(e1), ne, nn) for (e0, e1, ne, nn) in kernel_events]
    dense_ms = [(e0.elapsed_time(e1), n) for (e0, e1, n) in dense_events]
    last_default = dict(last)          # the timed region's last scores (default precision)
    gpu_scores = last["scores"].cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    gpu_ranks = last["ranks"].double().cpu().numpy() if gpu_scores is not None else None

    # the same step with the dense kernel's products on the f32 MFMA forms (after the timed region; the default runs them as exact
    # three-term f16 splits)
    dense_f32 = None
    if model.dense_precision != "f32" and not args.graphs and not args.no_dense_f32:
        saved_prec, saved_ev = model.dense_precision, (engine.KERNEL_EVENTS, engine.DENSE_EVENTS)
        model.dense_precision = "f32"
        engine.KERNEL_EVENTS = engine.DENSE_EVENTS = None
        k2 = max(1, min(args.steps, 10))
        step()
        while pending:
            pending.pop()[0].wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        while pending:
            pending.pop()[0].wait()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t2 = torch.tensor([time.perf_counter() - t1], device="cuda", dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        dense_f32 = dict(ms_per_step=float(t2[0]) / k2 * 1e3, steps=k2, value=total_edges / args.steps * k2 / float(t2[0]),
                         max_abs_score_diff_to_default=float((last["scores"] - last_default["scores"]).abs().max()))
        model.dense_precision = saved_prec
        engine.KERNEL_EVENTS, engine.DENSE_EVENTS = saved_ev

    # second half of the metric's name, after (never inside) the timed region
    family = None if args.no_family_eval else family_eval(dist, world, engine, cpu_leg=not args.no_cpu_baseline, rank=rank)

    if rank == 0:
        version = int(_lib.lib().rg_version())
        roof, per_hop = layer_rooflines(ev_ms, d, shape["n_layer"], stored_traffic(args.config, B, version))
        traffic_entry = stored_traffic(args.config, B, version)
        roof_dense = dense_roofline(dense_ms, d, shape["attn_dim"], shape["n_layer"], B, model.dense_precision) if dense_ms else None
        if roof_dense and traffic_entry:
            for name, rec in traffic_entry.get("dense", {}).items():          # the same PMC passes also saw the dense launches
                if name.endswith(roof_dense["kernel"]):
                    roof_dense["traffic"] = rec["hbm_bytes_per_launch"]
                    roof_dense["hbm_measured_frac"] = rec["hbm_bytes_per_launch"] / (roof_dense["avg_launch_ms"] * 1e-3) / HBM_PEAK
        cpu = parity = None
        if gpu_scores is not None:
            query, answer = loader.test_q, loader.test_a
            ans = [np.asarray(answer[i]) for i in q_idx]
            filt = [np.asarray(loader.filters[(int(s), int(r))]) for s, r in zip(subs, rels)]
            cpu, parity = cpu_baseline(kg, shape, model.state_dict(), np.asarray(subs), np.asarray(rels), ans, filt, gpu_scores, gpu_ranks,
                                       a_ptr.cpu().numpy())
        s = sums.double().cpu().numpy()
        out = {
            "metric": "edges aggregated/sec + MRR-eval queries/sec, family KG n_layer=3 at 1/2/4/8 GPU",
            "value": total_edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE_OF[dense_kernel_of(d, model.dense_precision)[1]], "data": "synthetic",
            "dense_precision": DENSE_NOTE[dense_kernel_of(d, model.dense_precision)[1]],
            "dense_f32": dense_f32,
            "config": {"workload": "%s synthetic KG %d entities / %d relations / %d triples (seed 1234), n_layer=%d hidden_dim=%d attn_dim=%d, "
                                   "eval step = expansion + fused layers + GRU/readout + filtered ranking, %d queries per GPU"
                                   % (args.config, kg.n_ent, kg.n_rel, shape["n_triples"], shape["n_layer"], d, shape["attn_dim"], B),
                       "batch_per_gpu": B, "global_batch": B * world,
                       "sharding": ("queries dealt over ranks by estimated cost (equal counts), %s (%s)"
                                    % ("scores all-gathered + 4 metric sums all-reduced" if args.eval_collective == "allgather"
                                       else "4 metric sums all-reduced only", "RCCL" if args.backend == "nccl" else args.backend + ", not RCCL"))
                       if world > 1 else "none"},
            "eval_queries_per_s": B * world * args.steps / dt,
            "edges_per_step": total_edges / args.steps,
            "mrr_of_random_init": float(s[0] / s[3]),
            "family_eval_queries_per_s": family["queries_per_s"] if family else None,
            "roofline": roof, "per_hop": per_hop, "roofline_dense": roof_dense, "cpu_baseline": cpu,
            "parity": parity, "family_eval": family, "rg_version": version,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
