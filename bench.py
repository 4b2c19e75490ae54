"""bench.py — edges aggregated/sec + MRR-eval queries/sec of the RED-GNN hot path on MI355X.

A *step* is one evaluation pass of the hot path over one batch of synthetic queries:
frontier expansion + n_layer fused message-passing layers + GRU/readout + filtered ranking
(RED_GNN_trans.forward + cal_ranks of the reference).  Workload = BASELINE.json configs[1]:
synthetic KG 10k entities / 50 relations / 200k triples (seed 1234), n_layer=3, hidden_dim=64,
attn_dim=5, batch of B queries per GPU.  Queries are sharded over ranks (weak scaling: B per GPU
fixed); for N>1 scores are all-gathered over RCCL and the 4 metric sums all-reduced.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the dominant
kernel (layer_fwd_kernel, HIP events on its stream) and `cpu_baseline` (the oracle on host cores).
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

HBM_PEAK = 8.0e12      # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; ~6.3e12 achievable)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, same guide: f32-input MFMA = the FP32 vector rate


class Params:
    def __init__(self, shape, n_rel):
        self.n_layer, self.hidden_dim, self.attn_dim = shape["n_layer"], shape["hidden_dim"], shape["attn_dim"]
        self.n_rel, self.act, self.dropout = n_rel, "relu", 0.0


def algorithmic_bytes(n_edges, n_nodes, d):
    """per edge one fp32 source row (4d B) + four int32 indices (16 B); per output node one fp32 row."""
    return n_edges * (4 * d + 16) + n_nodes * 4 * d


def cpu_baseline(kg, shape, state, subs, rels, ans, filt, budget_s=20.0):
    """The oracle (CPU restatement, validated against the reference) on the host cores: same KG, same
    weights, the first queries of the same batch.  Bounded sample: batches of 4 queries until ~budget_s."""
    from oracle import redgnn_oracle as orc
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(n_cpu, 64)))       # the cores this process may actually run on
    og = orc.OracleGraph(np.concatenate([orc.double_triple(kg.facts, kg.n_rel), orc.double_triple(kg.train, kg.n_rel)], 0),
                         kg.n_ent, kg.n_rel)
    p = {k: v.detach().cpu() for k, v in state.items()}
    bs, done, edges, t_tot = 4, 0, 0, 0.0
    while done + bs <= len(subs) and (t_tot < budget_s or done == 0):
        sl = slice(done, done + bs)
        t0 = time.perf_counter()
        trace = []
        sc = orc.forward(p, og, subs[sl], rels[sl], shape["n_layer"], act="relu", trace=trace).numpy()
        labels = np.zeros((bs, kg.n_ent)); fl = np.zeros((bs, kg.n_ent))
        for i in range(bs):
            labels[i, ans[done + i]] = 1
            fl[i, filt[done + i]] = 1
        orc.cal_ranks(sc, labels, fl)
        t_tot += time.perf_counter() - t0
        edges += sum(len(t["edges"]) for t in trace)
        done += bs
    return dict(value=edges / t_tot, unit="edges/s", cores=torch.get_num_threads(), kind="port",
                sample="%d of the batch's queries (batches of %d), same KG and weights, forward + cal_ranks, %.1f s; %.1f queries/s"
                       % (done, bs, t_tot, done / t_tot),
                queries_per_s=done / t_tot)


def family_eval(dist, world, engine):
    """BaseModel.evaluate (forward + filtered ranking, valid + test) on the real family graph with the reference's shape for
    BASELINE configs[0] (n_layer=3, hidden_dim=64, n_tbatch=50) and random-init weights: total queries per second over all
    ranks (evaluation batches are dealt round-robin).  The id triples are the committed fixture of the reference's data/family."""
    path = os.path.join(ROOT, "tests", "golden", "family_ids.npz")
    if not os.path.exists(path):
        return None
    from red_gnn_amd.base_model import BaseModel
    from red_gnn_amd.load_data import DataLoader
    saved = engine.KERNEL_EVENTS, engine.DENSE_EVENTS
    engine.KERNEL_EVENTS = engine.DENSE_EVENTS = None          # graph replay path
    try:
        loader = DataLoader(ids=dict(np.load(path)), verbose=False)

        class Opt:
            lr, decay_rate, lamb, hidden_dim, attn_dim, n_layer, dropout, act, n_batch, n_tbatch = 0.0036, 0.999, 1.7e-5, 64, 5, 3, 0.29, "relu", 20, 50
            n_rel = loader.n_rel

        torch.manual_seed(1234)
        bm = BaseModel(Opt, loader, dist=dist if world > 1 else None)
        for _ in range(3):                                      # the third pass of a batch shape captures its graph
            bm.evaluate()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
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
        nq = loader.n_valid + loader.n_test
        return dict(queries_per_s=nq * reps / dt, queries=nq, n_tbatch=50, hidden_dim=64, n_layer=3, seconds_per_pass=dt / reps,
                    valid_mrr_of_random_init=float(mrr), path="HIP graph replay per batch")
    finally:
        engine.KERNEL_EVENTS, engine.DENSE_EVENTS = saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="queries per GPU per step")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--family-eval", action="store_true", help="also time BaseModel.evaluate on the real family graph (n_tbatch=50, graph replay); "
                    "off by default so that a rocprofv3 run of the default command sees the C2 step's kernels only")
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
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from red_gnn_amd import engine
    from red_gnn_amd.load_data import DataLoader
    from red_gnn_amd.models import RED_GNN_trans
    from red_gnn_amd.sharding import shard_slice, gather_scores, reduce_metrics
    from red_gnn_amd.synthetic import SHAPES, make_shape
    from red_gnn_amd.utils import cal_ranks_csr

    shape = SHAPES[args.config]
    kg = make_shape(args.config, seed=1234)
    ids = dict(n_ent=kg.n_ent, n_rel=kg.n_rel, facts=kg.facts, train=kg.train, valid=kg.valid, test=kg.test)
    loader = DataLoader(ids=ids, verbose=False)
    torch.manual_seed(1234)
    model = RED_GNN_trans(Params(shape, kg.n_rel), loader).cuda().eval()
    model.use_graphs = args.graphs            # the default run keeps the eager path: its kernels are timed one by one below
    d = shape["hidden_dim"]

    B = args.batch
    n_q = loader.n_test
    lo, hi = shard_slice(B * world, world, rank)              # this rank's queries of the global batch
    q_idx = np.arange(lo, hi) % n_q
    subs, rels, a_ptr, a_idx, f_ptr, f_idx = loader.get_batch_csr(q_idx, data="test")

    kernel_events, dense_events = [], []
    if not args.no_kernel_events and not args.graphs:
        engine.KERNEL_EVENTS = kernel_events
        engine.DENSE_EVENTS = dense_events

    pending = []     # the previous step's all-gather (RCCL stream): it overlaps the next step's kernels

    def step():
        with torch.no_grad():
            scores = model(subs, rels, mode="test")
            ranks = cal_ranks_csr(scores, a_ptr, a_idx, f_ptr, f_idx)
            sums = torch.stack([(1.0 / ranks).sum(), (ranks <= 1).sum(), (ranks <= 10).sum(),
                                torch.tensor(float(ranks.numel()), device=ranks.device)])
            if dist is not None:
                while pending:
                    pending.pop()[0].wait()
                # north star: RCCL all-gather of the score shards over xGMI; asynchronous, joined one step later
                pending.append(gather_scores(scores, dist, async_op=True) + (scores,))
                sums = reduce_metrics(sums, dist)
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

    family = family_eval(dist, world, engine) if args.family_eval else None      # second half of the metric's name

    if rank == 0:
        # dominant kernel: layer_fwd_kernel (HIP events recorded around every launch of the timed steps)
        roof = None
        if kernel_events:
            k_ms = sum(e0.elapsed_time(e1) for (e0, e1, _, _) in kernel_events)
            k_bytes = sum(algorithmic_bytes(ne, nn, d) for (_, _, ne, nn) in kernel_events)
            k_edges = sum(ne for (_, _, ne, _) in kernel_events)
            n_launch = len(kernel_events)
            achieved = k_bytes / (k_ms * 1e-3)
            traffic = None
            prof = os.path.join(ROOT, "profiles", "traffic_layer_fwd.json")
            if os.path.exists(prof):
                with open(prof) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            roof = dict(bound="hbm", kernel="layer_fwd_kernel", achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=achieved / HBM_PEAK, traffic=traffic, launches=n_launch,
                        avg_launch_ms=k_ms / n_launch, algorithmic_bytes_per_launch=k_bytes / n_launch,
                        kernel_edges_per_s=k_edges / (k_ms * 1e-3))
        roof_dense = None
        if dense_events:
            # second kernel of the step: W_h + GRU + projections on f32 MFMA.  Algorithmic flops per node (what the
            # reference computes): 2*d*d*(1 + 3 + 3) for W_h, weight_ih, weight_hh, + 2*d*attn_dim for the hoisted Ws_attn
            d_ms = sum(e0.elapsed_time(e1) for (e0, e1, _) in dense_events)
            rows = sum(n for (_, _, n) in dense_events)
            flops = rows * (2.0 * d * d * 7 + 2.0 * d * shape["attn_dim"])
            roof_dense = dict(bound="mfma", kernel="dense_kernel", achieved=flops / (d_ms * 1e-3) / 1e12, peak=MFMA_F32_PEAK / 1e12,
                              unit="TFLOP/s", frac=flops / (d_ms * 1e-3) / MFMA_F32_PEAK, traffic=None, launches=len(dense_events),
                              avg_launch_ms=d_ms / len(dense_events), algorithmic_flops_per_launch=flops / len(dense_events))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            query, answer = loader.test_q, loader.test_a
            ans = [np.asarray(answer[i]) for i in q_idx]
            filt = [np.asarray(loader.filters[(int(s), int(r))]) for s, r in zip(subs, rels)]
            cpu = cpu_baseline(kg, shape, model.state_dict(), np.asarray(subs), np.asarray(rels), ans, filt)
        s = sums.double().cpu().numpy()
        out = {
            "metric": "edges aggregated/sec + MRR-eval queries/sec, family KG n_layer=3 at 1/2/4/8 GPU",
            "value": total_edges / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s synthetic KG %d entities / %d relations / %d triples (seed 1234), n_layer=%d hidden_dim=%d attn_dim=%d, "
                                   "eval step = expansion + fused layers + GRU/readout + filtered ranking, %d queries per GPU"
                                   % (args.config, kg.n_ent, kg.n_rel, shape["n_triples"], shape["n_layer"], d, shape["attn_dim"], B),
                       "batch_per_gpu": B, "global_batch": B * world, "sharding": ("queries over ranks, scores all-gathered (%s)" % ("RCCL" if args.backend == "nccl" else args.backend)) if world > 1 else "none"},
            "eval_queries_per_s": B * world * args.steps / dt,
            "edges_per_step": total_edges / args.steps,
            "mrr_of_random_init": float(s[0] / s[3]),
            "roofline": roof, "roofline_dense": roof_dense, "cpu_baseline": cpu, "family_eval": family,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
