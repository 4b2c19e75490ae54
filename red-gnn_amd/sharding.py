"""Query sharding over GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Every query builds its own subgraph (the batch index is carried through the whole path,
Static/transductive/models.py:73, load_data.py:115-123), so the batch shards with no data-path
collective: each rank holds a full replica of the KG and of the weights and runs the hot path on its
slice of the queries.  The only exchanges are at the edges of the path: an all-gather of the score
shards (evaluation, as BASELINE.json's north star specifies) or an all-reduce of four metric sums, and
an all-reduce of the parameter gradients in training (the loss is a sum over queries, so gradients add).
The functions take the process-group module as an argument so they run under gloo on CPU in tests.
"""
import torch


def shard_slice(n, world, rank):
    """Contiguous, balanced split of n items: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_cost(costs, world):
    """Greedy longest-processing-time split of queries by an estimated cost (e.g. out-degree of the
    query subject: per-query subgraphs differ by >2x).  Returns a list of index lists."""
    import numpy as np
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    loads = [0.0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(int(i))
        loads[r] += float(costs[i])
    return [sorted(p) for p in parts]


def shard_balanced(costs, world):
    """Equal-count split balanced by cost: queries sorted by descending cost are dealt in snake order (0..W-1, W-1..0, ...),
    so every rank gets n // world (+1 for the first n % world) queries and near-equal total cost.  Returns a list of sorted
    index lists; ``unshard_order(parts)`` gives the permutation that restores query order after a rank-order gather."""
    import numpy as np
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    parts = [[] for _ in range(world)]
    for j, i in enumerate(order):
        lap, pos = divmod(j, world)
        parts[pos if lap % 2 == 0 else world - 1 - pos].append(int(i))
    n = len(order)
    base, extra = divmod(n, world)
    # snake dealing leaves the counts equal when world divides n; otherwise move the surplus so that rank r holds base + (r < extra)
    want = [base + (1 if r < extra else 0) for r in range(world)]
    surplus = [i for r in range(world) for i in parts[r][want[r]:]]
    parts = [p[:want[r]] for r, p in enumerate(parts)]
    for r in range(world):
        while len(parts[r]) < want[r]:
            parts[r].append(surplus.pop())
    return [sorted(p) for p in parts]


def unshard_order(parts):
    """inv such that cat([x[p] for p in parts])[inv] == x  (restores query order after gather_scores over cost-sharded ranks)."""
    import numpy as np
    cat = np.concatenate([np.asarray(p, dtype=np.int64) for p in parts]) if parts else np.zeros(0, np.int64)
    inv = np.empty(len(cat), dtype=np.int64)
    inv[cat] = np.arange(len(cat))
    return inv


def query_costs(base_triples, n_ent, subs, hops=2):
    """Estimated cost of each query = number of edges its first ``hops`` expansions touch on the graph built from
    ``base_triples`` (inverse and identity rows added, load_data.py:69-79): hop 1 = out-degree of the subject, hop 2 = sum of the
    out-degrees of its neighbours (itself included through the identity row).  Host numpy, O(|KG|)."""
    import numpy as np
    return entity_costs(base_triples, n_ent, hops)[np.asarray(subs, dtype=np.int64)]


def entity_costs(base_triples, n_ent, hops=2):
    """query_costs for every entity as the query subject: one table per graph (the trainer builds it once per epoch)."""
    import numpy as np
    t = np.asarray(base_triples, dtype=np.int64).reshape(-1, 3)
    heads = np.concatenate([t[:, 0], t[:, 2], np.arange(n_ent)])
    tails = np.concatenate([t[:, 2], t[:, 0], np.arange(n_ent)])
    deg = np.bincount(heads, minlength=n_ent).astype(np.float64)
    cost = deg.copy()
    reach = deg
    for _ in range(hops - 1):
        nxt = np.zeros(n_ent)
        np.add.at(nxt, heads, reach[tails])
        reach = nxt
        cost = cost + reach
    return cost


def split_batch(costs, world, rank):
    """Positions of one batch that rank ``rank`` of ``world`` runs: equal counts, near-equal estimated cost (shard_balanced).
    Every rank computes the same split from the same costs; the union over ranks is the batch, the parts are disjoint."""
    import numpy as np
    if world == 1:
        return np.arange(len(costs))
    return np.asarray(shard_balanced(costs, world)[rank], dtype=np.int64)


def gather_scores(local_scores, dist, sizes=None, async_op=False):
    """All-gather of [b_r, n_ent] score shards into [sum b_r, n_ent] on every rank (rank order).
    ``async_op=True`` returns (work, out): the collective runs on the RCCL stream and overlaps whatever the
    caller launches next; call work.wait() before reading ``out`` (and keep ``local_scores`` alive until then)."""
    world = dist.get_world_size()
    if sizes is None:
        out = torch.empty((world * local_scores.shape[0], local_scores.shape[1]), dtype=local_scores.dtype, device=local_scores.device)
        work = dist.all_gather_into_tensor(out, local_scores.contiguous(), async_op=async_op)
        return (work, out) if async_op else out
    assert not async_op
    bufs = [torch.empty((s, local_scores.shape[1]), dtype=local_scores.dtype, device=local_scores.device) for s in sizes]
    dist.all_gather(bufs, local_scores.contiguous())
    return torch.cat(bufs, 0)


def reduce_metrics(sums, dist):
    """All-reduce of (sum 1/rank, #rank<=1, #rank<=10, count); MRR / H@k follow by division."""
    sums = sums.clone()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


def allreduce_gradients(params, dist):
    """One flat all-reduce (sum) of all parameter gradients (<= 0.7 MB for every preset)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
