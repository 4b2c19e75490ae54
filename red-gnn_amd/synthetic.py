"""Seeded synthetic knowledge graphs of the shapes named in BASELINE.json.

The generator follows SURVEY.md §8(d): heads/tails are drawn from a Zipf(1.0) law over a
random entity permutation mixed 50/50 with uniform draws (so there are hubs whose in-degree
is far above the mean, as in WN18RR), relations are Zipf(1.0), triples are de-duplicated, the
named triple count T is the *eval* graph's base set and is split facts:train = 3:1 (the
reference's split, README.md:38), and valid/test are 0.05*T extra triples each from the
same law.  Pure numpy; no device code.  Entity/relation names written by
``write_task_dir`` use the text layout read at Static/transductive/load_data.py:11-35,58-67.
"""
import os
from dataclasses import dataclass

import numpy as np

# (n_ent, n_rel, n_triples, n_layer, hidden_dim, attn_dim) for BASELINE.json configs[1..4]
SHAPES = {
    "C2": dict(n_ent=10_000, n_rel=50, n_triples=200_000, n_layer=3, hidden_dim=64, attn_dim=5),
    "C3": dict(n_ent=40_000, n_rel=11, n_triples=93_000, n_layer=5, hidden_dim=64, attn_dim=5),
    "C4": dict(n_ent=15_000, n_rel=237, n_triples=310_000, n_layer=4, hidden_dim=128, attn_dim=5),
    "C5": dict(n_ent=7_000, n_rel=230, n_triples=90_000, n_layer=5, hidden_dim=64, attn_dim=30, n_time=365),
    # the same ICEWS14 shape for the EXTRAPOLATION setting (Temporal/extrapolation/main.py: DP_steps = 3, reversed relations
    # added, hourly stamps at a granularity of 24): 2 x 90 k time-sorted rows, 460 relations + the self-loop relation
    "X": dict(n_ent=7_000, n_rel=230, n_triples=90_000, n_layer=3, hidden_dim=64, attn_dim=30, n_time=365, time_granularity=24),
}


@dataclass
class SyntheticKG:
    n_ent: int
    n_rel: int
    facts: np.ndarray   # int64 [n,3] (h, r, t)
    train: np.ndarray
    valid: np.ndarray
    test: np.ndarray


def _zipf_p(n):
    p = 1.0 / np.arange(1, n + 1, dtype=np.float64)
    return p / p.sum()


def make_synthetic_kg(n_ent, n_rel, n_triples, seed=1234, extra_frac=0.05):
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n_ent)
    p_ent = _zipf_p(n_ent)
    p_rel = _zipf_p(n_rel)
    n_extra = max(1, int(round(extra_frac * n_triples)))
    need = n_triples + 2 * n_extra

    def draw_ent(n):
        z = perm[rng.choice(n_ent, size=n, p=p_ent)]
        u = rng.integers(0, n_ent, size=n)
        return np.where(rng.random(n) < 0.5, z, u)

    chunks = []
    have = 0
    seen = None
    while have < need:
        n = int((need - have) * 1.5) + 1024
        h, t = draw_ent(n), draw_ent(n)
        r = rng.choice(n_rel, size=n, p=p_rel)
        keep = h != t
        trip = np.stack([h[keep], r[keep], t[keep]], 1).astype(np.int64)
        chunks.append(trip)
        allt = np.concatenate(chunks, 0)
        key = (allt[:, 0] * n_rel + allt[:, 1]) * n_ent + allt[:, 2]
        _, first = np.unique(key, return_index=True)
        first.sort()                      # keep first occurrences in draw order
        seen = allt[first]
        chunks = [seen]
        have = len(seen)
    seen = seen[:need]
    base = seen[:n_triples]
    n_fact = (n_triples * 3) // 4
    return SyntheticKG(n_ent=n_ent, n_rel=n_rel,
                       facts=base[:n_fact].copy(), train=base[n_fact:].copy(),
                       valid=seen[n_triples:n_triples + n_extra].copy(),
                       test=seen[n_triples + n_extra:].copy())


def make_shape(name, seed=1234):
    s = SHAPES[name]
    return make_synthetic_kg(s["n_ent"], s["n_rel"], s["n_triples"], seed=seed)


@dataclass
class SyntheticTKG:
    """Quadruple graph in the layout of Temporal/interpolation/graph.py:34-49: forward quads, their '~' inverses
    (relation + n_rel_base) and one identity row per entity (relation 2*n_rel_base, sentinel time id = n_time_base)."""
    n_ent: int
    n_rel: int          # relation ids in the graph incl. inverses and idd (rows of the relation tables = n_rel + 1)
    n_time: int         # time ids incl. the sentinel
    quads: np.ndarray   # int32 [n,4] (head, rel, tail, time id), identity rows last
    n_base: int         # number of forward quads (rows [0, n_base) of ``quads``)


def make_temporal_kg(n_ent, n_rel_base, n_time_base, n_quads, seed=1234):
    """ICEWS14-shaped synthetic (BASELINE configs[4]): heads/tails Zipf(1.0) over a random permutation mixed 50/50 with
    uniform draws (as make_synthetic_kg), relations and timestamps uniform."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n_ent)
    w = _zipf_p(n_ent)

    def ent(n):
        return np.where(rng.random(n) < 0.5, perm[rng.choice(n_ent, n, p=w)], rng.integers(0, n_ent, n))

    h, t = ent(n_quads), ent(n_quads)
    r = rng.integers(0, n_rel_base, n_quads)
    tau = rng.integers(0, n_time_base, n_quads)
    quads = np.stack([h, r, t, tau], 1)
    inv = np.stack([t, r + n_rel_base, h, tau], 1)
    n_rel = 2 * n_rel_base + 1
    idd = np.stack([np.arange(n_ent), np.full(n_ent, n_rel - 1), np.arange(n_ent), np.full(n_ent, n_time_base)], 1)
    return SyntheticTKG(n_ent=n_ent, n_rel=n_rel, n_time=n_time_base + 1,
                        quads=np.concatenate([quads, inv, idd], 0).astype(np.int32), n_base=n_quads)


def make_temporal_shape(name="C5", seed=1234):
    s = SHAPES[name]
    return make_temporal_kg(s["n_ent"], s["n_rel"], s.get("n_time", 365), s["n_triples"], seed=seed)


def make_extrapolation_shape(name="X", seed=1234):
    """(data int64 [2 n, 4] = (subject, relation, object, time) sorted by time with the reversed rows added (relation + n_rel), n_ent,
    n_rel_true = 2 n_rel, time_granularity): the layout of the reference's contents.data (Temporal/extrapolation/utils.py Data)."""
    s = SHAPES[name]
    tkg = make_temporal_kg(s["n_ent"], s["n_rel"], s.get("n_time", 365), s["n_triples"], seed=seed)
    gran = s.get("time_granularity", 24)
    rows = tkg.quads[:2 * tkg.n_base].astype(np.int64)                 # forward quads + their inverses (identity rows are per query)
    rng = np.random.default_rng(seed + 1)
    rows[:, 3] = rows[:, 3] * gran + np.tile(rng.integers(0, gran, tkg.n_base), 2)      # a forward row and its reverse share the stamp
    rows = rows[np.argsort(rows[:, 3], kind="stable")]
    return rows, s["n_ent"], 2 * s["n_rel"], gran


def write_task_dir(kg, task_dir):
    """Write ``kg`` in the reference's text format (entities.txt, relations.txt, *.txt triples)."""
    os.makedirs(task_dir, exist_ok=True)
    with open(os.path.join(task_dir, "entities.txt"), "w") as f:
        for i in range(kg.n_ent):
            f.write("e%d\n" % i)
    with open(os.path.join(task_dir, "relations.txt"), "w") as f:
        for i in range(kg.n_rel):
            f.write("r%d\n" % i)
    for name in ("facts", "train", "valid", "test"):
        with open(os.path.join(task_dir, name + ".txt"), "w") as f:
            for h, r, t in getattr(kg, name):
                f.write("e%d r%d e%d\n" % (h, r, t))
