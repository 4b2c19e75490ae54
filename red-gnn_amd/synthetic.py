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
    "C5": dict(n_ent=7_000, n_rel=230, n_triples=90_000, n_layer=5, hidden_dim=64, attn_dim=30),
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
