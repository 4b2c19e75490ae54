"""Shared helpers for the tests: fixture loading and oracle graphs built from id fixtures."""
import hashlib
import os

import numpy as np

from oracle import redgnn_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def params_of(fx):
    return {k[len("param::"):]: fx[k] for k in fx if k.startswith("param::")}


def grads_of(fx):
    return {k[len("grad::"):]: fx[k] for k in fx if k.startswith("grad::")}


def oracle_graph(ids, mode):
    """'train' graph = facts only (load_data.py:49); eval graph = facts + train (load_data.py:50)."""
    n_ent, n_rel = int(ids["n_ent"]), int(ids["n_rel"])
    if mode == "train":
        trip = orc.double_triple(ids["facts"], n_rel)
    else:
        trip = np.concatenate([orc.double_triple(ids["facts"], n_rel), orc.double_triple(ids["train"], n_rel)], 0)
    return orc.OracleGraph(trip, n_ent, n_rel)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def edge_multiset_hash(edges):
    e = np.asarray(edges)[:, :4].astype(np.int64)
    o = np.lexsort((e[:, 3], e[:, 2], e[:, 1], e[:, 0]))
    return sha(e[o])


def sorted_edges(edges):
    e = np.asarray(edges).astype(np.int64)
    o = np.lexsort(tuple(e[:, c] for c in reversed(range(e.shape[1]))))
    return e[o]
