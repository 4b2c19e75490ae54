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


def assert_close_fp32(actual, ref32, ref64, rtol, atol, what=""):
    """Tolerance for deep / wide fp32 sums (hub destinations with thousands of in-edges, 4-5 hops): every element of ``actual``
    is within rtol/atol of the reference algorithm evaluated in fp64 (``ref64``), OR no further from it than 4x the largest error
    the reference's own fp32 evaluation (``ref32``, the CPU path) makes on this tensor.  The reference's summation order is
    unspecified ("Determinism / tolerance"), so its fp32 result is itself only that close to the exact value."""
    actual, ref32, ref64 = (np.asarray(x, dtype=np.float64) for x in (actual, ref32, ref64))
    err = np.abs(actual - ref64)
    ok = err <= atol + rtol * np.abs(ref64)
    if ok.all():
        return
    ref_err = float(np.abs(ref32 - ref64).max())
    bad = ~ok & (err > 4.0 * ref_err)
    assert not bad.any(), ("%s: %d of %d elements beyond rtol=%g atol=%g of the fp64 reference AND beyond 4x the fp32 reference's own "
                           "max error %.3g (worst %.3g)" % (what, int(bad.sum()), err.size, rtol, atol, ref_err, float(err[bad].max())))
