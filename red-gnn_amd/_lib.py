"""ctypes binding of libredgnn.so — the C-ABI declared in include/redgnn.h.

This is the binding a maintainer of the reference would add.  There is no
CPU fallback: if the HIP library is missing, loading fails loudly.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RG_LIB") or os.path.join(_PKG, "libredgnn.so")      # RG_LIB: an alternative build (kernel A/B runs)

# every symbol include/redgnn.h declares (tests check the library exports all of them)
SYMBOLS = [
    "rg_last_error", "rg_version", "rg_hipgraph_fill_nodes",
    "rg_graph_create", "rg_graph_create_device", "rg_graph_export_packs", "rg_tgraph_create", "rg_tgraph_create_excluding", "rg_graph_destroy", "rg_graph_n_fact", "rg_graph_export",
    "rg_frontier_workspace_bytes", "rg_frontier_create", "rg_frontier_destroy", "rg_frontier_reset",
    "rg_frontier_reset_nodes", "rg_frontier_expand", "rg_frontier_nodes", "rg_frontier_edges_scratch_bytes", "rg_frontier_edges",
    "rg_layer_fwd_scratch_bytes", "rg_layer_fwd", "rg_layer_fwd_plan", "rg_tlayer_fwd", "rg_xlayer_fwd", "rg_frontier_set_window", "rg_layer_bwd_scratch_bytes", "rg_layer_bwd", "rg_tlayer_bwd_scratch_bytes", "rg_tlayer_bwd", "rg_xlayer_bwd", "rg_dense_fwd_supported", "rg_dense_scratch_bytes", "rg_dense_fwd", "rg_dense_fwd_dev", "rg_dense_train_fwd", "rg_dense_train_fwd_as", "rg_rows_addmm", "rg_dense_train_bwd", "rg_dense_train_bwd2", "rg_split3_roundtrip", "rg_split3_product_check", "rg_gram_tn_scratch_bytes", "rg_gram_tn", "rg_rank",
    "rg_frontier_expand_async", "rg_frontier_expand_nodes_async", "rg_frontier_set_edge_hint", "rg_frontier_count_ptr", "rg_frontier_level_counts", "rg_attn_tables",
]

_lib = None


class NativeError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built (``python -m red_gnn_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "libredgnn.so is missing at %s: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback." % LIB_PATH)
    # torch bundles its own HIP runtime (libamdhip64.so.7); it must be loaded first so that this
    # library binds to the SAME runtime (one device context, shared streams and allocations).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.rg_last_error.restype = C.c_char_p
    L.rg_version.restype = C.c_int
    L.rg_hipgraph_fill_nodes.argtypes = [vp]
    L.rg_graph_create.argtypes = [i32, i32, vp, i64, C.c_int, C.POINTER(vp)]
    L.rg_graph_create_device.argtypes = [i32, i32, vp, i64, C.c_int, vp, C.POINTER(vp)]
    L.rg_graph_export_packs.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), vp, vp, vp, vp]
    L.rg_tgraph_create.argtypes = [i32, i32, i32, vp, i64, C.POINTER(vp)]
    L.rg_tgraph_create_excluding.argtypes = [i32, i32, i32, vp, i64, vp, i64, C.POINTER(vp)]
    L.rg_graph_destroy.argtypes = [vp]
    L.rg_graph_n_fact.argtypes = [vp]
    L.rg_graph_n_fact.restype = i64
    L.rg_graph_export.argtypes = [vp, vp, vp, vp, vp]
    L.rg_frontier_workspace_bytes.argtypes = [i32, i32, i32]
    L.rg_frontier_workspace_bytes.restype = sz
    L.rg_frontier_create.argtypes = [i32, i32, i32, vp, sz, C.POINTER(vp)]
    L.rg_frontier_destroy.argtypes = [vp]
    L.rg_frontier_reset.argtypes = [vp, vp, vp]
    L.rg_frontier_reset_nodes.argtypes = [vp, vp, i64, vp]
    L.rg_frontier_expand.argtypes = [vp, vp, C.POINTER(i64), vp]
    L.rg_frontier_expand_async.argtypes = [vp, vp, vp]
    L.rg_frontier_expand_nodes_async.argtypes = [vp, vp, vp, vp, vp]
    L.rg_frontier_set_edge_hint.argtypes = [vp, i64]
    L.rg_attn_tables.argtypes = [i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rg_frontier_count_ptr.argtypes = [vp]
    L.rg_frontier_count_ptr.restype = vp
    L.rg_frontier_level_counts.argtypes = [vp, C.POINTER(i64), vp]
    L.rg_frontier_nodes.argtypes = [vp, vp, vp, vp, vp]
    L.rg_frontier_edges_scratch_bytes.argtypes = [i64]
    L.rg_frontier_edges_scratch_bytes.restype = sz
    L.rg_frontier_edges.argtypes = [vp, vp, i32, vp, i64, vp, vp, vp, vp]
    L.rg_layer_fwd_scratch_bytes.argtypes = [vp, vp, i32]
    L.rg_layer_fwd_scratch_bytes.restype = sz
    L.rg_layer_fwd.argtypes = [vp, vp, i32, i64, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, sz, i32, vp]
    L.rg_layer_fwd_plan.argtypes = [vp, vp, i32, i64, i64, i64, i32]
    L.rg_tlayer_fwd.argtypes = [vp, vp, i32, i64, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, sz, vp]
    L.rg_xlayer_fwd.argtypes = [vp, vp, i32, i64, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, sz, vp]
    L.rg_frontier_set_window.argtypes = [vp, vp, vp, i32]
    L.rg_layer_bwd_scratch_bytes.argtypes = [vp, vp, i32, i32]
    L.rg_layer_bwd_scratch_bytes.restype = sz
    L.rg_layer_bwd.argtypes = [vp, vp, i32, i64, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32,
                               vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.rg_tlayer_bwd_scratch_bytes.argtypes = [vp, vp, i32, i32]
    L.rg_tlayer_bwd_scratch_bytes.restype = sz
    L.rg_tlayer_bwd.argtypes = [vp, vp, i32, i64, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32,
                                vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.rg_xlayer_bwd.argtypes = [vp, vp, i32, i64, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32,
                                vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.rg_dense_fwd_supported.argtypes = [i32, i32]
    L.rg_dense_fwd.argtypes = [i64, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, i64, vp]
    L.rg_dense_scratch_bytes.argtypes = [i32, i32]
    L.rg_dense_scratch_bytes.restype = i64
    L.rg_dense_fwd_dev.argtypes = [i64, vp, i64, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, i32, vp, i64, vp]
    L.rg_dense_train_fwd.argtypes = [i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rg_dense_train_fwd_as.argtypes = [i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    L.rg_rows_addmm.argtypes = [vp, i64, vp, i64, i32, vp, i32, i64, vp, i64, vp]
    L.rg_dense_train_bwd.argtypes = [i64, i32, vp, vp, vp, vp, C.c_float, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rg_dense_train_bwd2.argtypes = [i64, i32, vp, vp, vp, vp, C.c_float, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rg_rank.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.rg_split3_roundtrip.argtypes = [vp, i64, i32, vp, vp, vp]
    L.rg_gram_tn_scratch_bytes.argtypes = [i32, i32]
    L.rg_gram_tn_scratch_bytes.restype = sz
    L.rg_gram_tn.argtypes = [vp, i64, i32, vp, i64, i32, i64, vp, vp, vp, sz, vp]
    L.rg_split3_product_check.argtypes = [i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise NativeError(lib().rg_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array; None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
