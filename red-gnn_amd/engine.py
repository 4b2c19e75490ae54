"""Thin host objects over the C-ABI handles: ``Graph`` (rg_graph) and ``Frontier`` (rg_frontier).

PyTorch is used here for device memory and the current HIP stream only; every computation
below is a call into libredgnn.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


# tests set this to 1 (per-query walk) or 2..5 (word-parallel, 32/16/8/4 queries per item) to force rg_layer_fwd's edge walk; 0 = the
# library picks from the hop's sizes
FORCE_WALK = 0

# bench.py sets these to lists to collect (start_event, end_event, n_edges, n_new) per rg_layer_fwd launch and
# (start_event, end_event, n_rows) per rg_dense_fwd launch
KERNEL_EVENTS = None
DENSE_EVENTS = None
# ... and (start_event, end_event, n_edges, n_old) per rg_layer_bwd call (its kernels: layer_bwd_kernel, bwd_combine_kernel, drel_kernel)
BWD_EVENTS = None


def _require_gpu(device):
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.NativeError("red_gnn_amd runs on an MI355X (device 'cuda'); got device %r. "
                               "There is no CPU path in the product — the CPU restatement lives in oracle/ for tests only."
                               % (device,))
    if not torch.cuda.is_available():
        raise _lib.NativeError("red_gnn_amd needs an MI355X: no ROCm device is visible to this process. The HIP path has "
                               "no CPU fallback (the CPU restatement in oracle/ is test infrastructure only).")
    return device


_GRAPH_SERIAL = [0]


def _next_serial():
    _GRAPH_SERIAL[0] += 1
    return _GRAPH_SERIAL[0]


class Graph:
    """Device-resident KG (inverse + identity rows added, CSR by head and by tail).
    Replaces load_data.py:69-81 (double_triple + load_graph).  ``serial`` is unique per object for the life of the process (caches key
    on it: id() of a collected graph can come back)."""

    def __init__(self, n_ent, n_rel, triples, add_inverse=True, device="cuda"):
        self.serial = _next_serial()
        self.device = _require_gpu(device)
        trip = np.ascontiguousarray(np.asarray(triples, dtype=np.int32).reshape(-1, 3))
        self.n_ent, self.n_rel = int(n_ent), int(n_rel)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().rg_graph_create(self.n_ent, self.n_rel, _lib.ptr(trip), len(trip),
                                                  1 if add_inverse else 0, C.byref(h)))
        self.handle = h
        self.n_fact = int(_lib.lib().rg_graph_n_fact(h))

    @classmethod
    def from_device(cls, n_ent, n_rel, triples_dev, add_inverse=True):
        """The same graph built on the device from a device int32 [n,3] tensor (rg_graph_create_device): the per-epoch rebuild of
        shuffle_train without a host round trip of the triples."""
        self = cls.__new__(cls)
        self.serial = _next_serial()
        self.device = _require_gpu(triples_dev.device)
        assert triples_dev.dtype == torch.int32 and triples_dev.dim() == 2 and triples_dev.shape[1] == 3
        trip = triples_dev.contiguous()
        self.n_ent, self.n_rel = int(n_ent), int(n_rel)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().rg_graph_create_device(self.n_ent, self.n_rel, _lib.ptr(trip), trip.shape[0], 1 if add_inverse else 0,
                                                         _lib.stream_ptr(), C.byref(h)))
        self.handle = h
        self.n_fact = int(_lib.lib().rg_graph_n_fact(h))
        return self

    def export_packs(self):
        """(vrows [n_vrows,4], ent [n_packs*128,2], pack [n_packs,4], rows [n_vrows,2]) of the word-parallel walk as numpy — for tests."""
        L = _lib.lib()
        n_packs, n_vrows = C.c_int32(), C.c_int32()
        _lib.check(L.rg_graph_export_packs(self.handle, C.byref(n_packs), C.byref(n_vrows), None, None, None, None))
        vr = np.empty((n_vrows.value, 4), np.int32)
        ent = np.empty((n_packs.value * 128, 2), np.int32)
        pack = np.empty((n_packs.value, 4), np.int32)
        rows = np.empty((n_vrows.value, 2), np.int32)
        _lib.check(L.rg_graph_export_packs(self.handle, None, None, _lib.ptr(ent), _lib.ptr(pack), _lib.ptr(rows), _lib.ptr(vr)))
        return vr, ent, pack, rows

    def export(self):
        """(out_ptr, out_rel_tail[n_fact,2], in_ptr, in_head_rel[n_fact,2]) as numpy — for tests."""
        op = np.empty(self.n_ent + 1, np.int32)
        ip = np.empty(self.n_ent + 1, np.int32)
        ort = np.empty((self.n_fact, 2), np.int32)
        ihr = np.empty((self.n_fact, 2), np.int32)
        _lib.check(_lib.lib().rg_graph_export(self.handle, _lib.ptr(op), _lib.ptr(ort), _lib.ptr(ip), _lib.ptr(ihr)))
        return op, ort, ip, ihr

    def close(self):
        if getattr(self, "handle", None) is not None and _lib._lib is not None:
            _lib.lib().rg_graph_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TemporalGraph(Graph):
    """Device-resident quadruple graph (head, rel, tail, time id) of T-RED-GNN (Temporal/interpolation/graph.py:34-49),
    used as given: the reference's array already holds the identity rows with the sentinel timestamp."""

    def __init__(self, n_ent, n_rela_rows, n_time, quads, device="cuda", exclude=None):
        """``exclude``: row indices of ``quads`` to leave out (the training mode's np.delete, done while the rows are read)."""
        self.serial = _next_serial()
        self.device = _require_gpu(device)
        q = quads if (isinstance(quads, np.ndarray) and quads.dtype == np.int32 and quads.flags.c_contiguous and quads.ndim == 2) \
            else np.ascontiguousarray(np.asarray(quads, dtype=np.int32).reshape(-1, 4))
        self.n_ent, self.n_rel, self.n_rela_rows, self.n_time = int(n_ent), 0, int(n_rela_rows), int(n_time)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            if exclude is None:
                _lib.check(_lib.lib().rg_tgraph_create(self.n_ent, self.n_rela_rows, self.n_time, _lib.ptr(q), len(q), C.byref(h)))
            else:
                ex = np.ascontiguousarray(np.asarray(exclude, dtype=np.int64).reshape(-1))
                _lib.check(_lib.lib().rg_tgraph_create_excluding(self.n_ent, self.n_rela_rows, self.n_time, _lib.ptr(q), len(q),
                                                                 _lib.ptr(ex), len(ex), C.byref(h)))
        self.handle = h
        self.n_fact = int(_lib.lib().rg_graph_n_fact(h))


class Frontier:
    """Per-batch visited-set state (levels of (batch, entity) node sets) in a torch-owned workspace.
    Replaces the state threaded through RED_GNN_trans.forward / DataLoader.get_neighbors
    (models.py:73-78, load_data.py:106-131)."""

    def __init__(self, n_ent, batch, n_levels=2, device="cuda"):
        self.device = _require_gpu(device)
        L = _lib.lib()
        self.n_ent, self.batch, self.n_levels = int(n_ent), int(batch), int(n_levels)
        nbytes = L.rg_frontier_workspace_bytes(self.n_ent, self.batch, self.n_levels)
        if nbytes == 0:
            raise _lib.NativeError("bad frontier shape n_ent=%d batch=%d n_levels=%d" % (n_ent, batch, n_levels))
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        aligned = (base + 255) // 256 * 256
        h = C.c_void_p()
        _lib.check(L.rg_frontier_create(self.n_ent, self.batch, self.n_levels, C.c_void_p(aligned), nbytes, C.byref(h)))
        self.handle = h
        self._counts = (C.c_int64 * 4)()
        self.level = -1
        self.n_new = self.n_old = self.n_edges = 0
        self.generation = 0      # bumped by every reset: the level bitmaps of an older generation are gone
        self.leases = 0          # live FrontierLease objects (autograd graphs whose backward still reads the level bitmaps)

    def reset(self, q_sub):
        """Level 0 = {(b, q_sub[b])} (models.py:73).  q_sub: int32 device tensor [batch]."""
        assert q_sub.dtype == torch.int32 and q_sub.is_cuda and q_sub.numel() == self.batch
        _lib.check(_lib.lib().rg_frontier_reset(self.handle, _lib.ptr(q_sub), _lib.stream_ptr()))
        self.level, self.n_new, self.n_old, self.n_edges = 0, self.batch, 0, 0
        self.generation += 1

    def reset_nodes(self, nodes):
        """Level 0 = an arbitrary node set, int32 device tensor [n,2] (batch, entity)."""
        assert nodes.dtype == torch.int32 and nodes.is_cuda and nodes.dim() == 2 and nodes.shape[1] == 2
        nodes = nodes.contiguous()
        _lib.check(_lib.lib().rg_frontier_reset_nodes(self.handle, _lib.ptr(nodes), nodes.shape[0], _lib.stream_ptr()))
        self.level, self.n_new, self.n_old, self.n_edges = 0, nodes.shape[0], 0, 0
        self.generation += 1

    def set_window(self, win_lo, win_hi, n_data):
        """Per-query data-row windows of the extrapolation setting (rg_frontier_set_window); int32 device tensors [batch], kept alive here."""
        if win_lo is None:
            self._window = None
            _lib.check(_lib.lib().rg_frontier_set_window(self.handle, None, None, 0))
            return
        assert win_lo.dtype == torch.int32 and win_hi.dtype == torch.int32 and win_lo.is_cuda and win_lo.numel() == self.batch == win_hi.numel()
        self._window = (win_lo.contiguous(), win_hi.contiguous())
        _lib.check(_lib.lib().rg_frontier_set_window(self.handle, _lib.ptr(self._window[0]), _lib.ptr(self._window[1]), int(n_data)))

    def expand(self, graph):
        """One hop.  Returns (n_new, n_edges, n_old); synchronises the stream once."""
        _lib.check(_lib.lib().rg_frontier_expand(self.handle, graph.handle, self._counts, _lib.stream_ptr()))
        self.n_new, self.n_edges, self.n_old, self.level = (int(self._counts[i]) for i in range(4))
        return self.n_new, self.n_edges, self.n_old

    def expand_async(self, graph):
        """One hop with no read-back (rg_frontier_expand_async): nothing synchronises; the sizes stay on the device
        (count_ptr, level_counts) and self.n_new / n_old / n_edges are unknown (-1) until level_counts() is called."""
        _lib.check(_lib.lib().rg_frontier_expand_async(self.handle, graph.handle, _lib.stream_ptr()))
        self.level += 1
        self.n_new = self.n_old = self.n_edges = -1

    def expand_nodes_async(self, graph, nodes, prev, edge_hint=None):
        """expand_async + nodes_into in one library call (rg_frontier_expand_nodes_async: one fused launch for small batches)."""
        _lib.check(_lib.lib().rg_frontier_expand_nodes_async(self.handle, graph.handle, _lib.ptr(nodes), _lib.ptr(prev), _lib.stream_ptr()))
        if edge_hint is not None and edge_hint >= 0:      # what an eager run of the same shape counted: tunes the next layer call's tickets
            _lib.check(_lib.lib().rg_frontier_set_edge_hint(self.handle, int(edge_hint)))
        self.level += 1
        self.n_new = self.n_old = self.n_edges = -1

    def count_ptr(self):
        """Device address of the newest level's node count (int32), for dense_fwd_dev."""
        return C.c_void_p(_lib.lib().rg_frontier_count_ptr(self.handle))

    def level_counts(self):
        """[(N_l, E_l) for l = 0..level] read back from the device (synchronises the stream)."""
        buf = (C.c_int64 * 32)()
        _lib.check(_lib.lib().rg_frontier_level_counts(self.handle, buf, _lib.stream_ptr()))
        return [(int(buf[2 * l]), int(buf[2 * l + 1])) for l in range(self.level + 1)]

    def nodes_into(self, nodes, prev):
        """rg_frontier_nodes into caller-owned buffers of capacity batch * n_ent rows (no size needed on the host)."""
        _lib.check(_lib.lib().rg_frontier_nodes(self.handle, _lib.ptr(nodes), _lib.ptr(prev), None, _lib.stream_ptr()))

    def nodes(self, want_prev=True, want_old_new=True):
        """(nodes int32 [n_new,2] sorted, prev_idx int32 [n_new] or None, old_nodes_new_idx int32 [n_old] or None)."""
        nodes = torch.empty((self.n_new, 2), dtype=torch.int32, device=self.device)
        prev = torch.empty(self.n_new, dtype=torch.int32, device=self.device) if want_prev else None
        old_new = (torch.empty(self.n_old, dtype=torch.int32, device=self.device)
                   if (want_old_new and self.level > 0) else None)
        _lib.check(_lib.lib().rg_frontier_nodes(self.handle, _lib.ptr(nodes), _lib.ptr(prev), _lib.ptr(old_new),
                                                _lib.stream_ptr()))
        return nodes, prev, old_new

    def edges(self, graph, nodes_new, level=None):
        """Materialised (batch, head, rel, tail, old_idx, new_idx) int32 [E,6] + row_ptr [n_new+1]
        of hop level-1 -> level (destination-segmented)."""
        level = self.level if level is None else level
        n_new = nodes_new.shape[0]
        L = _lib.lib()
        scratch = torch.empty(L.rg_frontier_edges_scratch_bytes(n_new), dtype=torch.uint8, device=self.device)
        row_ptr = torch.empty(n_new + 1, dtype=torch.int32, device=self.device)
        # pass 1: counts only -> E
        _lib.check(L.rg_frontier_edges(self.handle, graph.handle, level, _lib.ptr(nodes_new), n_new, None,
                                       _lib.ptr(row_ptr), _lib.ptr(scratch), _lib.stream_ptr()))
        n_e = int(row_ptr[-1].item())
        edges = torch.empty((n_e, 6), dtype=torch.int32, device=self.device)
        _lib.check(L.rg_frontier_edges(self.handle, graph.handle, level, _lib.ptr(nodes_new), n_new, _lib.ptr(edges),
                                       _lib.ptr(row_ptr), _lib.ptr(scratch), _lib.stream_ptr()))
        return edges, row_ptr

    def scratch(self, nbytes):
        """A reusable device scratch buffer of at least nbytes (kernels' partial-sum space)."""
        buf = getattr(self, "_scratch", None)
        if buf is None or buf.numel() < nbytes:
            buf = self._scratch = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return buf

    def close(self):
        if getattr(self, "handle", None) is not None and _lib._lib is not None:
            _lib.lib().rg_frontier_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrontierLease:
    """Held by the autograd contexts of one grad-enabled forward: while it lives, the frontier's level bitmaps are still
    needed by that graph's backward (rg_layer_bwd / rg_tlayer_bwd revisit every hop), so the per-shape caches hand out another
    Frontier instead of resetting this one (two forwards before the first backward: gradient accumulation, two losses).
    ``check()`` in backward turns any remaining misuse (a reset behind the lease's back) into a clear error."""
    __slots__ = ("frontier", "generation", "released")

    def __init__(self, frontier):
        self.frontier, self.generation, self.released = frontier, frontier.generation, False
        frontier.leases += 1

    def release(self):
        """Called when the first hop's backward has run (the last one of a backward pass): autograd frees saved tensors then, but
        keeps the contexts - and this object - for as long as the forward's output tensor lives.  A second backward through a
        retained graph is still guarded by check()."""
        if not self.released:
            self.released = True
            self.frontier.leases -= 1

    def check(self):
        if self.frontier.generation != self.generation:
            raise RuntimeError("red_gnn_amd: the frontier of this autograd graph was reset by a later forward (generation %d, now %d); "
                               "its level bitmaps are gone, the backward cannot run" % (self.generation, self.frontier.generation))

    def __del__(self):
        self.release()


class FrontierPool:
    """Frontiers per shape key; ``get`` returns one that no live autograd graph leases (allocating another if needed)."""

    def __init__(self, max_keys=64):      # (8 evaluation lanes x a few batch shapes)
        self.max_keys, self.pool = max_keys, {}

    def get(self, n_ent, batch, n_levels, device):
        # per stream: forwards enqueued on different streams (the evaluator's lanes) run concurrently on the device
        key = (n_ent, batch, n_levels, str(device), torch.cuda.current_stream(device).cuda_stream)
        frs = self.pool.get(key)
        if frs is None:
            if len(self.pool) >= self.max_keys:
                self.pool = {k: [f for f in v if f.leases > 0] for k, v in self.pool.items()}
                self.pool = {k: v for k, v in self.pool.items() if v}
            frs = self.pool[key] = []
        for fr in frs:
            if fr.leases == 0:
                return fr
        fr = Frontier(n_ent, batch, n_levels, device)
        frs.append(fr)
        return fr

    def clear(self):
        self.pool = {}


def layer_fwd(frontier, graph, level, nodes_new, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim):
    """agg [n_new, ld] = fused message passing of hop level-1 -> level (rg_layer_fwd)."""
    ld, ap = hidden.shape[1], a_s.shape[1]
    for t in (hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert rela.shape[1] == ld and a_r.shape[1] == ap and a_q.shape[1] == ap
    n_new = nodes_new.shape[0]
    agg = torch.empty((n_new, ld), dtype=torch.float32, device=hidden.device)
    nbytes = _lib.lib().rg_layer_fwd_scratch_bytes(frontier.handle, graph.handle, ld)
    scratch = frontier.scratch(nbytes)
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(_lib.lib().rg_layer_fwd(frontier.handle, graph.handle, level, n_new,
                                       _lib.ptr(hidden), _lib.ptr(rela), d, ld, _lib.ptr(a_s), _lib.ptr(a_r),
                                       _lib.ptr(a_q), ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim,
                                       _lib.ptr(agg), _lib.ptr(scratch), nbytes, FORCE_WALK, _lib.stream_ptr()))
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append((ev[0], ev[1], frontier.n_edges, n_new))
    return agg


def layer_fwd_plan(frontier, graph, level, n_old, n_new, n_edges, ld):
    """The walk rg_layer_fwd picks for a hop of these sizes (rg_layer_fwd_plan): recorded from an eager forward for graph replay."""
    return int(_lib.lib().rg_layer_fwd_plan(frontier.handle, graph.handle, level, n_old, n_new, n_edges, ld))


def layer_fwd_scratch_bytes(frontier, graph, ld):
    return _lib.lib().rg_layer_fwd_scratch_bytes(frontier.handle, graph.handle, ld)


def layer_fwd_into(frontier, graph, level, n_hint, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim, agg, scratch, walk=1):
    """rg_layer_fwd after expand_async: agg has room for batch * n_ent rows; the sizes are not known on the host, so the caller
    names the walk (recorded from an eager forward of the same shape) and n_hint (> 0) picks the per-query walk's flavour."""
    ld, ap = hidden.shape[1], a_s.shape[1]
    _lib.check(_lib.lib().rg_layer_fwd(frontier.handle, graph.handle, level, max(int(n_hint), 1),
                                       _lib.ptr(hidden), _lib.ptr(rela), d, ld, _lib.ptr(a_s), _lib.ptr(a_r),
                                       _lib.ptr(a_q), ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim,
                                       _lib.ptr(agg), _lib.ptr(scratch), scratch.numel(), FORCE_WALK or max(int(walk), 1), _lib.stream_ptr()))


def tlayer_fwd(frontier, graph, level, n_new, q_time, hidden_dir, rela_dir, time_dir, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim):
    """Temporal fused message passing (rg_tlayer_fwd): agg [n_new, ld]."""
    ld, ap = hidden_dir.shape[1], a_s.shape[1]
    for t in (hidden_dir, rela_dir, time_dir, a_s, a_r, a_q, w_alpha, b_alpha):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert q_time.dtype == torch.int32 and q_time.is_cuda
    agg = torch.empty((n_new, ld), dtype=torch.float32, device=hidden_dir.device)
    nbytes = _lib.lib().rg_layer_fwd_scratch_bytes(frontier.handle, graph.handle, ld)
    scratch = frontier.scratch(nbytes)
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(_lib.lib().rg_tlayer_fwd(frontier.handle, graph.handle, level, n_new, _lib.ptr(q_time), _lib.ptr(hidden_dir),
                                        _lib.ptr(rela_dir), _lib.ptr(time_dir), d, ld, _lib.ptr(a_s), _lib.ptr(a_r), _lib.ptr(a_q),
                                        ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim, _lib.ptr(agg), _lib.ptr(scratch),
                                        nbytes, _lib.stream_ptr()))
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append((ev[0], ev[1], frontier.n_edges, n_new))
    return agg


def xlayer_fwd(frontier, graph, level, n_new, q_time, loop_time, row_time, n_data, hidden_p, rela_p, time_p, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim):
    """Temporal-extrapolation fused message passing (rg_xlayer_fwd): agg [n_new, ld]."""
    ld, ap = hidden_p.shape[1], a_s.shape[1]
    for t in (hidden_p, rela_p, time_p, a_s, a_r, a_q, w_alpha, b_alpha):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    for t in (q_time, loop_time, row_time):
        assert t.dtype == torch.int32 and t.is_cuda and t.is_contiguous()
    agg = torch.empty((n_new, ld), dtype=torch.float32, device=hidden_p.device)
    nbytes = _lib.lib().rg_layer_fwd_scratch_bytes(frontier.handle, graph.handle, ld)
    scratch = frontier.scratch(nbytes)
    ev = None
    if KERNEL_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(_lib.lib().rg_xlayer_fwd(frontier.handle, graph.handle, level, n_new, _lib.ptr(q_time), _lib.ptr(loop_time), _lib.ptr(row_time),
                                        int(n_data), _lib.ptr(hidden_p), _lib.ptr(rela_p), _lib.ptr(time_p), time_p.shape[0], d, ld,
                                        _lib.ptr(a_s), _lib.ptr(a_r), _lib.ptr(a_q), ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim,
                                        _lib.ptr(agg), _lib.ptr(scratch), nbytes, _lib.stream_ptr()))
    if ev is not None:
        ev[1].record()
        KERNEL_EVENTS.append((ev[0], ev[1], frontier.n_edges, n_new))
    return agg


def layer_bwd(frontier, graph, level, nodes_old, hidden, rela, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim, grad_agg):
    """Adjoint of layer_fwd (rg_layer_bwd).  Returns grads of (hidden, rela, a_s, a_r, a_q, w_alpha, b_alpha)."""
    ld, ap = hidden.shape[1], a_s.shape[1]
    grad_agg = grad_agg.contiguous()
    n_old = nodes_old.shape[0]
    dev = hidden.device
    g_h = torch.empty_like(hidden)
    g_as = torch.empty_like(a_s)
    g_rela = torch.zeros_like(rela)
    g_ar = torch.zeros_like(a_r)
    g_aq = torch.empty_like(a_q)
    g_w = torch.zeros(attn_dim, dtype=torch.float32, device=dev)
    g_b = torch.zeros(1, dtype=torch.float32, device=dev)
    nbytes = _lib.lib().rg_layer_bwd_scratch_bytes(frontier.handle, graph.handle, ld, ap)
    scratch = frontier.scratch(nbytes)
    ev = None
    if BWD_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(_lib.lib().rg_layer_bwd(frontier.handle, graph.handle, level, n_old,
                                       _lib.ptr(hidden), _lib.ptr(rela), d, ld, _lib.ptr(a_s), _lib.ptr(a_r),
                                       _lib.ptr(a_q), ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim,
                                       _lib.ptr(grad_agg), _lib.ptr(g_h), _lib.ptr(g_rela), _lib.ptr(g_as),
                                       _lib.ptr(g_ar), _lib.ptr(g_aq), _lib.ptr(g_w), _lib.ptr(g_b), _lib.ptr(scratch), nbytes,
                                       _lib.stream_ptr()))
    if ev is not None:
        ev[1].record()
        BWD_EVENTS.append((ev[0], ev[1], level, n_old))
    return g_h, g_rela, g_as, g_ar, g_aq, g_w, g_b


def tlayer_bwd(frontier, graph, level, n_old, q_time, hidden_dir, rela_dir, time_dir, d, a_s, a_r, a_q, w_alpha, b_alpha, attn_dim,
               grad_agg):
    """Adjoint of tlayer_fwd (rg_tlayer_bwd).  Returns grads of (hidden_dir [3 n_old, ld], rela_dir, time_dir, a_s, a_r, a_q, w_alpha)."""
    ld, ap = hidden_dir.shape[1], a_s.shape[1]
    grad_agg = grad_agg.contiguous()
    dev = hidden_dir.device
    g_hd = torch.empty_like(hidden_dir)
    g_as = torch.empty_like(a_s)
    g_rd = torch.zeros_like(rela_dir)
    g_td = torch.zeros_like(time_dir)
    g_ar = torch.zeros_like(a_r)
    g_aq = torch.empty_like(a_q)
    g_w = torch.zeros(attn_dim, dtype=torch.float32, device=dev)
    nbytes = _lib.lib().rg_tlayer_bwd_scratch_bytes(frontier.handle, graph.handle, ld, ap)
    scratch = frontier.scratch(nbytes)
    _lib.check(_lib.lib().rg_tlayer_bwd(frontier.handle, graph.handle, level, n_old, _lib.ptr(q_time), _lib.ptr(hidden_dir),
                                        _lib.ptr(rela_dir), _lib.ptr(time_dir), d, ld, _lib.ptr(a_s), _lib.ptr(a_r), _lib.ptr(a_q),
                                        ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim, _lib.ptr(grad_agg), _lib.ptr(g_hd),
                                        _lib.ptr(g_rd), _lib.ptr(g_td), _lib.ptr(g_as), _lib.ptr(g_ar), _lib.ptr(g_aq), _lib.ptr(g_w),
                                        _lib.ptr(scratch), nbytes, _lib.stream_ptr()))
    return g_hd, g_rd, g_td, g_as, g_ar, g_aq, g_w


def xlayer_bwd(frontier, graph, level, n_old, q_time, loop_time, row_time, n_data, hidden_p, rela_p, time_p, d, a_s, a_r, a_q, w_alpha,
               b_alpha, attn_dim, grad_agg):
    """Adjoint of xlayer_fwd (rg_xlayer_bwd).  Returns grads of (hidden_p [n_old, ld], rela_p, time_p, a_s, a_r, a_q, w_alpha)."""
    ld, ap = hidden_p.shape[1], a_s.shape[1]
    grad_agg = grad_agg.contiguous()
    dev = hidden_p.device
    g_hp = torch.empty_like(hidden_p)
    g_as = torch.empty_like(a_s)
    g_rp = torch.zeros_like(rela_p)
    g_tp = torch.zeros_like(time_p)
    g_ar = torch.zeros_like(a_r)
    g_aq = torch.empty_like(a_q)
    g_w = torch.zeros(attn_dim, dtype=torch.float32, device=dev)
    nbytes = _lib.lib().rg_tlayer_bwd_scratch_bytes(frontier.handle, graph.handle, ld, ap)
    scratch = frontier.scratch(nbytes)
    _lib.check(_lib.lib().rg_xlayer_bwd(frontier.handle, graph.handle, level, n_old, _lib.ptr(q_time), _lib.ptr(loop_time), _lib.ptr(row_time),
                                        int(n_data), _lib.ptr(hidden_p), _lib.ptr(rela_p), _lib.ptr(time_p), time_p.shape[0], d, ld,
                                        _lib.ptr(a_s), _lib.ptr(a_r), _lib.ptr(a_q), ap, _lib.ptr(w_alpha), _lib.ptr(b_alpha), attn_dim,
                                        _lib.ptr(grad_agg), _lib.ptr(g_hp), _lib.ptr(g_rp), _lib.ptr(g_tp), _lib.ptr(g_as), _lib.ptr(g_ar),
                                        _lib.ptr(g_aq), _lib.ptr(g_w), _lib.ptr(scratch), nbytes, _lib.stream_ptr()))
    return g_hp, g_rp, g_tp, g_as, g_ar, g_aq, g_w


_GRAM_SCRATCH = {}


def gram_tn(g, x, colsum=False):
    """g^T x for node-row matrices g [N, m], x [N, n] (unit-stride columns; rows may be spaced: column blocks of wider buffers work
    without a copy), N in the millions: rg_gram_tn (exact fp32 MFMA products, rows read once, deterministic).  Returns out [m, n], or
    (out, column sums of g [m]) with colsum=True.  Products wider than 192 x 64 are tiled over column blocks."""
    assert g.is_cuda and x.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.shape[0] == x.shape[0]
    assert g.stride(1) == 1 and x.stride(1) == 1
    n_rows, m, n = g.shape[0], g.shape[1], x.shape[1]
    L = _lib.lib()
    out = torch.empty((m, n), dtype=torch.float32, device=g.device)
    cs = torch.empty(m, dtype=torch.float32, device=g.device) if colsum else None
    for m0 in range(0, m, 192):
        mc = min(192, m - m0)
        for n0 in range(0, n, 64):
            nc = min(64, n - n0)
            nbytes = L.rg_gram_tn_scratch_bytes(mc, nc)
            key = (str(g.device), torch.cuda.current_stream(g.device).cuda_stream)
            scratch = _GRAM_SCRATCH.get(key)
            if scratch is None or scratch.numel() < nbytes:
                scratch = _GRAM_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
            tile = out if (mc == m and nc == n) else torch.empty((mc, nc), dtype=torch.float32, device=g.device)
            cs_t = None
            if colsum and n0 == 0:
                cs_t = cs if mc == m else torch.empty(mc, dtype=torch.float32, device=g.device)
            gv, xv = g[:, m0:m0 + mc], x[:, n0:n0 + nc]
            _lib.check(L.rg_gram_tn(C.c_void_p(gv.data_ptr()), g.stride(0), mc, C.c_void_p(xv.data_ptr()), x.stride(0), nc, n_rows,
                                    _lib.ptr(tile), _lib.ptr(cs_t), _lib.ptr(scratch), scratch.numel(), _lib.stream_ptr()))
            if tile is not out:
                out[m0:m0 + mc, n0:n0 + nc] = tile
            if cs_t is not None and cs_t is not cs:
                cs[m0:m0 + mc] = cs_t
    return (out, cs) if colsum else out


_BLAS_CHOICE = [None]


def prefer_blas(n_rows):
    """Pick the GEMM library for the torch-side dense algebra of a training step from the number of node rows.
    hipBLASLt (torch's default on gfx950) searches its heuristics again for every new problem size: 75 us per call
    when the row count changes with every batch, as it does here, against 11 us through rocBLAS (tools/probe_small_gemm.py);
    with the family preset (20 queries per batch) that search was half of the step.  For millions of rows the batched
    weight-gradient products are 1.5x faster through hipBLASLt, and the call overhead no longer matters."""
    choice = "cublas" if n_rows < (1 << 17) else "cublaslt"      # torch's names: cublas = rocBLAS, cublaslt = hipBLASLt
    if _BLAS_CHOICE[0] != choice:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            torch.backends.cuda.preferred_blas_library(choice)
        _BLAS_CHOICE[0] = choice


def dense_supported(d, attn_dim):
    return bool(_lib.lib().rg_dense_fwd_supported(d, attn_dim))


DENSE_PRECISIONS = {"f32": 0, "f16x2": 1, "f16x3": 2}


def dense_scratch(d, precision, device):
    """Scratch of rg_dense_fwd for this width and precision (d = 128 with split products: the weights' split image), or None."""
    nbytes = int(_lib.lib().rg_dense_scratch_bytes(d, DENSE_PRECISIONS[precision]))
    return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None


def dense_fwd(agg, hidden_prev, prev_idx, d, W_h, act, gate, Ws_next=None, attn_dim=0, ap=0, W_final=None, nodes=None,
              n_ent=0, scores_all=None, precision="f32"):
    """Fused W_h + act + GRU step (+ next layer's a_s, + readout) on the matrix cores (rg_dense_fwd).  precision: "f32" = exact
    fp32 MFMA, "f16x3" = exact three-term f16 splits of every operand (fp32 arithmetic on the f16 pipe), "f16x2" = two-term f16
    splits (22 bits, fp32 accumulation); see include/redgnn.h.
    Returns (hidden_new [n, ld], a_s_next [n, ap] or None)."""
    n, ld = agg.shape
    hidden = torch.empty_like(agg)
    a_s = torch.empty((n, ap), dtype=torch.float32, device=agg.device) if Ws_next is not None else None
    c = lambda t: None if t is None else t.detach().contiguous()
    W_h, w_ih, w_hh, b_ih, b_hh = c(W_h), c(gate.weight_ih_l0), c(gate.weight_hh_l0), c(gate.bias_ih_l0), c(gate.bias_hh_l0)
    Ws_next, W_final = c(Ws_next), c(W_final)
    scratch = dense_scratch(d, precision, agg.device)
    ev = None
    if DENSE_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _lib.check(_lib.lib().rg_dense_fwd(n, d, ld, _lib.ptr(agg), _lib.ptr(hidden_prev), _lib.ptr(prev_idx), _lib.ptr(W_h),
                                       {"idd": 0, "relu": 1, "tanh": 2}[act], _lib.ptr(w_ih), _lib.ptr(w_hh), _lib.ptr(b_ih),
                                       _lib.ptr(b_hh), _lib.ptr(Ws_next), attn_dim, ap, _lib.ptr(a_s), _lib.ptr(W_final),
                                       _lib.ptr(nodes), n_ent, _lib.ptr(scores_all), _lib.ptr(hidden), DENSE_PRECISIONS[precision],
                                       _lib.ptr(scratch), 0 if scratch is None else scratch.numel(), _lib.stream_ptr()))
    if ev is not None:
        ev[1].record()
        DENSE_EVENTS.append((ev[0], ev[1], n))
    return hidden, a_s


def dense_fwd_dev(n_cap, count_ptr, agg, hidden_prev, prev_idx, d, W_h, act, gate, hidden_out, Ws_next=None, attn_dim=0, ap=0,
                  a_s_out=None, W_final=None, nodes=None, n_ent=0, scores_all=None, n_hint=0, precision="f32", scratch=None):
    """rg_dense_fwd_dev: the fused dense epilogue with the row count read on the device (buffers of capacity n_cap)."""
    ld = agg.shape[1]
    c = lambda t: None if t is None else t.detach().contiguous()
    W_h, w_ih, w_hh, b_ih, b_hh = c(W_h), c(gate.weight_ih_l0), c(gate.weight_hh_l0), c(gate.bias_ih_l0), c(gate.bias_hh_l0)
    Ws_next, W_final = c(Ws_next), c(W_final)
    if scratch is None:
        scratch = dense_scratch(d, precision, agg.device)
    _lib.check(_lib.lib().rg_dense_fwd_dev(n_cap, count_ptr, int(n_hint), d, ld, _lib.ptr(agg), _lib.ptr(hidden_prev), _lib.ptr(prev_idx),
                                           _lib.ptr(W_h), {"idd": 0, "relu": 1, "tanh": 2}[act], _lib.ptr(w_ih), _lib.ptr(w_hh),
                                           _lib.ptr(b_ih), _lib.ptr(b_hh), _lib.ptr(Ws_next), attn_dim, ap, _lib.ptr(a_s_out),
                                           _lib.ptr(W_final), _lib.ptr(nodes), n_ent, _lib.ptr(scores_all), _lib.ptr(hidden_out),
                                           DENSE_PRECISIONS[precision], _lib.ptr(scratch), 0 if scratch is None else scratch.numel(),
                                           _lib.stream_ptr()))


def dense_train_supported(d, act):
    return ((16 <= d <= 64 and d % 4 == 0) or d == 128) and act in ("idd", "relu", "tanh")


def dense_train_fwd(agg, hidden_prev, prev_idx, W_h, act, gate, mask=None, Ws_next=None):
    """rg_dense_train_fwd: (hidden_new [n,d], x [n,d], gates workspace [n,5d]) for the training step's dense part; with Ws_next
    [attn, d] (attn <= 16) also a_s [n, ap] = hidden_new Ws_next^T, the next layer's hoisted attention projection
    (rg_dense_train_fwd_as), as a fourth element (None without Ws_next)."""
    n, d = agg.shape
    dev = agg.device
    hidden = torch.empty((n, d), dtype=torch.float32, device=dev)
    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    ws = torch.empty((n, 5 * d), dtype=torch.float32, device=dev)
    c = lambda t: None if t is None else t.detach().contiguous()
    common = (n, d, _lib.ptr(c(agg)), _lib.ptr(c(hidden_prev)), _lib.ptr(prev_idx), _lib.ptr(c(W_h)),
              {"idd": 0, "relu": 1, "tanh": 2}[act], _lib.ptr(c(gate.weight_ih_l0)), _lib.ptr(c(gate.weight_hh_l0)),
              _lib.ptr(c(gate.bias_ih_l0)), _lib.ptr(c(gate.bias_hh_l0)), _lib.ptr(c(mask)))
    if Ws_next is None:
        _lib.check(_lib.lib().rg_dense_train_fwd(*common, _lib.ptr(hidden), _lib.ptr(x), _lib.ptr(ws), _lib.stream_ptr()))
        return hidden, x, ws, None
    attn = Ws_next.shape[0]
    ap = (attn + 3) // 4 * 4
    a_s = torch.empty((n, ap), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().rg_dense_train_fwd_as(*common, _lib.ptr(c(Ws_next)), attn, ap, _lib.ptr(hidden), _lib.ptr(x), _lib.ptr(ws),
                                                _lib.ptr(a_s), _lib.stream_ptr()))
    return hidden, x, ws, a_s


def rows_addmm(base, g, W):
    """base + g @ W for node-row matrices base [N, n], g [N, k] (k <= 32; rows may be spaced, unit-stride columns), W [k, n]: one HIP
    pass (rg_rows_addmm).  Returns a new [N, n] tensor."""
    assert base.is_cuda and base.dtype == g.dtype == W.dtype == torch.float32 and base.stride(1) == 1 and g.stride(1) == 1
    n_rows, n = base.shape
    k = W.shape[0]
    assert g.shape[0] == n_rows and g.shape[1] >= k and W.shape[1] == n
    out = torch.empty((n_rows, n), dtype=torch.float32, device=base.device)
    Wc = W.detach().contiguous()
    _lib.check(_lib.lib().rg_rows_addmm(C.c_void_p(base.data_ptr()), base.stride(0), C.c_void_p(g.data_ptr()), g.stride(0), k, _lib.ptr(Wc), n,
                                        n_rows, _lib.ptr(out), n, _lib.stream_ptr()))
    return out


def dense_train_bwd2(g_h, ws, x, mask, keep, act, W_h, w_ih, w_hh, prev_idx, n_old):
    """rg_dense_train_bwd2: (dgi [n,3d], dgh_n [n,d], dpre [n,d], dagg [n,d], g_prev [n_old,d]) - the hidden-side gate gradients as their
    n block only (r and z blocks = dgi's) and the carried state's gradient scattered to the previous frontier's rows by prev_idx."""
    n, d = x.shape
    dev = x.device
    f = lambda rows, cols: torch.empty((rows, cols), dtype=torch.float32, device=dev)
    dgi, dgh_n, dpre, dagg, g_prev = f(n, 3 * d), f(n, d), f(n, d), f(n, d), f(n_old, d)
    c = lambda t: None if t is None else t.detach().contiguous()
    _lib.check(_lib.lib().rg_dense_train_bwd2(n, d, _lib.ptr(c(g_h)), _lib.ptr(ws), _lib.ptr(x), _lib.ptr(c(mask)), float(keep),
                                              {"idd": 0, "relu": 1, "tanh": 2}[act], _lib.ptr(c(W_h)), _lib.ptr(c(w_ih)), _lib.ptr(c(w_hh)),
                                              _lib.ptr(prev_idx), _lib.ptr(dgi), _lib.ptr(dgh_n), _lib.ptr(dpre), _lib.ptr(dagg), _lib.ptr(g_prev),
                                              _lib.stream_ptr()))
    return dgi, dgh_n, dpre, dagg, g_prev


def dense_train_bwd_supported(d):
    return 16 <= d <= 64 and d % 4 == 0


def dense_train_bwd(g_h, ws, x, mask, keep, act, W_h, w_ih, w_hh):
    """rg_dense_train_bwd: (dgi [n,3d], dgh [n,3d], dpre [n,d], dagg [n,d], dh0 [n,d])."""
    n, d = x.shape
    dev = x.device
    f = lambda cols: torch.empty((n, cols), dtype=torch.float32, device=dev)
    dgi, dgh, dpre, dagg, dh0 = f(3 * d), f(3 * d), f(d), f(d), f(d)
    c = lambda t: None if t is None else t.detach().contiguous()
    _lib.check(_lib.lib().rg_dense_train_bwd(n, d, _lib.ptr(c(g_h)), _lib.ptr(ws), _lib.ptr(x), _lib.ptr(c(mask)), float(keep),
                                             {"idd": 0, "relu": 1, "tanh": 2}[act], _lib.ptr(c(W_h)), _lib.ptr(c(w_ih)), _lib.ptr(c(w_hh)),
                                             _lib.ptr(dgi), _lib.ptr(dgh), _lib.ptr(dpre), _lib.ptr(dagg), _lib.ptr(dh0), _lib.stream_ptr()))
    return dgi, dgh, dpre, dagg, dh0


def attn_tables(layers, q_rel, d, ld, attn_dim, ap):
    """Hoisted attention tables of all layers in ONE launch (rg_attn_tables): [(a_r [2R+1, ap], a_q [B, ap], rela padded to ld)] per layer.
    ``layers``: the GNNLayer modules (rela_embed, Wr_attn, Wqr_attn read in place: no stacking copies)."""
    L, dev = len(layers), q_rel.device
    n_rows, B = layers[0].rela_embed.weight.shape[0], q_rel.numel()
    a_r = torch.empty((L, n_rows, ap), dtype=torch.float32, device=dev)
    a_q = torch.empty((L, B, ap), dtype=torch.float32, device=dev)
    rela_p = torch.empty((L, n_rows, ld), dtype=torch.float32, device=dev) if ld != d else None
    arr = lambda ts: (C.c_void_p * L)(*[t.data_ptr() for t in ts])
    for l in layers:
        for t in (l.rela_embed.weight, l.Wr_attn.weight, l.Wqr_attn.weight, l.Wqr_attn.bias):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
    assert q_rel.dtype == torch.int64 and q_rel.is_contiguous()
    _lib.check(_lib.lib().rg_attn_tables(L, n_rows, B, d, ld, attn_dim, ap, arr([l.rela_embed.weight for l in layers]),
                                         arr([l.Wr_attn.weight for l in layers]), arr([l.Wqr_attn.weight for l in layers]),
                                         arr([l.Wqr_attn.bias for l in layers]), _lib.ptr(q_rel), _lib.ptr(a_r), _lib.ptr(a_q),
                                         _lib.ptr(rela_p), _lib.stream_ptr()))
    return [(a_r[i], a_q[i], rela_p[i] if rela_p is not None else layers[i].rela_embed.weight) for i in range(L)]


def rank(scores, ans_ptr, ans_idx, filt_ptr, filt_idx):
    """Filtered ranks (rg_rank) of every answer, fp32 [len(ans_idx)] in (query, answer) order."""
    assert scores.is_cuda and scores.dtype == torch.float32 and scores.is_contiguous()
    B, n_ent = scores.shape
    out = torch.empty(ans_idx.numel(), dtype=torch.float32, device=scores.device)
    _lib.check(_lib.lib().rg_rank(_lib.ptr(scores), B, n_ent, _lib.ptr(ans_ptr), _lib.ptr(ans_idx),
                                  _lib.ptr(filt_ptr), _lib.ptr(filt_idx), _lib.ptr(out), _lib.stream_ptr()))
    return out
