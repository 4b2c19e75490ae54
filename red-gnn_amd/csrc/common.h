// Shared internals of libredgnn.so (not part of the C-ABI; the ABI is include/redgnn.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "redgnn.h"

namespace rg {

void set_error(const char* fmt, ...);

#define RG_CHECK(cond, ...)                 \
  do {                                      \
    if (!(cond)) {                          \
      ::rg::set_error(__VA_ARGS__);         \
      return 1;                             \
    }                                       \
  } while (0)

#define RG_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      ::rg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                      __LINE__);                                                      \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

#define RG_LAUNCH_CHECK() RG_HIP(hipGetLastError())

// Device fills and small device-to-device copies as plain kernels.  hipMemsetAsync / hipMemcpyAsync would do, except inside a
// captured hipGraph: with ROCm 7.2 a graph holding several memset nodes zeroes correctly on its first launch only (later
// launches left stale bitmap words and wrote address-like garbage into the 1 KB counter block; tools/probe_graph.py), so
// every fill that can end up in a graph goes through these.  p 4-byte aligned, bytes a multiple of 4.
int zero_async(void* p, size_t bytes, hipStream_t s);
int zero2_async(void* p1, size_t bytes1, void* p2, size_t bytes2, hipStream_t s);   // two 16-B aligned ranges, one launch
int copy_words_async(void* dst, const void* src, int n_words, hipStream_t s);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- exclusive scan (device) --------------------------------------------------------------
// out[i] = sum_{j<i} f(in[j]);  *total_dev = sum of all.  `POPC` selects f = popcount.
// scratch: int32 device buffer of scan_scratch_elems(n) elements.
size_t scan_scratch_elems(int64_t n);
int scan_exclusive(const uint32_t* in, int32_t* out, int64_t n, bool popc, int32_t* total_dev,
                   int32_t* scratch, hipStream_t s);

}  // namespace rg

// best-fit placement of length-sorted virtual rows into packs (graph.hip; shared by the host and the device graph builders)
#include <vector>
namespace rg { int place_rows_best_fit(const int4* vrows, int32_t n_vrows, std::vector<int4>* place, std::vector<int32_t>* pack_nrows); }

// ---- handles ----------------------------------------------------------------------------------
// Virtual rows: the CSR rows of one direction cut into segments of at most RG_VROW_MAX entries and
// sorted by length (descending).  Neighbouring work items then have similar trip counts (no idle
// lane groups next to a hub) and a hub of in-degree 17k becomes 133 independent items.
// rows[i] = {entity, first entry, length, slot}: slot = -1 for an entity kept whole, else the index of
// this segment's partial sum inside the query's `n_slots` partial rows.
// split[i] = {entity, first slot, number of segments, 0}.
constexpr int RG_VROW_MAX = 128;   // must stay <= 255 (walk.h packs the length into 8 bits)
struct rg_vrows {
  int32_t n = 0, n_split = 0, n_slots = 0;
  int4* rows = nullptr;
  int4* split = nullptr;
};

// Packs of the word-parallel forward walk (layer_fwd_wp.hip): the virtual rows of the CSR-by-tail bin-packed into groups of
// RG_PACK entries, so that one wave covers a pack with RG_PACK/64 entries per lane and takes 32 queries at a time from the
// entity-major frontier bitmaps.  ent[p*RG_PACK + e] = {packed (rel << 20 | head) or -1 (padding), index of the entry's row
// inside the pack}; rows of a pack are contiguous in e, in CSR order.  pack[p] = {first row in `rows`, number of rows, slot of
// the partial sum if the pack is one segment of a cut row (then it is the pack's only row) else -1, that row's entity};
// rows[] = {entity, slot or -1}.
constexpr int RG_PACK = 128;
struct rg_packs {
  int32_t n = 0;
  int2* ent = nullptr;
  int4* pack = nullptr;
  int2* rows = nullptr;
};

struct rg_graph {
  void* arena = nullptr;     // the one device allocation all arrays below are slices of
  int32_t n_ent = 0, n_rel = 0;
  int32_t n_rela_rows = 0;   // rows of the relation table: 2*n_rel+1 (static), n_rel_total+1 (temporal)
  int32_t n_time = 0;        // temporal graphs: number of time ids (0 = static graph)
  int32_t* in_time = nullptr;  // temporal graphs: time id of every CSR-by-tail entry
  int32_t* out_time = nullptr; // ... and of every CSR-by-head entry (backward)
  int64_t n_fact = 0;
  int32_t max_in_deg = 0, max_out_deg = 0;
  // CSR by head: out_ptr[n_ent+1], out_rt[n_fact] = {rel, tail}
  int32_t* out_ptr = nullptr;
  int2* out_rt = nullptr;
  // CSR by tail: in_ptr[n_ent+1], in_hr[n_fact] = {head, rel}
  int32_t* in_ptr = nullptr;
  int2* in_hr = nullptr;
  // packed CSR-by-tail entries (rel << 20 | head), present when n_ent <= 2^20 and 2*n_rel+1 <= 2^12:
  // halves the structure bytes every query streams through its XCD's L2
  uint32_t* in_pk = nullptr;
  uint32_t* out_pk = nullptr;   // same packing for the CSR-by-head: (rel << 20 | tail)
  rg_vrows in_vr, out_vr;
  rg_packs in_pk_packs;         // static graphs with packed entries only (n = 0 otherwise)
  // CSR by relation (rows = relation ids): rel_ht[n_fact] = {head, tail}; its virtual rows carry the relation id in the
  // entity field.  Used by the backward's relation-gradient pass (one partial row per 128 edges instead of per edge).
  int32_t* rel_ptr = nullptr;
  int2* rel_ht = nullptr;
  rg_vrows rel_vr;
  // temporal graphs: time id of every CSR-by-relation entry, and a CSR by time id (rows = time ids; time_ht = {head, tail},
  // time_rel = relation): the temporal backward's table gradients are summed per 128-edge segment of one relation / one
  // time id (one row add per segment instead of 2*d float atomics per edge)
  int32_t* rel_tm = nullptr;
  int32_t* time_ptr = nullptr;
  int2* time_ht = nullptr;
  int32_t* time_rel = nullptr;
  rg_vrows time_vr;
};

// Frontier state, all inside the caller's workspace.
//   bitsT[2]   : entity-major visited bitmaps  uint32 [n_ent][BW]   (BW = ceil(B/32)); ping-pong
//   bm[level]  : batch-major packed {word, exclusive popcount prefix} int2 [B][W]  (W = ceil(n_ent/32));
//                prefix runs over the flattened [B][W] array, so rank(b,e) = prefix + popc(word below e)
//                is the node id in the reference's (batch, entity) order.
constexpr int RG_MAX_LEVELS = 16;
constexpr int RG_QSTRIDE = 32;       // ints between work-queue heads
constexpr size_t RG_QUEUE_BYTES = 8 * RG_QSTRIDE * sizeof(int32_t);
struct rg_frontier {
  int32_t n_ent = 0, B = 0, BW = 0, W = 0, n_levels = 0;
  uint32_t* bitsT[2] = {nullptr, nullptr};
  int2* bm[RG_MAX_LEVELS] = {};
  uint32_t* words_tmp = nullptr;     // [B][W] batch-major words before packing
  int32_t* prefix_tmp = nullptr;     // [B][W]
  int32_t* scan_scratch = nullptr;
  int32_t* counters = nullptr;       // device: [0]=N (int32), [1]=error flag, [2..3]=E (uint64), [4]=N of level 0, [64..]=per-level snapshots
  // per-query data-row windows [win_lo[b], win_hi[b]) of the extrapolation setting (rg_frontier_set_window; caller-owned device arrays,
  // NULL = none) and the number of data rows (edges whose row id is >= n_data are the always-valid self-loops)
  const int32_t* win_lo = nullptr;
  const int32_t* win_hi = nullptr;
  int32_t win_n_data = 0;
  mutable bool queues_clean = false; // the heads are known to be zero on the stream (the hop's last kernel cleared them): the next walk
                                     // launch skips its own clearing launch
  int32_t* queues = nullptr;         // device: the 8 per-XCD work-queue heads of the walks, RG_QSTRIDE ints apart (one 128-B line each:
                                     // returning atomics to ONE line serialise at ~50 per microsecond chip-wide, measured on the word-parallel
                                     // walk, whatever the number of distinct words in it)
  int64_t* counts_pinned = nullptr;  // host pinned [4]
  int level = -1;                    // newest level (absolute, not modulo)
  int tcur = 0;                      // which bitsT holds the newest level
  int64_t n_nodes[RG_MAX_LEVELS] = {};  // per slot
  int64_t n_edges = 0;
  int64_t edge_hint = -1;            // the caller's expectation of n_edges while it is unknown on the host (after rg_frontier_expand_async)
  const int2* bm_of(int lvl) const { return bm[lvl % n_levels]; }
};
