// Work distribution shared by the layer forward and backward kernels.
//
// The work space is (query b, virtual row vr) in query-major order; an item is live when (b, entity of vr) is
// in the frontier given by `bm_test`, and then carries R = {first CSR entry, length | (slot+1) << 8, b, node id}:
// node id = popcount rank of (b, entity); slot+1 = 0 for an entity kept whole, else 1 + the index of this
// segment's partial row inside the query's n_slots partial rows (walk_len / walk_part decode it).
//
// XCD x (HW_REG_XCC_ID) serves the x-th eighth of the work space from its own in-order queue, one block step
// at a time: a workgroup takes STEP consecutive items (DENSE: one per lane group; SPARSE: 64 per wave, one lane
// tests one item and survivors are compacted through LDS), the next ticket is prefetched while the step runs.
// An XCD's resident workgroups therefore span a few thousand consecutive items - a fraction of ONE query -
// so the per-query rows they gather from stay in that XCD's 4 MiB L2.  Queues only steer speed: a workgroup
// that finds its queue dry steals from the others; every item is processed exactly once under any placement.
#pragma once
#include "common.h"

namespace rg {

struct WalkArgs {
  int64_t n_items;   // B * n_vrows
  int32_t n_vrows;
  int32_t n_slots;
  const int4* vrows;
  const int2* bm_test;
  int W;
  int32_t* queues;   // 8 heads, RG_QSTRIDE ints apart: item offsets inside each eighth, zeroed before the launch
  bool queues_clean = false;   // host side: the heads are already zero on the stream (rg_frontier::queues_clean)
};

// DENSE walks take KPG items per lane group and block step: 1 for graphs of long rows (C2: 34 entries per row),
// RG_KPG_SHORT for graphs of short rows (WN18RR-like: 5 entries per row), where a block step of one 3-edge item per
// group is all barrier and load latency.  A template value, so the one-item walk keeps its constant-folded form.
constexpr int RG_KPG_SHORT = 8;

__device__ __forceinline__ int walk_len(const int4& R) { return R.y & 255; }
// row index of the item's result: >= 0 node id (whole entity), < 0: -(partial row) - 1
__device__ __forceinline__ int walk_out(const int4& R, int n_slots) {
  const int sp1 = R.y >> 8;
  return sp1 == 0 ? R.w : -(R.z * n_slots + sp1 - 1) - 1;
}

// run_item(const int4& R, bool live) is called by all 64 lanes of a wave with one item per group of G lanes.
// ALWAYS: every item is live (rows that are not tied to an entity of the frontier, e.g. the CSR by relation); R.w is then
// the row's own id (vrows[].x).
template <int G, bool DENSE, int KPG, int BLOCK, bool ALWAYS = false, typename F = void>
__device__ __forceinline__ void walk_items(const WalkArgs& A, int4* recs, F&& run_item) {
  constexpr int GW = 64 / G, WPB = BLOCK / 64;
  constexpr int STEP = DENSE ? WPB * GW * KPG : WPB * 64;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gi_w = lane / G;

  auto test_item = [&](int b, int vr, int4& rec) -> bool {
    const int4 row = A.vrows[vr];
    if constexpr (ALWAYS) { rec = make_int4(row.y, row.z, b, row.x); return true; }
    const int2 wp = A.bm_test[(int64_t)b * A.W + (row.x >> 5)];
    const uint32_t word = (uint32_t)wp.x, bit = row.x & 31;
    if (!((word >> bit) & 1u)) return false;
    const int o = wp.y + __popc(word & ((1u << bit) - 1u));
    rec = make_int4(row.y, row.z | ((row.w + 1) << 8), b, o);
    return true;
  };

  __shared__ int slot[2][4];   // {b0, vr0, count} of the step's first item
  int q = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID: speed only
  // thread 0: turn ticket `off` of queue q into a slot (or steal); count 0 = every queue is dry.  A dry queue is followed by ONE look at
  // all eight heads (independent loads, one round trip) and a ticket from the first queue that still has items: drawing a ticket from each
  // queue in turn was up to seven dependent returning atomics at the tail of every launch (20 us of a 50-query hop)
  auto resolve = [&](int off, int* out) {
    for (;;) {
      const int64_t qs = A.n_items * q / 8, ql = A.n_items * (q + 1) / 8 - qs;
      if (off < ql) {
        const int64_t item = qs + off;
        const int b0 = (int)(item / A.n_vrows);
        out[0] = b0; out[1] = (int)(item - (int64_t)b0 * A.n_vrows); out[2] = (int)min((int64_t)STEP, ql - off);
        return;
      }
      int head[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) head[k] = __hip_atomic_load(&A.queues[k * RG_QSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int best = -1;
#pragma unroll
      for (int j = 7; j >= 1; --j) {
        const int k = (q + j) & 7;
        if (head[k] < A.n_items * (k + 1) / 8 - A.n_items * k / 8) best = k;
      }
      if (best < 0) { out[2] = 0; return; }      // (heads only grow: dry stays dry)
      q = best;
      off = atomicAdd(&A.queues[q * RG_QSTRIDE], STEP);
    }
  };
  if (threadIdx.x == 0) resolve(atomicAdd(&A.queues[q * RG_QSTRIDE], STEP), slot[0]);
  int4* my_recs = recs + wv * 64;
  for (int p = 0;; p ^= 1) {
    __syncthreads();
    const int b0 = slot[p][0], vr0 = slot[p][1], cnt_items = slot[p][2];
    if (cnt_items == 0) break;
    int next_off = 0;
    if (threadIdx.x == 0) next_off = atomicAdd(&A.queues[q * RG_QSTRIDE], STEP);     // prefetch the next ticket

    if constexpr (DENSE) {
#pragma unroll 1
      for (int k = 0; k < KPG; ++k) {          // consecutive items go to different groups: equal lengths side by side
        const int idx = k * (WPB * GW) + wv * GW + gi_w;
        int b = b0, vr = vr0 + idx;
        while (vr >= A.n_vrows) { vr -= A.n_vrows; ++b; }
        int4 R = make_int4(0, 0, 0, 0);
        const bool live = idx < cnt_items && test_item(b, vr, R);
        if (!live) { R.x = 0; R.y = 0; }
        run_item(R, live);
      }
    } else {
      const int idx = wv * 64 + lane;
      int b = b0, vr = vr0 + idx;
      while (vr >= A.n_vrows) { vr -= A.n_vrows; ++b; }
      int4 rec = make_int4(0, 0, 0, 0);
      const bool ok = idx < cnt_items && test_item(b, vr, rec);
      const unsigned long long surv = __ballot(ok);
      const int n_surv = __popcll(surv);
      if (n_surv > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (ok) my_recs[__popcll(surv & ((1ull << lane) - 1ull))] = rec;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < n_surv; j += GW) {
          const bool live = j + gi_w < n_surv;
          const int4 R = live ? my_recs[j + gi_w] : make_int4(0, 0, 0, 0);
          run_item(R, live);
        }
      }
    }
    if (threadIdx.x == 0) resolve(next_off, slot[p ^ 1]);
  }
}

// items per lane group and block step in the dense walk: about 24 CSR entries' worth of rows
static inline int walk_kpg(int64_t n_fact, int32_t n_vrows) {
  return (double)n_fact / std::max(n_vrows, 1) < 12.0 ? RG_KPG_SHORT : 1;
}

// grid for a walk launch
static inline int walk_grid(int64_t n_items, int block, int g, bool dense, int per_cu, int kpg) {
  const int64_t steps = ceil_div(n_items, (int64_t)(block / 64) * (dense ? (64 / g) * kpg : 64));
  return (int)std::max<int64_t>(std::min<int64_t>(steps, 256 * per_cu), 1);
}

}  // namespace rg
