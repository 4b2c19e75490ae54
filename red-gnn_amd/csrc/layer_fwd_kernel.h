// Fused relational message passing, forward.
// Replaces GNNLayer.forward lines Static/transductive/models.py:29-39 — the five E x d gathers,
// the attention MLP on E rows and torch_scatter.scatter(reduce='sum') — with ONE kernel that
// never materialises an edge list or an E x d temporary:
//
//   for every query b and every virtual row (t, segment) of the KG's CSR-by-tail  [dense work space]:
//     if (b,t) is in the new frontier (bitmap test; o = its popcount rank = node id):
//       for every KG in-edge (h, r) -> t of the segment:
//         if (b,h) is in the previous frontier:
//           s      = rank of (b,h)
//           alpha  = sigmoid(w . relu(a_s[s] + a_r[r] + a_q[b]) + b_alpha)
//           acc   += alpha * (hidden[s] + rela[r])
//       agg[o] = acc       (or a partial row when t is a hub cut into segments; combined in order below)
//
// Mapping (wave64), 1024-thread workgroups at <= 64 VGPRs (32 waves per CU hide the gather latency):
//   * a destination item is owned by a group of G lanes, G*4 >= d floats, so a row is one coalesced
//     float4 per lane (d=64: 16 lanes x 16 B = 256 B per row, 4 destinations per wave);
//   * phase 1 runs lane-per-candidate (index math + attention scalar, G candidates at a time);
//     surviving edges are compacted into a per-group LDS strip; phase 2 runs group-per-edge
//     (row gather + FMA), four edges in flight per group;
//   * virtual rows are sorted by length, so the groups of a wave have equal trip counts, and a hub
//     of in-degree 17k is 133 independent items instead of one 17k-long serial loop;
//   * the work space is query-major; XCD x (HW_REG_XCC_ID) serves the x-th eighth of it from its own
//     in-order queue, one block step at a time: a 16-wave workgroup takes 64 consecutive items (DENSE:
//     one per lane group; SPARSE: 1024, one lane tests one item and survivors are compacted), the
//     next ticket is prefetched while the step runs.  An XCD's 64 resident workgroups therefore span
//     at most 4096 consecutive items - a third of ONE query - whose hidden slab (<= n_ent*d*4 B)
//     stays in that XCD's 4 MiB L2 while its destinations gather from it.  Queues only steer speed: a
//     workgroup that finds its queue dry steals from the others, every item is processed once under
//     any placement.
// rela / a_r / w_alpha live in LDS.  Sums run in CSR order: bitwise reproducible.
#pragma once
#include "walk.h"

namespace rgfwd {
namespace {   // internal linkage: the header is instantiated by layer_fwd.hip and tlayer_fwd.hip

struct FwdArgs {
  rg::WalkArgs walk;   // items tested against the NEW frontier (destinations)
  const int2* in_hr;
  const uint32_t* in_pk;
  const int2* bm_old;
  const int2* bm_new;
  int W;
  const float4* hidden;
  const float4* rela;
  int ld4;  // row stride of hidden / rela / agg in float4
  const float4* a_s;
  const float4* a_r;
  const float4* a_q;
  const float* w_alpha;
  const float* b_alpha;
  int attn_dim;
  int n_rela_rows;
  int rela_in_lds;
  float4* agg;
  float4* partial;
  // temporal variant (T-RED-GNN): time id of every CSR entry, query times, rows of W_dir time_embed
  const int32_t* in_time;
  const int32_t* q_time;
  int n_time;
  const float4* time_tab;   // [3 * n_time][ld4]
  // extrapolation variant (Temporal/extrapolation/model_cuda_new_embedding.py): in_time[c] = the data row of the edge (>= n_data: a
  // self-loop), valid for query b only inside its row window [win_lo[b], win_hi[b]); every edge lies in the past: one direction,
  // time_tab row = q_time[b] - row_time[data row]  (self-loops: q_time[b] - loop_time[b]), clamped to n_time - 1
  const int32_t* win_lo;
  const int32_t* win_hi;
  const int32_t* row_time;
  const int32_t* loop_time;
  int n_data;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

constexpr int FWD_BLOCK = 512;   // 3 workgroups per CU = 24 waves at <= 80 VGPRs (no spills)

template <int G, int AP4, bool PACKED, bool DENSE, int KPG, bool RELA_LDS, bool TEMPORAL>
__global__ __launch_bounds__(FWD_BLOCK, TEMPORAL ? 4 : (G >= 32 ? 4 : 6)) void layer_fwd_kernel(FwdArgs A) {
  extern __shared__ float4 lds[];
  constexpr int BLOCK = FWD_BLOCK;
  float4* stage = lds;                                   // [BLOCK] edge tuples {s, r, alpha, -}
  float4* ar_l = lds + BLOCK;                            // [n_rela_rows][AP4]
  float4* w_l = ar_l + A.n_rela_rows * AP4;              // [AP4]
  float4* rela_l = w_l + AP4;                            // [n_rela_rows][G] (optional)
  int4* recs = reinterpret_cast<int4*>(rela_l + (RELA_LDS ? A.n_rela_rows * G : 0));   // [BLOCK] (SPARSE only)

  for (int i = threadIdx.x; i < A.n_rela_rows * AP4; i += BLOCK) ar_l[i] = A.a_r[i];
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if constexpr (RELA_LDS) {
    for (int i = threadIdx.x; i < A.n_rela_rows * G; i += BLOCK) {
      const int r = i / G, c = i - r * G;
      rela_l[i] = c < A.ld4 ? A.rela[(int64_t)r * A.ld4 + c] : f4zero();
    }
  }
  __syncthreads();
  const float b_alpha = A.b_alpha[0];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane_g = lane & (G - 1), gi_w = lane / G;
  float4* my_stage = stage + wv * 64 + gi_w * G;
  const int gshift = lane & ~(G - 1);
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const bool row_lane = lane_g < A.ld4;
  const int lane_c = row_lane ? lane_g : A.ld4 - 1;   // loads never branch: idle lanes re-read the last float4

  // one destination item: sum over its candidate in-edges [beg, end) for query b
  auto run_item = [&](int beg, int end, int b) -> float4 {
    const float4* aq_p = A.a_q + (int64_t)b * AP4;
    float4 aq[AP4 <= 2 ? AP4 : 1];
    if constexpr (AP4 <= 2) {
#pragma unroll
      for (int k = 0; k < AP4; ++k) aq[k] = aq_p[k];
    }
    const int2* bm_row = A.bm_old + (int64_t)b * A.W;
    int qt = 0, wlo = 0, whi = 0;
    if constexpr (TEMPORAL) {
      qt = A.q_time[b];
      if (A.win_lo) { wlo = A.win_lo[b]; whi = A.win_hi[b]; }
    }
    float4 acc = f4zero();
    for (int c0 = beg; c0 < end; c0 += G) {
      // ---- phase 1: one candidate in-edge per lane ---------------------------------------------
      const int c = c0 + lane_g;
      bool valid = c < end;
      int s = 0, r = 0, trow = 0;
      float alpha = 0.f;
      if (valid) {
        int hd;
        if constexpr (PACKED) { const uint32_t pk = A.in_pk[c]; hd = pk & 0xFFFFF; r = pk >> 20; }
        else { const int2 hr = A.in_hr[c]; hd = hr.x; r = hr.y; }
        const int2 wp = bm_row[hd >> 5];
        const uint32_t word = (uint32_t)wp.x, bit = hd & 31;
        valid = (word >> bit) & 1u;
        if constexpr (TEMPORAL) {
          if (A.win_lo && valid) {            // the edge's data row must lie inside the query's time window (self-loops always do)
            const int row = A.in_time[c];
            valid = row >= A.n_data || (row >= wlo && row < whi);
          }
        }
        if (valid) {
          s = wp.y + __popc(word & ((1u << bit) - 1u));
          float z = b_alpha;
          // (a 32-wide attention row fully unrolled keeps sixteen loads live: 182 registers, two waves per SIMD; two steps at a time fit
          // 112 and four waves - same order of the sum)
#pragma clang loop unroll_count(AP4 >= 8 ? 2 : AP4)
          for (int k = 0; k < AP4; ++k) {
            const float4 as = A.a_s[(int64_t)s * AP4 + k];
            const float4 ar = ar_l[r * AP4 + k];
            const float4 w = w_l[k];
            float4 q;
            if constexpr (AP4 <= 2) q = aq[k]; else q = aq_p[k];
            z = fmaf(w.x, fmaxf(as.x + ar.x + q.x, 0.f), z);
            z = fmaf(w.y, fmaxf(as.y + ar.y + q.y, 0.f), z);
            z = fmaf(w.z, fmaxf(as.z + ar.z + q.z, 0.f), z);
            z = fmaf(w.w, fmaxf(as.w + ar.w + q.w, 0.f), z);
          }
          alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
          if constexpr (TEMPORAL) {
            // direction-specific linears hoisted per node / relation / |dt|: row = 3*id + dir, dir = past 0 / now 1 / future 2
            // (Temporal/interpolation/model_cuda.py:149-157)
            if (A.win_lo) {
              const int row = A.in_time[c];
              const int delta = qt - (row >= A.n_data ? A.loop_time[b] : A.row_time[row]);
              trow = min(max(delta, 0), A.n_time - 1);
            } else {
              const int dt = A.in_time[c] - qt;
              const int dir = dt > 0 ? 2 : (dt == 0 ? 1 : 0);
              s = s * 3 + dir;
              r = dir * A.n_rela_rows + r;
              trow = dir * A.n_time + (dt < 0 ? -dt : dt);
            }
          }
        }
      }
      const unsigned long long m = (__ballot(valid) >> gshift) & gmask;
      const int cnt = __popcll(m);
      const int pos = __popcll(m & ((1ull << lane_g) - 1ull));
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // previous round's reads are done
      __builtin_amdgcn_wave_barrier();
      if (lane_g >= cnt) my_stage[lane_g] = f4zero();          // pad tuples: alpha = 0, row 0
      if (valid) my_stage[pos] = make_float4(__int_as_float(s), __int_as_float(r), alpha, __int_as_float(trow));
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();

      // ---- phase 2: one edge per group step, 4 row gathers in flight ---------------------------
      for (int k = 0; k < cnt; k += 4) {
        float4 tp[4], hv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tp[u] = my_stage[k + u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          hv[u] = A.hidden[(int64_t)__float_as_int(tp[u].x) * A.ld4 + lane_c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ru = __float_as_int(tp[u].y);
          float4 rv;
          if constexpr (RELA_LDS) rv = rela_l[ru * G + lane_g];
          else rv = A.rela[(int64_t)ru * A.ld4 + lane_c];
          if constexpr (TEMPORAL) {
            const float4 tv = A.time_tab[(int64_t)__float_as_int(tp[u].w) * A.ld4 + lane_c];
            rv.x += tv.x; rv.y += tv.y; rv.z += tv.z; rv.w += tv.w;
          }
          const float al = tp[u].z;
          acc.x = fmaf(al, hv[u].x + rv.x, acc.x);
          acc.y = fmaf(al, hv[u].y + rv.y, acc.y);
          acc.z = fmaf(al, hv[u].z + rv.z, acc.z);
          acc.w = fmaf(al, hv[u].w + rv.w, acc.w);
        }
      }
    }
    return acc;
  };
  auto store_row = [&](int out, float4 acc) {
    if (!row_lane) return;
    if (out >= 0) A.agg[(int64_t)out * A.ld4 + lane_g] = acc;
    else A.partial[(int64_t)(-out - 1) * A.ld4 + lane_g] = acc;
  };
  rg::walk_items<G, DENSE, KPG, BLOCK>(A.walk, recs, [&](const int4& R, bool live) {
    const float4 acc = run_item(R.x, R.x + rg::walk_len(R), R.z);
    if (live) store_row(rg::walk_out(R, A.walk.n_slots), acc);
  });
}

// hubs cut into segments: agg[o] = sum of the segments' partial rows, in segment order.  `written` (word-parallel walk only):
// one byte per partial row, set when the segment had an edge for the query; the others were never written and count as zero.
__global__ void combine_kernel(const int4* __restrict__ split, int n_split, int n_slots, int B, const int2* __restrict__ bm_new,
                               int W, const float4* __restrict__ partial, float4* __restrict__ agg, int ld4,
                               const uint8_t* __restrict__ written) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t item = tid / ld4;
  const int c = (int)(tid - item * ld4);
  if (item >= (int64_t)B * n_split) return;
  const int b = (int)(item / n_split);
  const int4 se = split[item - (int64_t)b * n_split];
  const int2 wp = bm_new[(int64_t)b * W + (se.x >> 5)];
  const uint32_t word = (uint32_t)wp.x, bit = se.x & 31;
  if (!((word >> bit) & 1u)) return;
  const int o = wp.y + __popc(word & ((1u << bit) - 1u));
  const float4* p = partial + ((int64_t)b * n_slots + se.y) * ld4 + c;
  const uint8_t* wr = written ? written + (int64_t)b * n_slots + se.y : nullptr;
  // eight segments per trip: their rows are loaded together (a hub has up to 133 segments; one dependent load per segment made
  // this kernel 0.4 ms of the C2 step), then added in segment order.  The first written segment initialises the sum.
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  bool started = false;
  for (int k0 = 0; k0 < se.z; k0 += 8) {
    float4 v[8];
    bool on[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      on[u] = k0 + u < se.z && (!wr || wr[k0 + u]);
      v[u] = p[(int64_t)min(k0 + u, se.z - 1) * ld4];         // (always in range; used only where on[u])
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (!on[u]) continue;
      if (!started) { acc = v[u]; started = true; }
      else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  agg[(int64_t)o * ld4 + c] = acc;
}

inline int launch_combine(const FwdArgs& A, int B, const rg_vrows& vr, hipStream_t s, const uint8_t* written = nullptr) {
  if (vr.n_split > 0) {
    const int64_t threads = (int64_t)B * vr.n_split * A.ld4;
    hipLaunchKernelGGL(combine_kernel, dim3(rg::ceil_div(threads, 256)), dim3(256), 0, s, vr.split, vr.n_split, vr.n_slots, B,
                       A.bm_new, A.W, A.partial, A.agg, A.ld4, written);
    RG_LAUNCH_CHECK();
  }
  return 0;
}

template <int G, int AP4, bool PACKED, bool DENSE, int KPG, bool RELA_LDS, bool TEMPORAL>
int launch3(const FwdArgs& A, size_t lds, int B, const rg_vrows& vr, hipStream_t s) {
  constexpr int BLOCK = FWD_BLOCK;
  auto kern = layer_fwd_kernel<G, AP4, PACKED, DENSE, KPG, RELA_LDS, TEMPORAL>;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = lds <= 53 * 1024 ? 3 : (lds <= 80 * 1024 ? 2 : 1);
  const int grid = rg::walk_grid(A.walk.n_items, BLOCK, G, DENSE, per_cu, KPG);
  if (!A.walk.queues_clean && rg::zero_async(A.walk.queues, RG_QUEUE_BYTES, s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  return launch_combine(A, B, vr, s);
}

template <int G, int AP4, bool PACKED, bool DENSE, int KPG, bool TEMPORAL>
int launch2k(const FwdArgs& A, int B, const rg_vrows& vr, hipStream_t s) {
  size_t lds = (size_t)(FWD_BLOCK + A.n_rela_rows * AP4 + AP4) * sizeof(float4);
  if (!DENSE) lds += (size_t)FWD_BLOCK * sizeof(int4);
  const size_t rela_bytes = (size_t)A.n_rela_rows * G * sizeof(float4);
  RG_CHECK(lds <= 160 * 1024, "rg_layer_fwd: attention tables need %zu B of LDS (> 160 KiB)", lds);
  if constexpr (!TEMPORAL) {   // the temporal relation table has 3x the rows: left in L2
    if (lds + rela_bytes <= 53 * 1024) return launch3<G, AP4, PACKED, DENSE, KPG, true, false>(A, lds + rela_bytes, B, vr, s);   // 3 blocks per CU
  }
  return launch3<G, AP4, PACKED, DENSE, KPG, false, TEMPORAL>(A, lds, B, vr, s);
}

template <int G, int AP4, bool PACKED, bool DENSE, bool TEMPORAL>
int launch2(const FwdArgs& A, int B, const rg_vrows& vr, int kpg, hipStream_t s) {
  if constexpr (DENSE) {
    if (kpg > 1) return launch2k<G, AP4, PACKED, true, rg::RG_KPG_SHORT, TEMPORAL>(A, B, vr, s);
  }
  return launch2k<G, AP4, PACKED, DENSE, 1, TEMPORAL>(A, B, vr, s);
}

template <int G, int AP4, bool TEMPORAL>
int launch(const FwdArgs& A, int B, const rg_vrows& vr, bool dense, int kpg, hipStream_t s) {
  if (A.in_pk) return dense ? launch2<G, AP4, true, true, TEMPORAL>(A, B, vr, kpg, s) : launch2<G, AP4, true, false, TEMPORAL>(A, B, vr, kpg, s);
  return dense ? launch2<G, AP4, false, true, TEMPORAL>(A, B, vr, kpg, s) : launch2<G, AP4, false, false, TEMPORAL>(A, B, vr, kpg, s);
}

template <int G, bool TEMPORAL>
int launch_ap(const FwdArgs& A, int ap4, int B, const rg_vrows& vr, bool dense, int kpg, hipStream_t s) {
  switch (ap4) {
    case 1: return launch<G, 1, TEMPORAL>(A, B, vr, dense, kpg, s);
    case 2: return launch<G, 2, TEMPORAL>(A, B, vr, dense, kpg, s);
    case 3: return launch<G, 3, TEMPORAL>(A, B, vr, dense, kpg, s);
    case 4: return launch<G, 4, TEMPORAL>(A, B, vr, dense, kpg, s);
    case 8: return launch<G, 8, TEMPORAL>(A, B, vr, dense, kpg, s);
    default: rg::set_error("rg_layer_fwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}


// checks shared by the static and the temporal entry points; fills the common part of FwdArgs
inline int fill_common(const char* who, const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, int32_t d,
                       int32_t ld, int32_t ap, int32_t attn_dim, const void* scratch, size_t scratch_bytes, size_t need, FwdArgs* A) {
  RG_CHECK(g->n_ent == f->n_ent, "%s: graph has %d entities, frontier %d", who, g->n_ent, f->n_ent);
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1, "%s: level %d not resident (current %d, %d kept)",
           who, level, f->level, f->n_levels);
  // after rg_frontier_expand_async the host does not know the level's size: n_new is then only an estimate (> 0) that picks the
  // walk, and agg_out must have room for any outcome (B * n_ent rows)
  RG_CHECK(f->n_nodes[level % f->n_levels] < 0 || n_new == f->n_nodes[level % f->n_levels], "%s: n_new=%lld but level %d has %lld nodes",
           who, (long long)n_new, level, (long long)f->n_nodes[level % f->n_levels]);
  RG_CHECK(d > 0 && ld >= d && ld % 4 == 0 && ld >= 16 && ld <= 256, "%s: d=%d ld=%d (need ld%%4==0, 16<=ld<=256)", who, d, ld);
  RG_CHECK(attn_dim > 0 && ap >= attn_dim && ap % 4 == 0, "%s: attn_dim=%d ap=%d", who, attn_dim, ap);
  RG_CHECK(g->in_vr.n_slots == 0 || (scratch && scratch_bytes >= need), "%s: scratch %zu B < required %zu B", who, scratch_bytes, need);
  RG_CHECK((int64_t)f->B * std::max(g->in_vr.n_slots, 1) < ((int64_t)1 << 31) && g->in_vr.n_slots < (1 << 22),
           "%s: batch * hub segments overflows int32", who);
  const int64_t n_items = (int64_t)f->B * g->in_vr.n;
  RG_CHECK(n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "%s: work space too large for 32-bit queue tickets", who);
  A->walk.n_items = n_items; A->walk.n_vrows = g->in_vr.n; A->walk.n_slots = g->in_vr.n_slots; A->walk.vrows = g->in_vr.rows;
  A->walk.bm_test = f->bm_of(level); A->walk.W = f->W; A->walk.queues = f->queues;
  A->walk.queues_clean = f->queues_clean;      // (left by the hop's last kernel; any walk launch dirties them)
  f->queues_clean = false;
  A->in_hr = g->in_hr; A->in_pk = g->in_pk;
  A->bm_old = f->bm_of(level - 1); A->bm_new = f->bm_of(level); A->W = f->W;
  A->ld4 = ld / 4; A->attn_dim = attn_dim; A->n_rela_rows = g->n_rela_rows; A->rela_in_lds = 0;
  A->in_time = nullptr; A->q_time = nullptr; A->n_time = 0; A->time_tab = nullptr;
  A->win_lo = nullptr; A->win_hi = nullptr; A->row_time = nullptr; A->loop_time = nullptr; A->n_data = 0;
  return 0;
}

template <bool TEMPORAL>
int dispatch(const FwdArgs& A, int ld4, int ap4, int B, const rg_vrows& vr, bool dense, int kpg, hipStream_t s) {
  if (ld4 <= 4) return launch_ap<4, TEMPORAL>(A, ap4, B, vr, dense, kpg, s);
  if (ld4 <= 8) return launch_ap<8, TEMPORAL>(A, ap4, B, vr, dense, kpg, s);
  if (ld4 <= 16) return launch_ap<16, TEMPORAL>(A, ap4, B, vr, dense, kpg, s);
  if (ld4 <= 32) return launch_ap<32, TEMPORAL>(A, ap4, B, vr, dense, kpg, s);
  return launch_ap<64, TEMPORAL>(A, ap4, B, vr, dense, kpg, s);
}

}  // namespace
}  // namespace rgfwd
