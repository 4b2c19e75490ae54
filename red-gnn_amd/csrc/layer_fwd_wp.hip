// Fused relational message passing, forward, WORD-PARALLEL form: the same sums as layer_fwd_kernel.h
// (GNNLayer.forward, Static/transductive/models.py:29-39 incl. torch_scatter.scatter), enumerated from the SOURCE side of
// the frontier bitmaps instead of testing every candidate in-edge of every destination.
//
// Why: on an expanding hop the previous frontier is sparse while the new one is not.  The per-query walk tests each
// in-edge (h, r) -> t of each live destination (b, t) against the previous frontier; on C2's second hop 88 % of those 420 M
// tests fail.  The reference expands from the frontier's nodes (load_data.py:115-118); this kernel does so bit-parallel,
// 32 queries at a time, and still sums per destination in CSR order (bitwise the per-query walk's results, no atomics):
//
//   item = (query group gq of one bitmap word, pack of <= 128 CSR-by-tail entries = a few whole rows, common.h rg_packs)
//   lane l holds entries l and 64 + l of the pack:  word = bits_old[head][gq]    -> the queries for which the edge is valid
//   for every query bit b set in any lane's word:                              (scalar loop over the set bits only)
//       m = ballot(bit b of word)      -> the valid edges of query b in this pack, in CSR order, grouped by destination row
//       append them to the wave's LDS queue
//   queue full / item done:  phase 1: lane per queued edge: source node id (popcount rank), attention scalar
//                            phase 2: lane group per destination run: row gathers + FMAs, 4 in flight, result row stored
//
// A destination row cut into segments (hubs, > 128 in-edges) is a pack of its own; its partial sum is zero-filled when a
// live query has no edge in the segment, and combine_kernel (layer_fwd_kernel.h) adds the segments up as after the walk.
// Items are group-major and served per XCD from in-order queues (as walk.h): the hidden rows of one group's <= 32 sparse
// frontiers are what an XCD's L2 holds while it sweeps the packs; n_sub splits a word into 2 or 4 groups when they are not.
#include "layer_fwd_wp.h"

namespace rgwp {
namespace {

constexpr int WP_BLOCK = 512, WP_WAVES = WP_BLOCK / 64;
constexpr int QCAP = 256;   // queued edges per wave between flushes; >= RG_PACK, so one query's edges of a pack always fit
constexpr int RS_STRIDE = QCAP + 4;
static_assert(QCAP >= RG_PACK && RG_PACK == 128, "a lane holds exactly two entries of a pack");

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

template <int G, int AP4, bool RELA_LDS>
__global__ __launch_bounds__(WP_BLOCK, 2) void layer_fwd_wp_kernel(WpArgs A) {
  extern __shared__ float4 lds[];
  constexpr int GW = 64 / G;
  float4* stage = lds;                                                          // [WAVES][QCAP] {s | packed entry, r, alpha, key}
  int32_t* run_start = reinterpret_cast<int32_t*>(stage + WP_WAVES * QCAP);     // [WAVES][RS_STRIDE]
  float4* ar_l = reinterpret_cast<float4*>(run_start + WP_WAVES * RS_STRIDE);   // [n_rela_rows][AP4]
  float4* w_l = ar_l + A.n_rela_rows * AP4;                                     // [AP4]
  float4* rela_l = w_l + AP4;                                                   // [n_rela_rows][G] (optional)

  for (int i = threadIdx.x; i < A.n_rela_rows * AP4; i += WP_BLOCK) ar_l[i] = A.a_r[i];
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if constexpr (RELA_LDS) {
    for (int i = threadIdx.x; i < A.n_rela_rows * G; i += WP_BLOCK) {
      const int r = i / G, c = i - r * G;
      rela_l[i] = c < A.ld4 ? A.rela[(int64_t)r * A.ld4 + c] : f4zero();
    }
  }
  const float b_alpha = A.b_alpha[0];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane_g = lane & (G - 1), gi_w = lane / G;
  const bool row_lane = lane_g < A.ld4;
  const int lane_c = row_lane ? lane_g : A.ld4 - 1;      // loads never branch: idle lanes re-read the last float4
  const unsigned long long lt = (1ull << lane) - 1ull;
  float4* st = stage + wv * QCAP;
  int32_t* rs = run_start + wv * RS_STRIDE;
  const int qbits = 32 / A.n_sub;

  auto run_item = [&](long long item) {
    const int grp = (int)(item / A.n_packs);
    const int pi = __builtin_amdgcn_readfirstlane((int)(item - (long long)grp * A.n_packs));
    const int bw = __builtin_amdgcn_readfirstlane(grp / A.n_sub), sub = grp - bw * A.n_sub;
    const uint32_t qmask = A.n_sub == 1 ? 0xFFFFFFFFu : (((1u << qbits) - 1u) << (sub * qbits));
    const int2 e0 = A.ent[(int64_t)pi * RG_PACK + lane], e1 = A.ent[(int64_t)pi * RG_PACK + 64 + lane];
    uint32_t w0 = 0, w1 = 0;
    if (e0.x != -1) w0 = A.bits_old[(int64_t)(e0.x & 0xFFFFF) * A.BW + bw] & qmask;
    if (e1.x != -1) w1 = A.bits_old[(int64_t)(e1.x & 0xFFFFF) * A.BW + bw] & qmask;
    const int4 P = A.pack[pi];
    uint32_t todo = wave_or(w0 | w1);
    if (P.z >= 0) todo |= A.bits_new[(int64_t)P.w * A.BW + bw] & qmask;      // live queries of a cut row's segment
    todo = (uint32_t)__builtin_amdgcn_readfirstlane((int)todo);
    while (todo) {
      // ---- fill: the valid edges of query after query, in (query, CSR order) --------------------------------------------
      int qn = 0;
      while (todo) {
        const int b = __ffs((int)todo) - 1;
        const bool v0 = (w0 >> b) & 1u, v1 = (w1 >> b) & 1u;
        const unsigned long long m0 = __ballot(v0), m1 = __ballot(v1);
        const int c0 = __popcll(m0), c1 = __popcll(m1);
        if (c0 + c1 == 0) {   // live (only possible for a cut row), but no edge in this segment: its partial sum is zero
          if (lane < A.ld4) A.partial[((int64_t)(bw * 32 + b) * A.n_slots + P.z) * A.ld4 + lane] = f4zero();
          todo &= todo - 1;
          continue;
        }
        if (qn + c0 + c1 > QCAP) break;
        if (v0) st[qn + __popcll(m0 & lt)] = make_float4(__int_as_float(e0.x), 0.f, 0.f, __int_as_float((b << 8) | e0.y));
        if (v1) st[qn + c0 + __popcll(m1 & lt)] = make_float4(__int_as_float(e1.x), 0.f, 0.f, __int_as_float((b << 8) | e1.y));
        qn += c0 + c1;
        todo &= todo - 1;
      }
      if (qn == 0) continue;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // ---- phase 1: one queued edge per lane: source node id, attention scalar; destination runs ------------------------
      int n_runs = 0;
      for (int i0 = 0; i0 < qn; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < qn;
        const float4 t = valid ? st[i] : f4zero();
        const int key = __float_as_int(t.w);
        const int prev_key = (valid && i > 0) ? __float_as_int(st[i - 1].w) : -1;
        const bool head = valid && key != prev_key;
        const unsigned long long hm = __ballot(head);
        if (head) rs[n_runs + __popcll(hm & lt)] = i;
        n_runs += __popcll(hm);
        if (valid) {
          const uint32_t pk = (uint32_t)__float_as_int(t.x);
          const int hd = pk & 0xFFFFF, r = pk >> 20;
          const int bq = bw * 32 + (key >> 8);
          const int2 wp = A.bm_old[(int64_t)bq * A.W + (hd >> 5)];
          const uint32_t word = (uint32_t)wp.x, bit = hd & 31;
          const int s = wp.y + __popc(word & ((1u << bit) - 1u));
          const float4* aq_p = A.a_q + (int64_t)bq * AP4;
          float z = b_alpha;
#pragma unroll
          for (int k = 0; k < AP4; ++k) {
            const float4 as = A.a_s[(int64_t)s * AP4 + k];
            const float4 ar = ar_l[r * AP4 + k];
            const float4 w = w_l[k];
            const float4 q = aq_p[k];
            z = fmaf(w.x, fmaxf(as.x + ar.x + q.x, 0.f), z);
            z = fmaf(w.y, fmaxf(as.y + ar.y + q.y, 0.f), z);
            z = fmaf(w.z, fmaxf(as.z + ar.z + q.z, 0.f), z);
            z = fmaf(w.w, fmaxf(as.w + ar.w + q.w, 0.f), z);
          }
          const float alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
          st[i] = make_float4(__int_as_float(s), __int_as_float(r), alpha, t.w);
        }
      }
      if (lane == 0) rs[n_runs] = qn;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // ---- phase 2: one destination run per lane group: row gathers + FMAs in CSR order, 4 in flight --------------------
      for (int k = gi_w; k < n_runs; k += GW) {
        const int beg = rs[k], end = rs[k + 1];
        float4 acc = f4zero();
        for (int e = beg; e < end; e += 4) {
          float4 tp[4], hv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            tp[u] = st[min(e + u, end - 1)];
            if (e + u >= end) tp[u].z = 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) hv[u] = A.hidden[(int64_t)__float_as_int(tp[u].x) * A.ld4 + lane_c];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int ru = __float_as_int(tp[u].y);
            float4 rv;
            if constexpr (RELA_LDS) rv = rela_l[ru * G + lane_g];
            else rv = A.rela[(int64_t)ru * A.ld4 + lane_c];
            const float al = tp[u].z;
            acc.x = fmaf(al, hv[u].x + rv.x, acc.x);
            acc.y = fmaf(al, hv[u].y + rv.y, acc.y);
            acc.z = fmaf(al, hv[u].z + rv.z, acc.z);
            acc.w = fmaf(al, hv[u].w + rv.w, acc.w);
          }
        }
        const int key = __float_as_int(st[beg].w);
        const int2 dst = A.rows[P.x + (key & 255)];
        const int bq = bw * 32 + (key >> 8);
        if (row_lane) {
          if (dst.y < 0) {
            const int2 wp = A.bm_new[(int64_t)bq * A.W + (dst.x >> 5)];
            const int o = wp.y + __popc((uint32_t)wp.x & ((1u << (dst.x & 31)) - 1u));
            A.agg[(int64_t)o * A.ld4 + lane_g] = acc;
          } else {
            A.partial[((int64_t)bq * A.n_slots + dst.y) * A.ld4 + lane_g] = acc;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the queue is read out before the next fill overwrites it
      __builtin_amdgcn_wave_barrier();
    }
  };

  // ---- work distribution: XCD x serves the x-th eighth of the items from its own in-order queue (cf. walk.h), one item per
  // wave and block step; a workgroup whose queue is dry steals from the others (speed only: every item is taken once)
  constexpr int STEP = WP_WAVES;
  __shared__ long long slot_item[2];
  __shared__ int slot_cnt[2];
  int q = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID
  int n_dry = 0;
  auto resolve = [&](int off, int p) {
    for (;;) {
      const long long qs = A.n_items * q / 8, ql = A.n_items * (q + 1) / 8 - qs;
      if (off < ql) {
        slot_item[p] = qs + off;
        slot_cnt[p] = (int)min((long long)STEP, ql - off);
        return;
      }
      q = (q + 1) & 7;
      if (++n_dry == 8) { slot_cnt[p] = 0; return; }
      off = atomicAdd(&A.queues[q], STEP);
    }
  };
  if (threadIdx.x == 0) resolve(atomicAdd(&A.queues[q], STEP), 0);
  for (int p = 0;; p ^= 1) {
    __syncthreads();                                   // (the first one also publishes the LDS tables)
    const long long it0 = slot_item[p];
    const int cnt = slot_cnt[p];
    if (cnt == 0) break;
    int next_off = 0;
    if (threadIdx.x == 0) next_off = atomicAdd(&A.queues[q], STEP);     // prefetch the next ticket
    if (wv < cnt) run_item(it0 + wv);
    if (threadIdx.x == 0) resolve(next_off, p ^ 1);
  }
}

template <int G, int AP4, bool RELA_LDS>
int launch3(const WpArgs& A, size_t lds, hipStream_t s) {
  auto kern = layer_fwd_wp_kernel<G, AP4, RELA_LDS>;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(A.n_items, WP_WAVES), 256 * per_cu), 1);
  if (rg::zero_async(A.queues, 8 * sizeof(int32_t), s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WP_BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int G, int AP4>
int launch2(const WpArgs& A, hipStream_t s) {
  const size_t lds = (size_t)WP_WAVES * QCAP * sizeof(float4) + (size_t)WP_WAVES * RS_STRIDE * sizeof(int32_t) +
                     (size_t)(A.n_rela_rows * AP4 + AP4) * sizeof(float4);
  const size_t rela_bytes = (size_t)A.n_rela_rows * G * sizeof(float4);
  RG_CHECK(lds <= 160 * 1024, "rg_layer_fwd: attention tables need %zu B of LDS (> 160 KiB)", lds);
  if (lds + rela_bytes <= 80 * 1024) return launch3<G, AP4, true>(A, lds + rela_bytes, s);    // still two workgroups per CU
  return launch3<G, AP4, false>(A, lds, s);
}

template <int G>
int launch_ap(const WpArgs& A, int ap4, hipStream_t s) {
  switch (ap4) {
    case 1: return launch2<G, 1>(A, s);
    case 2: return launch2<G, 2>(A, s);
    case 3: return launch2<G, 3>(A, s);
    case 4: return launch2<G, 4>(A, s);
    case 8: return launch2<G, 8>(A, s);
    default: rg::set_error("rg_layer_fwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

}  // namespace

int launch(const WpArgs& A, int ap4, hipStream_t s) {
  RG_CHECK(A.n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "rg_layer_fwd: work space too large for 32-bit queue tickets");
  if (A.ld4 <= 4) return launch_ap<4>(A, ap4, s);
  if (A.ld4 <= 8) return launch_ap<8>(A, ap4, s);
  if (A.ld4 <= 16) return launch_ap<16>(A, ap4, s);
  if (A.ld4 <= 32) return launch_ap<32>(A, ap4, s);
  return launch_ap<64>(A, ap4, s);
}

}  // namespace rgwp
