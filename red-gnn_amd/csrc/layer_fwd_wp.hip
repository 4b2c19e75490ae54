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
//   fill     for every query bit b set in any lane's word (scalar loop over the set bits only):
//              m = ballot(bit b of word) = the valid edges of query b in this pack, in CSR order, grouped by destination row;
//              append {entry, b | row} to the wave's LDS queue (8 B each)
//   phase 1  lane per queued edge: source node id (popcount rank in the previous level), attention scalar, and the edge's
//            output row (rank of (b, t) in the new level, or the partial-sum slot of a cut row) -> 16-B tuples
//   phase 2  the lane groups split the queue at destination boundaries and stream through their shares, the next edge's row
//            in flight: acc += alpha (hidden[s] + rela[r]); a change of output row stores the finished sum
//
// A destination row cut into segments (hubs, > 128 in-edges) stores partial sums, flagged in `written`; combine_kernel
// (layer_fwd_kernel.h) adds up the flagged ones in segment order (a segment with no edge for a query stores nothing).
// Items are group-major; XCD x serves the x-th eighth of them from its own in-order queue, a wave at a time (cf. walk.h): the
// hidden rows of one group's <= 32 sparse frontiers are what an XCD's L2 holds while it sweeps the packs; n_sub splits a word
// into 2 or 4 groups when they would not fit.
#include "layer_fwd_wp.h"

namespace rgwp {
namespace {

constexpr int WP_BLOCK = 512, WP_WAVES = WP_BLOCK / 64;
// (four waves per SIMD = two workgroups per CU: the kernel waits 75 % of its wave cycles on C3's late hops - 23 % VALU-busy, 10.8
// wave-instructions per edge, profiles/r03/pmc_wp_issue_C3_B256_summary.json - but six / eight waves per SIMD were SLOWER on every hop of
// C3 (2.53 -> 2.64 / 2.85 ms on hop 4), of C2 (hop 1 1.73 -> 2.98 / 3.17 ms) and on family (338 k -> 326 k / 313 k queries/s): more
// waves are more query groups' source rows competing for an XCD's L2)
constexpr int QCAP = 256;   // queued edges per wave between flushes; >= RG_PACK, so one query's edges of a pack always fit
constexpr int WAVE_LDS = QCAP * 16 + 64;     // bytes: tuples [QCAP] (the 8-B fill queue aliases their upper half) + 4 head masks
static_assert(QCAP == 256 && RG_PACK == 128, "a lane holds two entries of a pack; phase 1 takes four queued edges per lane");

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o, 64);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// G lanes own a row, F float4 each (lane l holds columns l, l + G, ... in float4 units: a load instruction of the group
// covers G * 16 contiguous bytes); 64 / G destination runs are in flight per wave.  Every per-edge control and address
// instruction is shared by the 64 / G groups, so few lanes per row = few instructions per edge (d = 64: G = 4, F = 4: 16 edges
// per wave-instruction; the first version, 16 lanes x 1 float4 and one destination run per group at a time, spent 9.3 VALU
// instructions per edge against 4.4 in the per-query walk and was VALU-bound at 58 % busy).
// EXACT: the row is exactly G * F float4 wide (d = 64, 128): column offsets are instruction immediates.
template <int G, int F, int AP4, bool RELA_LDS, bool EXACT>
__global__ __launch_bounds__(WP_BLOCK, 4) void layer_fwd_wp_kernel(WpArgs A) {
  extern __shared__ float4 lds[];
  constexpr int GW = 64 / G, RW = G * F + 1;      // groups per wave; float4 per staged relation row (+1: rows of different groups
                                                  // start in different LDS banks)
  char* wave_lds = reinterpret_cast<char*>(lds);                                 // [WAVES][WAVE_LDS]
  float4* ar_l = reinterpret_cast<float4*>(wave_lds + WP_WAVES * WAVE_LDS);      // [n_rela_rows][AP4]
  float4* w_l = ar_l + A.n_rela_rows * AP4;                                      // [AP4]
  float4* rela_l = w_l + AP4;                                                    // [n_rela_rows][RW] (optional)

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
    for (int i = threadIdx.x; i < A.n_rela_rows * RW; i += WP_BLOCK) {
      const int r = i / RW, c = i - r * RW;
      rela_l[i] = c < A.ld4 ? A.rela[(int64_t)r * A.ld4 + c] : f4zero();
    }
  }
  __syncthreads();
  const float b_alpha = A.b_alpha[0];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane_g = lane & (G - 1), gi_w = lane / G;
  const unsigned long long lt = (1ull << lane) - 1ull;
  float4* st = reinterpret_cast<float4*>(wave_lds + wv * WAVE_LDS);            // tuples {hidden row offset, relation row offset, alpha, output row}
  int2* q8 = reinterpret_cast<int2*>(st) + QCAP;                               // fill queue {packed entry, b << 8 | row}: upper half of
                                                                               // the tuples' bytes; tuple i is written after entry i is read
  unsigned long long* hmask = reinterpret_cast<unsigned long long*>(st + QCAP);   // run heads of the queue, 64 entries per word
  const int qbits = 32 / A.n_sub;
  const uint32_t row_bytes = (uint32_t)A.ld4 * 16u;
  // this lane's F columns (float4 units): lane_g, lane_g + G, ...; columns beyond the row re-read its last float4 (loads never
  // branch) and are not stored.  col_d[f] = byte distance of column f from the lane's first column.
  const uint32_t lane_off = (uint32_t)lane_g * 16u;
  uint32_t col_d[F];
  bool col_ok[F];
#pragma unroll
  for (int f = 0; f < F; ++f) {
    const int c = lane_g + G * f;
    col_ok[f] = EXACT || c < A.ld4;
    col_d[f] = EXACT ? (uint32_t)(f * G * 16) : (uint32_t)((col_ok[f] ? c : A.ld4 - 1) - lane_g) * 16u;
  }
  const char* hidden_b = reinterpret_cast<const char*>(A.hidden);
  const char* rela_b = reinterpret_cast<const char*>(A.rela);
  const char* rela_lb = reinterpret_cast<const char*>(rela_l) + lane_g * 16;

  auto store_row = [&](int out, const float4 (&acc)[F]) {
    float4* row = out >= 0 ? A.agg + (int64_t)out * A.ld4 : A.partial + (int64_t)(-out - 1) * A.ld4;
#pragma unroll
    for (int f = 0; f < F; ++f)
      if (col_ok[f]) row[lane_g + G * f] = acc[f];
    if (out < 0 && lane_g == 0) A.written[-out - 1] = 1;
  };
  auto gather = [&](const float4& t, float4 (&hv)[F]) {
    const uint32_t off = (uint32_t)__float_as_int(t.x) + lane_off;      // 32-bit: uniform base + lane offset + immediate
#pragma unroll
    for (int f = 0; f < F; ++f) {
      if constexpr (EXACT) hv[f] = *reinterpret_cast<const float4*>(hidden_b + off + f * (G * 16));     // (added in 64 bits: folds)
      else hv[f] = *reinterpret_cast<const float4*>(hidden_b + (off + col_d[f]));
    }
  };
  auto consume = [&](const float4& t, const float4 (&hv)[F], float4 (&acc)[F]) {
    const float al = t.z;
    const uint32_t ro = (uint32_t)__float_as_int(t.y);
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float4 rv;
      if constexpr (RELA_LDS) rv = *reinterpret_cast<const float4*>(rela_lb + ro + f * (G * 16));
      else if constexpr (EXACT) rv = *reinterpret_cast<const float4*>(rela_b + (ro + lane_off) + f * (G * 16));
      else rv = *reinterpret_cast<const float4*>(rela_b + (ro + lane_off + col_d[f]));
      acc[f].x = fmaf(al, hv[f].x + rv.x, acc[f].x);
      acc[f].y = fmaf(al, hv[f].y + rv.y, acc[f].y);
      acc[f].z = fmaf(al, hv[f].z + rv.z, acc[f].z);
      acc[f].w = fmaf(al, hv[f].w + rv.w, acc[f].w);
    }
  };

  auto run_item = [&](long long item) {
    const int grp = (int)(item / A.n_packs);
    const int pi = __builtin_amdgcn_readfirstlane((int)(item - (long long)grp * A.n_packs));
    const int bw = __builtin_amdgcn_readfirstlane(grp / A.n_sub), sub = grp - bw * A.n_sub;
    const uint32_t qmask = A.n_sub == 1 ? 0xFFFFFFFFu : (((1u << qbits) - 1u) << (sub * qbits));
    const int2 e0 = A.ent[(int64_t)pi * RG_PACK + lane], e1 = A.ent[(int64_t)pi * RG_PACK + 64 + lane];
    uint32_t w0 = 0, w1 = 0;
    if (e0.x != -1) w0 = A.bits_old[(int64_t)(e0.x & 0xFFFFF) * A.BW + bw] & qmask;
    if (e1.x != -1) w1 = A.bits_old[(int64_t)(e1.x & 0xFFFFF) * A.BW + bw] & qmask;
    const int row0 = A.pack[pi].x;
    uint32_t todo = wave_or(w0 | w1);
    while (todo) {
      // ---- fill: the valid edges of query after query, in (query, CSR order) --------------------------------------------
      int qn = 0;
      while (todo) {
        const int b = __ffs((int)todo) - 1;
        const bool v0 = (w0 >> b) & 1u, v1 = (w1 >> b) & 1u;
        const unsigned long long m0 = __ballot(v0), m1 = __ballot(v1);
        const int c0 = __popcll(m0), c1 = __popcll(m1);
        if (qn + c0 + c1 > QCAP) break;
        if (v0) q8[qn + __popcll(m0 & lt)] = make_int2(e0.x, (b << 8) | e0.y);
        if (v1) q8[qn + c0 + __popcll(m1 & lt)] = make_int2(e1.x, (b << 8) | e1.y);
        qn += c0 + c1;
        todo &= todo - 1;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- phase 1: queued edges -> tuples, up to four per lane at once: their dependent load chains (frontier words and row records,
      // then the destination's word and the source's attention row) run side by side (two trips of two were 29 % of the kernel's cycles
      // on C2's second hop: one trip waited out the other's chains) ---------------------------------------------------------------
      {
        int2 en[4];
        int s[4], out[4];
        bool valid[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = j * 64 + lane;
          valid[j] = i < qn;
          en[j] = valid[j] ? q8[i] : make_int2(0, -1);
          const int prev_key = (valid[j] && i > 0) ? q8[i - 1].y : -2;
          const unsigned long long hm = __ballot(valid[j] && en[j].y != prev_key);
          if (lane == 0) hmask[j] = hm;
        }
        int2 wp[4], dst[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wp[j] = make_int2(0, 0); dst[j] = make_int2(0, 0);
          if (valid[j]) {
            const int hd = en[j].x & 0xFFFFF, bq = bw * 32 + (en[j].y >> 8);
            wp[j] = A.bm_old[(int64_t)bq * A.W + (hd >> 5)];
            dst[j] = A.rows[row0 + (en[j].y & 255)];                  // {entity, slot of a cut row's partial sum or -1}
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = 0; out[j] = 0;
          if (valid[j]) {
            const int hd = en[j].x & 0xFFFFF, bq = bw * 32 + (en[j].y >> 8);
            s[j] = wp[j].y + __popc((uint32_t)wp[j].x & ((1u << (hd & 31)) - 1u));
            if (dst[j].y < 0) {
              const int2 wn = A.bm_new[(int64_t)bq * A.W + (dst[j].x >> 5)];
              out[j] = wn.y + __popc((uint32_t)wn.x & ((1u << (dst[j].x & 31)) - 1u));
            } else {
              out[j] = -(bq * A.n_slots + dst[j].y) - 1;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (valid[j]) {
            const int r = (uint32_t)en[j].x >> 20;
            const float4* aq_p = A.a_q + (int64_t)(bw * 32 + (en[j].y >> 8)) * AP4;
            float z = b_alpha;
#pragma unroll
            for (int k = 0; k < AP4; ++k) {
              const float4 as = A.a_s[(int64_t)s[j] * AP4 + k];
              const float4 ar = ar_l[r * AP4 + k];
              const float4 w = w_l[k];
              const float4 q = aq_p[k];
              z = fmaf(w.x, fmaxf(as.x + ar.x + q.x, 0.f), z);
              z = fmaf(w.y, fmaxf(as.y + ar.y + q.y, 0.f), z);
              z = fmaf(w.z, fmaxf(as.z + ar.z + q.z, 0.f), z);
              z = fmaf(w.w, fmaxf(as.w + ar.w + q.w, 0.f), z);
            }
            const float alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
            const uint32_t hoff = (uint32_t)s[j] * row_bytes;                 // < 2^32: checked by the launcher
            const uint32_t roff = (uint32_t)r * (RELA_LDS ? (uint32_t)(RW * 16) : row_bytes);
            st[j * 64 + lane] = make_float4(__int_as_float((int)hoff), __int_as_float((int)roff), alpha, __int_as_float(out[j]));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- phase 2: the lane groups take contiguous shares of the queue, cut at destination boundaries, and stream through
      // them one edge per trip with the next edge's row already in flight ---------------------------------------------------
      auto next_head = [&](int c) -> int {       // first run head at or after queue position c (qn if none)
        int pos = qn;
#pragma unroll
        for (int w = 3; w >= 0; --w) {
          unsigned long long m = hmask[w];
          if (w == (c >> 6)) m &= ~0ull << (c & 63);
          else if (w < (c >> 6)) m = 0ull;
          if (m) pos = w * 64 + __ffsll((long long)m) - 1;
        }
        return pos;
      };
      int e = 0, end = qn;
      if constexpr (GW > 1) {
        e = gi_w == 0 ? 0 : next_head(gi_w * qn / GW);
        end = gi_w == GW - 1 ? qn : next_head((gi_w + 1) * qn / GW);
      }
      if (e < end) {
        float4 acc[F], hvA[F], hvB[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = f4zero();
        float4 tA = st[e], tB;
        int cur = __float_as_int(tA.w);
        gather(tA, hvA);
        for (;;) {
          // A is current; B = the next edge of the share (or A again past its end: loaded, never used)
          tB = st[min(e + 1, end - 1)];
          gather(tB, hvB);
          consume(tA, hvA, acc);
          if (e + 1 >= end || __float_as_int(tB.w) != cur) {
            store_row(cur, acc);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = f4zero();
            cur = __float_as_int(tB.w);
          }
          if (e + 1 >= end) break;
          tA = st[min(e + 2, end - 1)];
          gather(tA, hvA);
          consume(tB, hvB, acc);
          if (e + 2 >= end || __float_as_int(tA.w) != cur) {
            store_row(cur, acc);
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = f4zero();
            cur = __float_as_int(tA.w);
          }
          if (e + 2 >= end) break;
          e += 2;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the queue is read out before the next fill overwrites it
      __builtin_amdgcn_wave_barrier();
    }
  };

  // ---- work distribution: XCD x serves the x-th eighth of the items from its own in-order queue, A.ipt items per ticket and wave;
  // a wave whose queue is dry steals from the others (speed only: every item is taken exactly once under any placement)
  int q = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;   // HW_REG_XCC_ID
  auto ticket = [&]() -> int {
    int off = 0;
    if (lane == 0) off = atomicAdd(&A.queues[q * RG_QSTRIDE], A.ipt);
    return off;
  };
  // lane 0's ticket -> (first item, count) for the whole wave; count 0 = every queue is dry.  A dry queue is followed by one look at all
  // eight heads (lanes 0..7, one round trip) and a ticket from the first queue that still has items (cf. walk.h)
  auto resolve = [&](int off, long long* first) -> int {
    off = __builtin_amdgcn_readfirstlane(off);
    for (;;) {
      const long long qs = A.n_items * q / 8, ql = A.n_items * (q + 1) / 8 - qs;
      if (off < ql) {
        *first = qs + off;
        return (int)min((long long)A.ipt, ql - off);
      }
      const int k = (q + 1 + lane) & 7;                   // lanes 0..6: the other queues in stealing order
      bool has = false;
      if (lane < 7) has = __hip_atomic_load(&A.queues[k * RG_QSTRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < A.n_items * (k + 1) / 8 - A.n_items * k / 8;
      const unsigned long long m = __ballot(has);
      if (m == 0ull) return 0;                            // (heads only grow: dry stays dry)
      q = (q + 1 + (__ffsll((long long)m) - 1)) & 7;
      off = __builtin_amdgcn_readfirstlane(ticket());
    }
  };
  int off = ticket();
  for (;;) {
    long long first = 0;
    const int cnt = resolve(off, &first);
    if (cnt == 0) break;
    off = ticket();                                     // the next ticket is in flight while these items run
    for (int k = 0; k < cnt; ++k) run_item(first + k);
  }
}

template <int G, int F, int AP4, bool RELA_LDS, bool EXACT>
int launch4(const WpArgs& A, size_t lds, hipStream_t s) {
  auto kern = layer_fwd_wp_kernel<G, F, AP4, RELA_LDS, EXACT>;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(A.n_items, WP_WAVES * A.ipt), 256 * per_cu), 1);
  if (!A.queues_clean && rg::zero_async(A.queues, RG_QUEUE_BYTES, s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WP_BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int G, int F, int AP4, bool RELA_LDS>
int launch3(const WpArgs& A, size_t lds, hipStream_t s) {
  if constexpr (G * F >= 16) {      // d = 64, 128, 256: the common widths get the immediate-offset form
    if (A.ld4 == G * F) return launch4<G, F, AP4, RELA_LDS, true>(A, lds, s);
  }
  return launch4<G, F, AP4, RELA_LDS, false>(A, lds, s);
}

template <int G, int F, int AP4>
int launch2(const WpArgs& A, hipStream_t s) {
  const size_t lds = (size_t)WP_WAVES * WAVE_LDS + (size_t)(A.n_rela_rows * AP4 + AP4) * sizeof(float4);
  const size_t rela_bytes = (size_t)A.n_rela_rows * (G * F + 1) * sizeof(float4);
  RG_CHECK(lds <= 160 * 1024, "rg_layer_fwd: attention tables need %zu B of LDS (> 160 KiB)", lds);
  if (lds + rela_bytes <= 80 * 1024) return launch3<G, F, AP4, true>(A, lds + rela_bytes, s);    // still two workgroups per CU
  return launch3<G, F, AP4, false>(A, lds, s);
}

template <int G, int F>
int launch_ap(const WpArgs& A, int ap4, hipStream_t s) {
  switch (ap4) {
    case 1: return launch2<G, F, 1>(A, s);
    case 2: return launch2<G, F, 2>(A, s);
    case 3: return launch2<G, F, 3>(A, s);
    case 4: return launch2<G, F, 4>(A, s);
    case 8: return launch2<G, F, 8>(A, s);
    default: rg::set_error("rg_layer_fwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

}  // namespace

bool offsets_fit(int64_t n_old, int32_t ld) { return n_old >= 0 && (uint64_t)n_old * ld * sizeof(float) < ((uint64_t)1 << 32); }

// lane grouping per row width: two float4 per lane (one 128-byte line per group and load instruction at d = 64).  Measured on C2's
// second hop: (16 lanes, 1 float4) 1.79 ms, (8, 2) 1.75 ms, (4, 4) 2.12 ms (twice the L1 transactions per row, 16-way conflicts on
// the tuple reads) - profiles/r02/per_hop_wp_lane_grouping_variants.txt
int launch(const WpArgs& A, int ap4, hipStream_t s) {
  RG_CHECK(A.n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "rg_layer_fwd: work space too large for 32-bit queue tickets");
  if (A.ld4 <= 4) return launch_ap<2, 2>(A, ap4, s);
  if (A.ld4 <= 8) return launch_ap<4, 2>(A, ap4, s);
  if (A.ld4 <= 16) return launch_ap<8, 2>(A, ap4, s);
  if (A.ld4 <= 32) return launch_ap<16, 2>(A, ap4, s);
  return launch_ap<32, 2>(A, ap4, s);
}

}  // namespace rgwp
