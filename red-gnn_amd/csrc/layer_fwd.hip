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
// Mapping (wave64):
//   * a wave grabs 64 consecutive work items from an in-order queue; each lane tests one item
//     (visited bit + rank), survivors are compacted through a wave-private LDS strip;
//   * a survivor is owned by a group of G lanes, G*4 >= d floats, so a row is one coalesced float4
//     per lane (d=64: 16 lanes x 16 B = 256 B per row, 4 destinations per wave);
//   * phase 1 runs lane-per-candidate (index math + attention scalar, G candidates at a time);
//     surviving edges are compacted into a per-group LDS strip; phase 2 runs group-per-edge
//     (row gather + FMA), four edges in flight per group;
//   * virtual rows are sorted by length, so the groups of a wave have equal trip counts, and a hub
//     of in-degree 17k is 133 independent items instead of one 17k-long serial loop;
//   * work space is query-major; XCD x serves the x-th eighth of it from its own queue (in order),
//     so the hidden slab of the queries being processed (<= n_ent*d*4 B each) stays in that XCD's
//     4 MiB L2 while every destination of the query gathers from it.  Queues are only a speed
//     device: a wave that finds its queue empty steals from the others, so every item is processed
//     whatever the workgroup placement.
// rela / a_r / w_alpha live in LDS.  Sums run in CSR order: bitwise reproducible.
#include "common.h"

namespace {

struct FwdArgs {
  int64_t n_items;   // B * n_vrows
  int32_t n_vrows;
  int32_t n_slots;
  const int4* vrows;
  const int2* in_hr;
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
  int32_t* queues;  // [8], zeroed before the launch
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int G, int AP4, int BLOCK>
__global__ __launch_bounds__(BLOCK) void layer_fwd_kernel(FwdArgs A) {
  extern __shared__ float4 lds[];
  constexpr int GW = 64 / G;                             // destination groups per wave
  float4* stage = lds;                                   // [BLOCK] edge tuples {s, r, alpha, -}
  int4* recs = reinterpret_cast<int4*>(lds + BLOCK);     // [BLOCK] surviving items {beg, len, b, out}
  float4* ar_l = lds + 2 * BLOCK;                        // [n_rela_rows][AP4]
  float4* w_l = ar_l + A.n_rela_rows * AP4;              // [AP4]
  float4* rela_l = w_l + AP4;                            // [n_rela_rows][G] (optional)

  for (int i = threadIdx.x; i < A.n_rela_rows * AP4; i += BLOCK) ar_l[i] = A.a_r[i];
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if (A.rela_in_lds) {
    for (int i = threadIdx.x; i < A.n_rela_rows * G; i += BLOCK) {
      const int r = i / G, c = i - r * G;
      rela_l[i] = c < A.ld4 ? A.rela[(int64_t)r * A.ld4 + c] : f4zero();
    }
  }
  __syncthreads();
  const float b_alpha = A.b_alpha[0];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane_g = lane & (G - 1), gi_w = lane / G;
  int4* my_recs = recs + wv * 64;
  float4* my_stage = stage + wv * 64 + gi_w * G;
  const int gshift = lane & ~(G - 1);
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const bool row_lane = lane_g < A.ld4;

  // HW_REG_XCC_ID (id 20, 4 bits): which XCD this workgroup runs on.  Speed only.
  int q = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
  int n_dry = 0;

  for (;;) {
    // ---- grab 64 consecutive items from queue q (steal from the next queue when it is dry) --------
    int64_t qs = 0, ql = 0;
    int off = 0;
    for (;;) {
      qs = A.n_items * q / 8;
      ql = A.n_items * (q + 1) / 8 - qs;
      off = 0;
      if (lane == 0) off = atomicAdd(&A.queues[q], 64);
      off = __builtin_amdgcn_readfirstlane(off);
      if (off < ql) break;
      q = (q + 1) & 7;
      if (++n_dry == 8) return;      // every queue seen dry: all items are taken
    }

    // ---- filter: one item per lane -----------------------------------------------------------------
    bool ok = (int64_t)off + lane < ql;
    int4 rec = make_int4(0, 0, 0, 0);
    if (ok) {
      const int64_t item = qs + off + lane;
      const int b = (int)(item / A.n_vrows);
      const int vr = (int)(item - (int64_t)b * A.n_vrows);
      const int4 row = A.vrows[vr];
      const int2 wp = A.bm_new[(int64_t)b * A.W + (row.x >> 5)];
      const uint32_t word = (uint32_t)wp.x, bit = row.x & 31;
      ok = (word >> bit) & 1u;
      if (ok) {
        const int o = wp.y + __popc(word & ((1u << bit) - 1u));
        rec = make_int4(row.y, row.z, b, row.w < 0 ? o : -(b * A.n_slots + row.w) - 1);
      }
    }
    const unsigned long long surv = __ballot(ok);
    const int n_surv = __popcll(surv);
    if (n_surv == 0) continue;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (ok) my_recs[__popcll(surv & ((1ull << lane) - 1ull))] = rec;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // ---- survivors, GW at a time: one per lane group ---------------------------------------------------
    for (int j = 0; j < n_surv; j += GW) {
      const bool live = j + gi_w < n_surv;
      const int4 R = live ? my_recs[j + gi_w] : make_int4(0, 0, 0, 0);
      const int beg = R.x, end = R.x + R.y, b = R.z;
      float4 aq[AP4];
#pragma unroll
      for (int k = 0; k < AP4; ++k) aq[k] = A.a_q[(int64_t)b * AP4 + k];
      const int2* bm_row = A.bm_old + (int64_t)b * A.W;
      float4 acc = f4zero();

      for (int c0 = beg; c0 < end; c0 += G) {
        // ---- phase 1: one candidate in-edge per lane ---------------------------------------------
        const int c = c0 + lane_g;
        bool valid = c < end;
        int s = 0, r = 0;
        float alpha = 0.f;
        if (valid) {
          const int2 hr = A.in_hr[c];
          const int2 wp = bm_row[hr.x >> 5];
          const uint32_t word = (uint32_t)wp.x, bit = hr.x & 31;
          valid = (word >> bit) & 1u;
          if (valid) {
            s = wp.y + __popc(word & ((1u << bit) - 1u));
            r = hr.y;
            float z = b_alpha;
#pragma unroll
            for (int k = 0; k < AP4; ++k) {
              const float4 as = A.a_s[(int64_t)s * AP4 + k];
              const float4 ar = ar_l[r * AP4 + k];
              const float4 w = w_l[k];
              z = fmaf(w.x, fmaxf(as.x + ar.x + aq[k].x, 0.f), z);
              z = fmaf(w.y, fmaxf(as.y + ar.y + aq[k].y, 0.f), z);
              z = fmaf(w.z, fmaxf(as.z + ar.z + aq[k].z, 0.f), z);
              z = fmaf(w.w, fmaxf(as.w + ar.w + aq[k].w, 0.f), z);
            }
            alpha = 1.0f / (1.0f + expf(-z));
          }
        }
        const unsigned long long m = (__ballot(valid) >> gshift) & gmask;
        const int cnt = __popcll(m);
        const int pos = __popcll(m & ((1ull << lane_g) - 1ull));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // previous round's reads are done
        __builtin_amdgcn_wave_barrier();
        if (lane_g >= cnt) my_stage[lane_g] = f4zero();          // pad tuples: alpha = 0, row 0
        if (valid) my_stage[pos] = make_float4(__int_as_float(s), __int_as_float(r), alpha, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---- phase 2: one edge per group step, 4 in flight -------------------------------------
        for (int k = 0; k < cnt; k += 4) {
          float4 tp[4], hv[4], rv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) tp[u] = my_stage[k + u];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int su = __float_as_int(tp[u].x), ru = __float_as_int(tp[u].y);
            hv[u] = row_lane ? A.hidden[(int64_t)su * A.ld4 + lane_g] : f4zero();
            rv[u] = A.rela_in_lds ? rela_l[ru * G + lane_g]
                                  : (row_lane ? A.rela[(int64_t)ru * A.ld4 + lane_g] : f4zero());
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float al = tp[u].z;
            acc.x = fmaf(al, hv[u].x + rv[u].x, acc.x);
            acc.y = fmaf(al, hv[u].y + rv[u].y, acc.y);
            acc.z = fmaf(al, hv[u].z + rv[u].z, acc.z);
            acc.w = fmaf(al, hv[u].w + rv[u].w, acc.w);
          }
        }
      }
      if (live && row_lane) {
        if (R.w >= 0) A.agg[(int64_t)R.w * A.ld4 + lane_g] = acc;
        else A.partial[(int64_t)(-R.w - 1) * A.ld4 + lane_g] = acc;
      }
    }
  }
}

// hubs cut into segments: agg[o] = sum of the segments' partial rows, in segment order
__global__ void combine_kernel(const int4* __restrict__ split, int n_split, int n_slots, int B, const int2* __restrict__ bm_new,
                               int W, const float4* __restrict__ partial, float4* __restrict__ agg, int ld4) {
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
  float4 acc = p[0];
  for (int k = 1; k < se.z; ++k) {
    const float4 v = p[(int64_t)k * ld4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  agg[(int64_t)o * ld4 + c] = acc;
}

template <int G, int AP4>
int launch(const FwdArgs& A, int B, const rg_vrows& vr, hipStream_t s) {
  constexpr int BLOCK = 256;
  FwdArgs a = A;
  size_t lds = (size_t)(2 * BLOCK + A.n_rela_rows * AP4 + AP4) * sizeof(float4);
  const size_t rela_bytes = (size_t)A.n_rela_rows * G * sizeof(float4);
  a.rela_in_lds = (lds + rela_bytes <= 40 * 1024) ? 1 : 0;
  if (a.rela_in_lds) lds += rela_bytes;
  RG_CHECK(lds <= 64 * 1024, "rg_layer_fwd: attention tables need %zu B of LDS (> 64 KiB)", lds);
  RG_HIP(hipMemsetAsync(a.queues, 0, 8 * sizeof(int32_t), s));
  const int64_t n_batches = rg::ceil_div(A.n_items, 64);
  int grid = (int)std::min<int64_t>(rg::ceil_div(n_batches, BLOCK / 64), 256 * 8);
  grid = std::max(grid, 8);
  hipLaunchKernelGGL((layer_fwd_kernel<G, AP4, BLOCK>), dim3(grid), dim3(BLOCK), lds, s, a);
  RG_LAUNCH_CHECK();
  if (vr.n_split > 0) {
    const int64_t threads = (int64_t)B * vr.n_split * A.ld4;
    hipLaunchKernelGGL(combine_kernel, dim3(rg::ceil_div(threads, 256)), dim3(256), 0, s, vr.split, vr.n_split, vr.n_slots, B,
                       A.bm_new, A.W, A.partial, A.agg, A.ld4);
    RG_LAUNCH_CHECK();
  }
  return 0;
}

template <int G>
int launch_ap(const FwdArgs& A, int ap4, int B, const rg_vrows& vr, hipStream_t s) {
  switch (ap4) {
    case 1: return launch<G, 1>(A, B, vr, s);
    case 2: return launch<G, 2>(A, B, vr, s);
    case 3: return launch<G, 3>(A, B, vr, s);
    case 4: return launch<G, 4>(A, B, vr, s);
    case 8: return launch<G, 8>(A, B, vr, s);
    default: rg::set_error("rg_layer_fwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

}  // namespace

extern "C" size_t rg_layer_fwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld) {
  if (!f || !g) return 0;
  return (size_t)f->B * g->in_vr.n_slots * ld * sizeof(float) + 256;
}

extern "C" int rg_layer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const float* hidden,
                            const float* rela, int32_t d, int32_t ld, const float* a_s, const float* a_r,
                            const float* a_q, int32_t ap, const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                            float* agg_out, void* scratch, size_t scratch_bytes, void* stream) {
  RG_CHECK(f && g && hidden && rela && a_s && a_r && a_q && w_alpha && b_alpha && agg_out, "rg_layer_fwd: NULL argument");
  RG_CHECK(g->n_ent == f->n_ent, "rg_layer_fwd: graph has %d entities, frontier %d", g->n_ent, f->n_ent);
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1,
           "rg_layer_fwd: level %d not resident (current %d, %d kept)", level, f->level, f->n_levels);
  RG_CHECK(n_new == f->n_nodes[level % f->n_levels], "rg_layer_fwd: n_new=%lld but level %d has %lld nodes",
           (long long)n_new, level, (long long)f->n_nodes[level % f->n_levels]);
  RG_CHECK(d > 0 && ld >= d && ld % 4 == 0 && ld >= 16 && ld <= 256, "rg_layer_fwd: d=%d ld=%d (need ld%%4==0, 16<=ld<=256)", d, ld);
  RG_CHECK(attn_dim > 0 && ap >= attn_dim && ap % 4 == 0, "rg_layer_fwd: attn_dim=%d ap=%d", attn_dim, ap);
  RG_CHECK((((uintptr_t)hidden | (uintptr_t)rela | (uintptr_t)a_s | (uintptr_t)a_r | (uintptr_t)a_q | (uintptr_t)agg_out |
             (uintptr_t)scratch) & 15) == 0, "rg_layer_fwd: float buffers must be 16-B aligned");
  const size_t need = rg_layer_fwd_scratch_bytes(f, g, ld);
  RG_CHECK(g->in_vr.n_slots == 0 || (scratch && scratch_bytes >= need), "rg_layer_fwd: scratch %zu B < required %zu B",
           scratch_bytes, need);
  RG_CHECK((int64_t)f->B * std::max(g->in_vr.n_slots, 1) < ((int64_t)1 << 31), "rg_layer_fwd: batch * hub segments overflows int32");
  const int64_t n_items = (int64_t)f->B * g->in_vr.n;
  RG_CHECK(n_items / 8 + 64 < ((int64_t)1 << 31) - ((int64_t)1 << 24), "rg_layer_fwd: work space too large for 32-bit queues");
  if (n_new == 0) return 0;
  FwdArgs A;
  A.n_items = n_items; A.n_vrows = g->in_vr.n; A.n_slots = g->in_vr.n_slots; A.vrows = g->in_vr.rows;
  A.in_hr = g->in_hr;
  A.bm_old = f->bm_of(level - 1); A.bm_new = f->bm_of(level); A.W = f->W;
  A.hidden = (const float4*)hidden; A.rela = (const float4*)rela; A.ld4 = ld / 4;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha; A.attn_dim = attn_dim;
  A.n_rela_rows = 2 * g->n_rel + 1; A.rela_in_lds = 0;
  A.agg = (float4*)agg_out; A.partial = (float4*)scratch; A.queues = f->counters + 16;
  hipStream_t s = (hipStream_t)stream;
  const int ld4 = ld / 4;
  if (ld4 <= 4) return launch_ap<4>(A, ap / 4, f->B, g->in_vr, s);
  if (ld4 <= 8) return launch_ap<8>(A, ap / 4, f->B, g->in_vr, s);
  if (ld4 <= 16) return launch_ap<16>(A, ap / 4, f->B, g->in_vr, s);
  if (ld4 <= 32) return launch_ap<32>(A, ap / 4, f->B, g->in_vr, s);
  return launch_ap<64>(A, ap / 4, f->B, g->in_vr, s);
}
