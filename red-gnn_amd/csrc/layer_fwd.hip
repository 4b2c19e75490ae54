// Fused relational message passing, forward.
// Replaces GNNLayer.forward lines Static/transductive/models.py:29-39 — the five E x d gathers,
// the attention MLP on E rows and torch_scatter.scatter(reduce='sum') — with ONE kernel that
// never materialises an edge list or an E x d temporary:
//
//   for every destination node (b,t) of the new frontier (sorted, so queries are contiguous):
//     for every KG in-edge (h, r) -> t  (CSR by tail, shared by all queries):
//       if (b,h) is in the previous frontier (bitmap test):
//         s      = rank of (b,h)          (popcount prefix: the node id, no sort, no hash)
//         alpha  = sigmoid(w . relu(a_s[s] + a_r[r] + a_q[b]) + b_alpha)
//         acc   += alpha * (hidden[s] + rela[r])
//     agg[(b,t)] = acc                    (one plain store per row; deterministic order)
//
// Mapping (wave64): a destination is owned by a group of G lanes, G*4 >= d floats, so a row is
// one coalesced float4 per lane (d=64: 16 lanes x 16 B = 256 B per row, 4 destinations per wave).
// Phase 1 runs lane-per-candidate (index math + attention scalar, G candidates at a time);
// survivors are compacted into a per-group LDS staging strip; phase 2 runs group-per-edge
// (row gather + FMA), four edges in flight per group.  rela / a_r / w_alpha live in LDS.
// Work is dealt to the 8 XCDs in contiguous node ranges: a query's hidden slab (<= n_ent*d*4 B)
// then stays in that XCD's 4 MiB L2 while all its destinations gather from it.
#include "common.h"

namespace {

struct FwdArgs {
  const int32_t* nodes_new;
  int64_t n_new;
  const int32_t* in_ptr;
  const int2* in_hr;
  const int2* bm_old;
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
  int n_chunks;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int G, int AP4, int BLOCK>
__global__ __launch_bounds__(BLOCK) void layer_fwd_kernel(FwdArgs A) {
  extern __shared__ float4 lds[];
  constexpr int GPB = BLOCK / G;  // destinations per block iteration
  float4* stage = lds;                                   // [BLOCK] {s, r, alpha, -}
  float4* ar_l = lds + BLOCK;                            // [n_rela_rows][AP4]
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

  const int lane_g = threadIdx.x & (G - 1);
  const int gi = threadIdx.x / G;
  float4* my_stage = stage + gi * G;
  const int gshift = (threadIdx.x & 63) & ~(G - 1);      // first lane of my group inside the wave
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const bool row_lane = lane_g < A.ld4;

  // XCD-aware chunk walk: XCD x (= blockIdx % 8 under round-robin dispatch) owns a contiguous
  // eighth of the chunks; its blocks sweep that range front to back.
  const int x = blockIdx.x & 7, j0 = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int cpx = (A.n_chunks + 7) >> 3;
  const int c_end = min((x + 1) * cpx, A.n_chunks);

  for (int chunk = x * cpx + j0; chunk < c_end; chunk += nbx) {
    const int64_t item = (int64_t)chunk * GPB + gi;
    const bool live = item < A.n_new;
    int b = 0, t = 0, beg = 0, end = 0;
    if (live) {
      b = A.nodes_new[2 * item];
      t = A.nodes_new[2 * item + 1];
      beg = A.in_ptr[t];
      end = A.in_ptr[t + 1];
    }
    float4 aq[AP4];
#pragma unroll
    for (int k = 0; k < AP4; ++k) aq[k] = live ? A.a_q[(int64_t)b * AP4 + k] : f4zero();
    const int2* bm_row = A.bm_old + (int64_t)b * A.W;
    float4 acc = f4zero();

    for (int c0 = beg; c0 < end; c0 += G) {
      // ---- phase 1: one candidate in-edge per lane -------------------------------------------
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

      // ---- phase 2: one edge per group step, 4 in flight ---------------------------------------
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
    if (live && row_lane) A.agg[item * A.ld4 + lane_g] = acc;
  }
}

template <int G, int AP4>
int launch(const FwdArgs& A, hipStream_t s) {
  constexpr int BLOCK = 512;
  constexpr int GPB = BLOCK / G;
  FwdArgs a = A;
  a.n_chunks = (int)rg::ceil_div(A.n_new, GPB);
  size_t lds = (size_t)(BLOCK + A.n_rela_rows * AP4 + AP4) * sizeof(float4);
  const size_t rela_bytes = (size_t)A.n_rela_rows * G * sizeof(float4);
  a.rela_in_lds = (lds + rela_bytes <= 40 * 1024) ? 1 : 0;   // keep 4 blocks of 512 threads per CU
  if (a.rela_in_lds) lds += rela_bytes;
  RG_CHECK(lds <= 64 * 1024, "rg_layer_fwd: attention tables need %zu B of LDS (> 64 KiB)", lds);
  int grid = (int)std::min<int64_t>(a.n_chunks, 256 * 4);
  grid = (grid + 7) & ~7;
  hipLaunchKernelGGL((layer_fwd_kernel<G, AP4, BLOCK>), dim3(grid), dim3(BLOCK), lds, s, a);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int G>
int launch_ap(const FwdArgs& A, int ap4, hipStream_t s) {
  switch (ap4) {
    case 1: return launch<G, 1>(A, s);
    case 2: return launch<G, 2>(A, s);
    case 3: return launch<G, 3>(A, s);
    case 4: return launch<G, 4>(A, s);
    case 8: return launch<G, 8>(A, s);
    default: rg::set_error("rg_layer_fwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

}  // namespace

extern "C" int rg_layer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, const int32_t* nodes_new,
                            int64_t n_new, const float* hidden, const float* rela, int32_t d, int32_t ld,
                            const float* a_s, const float* a_r, const float* a_q, int32_t ap, const float* w_alpha,
                            const float* b_alpha, int32_t attn_dim, float* agg_out, void* stream) {
  RG_CHECK(f && g && nodes_new && hidden && rela && a_s && a_r && a_q && w_alpha && b_alpha && agg_out,
           "rg_layer_fwd: NULL argument");
  RG_CHECK(g->n_ent == f->n_ent, "rg_layer_fwd: graph has %d entities, frontier %d", g->n_ent, f->n_ent);
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1,
           "rg_layer_fwd: level %d not resident (current %d, %d kept)", level, f->level, f->n_levels);
  RG_CHECK(n_new == f->n_nodes[level % f->n_levels], "rg_layer_fwd: n_new=%lld but level %d has %lld nodes",
           (long long)n_new, level, (long long)f->n_nodes[level % f->n_levels]);
  RG_CHECK(d > 0 && ld >= d && ld % 4 == 0 && ld >= 16 && ld <= 256, "rg_layer_fwd: d=%d ld=%d (need ld%%4==0, 16<=ld<=256)", d, ld);
  RG_CHECK(attn_dim > 0 && ap >= attn_dim && ap % 4 == 0, "rg_layer_fwd: attn_dim=%d ap=%d", attn_dim, ap);
  RG_CHECK((((uintptr_t)hidden | (uintptr_t)rela | (uintptr_t)a_s | (uintptr_t)a_r | (uintptr_t)a_q | (uintptr_t)agg_out) & 15) == 0,
           "rg_layer_fwd: float buffers must be 16-B aligned");
  if (n_new == 0) return 0;
  FwdArgs A;
  A.nodes_new = nodes_new; A.n_new = n_new;
  A.in_ptr = g->in_ptr; A.in_hr = g->in_hr;
  A.bm_old = f->bm_of(level - 1); A.W = f->W;
  A.hidden = (const float4*)hidden; A.rela = (const float4*)rela; A.ld4 = ld / 4;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha; A.attn_dim = attn_dim;
  A.n_rela_rows = 2 * g->n_rel + 1; A.rela_in_lds = 0;
  A.agg = (float4*)agg_out; A.n_chunks = 0;
  hipStream_t s = (hipStream_t)stream;
  const int ld4 = ld / 4;
  if (ld4 <= 4) return launch_ap<4>(A, ap / 4, s);
  if (ld4 <= 8) return launch_ap<8>(A, ap / 4, s);
  if (ld4 <= 16) return launch_ap<16>(A, ap / 4, s);
  if (ld4 <= 32) return launch_ap<32>(A, ap / 4, s);
  return launch_ap<64>(A, ap / 4, s);
}
