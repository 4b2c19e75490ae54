// Fused relational message passing, backward (adjoint of layer_fwd.hip).
// Replaces what autograd replays for Static/transductive/models.py:29-39 (index / add / Linear x3 /
// sigmoid / mul / scatter) on E-row temporaries.  Source-pull formulation: every node (b,h) of the
// previous frontier walks its KG out-edges (CSR by head, cut into length-sorted virtual rows exactly as
// the forward's CSR by tail); every out-edge of a visited node is an edge of the hop, its destination id
// is the popcount rank of (b,t) in the new frontier.
//
//   per edge e=(s,r,o):   m = H[s] + Rel[r];  z = relu(a_s[s] + a_r[r] + a_q[b]);  alpha = sigma(w.z + b_alpha)
//     g_alpha = <G[o], m>                    g_p  = g_alpha * alpha (1 - alpha)     g_z = g_p * w * 1[z>0]
//     dH[s]   += alpha G[o]   (registers, one store per source row / segment: deterministic)
//     dA_s[s] += g_z          (registers -> one store per source / segment)
//     dRel[r] += alpha G[o]   (2R+1 rows only: privatised in LDS, flushed once per workgroup)
//     dA_r[r] += g_z          (LDS)            dw += g_p relu(z), db += g_p   (registers -> block reduce)
//   dA_q[b] = sum of dA_s over the nodes of query b is left to the caller (a segment sum).
// The projections a_s = H Ws^T etc. are differentiated by the caller (dense GEMMs).
// Work distribution: walk.h (in-order per-XCD queues; grad_agg rows of the query being processed stay in L2).
#include <stdlib.h>

#include "aq_sum.h"
#include "walk.h"

namespace {

struct BwdArgs {
  rg::WalkArgs walk;   // items tested against the OLD frontier (sources); vrows = CSR-by-head segments
  const int2* out_rt;
  const uint32_t* out_pk;
  const int2* bm_new;
  int W;
  const float4* hidden;
  const float4* rela;
  int ld4;
  const float4* a_s;
  const float4* a_r;
  const float4* a_q;
  const float* w_alpha;
  const float* b_alpha;
  int attn_dim;
  int n_rela_rows;
  const float4* grad_agg;
  float4* g_hidden;
  float4* g_hidden_part;  // [B*n_slots][ld4]
  float4* g_as;
  float4* g_as_part;      // [B*n_slots][AP4]
  float* g_rela;          // [n_rela_rows][ld]
  float* g_ar;            // [n_rela_rows][ap]
  float* g_w;
  float* g_b;
  int kpg;    // walk.h: items per lane group of the dense walk (8 on short-row graphs)
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(t);
}

// sum over the G lanes of a group; every lane gets the total
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (G >= 16) {
    v = dpp_add<0x128>(v);  // row_ror:8
    v = dpp_add<0x124>(v);  // row_ror:4
    v = dpp_add<0x122>(v);  // row_ror:2
    v = dpp_add<0x121>(v);  // row_ror:1
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
  } else {
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  }
  return v;
}

constexpr int BWD_BLOCK = 512;

// DREL: accumulate dRel inside this kernel (run-length + LDS atomics).  When false the separate relation-major pass
// (drel_kernel below) computes it and this kernel only needs rela rows for the attention gradient's dot product.
// (AP4 >= 4, attn_dim > 12: 4 x AP4 float4 of per-edge attention state; 256 VGPRs instead of spilling at 128)
// KPG: items per lane group and block step of the dense walk (walk.h): 8 on graphs of short rows, as in the forward
template <int G, int AP4, bool PACKED, bool DENSE, bool RELA_LDS, bool DREL, int KPG = 1>
__global__ __launch_bounds__(BWD_BLOCK, AP4 >= 4 ? 2 : 4) void layer_bwd_kernel(BwdArgs A) {
  extern __shared__ float4 lds[];
  constexpr int BLOCK = BWD_BLOCK;
  const int nr = A.n_rela_rows;
  float4* stage = lds;                      // [BLOCK] {o, r, alpha, g_alpha}
  float4* ar_l = stage + BLOCK;             // [nr][AP4]
  float4* w_l = ar_l + nr * AP4;            // [AP4]
  float4* gar_l = w_l + AP4;                // [nr][AP4]   grad a_r
  float4* red_l = gar_l + nr * AP4;         // [(BLOCK/64)][AP4 + 1] block reduction of dw, db
  float4* rela_l = red_l + (BLOCK / 64) * (AP4 + 1);            // [nr][G]  (RELA_LDS)
  // grad rela, component-major inside a row and rows 8 floats apart in bank space: the 16 lanes of a group add to 16
  // consecutive banks and four groups working on four different relations do not collide (ds_add_f32, 32 banks)
  constexpr int RS = 4 * G + 8;
  float* grela_l = reinterpret_cast<float*>(rela_l + (RELA_LDS ? nr * G : 0));   // [nr][RS]  (RELA_LDS)
  int4* recs = reinterpret_cast<int4*>(grela_l + ((RELA_LDS && DREL) ? ((nr * RS + 3) & ~3) : 0));   // [BLOCK] (SPARSE only)

  for (int i = threadIdx.x; i < nr * AP4; i += BLOCK) { ar_l[i] = A.a_r[i]; gar_l[i] = f4zero(); }
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if constexpr (RELA_LDS) {
    for (int i = threadIdx.x; i < nr * G; i += BLOCK) {
      const int r = i / G, c = i - r * G;
      rela_l[i] = c < A.ld4 ? A.rela[(int64_t)r * A.ld4 + c] : f4zero();
    }
    if constexpr (DREL) { for (int i = threadIdx.x; i < nr * RS; i += BLOCK) grela_l[i] = 0.f; }
  }
  __syncthreads();
  const float b_alpha = A.b_alpha[0];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane_g = lane & (G - 1), gi_w = lane / G;
  float4* my_stage = stage + wv * 64 + gi_w * G;
  const bool row_lane = lane_g < A.ld4;
  const int lane_c = row_lane ? lane_g : A.ld4 - 1;

  float4 gw[AP4];
#pragma unroll
  for (int k = 0; k < AP4; ++k) gw[k] = f4zero();
  float gb = 0.f;

  rg::walk_items<G, DENSE, KPG, BLOCK>(A.walk, recs, [&](const int4& R, bool live) {
    const int beg = R.x, end = R.x + rg::walk_len(R), b = R.z, s_node = R.w;
    float4 base[AP4], gas[AP4];
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      const float4 as = A.a_s[(int64_t)s_node * AP4 + k];
      const float4 aq = A.a_q[(int64_t)b * AP4 + k];
      base[k] = make_float4(as.x + aq.x, as.y + aq.y, as.z + aq.z, as.w + aq.w);
      gas[k] = f4zero();
    }
    const float4 hs = A.hidden[(int64_t)s_node * A.ld4 + lane_c];
    const int2* bm_row = A.bm_new + (int64_t)b * A.W;
    float4 acc = f4zero();
    // dRel: out-edges arrive sorted by relation, so alpha*G is summed in registers over a run of equal relation
    // and added to the LDS copy once per run (float LDS atomics are the slowest thing in this kernel)
    int run_r = -1;
    float4 racc = f4zero();
    auto flush_run = [&]() {
      if (DREL && run_r >= 0 && row_lane) {
        if constexpr (RELA_LDS) {
          float* gr = grela_l + run_r * RS + lane_g;
          atomicAdd(gr, racc.x); atomicAdd(gr + G, racc.y); atomicAdd(gr + 2 * G, racc.z); atomicAdd(gr + 3 * G, racc.w);
        } else {
          float* gr = A.g_rela + ((int64_t)run_r * A.ld4 + lane_g) * 4;
          atomicAdd(gr + 0, racc.x); atomicAdd(gr + 1, racc.y); atomicAdd(gr + 2, racc.z); atomicAdd(gr + 3, racc.w);
        }
      }
    };

    for (int c0 = beg; c0 < end; c0 += G) {
      // ---- phase 1: one out-edge per lane: destination id, attention ----------------------------
      const int c = c0 + lane_g;
      const bool valid = c < end;
      const int cnt = min(G, end - c0);
      int o = 0, r = 0;
      float alpha = 0.f;
      float4 zr[AP4];
#pragma unroll
      for (int k = 0; k < AP4; ++k) zr[k] = f4zero();
      if (valid) {
        int tl;
        if constexpr (PACKED) { const uint32_t pk = A.out_pk[c]; tl = pk & 0xFFFFF; r = pk >> 20; }
        else { const int2 rt = A.out_rt[c]; r = rt.x; tl = rt.y; }
        const int2 wp = bm_row[tl >> 5];
        o = wp.y + __popc((uint32_t)wp.x & ((1u << (tl & 31)) - 1u));
        float z = b_alpha;
#pragma unroll
        for (int k = 0; k < AP4; ++k) {
          const float4 ar = ar_l[r * AP4 + k];
          const float4 w = w_l[k];
          zr[k] = make_float4(fmaxf(base[k].x + ar.x, 0.f), fmaxf(base[k].y + ar.y, 0.f),
                              fmaxf(base[k].z + ar.z, 0.f), fmaxf(base[k].w + ar.w, 0.f));
          z = fmaf(w.x, zr[k].x, z);
          z = fmaf(w.y, zr[k].y, z);
          z = fmaf(w.z, zr[k].z, z);
          z = fmaf(w.w, zr[k].w, z);
        }
        alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      my_stage[lane_g] = make_float4(__int_as_float(o), __int_as_float(r), alpha, 0.f);  // pad lanes: alpha 0, row 0
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();

      // ---- phase 2: one edge per group step, 4 grad rows in flight ------------------------------------
      for (int k = 0; k < cnt; k += 4) {
        float4 tp[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tp[u] = my_stage[k + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = A.grad_agg[(int64_t)__float_as_int(tp[u].x) * A.ld4 + lane_c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float al = tp[u].z;
          const int ru = __float_as_int(tp[u].y);
          float4 rv;
          if constexpr (RELA_LDS) rv = rela_l[ru * G + lane_g];
          else rv = A.rela[(int64_t)ru * A.ld4 + lane_c];
          float dot = 0.f;
          if (row_lane) {
            dot = gv[u].x * (hs.x + rv.x);
            dot = fmaf(gv[u].y, hs.y + rv.y, dot);
            dot = fmaf(gv[u].z, hs.z + rv.z, dot);
            dot = fmaf(gv[u].w, hs.w + rv.w, dot);
          }
          dot = group_sum<G>(dot);
          if (lane_g == 0) reinterpret_cast<float*>(&my_stage[k + u])[3] = dot;
          const float4 ag = make_float4(al * gv[u].x, al * gv[u].y, al * gv[u].z, al * gv[u].w);
          acc.x += ag.x; acc.y += ag.y; acc.z += ag.z; acc.w += ag.w;
          if constexpr (DREL) {
            if (al != 0.f) {
              if (ru != run_r) { flush_run(); run_r = ru; racc = f4zero(); }
              racc.x += ag.x; racc.y += ag.y; racc.z += ag.z; racc.w += ag.w;
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();

      // ---- phase 3: back to one edge per lane: attention gradients --------------------------------
      if (valid) {
        const float g_alpha = reinterpret_cast<const float*>(&my_stage[lane_g])[3];
        const float g_p = g_alpha * alpha * (1.0f - alpha);
        gb += g_p;
#pragma unroll
        for (int k = 0; k < AP4; ++k) {
          const float4 w = w_l[k];
          gw[k].x = fmaf(g_p, zr[k].x, gw[k].x);
          gw[k].y = fmaf(g_p, zr[k].y, gw[k].y);
          gw[k].z = fmaf(g_p, zr[k].z, gw[k].z);
          gw[k].w = fmaf(g_p, zr[k].w, gw[k].w);
          const float4 gz = make_float4(zr[k].x > 0.f ? g_p * w.x : 0.f, zr[k].y > 0.f ? g_p * w.y : 0.f,
                                        zr[k].z > 0.f ? g_p * w.z : 0.f, zr[k].w > 0.f ? g_p * w.w : 0.f);
          gas[k].x += gz.x; gas[k].y += gz.y; gas[k].z += gz.z; gas[k].w += gz.w;
          float* ga = reinterpret_cast<float*>(&gar_l[r * AP4 + k]);
          if (gz.x != 0.f) atomicAdd(ga + 0, gz.x);
          if (gz.y != 0.f) atomicAdd(ga + 1, gz.y);
          if (gz.z != 0.f) atomicAdd(ga + 2, gz.z);
          if (gz.w != 0.f) atomicAdd(ga + 3, gz.w);
        }
      }
    }
    flush_run();
    // ---- per-source (or per-segment) results -------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      gas[k].x = group_sum<G>(gas[k].x);
      gas[k].y = group_sum<G>(gas[k].y);
      gas[k].z = group_sum<G>(gas[k].z);
      gas[k].w = group_sum<G>(gas[k].w);
    }
    if (live) {
      const int out = rg::walk_out(R, A.walk.n_slots);
      float4* hrow = out >= 0 ? A.g_hidden + (int64_t)out * A.ld4 : A.g_hidden_part + (int64_t)(-out - 1) * A.ld4;
      float4* arow = out >= 0 ? A.g_as + (int64_t)out * AP4 : A.g_as_part + (int64_t)(-out - 1) * AP4;
      if (row_lane) hrow[lane_g] = acc;
      if (lane_g == 0) {
#pragma unroll
        for (int k = 0; k < AP4; ++k) arow[k] = gas[k];
      }
    }
  });

  // ---- block-level flushes ------------------------------------------------------------------------------
  __syncthreads();
  for (int i = threadIdx.x; i < nr * AP4 * 4; i += BLOCK) {
    const float v = reinterpret_cast<float*>(gar_l)[i];
    if (v != 0.f) atomicAdd(A.g_ar + i, v);
  }
  if constexpr (RELA_LDS && DREL) {
    for (int i = threadIdx.x; i < nr * A.ld4 * 4; i += BLOCK) {
      const int r = i / (A.ld4 * 4), c = i - r * (A.ld4 * 4);
      const float v = grela_l[r * RS + (c & 3) * G + (c >> 2)];
      if (v != 0.f) atomicAdd(A.g_rela + i, v);
    }
  }
  // dw, db: wave reduce -> LDS -> first threads
  float vals[AP4 * 4 + 1];
#pragma unroll
  for (int k = 0; k < AP4; ++k) { vals[4 * k] = gw[k].x; vals[4 * k + 1] = gw[k].y; vals[4 * k + 2] = gw[k].z; vals[4 * k + 3] = gw[k].w; }
  vals[AP4 * 4] = gb;
  float* red = reinterpret_cast<float*>(red_l);
#pragma unroll
  for (int i = 0; i < AP4 * 4 + 1; ++i) {
    float v = vals[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv * (AP4 * 4 + 4) + i] = v;
  }
  __syncthreads();
  if (threadIdx.x < AP4 * 4 + 1) {
    float v = 0.f;
    for (int w = 0; w < BLOCK / 64; ++w) v += red[w * (AP4 * 4 + 4) + threadIdx.x];
    if (threadIdx.x == AP4 * 4) { if (v != 0.f) atomicAdd(A.g_b, v); }
    else if ((int)threadIdx.x < A.attn_dim && v != 0.f) atomicAdd(A.g_w + threadIdx.x, v);
  }
}


// ---- relation gradient, relation-major --------------------------------------------------------------------------
// dRel[r] = sum over queries b and KG edges (h, r, t) with (b,h) in the previous frontier of alpha * G[(b,t)].
// Items are (query, 128-edge segment of relation r's edge list): the segment's sum is built in registers exactly as
// the forward kernel builds a destination row (test + rank + attention per candidate lane, then row gathers of G), and
// added to the workgroup's LDS copy of dRel ONCE per segment - 1/128th of the LDS float atomics of the per-edge form,
// which were two thirds of the backward kernel's time.
struct DrelArgs {
  rg::WalkArgs walk;          // vrows = CSR-by-relation segments; always live
  const int2* rel_ht;
  const int2* bm_old;
  const int2* bm_new;
  int W;
  const float4* a_s;
  const float4* a_r;
  const float4* a_q;
  const float* w_alpha;
  const float* b_alpha;
  int attn_dim;
  int n_rela_rows;
  int ld4;
  const float4* grad_agg;
  float* g_rela;
};

// TABLE: the relation gradient is accumulated in an LDS copy of the table (one LDS row add per segment, one global add per
// block and row at the end).  When the table does not fit LDS (FB15k-237-like: 475 rows x 128) every segment's sum goes
// straight to global memory: still one row add per <= 128 edges.
template <int G, int AP4, bool TABLE>
__global__ __launch_bounds__(BWD_BLOCK, 4) void drel_kernel(DrelArgs A) {
  extern __shared__ float4 lds[];
  constexpr int BLOCK = BWD_BLOCK;
  constexpr int RS = 4 * G + 8;
  const int nr = A.n_rela_rows;
  float4* stage = lds;                                   // [BLOCK] {o, alpha}
  float4* ar_l = stage + BLOCK;                          // [nr][AP4]
  float4* w_l = ar_l + nr * AP4;                         // [AP4]
  float* grela_l = reinterpret_cast<float*>(w_l + AP4);  // [nr][RS]  (TABLE)
  for (int i = threadIdx.x; i < nr * AP4; i += BLOCK) ar_l[i] = A.a_r[i];
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if constexpr (TABLE) { for (int i = threadIdx.x; i < nr * RS; i += BLOCK) grela_l[i] = 0.f; }
  __syncthreads();
  const float b_alpha = A.b_alpha[0];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lane_g = lane & (G - 1), gi_w = lane / G;
  float4* my_stage = stage + wv * 64 + gi_w * G;
  const int gshift = lane & ~(G - 1);
  const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
  const bool row_lane = lane_g < A.ld4;
  const int lane_c = row_lane ? lane_g : A.ld4 - 1;

  rg::walk_items<G, true, 1, BLOCK, true>(A.walk, nullptr, [&](const int4& R, bool live) {
    const int beg = R.x, end = R.x + rg::walk_len(R), b = R.z, r = R.w;
    const int2* old_row = A.bm_old + (int64_t)b * A.W;
    const int2* new_row = A.bm_new + (int64_t)b * A.W;
    float4 base[AP4];
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      const float4 ar = ar_l[(live ? r : 0) * AP4 + k];
      const float4 aq = A.a_q[(int64_t)b * AP4 + k];
      base[k] = make_float4(ar.x + aq.x, ar.y + aq.y, ar.z + aq.z, ar.w + aq.w);
    }
    float4 acc = f4zero();
    bool any = false;
    for (int c0 = beg; c0 < end; c0 += G) {
      const int c = c0 + lane_g;
      bool valid = c < end;
      int o = 0;
      float alpha = 0.f;
      if (valid) {
        const int2 ht = A.rel_ht[c];
        const int2 wp = old_row[ht.x >> 5];
        const uint32_t word = (uint32_t)wp.x, bit = ht.x & 31;
        valid = (word >> bit) & 1u;
        if (valid) {
          const int s = wp.y + __popc(word & ((1u << bit) - 1u));
          const int2 wn = new_row[ht.y >> 5];
          o = wn.y + __popc((uint32_t)wn.x & ((1u << (ht.y & 31)) - 1u));
          float z = b_alpha;
#pragma unroll
          for (int k = 0; k < AP4; ++k) {
            const float4 as = A.a_s[(int64_t)s * AP4 + k];
            const float4 w = w_l[k];
            z = fmaf(w.x, fmaxf(as.x + base[k].x, 0.f), z);
            z = fmaf(w.y, fmaxf(as.y + base[k].y, 0.f), z);
            z = fmaf(w.z, fmaxf(as.z + base[k].z, 0.f), z);
            z = fmaf(w.w, fmaxf(as.w + base[k].w, 0.f), z);
          }
          alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        }
      }
      const unsigned long long m = (__ballot(valid) >> gshift) & gmask;
      const int cnt = __popcll(m);
      const int pos = __popcll(m & ((1ull << lane_g) - 1ull));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (lane_g >= cnt) my_stage[lane_g] = f4zero();
      if (valid) my_stage[pos] = make_float4(__int_as_float(o), alpha, 0.f, 0.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      any = any || cnt > 0;
      for (int k = 0; k < cnt; k += 4) {
        float4 tp[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tp[u] = my_stage[k + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = A.grad_agg[(int64_t)__float_as_int(tp[u].x) * A.ld4 + lane_c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float al = tp[u].y;
          acc.x = fmaf(al, gv[u].x, acc.x);
          acc.y = fmaf(al, gv[u].y, acc.y);
          acc.z = fmaf(al, gv[u].z, acc.z);
          acc.w = fmaf(al, gv[u].w, acc.w);
        }
      }
    }
    if (live && any && row_lane) {
      if constexpr (TABLE) {
        float* gr = grela_l + r * RS + lane_g;
        atomicAdd(gr, acc.x); atomicAdd(gr + G, acc.y); atomicAdd(gr + 2 * G, acc.z); atomicAdd(gr + 3 * G, acc.w);
      } else {
        float* gr = A.g_rela + ((int64_t)r * A.ld4 + lane_g) * 4;
        atomicAdd(gr + 0, acc.x); atomicAdd(gr + 1, acc.y); atomicAdd(gr + 2, acc.z); atomicAdd(gr + 3, acc.w);
      }
    }
  });

  if constexpr (TABLE) {
    __syncthreads();
    for (int i = threadIdx.x; i < nr * A.ld4 * 4; i += BLOCK) {
      const int r = i / (A.ld4 * 4), c = i - r * (A.ld4 * 4);
      const float v = grela_l[r * RS + (c & 3) * G + (c >> 2)];
      if (v != 0.f) atomicAdd(A.g_rela + i, v);
    }
  }
}

template <int G, int AP4, bool TABLE>
int launch_drel_t(const DrelArgs& A, hipStream_t s) {
  const size_t lds = (size_t)(BWD_BLOCK + A.n_rela_rows * AP4 + AP4) * sizeof(float4) +
                     (TABLE ? (size_t)A.n_rela_rows * (4 * G + 8) * sizeof(float) : 0);
  RG_CHECK(lds <= 160 * 1024, "rg_layer_bwd: attention table needs %zu B of LDS (> 160 KiB)", lds);
  auto kern = drel_kernel<G, AP4, TABLE>;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = rg::walk_grid(A.walk.n_items, BWD_BLOCK, G, true, lds <= 80 * 1024 ? 2 : 1, 1);
  if (rg::zero_async(A.walk.queues, RG_QUEUE_BYTES, s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int G, int AP4>
int launch_drel(const DrelArgs& A, hipStream_t s) {
  const size_t table_lds = (size_t)(BWD_BLOCK + A.n_rela_rows * AP4 + AP4) * sizeof(float4) + (size_t)A.n_rela_rows * (4 * G + 8) * sizeof(float);
  return table_lds <= 80 * 1024 ? launch_drel_t<G, AP4, true>(A, s) : launch_drel_t<G, AP4, false>(A, s);
}

template <int G>
int launch_drel_ap(const DrelArgs& A, int ap4, hipStream_t s) {
  switch (ap4) {
    case 1: return launch_drel<G, 1>(A, s);
    case 2: return launch_drel<G, 2>(A, s);
    case 3: return launch_drel<G, 3>(A, s);
    case 4: return launch_drel<G, 4>(A, s);
    case 8: return launch_drel<G, 8>(A, s);
    default: rg::set_error("rg_layer_bwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

// hub sources cut into segments: dH[s], dA_s[s] = sums of the segments' partial rows, in segment order
__global__ void bwd_combine_kernel(const int4* __restrict__ split, int n_split, int n_slots, int B, const int2* __restrict__ bm_old,
                                   int W, const float4* __restrict__ hpart, const float4* __restrict__ apart,
                                   float4* __restrict__ g_hidden, float4* __restrict__ g_as, int ld4, int ap4) {
  const int cols = ld4 + ap4;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t item = tid / cols;
  const int c = (int)(tid - item * cols);
  if (item >= (int64_t)B * n_split) return;
  const int b = (int)(item / n_split);
  const int4 se = split[item - (int64_t)b * n_split];
  const int2 wp = bm_old[(int64_t)b * W + (se.x >> 5)];
  const uint32_t word = (uint32_t)wp.x, bit = se.x & 31;
  if (!((word >> bit) & 1u)) return;
  const int s = wp.y + __popc(word & ((1u << bit) - 1u));
  const bool is_h = c < ld4;
  const int stride = is_h ? ld4 : ap4;
  const float4* p = (is_h ? hpart : apart) + ((int64_t)b * n_slots + se.y) * stride + (is_h ? c : c - ld4);
  float4 acc = p[0];
  for (int k = 1; k < se.z; ++k) {
    const float4 v = p[(int64_t)k * stride];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (is_h) g_hidden[(int64_t)s * ld4 + c] = acc;
  else g_as[(int64_t)s * ap4 + (c - ld4)] = acc;
}

template <int G, int AP4, bool PACKED, bool DENSE, bool RELA_LDS, bool DREL, int KPG = 1>
int launch3(const BwdArgs& A, size_t lds, int B, const rg_vrows& vr, const int2* bm_old, hipStream_t s) {
  auto kern = layer_bwd_kernel<G, AP4, PACKED, DENSE, RELA_LDS, DREL, KPG>;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = lds <= 80 * 1024 ? 2 : 1;
  const int grid = rg::walk_grid(A.walk.n_items, BWD_BLOCK, G, DENSE, per_cu, KPG);
  if (rg::zero_async(A.walk.queues, RG_QUEUE_BYTES, s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BWD_BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  if (vr.n_split > 0) {
    const int64_t threads = (int64_t)B * vr.n_split * (A.ld4 + AP4);
    hipLaunchKernelGGL(bwd_combine_kernel, dim3(rg::ceil_div(threads, 256)), dim3(256), 0, s, vr.split, vr.n_split, vr.n_slots, B,
                       bm_old, A.W, A.g_hidden_part, A.g_as_part, A.g_hidden, A.g_as, A.ld4, AP4);
    RG_LAUNCH_CHECK();
  }
  return 0;
}

template <int G, int AP4, bool PACKED, bool DENSE>
int launch2(const BwdArgs& A, int B, const rg_vrows& vr, const int2* bm_old, hipStream_t s) {
  size_t lds = (size_t)(BWD_BLOCK + 2 * A.n_rela_rows * AP4 + AP4 + (BWD_BLOCK / 64) * (AP4 + 1)) * sizeof(float4);
  if (!DENSE) lds += (size_t)BWD_BLOCK * sizeof(int4);
  const size_t rela_bytes = (size_t)A.n_rela_rows * G * sizeof(float4) + (((size_t)A.n_rela_rows * (4 * G + 8) + 3) & ~(size_t)3) * sizeof(float);
  RG_CHECK(lds <= 160 * 1024, "rg_layer_bwd: attention tables need %zu B of LDS (> 160 KiB)", lds);
  const size_t rela_only = (size_t)A.n_rela_rows * G * sizeof(float4);
  // dRel comes from the relation-major pass (drel_kernel); here the rela rows are only read
  if constexpr (DENSE) {
    if (A.kpg > 1) {
      if (lds + rela_only <= 80 * 1024)
        return launch3<G, AP4, PACKED, true, true, false, rg::RG_KPG_SHORT>(A, lds + rela_only + 64, B, vr, bm_old, s);
      return launch3<G, AP4, PACKED, true, false, false, rg::RG_KPG_SHORT>(A, lds, B, vr, bm_old, s);
    }
  }
  if (lds + rela_only <= 80 * 1024) return launch3<G, AP4, PACKED, DENSE, true, false>(A, lds + rela_only + 64, B, vr, bm_old, s);
  return launch3<G, AP4, PACKED, DENSE, false, false>(A, lds, B, vr, bm_old, s);
}

template <int G, int AP4>
int launch(const BwdArgs& A, int B, const rg_vrows& vr, const int2* bm_old, bool dense, hipStream_t s) {
  if (A.out_pk) return dense ? launch2<G, AP4, true, true>(A, B, vr, bm_old, s) : launch2<G, AP4, true, false>(A, B, vr, bm_old, s);
  return dense ? launch2<G, AP4, false, true>(A, B, vr, bm_old, s) : launch2<G, AP4, false, false>(A, B, vr, bm_old, s);
}

template <int G>
int launch_ap(const BwdArgs& A, int ap4, int B, const rg_vrows& vr, const int2* bm_old, bool dense, hipStream_t s) {
  switch (ap4) {
    case 1: return launch<G, 1>(A, B, vr, bm_old, dense, s);
    case 2: return launch<G, 2>(A, B, vr, bm_old, dense, s);
    case 3: return launch<G, 3>(A, B, vr, bm_old, dense, s);
    case 4: return launch<G, 4>(A, B, vr, bm_old, dense, s);
    case 8: return launch<G, 8>(A, B, vr, bm_old, dense, s);
    default: rg::set_error("rg_layer_bwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

}  // namespace

extern "C" size_t rg_layer_bwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld, int32_t ap) {
  if (!f || !g) return 0;
  return (size_t)f->B * g->out_vr.n_slots * (ld + ap) * sizeof(float) + 512;
}

extern "C" int rg_layer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, const float* hidden,
                            const float* rela, int32_t d, int32_t ld, const float* a_s, const float* a_r,
                            const float* a_q, int32_t ap, const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                            const float* grad_agg, float* grad_hidden, float* grad_rela, float* grad_a_s,
                            float* grad_a_r, float* grad_a_q, float* grad_w_alpha, float* grad_b_alpha, void* scratch,
                            size_t scratch_bytes, void* stream) {
  RG_CHECK(f && g && hidden && rela && a_s && a_r && a_q && w_alpha && b_alpha && grad_agg && grad_hidden && grad_rela &&
               grad_a_s && grad_a_r && grad_w_alpha && grad_b_alpha, "rg_layer_bwd: NULL argument");
  RG_CHECK(g->n_ent == f->n_ent, "rg_layer_bwd: graph has %d entities, frontier %d", g->n_ent, f->n_ent);
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1,
           "rg_layer_bwd: level %d not resident (current %d, %d kept)", level, f->level, f->n_levels);
  RG_CHECK(n_old == f->n_nodes[(level - 1) % f->n_levels], "rg_layer_bwd: n_old=%lld but level %d has %lld nodes",
           (long long)n_old, level - 1, (long long)f->n_nodes[(level - 1) % f->n_levels]);
  RG_CHECK(d > 0 && ld >= d && ld % 4 == 0 && ld >= 16 && ld <= 256, "rg_layer_bwd: d=%d ld=%d", d, ld);
  RG_CHECK(attn_dim > 0 && ap >= attn_dim && ap % 4 == 0, "rg_layer_bwd: attn_dim=%d ap=%d", attn_dim, ap);
  const size_t need = rg_layer_bwd_scratch_bytes(f, g, ld, ap);
  RG_CHECK(g->out_vr.n_slots == 0 || (scratch && scratch_bytes >= need), "rg_layer_bwd: scratch %zu B < required %zu B",
           scratch_bytes, need);
  RG_CHECK((int64_t)f->B * std::max(g->out_vr.n_slots, 1) < ((int64_t)1 << 31) && g->out_vr.n_slots < (1 << 22),
           "rg_layer_bwd: batch * hub segments overflows int32");
  const int64_t n_items = (int64_t)f->B * g->out_vr.n;
  RG_CHECK(n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "rg_layer_bwd: work space too large for 32-bit queue tickets");
  if (n_old == 0) return grad_a_q ? rg::launch_aq_sum(f->bm_of(level - 1), f->W, f->B, f->n_ent, 0, grad_a_s, ap, grad_a_q, (hipStream_t)stream) : 0;
  BwdArgs A;
  A.walk.n_items = n_items; A.walk.n_vrows = g->out_vr.n; A.walk.n_slots = g->out_vr.n_slots; A.walk.vrows = g->out_vr.rows;
  A.walk.bm_test = f->bm_of(level - 1); A.walk.W = f->W; A.walk.queues = f->queues; f->queues_clean = false;
  A.out_rt = g->out_rt; A.out_pk = g->out_pk;
  A.bm_new = f->bm_of(level); A.W = f->W;
  A.hidden = (const float4*)hidden; A.rela = (const float4*)rela; A.ld4 = ld / 4;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha; A.attn_dim = attn_dim;
  A.n_rela_rows = g->n_rela_rows;
  A.grad_agg = (const float4*)grad_agg; A.g_hidden = (float4*)grad_hidden; A.g_rela = grad_rela;
  A.g_as = (float4*)grad_a_s; A.g_ar = grad_a_r; A.g_w = grad_w_alpha; A.g_b = grad_b_alpha;
  A.kpg = rg::walk_kpg(g->n_fact, g->out_vr.n);
  A.g_hidden_part = (float4*)scratch;
  A.g_as_part = (float4*)((char*)scratch + rg::align_up((size_t)f->B * g->out_vr.n_slots * ld * sizeof(float), 256));
  hipStream_t s = (hipStream_t)stream;
  // every hop but the first walks densely: a sparse source set still carries hub rows of thousands of edges, and the
  // 64-items-per-lane filter of the sparse walk hands them to a few workgroups (measured 10x slower on C2 hop 1)
  const bool dense = n_old >= 4 * (int64_t)f->B;
  const int ld4 = ld / 4;
  const int2* bm_old = f->bm_of(level - 1);
  int rc;
  if (ld4 <= 4) rc = launch_ap<4>(A, ap / 4, f->B, g->out_vr, bm_old, dense, s);
  else if (ld4 <= 8) rc = launch_ap<8>(A, ap / 4, f->B, g->out_vr, bm_old, dense, s);
  else if (ld4 <= 16) rc = launch_ap<16>(A, ap / 4, f->B, g->out_vr, bm_old, dense, s);
  else if (ld4 <= 32) rc = launch_ap<32>(A, ap / 4, f->B, g->out_vr, bm_old, dense, s);
  else rc = launch_ap<64>(A, ap / 4, f->B, g->out_vr, bm_old, dense, s);
  if (rc) return rc;
  if (grad_a_q && rg::launch_aq_sum(bm_old, f->W, f->B, f->n_ent, n_old, grad_a_s, ap, grad_a_q, s)) return 1;
  // relation gradient, relation-major (see drel_kernel)
  DrelArgs D;
  D.walk.n_items = (int64_t)f->B * g->rel_vr.n; D.walk.n_vrows = g->rel_vr.n; D.walk.n_slots = 0; D.walk.vrows = g->rel_vr.rows;
  D.walk.bm_test = nullptr; D.walk.W = f->W; D.walk.queues = f->queues; f->queues_clean = false;
  RG_CHECK(D.walk.n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "rg_layer_bwd: relation work space too large for 32-bit queue tickets");
  D.rel_ht = g->rel_ht; D.bm_old = bm_old; D.bm_new = f->bm_of(level); D.W = f->W;
  D.a_s = (const float4*)a_s; D.a_r = (const float4*)a_r; D.a_q = (const float4*)a_q;
  D.w_alpha = w_alpha; D.b_alpha = b_alpha; D.attn_dim = attn_dim; D.n_rela_rows = g->n_rela_rows; D.ld4 = ld4;
  D.grad_agg = (const float4*)grad_agg; D.g_rela = grad_rela;
  if (ld4 <= 4) return launch_drel_ap<4>(D, ap / 4, s);
  if (ld4 <= 8) return launch_drel_ap<8>(D, ap / 4, s);
  if (ld4 <= 16) return launch_drel_ap<16>(D, ap / 4, s);
  if (ld4 <= 32) return launch_drel_ap<32>(D, ap / 4, s);
  return launch_drel_ap<64>(D, ap / 4, s);
}
