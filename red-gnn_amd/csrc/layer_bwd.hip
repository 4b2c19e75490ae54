// Fused relational message passing, backward (adjoint of layer_fwd.hip).
// Replaces what autograd replays for Static/transductive/models.py:29-39 (index / add / Linear x3 /
// sigmoid / mul / scatter) on E-row temporaries.  Source-pull formulation: every node (b,h) of the
// previous frontier walks its KG out-edges (CSR by head); every out-edge of a visited node is an
// edge of the hop, its destination id is the popcount rank of (b,t) in the new frontier.
//
//   per edge e=(s,r,o):   m = H[s] + Rel[r];  z = relu(a_s[s] + a_r[r] + a_q[b]);  alpha = sigma(w.z + b_alpha)
//     g_alpha = <G[o], m>                    g_p  = g_alpha * alpha (1 - alpha)     g_z = g_p * w * 1[z>0]
//     dH[s]   += alpha G[o]   (registers, one store per source row: deterministic)
//     dA_s[s] += g_z          (registers -> one store per source)
//     dRel[r] += alpha G[o]   (2R+1 rows only: privatised in LDS, flushed once per block)
//     dA_r[r] += g_z          (LDS)            dw += g_p relu(z), db += g_p   (registers -> block reduce)
//   dA_q[b] = sum of dA_s over the nodes of query b is left to the caller (a segment sum).
// The projections a_s = H Ws^T etc. are differentiated by the caller (dense GEMMs).
#include "common.h"

namespace {

struct BwdArgs {
  const int32_t* nodes_old;
  int64_t n_old;
  const int32_t* out_ptr;
  const int2* out_rt;
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
  int rela_in_lds;
  const float4* grad_agg;
  float4* g_hidden;
  float* g_rela;  // [n_rela_rows][ld]
  float4* g_as;
  float* g_ar;    // [n_rela_rows][ap]
  float* g_w;
  float* g_b;
  int n_chunks;
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

template <int G, int AP4, int BLOCK>
__global__ __launch_bounds__(BLOCK) void layer_bwd_kernel(BwdArgs A) {
  extern __shared__ float4 lds[];
  constexpr int GPB = BLOCK / G;
  const int nr = A.n_rela_rows;
  float4* stage = lds;                      // [BLOCK] {o, r, alpha, g_alpha}
  float4* ar_l = stage + BLOCK;             // [nr][AP4]
  float4* w_l = ar_l + nr * AP4;            // [AP4]
  float4* gar_l = w_l + AP4;                // [nr][AP4]   grad a_r
  float4* red_l = gar_l + nr * AP4;         // [(BLOCK/64)][AP4 + 1] block reduction of dw, db
  float4* rela_l = red_l + (BLOCK / 64) * (AP4 + 1);   // [nr][G]  (optional)
  float4* grela_l = rela_l + nr * G;                   // [nr][G]  (optional)

  for (int i = threadIdx.x; i < nr * AP4; i += BLOCK) { ar_l[i] = A.a_r[i]; gar_l[i] = f4zero(); }
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
  if (A.rela_in_lds) {
    for (int i = threadIdx.x; i < nr * G; i += BLOCK) {
      const int r = i / G, c = i - r * G;
      rela_l[i] = c < A.ld4 ? A.rela[(int64_t)r * A.ld4 + c] : f4zero();
      grela_l[i] = f4zero();
    }
  }
  __syncthreads();
  const float b_alpha = A.b_alpha[0];

  const int lane_g = threadIdx.x & (G - 1);
  const int gi = threadIdx.x / G;
  float4* my_stage = stage + gi * G;
  const bool row_lane = lane_g < A.ld4;

  float4 gw[AP4];
#pragma unroll
  for (int k = 0; k < AP4; ++k) gw[k] = f4zero();
  float gb = 0.f;

  const int x = blockIdx.x & 7, j0 = blockIdx.x >> 3, nbx = gridDim.x >> 3;
  const int cpx = (A.n_chunks + 7) >> 3;
  const int c_end = min((x + 1) * cpx, A.n_chunks);

  for (int chunk = x * cpx + j0; chunk < c_end; chunk += nbx) {
    const int64_t item = (int64_t)chunk * GPB + gi;
    const bool live = item < A.n_old;
    int b = 0, h = 0, beg = 0, end = 0;
    if (live) {
      b = A.nodes_old[2 * item];
      h = A.nodes_old[2 * item + 1];
      beg = A.out_ptr[h];
      end = A.out_ptr[h + 1];
    }
    float4 base[AP4], gas[AP4];
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      float4 as = live ? A.a_s[item * AP4 + k] : f4zero();
      const float4 aq = live ? A.a_q[(int64_t)b * AP4 + k] : f4zero();
      base[k] = make_float4(as.x + aq.x, as.y + aq.y, as.z + aq.z, as.w + aq.w);
      gas[k] = f4zero();
    }
    const float4 hs = (live && row_lane) ? A.hidden[item * A.ld4 + lane_g] : f4zero();
    const int2* bm_row = A.bm_new + (int64_t)b * A.W;
    float4 acc = f4zero();

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
        const int2 rt = A.out_rt[c];
        r = rt.x;
        const int2 wp = bm_row[rt.y >> 5];
        o = wp.y + __popc((uint32_t)wp.x & ((1u << (rt.y & 31)) - 1u));
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
        alpha = 1.0f / (1.0f + expf(-z));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      my_stage[lane_g] = make_float4(__int_as_float(o), __int_as_float(r), alpha, 0.f);  // pad lanes: alpha 0, row 0
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();

      // ---- phase 2: one edge per group step -------------------------------------------------------
      for (int k = 0; k < cnt; k += 2) {
        float4 tp[2], gv[2], rv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) tp[u] = my_stage[k + u];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ou = __float_as_int(tp[u].x), ru = __float_as_int(tp[u].y);
          gv[u] = row_lane ? A.grad_agg[(int64_t)ou * A.ld4 + lane_g] : f4zero();
          rv[u] = A.rela_in_lds ? rela_l[ru * G + lane_g] : (row_lane ? A.rela[(int64_t)ru * A.ld4 + lane_g] : f4zero());
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float al = tp[u].z;
          const int ru = __float_as_int(tp[u].y);
          float dot = gv[u].x * (hs.x + rv[u].x);
          dot = fmaf(gv[u].y, hs.y + rv[u].y, dot);
          dot = fmaf(gv[u].z, hs.z + rv[u].z, dot);
          dot = fmaf(gv[u].w, hs.w + rv[u].w, dot);
          dot = group_sum<G>(dot);
          if (lane_g == 0) reinterpret_cast<float*>(&my_stage[k + u])[3] = dot;
          const float4 ag = make_float4(al * gv[u].x, al * gv[u].y, al * gv[u].z, al * gv[u].w);
          acc.x += ag.x; acc.y += ag.y; acc.z += ag.z; acc.w += ag.w;
          if (al != 0.f && row_lane) {
            if (A.rela_in_lds) {
              float* gr = reinterpret_cast<float*>(&grela_l[ru * G + lane_g]);
              atomicAdd(gr + 0, ag.x); atomicAdd(gr + 1, ag.y); atomicAdd(gr + 2, ag.z); atomicAdd(gr + 3, ag.w);
            } else {
              float* gr = A.g_rela + ((int64_t)ru * A.ld4 + lane_g) * 4;
              atomicAdd(gr + 0, ag.x); atomicAdd(gr + 1, ag.y); atomicAdd(gr + 2, ag.z); atomicAdd(gr + 3, ag.w);
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
    // ---- per-source results -------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      gas[k].x = group_sum<G>(gas[k].x);
      gas[k].y = group_sum<G>(gas[k].y);
      gas[k].z = group_sum<G>(gas[k].z);
      gas[k].w = group_sum<G>(gas[k].w);
    }
    if (live) {
      if (row_lane) A.g_hidden[item * A.ld4 + lane_g] = acc;
      if (lane_g == 0) {
#pragma unroll
        for (int k = 0; k < AP4; ++k) A.g_as[item * AP4 + k] = gas[k];
      }
    }
  }

  // ---- block-level flushes ------------------------------------------------------------------------------
  __syncthreads();
  for (int i = threadIdx.x; i < nr * AP4 * 4; i += BLOCK) {
    const float v = reinterpret_cast<float*>(gar_l)[i];
    if (v != 0.f) atomicAdd(A.g_ar + i, v);
  }
  if (A.rela_in_lds) {
    for (int i = threadIdx.x; i < nr * A.ld4 * 4; i += BLOCK) {
      const int r = i / (A.ld4 * 4), c = i - r * (A.ld4 * 4);
      const float v = reinterpret_cast<float*>(grela_l)[r * G * 4 + c];
      if (v != 0.f) atomicAdd(A.g_rela + i, v);
    }
  }
  // dw, db: wave reduce -> LDS -> thread 0
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
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
    else if (threadIdx.x < A.attn_dim && v != 0.f) atomicAdd(A.g_w + threadIdx.x, v);
  }
}

template <int G, int AP4>
int launch(const BwdArgs& A, hipStream_t s) {
  constexpr int BLOCK = 512;
  constexpr int GPB = BLOCK / G;
  BwdArgs a = A;
  a.n_chunks = (int)rg::ceil_div(A.n_old, GPB);
  size_t lds = (size_t)(BLOCK + 2 * A.n_rela_rows * AP4 + AP4 + (BLOCK / 64) * (AP4 + 1)) * sizeof(float4);
  const size_t rela_bytes = 2 * (size_t)A.n_rela_rows * G * sizeof(float4);
  a.rela_in_lds = (lds + rela_bytes <= 80 * 1024) ? 1 : 0;
  if (a.rela_in_lds) lds += rela_bytes;
  RG_CHECK(lds <= 160 * 1024, "rg_layer_bwd: attention tables need %zu B of LDS", lds);
  if (lds > 64 * 1024)
    RG_HIP(hipFuncSetAttribute((const void*)layer_bwd_kernel<G, AP4, BLOCK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int grid = (int)std::min<int64_t>(a.n_chunks, 256 * 2);
  grid = (grid + 7) & ~7;
  hipLaunchKernelGGL((layer_bwd_kernel<G, AP4, BLOCK>), dim3(grid), dim3(BLOCK), lds, s, a);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int G>
int launch_ap(const BwdArgs& A, int ap4, hipStream_t s) {
  switch (ap4) {
    case 1: return launch<G, 1>(A, s);
    case 2: return launch<G, 2>(A, s);
    case 3: return launch<G, 3>(A, s);
    case 4: return launch<G, 4>(A, s);
    case 8: return launch<G, 8>(A, s);
    default: rg::set_error("rg_layer_bwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

}  // namespace

extern "C" int rg_layer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, const int32_t* nodes_old,
                            int64_t n_old, const float* hidden, const float* rela, int32_t d, int32_t ld,
                            const float* a_s, const float* a_r, const float* a_q, int32_t ap, const float* w_alpha,
                            const float* b_alpha, int32_t attn_dim, const float* grad_agg, float* grad_hidden,
                            float* grad_rela, float* grad_a_s, float* grad_a_r, float* grad_a_q, float* grad_w_alpha,
                            float* grad_b_alpha, void* stream) {
  RG_CHECK(f && g && nodes_old && hidden && rela && a_s && a_r && a_q && w_alpha && b_alpha && grad_agg && grad_hidden &&
               grad_rela && grad_a_s && grad_a_r && grad_w_alpha && grad_b_alpha,
           "rg_layer_bwd: NULL argument");
  (void)grad_a_q;  // dA_q[b] = segment sum of dA_s over the nodes of query b: done by the caller
  RG_CHECK(g->n_ent == f->n_ent, "rg_layer_bwd: graph has %d entities, frontier %d", g->n_ent, f->n_ent);
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1,
           "rg_layer_bwd: level %d not resident (current %d, %d kept)", level, f->level, f->n_levels);
  RG_CHECK(n_old == f->n_nodes[(level - 1) % f->n_levels], "rg_layer_bwd: n_old=%lld but level %d has %lld nodes",
           (long long)n_old, level - 1, (long long)f->n_nodes[(level - 1) % f->n_levels]);
  RG_CHECK(d > 0 && ld >= d && ld % 4 == 0 && ld >= 16 && ld <= 256, "rg_layer_bwd: d=%d ld=%d", d, ld);
  RG_CHECK(attn_dim > 0 && ap >= attn_dim && ap % 4 == 0, "rg_layer_bwd: attn_dim=%d ap=%d", attn_dim, ap);
  if (n_old == 0) return 0;
  BwdArgs A;
  A.nodes_old = nodes_old; A.n_old = n_old;
  A.out_ptr = g->out_ptr; A.out_rt = g->out_rt;
  A.bm_new = f->bm_of(level); A.W = f->W;
  A.hidden = (const float4*)hidden; A.rela = (const float4*)rela; A.ld4 = ld / 4;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha; A.attn_dim = attn_dim;
  A.n_rela_rows = 2 * g->n_rel + 1; A.rela_in_lds = 0;
  A.grad_agg = (const float4*)grad_agg; A.g_hidden = (float4*)grad_hidden; A.g_rela = grad_rela;
  A.g_as = (float4*)grad_a_s; A.g_ar = grad_a_r; A.g_w = grad_w_alpha; A.g_b = grad_b_alpha; A.n_chunks = 0;
  hipStream_t s = (hipStream_t)stream;
  const int ld4 = ld / 4;
  if (ld4 <= 4) return launch_ap<4>(A, ap / 4, s);
  if (ld4 <= 8) return launch_ap<8>(A, ap / 4, s);
  if (ld4 <= 16) return launch_ap<16>(A, ap / 4, s);
  if (ld4 <= 32) return launch_ap<32>(A, ap / 4, s);
  return launch_ap<64>(A, ap / 4, s);
}
