// Temporal (T-RED-GNN interpolation) layer backward: adjoint of rg_tlayer_fwd.
// Replaces what autograd would replay for Temporal/interpolation/model_cuda.py:149-160,192 on E-row temporaries.
// Same source-pull structure as layer_bwd.hip (walk over CSR-by-head segments of the previous frontier's nodes), with
//   m_e = hidden_dir[3 s + dir] + rela_dir[dir * n_rela_rows + r] + time_dir[dir * n_time + |dt|],   dt = time(e) - q_time[b]
//   d hidden_dir[3 s + dir] += alpha G[o]      (three register accumulators per source, one store per row)
//   d rela_dir / d time_dir rows += alpha G[o] (separate key-major passes, tkey_kernel: items are 128-edge segments of one
//                                               relation / one time id, so a whole segment lands in at most three rows / one row
//                                               and is added once; per-edge float atomics on a 2.5 k-row table cost 20x the forward)
//   g_alpha = <G[o], m_e>  ->  d a_s[s], d a_r[r], d w   exactly as in the static kernel.
// The direction linears and the attention's three blocks are differentiated by the caller (dense GEMMs).
// WIN = true is the adjoint of rg_xlayer_fwd (temporal EXTRAPOLATION, Temporal/extrapolation/model_cuda_new_embedding.py:186-239): an
// edge's time field is its data row, valid for query b inside the row window [win_lo[b], win_hi[b]) only (self-loops, row >= n_data,
// always); one direction (every edge lies in the past: past_linear), time row = clamp(q_time[b] - row_time[row], 0, n_time - 1)
// (self-loops: q_time[b] - loop_time[b]); hidden_dir / rela_dir / time_dir are then hidden_p [N_old] / rela_p / time_p [n_time].
#include "aq_sum.h"
#include "walk.h"

namespace {

struct TBwdArgs {
  rg::WalkArgs walk;   // items tested against the OLD frontier (sources); vrows = CSR-by-head segments
  const int2* out_rt;
  const int32_t* out_time;
  const int32_t* q_time;
  const int2* bm_new;
  int W;
  const float4* hidden_dir;   // [3 * N_old][ld4]
  const float4* rela_dir;     // [3 * n_rela_rows][ld4]
  const float4* time_dir;     // [3 * n_time][ld4]
  int ld4, n_rela_rows, n_time;
  const float4* a_s;
  const float4* a_r;
  const float4* a_q;
  const float* w_alpha;
  const float* b_alpha;
  int attn_dim;
  const float4* grad_agg;
  float4* g_hidden_dir;       // [N_old][3 * ld4]
  float4* g_hidden_part;      // [B * n_slots][3 * ld4]
  float4* g_as;
  float4* g_as_part;
  float* g_rela_dir;          // [3 * n_rela_rows][ld]
  float* g_time_dir;          // [3 * n_time][ld]
  float* g_ar;
  float* g_w;
  // WIN
  const int32_t* win_lo = nullptr;
  const int32_t* win_hi = nullptr;
  const int32_t* row_time = nullptr;
  const int32_t* loop_time = nullptr;
  int n_data = 0;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(t);
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (G >= 16) {
    v = dpp_add<0x128>(v);
    v = dpp_add<0x124>(v);
    v = dpp_add<0x122>(v);
    v = dpp_add<0x121>(v);
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
  } else {
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  }
  return v;
}

constexpr int TB_BLOCK = 512;

// wide attention MLPs (AP4 >= 4: ICEWS presets use attn_dim = 30) keep 4 x AP4 float4 of per-edge state in registers: 256 VGPRs
// (2 waves per SIMD; their LDS tables leave room for one workgroup per CU anyway) instead of spilling at 128
template <int G, int AP4, bool DENSE, bool WIN>
__global__ __launch_bounds__(TB_BLOCK, AP4 >= 4 ? 2 : 4) void tlayer_bwd_kernel(TBwdArgs A) {
  extern __shared__ float4 lds[];
  constexpr int BLOCK = TB_BLOCK;
  constexpr int ND = WIN ? 1 : 3;          // direction rows per source node
  const int nr = A.n_rela_rows;
  float4* stage = lds;                      // [BLOCK] {o -> g_alpha, rela row, alpha, time row * 4 + dir}
  float4* ar_l = stage + BLOCK;             // [nr][AP4]
  float4* w_l = ar_l + nr * AP4;            // [AP4]
  float4* gar_l = w_l + AP4;                // [nr][AP4]
  float4* red_l = gar_l + nr * AP4;         // [(BLOCK/64)][AP4 + 1]
  int4* recs = reinterpret_cast<int4*>(red_l + (BLOCK / 64) * (AP4 + 1));   // [BLOCK] (SPARSE only)

  for (int i = threadIdx.x; i < nr * AP4; i += BLOCK) { ar_l[i] = A.a_r[i]; gar_l[i] = f4zero(); }
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
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

  rg::walk_items<G, DENSE, 1, BLOCK>(A.walk, recs, [&](const int4& R, bool live) {
    const int beg = R.x, end = R.x + rg::walk_len(R), b = R.z, s_node = R.w;
    const int qt = A.q_time[b];
    int wlo = 0, whi = 0, lt = 0;
    if constexpr (WIN) { wlo = A.win_lo[b]; whi = A.win_hi[b]; lt = A.loop_time[b]; }
    float4 base[AP4], gas[AP4];
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      const float4 as = A.a_s[(int64_t)s_node * AP4 + k];
      const float4 aq = A.a_q[(int64_t)b * AP4 + k];
      base[k] = make_float4(as.x + aq.x, as.y + aq.y, as.z + aq.z, as.w + aq.w);
      gas[k] = f4zero();
    }
    float4 hs[ND], acc[ND];
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
      hs[dd] = A.hidden_dir[((int64_t)s_node * ND + dd) * A.ld4 + lane_c];
      acc[dd] = f4zero();
    }
    const int2* bm_row = A.bm_new + (int64_t)b * A.W;

    for (int c0 = beg; c0 < end; c0 += G) {
      // ---- phase 1: one out-edge per lane ---------------------------------------------------------
      const int c = c0 + lane_g;
      bool valid = c < end;
      const int cnt = min(G, end - c0);
      int o = 0, r = 0, rrow = 0, tdir = 0;
      float alpha = 0.f;
      float4 zr[AP4];
#pragma unroll
      for (int k = 0; k < AP4; ++k) zr[k] = f4zero();
      int erow = 0;
      if constexpr (WIN) {          // an edge outside the query's window is no edge: it stays in the round as a pad (alpha = 0, row 0)
        if (valid) {
          erow = A.out_time[c];
          valid = erow >= A.n_data || (erow >= wlo && erow < whi);
        }
      }
      if (valid) {
        const int2 rt = A.out_rt[c];
        r = rt.x;
        const int2 wp = bm_row[rt.y >> 5];
        o = wp.y + __popc((uint32_t)wp.x & ((1u << (rt.y & 31)) - 1u));
        if constexpr (WIN) {
          const int delta = qt - (erow >= A.n_data ? lt : A.row_time[erow]);
          rrow = r;
          tdir = min(max(delta, 0), A.n_time - 1) * 4;
        } else {
          const int dt = A.out_time[c] - qt;
          const int dir = dt > 0 ? 2 : (dt == 0 ? 1 : 0);
          rrow = dir * nr + r;
          tdir = (dir * A.n_time + (dt < 0 ? -dt : dt)) * 4 + dir;
        }
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
      my_stage[lane_g] = make_float4(__int_as_float(o), __int_as_float(rrow), alpha, __int_as_float(tdir));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();

      // ---- phase 2: one edge per group step -----------------------------------------------------------
      for (int k = 0; k < cnt; k += 2) {
        float4 tp[2], gv[2], rv[2], tv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) tp[u] = my_stage[k + u];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          gv[u] = A.grad_agg[(int64_t)__float_as_int(tp[u].x) * A.ld4 + lane_c];
          rv[u] = A.rela_dir[(int64_t)__float_as_int(tp[u].y) * A.ld4 + lane_c];
          tv[u] = A.time_dir[(int64_t)(__float_as_int(tp[u].w) >> 2) * A.ld4 + lane_c];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float al = tp[u].z;
          const int dir = __float_as_int(tp[u].w) & 3;
          float4 hsel = hs[0];
          if constexpr (!WIN) hsel = dir == 0 ? hs[0] : (dir == 1 ? hs[1] : hs[2]);
          float dot = 0.f;
          if (row_lane) {
            dot = gv[u].x * (hsel.x + rv[u].x + tv[u].x);
            dot = fmaf(gv[u].y, hsel.y + rv[u].y + tv[u].y, dot);
            dot = fmaf(gv[u].z, hsel.z + rv[u].z + tv[u].z, dot);
            dot = fmaf(gv[u].w, hsel.w + rv[u].w + tv[u].w, dot);
          }
          dot = group_sum<G>(dot);
          if (lane_g == 0) reinterpret_cast<float*>(&my_stage[k + u])[0] = dot;     // o is consumed: slot reused for g_alpha
          const float4 ag = make_float4(al * gv[u].x, al * gv[u].y, al * gv[u].z, al * gv[u].w);
#pragma unroll
          for (int dd = 0; dd < ND; ++dd) {
            const float m = (WIN || dir == dd) ? 1.f : 0.f;
            acc[dd].x = fmaf(m, ag.x, acc[dd].x); acc[dd].y = fmaf(m, ag.y, acc[dd].y);
            acc[dd].z = fmaf(m, ag.z, acc[dd].z); acc[dd].w = fmaf(m, ag.w, acc[dd].w);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();

      // ---- phase 3: attention gradients, one edge per lane ----------------------------------------------
      if (valid) {
        const float g_alpha = reinterpret_cast<const float*>(&my_stage[lane_g])[0];
        const float g_p = g_alpha * alpha * (1.0f - alpha);
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
#pragma unroll
    for (int k = 0; k < AP4; ++k) {
      gas[k].x = group_sum<G>(gas[k].x);
      gas[k].y = group_sum<G>(gas[k].y);
      gas[k].z = group_sum<G>(gas[k].z);
      gas[k].w = group_sum<G>(gas[k].w);
    }
    if (live) {
      const int out = rg::walk_out(R, A.walk.n_slots);
      float4* hrow = out >= 0 ? A.g_hidden_dir + (int64_t)out * ND * A.ld4 : A.g_hidden_part + (int64_t)(-out - 1) * ND * A.ld4;
      float4* arow = out >= 0 ? A.g_as + (int64_t)out * AP4 : A.g_as_part + (int64_t)(-out - 1) * AP4;
      if (row_lane) {
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) hrow[dd * A.ld4 + lane_g] = acc[dd];
      }
      if (lane_g == 0) {
#pragma unroll
        for (int k = 0; k < AP4; ++k) arow[k] = gas[k];
      }
    }
  });

  __syncthreads();
  for (int i = threadIdx.x; i < nr * AP4 * 4; i += BLOCK) {
    const float v = reinterpret_cast<float*>(gar_l)[i];
    if (v != 0.f) atomicAdd(A.g_ar + i, v);
  }
  float vals[AP4 * 4];
#pragma unroll
  for (int k = 0; k < AP4; ++k) { vals[4 * k] = gw[k].x; vals[4 * k + 1] = gw[k].y; vals[4 * k + 2] = gw[k].z; vals[4 * k + 3] = gw[k].w; }
  float* red = reinterpret_cast<float*>(red_l);
#pragma unroll
  for (int i = 0; i < AP4 * 4; ++i) {
    float v = vals[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wv * (AP4 * 4 + 4) + i] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < AP4 * 4 && (int)threadIdx.x < A.attn_dim) {
    float v = 0.f;
    for (int w = 0; w < BLOCK / 64; ++w) v += red[w * (AP4 * 4 + 4) + threadIdx.x];
    if (v != 0.f) atomicAdd(A.g_w + threadIdx.x, v);
  }
}

// hub sources cut into segments: sum the segments' partial rows in order (rows of `cols_h` float4 + ap4 float4)
__global__ void tbwd_combine_kernel(const int4* __restrict__ split, int n_split, int n_slots, int B, const int2* __restrict__ bm_old,
                                    int W, const float4* __restrict__ hpart, const float4* __restrict__ apart,
                                    float4* __restrict__ g_hidden, float4* __restrict__ g_as, int cols_h, int ap4) {
  const int cols = cols_h + ap4;
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
  const bool is_h = c < cols_h;
  const int stride = is_h ? cols_h : ap4;
  const float4* p = (is_h ? hpart : apart) + ((int64_t)b * n_slots + se.y) * stride + (is_h ? c : c - cols_h);
  float4 acc = p[0];
  for (int k = 1; k < se.z; ++k) {
    const float4 v = p[(int64_t)k * stride];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (is_h) g_hidden[(int64_t)s * cols_h + c] = acc;
  else g_as[(int64_t)s * ap4 + (c - cols_h)] = acc;
}

// ---- table gradients, key-major -------------------------------------------------------------------------------------------
// BY_TIME = false: items = (query, segment of relation r's edge list); an edge's direction follows from its time id, so the
//                  segment's alpha*G sums go to up to three rows  dir * n_rela_rows + r  of g_rela_dir.
// BY_TIME = true : items = (query, segment of time id tau's edge list); dt = tau - q_time[b] is the same for the whole segment,
//                  which therefore lands in the single row  dir * n_time + |dt|  of g_time_dir.
struct TKeyArgs {
  rg::WalkArgs walk;          // vrows = CSR-by-relation / CSR-by-time segments; always live
  const int2* ht;             // {head, tail} per entry
  const int32_t* aux;         // BY_TIME ? relation : time id, per entry
  const int32_t* q_time;
  const int2* bm_old;
  const int2* bm_new;
  int W;
  const float4* a_s;
  const float4* a_r;
  const float4* a_q;
  const float* w_alpha;
  const float* b_alpha;
  int attn_dim, n_rela_rows, n_time, ld4;
  const float4* grad_agg;
  float* g_table;             // g_rela_dir or g_time_dir
  // WIN
  const int32_t* win_lo = nullptr;
  const int32_t* win_hi = nullptr;
  const int32_t* row_time = nullptr;
  const int32_t* loop_time = nullptr;
  int n_data = 0;
};

template <int G, int AP4, bool BY_TIME, bool WIN>
__global__ __launch_bounds__(TB_BLOCK, 4) void tkey_kernel(TKeyArgs A) {
  extern __shared__ float4 lds[];
  constexpr int BLOCK = TB_BLOCK;
  const int nr = A.n_rela_rows;
  float4* stage = lds;                // [BLOCK] {o, alpha, dir}
  float4* ar_l = stage + BLOCK;       // [nr][AP4]
  float4* w_l = ar_l + nr * AP4;      // [AP4]
  for (int i = threadIdx.x; i < nr * AP4; i += BLOCK) ar_l[i] = A.a_r[i];
  if (threadIdx.x < AP4) {
    float w[4];
    for (int k = 0; k < 4; ++k) {
      const int j = threadIdx.x * 4 + k;
      w[k] = j < A.attn_dim ? A.w_alpha[j] : 0.f;
    }
    w_l[threadIdx.x] = make_float4(w[0], w[1], w[2], w[3]);
  }
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
    const int beg = R.x, b = R.z, key = R.w;
    int end = R.x + rg::walk_len(R);
    int wlo = 0, whi = 0;
    if constexpr (WIN) {
      wlo = A.win_lo[b]; whi = A.win_hi[b];
      if (BY_TIME && key < A.n_data && (key < wlo || key >= whi)) end = beg;      // the whole row lies outside the query's window
    }
    const int2* old_row = A.bm_old + (int64_t)b * A.W;
    const int2* new_row = A.bm_new + (int64_t)b * A.W;
    const int qt = A.q_time[b];
    float4 aq[AP4];
#pragma unroll
    for (int k = 0; k < AP4; ++k) aq[k] = A.a_q[(int64_t)b * AP4 + k];
    float4 acc[BY_TIME ? 1 : 3];
#pragma unroll
    for (int dd = 0; dd < (BY_TIME ? 1 : 3); ++dd) acc[dd] = f4zero();
    unsigned seen = 0u;      // directions with at least one edge (BY_TIME: bit 0)
    for (int c0 = beg; c0 < end; c0 += G) {
      const int c = c0 + lane_g;
      bool valid = c < end;
      int o = 0, dir = 0;
      float alpha = 0.f;
      if (valid) {
        const int2 ht = A.ht[c];
        const int2 wp = old_row[ht.x >> 5];
        const uint32_t word = (uint32_t)wp.x, bit = ht.x & 31;
        valid = (word >> bit) & 1u;
        if constexpr (WIN && !BY_TIME) {
          if (valid) { const int erow = A.aux[c]; valid = erow >= A.n_data || (erow >= wlo && erow < whi); }
        }
        if (valid) {
          const int s = wp.y + __popc(word & ((1u << bit) - 1u));
          const int2 wn = new_row[ht.y >> 5];
          o = wn.y + __popc((uint32_t)wn.x & ((1u << (ht.y & 31)) - 1u));
          const int other = A.aux[c];
          const int r = BY_TIME ? other : key;
          if constexpr (!BY_TIME && !WIN) { const int dt = other - qt; dir = dt > 0 ? 2 : (dt == 0 ? 1 : 0); }
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
          alpha = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
        }
      }
      const unsigned long long m = (__ballot(valid) >> gshift) & gmask;
      const int cnt = __popcll(m);
      const int pos = __popcll(m & ((1ull << lane_g) - 1ull));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (lane_g >= cnt) my_stage[lane_g] = f4zero();
      if (valid) my_stage[pos] = make_float4(__int_as_float(o), alpha, __int_as_float(dir), 0.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (int k = 0; k < cnt; k += 4) {
        float4 tp[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tp[u] = my_stage[k + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = A.grad_agg[(int64_t)__float_as_int(tp[u].x) * A.ld4 + lane_c];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float al = tp[u].y;
          if constexpr (BY_TIME) {
            acc[0].x = fmaf(al, gv[u].x, acc[0].x); acc[0].y = fmaf(al, gv[u].y, acc[0].y);
            acc[0].z = fmaf(al, gv[u].z, acc[0].z); acc[0].w = fmaf(al, gv[u].w, acc[0].w);
            if (k + u < cnt) seen |= 1u;
          } else {
            const int du = __float_as_int(tp[u].z);
#pragma unroll
            for (int dd = 0; dd < 3; ++dd) {
              const float mk = du == dd ? al : 0.f;
              acc[dd].x = fmaf(mk, gv[u].x, acc[dd].x); acc[dd].y = fmaf(mk, gv[u].y, acc[dd].y);
              acc[dd].z = fmaf(mk, gv[u].z, acc[dd].z); acc[dd].w = fmaf(mk, gv[u].w, acc[dd].w);
            }
            if (k + u < cnt) seen |= 1u << du;
          }
        }
      }
    }
    if (live && row_lane) {
      if constexpr (BY_TIME) {
        if (seen) {
          int trow;
          if constexpr (WIN) {
            trow = min(max(qt - (key >= A.n_data ? A.loop_time[b] : A.row_time[key]), 0), A.n_time - 1);
          } else {
            const int dt = key - qt;
            const int dir = dt > 0 ? 2 : (dt == 0 ? 1 : 0);
            trow = dir * A.n_time + (dt < 0 ? -dt : dt);
          }
          float* gr = A.g_table + ((int64_t)trow * A.ld4 + lane_g) * 4;
          atomicAdd(gr + 0, acc[0].x); atomicAdd(gr + 1, acc[0].y); atomicAdd(gr + 2, acc[0].z); atomicAdd(gr + 3, acc[0].w);
        }
      } else {
#pragma unroll
        for (int dd = 0; dd < (WIN ? 1 : 3); ++dd)
          if (seen & (1u << dd)) {
            float* gr = A.g_table + ((int64_t)(dd * nr + key) * A.ld4 + lane_g) * 4;
            atomicAdd(gr + 0, acc[dd].x); atomicAdd(gr + 1, acc[dd].y); atomicAdd(gr + 2, acc[dd].z); atomicAdd(gr + 3, acc[dd].w);
          }
      }
    }
  });
}

template <int G, int AP4, bool BY_TIME>
int launch_tkey(const TKeyArgs& A, hipStream_t s) {
  const size_t lds = (size_t)(TB_BLOCK + A.n_rela_rows * AP4 + AP4) * sizeof(float4);
  RG_CHECK(lds <= 160 * 1024, "rg_tlayer_bwd: attention table needs %zu B of LDS (> 160 KiB)", lds);
  auto kern = A.win_lo ? tkey_kernel<G, AP4, BY_TIME, true> : tkey_kernel<G, AP4, BY_TIME, false>;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = rg::walk_grid(A.walk.n_items, TB_BLOCK, G, true, lds <= 80 * 1024 ? 2 : 1, 1);
  if (rg::zero_async(A.walk.queues, RG_QUEUE_BYTES, s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TB_BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int G, bool BY_TIME>
int launch_tkey_ap(const TKeyArgs& A, int ap4, hipStream_t s) {
  switch (ap4) {
    case 1: return launch_tkey<G, 1, BY_TIME>(A, s);
    case 2: return launch_tkey<G, 2, BY_TIME>(A, s);
    case 3: return launch_tkey<G, 3, BY_TIME>(A, s);
    case 4: return launch_tkey<G, 4, BY_TIME>(A, s);
    case 8: return launch_tkey<G, 8, BY_TIME>(A, s);
    default: rg::set_error("rg_tlayer_bwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
}

template <bool BY_TIME>
int launch_tkey_g(const TKeyArgs& A, int ld4, int ap4, hipStream_t s) {
  if (ld4 <= 4) return launch_tkey_ap<4, BY_TIME>(A, ap4, s);
  if (ld4 <= 8) return launch_tkey_ap<8, BY_TIME>(A, ap4, s);
  if (ld4 <= 16) return launch_tkey_ap<16, BY_TIME>(A, ap4, s);
  if (ld4 <= 32) return launch_tkey_ap<32, BY_TIME>(A, ap4, s);
  return launch_tkey_ap<64, BY_TIME>(A, ap4, s);
}

template <int G, int AP4, bool DENSE>
int launch2(const TBwdArgs& A, int B, const rg_vrows& vr, const int2* bm_old, hipStream_t s) {
  size_t lds = (size_t)(TB_BLOCK + 2 * A.n_rela_rows * AP4 + AP4 + (TB_BLOCK / 64) * (AP4 + 1)) * sizeof(float4);
  if (!DENSE) lds += (size_t)TB_BLOCK * sizeof(int4);
  RG_CHECK(lds <= 160 * 1024, "rg_tlayer_bwd: attention tables need %zu B of LDS (> 160 KiB)", lds);
  auto kern = A.win_lo ? tlayer_bwd_kernel<G, AP4, DENSE, true> : tlayer_bwd_kernel<G, AP4, DENSE, false>;
  const int nd = A.win_lo ? 1 : 3;
  if (lds > 64 * 1024) RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = rg::walk_grid(A.walk.n_items, TB_BLOCK, G, DENSE, lds <= 80 * 1024 ? 2 : 1, 1);
  if (rg::zero_async(A.walk.queues, RG_QUEUE_BYTES, s)) return 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(TB_BLOCK), lds, s, A);
  RG_LAUNCH_CHECK();
  if (vr.n_split > 0) {
    const int64_t threads = (int64_t)B * vr.n_split * (nd * A.ld4 + AP4);
    hipLaunchKernelGGL(tbwd_combine_kernel, dim3(rg::ceil_div(threads, 256)), dim3(256), 0, s, vr.split, vr.n_split, vr.n_slots, B,
                       bm_old, A.W, A.g_hidden_part, A.g_as_part, A.g_hidden_dir, A.g_as, nd * A.ld4, AP4);
    RG_LAUNCH_CHECK();
  }
  return 0;
}

template <int G>
int launch_ap(const TBwdArgs& A, int ap4, int B, const rg_vrows& vr, const int2* bm_old, bool dense, hipStream_t s) {
#define RG_TB_CASE(K) case K: return dense ? launch2<G, K, true>(A, B, vr, bm_old, s) : launch2<G, K, false>(A, B, vr, bm_old, s);
  switch (ap4) {
    RG_TB_CASE(1) RG_TB_CASE(2) RG_TB_CASE(3) RG_TB_CASE(4) RG_TB_CASE(8)
    default: rg::set_error("rg_tlayer_bwd: padded attention dim %d not in {4,8,12,16,32}", ap4 * 4); return 1;
  }
#undef RG_TB_CASE
}

}  // namespace

extern "C" size_t rg_tlayer_bwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld, int32_t ap) {
  if (!f || !g) return 0;
  return (size_t)f->B * g->out_vr.n_slots * (3 * ld + ap) * sizeof(float) + 512;
}

namespace {
struct WinArgs {            // the extrapolation setting's extras; all null / 0 for the interpolation layer
  const int32_t* loop_time = nullptr;
  const int32_t* row_time = nullptr;
  int n_data = 0, n_tab = 0;
};

int tbwd_impl(const char* who, const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, const int32_t* q_time,
              const float* hidden_dir, const float* rela_dir, const float* time_dir, int32_t d, int32_t ld,
              const float* a_s, const float* a_r, const float* a_q, int32_t ap, const float* w_alpha,
              const float* b_alpha, int32_t attn_dim, const float* grad_agg, float* grad_hidden_dir,
              float* grad_rela_dir, float* grad_time_dir, float* grad_a_s, float* grad_a_r,
              float* grad_a_q, float* grad_w_alpha, void* scratch, size_t scratch_bytes, const WinArgs& win, void* stream) {
  const bool windowed = win.row_time != nullptr;
  const int nd = windowed ? 1 : 3;
  RG_CHECK(f && g && q_time && hidden_dir && rela_dir && time_dir && a_s && a_r && a_q && w_alpha && b_alpha && grad_agg &&
               grad_hidden_dir && grad_rela_dir && grad_time_dir && grad_a_s && grad_a_r && grad_w_alpha,
           "%s: NULL argument", who);
  RG_CHECK(g->out_time && g->rel_tm && g->time_ht && g->n_time > 0, "%s: the graph has no timestamps (build it with rg_tgraph_create)", who);
  RG_CHECK(g->n_ent == f->n_ent, "%s: graph has %d entities, frontier %d", who, g->n_ent, f->n_ent);
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1,
           "%s: level %d not resident (current %d, %d kept)", who, level, f->level, f->n_levels);
  RG_CHECK(n_old == f->n_nodes[(level - 1) % f->n_levels], "%s: n_old=%lld but level %d has %lld nodes", who,
           (long long)n_old, level - 1, (long long)f->n_nodes[(level - 1) % f->n_levels]);
  RG_CHECK(d > 0 && ld >= d && ld % 4 == 0 && ld >= 16 && ld <= 256, "%s: d=%d ld=%d", who, d, ld);
  RG_CHECK(attn_dim > 0 && ap >= attn_dim && ap % 4 == 0, "%s: attn_dim=%d ap=%d", who, attn_dim, ap);
  RG_CHECK((int64_t)f->B * f->n_ent * 3 < ((int64_t)1 << 31), "%s: 3 * batch * n_ent does not fit int32 row ids", who);
  const int n_time = windowed ? win.n_tab : g->n_time;
  RG_CHECK((int64_t)3 * n_time * 4 + 3 < ((int64_t)1 << 31), "%s: n_time too large", who);
  RG_CHECK(!windowed || (f->win_lo && f->win_hi && win.loop_time && win.n_tab > 0 && win.n_data >= 0),
           "%s: call rg_frontier_set_window first (and pass loop_time, n_tab)", who);
  const size_t need = rg_tlayer_bwd_scratch_bytes(f, g, ld, ap);
  RG_CHECK(g->out_vr.n_slots == 0 || (scratch && scratch_bytes >= need), "%s: scratch %zu B < required %zu B", who,
           scratch_bytes, need);
  RG_CHECK((int64_t)f->B * std::max(g->out_vr.n_slots, 1) < ((int64_t)1 << 31) && g->out_vr.n_slots < (1 << 22),
           "%s: batch * hub segments overflows int32", who);
  const int64_t n_items = (int64_t)f->B * g->out_vr.n;
  RG_CHECK(n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "%s: work space too large for 32-bit queue tickets", who);
  if (n_old == 0) return grad_a_q ? rg::launch_aq_sum(f->bm_of(level - 1), f->W, f->B, f->n_ent, 0, grad_a_s, ap, grad_a_q, (hipStream_t)stream) : 0;
  TBwdArgs A;
  A.walk.n_items = n_items; A.walk.n_vrows = g->out_vr.n; A.walk.n_slots = g->out_vr.n_slots; A.walk.vrows = g->out_vr.rows;
  A.walk.bm_test = f->bm_of(level - 1); A.walk.W = f->W; A.walk.queues = f->queues; f->queues_clean = false;
  A.out_rt = g->out_rt; A.out_time = g->out_time; A.q_time = q_time;
  A.bm_new = f->bm_of(level); A.W = f->W;
  A.hidden_dir = (const float4*)hidden_dir; A.rela_dir = (const float4*)rela_dir; A.time_dir = (const float4*)time_dir;
  A.ld4 = ld / 4; A.n_rela_rows = g->n_rela_rows; A.n_time = n_time;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha; A.attn_dim = attn_dim;
  A.grad_agg = (const float4*)grad_agg;
  A.g_hidden_dir = (float4*)grad_hidden_dir; A.g_as = (float4*)grad_a_s;
  A.g_rela_dir = grad_rela_dir; A.g_time_dir = grad_time_dir; A.g_ar = grad_a_r; A.g_w = grad_w_alpha;
  A.g_hidden_part = (float4*)scratch;
  A.g_as_part = (float4*)((char*)scratch + rg::align_up((size_t)f->B * g->out_vr.n_slots * nd * ld * sizeof(float), 256));
  if (windowed) { A.win_lo = f->win_lo; A.win_hi = f->win_hi; A.row_time = win.row_time; A.loop_time = win.loop_time; A.n_data = win.n_data; }
  hipStream_t s = (hipStream_t)stream;
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
  // table gradients, key-major (see tkey_kernel)
  TKeyArgs K;
  K.q_time = q_time; K.bm_old = bm_old; K.bm_new = f->bm_of(level); K.W = f->W;
  K.a_s = (const float4*)a_s; K.a_r = (const float4*)a_r; K.a_q = (const float4*)a_q;
  K.w_alpha = w_alpha; K.b_alpha = b_alpha; K.attn_dim = attn_dim; K.n_rela_rows = g->n_rela_rows; K.n_time = n_time; K.ld4 = ld4;
  K.grad_agg = (const float4*)grad_agg;
  if (windowed) { K.win_lo = f->win_lo; K.win_hi = f->win_hi; K.row_time = win.row_time; K.loop_time = win.loop_time; K.n_data = win.n_data; }
  K.walk.n_slots = 0; K.walk.bm_test = nullptr; K.walk.W = f->W; K.walk.queues = f->queues; f->queues_clean = false;
  K.walk.n_items = (int64_t)f->B * g->rel_vr.n; K.walk.n_vrows = g->rel_vr.n; K.walk.vrows = g->rel_vr.rows;
  RG_CHECK(K.walk.n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "%s: relation work space too large for 32-bit queue tickets", who);
  K.ht = g->rel_ht; K.aux = g->rel_tm; K.g_table = grad_rela_dir;
  if (launch_tkey_g<false>(K, ld4, ap / 4, s)) return 1;
  K.walk.n_items = (int64_t)f->B * g->time_vr.n; K.walk.n_vrows = g->time_vr.n; K.walk.vrows = g->time_vr.rows;
  RG_CHECK(K.walk.n_items / 8 + ((int64_t)1 << 26) < ((int64_t)1 << 31), "%s: time work space too large for 32-bit queue tickets", who);
  K.ht = g->time_ht; K.aux = g->time_rel; K.g_table = grad_time_dir;
  return launch_tkey_g<true>(K, ld4, ap / 4, s);
}
}  // namespace

extern "C" int rg_tlayer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, const int32_t* q_time,
                             const float* hidden_dir, const float* rela_dir, const float* time_dir, int32_t d, int32_t ld,
                             const float* a_s, const float* a_r, const float* a_q, int32_t ap, const float* w_alpha,
                             const float* b_alpha, int32_t attn_dim, const float* grad_agg, float* grad_hidden_dir,
                             float* grad_rela_dir, float* grad_time_dir, float* grad_a_s, float* grad_a_r,
                             float* grad_a_q, float* grad_w_alpha, void* scratch, size_t scratch_bytes, void* stream) {
  return tbwd_impl("rg_tlayer_bwd", f, g, level, n_old, q_time, hidden_dir, rela_dir, time_dir, d, ld, a_s, a_r, a_q, ap, w_alpha, b_alpha,
                   attn_dim, grad_agg, grad_hidden_dir, grad_rela_dir, grad_time_dir, grad_a_s, grad_a_r, grad_a_q, grad_w_alpha, scratch,
                   scratch_bytes, WinArgs(), stream);
}

// Adjoint of rg_xlayer_fwd (temporal extrapolation; see the header comment): grad_hidden_p [N_old, ld], grad_rela_p [n_rela_rows, ld],
// grad_time_p [n_tab, ld] (the last two zero-initialised by the caller, added into), the attention gradients as rg_tlayer_bwd.
extern "C" int rg_xlayer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, const int32_t* q_time,
                             const int32_t* loop_time, const int32_t* row_time, int32_t n_data, const float* hidden_p, const float* rela_p,
                             const float* time_p, int32_t n_tab, int32_t d, int32_t ld, const float* a_s, const float* a_r, const float* a_q,
                             int32_t ap, const float* w_alpha, const float* b_alpha, int32_t attn_dim, const float* grad_agg,
                             float* grad_hidden_p, float* grad_rela_p, float* grad_time_p, float* grad_a_s, float* grad_a_r, float* grad_a_q,
                             float* grad_w_alpha, void* scratch, size_t scratch_bytes, void* stream) {
  RG_CHECK(loop_time && row_time, "rg_xlayer_bwd: NULL argument");
  WinArgs w;
  w.loop_time = loop_time; w.row_time = row_time; w.n_data = n_data; w.n_tab = n_tab;
  return tbwd_impl("rg_xlayer_bwd", f, g, level, n_old, q_time, hidden_p, rela_p, time_p, d, ld, a_s, a_r, a_q, ap, w_alpha, b_alpha, attn_dim,
                   grad_agg, grad_hidden_p, grad_rela_p, grad_time_p, grad_a_s, grad_a_r, grad_a_q, grad_w_alpha, scratch, scratch_bytes, w, stream);
}
