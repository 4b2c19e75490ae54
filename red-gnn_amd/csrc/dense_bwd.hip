// Adjoint of the fused dense training step (rg_dense_train_fwd; models.py:41,81-83) for the node-row quantities:
// given d hidden_new it produces, per node,
//   the GRU gate gradients   dgi = (dr, dz, dn),  dgh = (dr, dz, dn * r)      (pre-activation gradients, [n, 3d] each)
//   d h0  = g * z + dgh W_hh,   dx = dgi W_ih,   dpre = dx * mask * act'(.),   dagg = dpre W_h
// in one f32-MFMA kernel.  It mirrors dense.hip: every product is computed transposed (out^T[col][node] = sum_k W^T[col][k]
// G[node][k]) so that the 16 nodes of a tile sit on the lanes and an accumulator tile is directly the next product's B
// fragment; here even the inputs arrive in that layout (lane = node, 4 consecutive columns per lane quarter: 64-byte row
// segments), so there is no LDS staging at all - LDS holds the three TRANSPOSED weight images for the whole kernel.
// The weight gradients (sums over all nodes of outer products) stay with the caller (row-chunked batched GEMMs on dgi, dgh,
// dpre).  Replaces aten's fused GRU-cell backward, three [n, .] GEMMs and the elementwise passes between them.
#include "dense_common.h"

namespace {

using namespace rg;

constexpr int DB_T = 512;

struct DenseBwdArgs {
  int64_t n;
  int d;
  const float* g_h;      // [n][d]
  const float* ws;       // [n][5][d] = {r, z, n, h0, hn_pre}
  const float* x;        // [n][d]  GRU input = act(pre) * mask
  const float* mask;     // [n][d] or null
  float keep;            // 1 - p (tanh: y = x * keep where kept)
  int act;
  const float* W_h;      // [d][d]
  const float* w_ih;     // [3d][d]
  const float* w_hh;
  float* dgi;            // [n][3d]
  float* dgh;
  float* dpre;           // [n][d]
  float* dagg;
  float* dh0;
  float* dgh_n = nullptr;          // [n][d]: only the n-gate part of dgh (its r and z parts equal dgi's); then dgh is not written
  const int32_t* prev_idx = nullptr;   // [n]: the node's row in the previous frontier or -1; then dh0 rows are scattered to dh0[prev_idx]
  int n_tiles;
};

// transposed images: row c (an input column of the weight), 16-B slot s (4 consecutive k), XOR-swizzled by the row inside
// groups of XM + 1 slots (XM + 1 = DP / 4 divides both row lengths)
template <int SLOTS, int XM>
__device__ __forceinline__ int swt(int row, int slot) { return row * SLOTS + (slot ^ (row & XM)); }

template <int NB>
__global__ __launch_bounds__(DB_T, 2) void dense_bwd_kernel(DenseBwdArgs A) {
  constexpr int DP = 16 * NB;
  constexpr int S3 = 3 * DP / 4, S1 = DP / 4;      // slots per row of the gate / W_h images
  constexpr int KS = DP / 4;
  constexpr int XM = DP / 4 - 1;
  constexpr int NW = DB_T / 64;
  extern __shared__ float4 lds[];
  float4* WihT = lds;                   // [DP rows c][S3]:  WihT[c][g*DP + k] = w_ih[g*d + k][c]
  float4* WhhT = WihT + DP * S3;
  float4* WhT = WhhT + DP * S3;         // [DP rows c][S1]:  WhT[c][k] = W_h[k][c]
  const int d = A.d;
  auto load_t = [&](float4* dst, const float* src, int gates, int slots) {
    for (int i = threadIdx.x; i < DP * slots; i += DB_T) {
      const int c = i / slots, sl = i - c * slots;
      float v[4];
      for (int j = 0; j < 4; ++j) {
        const int kk = sl * 4 + j;                 // g*DP + k
        const int g = kk / DP, k = kk - g * DP;
        v[j] = (c < d && k < d && g < gates) ? src[((int64_t)g * d + k) * d + c] : 0.f;
      }
      dst[c * slots + (sl ^ (c & XM))] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  load_t(WihT, A.w_ih, 3, S3);
  load_t(WhhT, A.w_hh, 3, S3);
  load_t(WhT, A.W_h, 1, S1);
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, hq = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto ld4 = [&](const float* base, int64_t row, int row_floats, int col) -> float4 {
    return *reinterpret_cast<const float4*>(base + row * row_floats + col);
  };
  auto mfma4 = [&](const float4& a, const float (&b)[4], f32x4 acc) -> f32x4 {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[3], acc, 0, 0, 0);
    return acc;
  };

  for (int t = blockIdx.x * NW + wv; t < A.n_tiles; t += gridDim.x * NW) {
    const int64_t node = (int64_t)t * 16 + li;
    const bool in_n = node < A.n;
    const int p_row = (A.prev_idx && in_n) ? A.prev_idx[node] : -1;
    f32x4 acc_dx[NB], acc_dh[NB];
#pragma unroll
    for (int o = 0; o < NB; ++o) { acc_dx[o] = zero4; acc_dh[o] = zero4; }
    float gz[KS];                                   // g * z, the direct part of d h0
    // the six row segments of block ob + 1 are requested before block ob's MFMAs are issued (explicit double buffer: left to
    // itself the compiler requests all four blocks at once and spills, fenced off it requests each block just in time)
    auto fetch = [&](int ob, float4 (&v)[6]) {
      const int col = 16 * ob + 4 * hq;
#pragma unroll
      for (int q = 0; q < 6; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in_n && col < d) {
        v[0] = ld4(A.g_h, node, d, col);
        v[1] = ld4(A.ws, node, 5 * d, col); v[2] = ld4(A.ws, node, 5 * d, d + col); v[3] = ld4(A.ws, node, 5 * d, 2 * d + col);
        v[4] = ld4(A.ws, node, 5 * d, 3 * d + col); v[5] = ld4(A.ws, node, 5 * d, 4 * d + col);
      }
    };
    float4 cur[6], nxt[6];
    fetch(0, cur);
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      const int col = 16 * ob + 4 * hq;
      const bool ok = in_n && col < d;
      if (ob + 1 < NB) fetch(ob + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      const float4 g = cur[0], r = cur[1], z = cur[2], nn = cur[3], h0 = cur[4], hp = cur[5];
      const float gv[4] = {g.x, g.y, g.z, g.w}, rv[4] = {r.x, r.y, r.z, r.w}, zv[4] = {z.x, z.y, z.z, z.w};
      const float nv[4] = {nn.x, nn.y, nn.z, nn.w}, hv[4] = {h0.x, h0.y, h0.z, h0.w}, pv[4] = {hp.x, hp.y, hp.z, hp.w};
      float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dng = gv[j] * (1.0f - zv[j]) * (1.0f - nv[j] * nv[j]);      // d n_pre
        dn[j] = dng;
        dr[j] = dng * pv[j] * rv[j] * (1.0f - rv[j]);                              // d r_pre
        dz[j] = gv[j] * (hv[j] - nv[j]) * zv[j] * (1.0f - zv[j]);                  // d z_pre
        dnr[j] = dng * rv[j];
        gz[4 * ob + j] = gv[j] * zv[j];
      }
      if (ok) {
        float* gi = A.dgi + node * (3 * (int64_t)d) + col;
        float* gh = A.dgh + node * (3 * (int64_t)d) + col;
        *reinterpret_cast<float4*>(gi) = make_float4(dr[0], dr[1], dr[2], dr[3]);
        *reinterpret_cast<float4*>(gi + d) = make_float4(dz[0], dz[1], dz[2], dz[3]);
        *reinterpret_cast<float4*>(gi + 2 * d) = make_float4(dn[0], dn[1], dn[2], dn[3]);
        if (A.dgh_n) {
          *reinterpret_cast<float4*>(A.dgh_n + node * (int64_t)d + col) = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
        } else {
          *reinterpret_cast<float4*>(gh) = make_float4(dr[0], dr[1], dr[2], dr[3]);
          *reinterpret_cast<float4*>(gh + d) = make_float4(dz[0], dz[1], dz[2], dz[3]);
          *reinterpret_cast<float4*>(gh + 2 * d) = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
        }
      }
      // dx^T += W_ih^T[:, (g, ob)] dgi_g ;  dh^T += W_hh^T[:, (g, ob)] dgh_g      (k-block = this lane quarter's 4 columns)
#pragma unroll
      for (int o = 0; o < NB; ++o) {
        const int row = 16 * o + li;
        const int s0 = (0 * DP + 16 * ob) / 4 + hq, s1 = (1 * DP + 16 * ob) / 4 + hq, s2 = (2 * DP + 16 * ob) / 4 + hq;
        const float4 a0 = WihT[swt<S3, XM>(row, s0)], a1 = WihT[swt<S3, XM>(row, s1)], a2 = WihT[swt<S3, XM>(row, s2)];
        const float4 b0 = WhhT[swt<S3, XM>(row, s0)], b1 = WhhT[swt<S3, XM>(row, s1)], b2 = WhhT[swt<S3, XM>(row, s2)];
        acc_dx[o] = mfma4(a0, dr, acc_dx[o]);
        acc_dh[o] = mfma4(b0, dr, acc_dh[o]);
        acc_dx[o] = mfma4(a1, dz, acc_dx[o]);
        acc_dh[o] = mfma4(b1, dz, acc_dh[o]);
        acc_dx[o] = mfma4(a2, dn, acc_dx[o]);
        acc_dh[o] = mfma4(b2, dnr, acc_dh[o]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 6; ++q) cur[q] = nxt[q];
    }

    // ---- d h0, dpre (dropout mask, activation derivative) --------------------------------------------------------------
    float dp[KS];
#pragma unroll
    for (int o = 0; o < NB; ++o) {
      const int col = 16 * o + 4 * hq;
      const bool ok = in_n && col < d;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
      if (ok) {
        xv = ld4(A.x, node, d, col);
        if (A.mask) mk = ld4(A.mask, node, d, col);
      }
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ms[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = acc_dx[o][j] * (A.mask ? ms[j] : 1.0f);
        if (A.act == 1) v = xs[j] > 0.f ? v : 0.f;
        else if (A.act == 2) { const float y = xs[j] * (A.mask ? A.keep : 1.0f); v *= 1.0f - y * y; }
        dp[4 * o + j] = v;
      }
      if (ok) {
        *reinterpret_cast<float4*>(A.dpre + node * d + col) = make_float4(dp[4 * o + 0], dp[4 * o + 1], dp[4 * o + 2], dp[4 * o + 3]);
        // (every node of the previous frontier is exactly one node of this one: with prev_idx the carried state's gradient lands in
        // its own row of the previous level directly - no [n, d] intermediate and no gather pass)
        const int64_t hrow = A.prev_idx ? (int64_t)p_row : node;
        if (hrow >= 0)
          *reinterpret_cast<float4*>(A.dh0 + hrow * d + col) =
              make_float4(acc_dh[o][0] + gz[4 * o + 0], acc_dh[o][1] + gz[4 * o + 1], acc_dh[o][2] + gz[4 * o + 2], acc_dh[o][3] + gz[4 * o + 3]);
      }
    }

    // ---- dagg^T = W_h^T dpre ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int o = 0; o < NB; ++o) {
      f32x4 acc = zero4;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        const float4 a = WhT[swt<S1, XM>(16 * o + li, 4 * kb + hq)];
        const float b[4] = {dp[4 * kb + 0], dp[4 * kb + 1], dp[4 * kb + 2], dp[4 * kb + 3]};
        acc = mfma4(a, b, acc);
      }
      const int col = 16 * o + 4 * hq;
      if (in_n && col < d) *reinterpret_cast<float4*>(A.dagg + node * d + col) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

template <int NB>
int launch_bwd(const DenseBwdArgs& A, hipStream_t s) {
  constexpr int DP = 16 * NB;
  const size_t lds = (size_t)(2 * DP * (3 * DP / 4) + DP * (DP / 4)) * sizeof(float4);
  RG_HIP(hipFuncSetAttribute((const void*)dense_bwd_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (int)std::min<int64_t>(rg::ceil_div(A.n_tiles, DB_T / 64), 256);
  hipLaunchKernelGGL((dense_bwd_kernel<NB>), dim3(grid), dim3(DB_T), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int rg_dense_train_bwd(int64_t n, int32_t d, const float* grad_hidden, const float* gates_ws, const float* x,
                                  const float* mask, float keep, int32_t act, const float* W_h, const float* w_ih,
                                  const float* w_hh, float* grad_gates_i, float* grad_gates_h, float* grad_pre, float* grad_agg,
                                  float* grad_h0, void* stream) {
  RG_CHECK(grad_hidden && gates_ws && x && W_h && w_ih && w_hh && grad_gates_i && grad_gates_h && grad_pre && grad_agg && grad_h0,
           "rg_dense_train_bwd: NULL argument");
  RG_CHECK(d >= 16 && d <= 64 && d % 4 == 0, "rg_dense_train_bwd: hidden_dim %d not supported (16..64, multiple of 4)", d);
  RG_CHECK(act >= 0 && act <= 2, "rg_dense_train_bwd: act=%d", act);
  RG_CHECK((((uintptr_t)grad_hidden | (uintptr_t)gates_ws | (uintptr_t)x | (uintptr_t)mask | (uintptr_t)grad_gates_i |
             (uintptr_t)grad_gates_h | (uintptr_t)grad_pre | (uintptr_t)grad_agg | (uintptr_t)grad_h0) & 15) == 0,
           "rg_dense_train_bwd: float buffers must be 16-B aligned");
  if (n == 0) return 0;
  DenseBwdArgs A;
  A.n = n; A.d = d; A.g_h = grad_hidden; A.ws = gates_ws; A.x = x; A.mask = mask; A.keep = keep; A.act = act;
  A.W_h = W_h; A.w_ih = w_ih; A.w_hh = w_hh;
  A.dgi = grad_gates_i; A.dgh = grad_gates_h; A.dpre = grad_pre; A.dagg = grad_agg; A.dh0 = grad_h0;
  A.n_tiles = (int)rg::ceil_div(n, 16);
  hipStream_t s = (hipStream_t)stream;
  return d <= 32 ? launch_bwd<2>(A, s) : launch_bwd<4>(A, s);
}

// rg_dense_train_bwd with two fewer passes over memory: grad_gates_hn [n, d] is only the n-gate block of the hidden-side gate gradients
// (their r and z blocks equal grad_gates_i's, which the caller reads instead), and with prev_idx [n] (a node's row in the previous
// frontier or -1) the carried state's gradient is written straight to grad_prev [n_old, d] (every previous node is exactly one node
// of this level, so every row of grad_prev is written exactly once); prev_idx NULL: grad_prev is [n, d] = grad_h0 as before.
extern "C" int rg_dense_train_bwd2(int64_t n, int32_t d, const float* grad_hidden, const float* gates_ws, const float* x,
                                   const float* mask, float keep, int32_t act, const float* W_h, const float* w_ih,
                                   const float* w_hh, const int32_t* prev_idx, float* grad_gates_i, float* grad_gates_hn, float* grad_pre,
                                   float* grad_agg, float* grad_prev, void* stream) {
  RG_CHECK(grad_hidden && gates_ws && x && W_h && w_ih && w_hh && grad_gates_i && grad_gates_hn && grad_pre && grad_agg && grad_prev,
           "rg_dense_train_bwd2: NULL argument");
  RG_CHECK(d >= 16 && d <= 64 && d % 4 == 0, "rg_dense_train_bwd2: hidden_dim %d not supported (16..64, multiple of 4)", d);
  RG_CHECK(act >= 0 && act <= 2, "rg_dense_train_bwd2: act=%d", act);
  RG_CHECK((((uintptr_t)grad_hidden | (uintptr_t)gates_ws | (uintptr_t)x | (uintptr_t)mask | (uintptr_t)grad_gates_i |
             (uintptr_t)grad_gates_hn | (uintptr_t)grad_pre | (uintptr_t)grad_agg | (uintptr_t)grad_prev) & 15) == 0,
           "rg_dense_train_bwd2: float buffers must be 16-B aligned");
  if (n == 0) return 0;
  DenseBwdArgs A;
  A.n = n; A.d = d; A.g_h = grad_hidden; A.ws = gates_ws; A.x = x; A.mask = mask; A.keep = keep; A.act = act;
  A.W_h = W_h; A.w_ih = w_ih; A.w_hh = w_hh;
  A.dgi = grad_gates_i; A.dgh = nullptr; A.dgh_n = grad_gates_hn; A.dpre = grad_pre; A.dagg = grad_agg; A.dh0 = grad_prev;
  A.prev_idx = prev_idx;
  A.n_tiles = (int)rg::ceil_div(n, 16);
  hipStream_t s = (hipStream_t)stream;
  return d <= 32 ? launch_bwd<2>(A, s) : launch_bwd<4>(A, s);
}
