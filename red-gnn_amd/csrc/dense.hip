// Fused dense epilogue of one layer on f32 MFMA (inference path).
// Replaces, per layer of RED_GNN_trans.forward (Static/transductive/models.py):
//   :41     hidden_new = act(W_h(message_agg))
//   :81     h0 = zeros(...).index_copy_(1, old_nodes_new_idx, h0)         (as a gather by prev_idx)
//   :82-84  single-step nn.GRU  (dropout is the identity in eval mode)
//   next layer's :36 Ws_attn(hs), hoisted to per-node  a_s = hidden Ws^T
//   :86-88  W_final(hidden) scattered into scores_all                    (last layer)
// i.e. three rocBLAS GEMMs + gru_cell + index_copy + fills + two N x 3d temporaries become one kernel
// that reads agg and the gathered old state once and writes hidden, a_s (and scores) once.
//
// Every product is computed TRANSPOSED with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains):
//   out^T[w_row][node] = sum_k W[w_row][k] * X[node][k]      A = W (LDS),  B = X rows (one node per lane)
// so the 16 nodes of a wave's tile sit on the lanes (node = lane & 15) for inputs AND outputs, the
// output's w_row index lives in the 4 accumulator registers of a 16-row block, and an accumulator tile is
// directly the B operand of the next product (no LDS round trip between W_h, the GRU gates and the
// projections): lane quarter hq = lane >> 4 owns the k-subset  k(hq, m) = 16*(m/4) + 4*hq + (m%4),
// which is exactly the set of accumulator rows the lane holds.  Weights stay in LDS for the whole
// kernel (XOR-swizzled 16-B slots, conflict-free ds_read_b128); one persistent 8-wave workgroup per CU
// (2 waves per SIMD: one wave's loads and transcendentals hide under the other's MFMAs).
#include "dense_common.h"

using namespace rg;

namespace {

// LDS images: rows of DP floats, 16-B slot index XOR-swizzled with the row so that 16 lanes reading the same
// logical slot of 16 different rows hit 16 different bank groups.
template <int DP>
__device__ __forceinline__ int sw(int row, int slot) { return row * (DP / 4) + (slot ^ (row & (DP / 4 - 1))); }

// 8 waves per workgroup (2 per SIMD, <= 256 VGPRs).  NB = DP / 16 in {2, 4}: d <= 64, where the 7 weight images fit LDS
// (d = 128 streams them: dense128.hip).
constexpr int DENSE_T = 512;

// TRAIN: the same kernel also applies the dropout mask to the GRU input and writes what the backward pass needs - the GRU input
// and the gate workspace in the layout of aten's fused GRU cell, so that its fused backward kernel can be reused - straight
// from the accumulator layout (lane = node, 4 consecutive columns per register quad: 64-byte row segments).
template <int NB, bool TRAIN>
__global__ __launch_bounds__(DENSE_T, 2) void dense_kernel(DenseArgs A) {
  constexpr int DP = 16 * NB;      // padded width
  constexpr int S = DP / 4;        // 16-B slots per row
  constexpr int KS = DP / 4;       // MFMA k-steps per product = B-fragment registers
  constexpr int NW = DENSE_T / 64;
  extern __shared__ float4 lds[];
  if (A.n_dev) { A.n = *A.n_dev; A.n_tiles = (int)((A.n + 15) / 16); }      // the grid was sized for the capacity
  float4* Wh_l = lds;                         // [DP rows][S]
  float4* Wih_l = Wh_l + DP * S;              // [3*DP][S]   gate g rows at g*DP
  float4* Whh_l = Wih_l + 3 * DP * S;         // [3*DP][S]
  float4* E_l = Whh_l + 3 * DP * S;           // [32][S]     rows 0..ap-1 = Ws, row 16 = W_final
  float* bias_l = reinterpret_cast<float*>(E_l + 32 * S);   // [4][DP]: b_ir+b_hr, b_iz+b_hz, b_in, b_hn
  float4* tiles = reinterpret_cast<float4*>(bias_l + 4 * DP);   // [NW waves][16 rows][S]

  const int d = A.d;
  // ---- weights -> LDS (zero padded, swizzled) ----------------------------------------------------------
  // (one 16-B load per slot when the rows are 16-B aligned: the staging of the 7 images is the fixed cost of a launch, 10 of the 30 us
  // a 50-query batch's launch takes)
  const bool vec4 = (d & 3) == 0;
  auto load_w = [&](float4* dst, const float* src, int rows_src, int row0_dst, int rows_dst) {
    const bool v4 = vec4 && ((uintptr_t)src & 15) == 0;
    for (int i = threadIdx.x; i < rows_dst * S; i += DENSE_T) {
      const int r = i / S, sl = i - r * S;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src && r < rows_src) {
        if (v4) {
          if (sl * 4 < d) o = *reinterpret_cast<const float4*>(src + (int64_t)r * d + sl * 4);
        } else {
          float v[4];
          for (int k = 0; k < 4; ++k) {
            const int c = sl * 4 + k;
            v[k] = c < d ? src[(int64_t)r * d + c] : 0.f;
          }
          o = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
      dst[sw<DP>(row0_dst + r, sl)] = o;
    }
  };
  load_w(Wh_l, A.W_h, d, 0, DP);
  for (int g = 0; g < 3; ++g) {
    load_w(Wih_l, A.w_ih + (int64_t)g * d * d, d, g * DP, DP);
    load_w(Whh_l, A.w_hh + (int64_t)g * d * d, d, g * DP, DP);
  }
  load_w(E_l, A.Ws, A.Ws ? A.attn : 0, 0, 16);
  load_w(E_l, A.W_final, A.W_final ? 1 : 0, 16, 16);
  for (int i = threadIdx.x; i < 4 * DP; i += DENSE_T) {
    const int g = i / DP, c = i - g * DP;
    float v = 0.f;
    if (c < d) {
      if (g == 0) v = A.b_ih[c] + A.b_hh[c];
      else if (g == 1) v = A.b_ih[d + c] + A.b_hh[d + c];
      else if (g == 2) v = A.b_ih[2 * d + c];
      else v = A.b_hh[2 * d + c];
    }
    bias_l[i] = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, hq = lane >> 4;
  float4* tile = tiles + wv * 16 * S;

  // B fragment of the wave's staged tile: registers 4*ob..4*ob+3 <- X[node li][16*ob + 4*hq + (0..3)]
  auto read_frag = [&](float (&f)[KS]) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      const float4 v = tile[sw<DP>(li, 4 * ob + hq)];
      f[4 * ob + 0] = v.x; f[4 * ob + 1] = v.y; f[4 * ob + 2] = v.z; f[4 * ob + 3] = v.w;
    }
  };
  // A fragment: 4 consecutive k of weight row `row`
  auto lda = [&](const float4* W_l, int row, int slot) -> float4 { return W_l[sw<DP>(row, slot)]; };
  // acc += W[row0 + (0..15)][:] . frag     (one 16-row block of a weight image)
  auto mma = [&](const float4* W_l, int row0, const float (&f)[KS], f32x4 acc) -> f32x4 {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const float4 a = W_l[sw<DP>(row0 + li, 4 * kb + hq)];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f[4 * kb + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f[4 * kb + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f[4 * kb + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f[4 * kb + 3], acc, 0, 0, 0);
    }
    return acc;
  };
  // three independent 16-row blocks against the same fragment, MFMAs interleaved so that no instruction
  // waits on its predecessor's accumulator (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency)
  auto mma3 = [&](const float4* Wl, int r0, int r1, int r2, const float (&f)[KS], f32x4& c0, f32x4& c1, f32x4& c2) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const float4 a0 = lda(Wl, r0 + li, 4 * kb + hq);
      const float4 a1 = lda(Wl, r1 + li, 4 * kb + hq);
      const float4 a2 = lda(Wl, r2 + li, 4 * kb + hq);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, f[4 * kb + 0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, f[4 * kb + 0], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, f[4 * kb + 0], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, f[4 * kb + 1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, f[4 * kb + 1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, f[4 * kb + 1], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, f[4 * kb + 2], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, f[4 * kb + 2], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, f[4 * kb + 2], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, f[4 * kb + 3], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, f[4 * kb + 3], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, f[4 * kb + 3], c2, 0, 0, 0);
    }
  };
  // accumulator row of register r of block ob in this lane: 16*ob + 4*hq + r
  auto bias_acc = [&](int g, int ob) -> f32x4 {
    const float4 v = *reinterpret_cast<const float4*>(bias_l + g * DP + 16 * ob + 4 * hq);
    f32x4 acc = {v.x, v.y, v.z, v.w};
    return acc;
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // tile loads (agg rows: coalesced; old state: rows gathered by prev_idx), software-pipelined one tile ahead
  constexpr int NL = 16 * S / 64;
  auto load_tile = [&](int t, float4 (&va)[NL], float4 (&vh)[NL], bool& has_old) {
    const int64_t row0 = (int64_t)t * 16;
    has_old = false;
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 64 + lane, r = e / S, sl = e - r * S;
      va[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      vh[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < A.n_tiles && row0 + r < A.n && sl < A.ld4) {
        va[it] = A.agg[(row0 + r) * A.ld4 + sl];
        if (A.prev_idx) {
          const int p = A.prev_idx[row0 + r];
          if (p >= 0) { vh[it] = A.hprev[(int64_t)p * A.ld4 + sl]; has_old = true; }
        }
      }
    }
  };

  // The two waves of a SIMD run the same program; started together they stay in lockstep (both in the MFMA phase,
  // then both in the transcendental/store phase, the matrix pipe idle).  Delaying the second half of the workgroup
  // by about half an MFMA phase lets one wave's epilogue run under the other's MFMAs (MI355X_MICROARCH.md, item 9).
  if (wv >= NW / 2) __builtin_amdgcn_s_sleep(127);
  float4 va[NL], vh[NL];
  bool has_old = false;
  const int t_step = gridDim.x * NW;
  int t = blockIdx.x * NW + wv;
  if (t < A.n_tiles) load_tile(t, va, vh, has_old);
  for (; t < A.n_tiles; t += t_step) {
    const int64_t row0 = (int64_t)t * 16;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 64 + lane, r = e / S, sl = e - r * S;
      tile[sw<DP>(r, sl)] = va[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float fx[KS];
    read_frag(fx);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 64 + lane, r = e / S, sl = e - r * S;
      tile[sw<DP>(r, sl)] = vh[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float hf[KS];
    read_frag(hf);
    // a tile of new nodes only (every node of the early hops) has h = 0: its three W_hh products vanish
    const bool any_old = __ballot(has_old) != 0ull;
    // next tile's loads fly under this tile's MFMAs
    load_tile(t + t_step, va, vh, has_old);

    // ---- stage 1: x = act(W_h agg)   (accumulators become the next B fragment) ---------------------------
    float xf[KS];
    {
      f32x4 acc[NB];
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) acc[ob] = zero4;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        float4 a[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) a[ob] = lda(Wh_l, 16 * ob + li, 4 * kb + hq);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].x, fx[4 * kb + 0], acc[ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].y, fx[4 * kb + 1], acc[ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].z, fx[4 * kb + 2], acc[ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob].w, fx[4 * kb + 3], acc[ob], 0, 0, 0);
      }
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[ob][r];
          if (A.act == 1) v = fmaxf(v, 0.f);
          else if (A.act == 2) v = fast_tanh(v);
          xf[4 * ob + r] = v;
        }
        if constexpr (TRAIN) {
          const int64_t node = row0 + li;
          const int col = 16 * ob + 4 * hq;
          if (node < A.n && col < d) {
            if (A.mask) {
              const float4 mk = *reinterpret_cast<const float4*>(A.mask + node * (A.ld4 * 4) + col);
              xf[4 * ob + 0] *= mk.x; xf[4 * ob + 1] *= mk.y; xf[4 * ob + 2] *= mk.z; xf[4 * ob + 3] *= mk.w;
            }
            *reinterpret_cast<float4*>(A.x_out + node * (A.ld4 * 4) + col) =
                make_float4(xf[4 * ob + 0], xf[4 * ob + 1], xf[4 * ob + 2], xf[4 * ob + 3]);
          }
        }
      }
    }

    // ---- GRU gates ([r; z; n] row blocks of weight_ih / weight_hh) -----------------------------------------
    float hn[KS];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 ar = bias_acc(0, ob), az = bias_acc(1, ob), ai = bias_acc(2, ob), ah = bias_acc(3, ob);
      mma3(Wih_l, 0 * DP + 16 * ob, 1 * DP + 16 * ob, 2 * DP + 16 * ob, xf, ar, az, ai);
      if (any_old) mma3(Whh_l, 0 * DP + 16 * ob, 1 * DP + 16 * ob, 2 * DP + 16 * ob, hf, ar, az, ah);
      float rgv[4], zgv[4], ngv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = fast_sigmoid(ar[r]), zg = fast_sigmoid(az[r]);
        const float ng = fast_tanh(ai[r] + rg * ah[r]);
        hn[4 * ob + r] = (1.0f - zg) * ng + zg * hf[4 * ob + r];
        rgv[r] = rg; zgv[r] = zg; ngv[r] = ng;
      }
      if constexpr (TRAIN) {
        const int64_t nd = row0 + li;
        const int col = 16 * ob + 4 * hq;
        if (nd < A.n && col < d) {
          float* w = A.ws_out + nd * (5 * (int64_t)d) + col;
          *reinterpret_cast<float4*>(w) = make_float4(rgv[0], rgv[1], rgv[2], rgv[3]);
          *reinterpret_cast<float4*>(w + d) = make_float4(zgv[0], zgv[1], zgv[2], zgv[3]);
          *reinterpret_cast<float4*>(w + 2 * d) = make_float4(ngv[0], ngv[1], ngv[2], ngv[3]);
          *reinterpret_cast<float4*>(w + 3 * d) = make_float4(hf[4 * ob + 0], hf[4 * ob + 1], hf[4 * ob + 2], hf[4 * ob + 3]);
          *reinterpret_cast<float4*>(w + 4 * d) = make_float4(ah[0], ah[1], ah[2], ah[3]);
        }
      }
    }

    // ---- projections of the new state: a_s (next layer) or score (last layer) ----------------------------------
    const int64_t node = row0 + li;
    if (A.Ws) {       // rows e = 4*hq + r of block 0
      const f32x4 ae = mma(E_l, 0, hn, zero4);
      if (node < A.n && 4 * hq < A.ap)
        reinterpret_cast<float4*>(A.a_s_out + node * A.ap)[hq] = make_float4(ae[0], ae[1], ae[2], ae[3]);
    }
    if (A.W_final) {  // row 16 = register 0 of quarter 0 of block 1
      const f32x4 ae = mma(E_l, 16, hn, zero4);
      if (node < A.n && hq == 0) {
        const int b = A.nodes[2 * node], e = A.nodes[2 * node + 1];
        A.scores[(int64_t)b * A.n_ent + e] = ae[0];
      }
    }

    // ---- new state: transpose through the tile, store coalesced rows ----------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
      tile[sw<DP>(li, 4 * ob + hq)] = make_float4(hn[4 * ob + 0], hn[4 * ob + 1], hn[4 * ob + 2], hn[4 * ob + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 16 * S / 64; ++it) {
      const int e = it * 64 + lane, r = e / S, sl = e - r * S;
      if (row0 + r < A.n && sl < A.ld4) A.hidden_out[(row0 + r) * A.ld4 + sl] = tile[sw<DP>(r, sl)];
    }
  }
}

template <int NB, bool TRAIN>
int launch(const DenseArgs& A, hipStream_t s) {
  constexpr int DP = 16 * NB, S = DP / 4, NW = DENSE_T / 64;
  const size_t lds = (size_t)(7 * DP * S + 32 * S) * sizeof(float4) + 4 * DP * sizeof(float) + (size_t)NW * 16 * S * sizeof(float4);
  RG_HIP(hipFuncSetAttribute((const void*)dense_kernel<NB, TRAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // one persistent workgroup per CU at most; with a device-side row count the grid follows the caller's expectation (+25 %) instead of
  // the capacity: a 50-query batch has a few hundred tiles, and 256 workgroups staging 123 KB of weights each kept every CU busy for
  // 30 us per launch while other streams' batches waited
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, rg::ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL((dense_kernel<NB, TRAIN>), dim3(grid), dim3(DENSE_T), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int64_t rg_dense_scratch_bytes(int32_t d, int32_t precision) {
  if (d != 128) return 0;
  return precision == 1 ? rg::dense128_split_scratch_bytes() : precision == 2 ? rg::dense128_split3_scratch_bytes() : 0;
}

extern "C" int rg_dense_fwd_supported(int32_t d, int32_t attn_dim) { return ((d >= 1 && d <= 64) || d == 128) && attn_dim <= 16; }

static int dense_fwd_impl(int64_t n, const int32_t* n_dev, int64_t n_hint, int32_t d, int32_t ld, const float* agg, const float* hidden_prev,
                          const int32_t* prev_idx, const float* W_h, int32_t act, const float* w_ih, const float* w_hh,
                          const float* b_ih, const float* b_hh, const float* Ws_next, int32_t attn_dim, int32_t ap,
                          float* a_s_out, const float* W_final, const int32_t* nodes, int32_t n_ent, float* scores_all,
                          float* hidden_out, int32_t precision, void* scratch, int64_t scratch_bytes, void* stream) {
  RG_CHECK(agg && W_h && w_ih && w_hh && b_ih && b_hh && hidden_out, "rg_dense_fwd: NULL argument");
  RG_CHECK(precision >= 0 && precision <= 2, "rg_dense_fwd: precision=%d (0 = f32 MFMA, 1 = two-term f16 split, 2 = exact three-term f16 split)",
           precision);
  RG_CHECK(!prev_idx || hidden_prev, "rg_dense_fwd: prev_idx given without hidden_prev");
  RG_CHECK((d >= 1 && d <= 64) || d == 128, "rg_dense_fwd: hidden_dim %d not supported by the fused kernel (<= 64, or 128)", d);
  RG_CHECK(ld >= d && ld % 4 == 0 && ld <= 128, "rg_dense_fwd: ld=%d", ld);
  RG_CHECK(act >= 0 && act <= 2, "rg_dense_fwd: act=%d", act);
  RG_CHECK(!Ws_next || (a_s_out && attn_dim >= 1 && attn_dim <= 16 && ap >= attn_dim && ap % 4 == 0 && ap <= 16),
           "rg_dense_fwd: attn_dim=%d ap=%d not supported (<= 16)", attn_dim, ap);
  RG_CHECK(!W_final || (nodes && scores_all && n_ent > 0), "rg_dense_fwd: readout needs nodes and scores_all");
  RG_CHECK((((uintptr_t)agg | (uintptr_t)hidden_prev | (uintptr_t)hidden_out | (uintptr_t)a_s_out) & 15) == 0,
           "rg_dense_fwd: float buffers must be 16-B aligned");
  if (n == 0) return 0;
  DenseArgs A;
  A.n = n; A.n_dev = n_dev; A.n_hint = n_hint; A.d = d; A.ld4 = ld / 4;
  A.agg = (const float4*)agg; A.hprev = (const float4*)hidden_prev; A.prev_idx = prev_idx;
  A.W_h = W_h; A.w_ih = w_ih; A.w_hh = w_hh; A.b_ih = b_ih; A.b_hh = b_hh;
  A.Ws = Ws_next; A.attn = attn_dim; A.ap = ap; A.a_s_out = a_s_out;
  A.W_final = W_final; A.nodes = nodes; A.n_ent = n_ent; A.scores = scores_all;
  A.hidden_out = (float4*)hidden_out; A.act = act;
  A.n_tiles = (int)rg::ceil_div(n, 16);
  hipStream_t s = (hipStream_t)stream;
  if (d == 128)
    return precision == 1 ? rg::dense128_split_launch(A, scratch, scratch_bytes, s)
         : precision == 2 ? rg::dense128_split3_launch(A, scratch, scratch_bytes, s) : rg::dense128_launch(A, s);
  if (precision == 1) return rg::dense_split_launch(A, s);
  if (precision == 2) return rg::dense_split3_launch(A, s);
  return d <= 32 ? launch<2, false>(A, s) : launch<4, false>(A, s);
}

extern "C" int rg_dense_fwd(int64_t n, int32_t d, int32_t ld, const float* agg, const float* hidden_prev,
                            const int32_t* prev_idx, const float* W_h, int32_t act, const float* w_ih, const float* w_hh,
                            const float* b_ih, const float* b_hh, const float* Ws_next, int32_t attn_dim, int32_t ap,
                            float* a_s_out, const float* W_final, const int32_t* nodes, int32_t n_ent, float* scores_all,
                            float* hidden_out, int32_t precision, void* scratch, int64_t scratch_bytes, void* stream) {
  return dense_fwd_impl(n, nullptr, 0, d, ld, agg, hidden_prev, prev_idx, W_h, act, w_ih, w_hh, b_ih, b_hh, Ws_next, attn_dim, ap,
                        a_s_out, W_final, nodes, n_ent, scores_all, hidden_out, precision, scratch, scratch_bytes, stream);
}

extern "C" int rg_dense_fwd_dev(int64_t n_cap, const int32_t* n_dev, int64_t n_hint, int32_t d, int32_t ld, const float* agg,
                                const float* hidden_prev, const int32_t* prev_idx, const float* W_h, int32_t act,
                                const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* Ws_next,
                                int32_t attn_dim, int32_t ap, float* a_s_out, const float* W_final, const int32_t* nodes,
                                int32_t n_ent, float* scores_all, float* hidden_out, int32_t precision, void* scratch,
                                int64_t scratch_bytes, void* stream) {
  RG_CHECK(n_dev != nullptr, "rg_dense_fwd_dev: n_dev is NULL");
  return dense_fwd_impl(n_cap, n_dev, n_hint, d, ld, agg, hidden_prev, prev_idx, W_h, act, w_ih, w_hh, b_ih, b_hh, Ws_next, attn_dim, ap,
                        a_s_out, W_final, nodes, n_ent, scores_all, hidden_out, precision, scratch, scratch_bytes, stream);
}

static int dense_train_fwd_impl(const char* who, int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                                const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih,
                                const float* b_hh, const float* mask, const float* Ws_next, int32_t attn_dim, int32_t ap,
                                float* hidden_out, float* x_out, float* gates_ws_out, float* a_s_out, void* stream) {
  RG_CHECK(agg && W_h && w_ih && w_hh && b_ih && b_hh && hidden_out && x_out && gates_ws_out, "%s: NULL argument", who);
  RG_CHECK(!prev_idx || hidden_prev, "%s: prev_idx given without hidden_prev", who);
  RG_CHECK((d >= 16 && d <= 64 && d % 4 == 0) || d == 128, "%s: hidden_dim %d not supported (16..64 in steps of 4, or 128)", who, d);
  RG_CHECK(act >= 0 && act <= 2, "%s: act=%d", who, act);
  RG_CHECK(!Ws_next || (a_s_out && attn_dim >= 1 && attn_dim <= 16 && ap >= attn_dim && ap % 4 == 0 && ap <= 16),
           "%s: Ws_next needs a_s_out, 1 <= attn_dim <= 16 and ap = attn_dim padded to a multiple of 4 (got attn_dim=%d ap=%d)", who, attn_dim, ap);
  RG_CHECK((((uintptr_t)agg | (uintptr_t)hidden_prev | (uintptr_t)hidden_out | (uintptr_t)x_out | (uintptr_t)gates_ws_out |
             (uintptr_t)mask | (uintptr_t)a_s_out) & 15) == 0, "%s: float buffers must be 16-B aligned", who);
  if (n == 0) return 0;
  DenseArgs A;
  A.n = n; A.n_dev = nullptr; A.d = d; A.ld4 = d / 4;
  A.agg = (const float4*)agg; A.hprev = (const float4*)hidden_prev; A.prev_idx = prev_idx;
  A.W_h = W_h; A.w_ih = w_ih; A.w_hh = w_hh; A.b_ih = b_ih; A.b_hh = b_hh;
  A.Ws = Ws_next; A.attn = Ws_next ? attn_dim : 0; A.ap = Ws_next ? ap : 0; A.a_s_out = Ws_next ? a_s_out : nullptr;
  A.W_final = nullptr; A.nodes = nullptr; A.n_ent = 0; A.scores = nullptr;
  A.hidden_out = (float4*)hidden_out; A.act = act;
  A.n_tiles = (int)rg::ceil_div(n, 16);
  A.mask = mask; A.x_out = x_out; A.ws_out = gates_ws_out;
  hipStream_t s = (hipStream_t)stream;
  if (d == 128) return rg::dense128_launch(A, s);
  return d <= 32 ? launch<2, true>(A, s) : launch<4, true>(A, s);
}

extern "C" int rg_dense_train_fwd(int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                                  const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih,
                                  const float* b_hh, const float* mask, float* hidden_out, float* x_out, float* gates_ws_out,
                                  void* stream) {
  return dense_train_fwd_impl("rg_dense_train_fwd", n, d, agg, hidden_prev, prev_idx, W_h, act, w_ih, w_hh, b_ih, b_hh, mask, nullptr, 0, 0,
                              hidden_out, x_out, gates_ws_out, nullptr, stream);
}

// ... and the next layer's attention projection of the new state (models.py:33, Ws_attn hoisted per node as in rg_dense_fwd):
// a_s_out [n, ap] = hidden_out Ws_next^T, columns attn_dim .. ap - 1 zero
extern "C" int rg_dense_train_fwd_as(int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                                     const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih,
                                     const float* b_hh, const float* mask, const float* Ws_next, int32_t attn_dim, int32_t ap,
                                     float* hidden_out, float* x_out, float* gates_ws_out, float* a_s_out, void* stream) {
  RG_CHECK(Ws_next && a_s_out, "rg_dense_train_fwd_as: NULL Ws_next / a_s_out (use rg_dense_train_fwd)");
  return dense_train_fwd_impl("rg_dense_train_fwd_as", n, d, agg, hidden_prev, prev_idx, W_h, act, w_ih, w_hh, b_ih, b_hh, mask, Ws_next,
                              attn_dim, ap, hidden_out, x_out, gates_ws_out, a_s_out, stream);
}
