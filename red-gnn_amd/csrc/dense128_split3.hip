// Fused dense epilogue of one layer at hidden_dim = 128 with fp32 arithmetic on the f16 matrix pipe: the exact three-term splits of
// dense_split3.hip / split3.h (operands hi + mid + lo f16 = all 24 bits, six partial products per product, fp32 accumulation, the
// weights' lo image as bf8) on the weight traffic of dense128_split.hip: the 7 x 128 x 128 weights are 560 KB as hi + mid + lo8 and
// stream through LDS in chunks of three 16-row blocks (30 KB), double-buffered, one barrier per chunk, the next chunk in flight from L2
// while the waves run the current one against their own 16-node tiles.  A small kernel writes the split image (already in the
// swizzled LDS layout: a chunk is three linear 10 KB copies) into a scratch buffer of the caller first.
// Per operand a wave holds 56 registers of fragments (4 k-steps x {hi, mid, lo: 4, bf8: 2}); two operands, the new state and the
// prefetches do not fit 256 registers, so the workgroup is FOUR waves, one per SIMD with the whole 512-register file each (two waves
// per SIMD gain almost nothing on these kernels: dense_split3.hip measured 8 waves per CU against 4 at 3.60 / 4.03 ms).
//   image: 256-byte header (1 / weight scale of the layer, of the projections), then 58 blocks of 10 KB =
//          [hi: 16 rows x 16 slots x 16 B][mid: the same][lo8: 16 rows x 128 B]
//   f16 parts: slot (4 s + hq) ^ (row & 15) of a row holds its weights for k = 16 (2 s + j / 4) + 4 hq + j % 4, j = 0..7 (k-step s,
//          lane quarter hq: the lane's accumulator rows, as in dense_split.hip)
//   lo8 part: the 16-B quad ((hq & 1) + 2 half + 4 (hq >> 1)) ^ (2 ((row >> 1) & 3)) of a row holds k-steps 2 half, 2 half + 1 of
//          quarter hq (8 bytes each): a ds_read_b128 per half, whose lane groups cover the 64 banks
//   blocks 0..7 W_h, 8 + 8 g + ob weight_ih, 32 + 8 g + ob weight_hh (gate g, output block ob), 56 Ws (rows < attn), 57 W_final (row 0).
#include <type_traits>
#include <utility>
#include "dense_common.h"
#include "split3.h"

namespace rg {
namespace {

using namespace rg::sp3;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long l2v __attribute__((ext_vector_type(2)));

#define RG_PIN(x) asm volatile("" : "+v"(x))      // see dense_split3.hip: pins the order of the instruction that produced x
#define RG_PIN_ACC(x) asm volatile("" : "+a"(x))  // the same for an MFMA accumulator, which lives in the accumulation registers here

template <class Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int DP = 128, NB = 8, KST = 4, KS = 32, NW = 4, T = 256;
constexpr int S = 32;                       // float4 per row of the node buffers
constexpr int F16_B = 4096;                 // bytes of one f16 part of a 16-row block
constexpr int LO_B = 2048;                  // bytes of the bf8 part
constexpr int BLK_B = 2 * F16_B + LO_B;     // 10 KB
constexpr int CHUNK_B = 3 * BLK_B;
constexpr int N_BLOCKS = 58;
constexpr int HDR_B = 256;
constexpr int PREP_T = 512;
constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ int lo_quad(int row, int hq, int half) { return (((hq & 1) + 2 * half + 4 * (hq >> 1)) ^ (2 * ((row >> 1) & 3))); }

// ---- the split image ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PREP_T) void prepare128x3_kernel(const float* __restrict__ W_h, const float* __restrict__ w_ih,
                                                               const float* __restrict__ w_hh, const float* __restrict__ Ws, int attn,
                                                               const float* __restrict__ W_final, char* __restrict__ image) {
  __shared__ uint32_t wmax_bits[2];
  if (threadIdx.x < 2) wmax_bits[threadIdx.x] = 0u;
  __syncthreads();
  // every block finds the largest magnitudes itself (117 k weights out of L2: cheaper than a second launch).  The projections (Ws,
  // W_final) get a scale of their own: the state they read does not depend on which of them a launch carries.
  float wm = 0.0f;
  auto scan = [&](const float* p, int n4) {
    for (int i = threadIdx.x; i < n4; i += PREP_T) {
      const float4 q = reinterpret_cast<const float4*>(p)[i];
      wm = fmaxf(fmaxf(wm, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    }
  };
  scan(W_h, DP * DP / 4); scan(w_ih, 3 * DP * DP / 4); scan(w_hh, 3 * DP * DP / 4);
  atomicMax(&wmax_bits[0], __float_as_uint(wm));
  wm = 0.0f;
  if (Ws) scan(Ws, attn * DP / 4);
  if (W_final) scan(W_final, DP / 4);
  atomicMax(&wmax_bits[1], __float_as_uint(wm));
  __syncthreads();
  const float sw_g = fit_weight_scale(__uint_as_float(wmax_bits[0])), sw_e = fit_weight_scale(__uint_as_float(wmax_bits[1]));
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0) { reinterpret_cast<float*>(image)[0] = 1.0f / sw_g; reinterpret_cast<float*>(image)[1] = 1.0f / sw_e; }
  const float sw = b < 56 ? sw_g : sw_e;
  const int r = threadIdx.x >> 5, ch = threadIdx.x & 31;          // row of the block, 4-float chunk of the row
  const float* src = nullptr;
  if (b < 8) src = W_h + (int64_t)(16 * b + r) * DP;
  else if (b < 32) src = w_ih + (int64_t)(16 * (b - 8) + r) * DP;
  else if (b < 56) src = w_hh + (int64_t)(16 * (b - 32) + r) * DP;
  else if (b == 56) src = (Ws && r < attn) ? Ws + (int64_t)r * DP : nullptr;
  else src = (W_final && r == 0) ? W_final : nullptr;
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (src) q = reinterpret_cast<const float4*>(src)[ch];
  h4 hi, mid;
  uint32_t lo8;
  split3_4_lo8(q.x * sw, q.y * sw, q.z * sw, q.w * sw, hi, mid, lo8);
  const int kb = ch >> 2, hq = ch & 3;              // k block of 16 (= 2 s + j / 4), lane quarter
  const int s = kb >> 1;
  const int slot = (4 * s + hq) ^ (r & 15);
  char* blk = image + HDR_B + (int64_t)b * BLK_B;
  reinterpret_cast<h4*>(blk + (r * 16 + slot) * 16)[kb & 1] = hi;
  reinterpret_cast<h4*>(blk + F16_B + (r * 16 + slot) * 16)[kb & 1] = mid;
  // lo8: quad of (hq, half = s / 2), then 8 bytes per k-step (s & 1), 4 bytes per half k-step (kb & 1)
  reinterpret_cast<uint32_t*>(blk + 2 * F16_B + r * 128 + lo_quad(r, hq, s >> 1) * 16 + (s & 1) * 8)[kb & 1] = lo8;
}

struct Frag4 {          // B operand of one node row: hi / mid / lo f16 and the bf8 form, per k-step
  h8 h[KST], m[KST], l[KST];
  long q[KST];
};

// ---- the layer ----------------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(T, 1) void dense128_split3_kernel(DenseArgs A, const char* __restrict__ image) {
  extern __shared__ float4 lds[];
  if (A.n_dev) { A.n = *A.n_dev; A.n_tiles = (int)((A.n + 15) / 16); }
  char* wbuf = reinterpret_cast<char*>(lds);                                  // [2][CHUNK_B]
  char* E_l = wbuf + 2 * CHUNK_B;                                             // blocks 56, 57
  float* bias_l = reinterpret_cast<float*>(E_l + 2 * BLK_B);                  // [4][DP], pre-multiplied by the exp2 factors of their gates

  const float inv_w = reinterpret_cast<const float*>(image)[0], inv_e = reinterpret_cast<const float*>(image)[1];
  for (int i = threadIdx.x; i < 2 * BLK_B / 16; i += T)
    reinterpret_cast<float4*>(E_l)[i] = reinterpret_cast<const float4*>(image + HDR_B + 56 * BLK_B)[i];
  for (int i = threadIdx.x; i < 4 * DP; i += T) {
    const int g = i / DP, c = i - g * DP;
    bias_l[i] = g == 0 ? -LOG2E * (A.b_ih[c] + A.b_hh[c]) : g == 1 ? -LOG2E * (A.b_ih[DP + c] + A.b_hh[DP + c])
              : g == 2 ? -2.0f * LOG2E * A.b_ih[2 * DP + c] : -2.0f * LOG2E * A.b_hh[2 * DP + c];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, hq = lane >> 4;

  // chunk loads: a block is 640 16-byte pieces: thread t copies pieces t, t + 256 and (t < 128) t + 512 of each of the chunk's blocks
  const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image), 0, HDR_B + N_BLOCKS * BLK_B, 0x00020000);
  const uint32_t ld_off = HDR_B + threadIdx.x * 16;
  const bool third = threadIdx.x < 128;
  u32x4 pre[9];
  auto issue = [&](int b0, int b1, int b2) {
    const int bb[3] = {b0, b1, b2};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      pre[3 * j] = __builtin_amdgcn_raw_buffer_load_b128(r_img, ld_off, bb[j] * BLK_B, 0);
      pre[3 * j + 1] = __builtin_amdgcn_raw_buffer_load_b128(r_img, ld_off + 4096, bb[j] * BLK_B, 0);
      // (beyond the block for t >= 128: still inside the image except for the very last block, where the descriptor's range check
      // returns zeros; the piece is not stored)
      pre[3 * j + 2] = __builtin_amdgcn_raw_buffer_load_b128(r_img, ld_off + 8192, bb[j] * BLK_B, 0);
    }
  };
  int cur = 0;
  auto drop = [&]() {          // the prefetched chunk into the other buffer
    u32x4* dst = reinterpret_cast<u32x4*>(wbuf + (cur ^ 1) * CHUNK_B) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dst[j * (BLK_B / 16)] = pre[3 * j];
      dst[j * (BLK_B / 16) + 256] = pre[3 * j + 1];
      if (third) dst[j * (BLK_B / 16) + 512] = pre[3 * j + 2];
    }
  };
  auto publish = [&]() { drop(); __syncthreads(); cur ^= 1; };

  uint32_t a_off[KST];         // the lane's A-fragment slot of k-step s inside a block's hi part (re-laundered per round: immediates, not registers)
#pragma unroll
  for (int s = 0; s < KST; ++s) a_off[s] = (uint32_t)(li * 16 + ((4 * s + hq) ^ li)) * 16u;
  uint32_t c_off[2];           // the lane's two lo8 quads
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) c_off[hf] = (uint32_t)(2 * F16_B + li * 128 + lo_quad(li, hq, hf) * 16);
  // c_j (f16 chain) and e_j (bf8 chain) += block j of the current chunk times the fragment, j = 0..2
  auto mma3 = [&](const Frag4& X, f32x4 (&c)[3], f32x4 (&e)[3]) {
    const char* wb = wbuf + cur * CHUNK_B;
    l2v wl[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      wl[j][0] = *reinterpret_cast<const l2v*>(wb + j * BLK_B + c_off[0]);
      wl[j][1] = *reinterpret_cast<const l2v*>(wb + j * BLK_B + c_off[1]);
    }
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      h8 wh[3], wm[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wh[j] = *reinterpret_cast<const h8*>(wb + j * BLK_B + a_off[s]);
        wm[j] = *reinterpret_cast<const h8*>(wb + j * BLK_B + F16_B + a_off[s]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) e[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wl[j][s >> 1][s & 1], X.q[s], e[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], X.l[s], c[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], X.m[s], c[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], X.h[s], c[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], X.m[s], c[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], X.h[s], c[j], 0, 0, 0);
    }
  };
  // mma3 with vector operations of the PREVIOUS output block's gate arithmetic placed behind its MFMAs, one MFMA at a time (a wave
  // alone on its SIMD pays 7 counter ticks per vector instruction in a vector-only stretch and ~2.5 behind an MFMA:
  // tools/hipcheck/mfma_valu_roles.hip): op(k) for k in [k0, k1), spread evenly over the 72 MFMAs
  auto mma3_ops = [&](const Frag4& X, f32x4 (&c)[3], f32x4 (&e)[3], auto K0, auto K1, auto&& op) {
    constexpr int k0 = decltype(K0)::value, k1 = decltype(K1)::value, NM = 18 * KST;
    const char* wb = wbuf + cur * CHUNK_B;
    l2v wl[3][2];
    h8 wh[3], wm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      wl[j][0] = *reinterpret_cast<const l2v*>(wb + j * BLK_B + c_off[0]);
      wl[j][1] = *reinterpret_cast<const l2v*>(wb + j * BLK_B + c_off[1]);
    }
    static_for<NM>([&](auto M) {
      constexpr int m = decltype(M)::value, s_ = m / 18, term = (m % 18) / 3, j = m % 3;
      if constexpr (m % 18 == 0) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
          wh[jj] = *reinterpret_cast<const h8*>(wb + jj * BLK_B + a_off[s_]);
          wm[jj] = *reinterpret_cast<const h8*>(wb + jj * BLK_B + F16_B + a_off[s_]);
        }
      }
      if constexpr (term == 0) { e[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wl[j][s_ >> 1][s_ & 1], X.q[s_], e[j], 0, 0, 0); RG_PIN_ACC(e[j]); }
      else if constexpr (term == 1) { c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], X.l[s_], c[j], 0, 0, 0); RG_PIN_ACC(c[j]); }
      else if constexpr (term == 2) { c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], X.m[s_], c[j], 0, 0, 0); RG_PIN_ACC(c[j]); }
      else if constexpr (term == 3) { c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[j], X.h[s_], c[j], 0, 0, 0); RG_PIN_ACC(c[j]); }
      else if constexpr (term == 4) { c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], X.m[s_], c[j], 0, 0, 0); RG_PIN_ACC(c[j]); }
      else { c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], X.h[s_], c[j], 0, 0, 0); RG_PIN_ACC(c[j]); }
      constexpr int a = k0 + (m * (k1 - k0)) / NM, b = k0 + ((m + 1) * (k1 - k0)) / NM;
      static_for<b - a>([&](auto Q) { op(std::integral_constant<int, a + decltype(Q)::value>{}); });
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto join = [&](f32x4& acc, const f32x4& acc8) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(acc8[r], LO8_INV, acc[r]);
  };
  auto mma_e = [&](int blk, const Frag4& X) -> f32x4 {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc8 = {0.f, 0.f, 0.f, 0.f};
    const char* wb = E_l + blk * BLK_B;
    const l2v w0 = *reinterpret_cast<const l2v*>(wb + c_off[0]), w1 = *reinterpret_cast<const l2v*>(wb + c_off[1]);
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const h8 wh = *reinterpret_cast<const h8*>(wb + a_off[s]), wm = *reinterpret_cast<const h8*>(wb + F16_B + a_off[s]);
      acc8 = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8((s < 2 ? w0 : w1)[s & 1], X.q[s], acc8, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.l[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm, X.m[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm, X.h[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.m[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.h[s], acc, 0, 0, 0);
    }
    join(acc, acc8);
    return acc;
  };
  auto row_max = [&](const float (&f)[KS], float m) -> float {
#pragma unroll
    for (int i = 0; i < KS; ++i) m = fmaxf(m, fabsf(f[i]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    return m;
  };
  auto split_frag = [&](const float (&f)[KS], float sc, Frag4& X) {
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f[8 * s + j] * sc;
      h4 h0, m0, l0, h1, m1, l1;
      split3_4(v[0], v[1], v[2], v[3], h0, m0, l0);
      split3_4(v[4], v[5], v[6], v[7], h1, m1, l1);
      X.h[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      X.m[s] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
      X.l[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      X.q[s] = (long)(((uint64_t)to_bf8x4(v[4], v[5], v[6], v[7]) << 32) | to_bf8x4(v[0], v[1], v[2], v[3]));
    }
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // rows in fragment layout: lane (li, hq) owns columns 16 ob + 4 hq .. +3 of node li
  uint32_t lane_off = (uint32_t)(li * S + hq);
  auto load_prev = [&](int t) -> int {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    int p = -1;
    if (A.prev_idx && ts < A.n_tiles && (int64_t)ts * 16 + li < A.n) p = (A.prev_idx + (int64_t)ts * 16)[li];
    return p;
  };
  auto load_agg = [&](int t, float4 (&va)[NB]) {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    const bool row_ok = ts < A.n_tiles && (int64_t)ts * 16 + li < A.n;
    const float4* arow = A.agg + (int64_t)ts * 16 * S;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) va[ob] = row_ok ? arow[lane_off + 4 * ob] : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  const int n_rounds = (A.n_tiles + NW - 1) / NW;
  float4 va[NB];
  int round = blockIdx.x;
  int p_cur = load_prev(round * NW + wv);
  load_agg(round * NW + wv, va);
  int p_next = load_prev((round + (int)gridDim.x) * NW + wv);
  if (round < n_rounds) issue(0, 1, 2);
  for (; round < n_rounds; round += gridDim.x) {
    const int ts = __builtin_amdgcn_readfirstlane(round * NW + wv);
    const int64_t row0 = (int64_t)ts * 16;
#pragma unroll
    for (int s = 0; s < KST; ++s) asm volatile("" : "+v"(a_off[s]));
    asm volatile("" : "+v"(c_off[0]));
    asm volatile("" : "+v"(c_off[1]));
    asm volatile("" : "+v"(lane_off));
    const bool node_ok = row0 + li < A.n;
    const int64_t node = row0 + li;

    // ---- this round's operands out of the prefetch registers; its old-state rows and the next round's agg rows go out now ------------
    float sc1, inv1;
    Frag4 X;
    {
      float fx[KS];
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) { fx[4 * ob] = va[ob].x; fx[4 * ob + 1] = va[ob].y; fx[4 * ob + 2] = va[ob].z; fx[4 * ob + 3] = va[ob].w; }
      row_scale(row_max(fx, 0.f), sc1, inv1);
      split_frag(fx, sc1, X);
    }
    float4 vh[NB];
    {
      const float4* hrow = A.hprev + ((int64_t)(p_cur < 0 ? 0 : p_cur) * S + hq);
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) vh[ob] = p_cur >= 0 ? hrow[4 * ob] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int has_old = p_cur >= 0;
    p_cur = p_next;
    load_agg((round + (int)gridDim.x) * NW + wv, va);
    p_next = load_prev((round + 2 * (int)gridDim.x) * NW + wv);
    int2 qe = make_int2(0, 0);
    if (A.W_final && node_ok && hq == 0) qe = reinterpret_cast<const int2*>(A.nodes)[node];

    // chunk 0 (in flight since the previous round's last chunk) becomes current; the vote is its barrier
    drop();
    const bool hh = __syncthreads_or(has_old) != 0;   // a round of new nodes only (early hops): h = 0, weight_hh is skipped
    cur ^= 1;

    // ---- stage 1: x = act(W_h agg) ------------------------------------------------------------------------------------------------
    float xf[KS];
    const float sc_out = inv1 * inv_w * (ACT == 2 ? -2.0f * LOG2E : 1.0f);
    auto act_store = [&](f32x4 acc, const f32x4& acc8, int ob) {
      join(acc, acc8);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r] * sc_out;
        if (ACT == 1) v = fmaxf(v, 0.f);
        else if (ACT == 2) v = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v)), -1.0f);
        xf[4 * ob + r] = v;
      }
    };
    {
      f32x4 c[3] = {zero4, zero4, zero4}, e[3] = {zero4, zero4, zero4};
      issue(3, 4, 5);
      mma3(X, c, e);
      act_store(c[0], e[0], 0); act_store(c[1], e[1], 1); act_store(c[2], e[2], 2);
      publish();
    }
    {
      f32x4 c[3] = {zero4, zero4, zero4}, e[3] = {zero4, zero4, zero4};
      issue(6, 7, 7);
      mma3(X, c, e);
      act_store(c[0], e[0], 3); act_store(c[1], e[1], 4); act_store(c[2], e[2], 5);
      publish();
    }
    {
      f32x4 c[3] = {zero4, zero4, zero4}, e[3] = {zero4, zero4, zero4};
      issue(8, 16, 24);
      mma3(X, c, e);          // (the third block repeats block 7: one chunk shape; its result is dropped)
      act_store(c[0], e[0], 6); act_store(c[1], e[1], 7);
      publish();
    }

    // ---- GRU gates: x and the old state share one row scale -----------------------------------------------------------------------------
    Frag4 H;
    float sc, inv;
    float hf[KS];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) { hf[4 * ob] = vh[ob].x; hf[4 * ob + 1] = vh[ob].y; hf[4 * ob + 2] = vh[ob].z; hf[4 * ob + 3] = vh[ob].w; }
    if (hh) {
      row_scale(row_max(hf, row_max(xf, 0.f)), sc, inv);
      split_frag(hf, sc, H);
    } else {
      row_scale(row_max(xf, 0.f), sc, inv);
    }
    split_frag(xf, sc, X);
    const float inv_s = inv * inv_w * -LOG2E, inv_t = inv * inv_w * (-2.0f * LOG2E);
    float hn[KS];
    // new-state rows go out through a buffer descriptor of this tile's valid rows (no branch inside the interleaved stream)
    const int rows_valid = (int)(A.n - row0 < 16 ? (A.n - row0 > 0 ? A.n - row0 : 0) : 16);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(A.hidden_out + row0 * S), 0,
                                                                            rows_valid * S * 16, 0x00020000);
    // The loop over output blocks is software-pipelined by one block: the gate arithmetic of block ob - 1 (joins of the bf8 chains, the
    // three gates, the new state and its store: 105 operations) is issued between the MFMAs of block ob's two chunks.
    auto gate_loop = [&](auto HH_) {
      constexpr bool HH = decltype(HH_)::value;
      f32x4 P[2][4][3];                               // [block parity][c (r, z, W_in x), e (their bf8 chains), cg (r, z, W_hn h), eg][3]
      static_for<NB + 1>([&](auto OB) {
        constexpr int ob = decltype(OB)::value, cs = ob & 1, ps = cs ^ 1, pb = ob - 1;
        constexpr int OPS_R = 25, NOPS = 4 + 4 * OPS_R + 1;
        float4 bia[4];
        float s1[4], a1[4], ar[4], az[4], ai[4], ag[4], t1[4], t2[4], e1[4], e2[4], rg[4], zg[4], ti[4], th[4], uu[4], e3[4], ng[4], hnv[4];
        auto gate_op = [&](auto K) {
          constexpr int k = decltype(K)::value;
          f32x4 (&c)[3] = P[ps][0]; f32x4 (&e)[3] = P[ps][1]; f32x4 (&cg)[3] = P[ps][2]; f32x4 (&eg)[3] = P[ps][3];
          if constexpr (k < 4) {
            bia[k] = *reinterpret_cast<const float4*>(bias_l + k * DP + 16 * pb + 4 * hq);
            RG_PIN(bia[k].x);
          } else if constexpr (k < 4 + 4 * OPS_R) {
            constexpr int j = (k - 4) / 4, r = (k - 4) % 4;
            const float bvr = r == 0 ? bia[0].x : r == 1 ? bia[0].y : r == 2 ? bia[0].z : bia[0].w;
            const float bvz = r == 0 ? bia[1].x : r == 1 ? bia[1].y : r == 2 ? bia[1].z : bia[1].w;
            const float bvi = r == 0 ? bia[2].x : r == 1 ? bia[2].y : r == 2 ? bia[2].z : bia[2].w;
            const float bvh = r == 0 ? bia[3].x : r == 1 ? bia[3].y : r == 2 ? bia[3].z : bia[3].w;
            // r and z gates: W_i. x + W_h. h in one sum (f16 chains added, bf8 chains added and joined); the n gate keeps its products apart
            if constexpr (j == 0) { s1[r] = HH ? e[0][r] + eg[0][r] : e[0][r]; RG_PIN(s1[r]); }
            else if constexpr (j == 1) { a1[r] = HH ? c[0][r] + cg[0][r] : c[0][r]; RG_PIN(a1[r]); }
            else if constexpr (j == 2) { ar[r] = fmaf(s1[r], LO8_INV, a1[r]); RG_PIN(ar[r]); }
            else if constexpr (j == 3) { s1[r] = HH ? e[1][r] + eg[1][r] : e[1][r]; RG_PIN(s1[r]); }
            else if constexpr (j == 4) { a1[r] = HH ? c[1][r] + cg[1][r] : c[1][r]; RG_PIN(a1[r]); }
            else if constexpr (j == 5) { az[r] = fmaf(s1[r], LO8_INV, a1[r]); RG_PIN(az[r]); }
            else if constexpr (j == 6) { ai[r] = fmaf(e[2][r], LO8_INV, c[2][r]); RG_PIN(ai[r]); }
            else if constexpr (j == 7) { ag[r] = HH ? fmaf(eg[2][r], LO8_INV, cg[2][r]) : 0.f; RG_PIN(ag[r]); }
            else if constexpr (j == 8) { t1[r] = fmaf(ar[r], inv_s, bvr); RG_PIN(t1[r]); }
            else if constexpr (j == 9) { e1[r] = __builtin_amdgcn_exp2f(t1[r]); RG_PIN(e1[r]); }
            else if constexpr (j == 10) { t2[r] = fmaf(az[r], inv_s, bvz); RG_PIN(t2[r]); }
            else if constexpr (j == 11) { e2[r] = __builtin_amdgcn_exp2f(t2[r]); RG_PIN(e2[r]); }
            else if constexpr (j == 12) { t1[r] = 1.0f + e1[r]; RG_PIN(t1[r]); }
            else if constexpr (j == 13) { t2[r] = 1.0f + e2[r]; RG_PIN(t2[r]); }
            else if constexpr (j == 14) { rg[r] = __builtin_amdgcn_rcpf(t1[r]); RG_PIN(rg[r]); }
            else if constexpr (j == 15) { zg[r] = __builtin_amdgcn_rcpf(t2[r]); RG_PIN(zg[r]); }
            else if constexpr (j == 16) { ti[r] = fmaf(ai[r], inv_t, bvi); RG_PIN(ti[r]); }
            else if constexpr (j == 17) { th[r] = fmaf(ag[r], inv_t, bvh); RG_PIN(th[r]); }
            else if constexpr (j == 18) { uu[r] = fmaf(rg[r], th[r], ti[r]); RG_PIN(uu[r]); }
            else if constexpr (j == 19) { e3[r] = __builtin_amdgcn_exp2f(uu[r]); RG_PIN(e3[r]); }
            else if constexpr (j == 20) { uu[r] = 1.0f + e3[r]; RG_PIN(uu[r]); }
            else if constexpr (j == 21) { e3[r] = __builtin_amdgcn_rcpf(uu[r]); RG_PIN(e3[r]); }
            else if constexpr (j == 22) { ng[r] = fmaf(2.0f, e3[r], -1.0f); RG_PIN(ng[r]); }
            else if constexpr (j == 23) { uu[r] = hf[4 * pb + r] - ng[r]; RG_PIN(uu[r]); }
            else { hnv[r] = fmaf(zg[r], uu[r], ng[r]); RG_PIN(hnv[r]); }           // (1 - z) n + z h  (the old state stays in registers)
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[4 * pb + r] = hnv[r];
            const u32x4 bits = {__float_as_uint(hnv[0]), __float_as_uint(hnv[1]), __float_as_uint(hnv[2]), __float_as_uint(hnv[3])};
            __builtin_amdgcn_raw_buffer_store_b128(bits, r_out, (lane_off + 4u * pb) * 16u, 0, 0);
          }
        };
        auto no_op = [&](auto) {};
        if constexpr (ob < NB) {
          constexpr bool last = ob == NB - 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) { P[cs][q][0] = zero4; P[cs][q][1] = zero4; P[cs][q][2] = zero4; }
          // operations of block ob - 1 behind this block's MFMAs: all of them behind the first chunk when there is no second one
          constexpr int n_first = ob == 0 ? 0 : (HH ? NOPS / 2 : NOPS);
          if constexpr (HH) issue(32 + ob, 40 + ob, 48 + ob);
          else if constexpr (!last) issue(8 + ob + 1, 16 + ob + 1, 24 + ob + 1);
          else issue(0, 1, 2);                                   // the next round's first chunk (dropped at its top)
          if constexpr (ob == 0) mma3_ops(X, P[cs][0], P[cs][1], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, no_op);
          else mma3_ops(X, P[cs][0], P[cs][1], std::integral_constant<int, 0>{}, std::integral_constant<int, n_first>{}, gate_op);
          if constexpr (HH || !last) publish(); else __syncthreads();
          if constexpr (HH) {
            if constexpr (!last) issue(8 + ob + 1, 16 + ob + 1, 24 + ob + 1);
            else issue(0, 1, 2);
            if constexpr (ob == 0) mma3_ops(H, P[cs][2], P[cs][3], std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, no_op);
            else mma3_ops(H, P[cs][2], P[cs][3], std::integral_constant<int, n_first>{}, std::integral_constant<int, NOPS>{}, gate_op);
            if constexpr (!last) publish(); else __syncthreads();
          }
        } else {
          static_for<NOPS>(gate_op);                             // the last block's gates: nothing left to hide them behind
        }
      });
    };
    if (hh) gate_loop(std::true_type{}); else gate_loop(std::false_type{});

    // ---- projections of the new state (at scale 2^14) -------------------------------------------------------------------------------------
    if (A.Ws || A.W_final) split_frag(hn, 16384.0f, X);
    constexpr float inv_n = 1.0f / 16384.0f;
    if (A.Ws) {
      const f32x4 ae = mma_e(0, X);
      const float se = inv_e * inv_n;
      if (node_ok && 4 * hq < A.ap)
        reinterpret_cast<float4*>(A.a_s_out + node * A.ap)[hq] = make_float4(ae[0] * se, ae[1] * se, ae[2] * se, ae[3] * se);
    }
    if (A.W_final) {
      const f32x4 ae = mma_e(1, X);
      if (node_ok && hq == 0) A.scores[(int64_t)qe.x * A.n_ent + qe.y] = ae[0] * (inv_e * inv_n);
    }
  }
}

template <int ACT>
int launch(const DenseArgs& A, const char* image, hipStream_t s) {
  const size_t lds = 2 * CHUNK_B + 2 * BLK_B + 4 * DP * sizeof(float);
  RG_HIP(hipFuncSetAttribute((const void*)dense128_split3_kernel<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL((dense128_split3_kernel<ACT>), dim3(grid), dim3(T), lds, s, A, image);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int64_t dense128_split3_scratch_bytes() { return HDR_B + (int64_t)N_BLOCKS * BLK_B; }

int dense128_split3_launch(const DenseArgs& A, void* scratch, int64_t scratch_bytes, hipStream_t s) {
  RG_CHECK(A.d == DP && A.ld4 == S, "rg_dense_fwd: the d = 128 kernel needs ld = 128 (got d=%d ld=%d)", A.d, A.ld4 * 4);
  RG_CHECK(scratch && scratch_bytes >= dense128_split3_scratch_bytes() && ((uintptr_t)scratch & 255) == 0,
           "rg_dense_fwd: precision 2 at d = 128 needs a 256-B aligned scratch of rg_dense_scratch_bytes(128, 2) = %lld bytes",
           (long long)dense128_split3_scratch_bytes());
  char* image = (char*)scratch;
  hipLaunchKernelGGL(prepare128x3_kernel, dim3(N_BLOCKS), dim3(PREP_T), 0, s, A.W_h, A.w_ih, A.w_hh, A.Ws, A.attn, A.W_final, image);
  RG_LAUNCH_CHECK();
  return A.act == 0 ? launch<0>(A, image, s) : A.act == 1 ? launch<1>(A, image, s) : launch<2>(A, image, s);
}

}  // namespace rg
