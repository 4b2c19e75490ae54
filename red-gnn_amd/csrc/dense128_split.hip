// Fused dense epilogue of one layer at hidden_dim = 128 with the products as two-term f16 splits (inference path).
// The arithmetic is dense_split.hip's (operands v = s (hi + lo), three v_mfma_f32_16x16x32_f16 per product into one fp32 accumulator, row
// scales for the activations and one power-of-two scale for all weights, exp2-folded gates, rows loaded and stored in fragment layout);
// the weight traffic is dense128.hip's: 7 x 128 x 128 weights are 458 KB as hi + lo halves and cannot live in LDS, so they stream
// through it in chunks of three 16-row blocks (24 KB), double-buffered, one barrier per chunk, the next chunk in flight from L2 while the
// eight waves of the workgroup run the current one against their own 16-node tiles.  Splitting the weights once per chunk and round
// would cost as many vector instructions as the rest of the kernel, so a small kernel writes the split image (already in the swizzled
// LDS layout: a chunk is a linear 24 KB copy) and the weight scale into a scratch buffer of the caller first; both run on the stream.
//   image: 256-byte header (1 / weight scale of the layer, of the projections), then 58 blocks of 8 KB = [hi: 16 rows x 16 slots x 16 B][lo: the same]; slot (4 s + hq) ^ (row & 15)
//   of a row holds its weights for k = 16 (2 s + j / 4) + 4 hq + j % 4, j = 0..7 (the lane's accumulator rows, as in dense_split.hip).
//   blocks 0..7 W_h, 8 + 8 g + ob weight_ih, 32 + 8 g + ob weight_hh (gate g, output block ob), 56 Ws (rows < attn), 57 W_final (row 0).
#include <type_traits>
#include "dense_common.h"

namespace rg {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DP = 128, NB = 8, KST = 4, KS = 32, NW = 8, T = 512;
constexpr int S = 32;                       // float4 per row of the node buffers
constexpr int BLK_B = 8192;                 // bytes of a 16-row block image
constexpr int CHUNK_B = 3 * BLK_B;
constexpr int N_BLOCKS = 58;
constexpr int HDR_B = 256;
constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ float resid_lo(h2 hi, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  return r;
}
__device__ __forceinline__ float resid_hi(h2 hi, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi), "v"(x));
  return r;
}
__device__ __forceinline__ void split4(float a, float b, float c, float d, h4& hi, h4& lo) {
  const f2v x0 = {a, b}, x1 = {c, d};
  const h2 h0 = __builtin_convertvector(x0, h2), h1 = __builtin_convertvector(x1, h2);
  const f2v r0 = {resid_lo(h0, a), resid_hi(h0, b)}, r1 = {resid_lo(h1, c), resid_hi(h1, d)};
  const h2 l0 = __builtin_convertvector(r0, h2), l1 = __builtin_convertvector(r1, h2);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
}
__device__ __forceinline__ void row_scale(float m, float& sc, float& inv) {
  uint32_t eb = (__float_as_uint(m) >> 23) & 0xffu;
  eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
  sc = __uint_as_float((268u - eb) << 23);
  inv = __uint_as_float((eb - 14u) << 23);
}

// ---- the split image ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(T) void prepare128_kernel(const float* __restrict__ W_h, const float* __restrict__ w_ih,
                                                        const float* __restrict__ w_hh, const float* __restrict__ Ws, int attn,
                                                        const float* __restrict__ W_final, char* __restrict__ image) {
  __shared__ uint32_t wmax_bits[2];
  if (threadIdx.x < 2) wmax_bits[threadIdx.x] = 0u;
  __syncthreads();
  // every block finds the largest magnitudes itself (117 k weights out of L2: cheaper than a second launch).  The projections (Ws,
  // W_final) get a scale of their own: the state they read does not depend on which of them a launch carries.
  float wm = 0.0f;
  auto scan = [&](const float* p, int n4) {
    for (int i = threadIdx.x; i < n4; i += T) {
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
  auto fit = [](float wmax) -> float {          // largest magnitude to [2^13, 2^14)
    if (!(wmax > 0.0f)) return 4096.0f;
    uint32_t eb = (__float_as_uint(wmax) >> 23) & 0xffu;
    eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
    return __uint_as_float((267u - eb) << 23);
  };
  const float sw_g = fit(__uint_as_float(wmax_bits[0])), sw_e = fit(__uint_as_float(wmax_bits[1]));
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
  h4 hi, lo;
  split4(q.x * sw, q.y * sw, q.z * sw, q.w * sw, hi, lo);
  const int kb = ch >> 2, hq = ch & 3;              // k block of 16, lane quarter
  const int slot = (4 * (kb >> 1) + hq) ^ (r & 15);
  char* blk = image + HDR_B + (int64_t)b * BLK_B;
  reinterpret_cast<h4*>(blk + (r * 16 + slot) * 16)[kb & 1] = hi;
  reinterpret_cast<h4*>(blk + 4096 + (r * 16 + slot) * 16)[kb & 1] = lo;
}

// ---- the layer ----------------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(T, 2) void dense128_split_kernel(DenseArgs A, const char* __restrict__ image) {
  extern __shared__ float4 lds[];
  if (A.n_dev) { A.n = *A.n_dev; A.n_tiles = (int)((A.n + 15) / 16); }
  char* wbuf = reinterpret_cast<char*>(lds);                                  // [2][CHUNK_B]
  char* E_l = wbuf + 2 * CHUNK_B;                                             // blocks 56, 57
  float* bias_l = reinterpret_cast<float*>(E_l + 2 * BLK_B);                  // [4][DP], pre-multiplied by the exp2 factors of their gates
  float4* stash = reinterpret_cast<float4*>(bias_l + 4 * DP);                 // [NW][NB][64]: each lane's old-state chunks

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
  float4* my_stash = stash + wv * NB * 64 + lane;

  // chunk loads: thread t copies the t-th 16 bytes of each of the chunk's three blocks (buffer loads: scalar block offsets)
  const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(image), 0, HDR_B + N_BLOCKS * BLK_B, 0x00020000);
  const uint32_t ld_off = HDR_B + threadIdx.x * 16;
  u32x4 pre[3];
  auto issue = [&](int b0, int b1, int b2) {
    pre[0] = __builtin_amdgcn_raw_buffer_load_b128(r_img, ld_off, b0 * BLK_B, 0);
    pre[1] = __builtin_amdgcn_raw_buffer_load_b128(r_img, ld_off, b1 * BLK_B, 0);
    pre[2] = __builtin_amdgcn_raw_buffer_load_b128(r_img, ld_off, b2 * BLK_B, 0);
  };
  int cur = 0;
  auto drop = [&]() {          // the prefetched chunk into the other buffer
    u32x4* dst = reinterpret_cast<u32x4*>(wbuf + (cur ^ 1) * CHUNK_B) + threadIdx.x;
    dst[0] = pre[0]; dst[BLK_B / 16] = pre[1]; dst[2 * BLK_B / 16] = pre[2];
  };
  auto publish = [&]() { drop(); __syncthreads(); cur ^= 1; };

  uint32_t a_off[KST];         // the lane's A-fragment slot of k-step s inside a block's hi part (re-laundered per round: immediates, not registers)
#pragma unroll
  for (int s = 0; s < KST; ++s) a_off[s] = (uint32_t)(li * 16 + ((4 * s + hq) ^ li)) * 16u;
  // c_j += block j of the current chunk times the fragment, j = 0..2
  auto mma3 = [&](const h8 (&fh)[KST], const h8 (&fl)[KST], f32x4& c0, f32x4& c1, f32x4& c2) {
    const char* wb = wbuf + cur * CHUNK_B;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const h8 h0 = *reinterpret_cast<const h8*>(wb + a_off[s]), l0 = *reinterpret_cast<const h8*>(wb + a_off[s] + 4096);
      const h8 h1 = *reinterpret_cast<const h8*>(wb + a_off[s] + BLK_B), l1 = *reinterpret_cast<const h8*>(wb + a_off[s] + BLK_B + 4096);
      const h8 h2_ = *reinterpret_cast<const h8*>(wb + a_off[s] + 2 * BLK_B), l2 = *reinterpret_cast<const h8*>(wb + a_off[s] + 2 * BLK_B + 4096);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, fh[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, fh[s], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2_, fh[s], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, fl[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, fl[s], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2_, fl[s], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, fh[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, fh[s], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l2, fh[s], c2, 0, 0, 0);
    }
  };
  auto mma_e = [&](int blk, const h8 (&fh)[KST], const h8 (&fl)[KST]) -> f32x4 {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const char* wb = E_l + blk * BLK_B;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const h8 wh = *reinterpret_cast<const h8*>(wb + a_off[s]), wl = *reinterpret_cast<const h8*>(wb + a_off[s] + 4096);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fh[s], acc, 0, 0, 0);
    }
    return acc;
  };
  auto row_max = [&](const float (&f)[KS], float m) -> float {
#pragma unroll
    for (int i = 0; i < KS; ++i) m = fmaxf(m, fabsf(f[i]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    return m;
  };
  auto split_frag = [&](const float (&f)[KS], float sc, h8 (&fh)[KST], h8 (&fl)[KST]) {
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      h4 h0, l0, h1, l1;
      split4(f[8 * s + 0] * sc, f[8 * s + 1] * sc, f[8 * s + 2] * sc, f[8 * s + 3] * sc, h0, l0);
      split4(f[8 * s + 4] * sc, f[8 * s + 5] * sc, f[8 * s + 6] * sc, f[8 * s + 7] * sc, h1, l1);
      fh[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      fl[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
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
    asm volatile("" : "+v"(lane_off));
    const bool node_ok = row0 + li < A.n;
    const int64_t node = row0 + li;

    // ---- this round's operands out of the prefetch registers; its old-state rows and the next round's agg rows go out now ------------
    float sc1, inv1;
    h8 xh[KST], xl[KST];
    {
      float fx[KS];
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) { fx[4 * ob] = va[ob].x; fx[4 * ob + 1] = va[ob].y; fx[4 * ob + 2] = va[ob].z; fx[4 * ob + 3] = va[ob].w; }
      row_scale(row_max(fx, 0.f), sc1, inv1);
      split_frag(fx, sc1, xh, xl);
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
    auto act_store = [&](const f32x4& acc, int ob) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r] * sc_out;
        if (ACT == 1) v = fmaxf(v, 0.f);
        else if (ACT == 2) v = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v)), -1.0f);
        xf[4 * ob + r] = v;
      }
    };
    {
      f32x4 c0 = zero4, c1 = zero4, c2 = zero4;
      issue(3, 4, 5);
      mma3(xh, xl, c0, c1, c2);
      act_store(c0, 0); act_store(c1, 1); act_store(c2, 2);
      publish();
    }
    {
      f32x4 c0 = zero4, c1 = zero4, c2 = zero4;
      issue(6, 7, 7);
      mma3(xh, xl, c0, c1, c2);
      act_store(c0, 3); act_store(c1, 4); act_store(c2, 5);
      publish();
    }
    {
      f32x4 c0 = zero4, c1 = zero4, c2 = zero4;
      issue(8, 16, 24);
      mma3(xh, xl, c0, c1, c2);          // (the third block repeats block 7: one chunk shape; its result is dropped)
      act_store(c0, 6); act_store(c1, 7);
      publish();
    }

    // ---- GRU gates: x and the old state share one row scale -----------------------------------------------------------------------------
    h8 gh[KST], gl[KST];
    float sc, inv;
    if (hh) {
      float hf[KS];
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        my_stash[ob * 64] = vh[ob];        // lane-private: read back per block for z * h
        hf[4 * ob] = vh[ob].x; hf[4 * ob + 1] = vh[ob].y; hf[4 * ob + 2] = vh[ob].z; hf[4 * ob + 3] = vh[ob].w;
      }
      row_scale(row_max(hf, row_max(xf, 0.f)), sc, inv);
      split_frag(hf, sc, gh, gl);
    } else {
      row_scale(row_max(xf, 0.f), sc, inv);
    }
    split_frag(xf, sc, xh, xl);
    const float inv_s = inv * inv_w * -LOG2E, inv_t = inv * inv_w * (-2.0f * LOG2E);
    h8 nh[KST], nl[KST];
    h4 nh0, nl0;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 ar = zero4, az = zero4, ai = zero4, ag = zero4;
      const bool last = ob == NB - 1;
      if (hh) issue(32 + ob, 40 + ob, 48 + ob);
      else if (!last) issue(8 + ob + 1, 16 + ob + 1, 24 + ob + 1);
      else issue(0, 1, 2);                                   // the next round's first chunk (dropped at its top)
      mma3(xh, xl, ar, az, ai);
      if (hh || !last) publish(); else __syncthreads();
      if (hh) {
        if (!last) issue(8 + ob + 1, 16 + ob + 1, 24 + ob + 1);
        else issue(0, 1, 2);
        mma3(gh, gl, ar, az, ag);
        if (!last) publish(); else __syncthreads();
      }
      const float4 br = *reinterpret_cast<const float4*>(bias_l + 0 * DP + 16 * ob + 4 * hq);
      const float4 bz = *reinterpret_cast<const float4*>(bias_l + 1 * DP + 16 * ob + 4 * hq);
      const float4 bi = *reinterpret_cast<const float4*>(bias_l + 2 * DP + 16 * ob + 4 * hq);
      const float4 bh = *reinterpret_cast<const float4*>(bias_l + 3 * DP + 16 * ob + 4 * hq);
      float4 ho = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hh) ho = my_stash[ob * 64];
      const float hov[4] = {ho.x, ho.y, ho.z, ho.w};
      const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w};
      const float biv[4] = {bi.x, bi.y, bi.z, bi.w}, bhv[4] = {bh.x, bh.y, bh.z, bh.w};
      float hnv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(ar[r], inv_s, brv[r])));
        const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(az[r], inv_s, bzv[r])));
        const float ti = fmaf(ai[r], inv_t, biv[r]);
        const float th = fmaf(ag[r], inv_t, bhv[r]);
        const float ng = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(rg, th, ti))), -1.0f);
        hnv[r] = fmaf(zg, hov[r] - ng, ng);       // (1 - z) n + z h
      }
      if (node_ok) (A.hidden_out + row0 * S)[lane_off + 4 * ob] = make_float4(hnv[0], hnv[1], hnv[2], hnv[3]);
      h4 ch, cl;
      split4(hnv[0], hnv[1], hnv[2], hnv[3], ch, cl);
      if (ob & 1) {
        nh[ob >> 1] = __builtin_shufflevector(nh0, ch, 0, 1, 2, 3, 4, 5, 6, 7);
        nl[ob >> 1] = __builtin_shufflevector(nl0, cl, 0, 1, 2, 3, 4, 5, 6, 7);
      } else {
        nh0 = ch; nl0 = cl;
      }
    }

    // ---- projections of the new state -------------------------------------------------------------------------------------------------
    if (A.Ws) {
      const f32x4 ae = mma_e(0, nh, nl);
      if (node_ok && 4 * hq < A.ap)
        reinterpret_cast<float4*>(A.a_s_out + node * A.ap)[hq] = make_float4(ae[0] * inv_e, ae[1] * inv_e, ae[2] * inv_e, ae[3] * inv_e);
    }
    if (A.W_final) {
      const f32x4 ae = mma_e(1, nh, nl);
      if (node_ok && hq == 0) A.scores[(int64_t)qe.x * A.n_ent + qe.y] = ae[0] * inv_e;
    }
  }
}

template <int ACT>
int launch(const DenseArgs& A, const char* image, hipStream_t s) {
  const size_t lds = 2 * CHUNK_B + 2 * BLK_B + 4 * DP * sizeof(float) + (size_t)NW * NB * 64 * sizeof(float4);
  RG_HIP(hipFuncSetAttribute((const void*)dense128_split_kernel<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL((dense128_split_kernel<ACT>), dim3(grid), dim3(T), lds, s, A, image);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int64_t dense128_split_scratch_bytes() { return HDR_B + (int64_t)N_BLOCKS * BLK_B; }

int dense128_split_launch(const DenseArgs& A, void* scratch, int64_t scratch_bytes, hipStream_t s) {
  RG_CHECK(A.d == DP && A.ld4 == S, "rg_dense_fwd: the d = 128 kernel needs ld = 128 (got d=%d ld=%d)", A.d, A.ld4 * 4);
  RG_CHECK(scratch && scratch_bytes >= dense128_split_scratch_bytes() && ((uintptr_t)scratch & 255) == 0,
           "rg_dense_fwd: precision 1 at d = 128 needs a 256-B aligned scratch of rg_dense_scratch_bytes(128, 1) = %lld bytes",
           (long long)dense128_split_scratch_bytes());
  char* image = (char*)scratch;
  hipLaunchKernelGGL(prepare128_kernel, dim3(N_BLOCKS), dim3(T), 0, s, A.W_h, A.w_ih, A.w_hh, A.Ws, A.attn, A.W_final, image);
  RG_LAUNCH_CHECK();
  return A.act == 0 ? launch<0>(A, image, s) : A.act == 1 ? launch<1>(A, image, s) : launch<2>(A, image, s);
}

}  // namespace rg
