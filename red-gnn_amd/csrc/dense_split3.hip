// Fused dense epilogue of one layer with fp32 arithmetic on the f16 matrix pipe: every operand as an EXACT three-term f16 split
// (split3.h), every product as its six partial products of order >= 2^-22 in one fp32 accumulator (inference path, d <= 64; the
// model's default).  Same operator as dense.hip (Static/transductive/models.py :41 W_h + act, :81-84 the gathered single-step GRU, the
// next layer's :36 Ws_attn(hs), :86-88 the readout) and the data flow of dense_split.hip (node = lane, accumulators feed the next
// product, rows loaded and stored in fragment layout, exp2-folded gates); what differs:
//   * operands are hi + mid + lo (33 >= 24 bits): the f32 operands of the reference's nn.Linear / nn.GRU are carried exactly, not
//     rounded to 22 bits as in the two-term kernel; products are good to 2^-31, sums are fp32 (the MFMA accumulator);
//   * the weights' lo image is bf8 (one significant bit per weight; 4 KB per 64 x 64 matrix instead of 8), multiplied on
//     v_mfma_f32_16x16x32_bf8_bf8 (same operand map as the f16 form: tools/hipcheck/bf8_layout.hip): hi + mid + lo8 images of the
//     seven matrices and the projections are 150 KB at d = 64 and stay resident in LDS;
//   * which leaves no LDS for the old state's lane-private copy: z * h takes h back from its own (exact) split fragments, and the old
//     rows of the NEXT tile are requested as soon as this tile's are split - a full tile of work ahead of their use.
// rg_dense_fwd(..., precision = 2).  precision 0 = v_mfma_f32_16x16x4_f32 (dense.hip), 1 = two-term splits (dense_split.hip).
#include <type_traits>
#include <utility>
#include "dense_common.h"
#include "split3.h"

using namespace rg;
using namespace rg::sp3;

namespace {

constexpr int DENSE_T = 512;
constexpr float LOG2E = 1.44269504088896340736f;

// DP in {32, 64}: padded width.  f16 weight images: per row DP values of hi and, in a second image, of mid; the 16-B slot (k-step s,
// lane quarter hq) holds the row's weights for k = 16 * (2s + j / 4) + 4 * hq + j % 4, j = 0..7 - the k order in which a lane holds
// its accumulator rows, so that accumulators convert in place into the next B fragment.  Slots are XOR-swizzled with the row so that
// the 16 rows read by a quarter wave cover the 16 bank groups.  The bf8 lo image has the same slots at 8 B each, swizzled so that bank-conflict
// free for the one read per product that fetches all its k-steps.
template <int DP>
struct Geo {
  static constexpr int SR = DP / 8;                  // slots per image row
  static constexpr int SH = DP == 64 ? 1 : 2;        // rows per 256 B of the f16 images
  static constexpr int KST = DP / 32;                // k-steps of 32 per product
  __device__ static __forceinline__ int at(int row, int slot) { return row * SR + (slot ^ ((row >> SH) & (SR - 1))); }
  // bf8 lo image, byte offset of the lane quarter hq's values of a row: DP = 64: 16 B = {k-step 0, k-step 1} (one ds_read_b128 per
  // product; its 4 x 16 lane groups {0-3, 12-15, 20-27}, ... then cover the 64 banks: slot = hq ^ f(row / 4), f = 0, 3, 2, 1);
  // DP = 32: 8 B (one k-step), slot = hq ^ 2 (row / 8)
  __device__ static __forceinline__ int at8(int row, int hq) {
    if (DP == 64) {
      const int g = (row >> 2) & 3, f = (4 - g) & 3;
      return row * 64 + ((hq ^ f) << 4);
    }
    return row * 32 + ((hq ^ (((row >> 3) & 1) << 1)) << 3);
  }
};

// an empty volatile asm that consumes and redefines a register: volatile asms keep their order, so the instruction that produced the value
// stays ahead of it and its users stay behind it - the pipelined gate loop fixes its instruction order with these
#define RG_PIN(x) asm volatile("" : "+v"(x))

template <class Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int KST>
struct Frag {          // B operand of one node row: hi / mid / lo f16 and the bf8 form of x / 2^8, per k-step
  h8 h[KST], m[KST], l[KST];
  long q[KST];
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// PROBE: the test-hook build (rg_split3_product_check): writes one of the products instead of the new state
template <int NB, int ACT, bool PROBE>
__global__ __launch_bounds__(DENSE_T, 2) void dense_split3_kernel(DenseArgs A) {
  constexpr int DP = 16 * NB;
  using G = Geo<DP>;
  constexpr int SR = G::SR, KST = G::KST;
  constexpr int S = DP / 4;        // float4 chunks per padded row
  constexpr int KS = DP / 4;       // values per lane of a fragment
  constexpr int NW = DENSE_T / 64;
  constexpr int IMG = DP * SR;     // slots per weight image part
  // byte offsets inside LDS: f16 parts {hi, mid} of W_h, W_ih (3 gates), W_hh (3 gates), E (32 rows: Ws, W_final at row 16), then the bf8 parts
  constexpr uint32_t O_WH = 0, O_WIH = 2 * IMG * 16, O_WHH = O_WIH + 6 * IMG * 16, O_E = O_WHH + 6 * IMG * 16;
  constexpr uint32_t P_W = IMG * 16, P_G = 3 * IMG * 16, P_E = 32 * SR * 16;      // hi -> mid distance of W_h, of a gate stack, of E
  constexpr uint32_t O_F16_END = O_E + 2 * P_E;
  constexpr uint32_t L_WH = 0, L_WIH = IMG * 8, L_WHH = L_WIH + 3 * IMG * 8, L_E = L_WHH + 3 * IMG * 8;   // inside the bf8 region
  constexpr uint32_t O_LO_END = O_F16_END + L_E + 32 * SR * 8;
  extern __shared__ float4 lds[];
  if (A.n_dev) { A.n = *A.n_dev; A.n_tiles = (int)((A.n + 15) / 16); }
  if (A.n_tiles <= 0) return;
  char* const lds_b = reinterpret_cast<char*>(lds);
  float* bias_l = reinterpret_cast<float*>(lds_b + O_LO_END);                     // [4][DP], pre-multiplied by the exp2 factors of their gates
  uint32_t* wmax_bits = reinterpret_cast<uint32_t*>(bias_l + 4 * DP);

  const int d = A.d;
  // ---- weights -> split images under a power-of-two scale that puts their largest magnitude into [2^13, 2^14) (a first pass over the
  // 7 d^2 weights finds it: the splits are exact for every weight within 2^-14 of the largest).  The projections (Ws, W_final) get
  // a scale of their own, so that the new state does not depend on which of them a launch carries.
  const bool vec4 = (d & 3) == 0;
  {
    if (threadIdx.x < 2) wmax_bits[threadIdx.x] = 0u;
    __syncthreads();
    float wm = 0.0f;
    auto scan = [&](const float* src, int n) {
      if (!src) return;
      if (vec4 && ((uintptr_t)src & 15) == 0) {
        for (int i = threadIdx.x; i < n / 4; i += DENSE_T) {
          const float4 q = reinterpret_cast<const float4*>(src)[i];
          wm = fmaxf(fmaxf(wm, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
        }
      } else {
        for (int i = threadIdx.x; i < n; i += DENSE_T) wm = fmaxf(wm, fabsf(src[i]));
      }
    };
    auto publish = [&](int slot) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
      if ((threadIdx.x & 63) == 0) atomicMax(wmax_bits + slot, __float_as_uint(wm));      // non-negative floats order like their bit patterns
      wm = 0.0f;
    };
    scan(A.W_h, d * d); scan(A.w_ih, 3 * d * d); scan(A.w_hh, 3 * d * d);
    publish(0);
    scan(A.Ws, A.attn * d); scan(A.W_final, d);
    publish(1);
    __syncthreads();
  }
  const float sw_g = fit_weight_scale(__uint_as_float(wmax_bits[0])), sw_e = fit_weight_scale(__uint_as_float(wmax_bits[1]));
  auto load_w = [&](uint32_t o_hi, uint32_t o_mid, uint32_t o_lo, const float* src, int rows_src, int row0_dst, int rows_dst, float sw) {
    const bool v4 = vec4 && ((uintptr_t)src & 15) == 0;
    h8* hi_img = reinterpret_cast<h8*>(lds_b + o_hi);
    h8* mid_img = reinterpret_cast<h8*>(lds_b + o_mid);
    char* lo_img = lds_b + O_F16_END + o_lo;
    for (int i = threadIdx.x; i < rows_dst * S; i += DENSE_T) {
      const int r = i / S, ch = i - r * S;         // ch: 4-float chunk = (block ob, quarter hq)
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (src && r < rows_src) {
        if (v4) {
          if (ch * 4 < d) {
            const float4 q = *reinterpret_cast<const float4*>(src + (int64_t)r * d + ch * 4);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
          }
        } else {
          for (int k = 0; k < 4; ++k) {
            const int c = ch * 4 + k;
            v[k] = c < d ? src[(int64_t)r * d + c] : 0.f;
          }
        }
      }
      h4 hi, mid;
      uint32_t lo8;
      split3_4_lo8(v[0] * sw, v[1] * sw, v[2] * sw, v[3] * sw, hi, mid, lo8);
      const int ob = ch >> 2, hq = ch & 3, row = row0_dst + r;
      const int slot = 4 * (ob >> 1) + hq;
      reinterpret_cast<h4*>(hi_img + G::at(row, slot))[ob & 1] = hi;
      reinterpret_cast<h4*>(mid_img + G::at(row, slot))[ob & 1] = mid;
      reinterpret_cast<uint32_t*>(lo_img + G::at8(row, hq))[ob] = lo8;          // [k-step ob / 2][half ob % 2]
    }
  };
  load_w(O_WH, O_WH + P_W, L_WH, A.W_h, d, 0, DP, sw_g);
  for (int g = 0; g < 3; ++g) {
    load_w(O_WIH, O_WIH + P_G, L_WIH, A.w_ih + (int64_t)g * d * d, d, g * DP, DP, sw_g);
    load_w(O_WHH, O_WHH + P_G, L_WHH, A.w_hh + (int64_t)g * d * d, d, g * DP, DP, sw_g);
  }
  load_w(O_E, O_E + P_E, L_E, A.Ws, A.Ws ? A.attn : 0, 0, 16, sw_e);
  load_w(O_E, O_E + P_E, L_E, A.W_final, A.W_final ? 1 : 0, 16, 16, sw_e);
  const float inv_w = 1.0f / sw_g, inv_we = 1.0f / sw_e;                     // exact: powers of two
  for (int i = threadIdx.x; i < 4 * DP; i += DENSE_T) {
    const int g = i / DP, c = i - g * DP;
    float v = 0.f;
    if (c < d) {
      if (g == 0) v = -LOG2E * (A.b_ih[c] + A.b_hh[c]);
      else if (g == 1) v = -LOG2E * (A.b_ih[d + c] + A.b_hh[d + c]);
      else if (g == 2) v = -2.0f * LOG2E * A.b_ih[2 * d + c];
      else v = -2.0f * LOG2E * A.b_hh[2 * d + c];
    }
    bias_l[i] = v;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, hq = lane >> 4;

  // largest magnitude of the lane's node row (the row is spread over the four lane quarters)
  auto row_max = [&](const float (&f)[KS], float m) -> float {
#pragma unroll
    for (int i = 0; i < KS; ++i) m = fmaxf(m, fabsf(f[i]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    return m;
  };
  // B fragments of f * sc
  // (the bf8 form is taken from the SAME scaled values: the 2^-8 that pairs it with the weights' lo image, stored times 2^8, is applied
  // where the bf8 chain joins the f16 chain - one fma per accumulator element instead of a multiplication per operand element)
  auto split_frag = [&](const float (&f)[KS], float sc, Frag<KST>& F) {
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f[8 * s + j] * sc;
      h4 h0, m0, l0, h1, m1, l1;
      split3_4(v[0], v[1], v[2], v[3], h0, m0, l0);
      split3_4(v[4], v[5], v[6], v[7], h1, m1, l1);
      F.h[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      F.m[s] = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
      F.l[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
      const uint32_t q0 = to_bf8x4(v[0], v[1], v[2], v[3]);
      const uint32_t q1 = to_bf8x4(v[4], v[5], v[6], v[7]);
      F.q[s] = (long)(((uint64_t)q1 << 32) | q0);
    }
  };
  // the lane's A-fragment slot of k-step s in a 16-row block (block bases are multiples of 16 rows: the swizzles depend on li only).
  // Kept as byte offsets that the tile loop re-launders every iteration, so that each fragment read is one ds_read with the image /
  // block offset as its immediate (hoisted out of the loop, the addresses of a tile would each take a register).  A ds offset
  // reaches 64 KiB: the f16 images beyond it go through a second base, the bf8 images through a third.
  uint32_t a_off[KST], b_off[KST];
#pragma unroll
  for (int s = 0; s < KST; ++s) {
    a_off[s] = (uint32_t)G::at(li, 4 * s + hq) * 16u;
    b_off[s] = a_off[s] + 65536u;
  }
  uint32_t c_off = O_F16_END + (uint32_t)G::at8(li, hq);
  const char* wbase = lds_b;
  // acc += the 16-row block at row0 of the image {o_hi, o_mid, o_lo} times the fragment; smallest partial products first.
  // The bf8 products keep an accumulator chain of their own that a vector add joins to the f16 chain: with both MFMA forms chained
  // through ONE accumulator (hipcc 7.2, gfx950) the sums came out wrong by tens of percent - with or without eight wait states on
  // either side of the bf8 instruction, and right again as soon as a branch separated the instructions (tools/r3_variants.sh).
  // (acc8: the bf8 chain of the accumulator; products that add up in one accumulator - W_ih x and W_hh h of a gate - share it, and
  // `join` brings it in once)
  auto mma = [&](uint32_t o_hi, uint32_t o_mid, uint32_t o_lo, int row0, const Frag<KST>& F, f32x4& acc, f32x4& acc8) {
    o_hi += (uint32_t)row0 * SR * 16u;
    o_mid += (uint32_t)row0 * SR * 16u;
    o_lo += (uint32_t)row0 * SR * 8u;
    typedef long lk __attribute__((ext_vector_type(KST)));
    const lk wl = *reinterpret_cast<const lk*>(wbase + (c_off + o_lo));

#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const h8 wh = o_hi < 65536u ? *reinterpret_cast<const h8*>(wbase + (a_off[s] + o_hi))
                                  : *reinterpret_cast<const h8*>(wbase + (b_off[s] + (o_hi - 65536u)));
      const h8 wm_ = o_mid < 65536u ? *reinterpret_cast<const h8*>(wbase + (a_off[s] + o_mid))
                                    : *reinterpret_cast<const h8*>(wbase + (b_off[s] + (o_mid - 65536u)));
      acc8 = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wl[s], F.q[s], acc8, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, F.l[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm_, F.m[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm_, F.h[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, F.m[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, F.h[s], acc, 0, 0, 0);
    }
  };
  auto join = [&](f32x4& acc, const f32x4& acc8) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(acc8[r], LO8_INV, acc[r]);
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // tile I/O in fragment layout: lane (li, hq) owns columns 16 ob + 4 hq .. +3 of node li.  A wave-instruction then touches 16 rows
  // in 64-B pieces (each 128-B line is completed by the neighbouring block's instruction, issued next).  The tile index is
  // wave-uniform: row bases are scalar, a lane adds one 32-bit offset and the block's immediate.
  uint32_t lane_off = (uint32_t)(li * A.ld4 + hq);      // float4 units inside the tile's 16 rows
  uint32_t col_ok = 0;                                  // bit ob: the lane's chunk of block ob lies inside the row
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) col_ok |= (4 * ob + hq < A.ld4 ? 1u : 0u) << ob;
  // Every load of the tile loop is unconditional, from an address clamped into its buffer (rows beyond n read row n - 1, tiles beyond the
  // last read the last, column chunks beyond the row read chunk 0): rows that do not exist are never stored, chunk 0 again leaves the
  // row's maximum where it was and meets zero weight columns, and a node without an old state gets a zero row scale.  No load sits
  // behind a lane mask, so the loop body stays a handful of basic blocks and its memory waits stay counted ones.
  const int64_t n_last = A.n - 1;
  auto tile_row = [&](int t) -> int64_t {              // the lane's (clamped) row of tile t
    int ts = __builtin_amdgcn_readfirstlane(t);
    ts = ts < A.n_tiles ? ts : A.n_tiles - 1;
    const int64_t r = (int64_t)ts * 16 + li;
    return r < n_last ? r : n_last;
  };
  uint32_t col_off[NB];                                // float4 offset of the lane's chunk of block ob inside a row (clamped)
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) col_off[ob] = ((col_ok >> ob) & 1u) ? (uint32_t)(4 * ob + hq) : (uint32_t)(hq < A.ld4 ? hq : 0);
  // (the index as loaded: `prev_of` masks it with the row's existence where it is USED, a tile later - a select next to the load would
  // wait for it on the spot, and with it for every older load and store of the wave)
  auto load_prev = [&](int t) -> int { return A.prev_idx ? A.prev_idx[tile_row(t)] : -1; };
  auto prev_of = [&](int t, int raw) -> int {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    return (ts < A.n_tiles && (int64_t)ts * 16 + li < A.n) ? raw : -1;
  };
  auto load_agg = [&](int t, float4 (&va)[NB]) {
    const float4* arow = A.agg + tile_row(t) * A.ld4;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) va[ob] = arow[col_off[ob]];
  };
  // p < 0 (a new node, or no row): the lane reads its agg row once more (some address it may read: the old-state buffer may be empty)
  // and the row gets a zero scale
  auto load_old = [&](int t, int p, float4 (&vh)[NB]) {
    if (!A.prev_idx) return;
    const float4* hrow = p < 0 ? A.agg + tile_row(t) * A.ld4 : A.hprev + (int64_t)p * A.ld4;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) vh[ob] = hrow[col_off[ob]];
  };

  if (wv >= NW / 2) __builtin_amdgcn_s_sleep(64);     // de-phase the two waves of a SIMD (see dense.hip)
  float4 va[NB], vh[NB];
  const int t_step = gridDim.x * NW;
  int t = blockIdx.x * NW + wv;
  int p_cur = prev_of(t, load_prev(t));
  load_agg(t, va);
  int p_next = load_prev(t + t_step);      // raw
  load_old(t, p_cur, vh);
  for (; t < A.n_tiles; t += t_step) {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    const int64_t row0 = (int64_t)ts * 16;
#pragma unroll
    for (int s = 0; s < KST; ++s) { asm volatile("" : "+v"(a_off[s])); asm volatile("" : "+v"(b_off[s])); }
    asm volatile("" : "+v"(c_off));
    asm volatile("" : "+v"(lane_off));
    const bool node_ok = row0 + li < A.n;
    const bool any_old = __ballot(p_cur >= 0) != 0ull;

    // ---- operands of this tile out of the prefetch registers; the next tile's agg rows fly under this tile's work ---------------------
    float sc1, inv1;
    Frag<KST> F;
    {
      float fx[KS];
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) { fx[4 * ob] = va[ob].x; fx[4 * ob + 1] = va[ob].y; fx[4 * ob + 2] = va[ob].z; fx[4 * ob + 3] = va[ob].w; }
      row_scale(row_max(fx, 0.f), sc1, inv1);
      split_frag(fx, sc1, F);
    }
    load_agg(t + t_step, va);
    // (the readout's (query, entity) pair too: fetched at the end it would make the wave wait out its own prefetch)
    const int64_t node = row0 + li;
    int2 qe = make_int2(0, 0);
    if (A.W_final) qe = reinterpret_cast<const int2*>(A.nodes)[node < n_last ? node : n_last];

    // ---- stage 1: x = act(W_h agg) ------------------------------------------------------------------------------------------
    float xf[KS];
    {
      const float sc_out = inv1 * inv_w * (ACT == 2 ? -2.0f * LOG2E : 1.0f);
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        f32x4 acc = zero4, acc8 = zero4;
        mma(O_WH, O_WH + P_W, L_WH, 16 * ob, F, acc, acc8);
        join(acc, acc8);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[r] * sc_out;
          if (ACT == 1) v = fmaxf(v, 0.f);
          else if (ACT == 2) v = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v)), -1.0f);   // tanh = 2 sigmoid(2x) - 1
          xf[4 * ob + r] = v;
        }
      }
    }

    // (taken here, ahead of this tile's stores: a wait for the pair after them would wait for them as well)
    const int64_t score_at = (int64_t)qe.x * A.n_ent + qe.y;

    if (PROBE && A.probe == 1) {      // test hook: the stage-1 product itself
#pragma unroll
      for (int ob = 0; ob < NB; ++ob)
        if (node_ok && ((col_ok >> ob) & 1u))
          (A.hidden_out + row0 * A.ld4)[lane_off + 4 * ob] = make_float4(xf[4 * ob], xf[4 * ob + 1], xf[4 * ob + 2], xf[4 * ob + 3]);
      p_cur = prev_of(t + t_step, p_next);
      load_old(t + t_step, p_cur, vh);
      p_next = load_prev(t + 2 * t_step);
      continue;
    }

    // ---- GRU gates: x and the old state share one row scale, so that W_ih x and W_hh h add inside the accumulators; the new state
    // goes out in fragment layout as its blocks complete, and into the projection fragments ------------------------------------------
    float hn[KS];
    // new-state rows go out through a buffer descriptor of this tile's valid rows: lanes beyond them (and columns beyond the row) are
    // dropped by its range check instead of by a branch, so that the whole gate loop is ONE scheduling region
    const int rows_valid = (int)(A.n - row0 < 16 ? (A.n - row0 > 0 ? A.n - row0 : 0) : 16);
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(A.hidden_out + row0 * A.ld4), 0, rows_valid * A.ld4 * 16, 0x00020000);
    auto gru = [&](auto has_old) {
      constexpr bool OLD = decltype(has_old)::value;
      Frag<KST> H;
      float sc, inv;
      if constexpr (OLD) {
        float hf[KS];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) { hf[4 * ob] = vh[ob].x; hf[4 * ob + 1] = vh[ob].y; hf[4 * ob + 2] = vh[ob].z; hf[4 * ob + 3] = vh[ob].w; }
        const bool is_old = p_cur >= 0;
        row_scale(fmaxf(is_old ? row_max(hf, 0.f) : 0.f, row_max(xf, 0.f)), sc, inv);
        split_frag(hf, is_old ? sc : 0.f, H);
      } else {
        row_scale(row_max(xf, 0.f), sc, inv);
      }
      // the next tile's old rows go out here, AHEAD of this tile's stores: the memory counter retires in order, so a wait for loads
      // issued after the stores (as a prefetch at the end of the tile was) waits out the stores' acknowledgements and the loads' whole
      // latency - with that order the loop ran at ~7 us per tile whatever it computed
      p_cur = prev_of(t + t_step, p_next);
      load_old(t + t_step, p_cur, vh);
      p_next = load_prev(t + 2 * t_step);
      split_frag(xf, sc, F);
      const float inv_s = inv * inv_w * -LOG2E, inv_t = inv * inv_w * (-2.0f * LOG2E);
      // The gate loop is software-pipelined by one block: the matrix products of block ob + 1 are issued together with the gate
      // arithmetic of block ob, and the scheduler is told to alternate them (one MFMA, then vector instructions): a lone vector
      // instruction costs ~4.5 cycles of the SIMD, but the two waves of a SIMD hide two per MFMA when both streams are finely mixed
      // (tools/hipcheck/mfma_valu_overlap.hip); long MFMA runs followed by long vector runs did not overlap at all.
      auto products = [&](int ob, f32x4& ar, f32x4& az, f32x4& ai, f32x4& ag) {
        ar = zero4; az = zero4; ai = zero4; ag = zero4;
        {
          f32x4 r8 = zero4;
          mma(O_WIH, O_WIH + P_G, L_WIH, 0 * DP + 16 * ob, F, ar, r8);
          if constexpr (OLD) mma(O_WHH, O_WHH + P_G, L_WHH, 0 * DP + 16 * ob, H, ar, r8);
          join(ar, r8);
        }
        {
          f32x4 z8 = zero4;
          mma(O_WIH, O_WIH + P_G, L_WIH, 1 * DP + 16 * ob, F, az, z8);
          if constexpr (OLD) mma(O_WHH, O_WHH + P_G, L_WHH, 1 * DP + 16 * ob, H, az, z8);
          join(az, z8);
        }
        {
          f32x4 i8 = zero4;
          mma(O_WIH, O_WIH + P_G, L_WIH, 2 * DP + 16 * ob, F, ai, i8);
          join(ai, i8);
        }
        if constexpr (OLD) {
          f32x4 g8 = zero4;
          mma(O_WHH, O_WHH + P_G, L_WHH, 2 * DP + 16 * ob, H, ag, g8);
          join(ag, g8);
        }
      };
      auto gates = [&](int ob, const f32x4& ar, const f32x4& az, const f32x4& ai, const f32x4& ag) {
        const float4 br = *reinterpret_cast<const float4*>(bias_l + 0 * DP + 16 * ob + 4 * hq);
        const float4 bz = *reinterpret_cast<const float4*>(bias_l + 1 * DP + 16 * ob + 4 * hq);
        const float4 bi = *reinterpret_cast<const float4*>(bias_l + 2 * DP + 16 * ob + 4 * hq);
        const float4 bh = *reinterpret_cast<const float4*>(bias_l + 3 * DP + 16 * ob + 4 * hq);
        // the old state of this block's four columns, back from its split: (mid + lo) is the first residual and hi + that the scaled
        // value, both exact
        float hos[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (OLD) {
          const int s = ob >> 1, j0 = 4 * (ob & 1);
          const h2 hh0 = {H.h[s][j0], H.h[s][j0 + 1]}, hh1 = {H.h[s][j0 + 2], H.h[s][j0 + 3]};
          const h2 hm0 = {H.m[s][j0], H.m[s][j0 + 1]}, hm1 = {H.m[s][j0 + 2], H.m[s][j0 + 3]};
          const h2 hl0 = {H.l[s][j0], H.l[s][j0 + 1]}, hl1 = {H.l[s][j0 + 2], H.l[s][j0 + 3]};
          hos[0] = add_hf_lo(hh0, add_hh_lo(hm0, hl0));
          hos[1] = add_hf_hi(hh0, add_hh_hi(hm0, hl0));
          hos[2] = add_hf_lo(hh1, add_hh_lo(hm1, hl1));
          hos[3] = add_hf_hi(hh1, add_hh_hi(hm1, hl1));
        }
        const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w};
        const float biv[4] = {bi.x, bi.y, bi.z, bi.w}, bhv[4] = {bh.x, bh.y, bh.z, bh.w};
        float hnv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // sigmoid(a) = 1 / (1 + 2^(-log2e a)),  tanh(a) = 2 / (1 + 2^(-2 log2e a)) - 1: the factors sit in the scales and biases
          const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(ar[r], inv_s, brv[r])));
          const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(az[r], inv_s, bzv[r])));
          const float ti = fmaf(ai[r], inv_t, biv[r]);
          const float th = OLD ? fmaf(ag[r], inv_t, bhv[r]) : bhv[r];
          const float ng = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(rg, th, ti))), -1.0f);
          hnv[r] = OLD ? fmaf(zg, fmaf(hos[r], inv, -ng), ng) : fmaf(zg, -ng, ng);       // (1 - z) n + z h
        }
        if constexpr (PROBE) {    // test hook: the gate products themselves (W_in x / W_hn h)
          if (A.probe >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) hnv[r] = (A.probe == 2 ? ai[r] : ag[r]) * (inv * inv_w);
          }
        }
        const u32x4 bits = {__float_as_uint(hnv[0]), __float_as_uint(hnv[1]), __float_as_uint(hnv[2]), __float_as_uint(hnv[3])};
        const uint32_t voff = ((col_ok >> ob) & 1u) ? (lane_off + 4u * ob) * 16u : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_buffer_store_b128(bits, r_out, voff, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) hn[4 * ob + r] = hnv[r];
      };
      f32x4 ar, az, ai, ag;
      __builtin_amdgcn_sched_barrier(0);
      products(0, ar, az, ai, ag);
      if constexpr (PROBE) {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
          gates(ob, ar, az, ai, ag);
          if (ob + 1 < NB) products(ob + 1, ar, az, ai, ag);
        }
      } else {
        // ---- the pipelined loop, written out at instruction granularity: unit u = one MFMA of block ob + 1's products followed by
        // at most two vector instructions of block ob's gate arithmetic, pinned by a scheduling barrier (the scheduler left to itself -
        // also under sched_group_barrier patterns - kept the MFMAs and the transcendentals in separate runs)
        constexpr int NCALL = OLD ? 6 : 3;                 // products of a block: {W_ih, W_hh} x {r, z}, W_in x, W_hn h
        constexpr int NU = NCALL * KST * 6;
        constexpr int OPS_R = OLD ? 19 : 15;               // gate operations per element
        constexpr int NOPS = 4 + 4 * OPS_R + 1;            // bias reads, 4 elements, the store
        constexpr int OPU = (NOPS + NU - 1) / NU;          // most vector operations behind one MFMA (the list is spread evenly over the block's MFMAs)
        static_for<NB>([&](auto OB) {
          constexpr int ob = decltype(OB)::value;
          constexpr bool more = ob + 1 < NB;
          f32x4 nacc[4] = {zero4, zero4, zero4, zero4}, nacc8[4] = {zero4, zero4, zero4, zero4};
          // gate state of block ob
          float4 bia[4];
          float t1[4], t2[4], e1[4], e2[4], rg[4], zg[4], ti[4], th[4], uu[4], e3[4], hm[4], hs[4], ng[4], hnv[4];
          auto gate_op = [&](auto K) {
            constexpr int k = decltype(K)::value;
            if constexpr (k < 4) {
              bia[k] = *reinterpret_cast<const float4*>(bias_l + k * DP + 16 * ob + 4 * hq);
              RG_PIN(bia[k].x);
            } else if constexpr (k < 4 + 4 * OPS_R) {
              constexpr int j = (k - 4) / 4, r = (k - 4) % 4;          // operation j of element r (operation-major: four independent chains)
              const float bvr = r == 0 ? bia[0].x : r == 1 ? bia[0].y : r == 2 ? bia[0].z : bia[0].w;
              const float bvz = r == 0 ? bia[1].x : r == 1 ? bia[1].y : r == 2 ? bia[1].z : bia[1].w;
              const float bvi = r == 0 ? bia[2].x : r == 1 ? bia[2].y : r == 2 ? bia[2].z : bia[2].w;
              const float bvh = r == 0 ? bia[3].x : r == 1 ? bia[3].y : r == 2 ? bia[3].z : bia[3].w;
              constexpr int sH = ob >> 1, jH = 4 * (ob & 1) + (r & ~1);
              if constexpr (OLD) {
                if constexpr (j == 0) { t1[r] = fmaf(ar[r], inv_s, bvr); RG_PIN(t1[r]); }
                else if constexpr (j == 1) { e1[r] = __builtin_amdgcn_exp2f(t1[r]); RG_PIN(e1[r]); }
                else if constexpr (j == 2) { t2[r] = fmaf(az[r], inv_s, bvz); RG_PIN(t2[r]); }
                else if constexpr (j == 3) { e2[r] = __builtin_amdgcn_exp2f(t2[r]); RG_PIN(e2[r]); }
                else if constexpr (j == 4) { t1[r] = 1.0f + e1[r]; RG_PIN(t1[r]); }
                else if constexpr (j == 5) { t2[r] = 1.0f + e2[r]; RG_PIN(t2[r]); }
                else if constexpr (j == 6) { rg[r] = __builtin_amdgcn_rcpf(t1[r]); RG_PIN(rg[r]); }
                else if constexpr (j == 7) { zg[r] = __builtin_amdgcn_rcpf(t2[r]); RG_PIN(zg[r]); }
                else if constexpr (j == 8) { ti[r] = fmaf(ai[r], inv_t, bvi); RG_PIN(ti[r]); }
                else if constexpr (j == 9) { th[r] = fmaf(ag[r], inv_t, bvh); RG_PIN(th[r]); }
                else if constexpr (j == 10) { uu[r] = fmaf(rg[r], th[r], ti[r]); RG_PIN(uu[r]); }
                else if constexpr (j == 11) { e3[r] = __builtin_amdgcn_exp2f(uu[r]); RG_PIN(e3[r]); }
                else if constexpr (j == 12) {
                  const h2 m2 = {H.m[sH][jH], H.m[sH][jH + 1]}, l2 = {H.l[sH][jH], H.l[sH][jH + 1]};
                  { hm[r] = (r & 1) ? add_hh_hi(m2, l2) : add_hh_lo(m2, l2); RG_PIN(hm[r]); }
                } else if constexpr (j == 13) { uu[r] = 1.0f + e3[r]; RG_PIN(uu[r]); }
                else if constexpr (j == 14) {
                  const h2 h2_ = {H.h[sH][jH], H.h[sH][jH + 1]};
                  { hs[r] = (r & 1) ? add_hf_hi(h2_, hm[r]) : add_hf_lo(h2_, hm[r]); RG_PIN(hs[r]); }
                } else if constexpr (j == 15) { e3[r] = __builtin_amdgcn_rcpf(uu[r]); RG_PIN(e3[r]); }
                else if constexpr (j == 16) { ng[r] = fmaf(2.0f, e3[r], -1.0f); RG_PIN(ng[r]); }
                else if constexpr (j == 17) { hs[r] = fmaf(hs[r], inv, -ng[r]); RG_PIN(hs[r]); }
                else { hnv[r] = fmaf(zg[r], hs[r], ng[r]); RG_PIN(hnv[r]); }                       // (1 - z) n + z h
              } else {
                if constexpr (j == 0) { t1[r] = fmaf(ar[r], inv_s, bvr); RG_PIN(t1[r]); }
                else if constexpr (j == 1) { e1[r] = __builtin_amdgcn_exp2f(t1[r]); RG_PIN(e1[r]); }
                else if constexpr (j == 2) { t2[r] = fmaf(az[r], inv_s, bvz); RG_PIN(t2[r]); }
                else if constexpr (j == 3) { e2[r] = __builtin_amdgcn_exp2f(t2[r]); RG_PIN(e2[r]); }
                else if constexpr (j == 4) { t1[r] = 1.0f + e1[r]; RG_PIN(t1[r]); }
                else if constexpr (j == 5) { t2[r] = 1.0f + e2[r]; RG_PIN(t2[r]); }
                else if constexpr (j == 6) { rg[r] = __builtin_amdgcn_rcpf(t1[r]); RG_PIN(rg[r]); }
                else if constexpr (j == 7) { zg[r] = __builtin_amdgcn_rcpf(t2[r]); RG_PIN(zg[r]); }
                else if constexpr (j == 8) { ti[r] = fmaf(ai[r], inv_t, bvi); RG_PIN(ti[r]); }
                else if constexpr (j == 9) { uu[r] = fmaf(rg[r], bvh, ti[r]); RG_PIN(uu[r]); }
                else if constexpr (j == 10) { e3[r] = __builtin_amdgcn_exp2f(uu[r]); RG_PIN(e3[r]); }
                else if constexpr (j == 11) { uu[r] = 1.0f + e3[r]; RG_PIN(uu[r]); }
                else if constexpr (j == 12) { e3[r] = __builtin_amdgcn_rcpf(uu[r]); RG_PIN(e3[r]); }
                else if constexpr (j == 13) { ng[r] = fmaf(2.0f, e3[r], -1.0f); RG_PIN(ng[r]); }
                else { hnv[r] = fmaf(zg[r], -ng[r], ng[r]); RG_PIN(hnv[r]); }
              }
            } else {
              const u32x4 bits = {__float_as_uint(hnv[0]), __float_as_uint(hnv[1]), __float_as_uint(hnv[2]), __float_as_uint(hnv[3])};
              const uint32_t voff = ((col_ok >> ob) & 1u) ? (lane_off + 4u * ob) * 16u : 0xFFFFFFF0u;
              __builtin_amdgcn_raw_buffer_store_b128(bits, r_out, voff, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r) hn[4 * ob + r] = hnv[r];
            }
          };
          if constexpr (!more) {
            static_for<NOPS>(gate_op);
          } else {
            typedef long lk __attribute__((ext_vector_type(KST)));
            h8 wh, wm_, nwh, nwm;
            lk wl, nwl;
            // fragment reads of product c, k-step s of block ob + 1
            auto fetch = [&](auto C, auto S_, h8& fh, h8& fm, lk& fl) {
              constexpr int c = decltype(C)::value, ks = decltype(S_)::value;
              constexpr bool isH = OLD && (c & 1);
              constexpr int gate = OLD ? c / 2 : c;
              constexpr uint32_t row_b = (uint32_t)(gate * DP + 16 * (ob + 1)) * SR;
              constexpr uint32_t o_hi = (isH ? O_WHH : O_WIH) + row_b * 16u, o_mid = o_hi + P_G, o_lo = (isH ? L_WHH : L_WIH) + row_b * 8u;
              fh = o_hi < 65536u ? *reinterpret_cast<const h8*>(wbase + (a_off[ks] + o_hi)) : *reinterpret_cast<const h8*>(wbase + (b_off[ks] + (o_hi - 65536u)));
              fm = o_mid < 65536u ? *reinterpret_cast<const h8*>(wbase + (a_off[ks] + o_mid)) : *reinterpret_cast<const h8*>(wbase + (b_off[ks] + (o_mid - 65536u)));
              if constexpr (ks == 0) fl = *reinterpret_cast<const lk*>(wbase + (c_off + o_lo));
            };
            fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, nwh, nwm, nwl);
            static_for<NU>([&](auto U) {
              constexpr int u = decltype(U)::value;
              constexpr int c = u / (6 * KST), ks = (u / 6) % KST, term = u % 6;
              constexpr bool isH = OLD && (c & 1);
              constexpr int gate = OLD ? c / 2 : c;
              constexpr int ai_ = (isH && gate == 2) ? 3 : gate;
              const Frag<KST>& X = isH ? H : F;
              if constexpr (term == 0) {
                wh = nwh; wm_ = nwm;
                if constexpr (ks == 0) wl = nwl;
                constexpr int nk = u / 6 + 1;                  // the next k-step's fragments go out now
                if constexpr (nk < NCALL * KST) fetch(std::integral_constant<int, nk / KST>{}, std::integral_constant<int, nk % KST>{}, nwh, nwm, nwl);
                nacc8[ai_] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wl[ks], X.q[ks], nacc8[ai_], 0, 0, 0);
              } else if constexpr (term == 1) nacc[ai_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.l[ks], nacc[ai_], 0, 0, 0);
              else if constexpr (term == 2) nacc[ai_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm_, X.m[ks], nacc[ai_], 0, 0, 0);
              else if constexpr (term == 3) nacc[ai_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm_, X.h[ks], nacc[ai_], 0, 0, 0);
              else if constexpr (term == 4) nacc[ai_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.m[ks], nacc[ai_], 0, 0, 0);
              else nacc[ai_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, X.h[ks], nacc[ai_], 0, 0, 0);
              if constexpr (term == 0) RG_PIN(nacc8[ai_]); else RG_PIN(nacc[ai_]);
              static_for<OPU>([&](auto Q) {
                constexpr int k_lo = (u * NOPS + NU - 1) / NU, k_hi = ((u + 1) * NOPS + NU - 1) / NU, k = k_lo + decltype(Q)::value;
                if constexpr (k < k_hi) gate_op(std::integral_constant<int, k>{});
              });
              __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int g = 0; g < 4; ++g) join(nacc[g], nacc8[g]);
            ar = nacc[0]; az = nacc[1]; ai = nacc[2]; ag = nacc[3];
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (any_old) gru(std::true_type{}); else gru(std::false_type{});

    // ---- projections of the new state (at scale 2^14: |h| <= 1 in the model; any finite state below 2 is carried) -----------------
    constexpr float inv_n = 1.0f / 16384.0f;
    if (A.Ws || A.W_final) split_frag(hn, 16384.0f, F);
    if (A.Ws) {       // rows e = 4*hq + r of block 0
      f32x4 acc = zero4, acc8 = zero4;
      mma(O_E, O_E + P_E, L_E, 0, F, acc, acc8);
      join(acc, acc8);
      const float sc_e = inv_we * inv_n;
      const __amdgpu_buffer_rsrc_t r_as = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(A.a_s_out + row0 * A.ap), 0,
                                                                            rows_valid * A.ap * 4, 0x00020000);
      const u32x4 bits = {__float_as_uint(acc[0] * sc_e), __float_as_uint(acc[1] * sc_e), __float_as_uint(acc[2] * sc_e), __float_as_uint(acc[3] * sc_e)};
      __builtin_amdgcn_raw_buffer_store_b128(bits, r_as, 4 * hq < A.ap ? (uint32_t)(li * A.ap + 4 * hq) * 4u : 0xFFFFFFF0u, 0, 0);
    }
    if (A.W_final) {  // row 16 = register 0 of quarter 0 of block 1
      f32x4 acc = zero4, acc8 = zero4;
      mma(O_E, O_E + P_E, L_E, 16, F, acc, acc8);
      join(acc, acc8);
      if (node_ok && hq == 0) A.scores[score_at] = acc[0] * (inv_we * inv_n);
    }
  }
}

template <int NB>
constexpr size_t lds_bytes() {
  constexpr int DP = 16 * NB, SR = DP / 8;
  return (size_t)(14 * DP * SR + 2 * 32 * SR) * 16 + (size_t)(7 * DP * SR + 32 * SR) * 8 + 4 * DP * sizeof(float) + 16;   // + the two maxima
}

template <int NB, int ACT, bool PROBE>
int launch(const DenseArgs& A, hipStream_t s) {
  constexpr int NW = DENSE_T / 64;
  constexpr size_t lds = lds_bytes<NB>();
  static_assert(lds <= 160 * 1024, "weight images exceed the CU's LDS");
  RG_HIP(hipFuncSetAttribute((const void*)dense_split3_kernel<NB, ACT, PROBE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, rg::ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL((dense_split3_kernel<NB, ACT, PROBE>), dim3(grid), dim3(DENSE_T), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int NB>
int launch_act(const DenseArgs& A, hipStream_t s) {
  if (A.probe) return launch<NB, 0, true>(A, s);        // (the hook runs with the identity activation)
  return A.act == 0 ? launch<NB, 0, false>(A, s) : A.act == 1 ? launch<NB, 1, false>(A, s) : launch<NB, 2, false>(A, s);
}

// rg_split3_roundtrip: the device split of n rows of `cols` floats (row scale as the kernels take it) and its reconstruction
__global__ void split3_roundtrip_kernel(const float* __restrict__ x, int64_t n, int cols, float* __restrict__ back, float* __restrict__ parts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.y + threadIdx.y;
  if (r >= n) return;
  float m = 0.f;
  for (int c = threadIdx.x; c < cols; c += 64) m = fmaxf(m, fabsf(x[r * cols + c]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sc, inv;
  row_scale(m, sc, inv);
  for (int c = 4 * threadIdx.x; c < cols; c += 256) {
    float v[4];
    for (int k = 0; k < 4; ++k) v[k] = c + k < cols ? x[r * cols + c + k] : 0.f;
    h4 hi, mid, lo;
    split3_4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc, hi, mid, lo);
    h4 whi, wmid;
    uint32_t lo8;
    split3_4_lo8(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc, whi, wmid, lo8);
    for (int k = 0; k < 4 && c + k < cols; ++k) {
      back[r * cols + c + k] = ((float)hi[k] + ((float)mid[k] + (float)lo[k])) * inv;
      if (parts) {
        float* p = parts + (r * cols + c + k) * 4;
        p[0] = (float)hi[k]; p[1] = (float)mid[k]; p[2] = (float)lo[k];
        // the bf8 byte of lo * 2^8, decoded: sign, 5 exponent bits (bias 15), 2 mantissa bits
        const uint32_t b = (lo8 >> (8 * k)) & 0xffu, e = (b >> 2) & 31u, f = b & 3u;
        const float mag = e ? ldexpf(1.0f + 0.25f * f, (int)e - 15) : ldexpf(0.25f * f, -14);
        p[3] = (b & 0x80u ? -mag : mag) * LO8_INV;
      }
    }
  }
}

}  // namespace

namespace rg {
int dense_split3_launch(const DenseArgs& A, hipStream_t s) { return A.d <= 32 ? launch_act<2>(A, s) : launch_act<4>(A, s); }
}  // namespace rg

extern "C" int rg_split3_product_check(int32_t which, int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                                       const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih,
                                       const float* b_hh, float* out, void* stream) {
  RG_CHECK(which >= 1 && which <= 3 && agg && W_h && w_ih && w_hh && b_ih && b_hh && out, "rg_split3_product_check: bad argument");
  RG_CHECK(d >= 4 && d <= 64 && d % 4 == 0 && act >= 0 && act <= 2, "rg_split3_product_check: d=%d act=%d", d, act);
  RG_CHECK(which != 3 || (hidden_prev && prev_idx), "rg_split3_product_check: which = 3 needs the old state");
  if (n == 0) return 0;
  DenseArgs A;
  A.n = n; A.n_dev = nullptr; A.d = d; A.ld4 = d / 4;
  A.agg = (const float4*)agg; A.hprev = (const float4*)hidden_prev; A.prev_idx = prev_idx;
  A.W_h = W_h; A.w_ih = w_ih; A.w_hh = w_hh; A.b_ih = b_ih; A.b_hh = b_hh;
  A.Ws = nullptr; A.attn = 0; A.ap = 0; A.a_s_out = nullptr; A.W_final = nullptr; A.nodes = nullptr; A.n_ent = 0; A.scores = nullptr;
  A.hidden_out = (float4*)out; A.act = act; A.n_tiles = (int)rg::ceil_div(n, 16);
  A.probe = which;
  return rg::dense_split3_launch(A, (hipStream_t)stream);
}

extern "C" int rg_split3_roundtrip(const float* x, int64_t n_rows, int32_t cols, float* back, float* parts, void* stream) {
  RG_CHECK(x && back && n_rows >= 0 && cols >= 1, "rg_split3_roundtrip: bad argument");
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(split3_roundtrip_kernel, dim3((unsigned)rg::ceil_div(n_rows, 4)), dim3(64, 4), 0, (hipStream_t)stream, x, n_rows, cols, back, parts);
  RG_LAUNCH_CHECK();
  return 0;
}
