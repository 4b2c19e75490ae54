// Fused dense epilogue of one layer with the products on the f16 matrix pipe as two-term splits (inference path, d <= 64).
// Same operator as dense.hip (Static/transductive/models.py :41 W_h + act, :81-84 the gathered single-step GRU, the next layer's :36
// Ws_attn(hs), :86-88 the readout), same transposed data flow (node = lane, accumulators feed the next product), but every fp32
// operand v is carried as   v = s * (hi + lo),   hi = fp16(v / s),  lo = fp16(v / s - hi)
// with s a power of two - per node row for the activations (the row's largest magnitude goes to [2^14, 2^15)), one for all the
// weights of the launch (their largest magnitude to at most 2^15) - so that hi + lo carries 22 significant bits of the row's /
// the weights' largest entries and lo stays a normal f16.  A product W X is three v_mfma_f32_16x16x32_f16 into one fp32 accumulator:
//   acc += W_hi X_hi + W_hi X_lo + W_lo X_hi            (the dropped W_lo X_lo is 2^-22 of a product)
// and W X = s_w^-1 s_x^-1 acc.  A 64-term dot product comes out within a few 1e-7 of its largest terms (fp32 FMA chains: about 1e-7);
// measured against fp64 the layer's outputs are as close as the exact kernel's (tools/probe_dense_precision.py: both are set by the
// v_exp / v_rcp forms of sigmoid and tanh), at 3/16 of the matrix-pipe time of v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md: the f32
// forms run at 1/16 of the f16 rate).  With the matrix phase that short the kernel is bound by instruction issue and its HBM rows,
// so the rest is written for few instructions: operand rows are loaded and stored directly in fragment layout (no transposition
// through LDS, no wave barriers), row bases are scalar, sigmoid / tanh take their -log2(e) factors from the pre-scaled biases and
// scales, and one wave keeps the next tile's rows and the prev_idx word of the tile after in flight.
// rg_dense_fwd(..., precision = 1) selects it; precision = 0 keeps the exact-fp32 kernel of dense.hip.
#include <type_traits>
#include "dense_common.h"

using namespace rg;

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int DENSE_T = 512;
constexpr float LOG2E = 1.44269504088896340736f;

// hi / lo halves of four (already scaled) floats
// (the residual x - float(hi) as one v_fma_mix_f32 per value - hi read as f16 in place - instead of a conversion and a subtraction:
// the compiler does not form it, and the splits are a third of the kernel's vector instructions)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
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
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(float a, float b, float c, float d, h4& hi, h4& lo) {
  const f2v x0 = {a, b}, x1 = {c, d};
  const h2 h0 = __builtin_convertvector(x0, h2), h1 = __builtin_convertvector(x1, h2);
  const f2v r0 = {resid_lo(h0, a), resid_hi(h0, b)}, r1 = {resid_lo(h1, c), resid_hi(h1, d)};
  const h2 l0 = __builtin_convertvector(r0, h2), l1 = __builtin_convertvector(r1, h2);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
}

// power-of-two scale that puts m (>= 0) into [2^14, 2^15), and its inverse; rows of (near) zeros keep a finite scale
__device__ __forceinline__ void row_scale(float m, float& sc, float& inv) {
  uint32_t eb = (__float_as_uint(m) >> 23) & 0xffu;
  eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
  sc = __uint_as_float((268u - eb) << 23);
  inv = __uint_as_float((eb - 14u) << 23);
}

// DP in {32, 64}: padded width.  Weight images in LDS: per row DP f16 of hi and, in a second image, of lo; the 16-B slot (k-step s,
// lane quarter hq) holds the row's weights for k = 16 * (2s + j / 4) + 4 * hq + j % 4, j = 0..7 - the k order in which a lane holds
// its accumulator rows, so that accumulators convert in place into the next B fragment.  Slots are XOR-swizzled with the row so that
// the 16 rows read by a quarter wave cover the 16 bank groups.
template <int DP>
struct Geo {
  static constexpr int SR = DP / 8;                  // 16-B slots per image row
  static constexpr int SH = DP == 64 ? 1 : 2;        // rows per 256 B of LDS
  static constexpr int KST = DP / 32;                // k-steps of 32 per product
  __device__ static __forceinline__ int at(int row, int slot) { return row * SR + (slot ^ ((row >> SH) & (SR - 1))); }
};

template <int NB, int ACT>
__global__ __launch_bounds__(DENSE_T, 2) void dense_split_kernel(DenseArgs A) {
  constexpr int DP = 16 * NB;
  using G = Geo<DP>;
  constexpr int SR = G::SR, KST = G::KST;
  constexpr int S = DP / 4;        // float4 chunks per padded row
  constexpr int KS = DP / 4;       // values per lane of a fragment
  constexpr int NW = DENSE_T / 64;
  constexpr int IMG = DP * SR;     // 16-B slots per weight image part
  extern __shared__ float4 lds[];
  if (A.n_dev) { A.n = *A.n_dev; A.n_tiles = (int)((A.n + 15) / 16); }
  h8* Wh_hi = reinterpret_cast<h8*>(lds);      // images: W_h, W_ih (3 gates), W_hh (3 gates), E (32 rows: Ws, W_final at row 16)
  h8* Wh_lo = Wh_hi + IMG;
  h8* Wih_hi = Wh_lo + IMG;
  h8* Wih_lo = Wih_hi + 3 * IMG;
  h8* Whh_hi = Wih_lo + 3 * IMG;
  h8* Whh_lo = Whh_hi + 3 * IMG;
  h8* E_hi = Whh_lo + 3 * IMG;
  h8* E_lo = E_hi + 32 * SR;
  float* bias_l = reinterpret_cast<float*>(E_lo + 32 * SR);     // [4][DP], pre-multiplied by the exp2 factors of their gates
  float4* stash = reinterpret_cast<float4*>(bias_l + 4 * DP);   // [NW][NB][64]: each lane's old-state chunks
  uint32_t* wmax_bits = reinterpret_cast<uint32_t*>(stash + NW * NB * 64);

  const int d = A.d;
  // ---- weights -> split f16 images, all under one power-of-two scale ----------------------------------------------------------
  // 2^12 suits weights of magnitude 2^-8 .. 8 (initialised or trained layers); the staging pass finds the true maximum on its way
  // and is repeated with a fitted scale when that guess would overflow f16 or leave the lo halves denormal.
  const bool vec4 = (d & 3) == 0;
  float sw = 4096.0f, wm = 0.0f;
  auto load_w = [&](h8* hi_img, h8* lo_img, const float* src, int rows_src, int row0_dst, int rows_dst) {
    const bool v4 = vec4 && ((uintptr_t)src & 15) == 0;
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
      wm = fmaxf(fmaxf(wm, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      h4 hi, lo;
      split4(v[0] * sw, v[1] * sw, v[2] * sw, v[3] * sw, hi, lo);
      const int ob = ch >> 2, hq = ch & 3, row = row0_dst + r;
      const int slot = G::at(row, 4 * (ob >> 1) + hq);
      reinterpret_cast<h4*>(hi_img + slot)[ob & 1] = hi;
      reinterpret_cast<h4*>(lo_img + slot)[ob & 1] = lo;
    }
  };
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (threadIdx.x == 0) *wmax_bits = 0u;
    __syncthreads();
    wm = 0.0f;
    load_w(Wh_hi, Wh_lo, A.W_h, d, 0, DP);
    for (int g = 0; g < 3; ++g) {
      load_w(Wih_hi, Wih_lo, A.w_ih + (int64_t)g * d * d, d, g * DP, DP);
      load_w(Whh_hi, Whh_lo, A.w_hh + (int64_t)g * d * d, d, g * DP, DP);
    }
    load_w(E_hi, E_lo, A.Ws, A.Ws ? A.attn : 0, 0, 16);
    load_w(E_hi, E_lo, A.W_final, A.W_final ? 1 : 0, 16, 16);
    atomicMax(wmax_bits, __float_as_uint(wm));      // non-negative floats order like their bit patterns
    __syncthreads();
    const float wmax = __uint_as_float(*wmax_bits);
    if (wmax == 0.0f || (wmax * sw <= 32768.0f && wmax * sw >= 16.0f)) break;
    uint32_t eb = (__float_as_uint(wmax) >> 23) & 0xffu;
    eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
    sw = __uint_as_float((267u - eb) << 23);        // largest magnitude to [2^13, 2^14)
    __syncthreads();
  }
  const float inv_w = 1.0f / sw;                     // exact: a power of two
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
  float4* my_stash = stash + wv * NB * 64 + lane;

  // largest magnitude of the lane's node row (the row is spread over the four lane quarters)
  auto row_max = [&](const float (&f)[KS], float m) -> float {
#pragma unroll
    for (int i = 0; i < KS; ++i) m = fmaxf(m, fabsf(f[i]));
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    return m;
  };
  // B fragments of f * sc
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
  // the lane's A-fragment slot of k-step s in a 16-row block (block bases are multiples of 16 rows: the swizzle depends on li only).
  // Kept as byte offsets that the tile loop re-launders every iteration, so that each fragment read is one ds_read_b128 with the
  // image / block offset as its immediate (hoisted out of the loop, the 116 addresses of a tile would each take a register).
  uint32_t a_off[KST], b_off[KST];
#pragma unroll
  for (int s = 0; s < KST; ++s) { a_off[s] = (uint32_t)G::at(li, 4 * s + hq) * 16u; b_off[s] = a_off[s] + 65536u; }
  const char* wbase = reinterpret_cast<const char*>(lds);
  // acc += the 16-row block at row0 of a weight image times the fragment
  // byte offsets of the image parts from the start of LDS
  constexpr uint32_t O_WH = 0, O_WIH = 2 * IMG * 16, O_WHH = O_WIH + 6 * IMG * 16, O_E = O_WHH + 6 * IMG * 16;
  constexpr uint32_t P_W = IMG * 16, P_G = 3 * IMG * 16, P_E = 32 * SR * 16;      // hi -> lo distance of W_h, of a gate stack, of E
  // acc += the 16-row block at row0 of the image at byte offset o_hi (its lo part at o_lo) times the fragment
  auto mma = [&](uint32_t o_hi, uint32_t o_lo, int row0, const h8 (&fh)[KST], const h8 (&fl)[KST], f32x4& acc) {
    o_hi += (uint32_t)row0 * SR * 16u;
    o_lo += (uint32_t)row0 * SR * 16u;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      // (a ds offset reaches 64 KiB: images beyond it go through the second base)
      const h8 wh = o_hi < 65536u ? *reinterpret_cast<const h8*>(wbase + (a_off[s] + o_hi))
                                  : *reinterpret_cast<const h8*>(wbase + (b_off[s] + (o_hi - 65536u)));
      const h8 wl = o_lo < 65536u ? *reinterpret_cast<const h8*>(wbase + (a_off[s] + o_lo))
                                  : *reinterpret_cast<const h8*>(wbase + (b_off[s] + (o_lo - 65536u)));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fh[s], acc, 0, 0, 0);
    }
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // tile I/O in fragment layout: lane (li, hq) owns columns 16 ob + 4 hq .. +3 of node li.  A wave-instruction then touches 16 rows
  // in 64-B pieces (each 128-B line is completed by the neighbouring block's instruction, issued next).  The tile index is
  // wave-uniform: row bases are scalar, a lane adds one 32-bit offset and the block's immediate.
  uint32_t lane_off = (uint32_t)(li * A.ld4 + hq);      // float4 units inside the tile's 16 rows
  uint32_t col_ok = 0;                                  // bit ob: the lane's chunk of block ob lies inside the row
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) col_ok |= (4 * ob + hq < A.ld4 ? 1u : 0u) << ob;
  auto load_prev = [&](int t) -> int {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    int p = -1;
    if (A.prev_idx && ts < A.n_tiles && (int64_t)ts * 16 + li < A.n) p = (A.prev_idx + (int64_t)ts * 16)[li];
    return p;
  };
  auto load_rows = [&](int t, int p, float4 (&va)[NB], float4 (&vh)[NB]) {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    const bool row_ok = ts < A.n_tiles && (int64_t)ts * 16 + li < A.n;
    const float4* arow = A.agg + (int64_t)ts * 16 * A.ld4;                  // scalar
    const float4* hrow = A.hprev + ((int64_t)(p < 0 ? 0 : p) * A.ld4 + hq);
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      va[ob] = make_float4(0.f, 0.f, 0.f, 0.f);
      vh[ob] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row_ok && ((col_ok >> ob) & 1u)) {
        va[ob] = arow[lane_off + 4 * ob];
        if (p >= 0) vh[ob] = hrow[4 * ob];
      }
    }
  };

  if (wv >= NW / 2) __builtin_amdgcn_s_sleep(64);     // de-phase the two waves of a SIMD (see dense.hip)
  float4 va[NB], vh[NB];
  const int t_step = gridDim.x * NW;
  int t = blockIdx.x * NW + wv;
  int p_cur = load_prev(t);
  load_rows(t, p_cur, va, vh);
  int p_next = load_prev(t + t_step);
  for (; t < A.n_tiles; t += t_step) {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    const int64_t row0 = (int64_t)ts * 16;
#pragma unroll
    for (int s = 0; s < KST; ++s) { asm volatile("" : "+v"(a_off[s])); asm volatile("" : "+v"(b_off[s])); }
    asm volatile("" : "+v"(lane_off));
    const bool node_ok = row0 + li < A.n;
    const bool any_old = __ballot(p_cur >= 0) != 0ull;

    // ---- operands of this tile out of the prefetch registers; the next tile's loads fly under this tile's work ---------------------
    float sc1, inv1;
    h8 fh[KST], fl[KST];
    {
      float fx[KS];
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) { fx[4 * ob] = va[ob].x; fx[4 * ob + 1] = va[ob].y; fx[4 * ob + 2] = va[ob].z; fx[4 * ob + 3] = va[ob].w; }
      row_scale(row_max(fx, 0.f), sc1, inv1);
      split_frag(fx, sc1, fh, fl);
    }
    if (any_old) {
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) my_stash[ob * 64] = vh[ob];      // lane-private slots: read back for the split and for z * h
    }
    p_cur = p_next;
    load_rows(t + t_step, p_cur, va, vh);
    p_next = load_prev(t + 2 * t_step);
    // (the readout's (query, entity) pair too: fetched at the end it would make the wave wait out its own prefetch)
    const int64_t node = row0 + li;
    int2 qe = make_int2(0, 0);
    if (A.W_final && node_ok && hq == 0) qe = reinterpret_cast<const int2*>(A.nodes)[node];

    // ---- stage 1: x = act(W_h agg) ------------------------------------------------------------------------------------------
    float xf[KS];
    {
      const float sc_out = inv1 * inv_w * (ACT == 2 ? -2.0f * LOG2E : 1.0f);
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        f32x4 acc = zero4;
        mma(O_WH, O_WH + P_W, 16 * ob, fh, fl, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[r] * sc_out;
          if (ACT == 1) v = fmaxf(v, 0.f);
          else if (ACT == 2) v = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v)), -1.0f);   // tanh = 2 sigmoid(2x) - 1
          xf[4 * ob + r] = v;
        }
      }
    }

    // ---- GRU gates: x and the old state share one row scale, so that W_ih x and W_hh h add inside the accumulators; the new state
    // goes out in fragment layout as its blocks complete, and into the projection fragments ------------------------------------------
    h8 nh[KST], nl[KST];
    auto gru = [&](auto has_old) {
      constexpr bool OLD = decltype(has_old)::value;
      h8 xh[KST], xl[KST], hh[KST], hl[KST];
      float sc, inv;
      if constexpr (OLD) {
        float hf[KS];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
          const float4 q = my_stash[ob * 64];
          hf[4 * ob] = q.x; hf[4 * ob + 1] = q.y; hf[4 * ob + 2] = q.z; hf[4 * ob + 3] = q.w;
        }
        row_scale(row_max(hf, row_max(xf, 0.f)), sc, inv);
        split_frag(hf, sc, hh, hl);
      } else {
        row_scale(row_max(xf, 0.f), sc, inv);
      }
      split_frag(xf, sc, xh, xl);
      const float inv_s = inv * inv_w * -LOG2E, inv_t = inv * inv_w * (-2.0f * LOG2E);
      h4 nh0, nl0;
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        f32x4 ar = zero4, az = zero4, ai = zero4, ag = zero4;
        mma(O_WIH, O_WIH + P_G, 0 * DP + 16 * ob, xh, xl, ar);
        if constexpr (OLD) mma(O_WHH, O_WHH + P_G, 0 * DP + 16 * ob, hh, hl, ar);
        mma(O_WIH, O_WIH + P_G, 1 * DP + 16 * ob, xh, xl, az);
        if constexpr (OLD) mma(O_WHH, O_WHH + P_G, 1 * DP + 16 * ob, hh, hl, az);
        mma(O_WIH, O_WIH + P_G, 2 * DP + 16 * ob, xh, xl, ai);
        if constexpr (OLD) mma(O_WHH, O_WHH + P_G, 2 * DP + 16 * ob, hh, hl, ag);
        const float4 br = *reinterpret_cast<const float4*>(bias_l + 0 * DP + 16 * ob + 4 * hq);
        const float4 bz = *reinterpret_cast<const float4*>(bias_l + 1 * DP + 16 * ob + 4 * hq);
        const float4 bi = *reinterpret_cast<const float4*>(bias_l + 2 * DP + 16 * ob + 4 * hq);
        const float4 bh = *reinterpret_cast<const float4*>(bias_l + 3 * DP + 16 * ob + 4 * hq);
        float4 ho = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (OLD) ho = my_stash[ob * 64];
        const float hov[4] = {ho.x, ho.y, ho.z, ho.w};
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
          hnv[r] = fmaf(zg, hov[r] - ng, ng);       // (1 - z) n + z h
        }
        if (node_ok && ((col_ok >> ob) & 1u))
          (A.hidden_out + row0 * A.ld4)[lane_off + 4 * ob] = make_float4(hnv[0], hnv[1], hnv[2], hnv[3]);
        h4 ch, cl;                                              // |h| <= 1 in the model; any finite state below 65504 is carried
        split4(hnv[0], hnv[1], hnv[2], hnv[3], ch, cl);
        if (ob & 1) {
          nh[ob >> 1] = __builtin_shufflevector(nh0, ch, 0, 1, 2, 3, 4, 5, 6, 7);
          nl[ob >> 1] = __builtin_shufflevector(nl0, cl, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
          nh0 = ch; nl0 = cl;
        }
      }
    };
    if (any_old) gru(std::true_type{}); else gru(std::false_type{});

    // ---- projections of the new state --------------------------------------------------------------------------------------------
    if (A.Ws) {       // rows e = 4*hq + r of block 0
      f32x4 acc = zero4;
      mma(O_E, O_E + P_E, 0, nh, nl, acc);
      if (node_ok && 4 * hq < A.ap)
        reinterpret_cast<float4*>(A.a_s_out + node * A.ap)[hq] = make_float4(acc[0] * inv_w, acc[1] * inv_w, acc[2] * inv_w, acc[3] * inv_w);
    }
    if (A.W_final) {  // row 16 = register 0 of quarter 0 of block 1
      f32x4 acc = zero4;
      mma(O_E, O_E + P_E, 16, nh, nl, acc);
      if (node_ok && hq == 0) A.scores[(int64_t)qe.x * A.n_ent + qe.y] = acc[0] * inv_w;
    }
  }
}

template <int NB, int ACT>
int launch(const DenseArgs& A, hipStream_t s) {
  constexpr int DP = 16 * NB, SR = DP / 8, NW = DENSE_T / 64;
  const size_t lds = (size_t)(2 * 7 * DP * SR + 2 * 32 * SR) * 16 + 4 * DP * sizeof(float) + (size_t)NW * NB * 64 * sizeof(float4) + 16;
  RG_HIP(hipFuncSetAttribute((const void*)dense_split_kernel<NB, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, rg::ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL((dense_split_kernel<NB, ACT>), dim3(grid), dim3(DENSE_T), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int NB>
int launch_act(const DenseArgs& A, hipStream_t s) {
  return A.act == 0 ? launch<NB, 0>(A, s) : A.act == 1 ? launch<NB, 1>(A, s) : launch<NB, 2>(A, s);
}

}  // namespace

namespace rg {
int dense_split_launch(const DenseArgs& A, hipStream_t s) { return A.d <= 32 ? launch_act<2>(A, s) : launch_act<4>(A, s); }
}  // namespace rg
