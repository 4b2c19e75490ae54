// Fused dense epilogue of one layer with the products on the f16 matrix pipe as two-term splits (inference path, d <= 64).
// Same operator as dense.hip (Static/transductive/models.py :41 W_h + act, :81-84 the gathered single-step GRU, the next layer's :36
// Ws_attn(hs), :86-88 the readout), same data flow (transposed products, node = lane, accumulators feed the next product), but every
// fp32 operand v is carried as   v = s * (hi + lo * 2^-11),   hi = fp16(v / s),  lo = fp16((v / s - hi) * 2^11)
// (s: a power of two per node row that puts the row's largest magnitude in [2^14, 2^15); weights are split unscaled), and a product
// W X is three v_mfma_f32_16x16x32_f16 with fp32 accumulation:   acc1 += W_hi X_hi;   acc2 += W_hi X_lo + W_lo X_hi;
// W X = s * (acc1 + acc2 * 2^-11).  hi + lo carries 22 significant bits of each operand and the dropped lo*lo term is 2^-22 of a
// product, so a 64-term dot product is accurate to about 4e-7 of its largest terms (fp32 FMA chains: about 1e-7): two orders inside
// the 1e-4 tolerance of the path, at 3/16 of the matrix-pipe time of v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md: the f32 forms run
// at 1/16 of the f16 rate).  The kernel is then bound by its HBM rows and its transcendentals instead of the matrix pipe.
// rg_dense_fwd(..., precision = 1) selects it; precision = 0 keeps the exact-fp32 kernel of dense.hip.
#include "dense_common.h"

using namespace rg;

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int DENSE_T = 512;
constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;

// hi / lo halves of four (already scaled) floats
__device__ __forceinline__ void split4(float a, float b, float c, float d, h4& hi, h4& lo) {
  const f4v x = {a, b, c, d};
  hi = __builtin_convertvector(x, h4);
  const f4v back = __builtin_convertvector(hi, f4v);
  const f4v r = (x - back) * LO_SCALE;
  lo = __builtin_convertvector(r, h4);
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

template <int DP>
__device__ __forceinline__ int swf(int row, int slot) { return row * (DP / 4) + (slot ^ (row & (DP / 4 - 1))); }   // f32 transposition tile

template <int NB>
__global__ __launch_bounds__(DENSE_T, 2) void dense_split_kernel(DenseArgs A) {
  constexpr int DP = 16 * NB;
  using G = Geo<DP>;
  constexpr int SR = G::SR, KST = G::KST;
  constexpr int S = DP / 4;        // float4 slots per f32 row
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
  float* bias_l = reinterpret_cast<float*>(E_lo + 32 * SR);     // [4][DP]
  float4* tiles = reinterpret_cast<float4*>(bias_l + 4 * DP);   // [NW][16][S]

  const int d = A.d;
  // ---- weights -> split f16 images --------------------------------------------------------------------------------------------
  const bool vec4 = (d & 3) == 0;
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
      h4 hi, lo;
      split4(v[0], v[1], v[2], v[3], hi, lo);
      const int ob = ch >> 2, hq = ch & 3, row = row0_dst + r;
      const int slot = G::at(row, 4 * (ob >> 1) + hq);
      reinterpret_cast<h4*>(hi_img + slot)[ob & 1] = hi;
      reinterpret_cast<h4*>(lo_img + slot)[ob & 1] = lo;
    }
  };
  load_w(Wh_hi, Wh_lo, A.W_h, d, 0, DP);
  for (int g = 0; g < 3; ++g) {
    load_w(Wih_hi, Wih_lo, A.w_ih + (int64_t)g * d * d, d, g * DP, DP);
    load_w(Whh_hi, Whh_lo, A.w_hh + (int64_t)g * d * d, d, g * DP, DP);
  }
  load_w(E_hi, E_lo, A.Ws, A.Ws ? A.attn : 0, 0, 16);
  load_w(E_hi, E_lo, A.W_final, A.W_final ? 1 : 0, 16, 16);
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

  auto read_frag = [&](float (&f)[KS]) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      const float4 v = tile[swf<DP>(li, 4 * ob + hq)];
      f[4 * ob + 0] = v.x; f[4 * ob + 1] = v.y; f[4 * ob + 2] = v.z; f[4 * ob + 3] = v.w;
    }
  };
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
  // (a1, a2) += the 16-row block at row0 of a weight image times the fragment
  // the lane's A-fragment slot of k-step s in a 16-row block (block bases are multiples of 16 rows: the swizzle depends on li only).
  // Kept as byte offsets that the tile loop re-launders every iteration, so that each fragment read is one ds_read_b128 with the
  // image / block offset as its immediate (hoisted out of the loop, the 116 addresses of a tile would each take a register).
  uint32_t a_off[KST];
#pragma unroll
  for (int s = 0; s < KST; ++s) a_off[s] = (uint32_t)G::at(li, 4 * s + hq) * 16u;
  const char* wbase = reinterpret_cast<const char*>(lds);
  auto mma = [&](const h8* hi_img, const h8* lo_img, int row0, const h8 (&fh)[KST], const h8 (&fl)[KST], f32x4& a1, f32x4& a2) {
    const uint32_t o_hi = (uint32_t)(reinterpret_cast<const char*>(hi_img) - wbase) + (uint32_t)row0 * SR * 16u;
    const uint32_t o_lo = (uint32_t)(reinterpret_cast<const char*>(lo_img) - wbase) + (uint32_t)row0 * SR * 16u;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      const h8 wh = *reinterpret_cast<const h8*>(wbase + (a_off[s] + o_hi));
      const h8 wl = *reinterpret_cast<const h8*>(wbase + (a_off[s] + o_lo));
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fh[s], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, fl[s], a2, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, fh[s], a2, 0, 0, 0);
    }
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // tile I/O: the wave's tile index is wave-uniform, so row bases are scalar and a lane adds one small 32-bit offset (per-lane 64-bit
  // addresses, kept over the whole loop, were what pushed the kernel past its 256 registers).  Loads run two tiles ahead for the
  // prev_idx words and one tile ahead for the rows, so that the gather of the old state never waits for its own index: with the
  // matrix phase this short, a wave that stalls on a load chain inside its tile loop is what the kernel's time would be made of.
  constexpr int NL = 16 * S / 64;      // wave-instructions per 16-row tile
  constexpr int RPI = 64 / S;          // rows per instruction
  const int lr = lane / S, lsl = lane - lr * S;
  uint32_t lane_row_off = (uint32_t)(lr * A.ld4 + lsl);          // float4 units inside the instruction's RPI rows (laundered per tile)
  const bool lane_col_ok = lsl < A.ld4;
  auto load_prev = [&](int t, int (&pv)[NL]) {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    const int64_t row0 = (int64_t)ts * 16;
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int64_t rb = row0 + it * RPI;
      pv[it] = -1;
      if (A.prev_idx && ts < A.n_tiles && rb + lr < A.n) pv[it] = (A.prev_idx + rb)[lr];
    }
  };
  auto load_rows = [&](int t, const int (&pv)[NL], float4 (&va)[NL], float4 (&vh)[NL], bool& has_old) {
    const int ts = __builtin_amdgcn_readfirstlane(t);
    const int64_t row0 = (int64_t)ts * 16;
    has_old = false;
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      va[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      vh[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int64_t rb = row0 + it * RPI;                       // scalar
      if (ts < A.n_tiles && rb + lr < A.n && lane_col_ok) {
        va[it] = (A.agg + rb * A.ld4)[lane_row_off];
        if (pv[it] >= 0) { vh[it] = A.hprev[(int64_t)pv[it] * A.ld4 + lsl]; has_old = true; }
      }
    }
  };

  if (wv >= NW / 2) __builtin_amdgcn_s_sleep(64);     // de-phase the two waves of a SIMD (see dense.hip)
  float4 va[NL], vh[NL];
  int pv[NL];
  bool has_old = false;
  const int t_step = gridDim.x * NW;
  int t = blockIdx.x * NW + wv;
  load_prev(t, pv);
  load_rows(t, pv, va, vh, has_old);
  load_prev(t + t_step, pv);
  for (; t < A.n_tiles; t += t_step) {
    const int64_t row0 = (int64_t)__builtin_amdgcn_readfirstlane(t) * 16;
#pragma unroll
    for (int s = 0; s < KST; ++s) asm volatile("" : "+v"(a_off[s]));
    asm volatile("" : "+v"(lane_row_off));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) tile[swf<DP>(it * RPI + lr, lsl)] = va[it];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float fx[KS];
    read_frag(fx);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) tile[swf<DP>(it * RPI + lr, lsl)] = vh[it];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool any_old = __ballot(has_old) != 0ull;
    // the next tile's rows and the prev_idx words of the one after fly under this tile's work
    load_rows(t + t_step, pv, va, vh, has_old);
    load_prev(t + 2 * t_step, pv);
    // (the readout's (query, entity) pair too: fetched at the end it would make the wave wait out its own prefetch)
    const int64_t node = row0 + li;
    int2 qe = make_int2(0, 0);
    if (A.W_final && node < A.n && hq == 0) qe = reinterpret_cast<const int2*>(A.nodes)[node];

    // ---- stage 1: x = act(W_h agg) ------------------------------------------------------------------------------------------
    float xf[KS];
    {
      float sc, inv;
      row_scale(row_max(fx, 0.f), sc, inv);
      h8 fh[KST], fl[KST];
      split_frag(fx, sc, fh, fl);
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        f32x4 a1 = zero4, a2 = zero4;
        mma(Wh_hi, Wh_lo, 16 * ob, fh, fl, a1, a2);
#pragma unroll
        for (int r = 0; r < 4; ++r) xf[4 * ob + r] = fmaf(a2[r], LO_INV, a1[r]) * inv;
        if (A.act == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xf[4 * ob + r] = fmaxf(xf[4 * ob + r], 0.f);
        } else if (A.act == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) xf[4 * ob + r] = fast_tanh(xf[4 * ob + r]);
        }
        __builtin_amdgcn_sched_barrier(0);     // one 16-row block at a time: interleaved blocks cost registers, not time
      }
    }

    // ---- GRU gates: x and the old state share one row scale, so that W_ih x and W_hh h add inside the accumulators.  The old state
    // stays in the tile: a block reads its 4 columns back for z * h and overwrites them with the new state, whose projection
    // fragments are built as the blocks complete ------------------------------------------------------------------------------
    h8 nh[KST], nl[KST];
    {
      h8 xh[KST], xl[KST], hh[KST], hl[KST];
      float sc, inv;
      {
        float hf[KS];
        read_frag(hf);
        row_scale(row_max(hf, row_max(xf, 0.f)), sc, inv);
        split_frag(xf, sc, xh, xl);
        if (any_old) split_frag(hf, sc, hh, hl);
      }
      h4 nh0, nl0;
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        f32x4 r1 = zero4, r2 = zero4, z1 = zero4, z2 = zero4, i1 = zero4, i2 = zero4, g1 = zero4, g2 = zero4;
        // gate by gate: with all 24 fragment reads of a block in flight at once the kernel would not fit its 256 registers
        mma(Wih_hi, Wih_lo, 0 * DP + 16 * ob, xh, xl, r1, r2);
        if (any_old) mma(Whh_hi, Whh_lo, 0 * DP + 16 * ob, hh, hl, r1, r2);
        __builtin_amdgcn_sched_barrier(0);
        mma(Wih_hi, Wih_lo, 1 * DP + 16 * ob, xh, xl, z1, z2);
        if (any_old) mma(Whh_hi, Whh_lo, 1 * DP + 16 * ob, hh, hl, z1, z2);
        __builtin_amdgcn_sched_barrier(0);
        mma(Wih_hi, Wih_lo, 2 * DP + 16 * ob, xh, xl, i1, i2);
        if (any_old) mma(Whh_hi, Whh_lo, 2 * DP + 16 * ob, hh, hl, g1, g2);
        __builtin_amdgcn_sched_barrier(0);
        const float4 br = *reinterpret_cast<const float4*>(bias_l + 0 * DP + 16 * ob + 4 * hq);
        const float4 bz = *reinterpret_cast<const float4*>(bias_l + 1 * DP + 16 * ob + 4 * hq);
        const float4 bi = *reinterpret_cast<const float4*>(bias_l + 2 * DP + 16 * ob + 4 * hq);
        const float4 bh = *reinterpret_cast<const float4*>(bias_l + 3 * DP + 16 * ob + 4 * hq);
        const float4 ho = tile[swf<DP>(li, 4 * ob + hq)];
        const float hov[4] = {ho.x, ho.y, ho.z, ho.w};
        const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w};
        const float biv[4] = {bi.x, bi.y, bi.z, bi.w}, bhv[4] = {bh.x, bh.y, bh.z, bh.w};
        float hnv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float rg = fast_sigmoid(fmaf(fmaf(r2[r], LO_INV, r1[r]), inv, brv[r]));
          const float zg = fast_sigmoid(fmaf(fmaf(z2[r], LO_INV, z1[r]), inv, bzv[r]));
          const float ai = fmaf(fmaf(i2[r], LO_INV, i1[r]), inv, biv[r]);
          const float ah = fmaf(fmaf(g2[r], LO_INV, g1[r]), inv, bhv[r]);
          const float ng = fast_tanh(ai + rg * ah);
          hnv[r] = (1.0f - zg) * ng + zg * hov[r];
        }
        tile[swf<DP>(li, 4 * ob + hq)] = make_float4(hnv[0], hnv[1], hnv[2], hnv[3]);
        h4 ch, cl;                                              // |h| <= 1: no row scale
        split4(hnv[0], hnv[1], hnv[2], hnv[3], ch, cl);
        if (ob & 1) {
          nh[ob >> 1] = __builtin_shufflevector(nh0, ch, 0, 1, 2, 3, 4, 5, 6, 7);
          nl[ob >> 1] = __builtin_shufflevector(nl0, cl, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
          nh0 = ch; nl0 = cl;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- projections of the new state --------------------------------------------------------------------------------------------
    if (A.Ws) {       // rows e = 4*hq + r of block 0
      f32x4 a1 = zero4, a2 = zero4;
      mma(E_hi, E_lo, 0, nh, nl, a1, a2);
      if (node < A.n && 4 * hq < A.ap)
        reinterpret_cast<float4*>(A.a_s_out + node * A.ap)[hq] =
            make_float4(fmaf(a2[0], LO_INV, a1[0]), fmaf(a2[1], LO_INV, a1[1]), fmaf(a2[2], LO_INV, a1[2]), fmaf(a2[3], LO_INV, a1[3]));
    }
    if (A.W_final) {  // row 16 = register 0 of quarter 0 of block 1
      f32x4 a1 = zero4, a2 = zero4;
      mma(E_hi, E_lo, 16, nh, nl, a1, a2);
      if (node < A.n && hq == 0) A.scores[(int64_t)qe.x * A.n_ent + qe.y] = fmaf(a2[0], LO_INV, a1[0]);
    }

    // ---- new state: the tile now holds it node-major; store coalesced rows ---------------------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int64_t rb = row0 + it * RPI;
      if (rb + lr < A.n && lane_col_ok) (A.hidden_out + rb * A.ld4)[lane_row_off] = tile[swf<DP>(it * RPI + lr, lsl)];
    }
  }
}

template <int NB>
int launch(const DenseArgs& A, hipStream_t s) {
  constexpr int DP = 16 * NB, S = DP / 4, SR = DP / 8, NW = DENSE_T / 64;
  const size_t lds = (size_t)(2 * 7 * DP * SR + 2 * 32 * SR) * 16 + 4 * DP * sizeof(float) + (size_t)NW * 16 * S * sizeof(float4);
  RG_HIP(hipFuncSetAttribute((const void*)dense_split_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, rg::ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL((dense_split_kernel<NB>), dim3(grid), dim3(DENSE_T), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

namespace rg {
int dense_split_launch(const DenseArgs& A, hipStream_t s) { return A.d <= 32 ? launch<2>(A, s) : launch<4>(A, s); }
}  // namespace rg
