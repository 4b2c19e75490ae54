// Exact three-term f16 splits of fp32 operands for the f16 matrix pipe (dense_split3.hip, dense128_split3.hip, rg_split3_roundtrip).
//
// gfx950 has no reduced-precision fast path for f32 inputs (no xf32) and its f32 MFMA forms run at 1/16 of the f16 rate.  To use the
// f16 pipe WITHOUT narrowing the arithmetic, an fp32 value v, scaled by a power of two s, is carried as
//     v s = hi + mid + lo,    hi = f16(v s),  mid = f16(v s - hi),  lo = f16(v s - hi - mid)
// Each residual is computed exactly in fp32 (v s - hi has at most 13 significant bits, v s - hi - mid at most 2), so the three f16
// values (11 + 11 + 11 significant bits) hold all 24 bits of v: the sum is v s EXACTLY, as long as hi does not overflow f16 and the
// last bit of v s is not below f16's smallest subnormal (2^-24).  With the largest magnitude of a row scaled into [2^14, 2^15) that is
// every element within 2^-15 of the row's largest; smaller elements are carried with an absolute error <= 2^-39 of the row's
// largest (2^-15 of one ulp of it).  lo is 0 or +-1, +-2 units of v's last place: one significant bit.
// A product W X of two such operands is the six partial products of order >= 2^-22,
//     W_hi X_hi + W_hi X_mid + W_mid X_hi + W_hi X_lo + W_mid X_mid + W_lo X_hi      (dropped: <= 3 * 2^-33 |W X|)
// accumulated in the MFMA's fp32 accumulator: operands exact, products good to 2^-31, sums in fp32 - fp32 arithmetic on the f16 pipe
// at 6/16 of the f32 forms' time.  The weights' lo part (one significant bit) is stored as bf8 (e5m2) scaled by 2^8 - the same
// exponent range as f16 subnormals reach - and multiplied on the bf8 form of the same MFMA with X rounded to bf8 (3 significant
// bits: an error below 2^-3 of a 2^-22 term; the 2^-8 comes back where that chain's sums join the others), which keeps the d = 64
// weight images inside the CU's 160 KB of LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rg {
namespace sp3 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

constexpr float LO8_SCALE = 256.0f;            // weights' lo part is stored as bf8(lo * 2^8); the bf8 products' sums are scaled back by 2^-8
constexpr float LO8_INV = 1.0f / 256.0f;

// x - float(hi.lo / hi.hi) in one v_fma_mix_f32 (exact: see above)
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
// float(a.lo) + float(b.lo), float(a.hi) + float(b.hi), float(a.lo) + c, float(a.hi) + c
__device__ __forceinline__ float add_hh_lo(h2 a, h2 b) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float add_hh_hi(h2 a, h2 b) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float add_hf_lo(h2 a, float c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(c));
  return r;
}
__device__ __forceinline__ float add_hf_hi(h2 a, float c) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(c));
  return r;
}

// two (already scaled) floats -> hi, mid and the second residual (= lo before its conversion)
__device__ __forceinline__ void split3_2(float a, float b, h2& hi, h2& mid, float& ra, float& rb) {
  const f2v x = {a, b};
  hi = __builtin_convertvector(x, h2);
  const f2v r1 = {resid_lo(hi, a), resid_hi(hi, b)};
  mid = __builtin_convertvector(r1, h2);
  ra = resid_lo(mid, r1[0]);
  rb = resid_hi(mid, r1[1]);
}
// four scaled floats -> hi, mid, lo (f16)
__device__ __forceinline__ void split3_4(float a, float b, float c, float d, h4& hi, h4& mid, h4& lo) {
  h2 h0, h1, m0, m1;
  float r0, r1, r2, r3;
  split3_2(a, b, h0, m0, r0, r1);
  split3_2(c, d, h1, m1, r2, r3);
  const f2v q0 = {r0, r1}, q1 = {r2, r3};
  const h2 l0 = __builtin_convertvector(q0, h2), l1 = __builtin_convertvector(q1, h2);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
  mid = __builtin_shufflevector(m0, m1, 0, 1, 2, 3);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
}
// four scaled floats -> hi, mid (f16) and lo as four bf8 bytes of lo * 2^8 (the weights' form)
__device__ __forceinline__ void split3_4_lo8(float a, float b, float c, float d, h4& hi, h4& mid, uint32_t& lo8) {
  h2 h0, h1, m0, m1;
  float r0, r1, r2, r3;
  split3_2(a, b, h0, m0, r0, r1);
  split3_2(c, d, h1, m1, r2, r3);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(r0 * LO8_SCALE, r1 * LO8_SCALE, w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(r2 * LO8_SCALE, r3 * LO8_SCALE, w, true);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
  mid = __builtin_shufflevector(m0, m1, 0, 1, 2, 3);
  lo8 = (uint32_t)w;
}
// four scaled floats (below 2^15) -> four bf8 bytes (the activations' partner of the weights' lo8 image)
__device__ __forceinline__ uint32_t to_bf8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
  return (uint32_t)w;
}

// power-of-two scale that puts m (>= 0) into [2^14, 2^15), and its inverse; rows of (near) zeros keep a finite scale
__device__ __forceinline__ void row_scale(float m, float& sc, float& inv) {
  uint32_t eb = (__float_as_uint(m) >> 23) & 0xffu;
  eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
  sc = __uint_as_float((268u - eb) << 23);
  inv = __uint_as_float((eb - 14u) << 23);
}
// weights: the largest magnitude to [2^13, 2^14)
__device__ __forceinline__ float fit_weight_scale(float wmax) {
  if (!(wmax > 0.0f)) return 4096.0f;
  uint32_t eb = (__float_as_uint(wmax) >> 23) & 0xffu;
  eb = eb < 15u ? 15u : (eb > 254u ? 254u : eb);
  return __uint_as_float((267u - eb) << 23);
}

}  // namespace sp3
}  // namespace rg
