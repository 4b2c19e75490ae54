// Which (lane, byte) of the 64-bit A / B operands of v_mfma_f32_16x16x32_bf8_bf8 holds which (row, k) / (k, col)?
// The guide gives the map for the f16 / bf16 forms only ("other dtypes: check the map with exact integer data").
// Hypothesis checked: lane l, byte j holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15] - the f16 form's map.
// Also checks v_cvt_pk_bf8_f32 (round to nearest even, byte order) on a few values.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// experiment e: 0..511 A one-hot (lane e / 8, byte e % 8), B all ones; 512..1023 the same with A and B swapped;
// 1024..2047: A one-hot at (lane 16 (x / 8), byte x % 8), B one-hot at (lane 16 (y / 8), byte y % 8) with x = (e - 1024) / 32, y = (e - 1024) % 32
__global__ void probe(float* out) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const uint64_t ones = 0x3c3c3c3c3c3c3c3cull;      // bf8 (e5m2) 1.0 = 0x3c
  uint64_t a = 0, b = 0;
  if (e < 512) { b = ones; if (lane == e / 8) a = 0x3cull << (8 * (e % 8)); }
  else if (e < 1024) { a = ones; if (lane == (e - 512) / 8) b = 0x3cull << (8 * (e % 8)); }
  else {
    const int x = (e - 1024) / 32, y = (e - 1024) % 32;
    if (lane == 16 * (x / 8)) a = 0x3cull << (8 * (x % 8));
    if (lane == 16 * (y / 8)) b = 0x3cull << (8 * (y % 8));
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8((long)a, (long)b, acc, 0, 0, 0);
  // C/D: col = lane & 15, row = 4 (lane >> 4) + r
  for (int r = 0; r < 4; ++r) out[(size_t)e * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

__global__ void cvt(const float* x, uint32_t* out, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(x[2 * i], x[2 * i + 1], w, false);
    w = __builtin_amdgcn_cvt_pk_bf8_f32(x[2 * i + 1], x[2 * i], w, true);
    out[i] = (uint32_t)w;
  }
}

int main() {
  const int E = 2048;
  float* d; CK(hipMalloc(&d, (size_t)E * 256 * 4));
  hipLaunchKernelGGL(probe, dim3(E), dim3(64), 0, 0, d);
  std::vector<float> h((size_t)E * 256);
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int e = 0; e < 512; ++e) {            // A one-hot: exactly one row of ones
    const int lane = e / 8, want = lane & 15;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      const float v = h[(size_t)e * 256 + i * 16 + j], w = i == want ? 1.f : 0.f;
      if (v != w) { if (bad < 10) printf("A one-hot lane %d byte %d: D[%d][%d] = %g (want %g)\n", lane, e % 8, i, j, v, w); ++bad; }
    }
  }
  for (int e = 512; e < 1024; ++e) {         // B one-hot: exactly one column of ones
    const int lane = (e - 512) / 8, want = lane & 15;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      const float v = h[(size_t)e * 256 + i * 16 + j], w = j == want ? 1.f : 0.f;
      if (v != w) { if (bad < 20) printf("B one-hot lane %d byte %d: D[%d][%d] = %g (want %g)\n", lane, e % 8, i, j, v, w); ++bad; }
    }
  }
  for (int e = 1024; e < 2048; ++e) {        // k pairing: D[0][0] = 1 iff x == y
    const int x = (e - 1024) / 32, y = (e - 1024) % 32;
    const float v = h[(size_t)e * 256], w = x == y ? 1.f : 0.f;
    if (v != w) { if (bad < 30) printf("k pairing A(k=%d) B(k=%d): D[0][0] = %g (want %g)\n", x, y, v, w); ++bad; }
  }
  printf("bf8 16x16x32 operand map = the f16 form's map: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);

  // conversions: value pairs -> bytes
  const float xs[16] = {1.0f, -2.0f, 1.125f, 1.375f, 1.625f, 1.875f, 3.0e-5f, 1.5258789e-05f /* 2^-16 */, 57344.f, 61440.f, 0.f, -0.f,
                        7.6293945e-06f /* 2^-17: half the smallest subnormal, ties to even -> 0 */, 2.2888184e-05f /* 1.5 * 2^-16 -> 2^-15 */, 0.75f, 96.f};
  float* dx; uint32_t* dout; CK(hipMalloc(&dx, sizeof(xs))); CK(hipMalloc(&dout, 8 * 4));
  CK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, dout, 8);
  uint32_t ho[8]; CK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
  for (int i = 0; i < 8; ++i)
    printf("cvt_pk_bf8_f32(%g, %g): low word bytes %02x %02x | high word (swapped args) %02x %02x\n", xs[2 * i], xs[2 * i + 1],
           ho[i] & 0xff, (ho[i] >> 8) & 0xff, (ho[i] >> 16) & 0xff, (ho[i] >> 24) & 0xff);
  return bad ? 2 : 0;
}
