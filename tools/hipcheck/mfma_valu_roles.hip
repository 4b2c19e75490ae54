// Partner waves in different ROLES: does a wave that issues only MFMAs (or MFMAs with a few vector instructions mixed in) keep the
// matrix pipe's full rate while the other wave of its SIMD issues only vector instructions - and what does the vector wave get?
// (Question behind an explicit anti-phase schedule of the dense kernels' two waves per SIMD.)  512-thread blocks, one per CU: waves w and
// w + 4 share a SIMD.  Role loops are separate code (one uniform branch outside the loops).
//   M(K): per unit one v_mfma_f32_16x16x32_f16 (4 accumulator chains round robin) + K independent v_fma_f32
//   V   : per unit 4 independent v_fma_f32;   T: per unit 1 v_exp_f32 + 3 v_fma_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// role_hi / role_lo: 0 = idle (exits at once), 1 = M(K), 2 = V, 3 = T, 4 = M(K) with ONE accumulator chain, 5 = M(K) with two
template <int K>
__global__ void roles_kernel(float* out, long long* cyc, int iters, int role_lo, int role_hi) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
  const int wv = threadIdx.x >> 6;
  const int role = wv < 4 ? role_lo : role_hi;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * K + k) & 7]) : "v"(v[(u + k + 3) & 7]));
      }
    }
  } else if (role == 4 || role == 5) {
    const int nch = role == 4 ? 1 : 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (nch == 1) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[0], 0, 0, 0);
        else if (u & 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[1], 0, 0, 0);
        else acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[0], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * K + k) & 7]) : "v"(v[(u + k + 3) & 7]));
      }
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(v[(u + 3) & 7]));
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v[u]));
#pragma unroll
        for (int k = 0; k < 3; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[4 + ((u + k) & 3)]) : "v"(v[(u + k + 1) & 3]));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int K>
int run(int role_lo, int role_hi, const char* what) {
  const int blocks = 256, iters = 4096;
  float* out; long long* cyc;
  CK(hipMalloc(&out, blocks * 512 * 4)); CK(hipMalloc(&cyc, blocks * 8 * 8));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((roles_kernel<K>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters, role_lo, role_hi);
  CK(hipDeviceSynchronize());
  static long long h[2048];
  CK(hipMemcpy(h, cyc, blocks * 8 * 8, hipMemcpyDeviceToHost));
  double lo = 0, hi = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[b * 8 + w];
  // units per loop iteration: M: 4 MFMA units; V: 16 fma; T: 4 (exp + 3 fma)
  printf("%-62s waves 0-3: %7.1f cycles per iteration   waves 4-7: %7.1f\n", what, lo / (blocks * 4) / iters, hi / (blocks * 4) / iters);
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  printf("per loop iteration: M(K) = 4 MFMA + 4K fma;  V = 16 fma;  T = 4 exp + 12 fma\n");
  run<0>(1, 0, "M(0) alone");
  run<2>(1, 0, "M(2) alone");
  run<0>(2, 0, "V alone");
  run<0>(3, 0, "T alone");
  run<0>(1, 1, "M(0) + M(0)");
  run<2>(1, 1, "M(2) + M(2)");
  run<0>(1, 2, "M(0) + V");
  run<1>(1, 2, "M(1) + V");
  run<2>(1, 2, "M(2) + V");
  run<0>(1, 3, "M(0) + T");
  run<2>(1, 3, "M(2) + T");
  run<0>(2, 2, "V + V");
  run<0>(4, 0, "M(0) one chain, alone");
  run<0>(5, 0, "M(0) two chains, alone");
  run<0>(4, 2, "M(0) one chain + V");
  run<0>(4, 4, "M(0) one chain + the same");
  run<2>(4, 4, "M(2) one chain + the same");
  run<2>(4, 0, "M(2) one chain, alone");
  return 0;
}
