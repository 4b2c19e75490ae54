// How many independent VALU instructions fit into the shadow of one v_mfma_f32_16x16x32_f16 in ONE wave's instruction stream, and
// what do two such waves on a SIMD do to each other?  (Question behind the dense kernels' "matrix time + vector time add up".)
//   per iteration: 1 MFMA (4 independent accumulator chains, round robin) + K v_fma_f32 on independent registers; 4096 iterations
//   waves per SIMD: 1 (256-thread blocks, one per CU) or 2 (512-thread blocks)
//   mode 2: waves 0-3 issue only MFMAs, waves 4-7 only VALU (its per-unit branches pollute the number: ignore)
// Measured (MI355X): one wave: 18 / 21 / 25 / 32 / 34 / 43 / 55 cycles per unit at K = 0 / 1 / 2 / 3 / 4 / 6 / 8 - a lone wave hides NO vector
// instruction behind its MFMAs (each costs ~4.5 cycles on top); two waves of the same finely interleaved program: 14.1 / 16.4 / 22.7 /
// 35.9 cycles per unit and SIMD at K = 0 / 2 / 4 / 8 - two vector instructions per MFMA are free, the rest cost ~4.5 cycles each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int K, int MODE>
__global__ void loop_kernel(float* out, long long* cyc, int iters) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
  const int wv = threadIdx.x >> 6;
  const bool do_m = MODE != 2 || wv < 4, do_v = MODE != 2 || wv >= 4;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (do_m) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
      if (do_v) {
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(u * K + k) & 7]) : "v"(v[(u + k + 3) & 7] ));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wv] = t1 - t0;
}

template <int K, int MODE>
int run(int threads, const char* what) {
  const int blocks = 256, iters = 4096;
  float* out; long long* cyc;
  CK(hipMalloc(&out, blocks * threads * 4)); CK(hipMalloc(&cyc, blocks * 8 * 8));
  hipLaunchKernelGGL((loop_kernel<K, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((loop_kernel<K, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  CK(hipDeviceSynchronize());
  long long h[2048];
  CK(hipMemcpy(h, cyc, blocks * (threads / 64) * 8, hipMemcpyDeviceToHost));
  double s = 0; const int n = blocks * (threads / 64);
  for (int i = 0; i < n; ++i) s += h[i];
  printf("%-44s K=%d: %.1f cycles per (MFMA + K VALU) per wave\n", what, K, s / n / (iters * 4.0));
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

int main() {
  run<0, 0>(256, "1 wave/SIMD, MFMA only"); run<1, 0>(256, "1 wave/SIMD"); run<2, 0>(256, "1 wave/SIMD"); run<3, 0>(256, "1 wave/SIMD");
  run<4, 0>(256, "1 wave/SIMD"); run<6, 0>(256, "1 wave/SIMD"); run<8, 0>(256, "1 wave/SIMD");
  run<0, 0>(512, "2 waves/SIMD, MFMA only"); run<2, 0>(512, "2 waves/SIMD, same program"); run<4, 0>(512, "2 waves/SIMD, same program");
  run<8, 0>(512, "2 waves/SIMD, same program");
  run<2, 2>(512, "2 waves/SIMD: w0-3 MFMA only, w4-7 VALU only"); run<4, 2>(512, "2 waves/SIMD: w0-3 MFMA only, w4-7 VALU only");
  run<8, 2>(512, "2 waves/SIMD: w0-3 MFMA only, w4-7 VALU only");
  return 0;
}
