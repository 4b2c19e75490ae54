// global_load_lds_dwordx4 by inline asm (hipcc does not count it): where does lane l's 16 bytes land, is M0 the byte base, do pieces beyond
// 64 KiB of LDS work, and does a counted s_waitcnt vmcnt(N) retire the OLDER pieces while younger ones are in flight?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void k(const char* src, char* dst, int lds_off, int pieces) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned base = (unsigned)(uintptr_t)lds + lds_off;
  for (int p = wv; p < pieces; p += 4) glds16(src + p * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + p * 1024));
  // a second, younger batch (to another region) that the counted wait leaves in flight
  const int mine = (pieces - wv + 3) / 4;
  for (int p = wv; p < pieces; p += 4) glds16(src + p * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + (pieces + p) * 1024));
  if (mine == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if (mine == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (mine == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  for (int i = threadIdx.x; i < pieces * 64; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(lds + lds_off)[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
  const int pieces = 14, bytes = pieces * 1024;
  static char h[16384], out[16384];
  for (int i = 0; i < bytes; ++i) h[i] = (char)(i * 7 + (i >> 8));
  char *src, *dst;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
  CK(hipMemcpy(src, h, bytes, hipMemcpyHostToDevice));
  for (int off : {0, 30720, 61440, 92160, 120000 / 1024 * 1024}) {
    CK(hipMemset(dst, 0, bytes));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, src, dst, off, pieces);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dst, bytes, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < bytes; ++i) bad += out[i] != h[i];
    printf("LDS offset %6d: %d of %d bytes differ\n", off, bad, bytes);
  }
  return 0;
}
