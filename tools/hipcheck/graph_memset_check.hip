// Does a memset node captured from hipMemsetAsync run on every replay of a hipGraph?  (Diagnostic for the graph-replay path.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(int* p, int n, int v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void add1(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main() {
  const int sizes[3] = {8, 256, 1 << 20};
  for (int si = 0; si < 3; ++si) {
    const int n = sizes[si];
    int *buf, *snap, host[4];
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&snap, 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(buf, 0, (size_t)n * 4, s));
    hipLaunchKernelGGL(add1, dim3((n + 255) / 256), dim3(256), 0, s, buf, n);
    CK(hipMemcpyAsync(snap, buf, 16, hipMemcpyDeviceToDevice, s));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 4; ++rep) {
      hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, 7);
      hipLaunchKernelGGL(fill, dim3(1), dim3(4), 0, s, snap, 4, -5);
      CK(hipGraphLaunch(ge, s));
      CK(hipMemcpyAsync(host, snap, 16, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      int last;
      CK(hipMemcpy(&last, buf + n - 1, 4, hipMemcpyDeviceToHost));
      printf("n=%d replay %d: snap[0]=%d buf[n-1]=%d (expect 1 1)\n", n, rep, host[0], last);
    }
  }
  return 0;
}
