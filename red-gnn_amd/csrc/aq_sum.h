// grad_a_q[b] = sum of grad_a_s over the nodes of query b.  a_s[s] and a_q[b] enter the attention only as a sum, so
// the query projection's gradient is a segment sum of the node projection's; frontier nodes are sorted by query, and
// the rank of query b's first node is the popcount prefix of its first bitmap word, so the segments need no search.
#pragma once
#include "common.h"

namespace rg {
namespace {   // one copy per translation unit

constexpr int AQ_ROWS = 512;   // rows per workgroup

__global__ __launch_bounds__(256) void aq_sum_kernel(const int2* __restrict__ bm_old, int W, int B, int64_t n_old,
                                                     const float4* __restrict__ g_as, int ap4, float* __restrict__ g_aq) {
  __shared__ float4 part[256];
  const int b = blockIdx.x;
  const int64_t start = bm_old[(int64_t)b * W].y;
  const int64_t end = b + 1 < B ? (int64_t)bm_old[(int64_t)(b + 1) * W].y : n_old;
  const int64_t r0 = start + (int64_t)blockIdx.y * AQ_ROWS;
  const int64_t r1 = r0 + AQ_ROWS < end ? r0 + AQ_ROWS : end;
  if (r0 >= r1) return;
  const int lanes = 256 / ap4, col = threadIdx.x % ap4, sub = threadIdx.x / ap4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sub < lanes) {
    for (int64_t r = r0 + sub; r < r1; r += lanes) {
      const float4 v = g_as[r * ap4 + col];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < ap4) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < lanes; ++k) {
      const float4 v = part[k * ap4 + threadIdx.x];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* o = g_aq + ((int64_t)b * ap4 + threadIdx.x) * 4;
    atomicAdd(o + 0, s.x); atomicAdd(o + 1, s.y); atomicAdd(o + 2, s.z); atomicAdd(o + 3, s.w);
  }
}

// grad_a_q [B, ap] is WRITTEN (zero-filled here)
static inline int launch_aq_sum(const int2* bm_old, int W, int B, int32_t n_ent, int64_t n_old, const float* g_as, int ap,
                                float* g_aq, hipStream_t s) {
  if (zero_async(g_aq, (size_t)B * ap * sizeof(float), s)) return 1;
  if (n_old == 0) return 0;
  const dim3 grid((unsigned)B, (unsigned)ceil_div(n_ent, AQ_ROWS));
  hipLaunchKernelGGL(aq_sum_kernel, grid, dim3(256), 0, s, bm_old, W, B, n_old, (const float4*)g_as, ap / 4, g_aq);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace rg
