// Filtered ranking on the device.
// Replaces cal_ranks (Static/transductive/utils.py:7-14: two scipy.rankdata sorts over the
// [B, n_ent] score matrix) and the python filter loop of base_model.py:107-115 by counting:
//   s'      = fl32(fl32(s - rowmin) + 1e-8)
//   rank(a) = #{j not in filter: s'_j > s'_a} + (#{j: s'_j == s'_a} + 1) / 2
// which equals  rankdata(-s', 'average')[a] - rankdata(-(s'*filter), 'min')[a] + 1  because
// every answer is in its query's filter set and s' > 0 everywhere.
#include "common.h"

namespace {

constexpr int RT = 256;

__device__ __forceinline__ float shifted(float s, float mn) { return __fadd_rn(__fsub_rn(s, mn), 1e-8f); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__global__ __launch_bounds__(RT) void rank_kernel(const float* __restrict__ scores, int n_ent,
                                                  const int32_t* __restrict__ ans_ptr, const int32_t* __restrict__ ans_idx,
                                                  const int32_t* __restrict__ filt_ptr, const int32_t* __restrict__ filt_idx,
                                                  float* __restrict__ ranks) {
  __shared__ float s_min[RT / 64];
  __shared__ int s_cnt[3][RT / 64];
  const int q = blockIdx.x;
  const float* row = scores + (int64_t)q * n_ent;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

  float mn = INFINITY;
  for (int j = threadIdx.x; j < n_ent; j += RT) mn = fminf(mn, row[j]);
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_down(mn, o, 64));
  if (lane == 0) s_min[w] = mn;
  __syncthreads();
  mn = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));

  const int fb = filt_ptr[q], fe = filt_ptr[q + 1];
  for (int ai = ans_ptr[q]; ai < ans_ptr[q + 1]; ++ai) {
    const int a = ans_idx[ai];
    const float sa = shifted(row[a], mn);
    int gt = 0, eq = 0, gtf = 0;
    for (int j = threadIdx.x; j < n_ent; j += RT) {
      const float sj = shifted(row[j], mn);
      gt += sj > sa;
      eq += sj == sa;
    }
    for (int j = fb + threadIdx.x; j < fe; j += RT) gtf += shifted(row[filt_idx[j]], mn) > sa;
    gt = wave_sum(gt); eq = wave_sum(eq); gtf = wave_sum(gtf);
    __syncthreads();
    if (lane == 0) { s_cnt[0][w] = gt; s_cnt[1][w] = eq; s_cnt[2][w] = gtf; }
    __syncthreads();
    if (threadIdx.x == 0) {
      int G = 0, E = 0, F = 0;
      for (int i = 0; i < RT / 64; ++i) { G += s_cnt[0][i]; E += s_cnt[1][i]; F += s_cnt[2][i]; }
      ranks[ai] = (float)(G - F) + (float)(E + 1) * 0.5f;
    }
  }
}

}  // namespace

extern "C" int rg_rank(const float* scores, int32_t batch, int32_t n_ent, const int32_t* ans_ptr, const int32_t* ans_idx,
                       const int32_t* filt_ptr, const int32_t* filt_idx, float* ranks_out, void* stream) {
  RG_CHECK(scores && ans_ptr && ans_idx && filt_ptr && filt_idx && ranks_out, "rg_rank: NULL argument");
  RG_CHECK(batch > 0 && n_ent > 0, "rg_rank: batch=%d n_ent=%d", batch, n_ent);
  hipLaunchKernelGGL(rank_kernel, dim3(batch), dim3(RT), 0, (hipStream_t)stream, scores, n_ent, ans_ptr, ans_idx, filt_ptr,
                     filt_idx, ranks_out);
  RG_LAUNCH_CHECK();
  return 0;
}
