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

// STAGED: the query's score row is copied to LDS on the pass that finds its minimum and every answer's counting pass reads it from
// there (a row is read from memory once instead of once per answer + 1; rows up to 16 k entities)
template <bool STAGED>
__global__ __launch_bounds__(RT) void rank_kernel(const float* __restrict__ scores, int n_ent,
                                                  const int32_t* __restrict__ ans_ptr, const int32_t* __restrict__ ans_idx,
                                                  const int32_t* __restrict__ filt_ptr, const int32_t* __restrict__ filt_idx,
                                                  float* __restrict__ ranks) {
  extern __shared__ float s_row[];
  __shared__ float s_min[RT / 64];
  __shared__ int s_cnt[3][RT / 64];
  const int q = blockIdx.x;
  const float* grow = scores + (int64_t)q * n_ent;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

  float mn = INFINITY;
  for (int j = threadIdx.x; j < n_ent; j += RT) {
    const float v = grow[j];
    if (STAGED) s_row[j] = v;
    mn = fminf(mn, v);
  }
  const float* row = STAGED ? s_row : grow;
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
  if (n_ent <= 16384)
    hipLaunchKernelGGL(rank_kernel<true>, dim3(batch), dim3(RT), (size_t)n_ent * sizeof(float), (hipStream_t)stream, scores, n_ent, ans_ptr,
                       ans_idx, filt_ptr, filt_idx, ranks_out);
  else
    hipLaunchKernelGGL(rank_kernel<false>, dim3(batch), dim3(RT), 0, (hipStream_t)stream, scores, n_ent, ans_ptr, ans_idx, filt_ptr,
                       filt_idx, ranks_out);
  RG_LAUNCH_CHECK();
  return 0;
}

// ---- hoisted attention tables of all layers in one launch (include/redgnn.h: rg_attn_tables) ---------------------------------
namespace {
constexpr int RG_MAX_LAYERS = 16;
struct TablesArgs {
  const float* rela[RG_MAX_LAYERS];
  const float* Wr[RG_MAX_LAYERS];
  const float* Wqr[RG_MAX_LAYERS];
  const float* bqr[RG_MAX_LAYERS];
  const int64_t* q_rel;
  float* a_r; float* a_q; float* rela_pad;
  int n_layer, n_rows, batch, d, ld, attn, ap;
};

// one thread per output element: [l][row][j] of a_r (rows = relations) and a_q (rows = queries), then rela_pad's elements
__global__ void attn_tables_kernel(TablesArgs A) {
  const int64_t per_layer_a = (int64_t)(A.n_rows + A.batch) * A.ap;
  const int64_t n_a = per_layer_a * A.n_layer;
  const int64_t n_pad = A.rela_pad ? (int64_t)A.n_layer * A.n_rows * A.ld : 0;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_a) {
    const int l = (int)(i / per_layer_a);
    const int64_t k = i - (int64_t)l * per_layer_a;
    const int row = (int)(k / A.ap), j = (int)(k - (int64_t)row * A.ap);
    const bool is_q = row >= A.n_rows;
    float acc = 0.f;
    if (j < A.attn) {
      const int r = is_q ? (int)A.q_rel[row - A.n_rows] : row;
      const float* x = A.rela[l] + (int64_t)r * A.d;
      const float* w = (is_q ? A.Wqr[l] : A.Wr[l]) + (int64_t)j * A.d;
      for (int c = 0; c < A.d; ++c) acc = fmaf(x[c], w[c], acc);
      if (is_q) acc += A.bqr[l][j];
    }
    if (is_q) A.a_q[((int64_t)l * A.batch + (row - A.n_rows)) * A.ap + j] = acc;
    else A.a_r[((int64_t)l * A.n_rows + row) * A.ap + j] = acc;
  } else if (i < n_a + n_pad) {
    const int64_t k = i - n_a;
    const int c = (int)(k % A.ld);
    const int64_t lr = k / A.ld;
    const int l = (int)(lr / A.n_rows), r = (int)(lr - (int64_t)l * A.n_rows);
    A.rela_pad[k] = c < A.d ? A.rela[l][(int64_t)r * A.d + c] : 0.f;
  }
}
}  // namespace

extern "C" int rg_attn_tables(int32_t n_layer, int32_t n_rela_rows, int32_t batch, int32_t d, int32_t ld, int32_t attn_dim, int32_t ap,
                              const float* const* rela, const float* const* Wr, const float* const* Wqr, const float* const* bqr,
                              const int64_t* q_rel, float* a_r_out, float* a_q_out, float* rela_pad_out, void* stream) {
  RG_CHECK(n_layer >= 1 && n_layer <= RG_MAX_LAYERS, "rg_attn_tables: n_layer=%d not in 1..%d", n_layer, RG_MAX_LAYERS);
  RG_CHECK(rela && Wr && Wqr && bqr && q_rel && a_r_out && a_q_out, "rg_attn_tables: NULL argument");
  RG_CHECK(n_rela_rows > 0 && batch > 0 && d > 0 && ld >= d && attn_dim > 0 && ap >= attn_dim, "rg_attn_tables: bad shape");
  RG_CHECK(ld == d || rela_pad_out, "rg_attn_tables: rela_pad_out needed when ld != d");
  TablesArgs A;
  for (int l = 0; l < n_layer; ++l) {
    RG_CHECK(rela[l] && Wr[l] && Wqr[l] && bqr[l], "rg_attn_tables: NULL pointer for layer %d", l);
    A.rela[l] = rela[l]; A.Wr[l] = Wr[l]; A.Wqr[l] = Wqr[l]; A.bqr[l] = bqr[l];
  }
  A.q_rel = q_rel; A.a_r = a_r_out; A.a_q = a_q_out; A.rela_pad = ld == d ? nullptr : rela_pad_out;
  A.n_layer = n_layer; A.n_rows = n_rela_rows; A.batch = batch; A.d = d; A.ld = ld; A.attn = attn_dim; A.ap = ap;
  const int64_t n = (int64_t)(n_rela_rows + batch) * ap * n_layer + (A.rela_pad ? (int64_t)n_layer * n_rela_rows * ld : 0);
  hipLaunchKernelGGL(attn_tables_kernel, dim3((unsigned)rg::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
  RG_LAUNCH_CHECK();
  return 0;
}
