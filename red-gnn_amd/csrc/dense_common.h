// Shared by the fused dense kernels (dense.hip: d <= 64 with the weights resident in LDS; dense128.hip: d = 128 with the
// weights streamed through LDS).
#pragma once
#include "common.h"

namespace rg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DenseArgs {
  int64_t n;               // number of node rows, or their capacity when n_dev is given
  const int32_t* n_dev;    // device-side count (after rg_frontier_expand_async), or null
  int d, ld4;              // true width, row stride in float4
  const float4* agg;
  const float4* hprev;     // [n_old][ld4]
  const int32_t* prev_idx; // [n] or null (all new)
  const float* W_h;        // [d][d]
  const float* w_ih;       // [3d][d]
  const float* w_hh;
  const float* b_ih;       // [3d]
  const float* b_hh;
  const float* Ws;         // [attn][d] or null
  int attn, ap;
  float* a_s_out;          // [n][ap]
  const float* W_final;    // [d] or null
  const int32_t* nodes;    // [n][2]
  int n_ent;
  float* scores;           // [B*n_ent]
  float4* hidden_out;      // [n][ld4]
  int act;                 // 0 idd, 1 relu, 2 tanh
  int n_tiles;
  int64_t n_hint = 0;      // host side: expected number of rows when n is only a capacity (sizes the grid; any value is correct)
  // training variant (rg_dense_train_fwd): dropout mask in, GRU input and gate workspace out
  const float* mask = nullptr;   // [n][ld] 0 or 1/(1-p), or null
  float* x_out = nullptr;        // [n][ld]  act(W_h agg) * mask
  int probe = 0;                 // three-term kernel, test hook (rg_split3_product_check): 1 = hidden_out <- act(W_h agg), 2 = W_in x, 3 = W_hn h
  float* ws_out = nullptr;       // [n][5][d] = {r, z, n, h0, W_hn h0 + b_hn}: the workspace layout of aten's fused GRU cell
};

// v_exp_f32 / v_rcp_f32 forms (1 ulp each; __builtin_amdgcn_rcpf, not the correctly rounded __frcp_rn which expands
// to a full division): far inside the 1e-4 relative tolerance of the path
static __device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
static __device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));            // in (0, 1]: no overflow
  return copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}

// d = 128 (dense128.hip)
int dense128_launch(const DenseArgs& A, hipStream_t s);
// d <= 64 with the products as two-term f16 splits (dense_split.hip)
int dense_split_launch(const DenseArgs& A, hipStream_t s);
// d <= 64 with the products as exact three-term f16 splits = fp32 arithmetic on the f16 pipe (dense_split3.hip)
int dense_split3_launch(const DenseArgs& A, hipStream_t s);
// d = 128 with exact three-term splits, weights streamed from a split image in a caller-provided scratch (dense128_split3.hip)
int64_t dense128_split3_scratch_bytes();
int dense128_split3_launch(const DenseArgs& A, void* scratch, int64_t scratch_bytes, hipStream_t s);
// d = 128 with split products: the weights' split image goes through a caller-provided scratch (dense128_split.hip)
int64_t dense128_split_scratch_bytes();
int dense128_split_launch(const DenseArgs& A, void* scratch, int64_t scratch_bytes, hipStream_t s);

}  // namespace rg
