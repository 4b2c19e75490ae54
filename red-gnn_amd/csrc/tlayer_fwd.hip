// Temporal (T-RED-GNN interpolation) entry point of the fused forward kernel (kernel: layer_fwd_kernel.h).
// Replaces Temporal/interpolation/model_cuda.py:149-160,192 per layer: relative-time embedding gather,
// past/now/future linears, attention MLP on [h_s | rel | rel_q] and the torch_scatter sum.
// The direction-specific linears are hoisted by linearity, W_dir(h + r + tau) = W_dir h + W_dir r + W_dir tau:
// the caller passes hidden_dir [3*N_old, ld] (row 3*s + dir), rela_dir [3*n_rela_rows, ld] (row dir*n_rela_rows + r)
// and time_dir [3*n_time, ld] (row dir*n_time + |dt|), dir = 0 past (dt<0), 1 now, 2 future.
#include "layer_fwd_kernel.h"

extern "C" int rg_tlayer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const int32_t* q_time,
                             const float* hidden_dir, const float* rela_dir, const float* time_dir, int32_t d, int32_t ld,
                             const float* a_s, const float* a_r, const float* a_q, int32_t ap, const float* w_alpha,
                             const float* b_alpha, int32_t attn_dim, float* agg_out, void* scratch, size_t scratch_bytes,
                             void* stream) {
  RG_CHECK(f && g && q_time && hidden_dir && rela_dir && time_dir && a_s && a_r && a_q && w_alpha && b_alpha && agg_out,
           "rg_tlayer_fwd: NULL argument");
  RG_CHECK(g->in_time && g->n_time > 0, "rg_tlayer_fwd: the graph has no timestamps (build it with rg_tgraph_create)");
  RG_CHECK((int64_t)f->B * f->n_ent * 3 < ((int64_t)1 << 31), "rg_tlayer_fwd: 3 * batch * n_ent does not fit int32 row ids");
  RG_CHECK((((uintptr_t)hidden_dir | (uintptr_t)rela_dir | (uintptr_t)time_dir | (uintptr_t)a_s | (uintptr_t)a_r |
             (uintptr_t)a_q | (uintptr_t)agg_out | (uintptr_t)scratch) & 15) == 0, "rg_tlayer_fwd: float buffers must be 16-B aligned");
  rgfwd::FwdArgs A;
  if (rgfwd::fill_common("rg_tlayer_fwd", f, g, level, n_new, d, ld, ap, attn_dim, scratch, scratch_bytes,
                         rg_layer_fwd_scratch_bytes(f, g, ld), &A)) return 1;
  if (n_new == 0) return 0;
  A.hidden = (const float4*)hidden_dir; A.rela = (const float4*)rela_dir;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha;
  A.agg = (float4*)agg_out; A.partial = (float4*)scratch;
  A.in_time = g->in_time; A.q_time = q_time; A.n_time = g->n_time; A.time_tab = (const float4*)time_dir;
  const bool dense = n_new * 4 >= (int64_t)f->B * f->n_ent;
  return rgfwd::dispatch<true>(A, ld / 4, ap / 4, f->B, g->in_vr, dense, rg::walk_kpg(g->n_fact, g->in_vr.n), (hipStream_t)stream);
}

// Temporal EXTRAPOLATION entry point (Temporal/extrapolation/model_cuda_new_embedding.py:186-226 per layer): the same fused kernel
// with every query restricted to the data rows of its time window (rg_frontier_set_window) and one direction matrix (all edges
// lie in the past: past_linear only), hoisted as W_past (h + r + tau) = W_past h + W_past r + W_past tau:
//   hidden_p [N_old, ld] = W_past h;  rela_p [n_rela_rows, ld] = W_past rela;  time_p [n_tab, ld] = W_past time_embed(delta), delta = 0..n_tab-1
//   delta(edge, b) = q_time[b] - row_time[data row]   (self-loops, data row >= n_data: q_time[b] - loop_time[b]), clamped to n_tab - 1.
extern "C" int rg_xlayer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const int32_t* q_time,
                             const int32_t* loop_time, const int32_t* row_time, int32_t n_data, const float* hidden_p, const float* rela_p,
                             const float* time_p, int32_t n_tab, int32_t d, int32_t ld, const float* a_s, const float* a_r, const float* a_q,
                             int32_t ap, const float* w_alpha, const float* b_alpha, int32_t attn_dim, float* agg_out, void* scratch,
                             size_t scratch_bytes, void* stream) {
  RG_CHECK(f && g && q_time && loop_time && row_time && hidden_p && rela_p && time_p && a_s && a_r && a_q && w_alpha && b_alpha && agg_out,
           "rg_xlayer_fwd: NULL argument");
  RG_CHECK(g->in_time && g->n_time > 0, "rg_xlayer_fwd: the graph has no row ids (build it with rg_tgraph_create, time field = data row)");
  RG_CHECK(f->win_lo && f->win_hi, "rg_xlayer_fwd: call rg_frontier_set_window first");
  RG_CHECK(n_tab > 0 && n_data >= 0, "rg_xlayer_fwd: n_tab=%d n_data=%d", n_tab, n_data);
  RG_CHECK((((uintptr_t)hidden_p | (uintptr_t)rela_p | (uintptr_t)time_p | (uintptr_t)a_s | (uintptr_t)a_r | (uintptr_t)a_q |
             (uintptr_t)agg_out | (uintptr_t)scratch) & 15) == 0, "rg_xlayer_fwd: float buffers must be 16-B aligned");
  rgfwd::FwdArgs A;
  if (rgfwd::fill_common("rg_xlayer_fwd", f, g, level, n_new, d, ld, ap, attn_dim, scratch, scratch_bytes,
                         rg_layer_fwd_scratch_bytes(f, g, ld), &A)) return 1;
  if (n_new == 0) return 0;
  A.hidden = (const float4*)hidden_p; A.rela = (const float4*)rela_p;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha;
  A.agg = (float4*)agg_out; A.partial = (float4*)scratch;
  A.in_time = g->in_time; A.q_time = q_time; A.n_time = n_tab; A.time_tab = (const float4*)time_p;
  A.win_lo = f->win_lo; A.win_hi = f->win_hi; A.row_time = row_time; A.loop_time = loop_time; A.n_data = n_data;
  const bool dense = n_new * 4 >= (int64_t)f->B * f->n_ent;
  return rgfwd::dispatch<true>(A, ld / 4, ap / 4, f->B, g->in_vr, dense, rg::walk_kpg(g->n_fact, g->in_vr.n), (hipStream_t)stream);
}
