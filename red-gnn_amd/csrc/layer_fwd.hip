// Static RED-GNN entry point of the fused forward kernel (kernel: layer_fwd_kernel.h).
#include "layer_fwd_kernel.h"

extern "C" size_t rg_layer_fwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld) {
  if (!f || !g) return 0;
  return (size_t)f->B * g->in_vr.n_slots * ld * sizeof(float) + 256;
}

extern "C" int rg_layer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const float* hidden,
                            const float* rela, int32_t d, int32_t ld, const float* a_s, const float* a_r,
                            const float* a_q, int32_t ap, const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                            float* agg_out, void* scratch, size_t scratch_bytes, void* stream) {
  RG_CHECK(f && g && hidden && rela && a_s && a_r && a_q && w_alpha && b_alpha && agg_out, "rg_layer_fwd: NULL argument");
  RG_CHECK((((uintptr_t)hidden | (uintptr_t)rela | (uintptr_t)a_s | (uintptr_t)a_r | (uintptr_t)a_q | (uintptr_t)agg_out |
             (uintptr_t)scratch) & 15) == 0, "rg_layer_fwd: float buffers must be 16-B aligned");
  rgfwd::FwdArgs A;
  if (rgfwd::fill_common("rg_layer_fwd", f, g, level, n_new, d, ld, ap, attn_dim, scratch, scratch_bytes,
                         rg_layer_fwd_scratch_bytes(f, g, ld), &A)) return 1;
  if (n_new == 0) return 0;
  A.hidden = (const float4*)hidden; A.rela = (const float4*)rela;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha;
  A.agg = (float4*)agg_out; A.partial = (float4*)scratch;
  // dense walk when at least a quarter of all (query, entity) pairs are visited; else filter 64 items per wave
  // (after rg_frontier_expand_async n_new is the caller's estimate: it only picks the walk, both are exact)
  const bool dense = n_new * 4 >= (int64_t)f->B * f->n_ent;
  return rgfwd::dispatch<false>(A, ld / 4, ap / 4, f->B, g->in_vr, dense, rg::walk_kpg(g->n_fact, g->in_vr.n), (hipStream_t)stream);
}
