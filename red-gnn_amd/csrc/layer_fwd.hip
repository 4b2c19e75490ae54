// Static RED-GNN entry point of the fused forward kernels (per-query walk: layer_fwd_kernel.h; word-parallel form for hops
// whose source frontier is sparse: layer_fwd_wp.hip).
#include "layer_fwd_kernel.h"
#include "layer_fwd_wp.h"

extern "C" size_t rg_layer_fwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld) {
  if (!f || !g) return 0;
  // partial sums of cut rows [B][n_slots][ld], then one "written" byte per partial row (word-parallel walk)
  return rg::align_up((size_t)f->B * g->in_vr.n_slots * ld * sizeof(float), 256) + rg::align_up((size_t)f->B * g->in_vr.n_slots, 256) + 256;
}

// walk codes: 1 = per-query walk; 2, 3, 4, 5 = word-parallel with 32, 16, 8, 4 queries per item
static int plan_walk(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, int64_t n_new, int64_t n_edges, int32_t ld) {
  if (g->in_pk_packs.n == 0 || level != f->level || n_old < 0 || n_new <= 0 || n_edges < 0 || !rgwp::offsets_fit(n_old, ld)) return 1;
  // per-query walk: tests every in-edge of every live destination (~ n_new * mean in-degree candidates) after testing all
  // B * n_vrows items; word-parallel: touches the valid edges only, 32 queries' source rows per XCD at a time
  const double candidates = (double)n_new * (double)g->n_fact / (double)g->n_ent;
  const bool tiny = n_new * 16 < (int64_t)f->B * f->n_ent;
  // graphs of short rows (WN18RR-like: 5 in-edges per entity): the per-query walk pays its per-destination work (bitmap test,
  // ranks, lane-group set-up) for a handful of edges, the word-parallel walk packs rows into full waves: measured on C3 (B = 256)
  // it is 1.2-1.6x faster on every hop, saturated ones included (on C2 / C4, 41 in-edges per entity, 2x slower there)
  const bool short_rows = (double)g->n_fact < 12.0 * (double)g->n_ent;
  if (!tiny && !short_rows && (double)n_edges >= 0.4 * candidates) return 1;
  const double group_bytes = (double)n_old * ld * sizeof(float) / f->BW;      // source rows of one bitmap word's queries
  int code = group_bytes <= 3.0 * (1 << 20) ? 2 : (group_bytes <= 6.0 * (1 << 20) ? 3 : 4);
  // small batches: fewer items than the chip has wave slots -> the launch lasts as long as its heaviest item; smaller query groups
  // (more, lighter items) shorten that critical path (family, 50 queries: 800 items of ~170 edges -> 3200 of ~43)
  // (only while an item still carries a few hundred edges: lighter items cost more in tickets and fixed per-item work than they balance)
  auto items = [&](int c) { return (int64_t)f->BW * (1 << (c - 2)) * g->in_pk_packs.n; };
  // (2 and 1 queries per item - walk codes 6, 7 - measured neutral on family / C3 at 50-64 queries: not picked automatically)
  while (code < 5 && items(code) < 16384 && n_edges > 256 * items(code)) ++code;
  return code;
}

extern "C" int rg_layer_fwd_plan(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, int64_t n_new,
                                 int64_t n_edges, int32_t ld) {
  if (!f || !g) return 1;
  return plan_walk(f, g, level, n_old, n_new, n_edges, ld);
}

extern "C" int rg_layer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const float* hidden,
                            const float* rela, int32_t d, int32_t ld, const float* a_s, const float* a_r,
                            const float* a_q, int32_t ap, const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                            float* agg_out, void* scratch, size_t scratch_bytes, int32_t walk, void* stream) {
  RG_CHECK(f && g && hidden && rela && a_s && a_r && a_q && w_alpha && b_alpha && agg_out, "rg_layer_fwd: NULL argument");
  RG_CHECK((((uintptr_t)hidden | (uintptr_t)rela | (uintptr_t)a_s | (uintptr_t)a_r | (uintptr_t)a_q | (uintptr_t)agg_out |
             (uintptr_t)scratch) & 15) == 0, "rg_layer_fwd: float buffers must be 16-B aligned");
  RG_CHECK(walk >= 0 && walk <= 7, "rg_layer_fwd: walk=%d not in 0..7", walk);
  rgfwd::FwdArgs A;
  if (rgfwd::fill_common("rg_layer_fwd", f, g, level, n_new, d, ld, ap, attn_dim, scratch, scratch_bytes,
                         rg_layer_fwd_scratch_bytes(f, g, ld), &A)) return 1;
  if (n_new == 0) return 0;
  A.hidden = (const float4*)hidden; A.rela = (const float4*)rela;
  A.a_s = (const float4*)a_s; A.a_r = (const float4*)a_r; A.a_q = (const float4*)a_q;
  A.w_alpha = w_alpha; A.b_alpha = b_alpha;
  A.agg = (float4*)agg_out; A.partial = (float4*)scratch;
  hipStream_t s = (hipStream_t)stream;
  if (walk == 0)      // sizes known on the host (after rg_frontier_expand): pick; after expand_async the caller says which
    walk = plan_walk(f, g, level, f->n_nodes[(level - 1) % f->n_levels], f->n_nodes[level % f->n_levels],
                     level == f->level ? f->n_edges : -1, ld);
  if (walk >= 2 && f->n_nodes[(level - 1) % f->n_levels] < 0 && !rgwp::offsets_fit((int64_t)f->B * f->n_ent, ld))
    walk = 1;       // sizes unknown on the host (expand_async) and the full grid would overflow the 32-bit row offsets: per-query walk
  if (walk >= 2) {
    RG_CHECK(g->in_pk_packs.n > 0, "rg_layer_fwd: the word-parallel walk needs a static graph with packed entries");
    RG_CHECK(level == f->level, "rg_layer_fwd: the word-parallel walk reads the entity-major bitmaps of the newest hop only "
             "(level %d, newest %d)", level, f->level);
    // (after rg_frontier_expand_async the previous level's size is not known here: bounded by the full grid)
    const int64_t n_old = f->n_nodes[(level - 1) % f->n_levels] >= 0 ? f->n_nodes[(level - 1) % f->n_levels] : (int64_t)f->B * f->n_ent;
    RG_CHECK(rgwp::offsets_fit(n_old, ld), "rg_layer_fwd: the word-parallel walk addresses the previous level by 32-bit byte offsets "
             "(%lld rows x %d floats is 4 GiB or more)", (long long)n_old, ld);
    rgwp::WpArgs W;
    W.n_sub = 1 << (walk - 2);
    W.n_packs = g->in_pk_packs.n; W.BW = f->BW; W.W = f->W; W.n_slots = g->in_vr.n_slots;
    W.n_items = (int64_t)f->BW * W.n_sub * W.n_packs;
    W.ent = g->in_pk_packs.ent; W.pack = g->in_pk_packs.pack; W.rows = g->in_pk_packs.rows;
    W.bits_old = f->bitsT[f->tcur ^ 1]; W.bits_new = f->bitsT[f->tcur];
    W.bm_old = A.bm_old; W.bm_new = A.bm_new;
    W.hidden = A.hidden; W.rela = A.rela; W.ld4 = A.ld4; W.a_s = A.a_s; W.a_r = A.a_r; W.a_q = A.a_q;
    W.w_alpha = w_alpha; W.b_alpha = b_alpha; W.attn_dim = attn_dim; W.n_rela_rows = g->n_rela_rows;
    W.agg = A.agg; W.partial = A.partial; W.queues = f->queues; W.queues_clean = A.walk.queues_clean;
    // a ticket is a returning atomic: about 200 edges' worth of items each (light items: hop 0; heavy ones are taken one by one)
    const int64_t n_e = level == f->level ? (f->n_edges >= 0 ? f->n_edges : f->edge_hint) : -1;
    const int64_t by_work = n_e < 0 ? 2 : 200 * W.n_items / std::max<int64_t>(n_e, 1);
    W.ipt = (int32_t)std::max<int64_t>(std::min<int64_t>(std::min<int64_t>(by_work, W.n_items / 8192), 32), 1);   // ... and never fewer tickets than waves
    const size_t n_part = (size_t)f->B * g->in_vr.n_slots;
    W.written = n_part ? (uint8_t*)scratch + rg::align_up(n_part * ld * sizeof(float), 256) : nullptr;
    if (n_part && rg::zero_async(W.written, rg::align_up(n_part, 256), s)) return 1;
    if (rgwp::launch(W, ap / 4, s)) return 1;
    return rgfwd::launch_combine(A, f->B, g->in_vr, s, W.written);
  }
  // dense walk when at least a quarter of all (query, entity) pairs are visited; else filter 64 items per wave
  // (after rg_frontier_expand_async n_new is the caller's estimate: it only picks the walk, both are exact)
  const bool dense = n_new * 4 >= (int64_t)f->B * f->n_ent;
  return rgfwd::dispatch<false>(A, ld / 4, ap / 4, f->B, g->in_vr, dense, rg::walk_kpg(g->n_fact, g->in_vr.n), s);
}
