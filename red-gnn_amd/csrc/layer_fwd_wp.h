// Word-parallel form of the fused forward message passing (kernel: layer_fwd_wp.hip), used on hops whose source frontier is
// sparse.  Internal to libredgnn.so; the C-ABI entry point is rg_layer_fwd (layer_fwd.hip), which picks the walk.
#pragma once
#include "common.h"

namespace rgwp {

struct WpArgs {
  // work space: (query group, pack) in group-major order; a group = 32 / n_sub consecutive queries of one bitmap word
  int64_t n_items;
  int32_t n_packs, n_sub, BW, W, n_slots;
  const int2* ent;
  const int4* pack;
  const int2* rows;
  const uint32_t* bits_old;   // entity-major frontier bitmaps [n_ent][BW]: previous level / new level
  const uint32_t* bits_new;
  const int2* bm_old;         // batch-major {word, popcount prefix} of both levels: node ids
  const int2* bm_new;
  const float4* hidden;
  const float4* rela;
  int ld4;
  const float4* a_s;
  const float4* a_r;
  const float4* a_q;
  const float* w_alpha;
  const float* b_alpha;
  int attn_dim, n_rela_rows;
  float4* agg;
  float4* partial;
  uint8_t* written;           // [B * n_slots]: set where a partial sum was stored (zeroed by the caller)
  int32_t* queues;            // 8 heads, RG_QSTRIDE ints apart, zeroed by the launcher
  int32_t ipt;                // items per queue ticket
  bool queues_clean;          // host side: the heads are already zero on the stream
};

// launches the kernel (and nothing else: the caller runs combine_kernel for cut rows, as after the per-query walk)
int launch(const WpArgs& A, int ap4, hipStream_t s);
// the kernel addresses the previous level's rows by 32-bit byte offsets: n_old * ld * 4 must stay below 4 GiB
bool offsets_fit(int64_t n_old, int32_t ld);

}  // namespace rgwp
