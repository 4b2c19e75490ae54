/*
 * redgnn.h — C-ABI of the MI355X-native RED-GNN hot path (libredgnn.so).
 *
 * The reference (LARS-research/RED-GNN) is pure Python and has no FFI; its boundary is the
 * nn.Module API of RED_GNN_trans plus the DataLoader.get_neighbors callback
 * (Static/transductive/models.py:45-89, load_data.py:106-131).  This header is the native
 * boundary *behind* that API: every entry point names the reference code it replaces.
 * All paths below are relative to Static/transductive/ of the reference.
 *
 * Conventions
 *   - plain C types only: device pointers (const float* / const int32_t*), sizes, an opaque
 *     stream (hipStream_t passed as void*).  No torch types.
 *   - every function returns 0 on success, non-zero on error; rg_last_error() then returns a
 *     thread-local message.  Nothing is printed, nothing aborts.
 *   - buffers are caller-owned.  The library allocates device memory only inside rg_graph
 *     handles; per-batch state lives in a caller-provided workspace (rg_frontier).
 *   - launches go to the stream passed in and are asynchronous, except rg_frontier_expand,
 *     which returns the new node/edge counts and therefore synchronises that stream once.
 *   - indices are int32 (requires B*n_ent < 2^31 and n_fact < 2^31; checked).
 *   - node order is the reference's: sorted by (batch_idx, entity); node id = rank in that order.
 */
#ifndef REDGNN_H
#define REDGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rg_graph rg_graph;        /* device-resident KG: CSR by head and CSR by tail */
typedef struct rg_frontier rg_frontier;  /* per-batch visited-set state inside a caller workspace */

const char* rg_last_error(void);
int rg_version(void);
/* number of memset nodes in a captured hipGraph_t (-1 on error).  The library issues its fills as kernels because replayed graphs
 * with several memset nodes misbehaved on ROCm 7.2; the host side asserts that nothing else put one into a captured forward. */
int rg_hipgraph_fill_nodes(void* hip_graph);

/* ---- graph build: replaces load_data.py:69-81 (double_triple + load_graph) ------------------
 * triples: HOST int32 [n,3] = (head, rel, tail) base triples.  If add_inverse != 0 the inverse
 * (tail, rel+n_rel, head) of every triple is added (load_data.py:69-74).  One identity row
 * (e, 2*n_rel, e) per entity is always added (load_data.py:77-79).  Fact-row order is kept
 * inside every CSR row so sums are reproducible. */
int rg_graph_create(int32_t n_ent, int32_t n_rel, const int32_t* triples_host, int64_t n,
                    int add_inverse, rg_graph** out);
/* The same graph from a DEVICE triple array (int32 [n,3]), built on the device: the per-epoch rebuild of shuffle_train
 * (load_data.py:152-164: permute facts + train, re-split 3:1, load_graph) without moving the triples or the CSR arrays through the host.
 * Every array of the result equals rg_graph_create's on the same triples.  Synchronises `stream` (it returns counts). */
int rg_graph_create_device(int32_t n_ent, int32_t n_rel, const int32_t* triples_dev, int64_t n,
                           int add_inverse, void* stream, rg_graph** out);
/* copy the word-parallel walk's packs back (tests): sizes first (out arrays NULL), then ent int32 [n_packs*128, 2], pack int32 [n_packs, 4],
 * rows int32 [n_vrows, 2]; and the virtual rows of the CSR-by-tail, int32 [n_vrows, 4]. */
int rg_graph_export_packs(const rg_graph* g, int32_t* n_packs, int32_t* n_vrows, int32_t* ent_host, int32_t* pack_host, int32_t* rows_host,
                          int32_t* vrows_host);
/* temporal graph (T-RED-GNN interpolation): replaces the per-call coo_matrix build of
 * Temporal/interpolation/model_cuda.py:121-126 over the quadruple array of graph.py:34-49.
 * quads: HOST int32 [n,4] = (head, rel, tail, time id), used as given (the reference's graph array already
 * holds the identity rows with the sentinel timestamp); n_rela_rows = rows of the relation tables. */
int rg_tgraph_create(int32_t n_ent, int32_t n_rela_rows, int32_t n_time, const int32_t* quads_host, int64_t n,
                     rg_graph** out);
/* The same graph without the rows listed in exclude_rows_host (row indices into quads, duplicates allowed): the training
 * mode of T_RED_GNN.forward, `dataset = np.delete(self.dataset, batch['example_idx'], axis=0)` (model_cuda.py:103-104),
 * without a host copy of the array per batch. */
int rg_tgraph_create_excluding(int32_t n_ent, int32_t n_rela_rows, int32_t n_time, const int32_t* quads_host, int64_t n,
                               const int64_t* exclude_rows_host, int64_t n_exclude, rg_graph** out);
int rg_graph_destroy(rg_graph* g);
int64_t rg_graph_n_fact(const rg_graph* g);      /* rows incl. inverse + identity (load_data.py:80) */
/* copy the device CSR back (tests): ptr arrays have n_ent+1 entries, pair arrays 2*n_fact. */
int rg_graph_export(const rg_graph* g, int32_t* out_ptr_host, int32_t* out_rel_tail_host,
                    int32_t* in_ptr_host, int32_t* in_head_rel_host);

/* ---- frontier expansion: replaces DataLoader.get_neighbors, load_data.py:106-131 ------------
 * A frontier keeps `n_levels` visited-set snapshots: level 0 = the query nodes, level k = after
 * k hops.  Levels are stored modulo n_levels: 2 is enough for inference (ping-pong), training
 * keeps n_layer+1 so that rg_layer_bwd can revisit every hop. */
size_t rg_frontier_workspace_bytes(int32_t n_ent, int32_t batch, int32_t n_levels);
/* workspace: device memory of at least rg_frontier_workspace_bytes(), 256-B aligned. */
int rg_frontier_create(int32_t n_ent, int32_t batch, int32_t n_levels, void* workspace_dev,
                       size_t workspace_bytes, rg_frontier** out);
int rg_frontier_destroy(rg_frontier* f);
/* level 0 frontier {(b, q_sub[b])}: models.py:73.  q_sub: device int32 [batch]. */
int rg_frontier_reset(rg_frontier* f, const int32_t* q_sub_dev, void* stream);
/* level 0 frontier from an arbitrary node set (the general form get_neighbors accepts,
 * load_data.py:106,115): nodes device int32 [n,2] = (batch, entity), batch < `batch`, no duplicates. */
int rg_frontier_reset_nodes(rg_frontier* f, const int32_t* nodes_dev, int64_t n, void* stream);
/* one hop: new frontier = tails of all out-edges of the current one (identity edges keep the
 * old nodes).  counts_host[0] = N_new, counts_host[1] = E (edges of this hop),
 * counts_host[2] = N_old, counts_host[3] = new level.  Synchronises `stream`. */
int rg_frontier_expand(rg_frontier* f, const rg_graph* g, int64_t* counts_host, void* stream);
/* The same hop without the read-back: nothing synchronises, so a whole forward (models.py:69-88) can be enqueued - or
 * captured into a hipGraph - in one go.  The level's N stays on the device (rg_frontier_count_ptr; consumed by
 * rg_dense_fwd_dev), and while a level's size is unknown on the host rg_layer_fwd / rg_tlayer_fwd take their n_new as
 * an estimate (> 0, it only picks the walk) and need agg_out sized for batch * n_ent rows.
 * rg_frontier_level_counts reads N and E of levels 0..current back: counts_host[2*l] = N_l, [2*l+1] = E_l
 * (room for 2 * 16 values; synchronises `stream`). */
int rg_frontier_expand_async(rg_frontier* f, const rg_graph* g, void* stream);
/* rg_frontier_expand_async + rg_frontier_nodes (nodes_out [batch*n_ent, 2], prev_idx_out [batch*n_ent]; either may be NULL) in one
 * call: for small batches (batch * ceil(n_ent/32) <= 12288 words) the level build and the node list are ONE single-workgroup launch
 * instead of six, which is what a replayed graph at the reference's n_tbatch = 50 is made of (launch latency, not work). */
int rg_frontier_expand_nodes_async(rg_frontier* f, const rg_graph* g, int32_t* nodes_out, int32_t* prev_idx_out, void* stream);
/* After an asynchronous expansion the hop's edge count is not known on the host; a caller that knows what to expect (from an eager run of
 * the same shape) says so: it only tunes the work distribution of the next rg_layer_fwd (any value is correct).  Cleared by the next expansion. */
int rg_frontier_set_edge_hint(rg_frontier* f, int64_t n_edges);
const int32_t* rg_frontier_count_ptr(const rg_frontier* f);
int rg_frontier_level_counts(const rg_frontier* f, int64_t* counts_host, void* stream);
/* nodes of the current level: nodes_out int32 [N_new,2] = (batch, entity) sorted
 * (== tail_nodes, load_data.py:123); prev_idx_out int32 [N_new] = index of the node in the
 * previous level or -1; old_nodes_new_idx_out int32 [N_old] (load_data.py:127-129).
 * Any pointer may be NULL. */
int rg_frontier_nodes(const rg_frontier* f, int32_t* nodes_out, int32_t* prev_idx_out,
                      int32_t* old_nodes_new_idx_out, void* stream);
/* materialised edge list of hop `level-1 -> level` (API parity with sampled_edges,
 * load_data.py:118-125): edges_out int32 [E,6] = (batch, head, rel, tail, old_idx, new_idx),
 * grouped by new_idx (destination-segmented); row_ptr_out int32 [N_new+1].
 * nodes_new: device int32 [N_new,2] from rg_frontier_nodes.  scratch: device memory of
 * rg_frontier_edges_scratch_bytes(N_new) bytes. */
size_t rg_frontier_edges_scratch_bytes(int64_t n_new);
int rg_frontier_edges(const rg_frontier* f, const rg_graph* g, int32_t level,
                      const int32_t* nodes_new, int64_t n_new, int32_t* edges_out,
                      int32_t* row_ptr_out, void* scratch_dev, void* stream);

/* ---- layer forward: replaces GNNLayer.forward models.py:29-39 incl. torch_scatter.scatter ----
 * For hop level-1 -> level:
 * agg[o] = sum over in-edges e=(s,r,o) of alpha_e * (hidden[s] + rela[r]),
 * alpha_e = sigmoid(w_alpha . relu(a_s[s] + a_r[r] + a_q[b]) + b_alpha), with the hoisted
 * projections a_s = hidden Ws^T [N_old,ap], a_r = rela Wr^T [2R+1,ap], a_q = rela[q_rel] Wqr^T + b [B,ap]
 * (ap = attention dim padded to a multiple of 4, pad columns zero).  Edges are enumerated
 * from the CSR-by-tail of `g` and the frontier bitmaps; no edge list is materialised.
 * hidden [N_old, ld], rela [2R+1, ld], agg_out [N_new, ld] (every row is written);
 * ld % 4 == 0, ld >= d, pad columns of hidden/rela must be zero.  n_new is checked against the
 * frontier.  scratch: device memory of rg_layer_fwd_scratch_bytes() bytes (partial sums of hub
 * destinations that are cut into segments), 16-B aligned.
 *
 * walk: how the edges are enumerated (the sums and their order are the same; results are bitwise equal):
 *   0  let the library pick from the sizes of the hop (known on the host after rg_frontier_expand);
 *   1  per-query walk: every live destination tests its KG in-edges against the previous frontier;
 *   2 .. 7  word-parallel walk for hops whose SOURCE frontier is sparse (as the reference expands from the frontier's
 *      nodes, load_data.py:115-118): 32 / 16 / 8 / 4 / 2 / 1 queries per work item straight from the entity-major bitmaps; only for
 *      level == the newest hop of a static graph.
 * rg_layer_fwd_plan returns what walk 0 would pick for given sizes (n_old, n_new nodes, n_edges of the hop): callers that
 * enqueue without read-backs (rg_frontier_expand_async) record it from an eager run and pass it explicitly. */
size_t rg_layer_fwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld);
int rg_layer_fwd_plan(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, int64_t n_new,
                      int64_t n_edges, int32_t ld);
int rg_layer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new,
                 const float* hidden, const float* rela, int32_t d, int32_t ld,
                 const float* a_s, const float* a_r, const float* a_q, int32_t ap,
                 const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                 float* agg_out, void* scratch_dev, size_t scratch_bytes, int32_t walk, void* stream);

/* ---- hoisted attention tables of ALL layers in one launch: models.py:33,36 (Wr_attn, Wqr_attn applied per relation / per query
 * instead of per edge).  For layer l: a_r_out[l][r][ap] = rela[l][r] . Wr[l][j], a_q_out[l][b][ap] = rela[l][q_rel[b]] . Wqr[l][j] + bqr[l][j]
 * (columns j >= attn_dim zero), rela_pad_out[l][r][ld] = rela[l][r] zero-padded to ld columns (NULL when ld == d).
 * rela / Wr / Wqr / bqr: HOST arrays of n_layer device pointers (tables [n_rela_rows, d], weights [attn_dim, d], bias [attn_dim]);
 * q_rel device int64 [batch].  n_layer <= 16. */
int rg_attn_tables(int32_t n_layer, int32_t n_rela_rows, int32_t batch, int32_t d, int32_t ld, int32_t attn_dim, int32_t ap,
                   const float* const* rela, const float* const* Wr, const float* const* Wqr, const float* const* bqr,
                   const int64_t* q_rel, float* a_r_out, float* a_q_out, float* rela_pad_out, void* stream);

/* ---- temporal layer forward: replaces Temporal/interpolation/model_cuda.py:149-160,192 ------------------
 * agg[o] = sum_e alpha_e * (hidden_dir[3 s + dir_e] + rela_dir[dir_e * n_rela_rows + r] + time_dir[dir_e * n_time + |dt_e|]),
 * dt_e = time(e) - q_time[b], dir = 0 (dt<0, past) / 1 (dt=0, now) / 2 (dt>0, future): the three direction
 * linears applied per node / relation / |dt| by the caller (W(h+r+tau) = Wh + Wr + Wtau);
 * alpha_e as in rg_layer_fwd with a_s, a_r, a_q the three blocks of attention_1 (no biases: pass b_alpha = 0).
 * q_time device int32 [B]; hidden_dir [3*N_old, ld]; rela_dir [3*n_rela_rows, ld]; time_dir [3*n_time, ld].
 * Scratch as rg_layer_fwd_scratch_bytes(). */
int rg_tlayer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const int32_t* q_time,
                  const float* hidden_dir, const float* rela_dir, const float* time_dir, int32_t d, int32_t ld,
                  const float* a_s, const float* a_r, const float* a_q, int32_t ap,
                  const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                  float* agg_out, void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- temporal EXTRAPOLATION (Temporal/extrapolation/model_cuda_new_embedding.py:135-265): every query sees only the data rows of
 * its own time window, `self.dataset[time_offset_list[begin]:time_offset_list[cur_t]]` (:167-171), plus one self-loop per entity.
 * The graph is an rg_tgraph whose quadruples carry, in the time field, the index of the edge's row in the time-sorted data array
 * (self-loops: any value >= n_data).  rg_frontier_set_window gives the frontier the per-query row windows [win_lo[b], win_hi[b])
 * (device int32 [batch], caller-owned, NULL = no windows): rg_frontier_expand* then follow only edges valid for the query, and
 * rg_xlayer_fwd is the layer (:186-226) over those edges.  All edges lie in the past, so one direction matrix applies (past_linear):
 * hidden_p = W_past h [N_old, ld], rela_p = W_past rela [n_rela_rows, ld], time_p [n_tab, ld] = W_past time_embed(delta) for
 * delta = 0..n_tab-1; delta(edge, b) = q_time[b] - row_time[data row] (self-loops: q_time[b] - loop_time[b]), clamped to n_tab - 1.
 * Attention as rg_tlayer_fwd.  Its adjoint is rg_xlayer_bwd (below, with rg_tlayer_bwd). */
int rg_frontier_set_window(rg_frontier* f, const int32_t* win_lo_dev, const int32_t* win_hi_dev, int32_t n_data);
int rg_xlayer_fwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_new, const int32_t* q_time,
                  const int32_t* loop_time, const int32_t* row_time, int32_t n_data,
                  const float* hidden_p, const float* rela_p, const float* time_p, int32_t n_tab, int32_t d, int32_t ld,
                  const float* a_s, const float* a_r, const float* a_q, int32_t ap,
                  const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                  float* agg_out, void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- layer backward: adjoint of rg_layer_fwd (autograd of models.py:29-39) --------------------
 * grad_agg [N_new, ld].  grad_hidden [N_old, ld] and grad_a_s [N_old, ap] are WRITTEN (every row);
 * grad_rela [2R+1, ld], grad_a_r [2R+1, ap], grad_w_alpha [attn_dim], grad_b_alpha [1] are ACCUMULATED
 * into (caller zero-fills).  grad_a_q [B, ap] (may be NULL) is WRITTEN: the per-query segment sum of grad_a_s
 * (a_s[s] and a_q[b] enter the attention as a sum).  n_old is checked against the frontier.
 * scratch: rg_layer_bwd_scratch_bytes() bytes. */
size_t rg_layer_bwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld, int32_t ap);
int rg_layer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old,
                 const float* hidden, const float* rela, int32_t d, int32_t ld,
                 const float* a_s, const float* a_r, const float* a_q, int32_t ap,
                 const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                 const float* grad_agg,
                 float* grad_hidden, float* grad_rela, float* grad_a_s, float* grad_a_r, float* grad_a_q,
                 float* grad_w_alpha, float* grad_b_alpha,
                 void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- temporal layer backward: adjoint of rg_tlayer_fwd (autograd of Temporal/interpolation/model_cuda.py:149-160,192) ----
 * Arguments as rg_tlayer_fwd.  grad_hidden_dir [N_old, 3*ld] (row s = the three direction rows 3s..3s+2) and
 * grad_a_s [N_old, ap] are WRITTEN (every row); grad_rela_dir [3*n_rela_rows, ld], grad_time_dir [3*n_time, ld],
 * grad_a_r [n_rela_rows, ap], grad_w_alpha [attn_dim] are ACCUMULATED into (caller zero-fills); grad_a_q [B, ap] (may
 * be NULL) is WRITTEN, the per-query segment sum of grad_a_s.  The direction linears are differentiated by the caller
 * (three GEMMs on these sums).  Needs a graph from rg_tgraph_create. */
size_t rg_tlayer_bwd_scratch_bytes(const rg_frontier* f, const rg_graph* g, int32_t ld, int32_t ap);
int rg_tlayer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, const int32_t* q_time,
                  const float* hidden_dir, const float* rela_dir, const float* time_dir, int32_t d, int32_t ld,
                  const float* a_s, const float* a_r, const float* a_q, int32_t ap,
                  const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                  const float* grad_agg,
                  float* grad_hidden_dir, float* grad_rela_dir, float* grad_time_dir, float* grad_a_s, float* grad_a_r,
                  float* grad_a_q, float* grad_w_alpha,
                  void* scratch_dev, size_t scratch_bytes, void* stream);

/* Adjoint of rg_xlayer_fwd (training of the extrapolation setting, Temporal/extrapolation/main.py:296-320 around
 * model_cuda_new_embedding.py:186-239).  Arguments as rg_xlayer_fwd (the frontier still carries the batch's windows: rg_frontier_set_window).
 * grad_hidden_p [N_old, ld] and grad_a_s [N_old, ap] are WRITTEN; grad_rela_p [n_rela_rows, ld], grad_time_p [n_tab, ld],
 * grad_a_r [n_rela_rows, ap], grad_w_alpha [attn_dim] are ACCUMULATED into (caller zero-fills); grad_a_q [B, ap] (may be NULL) is
 * WRITTEN.  past_linear and the periodic time embedding are differentiated by the caller.  scratch: rg_tlayer_bwd_scratch_bytes(). */
int rg_xlayer_bwd(const rg_frontier* f, const rg_graph* g, int32_t level, int64_t n_old, const int32_t* q_time,
                  const int32_t* loop_time, const int32_t* row_time, int32_t n_data,
                  const float* hidden_p, const float* rela_p, const float* time_p, int32_t n_tab, int32_t d, int32_t ld,
                  const float* a_s, const float* a_r, const float* a_q, int32_t ap,
                  const float* w_alpha, const float* b_alpha, int32_t attn_dim,
                  const float* grad_agg,
                  float* grad_hidden_p, float* grad_rela_p, float* grad_time_p, float* grad_a_s, float* grad_a_r,
                  float* grad_a_q, float* grad_w_alpha,
                  void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- dense epilogue of a layer (inference): replaces models.py:41 (W_h + act), :81 (h0 index_copy_, as a
 * gather by prev_idx), :82-84 (single-step nn.GRU; dropout = identity in eval), the next layer's
 * Ws_attn projection (:36, hoisted per node) and, on the last layer, :86-88 (W_final + scatter into
 * scores_all).  One f32-MFMA kernel; weights in the layouts of the reference's state dict:
 * W_h [d,d], weight_ih_l0 / weight_hh_l0 [3d,d] (gate rows r,z,n), bias_* [3d], Ws_next [attn,d] or NULL,
 * W_final [d] or NULL.  agg [n,ld], hidden_prev [n_old,ld], prev_idx int32 [n] (-1 = new node; NULL = all
 * new), a_s_out [n,ap], nodes int32 [n,2], scores_all [B*n_ent] (pre-zeroed; only visited entries are
 * written), hidden_out [n,ld].  act: 0 identity, 1 relu, 2 tanh.  Supported: d <= 64 or d == 128, attn_dim <= 16
 * (rg_dense_fwd_supported); other shapes return an error and the caller keeps its own dense path. */
int rg_dense_fwd_supported(int32_t d, int32_t attn_dim);
int rg_dense_fwd(int64_t n, int32_t d, int32_t ld, const float* agg, const float* hidden_prev,
                 const int32_t* prev_idx, const float* W_h, int32_t act,
                 const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                 const float* Ws_next, int32_t attn_dim, int32_t ap, float* a_s_out,
                 const float* W_final, const int32_t* nodes, int32_t n_ent, float* scores_all,
                 float* hidden_out, int32_t precision, void* scratch, int64_t scratch_bytes, void* stream);
/* scratch: device memory of rg_dense_scratch_bytes(d, precision) bytes, 256-B aligned (0 bytes / NULL for every case but d = 128 with
 * precision 1 or 2, whose weights stream through LDS from a split image written there first).  Contents are dead after the call. */
int64_t rg_dense_scratch_bytes(int32_t d, int32_t precision);
/* precision: how the matrix products are evaluated.
 *   0  v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation (what the reference's fp32 GEMMs compute up to sum order)
 *   1  every fp32 operand as a two-term f16 split (22 significant bits, per-row power-of-two scaling), three
 *      v_mfma_f32_16x16x32_f16 per product with fp32 accumulation: errors of a few 1e-7 of a dot product's largest terms instead
 *      of 1e-7, at 3/16 of the matrix-pipe time
 *   2  every fp32 operand as an EXACT three-term f16 split (hi + mid + lo = all 24 bits; csrc/split3.h), the six partial products of
 *      order >= 2^-22 per product on v_mfma_f32_16x16x32_f16 / _bf8_bf8 with fp32 accumulation: fp32 arithmetic (operands exact,
 *      products good to 2^-31, fp32 sums) at 6/16 of the matrix-pipe time of precision 0.  The model's default. */

/* Test hook of precision 2's operand form: splits each of the n_rows rows of `cols` floats on the device exactly as the dense kernels
 * do (row scale = the power of two that takes the row's largest magnitude to [2^14, 2^15)) and writes back[n_rows, cols] =
 * (hi + mid + lo) / scale - bitwise equal to x for every element within 2^-15 of its row's largest magnitude (smaller ones: within 2^-39 of that largest) - and, if parts is not
 * NULL, parts[n_rows, cols, 4] = {hi, mid, lo, the bf8 (weights') form of lo}, scaled. */
int rg_split3_roundtrip(const float* x, int64_t n_rows, int32_t cols, float* back, float* parts, void* stream);
/* Test hook of precision 2's products: runs the d <= 64 kernel of rg_dense_fwd(precision = 2) on rows of d floats (ld = d) and writes
 * out[n, d] = one of its matrix products instead of the new state: which = 1: act(W_h agg); 2: weight_ih[2d:3d] x with
 * x = act(W_h agg); 3: weight_hh[2d:3d] h0 (h0 = hidden_prev gathered by prev_idx).  With one-hot operand rows (times powers of two)
 * a product is a column of the weight matrix and must come out bit for bit: the six partial products of csrc/split3.h are all there. */
int rg_split3_product_check(int32_t which, int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                            const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                            float* out, void* stream);

/* ---- weight gradients of the dense training step: out[m, n] = G^T X (row-major) for G [n_rows, m] (row stride ldg), X [n_rows, n] (row
 * stride ldx), n_rows in the millions, m <= 192, n <= 64 (wider products: call per column block) - autograd of models.py:41 (W_h),
 * :83 (weight_ih / weight_hh of the GRU) and the hoisted :36 Ws_attn.  colsum [m] (may be NULL) = column sums of G (the bias
 * gradients).  Exact fp32 products on v_mfma_f32_16x16x4_f32, rows read once, per-wave partial results added in a fixed order
 * (bitwise reproducible).  scratch: rg_gram_tn_scratch_bytes(m, n) bytes. */
size_t rg_gram_tn_scratch_bytes(int32_t m, int32_t n);
int rg_gram_tn(const float* g, int64_t ldg, int32_t m, const float* x, int64_t ldx, int32_t n, int64_t n_rows, float* out, float* colsum,
               void* scratch_dev, size_t scratch_bytes, void* stream);

/* ---- filtered ranking: replaces utils.py:7-14 cal_ranks (+ the filter loop base_model.py:107-115)
 * scores device fp32 [B, n_ent]; answers / filters as CSR over queries (device int32):
 * ans_ptr [B+1], ans_idx [ans_ptr[B]], filt_ptr [B+1], filt_idx [..].  ranks_out device fp32
 * [ans_ptr[B]] in (query, answer-list) order: rank = #{j not in filter: s'_j > s'_a} +
 * (#{j: s'_j == s'_a} + 1)/2 with s' = fl32(fl32(s - rowmin) + 1e-8). */
int rg_rank(const float* scores, int32_t batch, int32_t n_ent,
            const int32_t* ans_ptr, const int32_t* ans_idx,
            const int32_t* filt_ptr, const int32_t* filt_idx,
            float* ranks_out, void* stream);

/* rg_dense_fwd with the node count read on the device (after rg_frontier_expand_async): n_cap = capacity of the row buffers,
 * n_dev = rg_frontier_count_ptr() of the frontier whose newest level the rows belong to, n_hint = the row count the caller expects
 * (0 = unknown): it only sizes the grid, every row count up to n_cap is processed correctly. */
int rg_dense_fwd_dev(int64_t n_cap, const int32_t* n_dev, int64_t n_hint, int32_t d, int32_t ld, const float* agg, const float* hidden_prev,
                     const int32_t* prev_idx, const float* W_h, int32_t act,
                     const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                     const float* Ws_next, int32_t attn_dim, int32_t ap, float* a_s_out,
                     const float* W_final, const int32_t* nodes, int32_t n_ent, float* scores_all,
                     float* hidden_out, int32_t precision, void* scratch, int64_t scratch_bytes, void* stream);

/* ---- dense step of a layer in training: models.py:41 (W_h + act), :81 (h0 carry), :82 (dropout, as a given mask: 0 or
 * 1/(1-p) per element, or NULL), :83 (single-step GRU) in one f32-MFMA kernel that also leaves what the backward pass needs:
 * x_out [n,d] = the GRU input, and gates_ws_out [n,5,d] = {r, z, n, h0, W_hn h0 + b_hn}, the workspace layout of PyTorch's fused
 * GRU cell (so its fused backward kernel applies).  d in 16..64 (multiple of 4) or 128; rows are d floats wide (no padding). */
int rg_dense_train_fwd(int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                       const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       const float* mask, float* hidden_out, float* x_out, float* gates_ws_out, void* stream);

/* rg_dense_train_fwd that also emits the NEXT layer's hoisted attention projection of the new state, as rg_dense_fwd does in inference:
 * a_s_out [n, ap] = hidden_out Ws_next^T (Ws_next [attn_dim, d], models.py:16,33; columns attn_dim .. ap - 1 are zero; attn_dim <= 16,
 * ap = attn_dim rounded up to a multiple of 4). */
int rg_dense_train_fwd_as(int64_t n, int32_t d, const float* agg, const float* hidden_prev, const int32_t* prev_idx,
                          const float* W_h, int32_t act, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                          const float* mask, const float* Ws_next, int32_t attn_dim, int32_t ap,
                          float* hidden_out, float* x_out, float* gates_ws_out, float* a_s_out, void* stream);

/* out[r, :n] = base[r, :n] + g[r, :k] W[:k, :n] for n_rows node rows (row strides ldb / ldg / ldo in floats): the new state's gradient
 * through a_s = hidden Ws_attn^T added to the gradient it already carries (autograd of models.py:33).  k <= 32, n <= 128, n % 4 == 0;
 * out may alias base. */
int rg_rows_addmm(const float* base, int64_t ldb, const float* g, int64_t ldg, int32_t k, const float* W, int32_t n, int64_t n_rows,
                  float* out, int64_t ldo, void* stream);

/* Adjoint of rg_dense_train_fwd for the node-row quantities (autograd of models.py:41,81-83): from grad_hidden [n,d] and the saved
 * x / gates_ws (/ mask, keep = 1 - p) it writes grad_gates_i, grad_gates_h [n,3d] (GRU pre-activation gradients, for the weight and
 * bias gradients the caller forms), grad_pre [n,d] (gradient at W_h's output, for dW_h), grad_agg [n,d] and grad_h0 [n,d] (the
 * carried state's gradient, to be gathered back to the previous frontier by old_nodes_new_idx).  d in 16..64, multiple of 4. */
int rg_dense_train_bwd(int64_t n, int32_t d, const float* grad_hidden, const float* gates_ws, const float* x, const float* mask,
                       float keep, int32_t act, const float* W_h, const float* w_ih, const float* w_hh,
                       float* grad_gates_i, float* grad_gates_h, float* grad_pre, float* grad_agg, float* grad_h0, void* stream);

/* rg_dense_train_bwd with two fewer passes over memory: grad_gates_hn [n,d] is only the n-gate block of the hidden-side gate gradients
 * (their r and z blocks equal grad_gates_i's), and with prev_idx [n] (a node's row in the previous frontier or -1: rg_frontier_nodes)
 * the carried state's gradient goes straight to grad_prev [n_old,d] (every row written exactly once; autograd of models.py:81's
 * index_copy); prev_idx NULL: grad_prev is [n,d] = grad_h0. */
int rg_dense_train_bwd2(int64_t n, int32_t d, const float* grad_hidden, const float* gates_ws, const float* x, const float* mask,
                        float keep, int32_t act, const float* W_h, const float* w_ih, const float* w_hh, const int32_t* prev_idx,
                        float* grad_gates_i, float* grad_gates_hn, float* grad_pre, float* grad_agg, float* grad_prev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REDGNN_H */
