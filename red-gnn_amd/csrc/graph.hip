// Graph build + error plumbing + device scan.
// Replaces Static/transductive/load_data.py:69-81 (double_triple, load_graph): the KG with
// inverse and identity rows, held on the device as CSR-by-head and CSR-by-tail.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace rg {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ------------------------------------------------------------------------------------------
// Exclusive scan: reduce-then-scan, 256 threads x 8 items per block, recursive on block sums.
// ------------------------------------------------------------------------------------------
constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

template <bool POPC>
__device__ __forceinline__ int32_t scan_load(const uint32_t* in, int64_t i, int64_t n) {
  if (i >= n) return 0;
  uint32_t v = in[i];
  return POPC ? __popc(v) : (int32_t)v;
}

__device__ __forceinline__ int32_t wave_incl_scan(int32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *block_total valid in all threads
__device__ __forceinline__ int32_t block_excl_scan(int32_t v, int32_t* block_total) {
  __shared__ int32_t wsum[SCAN_T / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t inc = wave_incl_scan(v);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_T / 64; ++i) {
    int32_t x = wsum[i];
    if (i < w) base += x;
    tot += x;
  }
  __syncthreads();
  *block_total = tot;
  return base + inc - v;
}

template <bool POPC>
__global__ __launch_bounds__(SCAN_T) void scan_reduce_kernel(const uint32_t* in, int64_t n, int32_t* sums) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) acc += scan_load<POPC>(in, base + i, n);
  int32_t tot;
  block_excl_scan(acc, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// offsets == nullptr: single-block scan (n <= SCAN_TILE), writes *total.
template <bool POPC>
__global__ __launch_bounds__(SCAN_T) void scan_apply_kernel(const uint32_t* in, int32_t* out, int64_t n,
                                                            const int32_t* offsets, int32_t* total) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t acc = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = scan_load<POPC>(in, base + i, n);
    acc += v[i];
  }
  int32_t tot;
  int32_t ex = block_excl_scan(acc, &tot);
  if (offsets) ex += offsets[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
  if (!offsets && total && threadIdx.x == 0) *total = tot;
}

__global__ void zero_words_kernel(uint32_t* __restrict__ p, int64_t n_words) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) p[i] = 0u;
}
__global__ void zero_quads_kernel(uint4* __restrict__ p, int64_t n_quads) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}

int zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return 0;
  RG_CHECK(((uintptr_t)p & 3) == 0 && (bytes & 3) == 0, "zero_async: unaligned fill (%p, %zu B)", p, bytes);
  if (((uintptr_t)p & 15) == 0 && (bytes & 15) == 0 && bytes >= 4096) {
    const int64_t n = (int64_t)(bytes / 16);
    hipLaunchKernelGGL(zero_quads_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, (uint4*)p, n);
  } else {
    const int64_t n = (int64_t)(bytes / 4);
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, (uint32_t*)p, n);
  }
  RG_LAUNCH_CHECK();
  return 0;
}

__global__ void zero2_quads_kernel(uint4* __restrict__ p1, int64_t n1, uint4* __restrict__ p2, int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += stride) {
    if (i < n1) p1[i] = make_uint4(0u, 0u, 0u, 0u);
    else p2[i - n1] = make_uint4(0u, 0u, 0u, 0u);
  }
}

int zero2_async(void* p1, size_t bytes1, void* p2, size_t bytes2, hipStream_t s) {
  if ((((uintptr_t)p1 | (uintptr_t)p2) & 15) != 0 || ((bytes1 | bytes2) & 15) != 0) return zero_async(p1, bytes1, s) || zero_async(p2, bytes2, s);
  const int64_t n = (int64_t)((bytes1 + bytes2) / 16);
  if (n == 0) return 0;
  hipLaunchKernelGGL(zero2_quads_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 4096)), dim3(256), 0, s, (uint4*)p1,
                     (int64_t)(bytes1 / 16), (uint4*)p2, (int64_t)(bytes2 / 16));
  RG_LAUNCH_CHECK();
  return 0;
}

int copy_words_async(void* dst, const void* src, int n_words, hipStream_t s) {
  RG_CHECK(n_words > 0 && n_words <= 256, "copy_words_async: %d words", n_words);
  hipLaunchKernelGGL(copy_words_kernel, dim3(1), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, n_words);
  RG_LAUNCH_CHECK();
  return 0;
}

size_t scan_scratch_elems(int64_t n) {
  size_t tot = 0;
  while (n > SCAN_TILE) {
    n = ceil_div(n, SCAN_TILE);
    tot += align_up((size_t)n, 64) * 2;  // sums + scanned sums per level
  }
  return tot + 64;
}

int scan_exclusive(const uint32_t* in, int32_t* out, int64_t n, bool popc, int32_t* total_dev,
                   int32_t* scratch, hipStream_t s) {
  if (n <= 0) {
    if (total_dev && zero_async(total_dev, sizeof(int32_t), s)) return 1;
    return 0;
  }
  const int64_t nb = ceil_div(n, SCAN_TILE);
  if (nb == 1) {
    if (popc)
      hipLaunchKernelGGL(scan_apply_kernel<true>, dim3(1), dim3(SCAN_T), 0, s, in, out, n, nullptr, total_dev);
    else
      hipLaunchKernelGGL(scan_apply_kernel<false>, dim3(1), dim3(SCAN_T), 0, s, in, out, n, nullptr, total_dev);
    RG_LAUNCH_CHECK();
    return 0;
  }
  int32_t* sums = scratch;
  int32_t* sums_scanned = scratch + align_up((size_t)nb, 64);
  int32_t* next = sums_scanned + align_up((size_t)nb, 64);
  if (popc)
    hipLaunchKernelGGL(scan_reduce_kernel<true>, dim3(nb), dim3(SCAN_T), 0, s, in, n, sums);
  else
    hipLaunchKernelGGL(scan_reduce_kernel<false>, dim3(nb), dim3(SCAN_T), 0, s, in, n, sums);
  RG_LAUNCH_CHECK();
  if (scan_exclusive((const uint32_t*)sums, sums_scanned, nb, false, total_dev, next, s)) return 1;
  if (popc)
    hipLaunchKernelGGL(scan_apply_kernel<true>, dim3(nb), dim3(SCAN_T), 0, s, in, out, n, sums_scanned, nullptr);
  else
    hipLaunchKernelGGL(scan_apply_kernel<false>, dim3(nb), dim3(SCAN_T), 0, s, in, out, n, sums_scanned, nullptr);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace rg

// cut CSR rows into length-sorted virtual rows (see common.h); the arrays are uploaded with the rest of the graph
struct HostVrows { std::vector<int4> rows, split; };
static void host_vrows(const std::vector<int32_t>& ptr, int32_t n_ent, HostVrows* h, rg_vrows* out) {
  std::vector<int4>& rows = h->rows;
  std::vector<int4>& split = h->split;
  int32_t slots = 0;
  for (int32_t e = 0; e < n_ent; ++e) {
    const int32_t beg = ptr[e], len = ptr[e + 1] - ptr[e];
    if (len <= RG_VROW_MAX) {
      rows.push_back(make_int4(e, beg, len, -1));
    } else {
      const int32_t n_seg = (len + RG_VROW_MAX - 1) / RG_VROW_MAX;
      split.push_back(make_int4(e, slots, n_seg, 0));
      for (int32_t k = 0; k < n_seg; ++k) {
        const int32_t b = beg + k * RG_VROW_MAX;
        rows.push_back(make_int4(e, b, std::min(RG_VROW_MAX, beg + len - b), slots + k));
      }
      slots += n_seg;
    }
  }
  std::stable_sort(rows.begin(), rows.end(), [](const int4& a, const int4& b) { return a.z > b.z; });
  out->n = (int32_t)rows.size(); out->n_split = (int32_t)split.size(); out->n_slots = slots;
}

// bin-pack the virtual rows of the CSR-by-tail into packs of RG_PACK entries (see common.h): best fit over the rows in
// descending length; a segment of a cut row keeps a pack to itself
struct HostPacks { std::vector<int2> ent; std::vector<int4> pack; std::vector<int2> rows; };

namespace rg {
// place[i] = {pack, first entry inside the pack, row index inside the pack, 0} (pack = -1 for an empty row); returns the pack count.
// Best fit over the rows in their (descending length) order; inherently sequential, a few hundred microseconds for 40 k rows on a host
// core (one device thread needed 14.5 ms for the same loop: rg_graph_create_device ships the row descriptors here and back instead).
int place_rows_best_fit(const int4* vrows, int32_t n_vrows, std::vector<int4>* place, std::vector<int32_t>* pack_nrows) {
  std::vector<int32_t> open_by_room[RG_PACK + 1];       // packs with exactly that much room left (LIFO)
  std::vector<int32_t> fill;
  place->assign(n_vrows, make_int4(-1, 0, 0, 0));
  pack_nrows->clear();
  for (int32_t i = 0; i < n_vrows; ++i) {
    const int4& r = vrows[i];
    if (r.z <= 0) continue;
    int32_t p;
    if (r.w >= 0) { p = (int32_t)fill.size(); fill.push_back(0); pack_nrows->push_back(0); }
    else {
      int room = r.z;
      while (room <= RG_PACK && open_by_room[room].empty()) ++room;
      if (room > RG_PACK) { p = (int32_t)fill.size(); fill.push_back(0); pack_nrows->push_back(0); room = RG_PACK; }
      else { p = open_by_room[room].back(); open_by_room[room].pop_back(); }
      if (room - r.z > 0) open_by_room[room - r.z].push_back(p);
    }
    (*place)[i] = make_int4(p, fill[p], (*pack_nrows)[p], 0);
    fill[p] += r.z;
    (*pack_nrows)[p] += 1;
  }
  return (int)fill.size();
}
}  // namespace rg

static void host_packs(const std::vector<int4>& vrows, const std::vector<uint32_t>& in_pk, HostPacks* hp, rg_packs* out) {
  std::vector<int4> place;
  std::vector<int32_t> pack_nrows;
  const size_t n = (size_t)rg::place_rows_best_fit(vrows.data(), (int32_t)vrows.size(), &place, &pack_nrows);
  std::vector<int32_t> row0(n + 1, 0);
  for (size_t p = 0; p < n; ++p) row0[p + 1] = row0[p] + pack_nrows[p];
  hp->ent.assign(n * RG_PACK, make_int2(-1, 0));
  hp->pack.resize(n);
  hp->rows.resize(row0[n]);
  for (size_t i = 0; i < vrows.size(); ++i) {
    const int4& r = vrows[i];
    const int4& pl = place[i];
    if (pl.x < 0) continue;
    for (int32_t j = 0; j < r.z; ++j) hp->ent[(size_t)pl.x * RG_PACK + pl.y + j] = make_int2((int32_t)in_pk[r.y + j], pl.z);
    hp->rows[row0[pl.x] + pl.z] = make_int2(r.x, r.w);
    if (pl.z == 0) hp->pack[pl.x] = make_int4(row0[pl.x], pack_nrows[pl.x], r.w, r.x);
  }
  out->n = (int32_t)n;
}

// One device allocation per graph: every array is a slice of it, filled by one host-to-device copy.  (Two dozen hipMalloc /
// hipMemcpy pairs - and as many synchronising hipFree calls on destruction - were most of the cost of building a graph, which
// temporal training does once per batch.)
struct Arena {
  struct Item { void** field; const void* src; size_t bytes, off; };
  std::vector<Item> items;
  size_t total = 0;
  template <typename P, typename V>
  void add(P** field, const V& v) {
    const size_t bytes = v.size() * sizeof(typename V::value_type);
    items.push_back({(void**)field, (const void*)v.data(), bytes, total});
    total += rg::align_up(std::max<size_t>(bytes, 16), 256);
  }
};

extern "C" {

const char* rg_last_error(void) { return rg::g_err.c_str(); }
int rg_version(void) { return 10; }   // bumped whenever a kernel on the bench path changes: keys profiles/traffic_layer_fwd.json

// rows (H, R, T [, TIME]) -> device CSRs, packed entries, virtual rows
static int build_graph(int32_t n_ent, int32_t n_rel, int32_t n_rela_rows, const std::vector<int32_t>& H,
                       const std::vector<int32_t>& R, const std::vector<int32_t>& T, const std::vector<int32_t>* TIME,
                       int32_t n_time, rg_graph** out) {
  const int64_t n_fact = (int64_t)H.size();
  std::vector<int32_t> out_ptr(n_ent + 1, 0), in_ptr(n_ent + 1, 0);
  for (int64_t i = 0; i < n_fact; ++i) { out_ptr[H[i] + 1]++; in_ptr[T[i] + 1]++; }
  int32_t max_in = 0, max_out = 0;
  for (int32_t e = 0; e < n_ent; ++e) {
    max_out = out_ptr[e + 1] > max_out ? out_ptr[e + 1] : max_out;
    max_in = in_ptr[e + 1] > max_in ? in_ptr[e + 1] : max_in;
    out_ptr[e + 1] += out_ptr[e];
    in_ptr[e + 1] += in_ptr[e];
  }
  std::vector<int2> out_rt(n_fact), in_hr(n_fact);
  std::vector<int32_t> in_time(TIME ? n_fact : 0), out_time(TIME ? n_fact : 0);
  {
    std::vector<int32_t> po(out_ptr.begin(), out_ptr.end() - 1), pi(in_ptr.begin(), in_ptr.end() - 1);
    for (int64_t i = 0; i < n_fact; ++i) {  // stable: fact-row order inside each CSR row
      const int32_t q = pi[T[i]]++;
      in_hr[q] = make_int2(H[i], R[i]);
      if (TIME) in_time[q] = (*TIME)[i];
    }
    // CSR-by-head rows ordered by relation (then fact order): the backward kernel then sees runs of equal relation
    // and adds one partial sum per run (not per edge) to the privatised relation gradient.  Two stable counting
    // passes (by relation, then by head) instead of a comparison sort per row.
    std::vector<int32_t> by_rel(n_fact), out_fact(n_fact);
    {
      std::vector<int32_t> pr(n_rela_rows + 1, 0);
      for (int64_t i = 0; i < n_fact; ++i) pr[R[i] + 1]++;
      for (int32_t r = 0; r < n_rela_rows; ++r) pr[r + 1] += pr[r];
      for (int64_t i = 0; i < n_fact; ++i) by_rel[pr[R[i]]++] = (int32_t)i;
    }
    for (int64_t k = 0; k < n_fact; ++k) out_fact[po[H[by_rel[k]]]++] = by_rel[k];
    for (int64_t j = 0; j < n_fact; ++j) {
      const int32_t i = out_fact[j];
      out_rt[j] = make_int2(R[i], T[i]);
      if (TIME) out_time[j] = (*TIME)[i];
    }
  }
  rg_graph* g = new rg_graph();
  g->n_ent = n_ent; g->n_rel = n_rel; g->n_rela_rows = n_rela_rows; g->n_time = n_time;
  g->n_fact = n_fact; g->max_in_deg = max_in; g->max_out_deg = max_out;
  Arena A;
  A.add(&g->out_ptr, out_ptr); A.add(&g->in_ptr, in_ptr); A.add(&g->out_rt, out_rt); A.add(&g->in_hr, in_hr);
  if (TIME) { A.add(&g->in_time, in_time); A.add(&g->out_time, out_time); }
  std::vector<uint32_t> in_pk, out_pk;
  if (n_ent <= (1 << 20) && n_rela_rows <= (1 << 12)) {
    in_pk.resize(n_fact); out_pk.resize(n_fact);
    for (int64_t i = 0; i < n_fact; ++i) {
      in_pk[i] = ((uint32_t)in_hr[i].y << 20) | (uint32_t)in_hr[i].x;
      out_pk[i] = ((uint32_t)out_rt[i].x << 20) | (uint32_t)out_rt[i].y;
    }
    A.add(&g->in_pk, in_pk); A.add(&g->out_pk, out_pk);
  }
  // CSR by relation (and, for temporal graphs, by time id)
  std::vector<int32_t> rel_ptr(n_rela_rows + 1, 0), rel_tm(TIME ? n_fact : 0);
  std::vector<int2> rel_ht(n_fact);
  for (int64_t i = 0; i < n_fact; ++i) rel_ptr[R[i] + 1]++;
  for (int32_t r = 0; r < n_rela_rows; ++r) rel_ptr[r + 1] += rel_ptr[r];
  {
    std::vector<int32_t> pr(rel_ptr.begin(), rel_ptr.end() - 1);
    for (int64_t i = 0; i < n_fact; ++i) {
      const int32_t q = pr[R[i]]++;
      rel_ht[q] = make_int2(H[i], T[i]);
      if (TIME) rel_tm[q] = (*TIME)[i];
    }
  }
  A.add(&g->rel_ptr, rel_ptr); A.add(&g->rel_ht, rel_ht);
  std::vector<int32_t> time_ptr, time_rel;
  std::vector<int2> time_ht;
  HostVrows hv_in, hv_out, hv_rel, hv_time;
  if (TIME) {
    A.add(&g->rel_tm, rel_tm);
    time_ptr.assign(n_time + 1, 0); time_rel.resize(n_fact); time_ht.resize(n_fact);
    for (int64_t i = 0; i < n_fact; ++i) time_ptr[(*TIME)[i] + 1]++;
    for (int32_t t = 0; t < n_time; ++t) time_ptr[t + 1] += time_ptr[t];
    std::vector<int32_t> pt(time_ptr.begin(), time_ptr.end() - 1);
    for (int64_t i = 0; i < n_fact; ++i) {
      const int32_t q = pt[(*TIME)[i]]++;
      time_ht[q] = make_int2(H[i], T[i]);
      time_rel[q] = R[i];
    }
    A.add(&g->time_ptr, time_ptr); A.add(&g->time_ht, time_ht); A.add(&g->time_rel, time_rel);
    host_vrows(time_ptr, n_time, &hv_time, &g->time_vr);
    A.add(&g->time_vr.rows, hv_time.rows); A.add(&g->time_vr.split, hv_time.split);
  }
  host_vrows(in_ptr, n_ent, &hv_in, &g->in_vr);
  host_vrows(out_ptr, n_ent, &hv_out, &g->out_vr);
  host_vrows(rel_ptr, n_rela_rows, &hv_rel, &g->rel_vr);
  A.add(&g->in_vr.rows, hv_in.rows); A.add(&g->in_vr.split, hv_in.split);
  A.add(&g->out_vr.rows, hv_out.rows); A.add(&g->out_vr.split, hv_out.split);
  A.add(&g->rel_vr.rows, hv_rel.rows); A.add(&g->rel_vr.split, hv_rel.split);
  HostPacks hp;
  if (!TIME && !in_pk.empty() && n_rela_rows < (1 << 12)) {    // (rel 4095 | head 2^20-1 would read as the padding entry -1)      // (the temporal layer kernel has no word-parallel form; its graphs are rebuilt per batch)
    host_packs(hv_in.rows, in_pk, &hp, &g->in_pk_packs);
    A.add(&g->in_pk_packs.ent, hp.ent); A.add(&g->in_pk_packs.pack, hp.pack); A.add(&g->in_pk_packs.rows, hp.rows);
  }

  std::vector<char> staging(A.total);
  for (const Arena::Item& it : A.items)
    if (it.bytes) memcpy(staging.data() + it.off, it.src, it.bytes);
  hipError_t e = hipMalloc(&g->arena, A.total);
  if (e == hipSuccess) e = hipMemcpy(g->arena, staging.data(), A.total, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    rg::set_error("rg_graph_create: uploading %zu B failed: %s", A.total, hipGetErrorString(e));
    rg_graph_destroy(g);
    return 1;
  }
  for (const Arena::Item& it : A.items) *it.field = (char*)g->arena + it.off;
  *out = g;
  return 0;
}

int rg_graph_create(int32_t n_ent, int32_t n_rel, const int32_t* triples, int64_t n, int add_inverse,
                    rg_graph** out) {
  RG_CHECK(out != nullptr, "rg_graph_create: out is NULL");
  *out = nullptr;
  RG_CHECK(n_ent > 0 && n_rel > 0, "rg_graph_create: n_ent=%d n_rel=%d must be positive", n_ent, n_rel);
  RG_CHECK(n >= 0 && (n == 0 || triples != nullptr), "rg_graph_create: bad triples (n=%lld)", (long long)n);
  const int64_t n_fact = (add_inverse ? 2 * n : n) + n_ent;
  RG_CHECK(n_fact < (int64_t)1 << 31, "rg_graph_create: %lld fact rows do not fit int32", (long long)n_fact);
  // rows in the reference's order (load_data.py:69-80): triples, inverses, identity
  std::vector<int32_t> H(n_fact), R(n_fact), T(n_fact);
  const int32_t max_rel = add_inverse ? n_rel : 2 * n_rel;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t h = triples[3 * i], r = triples[3 * i + 1], t = triples[3 * i + 2];
    RG_CHECK(h >= 0 && h < n_ent && t >= 0 && t < n_ent && r >= 0 && r < max_rel,
             "rg_graph_create: triple %lld = (%d,%d,%d) out of range", (long long)i, h, r, t);
    H[i] = h; R[i] = r; T[i] = t;
    if (add_inverse) { H[n + i] = t; R[n + i] = r + n_rel; T[n + i] = h; }
  }
  const int64_t id0 = n_fact - n_ent;
  for (int32_t e = 0; e < n_ent; ++e) { H[id0 + e] = e; R[id0 + e] = 2 * n_rel; T[id0 + e] = e; }
  return build_graph(n_ent, n_rel, 2 * n_rel + 1, H, R, T, nullptr, 0, out);
}

static int tgraph_create(const char* who, int32_t n_ent, int32_t n_rela_rows, int32_t n_time, const int32_t* quads, int64_t n,
                         const int64_t* exclude, int64_t n_exclude, rg_graph** out) {
  RG_CHECK(out != nullptr, "%s: out is NULL", who);
  *out = nullptr;
  RG_CHECK(n_ent > 0 && n_rela_rows > 0 && n_time > 0, "%s: n_ent=%d n_rela_rows=%d n_time=%d must be positive", who, n_ent, n_rela_rows, n_time);
  RG_CHECK(n >= 0 && n < ((int64_t)1 << 31) && (n == 0 || quads != nullptr), "%s: bad quadruples (n=%lld)", who, (long long)n);
  RG_CHECK(n_exclude >= 0 && (n_exclude == 0 || exclude != nullptr), "%s: bad exclusion list", who);
  std::vector<char> drop(n_exclude ? n : 0, 0);
  for (int64_t k = 0; k < n_exclude; ++k) {
    RG_CHECK(exclude[k] >= 0 && exclude[k] < n, "%s: excluded row %lld out of range (n=%lld)", who, (long long)exclude[k], (long long)n);
    drop[exclude[k]] = 1;
  }
  std::vector<int32_t> H, R, T, TM;
  H.reserve(n); R.reserve(n); T.reserve(n); TM.reserve(n);
  for (int64_t i = 0; i < n; ++i) {
    if (n_exclude && drop[i]) continue;
    const int32_t h = quads[4 * i], r = quads[4 * i + 1], t = quads[4 * i + 2], tm = quads[4 * i + 3];
    RG_CHECK(h >= 0 && h < n_ent && t >= 0 && t < n_ent && r >= 0 && r < n_rela_rows && tm >= 0 && tm < n_time,
             "%s: quadruple %lld = (%d,%d,%d,%d) out of range", who, (long long)i, h, r, t, tm);
    H.push_back(h); R.push_back(r); T.push_back(t); TM.push_back(tm);
  }
  return build_graph(n_ent, 0, n_rela_rows, H, R, T, &TM, n_time, out);
}

int rg_tgraph_create(int32_t n_ent, int32_t n_rela_rows, int32_t n_time, const int32_t* quads, int64_t n, rg_graph** out) {
  return tgraph_create("rg_tgraph_create", n_ent, n_rela_rows, n_time, quads, n, nullptr, 0, out);
}

int rg_tgraph_create_excluding(int32_t n_ent, int32_t n_rela_rows, int32_t n_time, const int32_t* quads, int64_t n,
                               const int64_t* exclude_rows, int64_t n_exclude, rg_graph** out) {
  return tgraph_create("rg_tgraph_create_excluding", n_ent, n_rela_rows, n_time, quads, n, exclude_rows, n_exclude, out);
}

int rg_graph_destroy(rg_graph* g) {
  if (!g) return 0;
  if (g->arena) (void)hipFree(g->arena);      // every array of the graph is a slice of this one allocation
  delete g;
  return 0;
}

int64_t rg_graph_n_fact(const rg_graph* g) { return g ? g->n_fact : -1; }

int rg_graph_export(const rg_graph* g, int32_t* out_ptr, int32_t* out_rt, int32_t* in_ptr, int32_t* in_hr) {
  RG_CHECK(g != nullptr, "rg_graph_export: graph is NULL");
  if (out_ptr) RG_HIP(hipMemcpy(out_ptr, g->out_ptr, (g->n_ent + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (in_ptr) RG_HIP(hipMemcpy(in_ptr, g->in_ptr, (g->n_ent + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (out_rt) RG_HIP(hipMemcpy(out_rt, g->out_rt, g->n_fact * sizeof(int2), hipMemcpyDeviceToHost));
  if (in_hr) RG_HIP(hipMemcpy(in_hr, g->in_hr, g->n_fact * sizeof(int2), hipMemcpyDeviceToHost));
  return 0;
}

// number of memset / memcpy nodes of a captured hipGraph (models._GraphedInference asserts 0: with ROCm 7.2 a replayed graph that holds
// several memset nodes zeroed correctly on its first launch only, see common.h zero_async); -1 on error
int rg_hipgraph_fill_nodes(void* graph) {
  size_t n = 0;
  if (hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) { rg::set_error("rg_hipgraph_fill_nodes: hipGraphGetNodes failed"); return -1; }
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes((hipGraph_t)graph, nodes.data(), &n) != hipSuccess) { rg::set_error("rg_hipgraph_fill_nodes: hipGraphGetNodes failed"); return -1; }
  int count = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { rg::set_error("rg_hipgraph_fill_nodes: hipGraphNodeGetType failed"); return -1; }
    if (t == hipGraphNodeTypeMemset) ++count;
  }
  return count;
}

int rg_graph_export_packs(const rg_graph* g, int32_t* n_packs, int32_t* n_vrows, int32_t* ent, int32_t* pack, int32_t* rows, int32_t* vrows) {
  RG_CHECK(g != nullptr, "rg_graph_export_packs: graph is NULL");
  if (n_packs) *n_packs = g->in_pk_packs.n;
  if (n_vrows) *n_vrows = g->in_vr.n;
  if (ent && g->in_pk_packs.n) RG_HIP(hipMemcpy(ent, g->in_pk_packs.ent, (size_t)g->in_pk_packs.n * RG_PACK * sizeof(int2), hipMemcpyDeviceToHost));
  if (pack && g->in_pk_packs.n) RG_HIP(hipMemcpy(pack, g->in_pk_packs.pack, (size_t)g->in_pk_packs.n * sizeof(int4), hipMemcpyDeviceToHost));
  if (rows && g->in_pk_packs.n) RG_HIP(hipMemcpy(rows, g->in_pk_packs.rows, (size_t)g->in_vr.n * sizeof(int2), hipMemcpyDeviceToHost));
  if (vrows) RG_HIP(hipMemcpy(vrows, g->in_vr.rows, (size_t)g->in_vr.n * sizeof(int4), hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
