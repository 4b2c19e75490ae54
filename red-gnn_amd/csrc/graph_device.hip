// Graph build on the device: the same rg_graph as rg_graph_create (graph.hip), from a DEVICE triple array.
// Replaces the per-epoch rebuild of DataLoader.shuffle_train (Static/transductive/load_data.py:152-164: permute facts + train,
// re-split 3:1, load_graph :76-81) without copying the triples to the host and back: rows (triples, inverses, identity), the
// three CSRs (by tail in fact order; by head ordered by relation then fact order; by relation), packed entries, length-sorted
// virtual rows and the word-parallel walk's packs are produced by stable radix sorts (hipCUB), scans and a few kernels; one
// small read-back per CSR returns the counts that size the arena.  Every array equals the host builder's bit for bit (test:
// test_device_graph_build_equals_host_build), including the packs: their best-fit placement is inherently sequential, so the 16-byte
// descriptors of the length-sorted rows (0.6 MB for 40 k rows) visit the host for it - the triples and the CSR arrays never do.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

struct Tmp {                      // temporaries of one build: slices of one device block (hipMalloc / hipFree synchronise and cost
  std::vector<void*> ptrs;        // ~50 us each; sixty of them were half of the build), individual allocations beyond it
  char* block = nullptr;
  size_t block_bytes = 0, used = 0;
  explicit Tmp(size_t bytes) {
    if (hipMalloc((void**)&block, bytes) == hipSuccess) block_bytes = bytes; else block = nullptr;
  }
  template <typename T>
  T* get(size_t n) {
    const size_t bytes = rg::align_up(std::max<size_t>(n, 1) * sizeof(T), 256);
    if (used + bytes <= block_bytes) { T* p = (T*)(block + used); used += bytes; return p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return (T*)p;
  }
  ~Tmp() {
    for (void* p : ptrs) (void)hipFree(p);
    if (block) (void)hipFree(block);
  }
};

__global__ void rows_kernel(const int32_t* __restrict__ trip, int64_t n, int add_inverse, int n_ent, int n_rel, int max_rel,
                            int32_t* __restrict__ H, int32_t* __restrict__ R, int32_t* __restrict__ T, int64_t n_fact, int32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_fact) return;
  const int64_t n_dir = add_inverse ? 2 * n : n;
  if (i < n_dir) {                       // rows in the reference's order (load_data.py:69-80): triples, inverses, identity
    const int64_t j = i < n ? i : i - n;
    const int32_t h = trip[3 * j], r = trip[3 * j + 1], t = trip[3 * j + 2];
    if (h < 0 || h >= n_ent || t < 0 || t >= n_ent || r < 0 || r >= max_rel) { atomicMax(err, 1 + (int32_t)std::min<int64_t>(j, 0x7ffffffe)); H[i] = R[i] = T[i] = 0; return; }
    if (i < n) { H[i] = h; R[i] = r; T[i] = t; }
    else { H[i] = t; R[i] = r + n_rel; T[i] = h; }
  } else {
    const int32_t e = (int32_t)(i - n_dir);
    H[i] = e; R[i] = 2 * n_rel; T[i] = e;
  }
}

__global__ void iota_kernel(uint32_t* __restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

template <int MODE>   // key of row i: 0 = tail, 1 = (head, relation), 2 = relation
__global__ void keys_kernel(const int32_t* __restrict__ H, const int32_t* __restrict__ R, const int32_t* __restrict__ T, int rel_bits,
                            uint64_t* __restrict__ keys, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = MODE == 0 ? (uint64_t)T[i] : (MODE == 1 ? (((uint64_t)H[i] << rel_bits) | (uint64_t)R[i]) : (uint64_t)R[i]);
}

__global__ void hist_kernel(const int32_t* __restrict__ key, int64_t n, uint32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&cnt[key[i]], 1u);
}

__global__ void gather_pairs_kernel(const uint32_t* __restrict__ perm, const int32_t* __restrict__ A, const int32_t* __restrict__ B,
                                    int2* __restrict__ out, uint32_t* __restrict__ pk, int pk_mode, int64_t n) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const uint32_t i = perm[q];
  const int2 v = make_int2(A[i], B[i]);
  out[q] = v;
  if (pk) pk[q] = pk_mode == 0 ? (((uint32_t)v.y << 20) | (uint32_t)v.x)      // in:  (rel << 20 | head) from {head, rel}
                               : (((uint32_t)v.x << 20) | (uint32_t)v.y);     // out: (rel << 20 | tail) from {rel, tail}
}

// ---- virtual rows (graph.hip host_vrows): per entity its segment count / whether it is cut, then rows and split entries --------
__global__ void vrow_count_kernel(const int32_t* __restrict__ ptr, int n_rows, uint32_t* __restrict__ n_seg, uint32_t* __restrict__ is_split,
                                  uint32_t* __restrict__ split_seg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rows) return;
  const int len = ptr[e + 1] - ptr[e];
  const int cut = len > RG_VROW_MAX;
  const int k = cut ? (len + RG_VROW_MAX - 1) / RG_VROW_MAX : 1;
  n_seg[e] = k; is_split[e] = cut; split_seg[e] = cut ? k : 0;
}

__global__ void vrow_emit_kernel(const int32_t* __restrict__ ptr, int n_rows, const int32_t* __restrict__ row_off, const int32_t* __restrict__ split_off,
                                 const int32_t* __restrict__ slot_off, int4* __restrict__ rows, uint32_t* __restrict__ keys, int4* __restrict__ split) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rows) return;
  const int beg = ptr[e], len = ptr[e + 1] - ptr[e];
  if (len <= RG_VROW_MAX) {
    rows[row_off[e]] = make_int4(e, beg, len, -1);
    keys[row_off[e]] = (uint32_t)(RG_VROW_MAX - len);
  } else {
    const int k = (len + RG_VROW_MAX - 1) / RG_VROW_MAX;
    split[split_off[e]] = make_int4(e, slot_off[e], k, 0);
    for (int j = 0; j < k; ++j) {
      const int b = beg + j * RG_VROW_MAX, l = min(RG_VROW_MAX, beg + len - b);
      rows[row_off[e] + j] = make_int4(e, b, l, slot_off[e] + j);
      keys[row_off[e] + j] = (uint32_t)(RG_VROW_MAX - l);
    }
  }
}

__global__ void gather_rows_kernel(const uint32_t* __restrict__ perm, const int4* __restrict__ in, int4* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

// ---- packs (graph.hip host_packs): the best-fit placement is sequential; the 16-byte descriptors of the length-sorted rows go to the
// host for it (rg::place_rows_best_fit, the host builder's own routine) and the placements come back; entries are written here ---------
__global__ void pack_emit_kernel(const int4* __restrict__ vrows, const int4* __restrict__ place, int n_vrows, const int32_t* __restrict__ pack_row0,
                                 const uint32_t* __restrict__ in_pk, int2* __restrict__ ent, int4* __restrict__ pack, int2* __restrict__ rows) {
  const int i = blockIdx.x;                  // one workgroup per virtual row
  if (i >= n_vrows) return;
  const int4 r = vrows[i];
  const int4 pl = place[i];
  if (pl.x < 0) return;
  for (int j = threadIdx.x; j < r.z; j += blockDim.x) ent[(int64_t)pl.x * RG_PACK + pl.y + j] = make_int2((int)in_pk[r.y + j], pl.z);
  if (threadIdx.x == 0) {
    rows[pack_row0[pl.x] + pl.z] = make_int2(r.x, r.w);
    if (pl.z == 0) pack[pl.x] = make_int4(pack_row0[pl.x], 0, r.w, r.x);       // .y (row count) is filled by pack_count_kernel
  }
}

__global__ void pack_count_kernel(const int32_t* __restrict__ pack_nrows, int4* __restrict__ pack, int n_packs) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_packs) pack[p].y = pack_nrows[p];
}

__global__ void fill_int2_kernel(int2* __restrict__ p, int64_t n, int2 v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void max_kernel(const int32_t* __restrict__ ptr, int n_rows, int32_t* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_rows) atomicMax(out, ptr[e + 1] - ptr[e]);
}

#define DEV_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      ::rg::set_error("rg_graph_create_device: %s failed: %s", #expr, hipGetErrorString(e_)); \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

inline unsigned blocks(int64_t n, int t = 256) { return (unsigned)std::max<int64_t>(rg::ceil_div(n, t), 1); }

// stable sort of the row indices by a key of `bits` bits; perm_out receives the sorted indices
static int sort_perm(Tmp& tmp, const uint64_t* keys, int bits, int64_t n, uint32_t* perm_out, hipStream_t s) {
  uint64_t* keys_out = tmp.get<uint64_t>(n);
  uint32_t* iota = tmp.get<uint32_t>(n);
  if (!keys_out || !iota) { rg::set_error("rg_graph_create_device: out of device memory"); return 1; }
  hipLaunchKernelGGL(iota_kernel, dim3(blocks(n)), dim3(256), 0, s, iota, n);
  size_t bytes = 0;
  DEV_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, keys_out, iota, perm_out, (int)n, 0, bits, s));
  char* scratch = tmp.get<char>(bytes);
  if (!scratch) { rg::set_error("rg_graph_create_device: out of device memory"); return 1; }
  DEV_HIP(hipcub::DeviceRadixSort::SortPairs(scratch, bytes, keys, keys_out, iota, perm_out, (int)n, 0, bits, s));
  return 0;
}

// ptr[0..n_rows] = exclusive scan of the histogram of key over [0, n_rows)
static int build_ptr(Tmp& tmp, const int32_t* key, int64_t n, int n_rows, int32_t* ptr, hipStream_t s) {
  uint32_t* cnt = tmp.get<uint32_t>(n_rows + 1);
  int32_t* scr = tmp.get<int32_t>(rg::scan_scratch_elems(n_rows + 1));
  if (!cnt || !scr) { rg::set_error("rg_graph_create_device: out of device memory"); return 1; }
  DEV_HIP(hipMemsetAsync(cnt, 0, (size_t)(n_rows + 1) * 4, s));
  hipLaunchKernelGGL(hist_kernel, dim3(blocks(n)), dim3(256), 0, s, key, n, cnt);
  return rg::scan_exclusive(cnt, ptr, n_rows + 1, false, nullptr, scr, s);
}

struct VrowsDev { int4* rows = nullptr; int4* split = nullptr; int32_t* counts = nullptr; int cap_rows = 0, cap_split = 0; };

// virtual rows of one CSR into temporaries; counts (device) = {n_rows, n_split, n_slots}
static int build_vrows(Tmp& tmp, const int32_t* ptr, int n_rows, int64_t n_fact, VrowsDev* out, hipStream_t s) {
  uint32_t* n_seg = tmp.get<uint32_t>(n_rows + 1);
  uint32_t* is_split = tmp.get<uint32_t>(n_rows + 1);
  uint32_t* split_seg = tmp.get<uint32_t>(n_rows + 1);
  int32_t* row_off = tmp.get<int32_t>(n_rows + 1);
  int32_t* split_off = tmp.get<int32_t>(n_rows + 1);
  int32_t* slot_off = tmp.get<int32_t>(n_rows + 1);
  int32_t* scr = tmp.get<int32_t>(rg::scan_scratch_elems(n_rows + 1));
  out->cap_rows = (int)(n_rows + n_fact / RG_VROW_MAX + 2);
  out->cap_split = (int)(n_fact / (RG_VROW_MAX + 1) + 2);
  int4* rows_u = tmp.get<int4>(out->cap_rows);
  uint32_t* keys = tmp.get<uint32_t>(out->cap_rows);
  uint32_t* keys_o = tmp.get<uint32_t>(out->cap_rows);
  uint32_t* iota = tmp.get<uint32_t>(out->cap_rows);
  uint32_t* perm = tmp.get<uint32_t>(out->cap_rows);
  out->rows = tmp.get<int4>(out->cap_rows);
  out->split = tmp.get<int4>(out->cap_split);
  out->counts = tmp.get<int32_t>(4);
  if (!n_seg || !is_split || !split_seg || !row_off || !split_off || !slot_off || !scr || !rows_u || !keys || !keys_o || !iota || !perm ||
      !out->rows || !out->split || !out->counts) { rg::set_error("rg_graph_create_device: out of device memory"); return 1; }
  hipLaunchKernelGGL(vrow_count_kernel, dim3(blocks(n_rows)), dim3(256), 0, s, ptr, n_rows, n_seg, is_split, split_seg);
  if (rg::scan_exclusive(n_seg, row_off, n_rows, false, &out->counts[0], scr, s)) return 1;
  if (rg::scan_exclusive(is_split, split_off, n_rows, false, &out->counts[1], scr, s)) return 1;
  if (rg::scan_exclusive(split_seg, slot_off, n_rows, false, &out->counts[2], scr, s)) return 1;
  hipLaunchKernelGGL(vrow_emit_kernel, dim3(blocks(n_rows)), dim3(256), 0, s, ptr, n_rows, row_off, split_off, slot_off, rows_u, keys, out->split);
  // the row count is needed by the sort: it is bounded by cap_rows, and unused tail keys sort last (key = max)
  int32_t n_v = 0;
  DEV_HIP(hipMemcpyAsync(&n_v, &out->counts[0], 4, hipMemcpyDeviceToHost, s));
  DEV_HIP(hipStreamSynchronize(s));
  hipLaunchKernelGGL(iota_kernel, dim3(blocks(n_v)), dim3(256), 0, s, iota, (int64_t)n_v);
  size_t bytes = 0;
  DEV_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, keys_o, iota, perm, n_v, 0, 8, s));
  char* scratch = tmp.get<char>(bytes);
  if (!scratch) { rg::set_error("rg_graph_create_device: out of device memory"); return 1; }
  DEV_HIP(hipcub::DeviceRadixSort::SortPairs(scratch, bytes, keys, keys_o, iota, perm, n_v, 0, 8, s));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks(n_v)), dim3(256), 0, s, perm, rows_u, out->rows, n_v);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int rg_graph_create_device(int32_t n_ent, int32_t n_rel, const int32_t* triples_dev, int64_t n, int add_inverse, void* stream,
                                      rg_graph** out) {
  RG_CHECK(out != nullptr, "rg_graph_create_device: out is NULL");
  *out = nullptr;
  RG_CHECK(n_ent > 0 && n_rel > 0, "rg_graph_create_device: n_ent=%d n_rel=%d must be positive", n_ent, n_rel);
  RG_CHECK(n >= 0 && (n == 0 || triples_dev != nullptr), "rg_graph_create_device: bad triples (n=%lld)", (long long)n);
  const int64_t n_fact = (add_inverse ? 2 * n : n) + n_ent;
  RG_CHECK(n_fact < (int64_t)1 << 31, "rg_graph_create_device: %lld fact rows do not fit int32", (long long)n_fact);
  hipStream_t s = (hipStream_t)stream;
  const int n_rela_rows = 2 * n_rel + 1;
  const int max_rel = add_inverse ? n_rel : 2 * n_rel;
  int rel_bits = 1, ent_bits = 1;
  while ((1 << rel_bits) < n_rela_rows) ++rel_bits;
  while (((int64_t)1 << ent_bits) < n_ent) ++ent_bits;
  Tmp tmp((size_t)n_fact * 160 + (size_t)n_ent * 96 + ((size_t)8 << 20));       // (rows, keys, permutations, CSRs, sort scratch, virtual rows)
  int32_t* H = tmp.get<int32_t>(n_fact); int32_t* R = tmp.get<int32_t>(n_fact); int32_t* T = tmp.get<int32_t>(n_fact);
  uint64_t* keys = tmp.get<uint64_t>(n_fact);
  uint32_t* perm_in = tmp.get<uint32_t>(n_fact); uint32_t* perm_out = tmp.get<uint32_t>(n_fact); uint32_t* perm_rel = tmp.get<uint32_t>(n_fact);
  int32_t* flags = tmp.get<int32_t>(16);      // [0] error, [1] max in-degree, [2] max out-degree, [3] n_packs
  int32_t* out_ptr = tmp.get<int32_t>(n_ent + 2); int32_t* in_ptr = tmp.get<int32_t>(n_ent + 2); int32_t* rel_ptr = tmp.get<int32_t>(n_rela_rows + 2);
  int2* out_rt = tmp.get<int2>(n_fact); int2* in_hr = tmp.get<int2>(n_fact); int2* rel_ht = tmp.get<int2>(n_fact);
  const bool packed = n_ent <= (1 << 20) && n_rela_rows <= (1 << 12);
  uint32_t* in_pk = packed ? tmp.get<uint32_t>(n_fact) : nullptr; uint32_t* out_pk = packed ? tmp.get<uint32_t>(n_fact) : nullptr;
  RG_CHECK(H && R && T && keys && perm_in && perm_out && perm_rel && flags && out_ptr && in_ptr && rel_ptr && out_rt && in_hr && rel_ht &&
           (!packed || (in_pk && out_pk)), "rg_graph_create_device: out of device memory");
  DEV_HIP(hipMemsetAsync(flags, 0, 64, s));
  hipLaunchKernelGGL(rows_kernel, dim3(blocks(n_fact)), dim3(256), 0, s, triples_dev, n, add_inverse, n_ent, n_rel, max_rel, H, R, T, n_fact, flags);
  // CSR by tail (fact order inside a row), by head (relation, then fact order), by relation (fact order)
  hipLaunchKernelGGL(keys_kernel<0>, dim3(blocks(n_fact)), dim3(256), 0, s, H, R, T, rel_bits, keys, n_fact);
  if (sort_perm(tmp, keys, ent_bits, n_fact, perm_in, s)) return 1;
  hipLaunchKernelGGL(keys_kernel<1>, dim3(blocks(n_fact)), dim3(256), 0, s, H, R, T, rel_bits, keys, n_fact);
  if (sort_perm(tmp, keys, ent_bits + rel_bits, n_fact, perm_out, s)) return 1;
  hipLaunchKernelGGL(keys_kernel<2>, dim3(blocks(n_fact)), dim3(256), 0, s, H, R, T, rel_bits, keys, n_fact);
  if (sort_perm(tmp, keys, rel_bits, n_fact, perm_rel, s)) return 1;
  hipLaunchKernelGGL(gather_pairs_kernel, dim3(blocks(n_fact)), dim3(256), 0, s, perm_in, H, R, in_hr, in_pk, 0, n_fact);
  hipLaunchKernelGGL(gather_pairs_kernel, dim3(blocks(n_fact)), dim3(256), 0, s, perm_out, R, T, out_rt, out_pk, 1, n_fact);
  hipLaunchKernelGGL(gather_pairs_kernel, dim3(blocks(n_fact)), dim3(256), 0, s, perm_rel, H, T, rel_ht, (uint32_t*)nullptr, 0, n_fact);
  if (build_ptr(tmp, T, n_fact, n_ent, in_ptr, s) || build_ptr(tmp, H, n_fact, n_ent, out_ptr, s) || build_ptr(tmp, R, n_fact, n_rela_rows, rel_ptr, s)) return 1;
  hipLaunchKernelGGL(max_kernel, dim3(blocks(n_ent)), dim3(256), 0, s, in_ptr, n_ent, &flags[1]);
  hipLaunchKernelGGL(max_kernel, dim3(blocks(n_ent)), dim3(256), 0, s, out_ptr, n_ent, &flags[2]);
  VrowsDev vin, vout, vrel;
  if (build_vrows(tmp, in_ptr, n_ent, n_fact, &vin, s) || build_vrows(tmp, out_ptr, n_ent, n_fact, &vout, s) ||
      build_vrows(tmp, rel_ptr, n_rela_rows, n_fact, &vrel, s)) return 1;
  int32_t host[16];
  DEV_HIP(hipMemcpyAsync(host, vin.counts, 12, hipMemcpyDeviceToHost, s));
  DEV_HIP(hipMemcpyAsync(host + 4, vout.counts, 12, hipMemcpyDeviceToHost, s));
  DEV_HIP(hipMemcpyAsync(host + 8, vrel.counts, 12, hipMemcpyDeviceToHost, s));
  DEV_HIP(hipMemcpyAsync(host + 12, flags, 16, hipMemcpyDeviceToHost, s));
  DEV_HIP(hipStreamSynchronize(s));
  RG_CHECK(host[12] == 0, "rg_graph_create_device: triple %d out of range", host[12] - 1);
  // packs of the word-parallel walk
  const bool want_packs = packed && n_rela_rows < (1 << 12);
  int4* place = nullptr; int32_t* pack_nrows = nullptr; int32_t* pack_row0 = nullptr;
  int n_packs = 0;
  if (want_packs) {
    const int nv = host[0];
    place = tmp.get<int4>(nv);
    pack_nrows = tmp.get<int32_t>(nv + 1);
    pack_row0 = tmp.get<int32_t>(nv + 1);
    int32_t* scr = tmp.get<int32_t>(rg::scan_scratch_elems(nv + 1));
    RG_CHECK(place && pack_nrows && pack_row0 && scr, "rg_graph_create_device: out of device memory");
    std::vector<int4> vrows_h(nv), place_h;
    std::vector<int32_t> nrows_h;
    DEV_HIP(hipMemcpyAsync(vrows_h.data(), vin.rows, (size_t)nv * sizeof(int4), hipMemcpyDeviceToHost, s));
    DEV_HIP(hipStreamSynchronize(s));
    n_packs = rg::place_rows_best_fit(vrows_h.data(), nv, &place_h, &nrows_h);
    DEV_HIP(hipMemcpyAsync(place, place_h.data(), (size_t)nv * sizeof(int4), hipMemcpyHostToDevice, s));
    if (n_packs) DEV_HIP(hipMemcpyAsync(pack_nrows, nrows_h.data(), (size_t)n_packs * sizeof(int32_t), hipMemcpyHostToDevice, s));
    DEV_HIP(hipStreamSynchronize(s));          // (the host vectors go out of scope)
    if (rg::scan_exclusive((const uint32_t*)pack_nrows, pack_row0, n_packs, false, nullptr, scr, s)) return 1;
  }
  // ---- the arena: same slices as the host builder's ----------------------------------------------------------------------
  rg_graph* g = new rg_graph();
  g->n_ent = n_ent; g->n_rel = n_rel; g->n_rela_rows = n_rela_rows; g->n_time = 0; g->n_fact = n_fact;
  g->max_in_deg = host[13]; g->max_out_deg = host[14];
  g->in_vr.n = host[0]; g->in_vr.n_split = host[1]; g->in_vr.n_slots = host[2];
  g->out_vr.n = host[4]; g->out_vr.n_split = host[5]; g->out_vr.n_slots = host[6];
  g->rel_vr.n = host[8]; g->rel_vr.n_split = host[9]; g->rel_vr.n_slots = host[10];
  g->in_pk_packs.n = n_packs;
  struct Item { void** field; const void* src; size_t bytes, off; };
  std::vector<Item> items;
  size_t total = 0;
  auto add = [&](void** field, const void* src, size_t bytes) {
    items.push_back({field, src, bytes, total});
    total += rg::align_up(std::max<size_t>(bytes, 16), 256);
  };
  add((void**)&g->out_ptr, out_ptr, (size_t)(n_ent + 1) * 4); add((void**)&g->in_ptr, in_ptr, (size_t)(n_ent + 1) * 4);
  add((void**)&g->out_rt, out_rt, (size_t)n_fact * 8); add((void**)&g->in_hr, in_hr, (size_t)n_fact * 8);
  if (packed) { add((void**)&g->in_pk, in_pk, (size_t)n_fact * 4); add((void**)&g->out_pk, out_pk, (size_t)n_fact * 4); }
  add((void**)&g->rel_ptr, rel_ptr, (size_t)(n_rela_rows + 1) * 4); add((void**)&g->rel_ht, rel_ht, (size_t)n_fact * 8);
  add((void**)&g->in_vr.rows, vin.rows, (size_t)g->in_vr.n * 16); add((void**)&g->in_vr.split, vin.split, (size_t)g->in_vr.n_split * 16);
  add((void**)&g->out_vr.rows, vout.rows, (size_t)g->out_vr.n * 16); add((void**)&g->out_vr.split, vout.split, (size_t)g->out_vr.n_split * 16);
  add((void**)&g->rel_vr.rows, vrel.rows, (size_t)g->rel_vr.n * 16); add((void**)&g->rel_vr.split, vrel.split, (size_t)g->rel_vr.n_split * 16);
  const size_t ent_off = total;
  if (n_packs > 0) {
    add((void**)&g->in_pk_packs.ent, nullptr, (size_t)n_packs * RG_PACK * 8);
    add((void**)&g->in_pk_packs.pack, nullptr, (size_t)n_packs * 16);
    add((void**)&g->in_pk_packs.rows, nullptr, (size_t)g->in_vr.n * 8);
  }
  hipError_t e = hipMalloc(&g->arena, total);
  if (e != hipSuccess) {
    rg::set_error("rg_graph_create_device: allocating %zu B failed: %s", total, hipGetErrorString(e));
    delete g;
    return 1;
  }
  for (const Item& it : items) {
    *it.field = (char*)g->arena + it.off;
    if (it.src && it.bytes) (void)hipMemcpyAsync(*it.field, it.src, it.bytes, hipMemcpyDeviceToDevice, s);
  }
  (void)ent_off;
  if (n_packs > 0) {
    hipLaunchKernelGGL(fill_int2_kernel, dim3(blocks((int64_t)n_packs * RG_PACK)), dim3(256), 0, s, g->in_pk_packs.ent, (int64_t)n_packs * RG_PACK,
                       make_int2(-1, 0));
    hipLaunchKernelGGL(pack_emit_kernel, dim3(g->in_vr.n), dim3(64), 0, s, g->in_vr.rows, place, g->in_vr.n, pack_row0, g->in_pk, g->in_pk_packs.ent,
                       g->in_pk_packs.pack, g->in_pk_packs.rows);
    hipLaunchKernelGGL(pack_count_kernel, dim3(blocks(n_packs)), dim3(256), 0, s, pack_nrows, g->in_pk_packs.pack, n_packs);
  }
  e = hipStreamSynchronize(s);                 // the temporaries are freed on return
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    rg::set_error("rg_graph_create_device: build failed: %s", hipGetErrorString(e));
    rg_graph_destroy(g);
    return 1;
  }
  *out = g;
  return 0;
}
