// Frontier expansion on the device.
// Replaces DataLoader.get_neighbors (Static/transductive/load_data.py:106-131): the scipy
// one-hot SpMM + np.nonzero + two torch.unique(dim=0) sorts become bit-parallel set algebra:
//   * the visited set of all B queries is an entity-major bitmap [n_ent][B bits]; one hop is
//     new[t] = OR_{h in in(t)} old[h]  (B queries per 32-bit word; identity rows keep old nodes),
//   * transposed to batch-major words, a popcount prefix sum gives every (batch, entity) pair its
//     rank in the reference's sorted order (torch.unique(dim=0, sorted=True), load_data.py:122-123)
//     without sorting anything.
#include <string.h>

#include "common.h"

namespace {

// ---- level 0: one node (b, q_sub[b]) per query ---------------------------------------------------
// Level 0 of a query batch is one node per query, so its batch-major image needs no transpose / scan / pack passes:
// bm[0][b][w] = {bit of q_sub[b] if it falls in word w, b + (w beyond that word)}; the same kernel sets the entity-major bits
// (bitsT zeroed before) and the counters (N = B; the error flag for an id out of range).
__global__ void reset_level0_kernel(const int32_t* __restrict__ q_sub, int B, int n_ent, int BW, int W, uint32_t* __restrict__ bitsT,
                                    int2* __restrict__ bm0, int32_t* __restrict__ counters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * W) return;
  const int b = (int)(i / W), w = (int)(i - (int64_t)b * W);
  const int e = q_sub[b];
  const bool ok = e >= 0 && e < n_ent;
  const int ew = ok ? e >> 5 : W;                    // an invalid start node contributes no bit (and raises the flag)
  // nodes of earlier queries: b minus the invalid ones among them would be exact, but an invalid id is an error anyway
  bm0[i] = make_int2(w == ew ? (int)(1u << (e & 31)) : 0, b + (w > ew ? 1 : 0));
  if (w == 0) {
    if (ok) atomicOr(&bitsT[(int64_t)e * BW + (b >> 5)], 1u << (b & 31));
    else atomicOr((unsigned*)&counters[1], 1u);
    if (b == 0) { counters[0] = B; counters[4] = B; }
  }
}

// end of a hop: snapshot {N, error flag, E} of the new level for rg_frontier_level_counts (per-level slot) and for
// rg_frontier_expand's read-back (slot 8), then clear the E accumulator for the next hop's prologue and the work-queue heads
// for the layer kernel that follows (one launch less per walk)
__device__ __forceinline__ void end_of_hop(int32_t* __restrict__ counters, int32_t* __restrict__ queues, int slot, int t) {
  if (t < 4) {
    const int32_t v = counters[t];
    if (slot >= 0) counters[slot + t] = v;
    counters[8 + t] = v;
  }
  if (t >= 8 && t < 16) queues[(t - 8) * RG_QSTRIDE] = 0;
}
__global__ void snapshot_kernel(int32_t* __restrict__ counters, int32_t* __restrict__ queues, int slot) {
  end_of_hop(counters, queues, slot, threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0) { counters[2] = 0; counters[3] = 0; }
}

__global__ void reset_nodes_kernel(const int32_t* __restrict__ nodes, int64_t n, int B, int n_ent, int BW,
                                   uint32_t* __restrict__ bitsT, int32_t* counters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = nodes[2 * i], e = nodes[2 * i + 1];
  if (e < 0 || e >= n_ent || b < 0 || b >= B) { atomicOr((unsigned*)&counters[1], 1u); return; }
  atomicOr(&bitsT[(int64_t)e * BW + (b >> 5)], 1u << (b & 31));
}

// ---- one hop on the entity-major bitmap: new[t] = OR over in-edges (h -> t) of old[h] --------------
// One wave per virtual row of the CSR-by-tail (<= 128 in-edges; a hub of in-degree 17k is 133 waves, not
// one 17k-long chain of dependent loads).  WL lanes span the B/32 words of a bitmap row, 64/WL lanes
// stride over the in-edges, 4 independent loads in flight each; rows are merged with atomicOr
// (order-free), newT is zeroed by the caller.
__global__ __launch_bounds__(256) void hop_or_kernel(const int4* __restrict__ vrows, int n_vrows,
                                                     const int2* __restrict__ in_hr, const uint32_t* __restrict__ oldT,
                                                     uint32_t* __restrict__ newT, int BW, int WL) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n_vrows) return;
  const int4 row = vrows[wave];
  const int t = row.x, beg = row.y, end = row.y + row.z;
  const int wl = lane & (WL - 1);
  const int el = lane / WL, EL = 64 / WL;
  for (int w0 = 0; w0 < BW; w0 += WL) {
    const int w = w0 + wl;
    uint32_t acc = 0;
    if (w < BW) {
      for (int j = beg + el; j < end; j += 4 * EL) {
        int hh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) hh[u] = (j + u * EL < end) ? in_hr[j + u * EL].x : -1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (hh[u] >= 0) acc |= oldT[(int64_t)hh[u] * BW + w];
      }
    }
    for (int o = WL; o < 64; o <<= 1) acc |= __shfl_xor(acc, o, 64);
    if (el == 0 && w < BW && acc) atomicOr(&newT[(int64_t)t * BW + w], acc);
  }
}

// ---- the hop under per-query time windows (temporal extrapolation): edge j = data row in_time[j] counts for query b only if
// win_lo[b] <= row < win_hi[b] (rows >= n_data are the self-loops: always).  Same lane mapping as hop_or_kernel; the 32-query
// mask of an (edge, word) pair is built from the window bounds only when the source has a bit in that word.  E (valid edges of
// the hop) is counted here: the prologue's out-degree sum would count rows outside the windows.
__global__ __launch_bounds__(256) void hop_or_window_kernel(const int4* __restrict__ vrows, int n_vrows, const int2* __restrict__ in_hr,
                                                            const int32_t* __restrict__ in_row, const uint32_t* __restrict__ oldT,
                                                            uint32_t* __restrict__ newT, int BW, int WL, int B, const int32_t* __restrict__ win_lo,
                                                            const int32_t* __restrict__ win_hi, int n_data, unsigned long long* total) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n_vrows) return;
  const int4 row = vrows[wave];
  const int t = row.x, beg = row.y, end = row.y + row.z;
  const int wl = lane & (WL - 1);
  const int el = lane / WL, EL = 64 / WL;
  unsigned long long cnt = 0;
  for (int w0 = 0; w0 < BW; w0 += WL) {
    const int w = w0 + wl;
    uint32_t acc = 0;
    if (w < BW) {
      for (int j = beg + el; j < end; j += EL) {
        uint32_t v = oldT[(int64_t)in_hr[j].x * BW + w];
        if (v) {
          const int r = in_row[j];
          if (r < n_data) {
            uint32_t m = 0;
            for (uint32_t rest = v; rest; rest &= rest - 1) {
              const int bit = __ffs((int)rest) - 1, b = w * 32 + bit;
              if (b < B && r >= win_lo[b] && r < win_hi[b]) m |= 1u << bit;
            }
            v = m;
          }
          acc |= v;
          cnt += __popc(v);
        }
      }
    }
    for (int o = WL; o < 64; o <<= 1) acc |= __shfl_xor(acc, o, 64);
    if (el == 0 && w < BW && acc) atomicOr(&newT[(int64_t)t * BW + w], acc);
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if (lane == 0 && cnt) atomicAdd(total, cnt);
}

// ---- hop prologue: E += sum over visited (b,h) of outdeg(h) (the accumulator is cleared by the previous hop's end / the
// reset), and the new level's entity-major bitmap is zeroed for hop_or_kernel's atomicOr ------------------------------------
__global__ __launch_bounds__(256) void hop_prologue_kernel(const int32_t* __restrict__ out_ptr,
                                                           const uint32_t* __restrict__ oldT, uint32_t* __restrict__ newT,
                                                           int n_ent, int BW, unsigned long long* total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = 0;
  if (i < (int64_t)n_ent * BW) {
    const int h = (int)(i / BW);
    if (out_ptr) v = (unsigned long long)(out_ptr[h + 1] - out_ptr[h]) * __popc(oldT[i]);
    newT[i] = 0u;
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __shared__ unsigned long long ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = ws[0] + ws[1] + ws[2] + ws[3];
    if (s) atomicAdd(total, s);
  }
}

// ---- 32x32 bit transposes: entity-major [n_ent][BW] -> batch-major words [B][W] --------------------
// One wave per (batch word bw, pair of entity words): lane l holds row e = 64*pair + l.
__global__ __launch_bounds__(256) void transpose_kernel(const uint32_t* __restrict__ bitsT, uint32_t* __restrict__ words,
                                                        int n_ent, int B, int BW, int W, int n_pairs) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= BW * n_pairs) return;
  const int bw = wave / n_pairs, pair = wave - bw * n_pairs;
  const int e = pair * 64 + lane;
  const uint32_t v = (e < n_ent) ? bitsT[(int64_t)e * BW + bw] : 0u;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int b = 0; b < 32; ++b) {
    const unsigned long long m = __ballot((v >> b) & 1u);
    if (lane == b) { lo = (uint32_t)m; hi = (uint32_t)(m >> 32); }
  }
  const int batch = bw * 32 + lane;
  if (lane < 32 && batch < B) {
    const int ew = pair * 2;
    words[(int64_t)batch * W + ew] = lo;
    if (ew + 1 < W) words[(int64_t)batch * W + ew + 1] = hi;
  }
}

__global__ void pack_kernel(const uint32_t* __restrict__ words, const int32_t* __restrict__ prefix,
                            int2* __restrict__ bm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) bm[i] = make_int2((int)words[i], prefix[i]);
}

// ---- node list of the newest level + links to the previous level -----------------------------------
// one (word, bit) pair per thread: the set bits of a word are consecutive node ids, so a wave's stores are contiguous
__global__ void emit_nodes_kernel(const int2* __restrict__ bm_new, const int2* __restrict__ bm_old, int B, int W,
                                  int32_t* __restrict__ nodes, int32_t* __restrict__ prev_idx,
                                  int32_t* __restrict__ old_new) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t >> 5;
  const int pos = (int)(t & 31);
  if (i >= (int64_t)B * W) return;
  const int2 nw = bm_new[i];
  const uint32_t bits = (uint32_t)nw.x;
  if (!((bits >> pos) & 1u)) return;
  const int64_t idx = nw.y + __popc(bits & ((1u << pos) - 1u));
  const int b = (int)(i / W), w = (int)(i - (int64_t)b * W);
  if (nodes) reinterpret_cast<int2*>(nodes)[idx] = make_int2(b, w * 32 + pos);
  if (prev_idx || old_new) {
    int prev = -1;
    if (bm_old) {
      const int2 ow = bm_old[i];
      if (((uint32_t)ow.x >> pos) & 1u) prev = ow.y + __popc((uint32_t)ow.x & ((1u << pos) - 1u));
    }
    if (prev_idx) prev_idx[idx] = prev;
    if (old_new && prev >= 0) old_new[prev] = (int32_t)idx;
  }
}

// ---- materialised edges (compatibility path) -------------------------------------------------------
__device__ __forceinline__ int rank_in(const int2* __restrict__ bm, int b, int W, int e, bool* present) {
  const int2 wp = bm[(int64_t)b * W + (e >> 5)];
  const uint32_t word = (uint32_t)wp.x;
  *present = (word >> (e & 31)) & 1u;
  return wp.y + __popc(word & ((1u << (e & 31)) - 1u));
}

__global__ void edge_count_kernel(const int32_t* __restrict__ nodes_new, int64_t n_new, const int32_t* __restrict__ in_ptr,
                                  const int2* __restrict__ in_hr, const int2* __restrict__ bm_old, int W,
                                  uint32_t* __restrict__ deg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int b = nodes_new[2 * i], t = nodes_new[2 * i + 1];
  int c = 0;
  for (int j = in_ptr[t]; j < in_ptr[t + 1]; ++j) {
    bool p;
    rank_in(bm_old, b, W, in_hr[j].x, &p);
    c += p;
  }
  deg[i] = c;
}

__global__ void edge_fill_kernel(const int32_t* __restrict__ nodes_new, int64_t n_new, const int32_t* __restrict__ in_ptr,
                                 const int2* __restrict__ in_hr, const int2* __restrict__ bm_old, int W,
                                 const int32_t* __restrict__ row_ptr, int32_t* __restrict__ edges) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_new) return;
  const int b = nodes_new[2 * i], t = nodes_new[2 * i + 1];
  int64_t o = row_ptr[i];
  for (int j = in_ptr[t]; j < in_ptr[t + 1]; ++j) {
    const int2 hr = in_hr[j];
    bool p;
    const int s = rank_in(bm_old, b, W, hr.x, &p);
    if (p) {
      int32_t* e = edges + 6 * o;
      e[0] = b; e[1] = hr.x; e[2] = hr.y; e[3] = t; e[4] = s; e[5] = (int32_t)i;
      ++o;
    }
  }
}

// ---- small batches: transpose + popcount scan + pack + end of hop (+ node list) in ONE workgroup --------------------------
// At the reference's evaluation batch sizes (n_tbatch = 50: 50 x 94 words on family) the five kernels above are a few
// microseconds of work each and ~7 us of launch latency each inside a replayed HIP graph; the batch-major words fit LDS.
constexpr int SMALL_T = 1024;
constexpr int SMALL_MAX_WORDS = 12288;      // [B][W] words in LDS (48 KB)

__global__ __launch_bounds__(SMALL_T) void build_level_small_kernel(const uint32_t* __restrict__ bitsT, int n_ent, int B, int BW, int W,
                                                                    int n_pairs, int2* __restrict__ bm_new, const int2* __restrict__ bm_old,
                                                                    int32_t* __restrict__ counters, int32_t* __restrict__ queues, int slot,
                                                                    int32_t* __restrict__ nodes, int32_t* __restrict__ prev_idx,
                                                                    int32_t* __restrict__ old_new) {
  extern __shared__ uint32_t words[];       // [B * W]
  __shared__ int32_t wsum[SMALL_T / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nw = B * W;
  for (int task = wv; task < BW * n_pairs; task += SMALL_T / 64) {       // as transpose_kernel, one (batch word, entity pair) per wave
    const int bw = task / n_pairs, pair = task - bw * n_pairs;
    const int e = pair * 64 + lane;
    const uint32_t v = (e < n_ent) ? bitsT[(int64_t)e * BW + bw] : 0u;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < 32; ++b) {
      const unsigned long long m = __ballot((v >> b) & 1u);
      if (lane == b) { lo = (uint32_t)m; hi = (uint32_t)(m >> 32); }
    }
    const int batch = bw * 32 + lane;
    if (lane < 32 && batch < B) {
      words[batch * W + 2 * pair] = lo;
      if (2 * pair + 1 < W) words[batch * W + 2 * pair + 1] = hi;
    }
  }
  __syncthreads();
  const int per = (nw + SMALL_T - 1) / SMALL_T;
  const int beg = min((int)threadIdx.x * per, nw), end = min(beg + per, nw);
  int32_t sum = 0;
  for (int i = beg; i < end; ++i) sum += __popc(words[i]);
  int32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int32_t pre = inc - sum, total = 0;
  for (int w = 0; w < SMALL_T / 64; ++w) {
    const int32_t x = wsum[w];
    if (w < wv) pre += x;
    total += x;
  }
  for (int i = beg; i < end; ++i) {
    uint32_t w = words[i];
    bm_new[i] = make_int2((int)w, pre);
    if (nodes || prev_idx || old_new) {
      const int b = i / W, e0 = (i - b * W) * 32;
      uint32_t ow = 0; int op = 0;
      if ((prev_idx || old_new) && bm_old) { const int2 o = bm_old[i]; ow = (uint32_t)o.x; op = o.y; }
      int idx = pre;
      for (uint32_t rest = w; rest; rest &= rest - 1, ++idx) {
        const int pos = __ffs((int)rest) - 1;
        if (nodes) reinterpret_cast<int2*>(nodes)[idx] = make_int2(b, e0 + pos);
        const int prev = ((ow >> pos) & 1u) ? op + __popc(ow & ((1u << pos) - 1u)) : -1;
        if (prev_idx) prev_idx[idx] = prev;
        if (old_new && prev >= 0) old_new[prev] = idx;
      }
    }
    pre += __popc(w);
  }
  // end of hop (cf. end_of_hop): N is this kernel's total, the error flag and E were left by earlier kernels
  const int t = threadIdx.x;
  if (t < 4) {
    const int32_t v = t == 0 ? total : counters[t];
    if (t == 0) counters[0] = total;
    if (slot >= 0) counters[slot + t] = v;
    counters[8 + t] = v;
  }
  if (t >= 8 && t < 16) queues[(t - 8) * RG_QSTRIDE] = 0;
  __syncthreads();
  if (t == 0) { counters[2] = 0; counters[3] = 0; }
}

__global__ void set_last_kernel(int32_t* row_ptr, int64_t n, const int32_t* total) { row_ptr[n] = *total; }

size_t ws_layout(int32_t n_ent, int32_t B, int32_t n_levels, size_t* off /*[8]*/) {
  const size_t BW = (B + 31) / 32, W = (n_ent + 31) / 32;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = rg::align_up(o + bytes, 256); return r; };
  off[0] = take((size_t)n_ent * BW * 4);                     // bitsT[0]
  off[1] = take((size_t)n_ent * BW * 4);                     // bitsT[1]
  off[2] = take((size_t)B * W * 4);                          // words_tmp
  off[3] = take((size_t)B * W * 4);                          // prefix_tmp
  off[4] = take(rg::scan_scratch_elems((int64_t)B * W) * 4); // scan scratch
  off[5] = take(1024 + RG_QUEUE_BYTES);                      // counters + per-level snapshots (1 KB), then the work-queue heads
  off[6] = take((size_t)n_levels * B * W * 8);               // bm levels
  return o;
}

}  // namespace

extern "C" {

size_t rg_frontier_workspace_bytes(int32_t n_ent, int32_t batch, int32_t n_levels) {
  if (n_ent <= 0 || batch <= 0 || n_levels < 2 || n_levels > RG_MAX_LEVELS) return 0;
  size_t off[8];
  return ws_layout(n_ent, batch, n_levels, off);
}

int rg_frontier_create(int32_t n_ent, int32_t batch, int32_t n_levels, void* ws, size_t ws_bytes, rg_frontier** out) {
  RG_CHECK(out != nullptr, "rg_frontier_create: out is NULL");
  *out = nullptr;
  RG_CHECK(n_ent > 0 && batch > 0, "rg_frontier_create: n_ent=%d batch=%d must be positive", n_ent, batch);
  RG_CHECK(n_levels >= 2 && n_levels <= RG_MAX_LEVELS, "rg_frontier_create: n_levels=%d not in [2,%d]", n_levels, RG_MAX_LEVELS);
  RG_CHECK((int64_t)n_ent * batch < ((int64_t)1 << 31), "rg_frontier_create: batch*n_ent = %lld does not fit int32",
           (long long)n_ent * batch);
  size_t off[8];
  const size_t need = ws_layout(n_ent, batch, n_levels, off);
  RG_CHECK(ws != nullptr && ws_bytes >= need, "rg_frontier_create: workspace %zu B < required %zu B", ws_bytes, need);
  RG_CHECK(((uintptr_t)ws & 255) == 0, "rg_frontier_create: workspace must be 256-B aligned");
  rg_frontier* f = new rg_frontier();
  f->n_ent = n_ent; f->B = batch; f->BW = (batch + 31) / 32; f->W = (n_ent + 31) / 32; f->n_levels = n_levels;
  char* base = (char*)ws;
  f->bitsT[0] = (uint32_t*)(base + off[0]);
  f->bitsT[1] = (uint32_t*)(base + off[1]);
  f->words_tmp = (uint32_t*)(base + off[2]);
  f->prefix_tmp = (int32_t*)(base + off[3]);
  f->scan_scratch = (int32_t*)(base + off[4]);
  f->counters = (int32_t*)(base + off[5]);
  f->queues = f->counters + 256;
  for (int l = 0; l < n_levels; ++l) f->bm[l] = (int2*)(base + off[6] + (size_t)l * batch * f->W * 8);
  if (hipHostMalloc((void**)&f->counts_pinned, 1024, hipHostMallocDefault) != hipSuccess) {
    delete f;
    rg::set_error("rg_frontier_create: hipHostMalloc failed");
    return 1;
  }
  *out = f;
  return 0;
}

int rg_frontier_destroy(rg_frontier* f) {
  if (!f) return 0;
  if (f->counts_pinned) (void)hipHostFree(f->counts_pinned);
  delete f;
  return 0;
}

// bitsT[tcur] -> bm[level % n_levels] with prefix; leaves N in counters[0]
static int build_level(rg_frontier* f, hipStream_t s) {
  const int n_pairs = (f->W + 1) / 2;
  const int64_t waves = (int64_t)f->BW * n_pairs;
  hipLaunchKernelGGL(transpose_kernel, dim3(rg::ceil_div(waves, 4)), dim3(256), 0, s, f->bitsT[f->tcur], f->words_tmp,
                     f->n_ent, f->B, f->BW, f->W, n_pairs);
  RG_LAUNCH_CHECK();
  const int64_t nw = (int64_t)f->B * f->W;
  if (rg::scan_exclusive(f->words_tmp, f->prefix_tmp, nw, true, &f->counters[0], f->scan_scratch, s)) return 1;
  hipLaunchKernelGGL(pack_kernel, dim3(rg::ceil_div(nw, 256)), dim3(256), 0, s, f->words_tmp, f->prefix_tmp,
                     f->bm[f->level % f->n_levels], nw);
  RG_LAUNCH_CHECK();
  return 0;
}

int rg_frontier_reset(rg_frontier* f, const int32_t* q_sub, void* stream) {
  RG_CHECK(f && q_sub, "rg_frontier_reset: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  f->level = 0; f->tcur = 0; f->n_edges = 0;
  if (rg::zero2_async(f->bitsT[0], rg::align_up((size_t)f->n_ent * f->BW * 4, 16), f->counters, 1024, s)) return 1;
  f->queues_clean = false;
  const int64_t nw = (int64_t)f->B * f->W;
  hipLaunchKernelGGL(reset_level0_kernel, dim3(rg::ceil_div(nw, 256)), dim3(256), 0, s, q_sub, f->B, f->n_ent, f->BW, f->W,
                     f->bitsT[0], f->bm[0], f->counters);
  RG_LAUNCH_CHECK();
  f->n_nodes[0] = f->B;   // one node per query; an out-of-range q_sub is reported by the next expand
  return 0;
}

int rg_frontier_set_window(rg_frontier* f, const int32_t* win_lo, const int32_t* win_hi, int32_t n_data) {
  RG_CHECK(f != nullptr, "rg_frontier_set_window: frontier is NULL");
  RG_CHECK((win_lo == nullptr) == (win_hi == nullptr) && n_data >= 0, "rg_frontier_set_window: give both bounds (or neither)");
  f->win_lo = win_lo; f->win_hi = win_hi; f->win_n_data = n_data;
  return 0;
}

int rg_frontier_reset_nodes(rg_frontier* f, const int32_t* nodes, int64_t n, void* stream) {
  RG_CHECK(f && (nodes || n == 0) && n >= 0, "rg_frontier_reset_nodes: bad argument");
  hipStream_t s = (hipStream_t)stream;
  f->level = 0; f->tcur = 0; f->n_edges = 0;
  if (rg::zero2_async(f->bitsT[0], rg::align_up((size_t)f->n_ent * f->BW * 4, 16), f->counters, 1024, s)) return 1;
  f->queues_clean = false;
  if (n > 0) {
    hipLaunchKernelGGL(reset_nodes_kernel, dim3(rg::ceil_div(n, 256)), dim3(256), 0, s, nodes, n, f->B, f->n_ent, f->BW,
                       f->bitsT[0], f->counters);
    RG_LAUNCH_CHECK();
  }
  if (build_level(f, s)) return 1;
  if (rg::copy_words_async(&f->counters[4], &f->counters[0], 1, s)) return 1;
  f->n_nodes[0] = n;   // duplicates or out-of-range ids are reported by the next expand
  return 0;
}

// enqueue one expansion: bitsT[tcur] -> bitsT[tcur^1] -> bm[level+1]; N and E of the new level are left in counters[8], [10..11]
// and snapshotted into counters[64 + 8*level ..] for rg_frontier_level_counts; the work-queue heads are left zeroed.
// nodes / prev_idx (optional): the new level's node list as rg_frontier_nodes writes it, fused into the same launch for small batches.
static int enqueue_expand(rg_frontier* f, const rg_graph* g, hipStream_t s, int32_t* nodes = nullptr, int32_t* prev_idx = nullptr) {
  const uint32_t* oldT = f->bitsT[f->tcur];
  uint32_t* newT = f->bitsT[f->tcur ^ 1];
  int WL = 1;
  while (WL < f->BW && WL < 64) WL <<= 1;
  const bool windowed = f->win_lo != nullptr;
  RG_CHECK(!windowed || g->in_time, "rg_frontier_expand: time windows are set but the graph carries no data-row ids (rg_tgraph_create)");
  hipLaunchKernelGGL(hop_prologue_kernel, dim3(rg::ceil_div((int64_t)f->n_ent * f->BW, 256)), dim3(256), 0, s,
                     windowed ? (const int32_t*)nullptr : g->out_ptr, oldT, newT, f->n_ent, f->BW, (unsigned long long*)&f->counters[2]);
  RG_LAUNCH_CHECK();
  if (windowed)
    hipLaunchKernelGGL(hop_or_window_kernel, dim3(rg::ceil_div(g->in_vr.n, 4)), dim3(256), 0, s, g->in_vr.rows, g->in_vr.n, g->in_hr,
                       g->in_time, oldT, newT, f->BW, WL, f->B, f->win_lo, f->win_hi, f->win_n_data, (unsigned long long*)&f->counters[2]);
  else
    hipLaunchKernelGGL(hop_or_kernel, dim3(rg::ceil_div(g->in_vr.n, 4)), dim3(256), 0, s, g->in_vr.rows, g->in_vr.n, g->in_hr,
                       oldT, newT, f->BW, WL);
  RG_LAUNCH_CHECK();
  f->tcur ^= 1;
  f->level += 1;
  const int slot = f->level < RG_MAX_LEVELS ? 64 + 8 * f->level : -1;
  const int64_t nw = (int64_t)f->B * f->W;
  const int2* bm_old = f->bm_of(f->level - 1);
  if (nw <= SMALL_MAX_WORDS) {
    // one workgroup writes a node list at one CU's store rate (64 B/clk: the 150 k nodes of a saturated 50-query family level took 20 of
    // the kernel's 29 us, on the dependent chain of every replayed forward): past a few thousand words the list is a launch of its own
    const bool split_emit = (nodes || prev_idx) && nw > 1024;
    hipLaunchKernelGGL(build_level_small_kernel, dim3(1), dim3(SMALL_T), (size_t)nw * 4, s, f->bitsT[f->tcur], f->n_ent, f->B, f->BW,
                       f->W, (f->W + 1) / 2, f->bm[f->level % f->n_levels], bm_old, f->counters, f->queues, slot,
                       split_emit ? (int32_t*)nullptr : nodes, split_emit ? (int32_t*)nullptr : prev_idx, (int32_t*)nullptr);
    RG_LAUNCH_CHECK();
    if (split_emit) {
      hipLaunchKernelGGL(emit_nodes_kernel, dim3(rg::ceil_div(nw * 32, 256)), dim3(256), 0, s, f->bm_of(f->level), bm_old, f->B, f->W,
                         nodes, prev_idx, (int32_t*)nullptr);
      RG_LAUNCH_CHECK();
    }
  } else {
    if (build_level(f, s)) return 1;
    hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(64), 0, s, f->counters, f->queues, slot);
    RG_LAUNCH_CHECK();
    if (nodes || prev_idx) {
      hipLaunchKernelGGL(emit_nodes_kernel, dim3(rg::ceil_div(nw * 32, 256)), dim3(256), 0, s, f->bm_of(f->level), bm_old, f->B, f->W,
                         nodes, prev_idx, (int32_t*)nullptr);
      RG_LAUNCH_CHECK();
    }
  }
  f->queues_clean = true;
  return 0;
}

int rg_frontier_expand(rg_frontier* f, const rg_graph* g, int64_t* counts_host, void* stream) {
  RG_CHECK(f && g, "rg_frontier_expand: NULL argument");
  RG_CHECK(f->level >= 0, "rg_frontier_expand: call rg_frontier_reset first");
  RG_CHECK(g->n_ent == f->n_ent, "rg_frontier_expand: graph has %d entities, frontier %d", g->n_ent, f->n_ent);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_old = f->n_nodes[f->level % f->n_levels];
  if (enqueue_expand(f, g, s)) return 1;
  RG_HIP(hipMemcpyAsync(f->counts_pinned, f->counters, 64, hipMemcpyDeviceToHost, s));
  RG_HIP(hipStreamSynchronize(s));
  const int32_t* c32 = (const int32_t*)f->counts_pinned;
  RG_CHECK(c32[1] == 0, "rg_frontier_expand: a start node had batch or entity id out of range");
  RG_CHECK(f->level != 1 || n_old < 0 || c32[4] == n_old, "rg_frontier_expand: %lld start nodes given but %d distinct (duplicates?)",
           (long long)n_old, c32[4]);
  const int64_t n_new = c32[8];           // the hop's end left {N, flag, E} in slot 8 (and cleared the E accumulator)
  int64_t e;
  memcpy(&e, &c32[10], 8);
  f->n_nodes[f->level % f->n_levels] = n_new;
  f->n_edges = e;
  if (counts_host) { counts_host[0] = n_new; counts_host[1] = e; counts_host[2] = n_old; counts_host[3] = f->level; }
  return 0;
}

int rg_frontier_expand_async(rg_frontier* f, const rg_graph* g, void* stream) {
  return rg_frontier_expand_nodes_async(f, g, nullptr, nullptr, stream);
}

int rg_frontier_expand_nodes_async(rg_frontier* f, const rg_graph* g, int32_t* nodes_out, int32_t* prev_idx_out, void* stream) {
  RG_CHECK(f && g, "rg_frontier_expand_async: NULL argument");
  RG_CHECK(f->level >= 0, "rg_frontier_expand_async: call rg_frontier_reset first");
  RG_CHECK(g->n_ent == f->n_ent, "rg_frontier_expand_async: graph has %d entities, frontier %d", g->n_ent, f->n_ent);
  RG_CHECK(f->level + 1 < RG_MAX_LEVELS, "rg_frontier_expand_async: at most %d levels", RG_MAX_LEVELS - 1);
  RG_CHECK(!nodes_out || ((uintptr_t)nodes_out & 7) == 0, "rg_frontier_expand_nodes_async: nodes_out must be 8-byte aligned");
  if (enqueue_expand(f, g, (hipStream_t)stream, nodes_out, prev_idx_out)) return 1;
  f->n_nodes[f->level % f->n_levels] = -1;      // unknown on the host: the layer calls take their n as a capacity hint
  f->n_edges = -1;
  f->edge_hint = -1;
  return 0;
}

int rg_frontier_set_edge_hint(rg_frontier* f, int64_t n_edges) {
  RG_CHECK(f != nullptr, "rg_frontier_set_edge_hint: NULL frontier");
  f->edge_hint = n_edges;
  return 0;
}

const int32_t* rg_frontier_count_ptr(const rg_frontier* f) { return f ? f->counters : nullptr; }

int rg_frontier_level_counts(const rg_frontier* f, int64_t* counts_host, void* stream) {
  RG_CHECK(f && counts_host, "rg_frontier_level_counts: NULL argument");
  RG_CHECK(f->level >= 0 && f->level < RG_MAX_LEVELS, "rg_frontier_level_counts: level %d", f->level);
  hipStream_t s = (hipStream_t)stream;
  int32_t* host = (int32_t*)f->counts_pinned + 16;      // pinned: [0..7] is rg_frontier_expand's, snapshots behind it
  RG_HIP(hipMemcpyAsync(host, &f->counters[64], sizeof(int32_t) * 8 * (f->level + 1), hipMemcpyDeviceToHost, s));
  RG_HIP(hipStreamSynchronize(s));
  counts_host[0] = f->n_nodes[0]; counts_host[1] = 0;
  for (int l = 1; l <= f->level; ++l) {
    const int32_t* c = host + 8 * l;
    RG_CHECK(c[1] == 0, "rg_frontier_level_counts: a start node had batch or entity id out of range");
    counts_host[2 * l] = c[0];
    memcpy(&counts_host[2 * l + 1], &c[2], 8);
  }
  return 0;
}

int rg_frontier_nodes(const rg_frontier* f, int32_t* nodes, int32_t* prev_idx, int32_t* old_new, void* stream) {
  RG_CHECK(f && f->level >= 0, "rg_frontier_nodes: frontier not initialised");
  hipStream_t s = (hipStream_t)stream;
  const int2* bm_old = f->level > 0 ? f->bm_of(f->level - 1) : nullptr;
  RG_CHECK(!(old_new && !bm_old), "rg_frontier_nodes: level 0 has no previous level");
  const int64_t nw = (int64_t)f->B * f->W;
  RG_CHECK(!nodes || ((uintptr_t)nodes & 7) == 0, "rg_frontier_nodes: nodes_out must be 8-byte aligned");
  hipLaunchKernelGGL(emit_nodes_kernel, dim3(rg::ceil_div(nw * 32, 256)), dim3(256), 0, s, f->bm_of(f->level), bm_old, f->B, f->W,
                     nodes, prev_idx, old_new);
  RG_LAUNCH_CHECK();
  return 0;
}

size_t rg_frontier_edges_scratch_bytes(int64_t n_new) {
  return rg::align_up((size_t)(n_new + 1) * 4, 256) + rg::scan_scratch_elems(n_new) * 4 + 256;
}

int rg_frontier_edges(const rg_frontier* f, const rg_graph* g, int32_t level, const int32_t* nodes_new, int64_t n_new,
                      int32_t* edges, int32_t* row_ptr, void* scratch, void* stream) {
  RG_CHECK(f && g && nodes_new && row_ptr && scratch, "rg_frontier_edges: NULL argument");
  RG_CHECK(level >= 1 && level <= f->level && level > f->level - f->n_levels + 1,
           "rg_frontier_edges: level %d not resident (current %d, %d kept)", level, f->level, f->n_levels);
  RG_CHECK(n_new == f->n_nodes[level % f->n_levels], "rg_frontier_edges: n_new=%lld but level %d has %lld nodes",
           (long long)n_new, level, (long long)f->n_nodes[level % f->n_levels]);
  hipStream_t s = (hipStream_t)stream;
  uint32_t* deg = (uint32_t*)scratch;
  int32_t* scan_scr = (int32_t*)((char*)scratch + rg::align_up((size_t)(n_new + 1) * 4, 256));
  int32_t* total = scan_scr;          // first element; scan scratch proper starts after
  const int2* bm_old = f->bm_of(level - 1);
  if (n_new == 0) return rg::zero_async(row_ptr, 4, s);
  hipLaunchKernelGGL(edge_count_kernel, dim3(rg::ceil_div(n_new, 256)), dim3(256), 0, s, nodes_new, n_new, g->in_ptr,
                     g->in_hr, bm_old, f->W, deg);
  RG_LAUNCH_CHECK();
  if (rg::scan_exclusive(deg, row_ptr, n_new, false, total, scan_scr + 64, s)) return 1;
  hipLaunchKernelGGL(set_last_kernel, dim3(1), dim3(1), 0, s, row_ptr, n_new, total);
  RG_LAUNCH_CHECK();
  if (edges) {
    hipLaunchKernelGGL(edge_fill_kernel, dim3(rg::ceil_div(n_new, 256)), dim3(256), 0, s, nodes_new, n_new, g->in_ptr,
                       g->in_hr, bm_old, f->W, row_ptr, edges);
    RG_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
