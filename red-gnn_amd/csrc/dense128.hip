// Fused dense epilogue of one layer at hidden_dim = 128 (inference): same arithmetic and register layout as dense.hip
// (transposed f32-MFMA products, an accumulator tile is the next product's B fragment), but the 448 KB of W_h /
// weight_ih / weight_hh cannot live in the CU's 160 KB of LDS.  They are STREAMED through it instead:
//
//   chunk = three 16-row blocks of a weight matrix (24 KB), double-buffered.  While the eight waves of the workgroup run
//   the 96 MFMAs of chunk c each against their own 16-node tile, every thread has the three float4 of chunk c+1 in flight
//   from L2 (coalesced 512-B rows) and drops them into the other buffer afterwards; one barrier per chunk.
//   Chunk order of a round: W_h blocks (0,1,2) (3,4,5) (6,7,-), then for every 16-row output block ob the r/z/n blocks
//   of weight_ih and - unless no node of the round has an old state - of weight_hh: 19 (11) chunks for 128 nodes.
//
// The L2->LDS stream is 456 KB per round of ~57 k MFMA cycles per wave: 8 B/clk per CU, nowhere near a limit; what it
// buys over reading A fragments straight from L2 (the round-1 kernel, 0.40 of the f32 MFMA peak) is that no MFMA ever
// waits on a global load, and two waves per SIMD at 256 VGPRs instead of one at 512.
#include "dense_common.h"

namespace rg {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DP = 128, S = 32, KS = 32, NB = 8, NW = 8, T = 512;
constexpr int BLK = 16 * S;       // float4 per 16-row weight block
constexpr int CHUNK = 3 * BLK;    // float4 per chunk (24 KB)
constexpr int NL = 16 * S / 64;   // float4 per lane of a 16-row tile

__device__ __forceinline__ int swz(int row, int slot) { return row * S + (slot ^ (row & (S - 1))); }

// TRAIN: as in dense.hip (dropout mask in; GRU input and gate workspace out, straight from the accumulator layout)
template <bool TRAIN>
__global__ __launch_bounds__(T, 2) void dense128_kernel(DenseArgs A) {
  extern __shared__ float4 lds[];
  if (A.n_dev) { A.n = *A.n_dev; A.n_tiles = (int)((A.n + 15) / 16); }      // the grid was sized for the capacity
  float4* wbuf = lds;                                            // [2][CHUNK]
  float4* E_l = wbuf + 2 * CHUNK;                                // [32][S]  rows 0..ap-1 = Ws, row 16 = W_final
  float* bias_l = reinterpret_cast<float*>(E_l + 32 * S);        // [4][DP]: b_ir+b_hr, b_iz+b_hz, b_in, b_hn
  float4* tiles = reinterpret_cast<float4*>(bias_l + 4 * DP);    // [NW][16][S]

  for (int i = threadIdx.x; i < 32 * S; i += T) {
    const int r = i / S, sl = i - r * S;
    const float* src = r < 16 ? (A.Ws && r < A.attn ? A.Ws + (int64_t)r * DP : nullptr) : (r == 16 ? A.W_final : nullptr);
    E_l[swz(r, sl)] = src ? *reinterpret_cast<const float4*>(src + 4 * sl) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = threadIdx.x; i < 4 * DP; i += T) {
    const int g = i / DP, c = i - g * DP;
    bias_l[i] = g == 0 ? A.b_ih[c] + A.b_hh[c] : g == 1 ? A.b_ih[DP + c] + A.b_hh[DP + c] : g == 2 ? A.b_ih[2 * DP + c] : A.b_hh[2 * DP + c];
  }
  __syncthreads();

  const int lane0 = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int lane = lane0, li = lane0 & 15, hq = lane0 >> 4;      // re-derived (opaquely) at the top of every round, see below
  float4* tile = tiles + wv * 16 * S;
  // this thread's float4 of every 16-row block of a chunk: row tid / 32, slot tid % 32
  const uint32_t ld_off = ((threadIdx.x >> 5) * DP + (threadIdx.x & 31) * 4) * sizeof(float);   // bytes: scalar base + 32-bit lane offset
  const int st_off = swz(threadIdx.x >> 5, threadIdx.x & 31);

  // chunk loads go through buffer descriptors (scalar base + scalar block offset + one 32-bit lane offset): with flat
  // addresses the compiler keeps a 64-bit VGPR pair per block of the round's 19 chunks and spills them
  const __amdgpu_buffer_rsrc_t r_wh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.W_h), 0, DP * DP * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_ih = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.w_ih), 0, 3 * DP * DP * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_hh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.w_hh), 0, 3 * DP * DP * 4, 0x00020000);
  u32x4 pre[3];
  // b0..b2: first rows of the chunk's three 16-row blocks
  auto issue = [&](const __amdgpu_buffer_rsrc_t& r, int b0, int b1, int b2) {
    pre[0] = __builtin_amdgcn_raw_buffer_load_b128(r, ld_off, b0 * DP * 4, 0);
    pre[1] = __builtin_amdgcn_raw_buffer_load_b128(r, ld_off, b1 * DP * 4, 0);
    pre[2] = __builtin_amdgcn_raw_buffer_load_b128(r, ld_off, b2 * DP * 4, 0);
  };
  auto as_f4 = [](const u32x4& v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); };
  int cur = 0;
  auto publish = [&]() {       // the prefetched chunk becomes the current one
    float4* dst = wbuf + (cur ^ 1) * CHUNK + st_off;
    dst[0] = as_f4(pre[0]); dst[BLK] = as_f4(pre[1]); dst[2 * BLK] = as_f4(pre[2]);
    __syncthreads();
    cur ^= 1;
  };
  // c_j += block j of the current chunk . frag, three independent MFMA chains interleaved
  auto mma3 = [&](const float (&f)[KS], f32x4& c0, f32x4& c1, f32x4& c2) {
    const float4* wb = wbuf + cur * CHUNK + li * S;
    float4 a0 = wb[hq ^ li], a1 = wb[BLK + (hq ^ li)], a2 = wb[2 * BLK + (hq ^ li)];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      // the next k-block's A fragments are read while this one's MFMAs run; the scheduling fence keeps the compiler from
      // hoisting all eight blocks' reads (96 registers) to the top
      const int sn = (4 * (kb + 1 < NB ? kb + 1 : kb) + hq) ^ li;
      const float4 n0 = wb[sn], n1 = wb[BLK + sn], n2 = wb[2 * BLK + sn];
      __builtin_amdgcn_sched_barrier(0);      // (the scheduler would otherwise sink these reads to the end of the block)
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, f[4 * kb + 0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, f[4 * kb + 0], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, f[4 * kb + 0], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, f[4 * kb + 1], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, f[4 * kb + 1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, f[4 * kb + 1], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, f[4 * kb + 2], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, f[4 * kb + 2], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, f[4 * kb + 2], c2, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, f[4 * kb + 3], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, f[4 * kb + 3], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, f[4 * kb + 3], c2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = n0; a1 = n1; a2 = n2;
    }
  };
  auto mma_e = [&](int row0, const float (&f)[KS]) -> f32x4 {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const float4 a = E_l[swz(row0 + li, 4 * kb + hq)];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f[4 * kb + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f[4 * kb + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f[4 * kb + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f[4 * kb + 3], acc, 0, 0, 0);
    }
    return acc;
  };
  auto bias_acc = [&](int g, int ob) -> f32x4 {
    const float4 v = *reinterpret_cast<const float4*>(bias_l + g * DP + 16 * ob + 4 * hq);
    f32x4 acc = {v.x, v.y, v.z, v.w};
    return acc;
  };
  auto read_frag = [&](float (&f)[KS]) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      const float4 v = tile[swz(li, 4 * ob + hq)];
      f[4 * ob + 0] = v.x; f[4 * ob + 1] = v.y; f[4 * ob + 2] = v.z; f[4 * ob + 3] = v.w;
    }
  };
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  };

  const int n_rounds = (A.n_tiles + NW - 1) / NW;
  for (int round = blockIdx.x; round < n_rounds; round += gridDim.x) {
    const int t = round * NW + wv;
    const int64_t row0 = (int64_t)t * 16;
    // lane coordinates made opaque once per round: the ~50 swizzled LDS offsets derived from them are two instructions each,
    // but as loop invariants the compiler computes them all before the round loop and spills them across it
    lane = lane0; li = lane0 & 15; hq = lane0 >> 4;
    asm volatile("" : "+v"(lane), "+v"(li), "+v"(hq));
    issue(r_wh, 0, 16, 32);          // chunk 0 flies under the tile loads

    // ---- this wave's tile: agg rows (coalesced) and the old state gathered by prev_idx ----------------------
    float fx[KS], hf[KS];
    int has_old = 0;
    {
      // agg rows and the old-state indices first, the gathered old rows second: both tiles in flight at once would need
      // 64 registers next to the fragments
      float4 v[NL];
      int prev[NL];
#pragma unroll
      for (int it = 0; it < NL; ++it) {
        const int e = it * 64 + lane, r = e / S, sl = e - r * S;
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        prev[it] = -1;
        if (row0 + r < A.n) {
          v[it] = A.agg[(row0 + r) * S + sl];
          if (A.prev_idx) prev[it] = A.prev_idx[row0 + r];
        }
      }
      wave_sync();
#pragma unroll
      for (int it = 0; it < NL; ++it) {
        const int e = it * 64 + lane, r = e / S, sl = e - r * S;
        tile[swz(r, sl)] = v[it];
      }
#pragma unroll
      for (int it = 0; it < NL; ++it) {
        const int sl = lane & (S - 1);
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prev[it] >= 0) { v[it] = A.hprev[(int64_t)prev[it] * S + sl]; has_old = 1; }
      }
      wave_sync();
      read_frag(fx);
      wave_sync();
#pragma unroll
      for (int it = 0; it < NL; ++it) {
        const int e = it * 64 + lane, r = e / S, sl = e - r * S;
        tile[swz(r, sl)] = v[it];
      }
      wave_sync();
      read_frag(hf);
    }
    {   // chunk 0 -> buffer 0 (the previous round ended on a barrier); the vote is the barrier that publishes it
      float4* dst = wbuf + st_off;
      dst[0] = as_f4(pre[0]); dst[BLK] = as_f4(pre[1]); dst[2 * BLK] = as_f4(pre[2]);
      cur = 0;
    }
    const bool hh = __syncthreads_or(has_old) != 0;   // a round of new nodes only (early hops): h = 0, weight_hh is skipped

    // ---- stage 1: x = act(W_h agg) ------------------------------------------------------------------------------
    float xf[KS];
    auto act_store = [&](const f32x4& acc, int ob) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[r];
        if (A.act == 1) v = fmaxf(v, 0.f);
        else if (A.act == 2) v = fast_tanh(v);
        xf[4 * ob + r] = v;
      }
      if constexpr (TRAIN) {
        const int64_t nd = row0 + li;
        const int col = 16 * ob + 4 * hq;
        if (nd < A.n) {
          if (A.mask) {
            const float4 mk = *reinterpret_cast<const float4*>(A.mask + nd * DP + col);
            xf[4 * ob + 0] *= mk.x; xf[4 * ob + 1] *= mk.y; xf[4 * ob + 2] *= mk.z; xf[4 * ob + 3] *= mk.w;
          }
          *reinterpret_cast<float4*>(A.x_out + nd * DP + col) = make_float4(xf[4 * ob + 0], xf[4 * ob + 1], xf[4 * ob + 2], xf[4 * ob + 3]);
        }
      }
    };
    {
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
      issue(r_wh, 48, 64, 80);
      mma3(fx, c0, c1, c2);
      act_store(c0, 0); act_store(c1, 1); act_store(c2, 2);
      publish();
    }
    {
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
      issue(r_wh, 96, 112, 112);
      mma3(fx, c0, c1, c2);
      act_store(c0, 3); act_store(c1, 4); act_store(c2, 5);
      publish();
    }
    {
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
      issue(r_ih, 0, DP, 2 * DP);
      mma3(fx, c0, c1, c2);          // the third block repeats block 7 (keeps one chunk shape); its result is dropped
      act_store(c0, 6); act_store(c1, 7);
      publish();
    }

    // ---- GRU gates, one 16-row output block at a time -------------------------------------------------------------
    float hn[KS];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 ar = bias_acc(0, ob), az = bias_acc(1, ob), ai = bias_acc(2, ob), ah = bias_acc(3, ob);
      const bool last = ob == NB - 1;
      if (hh) issue(r_hh, 16 * ob, DP + 16 * ob, 2 * DP + 16 * ob);
      else if (!last) issue(r_ih, 16 * (ob + 1), DP + 16 * (ob + 1), 2 * DP + 16 * (ob + 1));
      mma3(xf, ar, az, ai);
      if (hh || !last) publish(); else __syncthreads();
      if (hh) {
        if (!last) issue(r_ih, 16 * (ob + 1), DP + 16 * (ob + 1), 2 * DP + 16 * (ob + 1));
        mma3(hf, ar, az, ah);
        if (!last) publish(); else __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);      // keep one block's gate math from being spread over its neighbours' MFMAs:
#pragma unroll                                // the hoisted temporaries of eight blocks do not fit 256 registers
      for (int r = 0; r < 4; ++r) {
        const float rg = fast_sigmoid(ar[r]), zg = fast_sigmoid(az[r]);
        const float ng = fast_tanh(ai[r] + rg * ah[r]);
        hn[4 * ob + r] = (1.0f - zg) * ng + zg * hf[4 * ob + r];
        if constexpr (TRAIN) { ar[r] = rg; az[r] = zg; ai[r] = ng; }
      }
      if constexpr (TRAIN) {
        const int64_t nd = row0 + li;
        if (nd < A.n) {
          float* w = A.ws_out + nd * (5 * (int64_t)DP) + 16 * ob + 4 * hq;
          *reinterpret_cast<float4*>(w) = make_float4(ar[0], ar[1], ar[2], ar[3]);
          *reinterpret_cast<float4*>(w + DP) = make_float4(az[0], az[1], az[2], az[3]);
          *reinterpret_cast<float4*>(w + 2 * DP) = make_float4(ai[0], ai[1], ai[2], ai[3]);
          *reinterpret_cast<float4*>(w + 3 * DP) = make_float4(hf[4 * ob + 0], hf[4 * ob + 1], hf[4 * ob + 2], hf[4 * ob + 3]);
          *reinterpret_cast<float4*>(w + 4 * DP) = make_float4(ah[0], ah[1], ah[2], ah[3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- projections of the new state: a_s (next layer) or score (last layer) -------------------------------------
    const int64_t node = row0 + li;
    if (A.Ws) {
      const f32x4 ae = mma_e(0, hn);
      if (node < A.n && 4 * hq < A.ap)
        reinterpret_cast<float4*>(A.a_s_out + node * A.ap)[hq] = make_float4(ae[0], ae[1], ae[2], ae[3]);
    }
    if (A.W_final) {
      const f32x4 ae = mma_e(16, hn);
      if (node < A.n && hq == 0) {
        const int b = A.nodes[2 * node], e = A.nodes[2 * node + 1];
        A.scores[(int64_t)b * A.n_ent + e] = ae[0];
      }
    }

    // ---- new state: transpose through the tile, store coalesced rows ----------------------------------------------
    wave_sync();
#pragma unroll
    for (int ob = 0; ob < NB; ++ob)
      tile[swz(li, 4 * ob + hq)] = make_float4(hn[4 * ob + 0], hn[4 * ob + 1], hn[4 * ob + 2], hn[4 * ob + 3]);
    wave_sync();
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int e = it * 64 + lane, r = e / S, sl = e - r * S;
      if (row0 + r < A.n) A.hidden_out[(row0 + r) * S + sl] = tile[swz(r, sl)];
    }
  }
}

}  // namespace

int dense128_launch(const DenseArgs& A, hipStream_t s) {
  RG_CHECK(A.d == DP && A.ld4 == S, "rg_dense_fwd: the d = 128 kernel needs ld = 128 (got d=%d ld=%d)", A.d, A.ld4 * 4);
  const size_t lds = (size_t)(2 * CHUNK + 32 * S + NW * 16 * S) * sizeof(float4) + 4 * DP * sizeof(float);
  auto kern = A.ws_out ? dense128_kernel<true> : dense128_kernel<false>;
  RG_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t tiles = A.n_dev && A.n_hint > 0 ? std::min<int64_t>(A.n_tiles, ceil_div(A.n_hint + A.n_hint / 4, 16)) : A.n_tiles;
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(ceil_div(tiles, NW), 256), 1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

}  // namespace rg
