// Weight gradients of the dense training step: out[m, n] = G^T X for node-row matrices G [N, m], X [N, n] with N in the millions and
// m, n <= 192 / 64 (autograd of Static/transductive/models.py:41 W_h and :83 the GRU's weight_ih / weight_hh; the hoisted Ws_attn),
// optionally with the column sums of G (the bias gradients).  A library GEMM lands such a product on a handful of workgroups
// (tools/probe_wgrad.py); round 1 issued it as 256 row chunks of a batched GEMM plus a sum (Tensile + aten reduce: ~6 ms of a 26 ms
// C2 / B = 256 training step).  Here every wave of a 1024-wave grid walks its own contiguous rows, four at a time as the K dimension
// of v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 sums: the arithmetic of the GEMM it replaces), with the whole m x n result
// in its accumulators (one wave per SIMD, up to 48 accumulator tiles: the f32 matrix pipe runs at its issue rate); rows are read
// exactly once.  Each wave then stores its partial result and a second launch adds the 1024 partials in wave order: the sums are
// bitwise reproducible (no float atomics).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GRAM_T = 256;            // 4 waves per workgroup, one per SIMD
constexpr int GRAM_BLOCKS = 256;
constexpr int GRAM_WAVES = GRAM_BLOCKS * GRAM_T / 64;

struct GramArgs {
  const float* g;
  const float* x;
  int64_t ldg, ldx, n_rows, rows_per_wave;
  int m, n;
  float* partial;      // [GRAM_WAVES][MB * 16][NB * 16 (+ 16 when colsum)]
  int colsum;
};

// MB, NB: 16-wide blocks of m and n (m <= 16 MB, n <= 16 NB; lanes beyond m / n feed zeros)
template <int MB, int NB>
__global__ __launch_bounds__(GRAM_T, 1) void gram_tn_kernel(GramArgs A) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (GRAM_T / 64) + (threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int64_t r_beg = (int64_t)wave * A.rows_per_wave;
  const int64_t r_end = r_beg + A.rows_per_wave < A.n_rows ? r_beg + A.rows_per_wave : A.n_rows;
  f32x4 acc[MB][NB];
  float cs[MB];
#pragma unroll
  for (int a = 0; a < MB; ++a) {
    cs[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bool m_ok[MB], n_ok[NB];
#pragma unroll
  for (int a = 0; a < MB; ++a) m_ok[a] = 16 * a + i < A.m;
#pragma unroll
  for (int b = 0; b < NB; ++b) n_ok[b] = 16 * b + i < A.n;
  // U K steps (4 U rows) per iteration: U (MB + NB) loads in flight ahead of U MB NB MFMAs (narrow products are latency bound: deeper)
  constexpr int U = MB >= 9 ? 2 : (MB >= 6 ? 3 : (MB >= 3 ? 4 : 8));
  for (int64_t r0 = r_beg; r0 < r_end; r0 += 4 * U) {
    float av[U][MB], bv[U][NB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + 4 * u + kq;
      const bool ok = r < r_end;
      const float* gr = A.g + r * A.ldg + i;
      const float* xr = A.x + r * A.ldx + i;
#pragma unroll
      for (int a = 0; a < MB; ++a) av[u][a] = (ok && m_ok[a]) ? gr[16 * a] : 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) bv[u][b] = (ok && n_ok[b]) ? xr[16 * b] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int a = 0; a < MB; ++a) {
        cs[a] += av[u][a];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][a], bv[u][b], acc[a][b], 0, 0, 0);
      }
    }
  }
  // the wave's partial: C/D layout col = lane & 15, row = 4 (lane >> 4) + reg
  const int pcols = NB * 16 + (A.colsum ? 16 : 0);
  float* P = A.partial + (int64_t)wave * (MB * 16) * pcols;
#pragma unroll
  for (int a = 0; a < MB; ++a) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(int64_t)(16 * a + 4 * kq + r) * pcols + 16 * b + i] = acc[a][b][r];
    }
    if (A.colsum) {
      // cs[a] of lane (i, kq) = the lane's share of column 16 a + i: the four row quarters add up in the combine (slots 0..3 of the row)
      P[(int64_t)(16 * a + i) * pcols + NB * 16 + kq] = cs[a];
    }
  }
}

// out[e] = the waves' partials added in a fixed order: four interleaved chains (waves w = c, c + 4, c + 8, ...) per element by four
// threads, chained (((c0 + c1) + c2) + c3) through LDS - always the same association, so the result is bitwise reproducible; a single
// chain per element (1024 dependent adds behind strided loads) took longer than the products themselves.
// colsum[row] = the same over the four row-quarter slots of every wave.
constexpr int COMB_T = 256;      // 64 elements x 4 chains

__global__ __launch_bounds__(COMB_T) void gram_combine_kernel(const float* __restrict__ partial, int n_waves, int mp, int pcols, int m, int n,
                                                              float* __restrict__ out, float* __restrict__ colsum) {
  __shared__ float red[COMB_T];
  const int el = threadIdx.x & 63, chain = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const int row = e / pcols, col = e - row * pcols;
  const int cs0 = pcols - (colsum ? 16 : 0);
  const bool is_cs = colsum && col == cs0;
  const bool live = row < m && (col < n || is_cs);
  float s = 0.f;
  if (live) {
    const int64_t stride = (int64_t)mp * pcols;
    const float* p = partial + (int64_t)row * pcols + col + chain * stride;
    if (!is_cs) {
#pragma unroll 8
      for (int w = chain; w < n_waves; w += 4, p += 4 * stride) s += p[0];
    } else {
      for (int w = chain; w < n_waves; w += 4, p += 4 * stride) s += (p[0] + p[1]) + (p[2] + p[3]);
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (chain == 0 && live) {
    const float t = ((red[el] + red[64 + el]) + red[128 + el]) + red[192 + el];
    if (is_cs) colsum[row] = t;
    else out[(int64_t)row * n + col] = t;
  }
}

template <int MB, int NB>
int launch(const GramArgs& A, hipStream_t s) {
  hipLaunchKernelGGL((gram_tn_kernel<MB, NB>), dim3(GRAM_BLOCKS), dim3(GRAM_T), 0, s, A);
  RG_LAUNCH_CHECK();
  return 0;
}

template <int MB>
int launch_n(const GramArgs& A, int nb, hipStream_t s) {
  switch (nb) {
    case 1: return launch<MB, 1>(A, s);
    case 2: return launch<MB, 2>(A, s);
    case 3: return launch<MB, 3>(A, s);
    default: return launch<MB, 4>(A, s);
  }
}

}  // namespace

static int mb_instance(int mb) { return mb <= 4 ? mb : (mb <= 6 ? 6 : (mb <= 9 ? 9 : 12)); }      // the MB the kernel is instantiated for

extern "C" size_t rg_gram_tn_scratch_bytes(int32_t m, int32_t n) {
  if (m < 1 || n < 1 || m > 192 || n > 64) return 0;
  const int mb = mb_instance((m + 15) / 16), nb = (n + 15) / 16;
  return (size_t)GRAM_WAVES * (mb * 16) * (nb * 16 + 16) * sizeof(float);
}

extern "C" int rg_gram_tn(const float* g, int64_t ldg, int32_t m, const float* x, int64_t ldx, int32_t n, int64_t n_rows, float* out,
                          float* colsum, void* scratch, size_t scratch_bytes, void* stream) {
  RG_CHECK(g && x && out && n_rows >= 0, "rg_gram_tn: NULL argument");
  RG_CHECK(m >= 1 && m <= 192 && n >= 1 && n <= 64, "rg_gram_tn: m=%d (<= 192) n=%d (<= 64): tile wider products over column blocks", m, n);
  RG_CHECK(ldg >= m && ldx >= n, "rg_gram_tn: ldg=%lld ldx=%lld", (long long)ldg, (long long)ldx);
  RG_CHECK(scratch && scratch_bytes >= rg_gram_tn_scratch_bytes(m, n), "rg_gram_tn: scratch %zu B < required %zu B", scratch_bytes,
           rg_gram_tn_scratch_bytes(m, n));
  const int mb = (m + 15) / 16, nb = (n + 15) / 16;
  GramArgs A;
  A.g = g; A.x = x; A.ldg = ldg; A.ldx = ldx; A.n_rows = n_rows; A.m = m; A.n = n;
  A.rows_per_wave = (rg::ceil_div(std::max<int64_t>(n_rows, 1), GRAM_WAVES) + 95) / 96 * 96;      // a multiple of every 4 U
  A.partial = (float*)scratch; A.colsum = colsum ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  switch (mb) {
    case 1: rc = launch_n<1>(A, nb, s); break;
    case 2: rc = launch_n<2>(A, nb, s); break;
    case 3: rc = launch_n<3>(A, nb, s); break;
    case 4: rc = launch_n<4>(A, nb, s); break;
    case 5: case 6: rc = launch_n<6>(A, nb, s); break;
    case 7: case 8: case 9: rc = launch_n<9>(A, nb, s); break;
    default: rc = launch_n<12>(A, nb, s); break;
  }
  if (rc) return rc;
  const int mb_used = mb_instance(mb);      // the partials have the geometry of the instantiation that ran
  const int pcols = nb * 16 + (colsum ? 16 : 0);
  const int elems = mb_used * 16 * pcols;
  hipLaunchKernelGGL(gram_combine_kernel, dim3(rg::ceil_div(elems, 64)), dim3(COMB_T), 0, s, (const float*)scratch, GRAM_WAVES, mb_used * 16,
                     pcols, m, n, out, colsum);
  RG_LAUNCH_CHECK();
  return 0;
}

// out[r, :n] = base[r, :n] + g[r, :k] W[:k, :n]  for node-row matrices (the gradient of the new state through the next layer's hoisted
// attention projection, added to the gradient it already carries: autograd of models.py:33 for a_s = hidden Ws_attn^T).  k <= 32,
// n <= 128 and a multiple of 4; W in LDS, one thread per four columns of a row: the pass is bound by its 2 n + k floats per row.
namespace {
__global__ __launch_bounds__(256) void rows_addmm_kernel(const float* __restrict__ base, int64_t ldb, const float* __restrict__ g, int64_t ldg,
                                                         int k, const float* __restrict__ W, int n, int64_t n_rows, float* __restrict__ out,
                                                         int64_t ldo) {
  __shared__ float4 w_l[32 * 32];
  const int n4 = n / 4;
  for (int i = threadIdx.x; i < k * n4; i += 256) w_l[i] = reinterpret_cast<const float4*>(W)[i];
  __syncthreads();
  const int64_t per_block = 256 / n4;                    // rows per workgroup pass
  const int c = threadIdx.x % n4, rl = threadIdx.x / n4;
  if (rl >= per_block) return;
  for (int64_t r = (int64_t)blockIdx.x * per_block + rl; r < n_rows; r += (int64_t)gridDim.x * per_block) {
    float4 acc = *reinterpret_cast<const float4*>(base + r * ldb + 4 * c);
    const float* gr = g + r * ldg;
    for (int j = 0; j < k; ++j) {
      const float gj = gr[j];
      const float4 w = w_l[j * n4 + c];
      acc.x = fmaf(gj, w.x, acc.x); acc.y = fmaf(gj, w.y, acc.y); acc.z = fmaf(gj, w.z, acc.z); acc.w = fmaf(gj, w.w, acc.w);
    }
    *reinterpret_cast<float4*>(out + r * ldo + 4 * c) = acc;
  }
}
}  // namespace

extern "C" int rg_rows_addmm(const float* base, int64_t ldb, const float* g, int64_t ldg, int32_t k, const float* W, int32_t n,
                             int64_t n_rows, float* out, int64_t ldo, void* stream) {
  RG_CHECK(base && g && W && out && n_rows >= 0, "rg_rows_addmm: NULL argument");
  RG_CHECK(k >= 1 && k <= 32 && n >= 4 && n <= 128 && n % 4 == 0, "rg_rows_addmm: k=%d (1..32) n=%d (4..128, multiple of 4)", k, n);
  RG_CHECK(ldb >= n && ldo >= n && ldg >= k && ldb % 4 == 0 && ldo % 4 == 0, "rg_rows_addmm: ldb=%lld ldo=%lld (>= n, multiples of 4) ldg=%lld",
           (long long)ldb, (long long)ldo, (long long)ldg);
  RG_CHECK((((uintptr_t)base | (uintptr_t)out | (uintptr_t)W) & 15) == 0, "rg_rows_addmm: base / out / W must be 16-B aligned");
  if (n_rows == 0) return 0;
  const int64_t per_block = 256 / (n / 4);
  const int grid = (int)std::max<int64_t>(std::min<int64_t>(rg::ceil_div(n_rows, per_block), 256 * 8), 1);
  hipLaunchKernelGGL(rows_addmm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, base, ldb, g, ldg, k, W, n, n_rows, out, ldo);
  RG_LAUNCH_CHECK();
  return 0;
}
