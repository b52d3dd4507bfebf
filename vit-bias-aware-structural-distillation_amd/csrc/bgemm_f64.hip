// Batched GEMM with fp64 accumulation on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64): C[b] = op(A[b]) op(B[b]), A/B fp32 or fp64,
// C fp32 or fp64, all row-major with explicit leading dimensions and batch
// strides.  Used by the Procrustes core for the products that must not be
// rounded to fp32 (cross-covariance, its Gram matrix, L^-1 cross): rocBLAS
// runs fp64 strided-batched GEMMs of these sizes as one launch per matrix
// (12k launches per step in the round-1 profile).
//
// Workgroup = 256 threads = 4 waves, 64x64 output tile (each wave 32x32 =
// 2x2 MFMA tiles), K chunks of 16 staged through LDS in fp64, k-major so the
// A/B fragments (one f64 per lane: A[i = lane&15][k = lane>>4]) are read from
// consecutive addresses.
#include <stdlib.h>
#include "basd_common.h"

namespace basd {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int BT = 64;      // tile
constexpr int BK = 16;      // K chunk
constexpr int BLD = BT + 2; // LDS row stride (doubles)

template <typename T>
__device__ __forceinline__ double ld_f64(const T* p) { return (double)(*p); }

// XCD-aware 1-D grid: workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  With the natural
// (n tile, m tile, matrix) grid the 12 tiles of one 196 x 192 product sat on different XCDs and every one fetched its
// operand rows again: 2.2 - 2.7 GB counted per launch against 0.8 GB algorithmic, and the launches ran at the HBM
// roof (6 TB/s), not the fp64 MFMA's.  Here block = ((matrix / 8) * tiles + tile) * 8 + matrix % 8: all tiles of a
// matrix share one XCD (one L2) and are adjacent in dispatch order.
__device__ __forceinline__ bool tile_of_block(int M, int N, int batch, int& m0, int& n0, int& mat) {
  const int tn = (N + BT - 1) / BT, tiles = tn * ((M + BT - 1) / BT);
  const int lin = blockIdx.x;
  mat = (lin / (8 * tiles)) * 8 + (lin & 7);
  if (mat >= batch) return false;
  const int tile = (lin >> 3) % tiles;
  m0 = (tile / tn) * BT;
  n0 = (tile % tn) * BT;
  return true;
}

// stage a [BK x BT] (k-major) tile of op(X) into LDS.
//   trans == 0: X is [rows = tile dim (M or N), cols = K] row-major -> element (k, i) = X[(t0 + i) * ld + k0 + k]
//   trans == 1: X is [K, tile dim] row-major                       -> element (k, i) = X[(k0 + k) * ld + t0 + i]
template <typename T>
__device__ __forceinline__ void stage(const T* __restrict__ x, int ld, int trans, int t0, int tdim, int k0, int kdim,
                                      double* __restrict__ dst, int tid) {
  if (trans) {
#pragma unroll
    for (int e = tid; e < BK * BT; e += 256) {
      const int k = e / BT, i = e - k * BT;
      const int gk = k0 + k, gi = t0 + i;
      dst[k * BLD + i] = (gk < kdim && gi < tdim) ? ld_f64(x + (size_t)gk * ld + gi) : 0.0;
    }
  } else {
#pragma unroll
    for (int e = tid; e < BK * BT; e += 256) {
      const int i = e / BK, k = e - i * BK;
      const int gk = k0 + k, gi = t0 + i;
      dst[k * BLD + i] = (gk < kdim && gi < tdim) ? ld_f64(x + (size_t)gi * ld + gk) : 0.0;
    }
  }
}

template <typename TA, typename TB, typename TC>
__global__ __launch_bounds__(256) void bgemm_f64_kernel(const TA* __restrict__ a, int64_t sa, int lda, int ta,
                                                        const TB* __restrict__ b, int64_t sb, int ldb, int tb,
                                                        TC* __restrict__ c, int64_t sc, int ldc, int M, int N,
                                                        int K, int sym, int batch, const int32_t* __restrict__ skip) {
  __shared__ double As[BK * BLD];
  __shared__ double Bs[BK * BLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int m0, n0, mat;
  if (!tile_of_block(M, N, batch, m0, n0, mat)) return;
  if (skip != nullptr && skip[mat] != 0) return;   // masked problem: its output is left untouched
  if (sym && n0 > m0) return;   // C = X X^T: lower tiles only, mirrored on store
  const TA* A = a + (size_t)mat * sa;
  const TB* B = b + (size_t)mat * sb;
  TC* C = c + (size_t)mat * sc;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    stage<TA>(A, lda, ta, m0, M, k0, K, As, tid);
    // op(B) is K x N: tb == 0 -> B stored [K, N] (k-major rows) == "trans" layout of stage()
    stage<TB>(B, ldb, tb ? 0 : 1, n0, N, k0, K, Bs, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      double av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[kr * BLD + wm + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Bs[kr * BLD + wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = m0 + wm + i * 16 + (lane >> 4) + 4 * reg;
        const int col = n0 + wn + j * 16 + (lane & 15);
        if (r < M && col < N) {
          C[(size_t)r * ldc + col] = (TC)acc[i][j][reg];
          if (sym && n0 != m0) C[(size_t)col * ldc + r] = (TC)acc[i][j][reg];
        }
      }
}


// ---------------------------------------------------------------------------------------------
// Pipelined variant for aligned shapes (M, N, K, leading dimensions and batch strides multiples
// of 4; 32-byte aligned bases; ragged 64-tiles at the M / N edges read zeros) -- every product of
// the Procrustes core, including the 196-token Gram matrices.
// The kernel above stages element-wise with bounds checks (98 VGPRs + 32 AGPRs: 3-4 waves per
// SIMD) and waits for its global loads between two barriers per K chunk: 46-58 % of the measured
// 47 TFLOP/s fp64-MFMA ceiling.  Here each thread fetches one 4-element vector of A and of B for
// chunk c+1 BEFORE the 16 MFMAs of chunk c, parks it in registers, and stores it to the other LDS
// buffer afterwards: one barrier per chunk, the loads fly under the matrix work.
// The transposes are template parameters (no runtime branches in the staging code).
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4))); };

// KMAJ == 1: the operand tile is contiguous along the tile dimension  -> element (k, i) = x[(k0 + k) * ld + t0 + i]
// KMAJ == 0: the operand tile is contiguous along k                  -> element (k, i) = x[(t0 + i) * ld + k0 + k]
template <typename T, int KMAJ>
__device__ __forceinline__ typename Vec4<T>::type fetch4(const T* __restrict__ x, int ld, int t0, int tdim, int k0, int K,
                                                         int tid) {
  typedef typename Vec4<T>::type V;
  // tdim and K are multiples of 4: a 4-vector is entirely inside or entirely outside the matrix (ragged edge tiles
  // of e.g. the 196-token Gram matrices read zeros there)
  if (KMAJ) {
    const int k = tid >> 4, i4 = (tid & 15) * 4;
    if (k0 + k < K && t0 + i4 < tdim) return *reinterpret_cast<const V*>(x + (size_t)(k0 + k) * ld + t0 + i4);
  } else {
    const int i = tid >> 2, kq = (tid & 3) * 4;
    if (k0 + kq < K && t0 + i < tdim) return *reinterpret_cast<const V*>(x + (size_t)(t0 + i) * ld + k0 + kq);
  }
  return (V){0, 0, 0, 0};
}
template <typename T, int KMAJ>
__device__ __forceinline__ void park4(typename Vec4<T>::type v, double* __restrict__ dst, int tid) {
  if (KMAJ) {
    const int k = tid >> 4, i4 = (tid & 15) * 4;
    double2* d = reinterpret_cast<double2*>(dst + k * BLD + i4);
    d[0] = make_double2((double)v.x, (double)v.y);
    d[1] = make_double2((double)v.z, (double)v.w);
  } else {
    const int i = tid >> 2, kq = (tid & 3) * 4;
    dst[(kq + 0) * BLD + i] = (double)v.x;
    dst[(kq + 1) * BLD + i] = (double)v.y;
    dst[(kq + 2) * BLD + i] = (double)v.z;
    dst[(kq + 3) * BLD + i] = (double)v.w;
  }
}

template <typename TA, typename TB, typename TC, int TRA, int TRB>
__global__ __launch_bounds__(256) void bgemm_f64_fast_kernel(const TA* __restrict__ a, int64_t sa, int lda,
                                                             const TB* __restrict__ b, int64_t sb, int ldb,
                                                             TC* __restrict__ c, int64_t sc, int ldc, int M, int N,
                                                             int K, int sym, int batch, const int32_t* __restrict__ skip) {
  __shared__ __align__(16) double As[2][BK * BLD];
  __shared__ __align__(16) double Bs[2][BK * BLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int m0, n0, mat;
  if (!tile_of_block(M, N, batch, m0, n0, mat)) return;
  if (skip != nullptr && skip[mat] != 0) return;   // masked problem: its output is left untouched
  if (sym && n0 > m0) return;
  const TA* A = a + (size_t)mat * sa;
  const TB* B = b + (size_t)mat * sb;
  TC* C = c + (size_t)mat * sc;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // 16 x 16 sub-tiles that need no MFMA (round 4): those entirely outside the matrix -- M = N = 196 is 12.25 sub-tiles:
  // the fourth 64-row tile holds 4 valid rows, and with the padded tiles executed in full the symmetric Gram of the
  // Procrustes chain did 160 sub-tile products per matrix for 91 useful ones -- and, in the diagonal tiles of a
  // symmetric product, those strictly above the diagonal (mirrored from their transposes at the store).  Wave-uniform.
  bool need[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r0 = m0 + wm + i * 16, c0 = n0 + wn + j * 16;
      need[i][j] = r0 < M && c0 < N && !(sym && m0 == n0 && c0 > r0);
    }
  // op(A) tile: TRA == 0 -> A is [M, K] (contiguous along k); TRA == 1 -> A is [K, M]
  // op(B) tile: TRB == 0 -> B is [K, N] (contiguous along the tile dim); TRB == 1 -> B is [N, K]
  typename Vec4<TA>::type ra = fetch4<TA, TRA>(A, lda, m0, M, 0, K, tid);
  typename Vec4<TB>::type rb = fetch4<TB, 1 - TRB>(B, ldb, n0, N, 0, K, tid);
  park4<TA, TRA>(ra, As[0], tid);
  park4<TB, 1 - TRB>(rb, Bs[0], tid);
  __syncthreads();
  const int nchunk = (K + BK - 1) / BK;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cur = ch & 1;
    const bool more = ch + 1 < nchunk;
    if (more) {
      ra = fetch4<TA, TRA>(A, lda, m0, M, (ch + 1) * BK, K, tid);
      rb = fetch4<TB, 1 - TRB>(B, ldb, n0, N, (ch + 1) * BK, K, tid);
    }
    const double* as = As[cur];
    const double* bs = Bs[cur];
    // a ragged last chunk (K = 196: four valid k of sixteen) only runs the k steps that hold data
    const int kk_end = (K - ch * BK + 3) >> 2;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      if (kk >= kk_end) break;
      const int kr = kk * 4 + (lane >> 4);
      double av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = as[kr * BLD + wm + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = bs[kr * BLD + wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (need[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      park4<TA, TRA>(ra, As[cur ^ 1], tid);
      park4<TB, 1 - TRB>(rb, Bs[cur ^ 1], tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = m0 + wm + i * 16 + (lane >> 4) + 4 * reg;
        const int col = n0 + wn + j * 16 + (lane & 15);
        if (need[i][j] && r < M && col < N) {
          C[(size_t)r * ldc + col] = (TC)acc[i][j][reg];
          // mirror: off-diagonal tiles, and the strictly lower sub-tiles of a diagonal tile
          if (sym && (n0 != m0 || n0 + wn + j * 16 < m0 + wm + i * 16)) C[(size_t)col * ldc + r] = (TC)acc[i][j][reg];
        }
      }
}

template <typename TA, typename TB, typename TC>
static void launch_bgemm_fast(const void* a, int64_t sa, int lda, int ta, const void* b, int64_t sb, int ldb, int tb,
                              void* c, int64_t sc, int ldc, int batch, int M, int N, int K, int sym, const int32_t* skip,
                              hipStream_t st) {
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * ((N + BT - 1) / BT) * ((M + BT - 1) / BT)));
#define BASD_BGF(TRA, TRB)                                                                                   \
  hipLaunchKernelGGL((bgemm_f64_fast_kernel<TA, TB, TC, TRA, TRB>), grid, dim3(256), 0, st, (const TA*)a, sa, lda, \
                     (const TB*)b, sb, ldb, (TC*)c, sc, ldc, M, N, K, sym, batch, skip)
  if (ta) { if (tb) BASD_BGF(1, 1); else BASD_BGF(1, 0); }
  else    { if (tb) BASD_BGF(0, 1); else BASD_BGF(0, 0); }
#undef BASD_BGF
}

// ---------------------------------------------------------------------------------------------
// Symmetric Gram C = X X^T of row-major X [n, K] (round 4): the Gt = t_w t_w^T of the feature-side Procrustes chain
// (n = 196 tokens, K = 768) was the most expensive fp64 launch of a step (1.5 ms per 1024 matrices, 26 TF/s of useful
// flops against the 45 the kernel above reaches on full tiles), because of how 196 = 3 x 64 + 4 falls on 64 x 64
// tiles: a diagonal tile needs 10 of its 16 sub-tiles but runs as long as a full one (the wave that owns the lower
// left quarter does four MFMAs per k step whatever the others skip), and the fourth tile row -- four valid rows -- costs
// 1.75 tile times.  Here a matrix is
//   * nt (nt - 1) / 2 full off-diagonal tiles (as above),
//   * nt diagonal tiles with the 10 sub-tiles dealt 3 / 3 / 2 / 2 to the waves (0.75 tile times; ONE staged operand,
//     every fragment read serves as row and as column operand),
//   * one strip workgroup for the r = n - 64 nt <= 8 trailing rows on the fp64 VALU (the strip rows in LDS as fp64,
//     one column per lane: ~0.4 tile times instead of 1.75),
// i.e. 5.6 tile times instead of 7.75.
template <typename TA, int W>
__device__ __forceinline__ void gram_diag_tile(const TA* __restrict__ X, int ldx, int m0, int n, int K,
                                               double* __restrict__ C, int ldc, double* __restrict__ As0, int tid) {
  // reads RS[0 .. NR), products (RS[MA[i]], RS[MB[i]]) -- see the table in the header comment
  constexpr int NR = (W == 0 || W == 3) ? 2 : 3;
  constexpr int NM = (W < 2) ? 3 : 2;
  constexpr int RS[3] = {W == 0 ? 0 : (W == 1 ? 2 : 3), W == 0 ? 1 : (W == 3 ? 2 : 0), W == 0 ? 0 : (W == 3 ? 0 : 1)};
  constexpr int MA[3] = {0, W == 0 ? 1 : 0, W == 0 ? 1 : 0};
  constexpr int MB[3] = {W == 0 ? 0 : 1, W == 0 ? 0 : (W == 3 ? 0 : 2), W == 0 ? 1 : 0};
  const int lane = tid & 63;
  f64x4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  typename Vec4<TA>::type ra = fetch4<TA, 0>(X, ldx, m0, n, 0, K, tid);
  park4<TA, 0>(ra, As0, tid);
  __syncthreads();
  const int nchunk = (K + BK - 1) / BK;
  for (int ch = 0; ch < nchunk; ++ch) {
    const int cur = ch & 1;
    const bool more = ch + 1 < nchunk;
    if (more) ra = fetch4<TA, 0>(X, ldx, m0, n, (ch + 1) * BK, K, tid);
    const double* as = As0 + cur * (BK * BLD);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      double f[3];
#pragma unroll
      for (int i = 0; i < NR; ++i) f[i] = as[kr * BLD + RS[i] * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[MA[i]], f[MB[i]], acc[i], 0, 0, 0);
    }
    if (more) park4<TA, 0>(ra, As0 + (cur ^ 1) * (BK * BLD), tid);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    const int si = RS[MA[i]], sj = RS[MB[i]];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r = m0 + si * 16 + (lane >> 4) + 4 * reg;
      const int col = m0 + sj * 16 + (lane & 15);
      C[(size_t)r * ldc + col] = acc[i][reg];
      if (si != sj) C[(size_t)col * ldc + r] = acc[i][reg];
    }
  }
}

constexpr int GRAM_RMAX = 8;                         // strip rows the VALU path takes

template <typename TA>
__global__ __launch_bounds__(256) void gram_rows_f64_kernel(const TA* __restrict__ x, int64_t sx, int ldx,
                                                            double* __restrict__ c, int64_t sc, int ldc, int n, int K,
                                                            int batch) {
  extern __shared__ __align__(16) double gsm[];
  double* As0 = gsm;                                 // [2][BK * BLD]
  double* Bs0 = gsm + 2 * BK * BLD;                  // [2][BK * BLD] (off-diagonal tiles); the strip: [r][K] from gsm
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = n / BT, rs = n - nt * BT;
  const int n_off = nt * (nt - 1) / 2;
  const int items = n_off + nt + (rs ? 1 : 0);
  const int lin = blockIdx.x;
  const int mat = (lin / (8 * items)) * 8 + (lin & 7);     // all items of a matrix on one XCD (see tile_of_block)
  if (mat >= batch) return;
  const int item = (lin >> 3) % items;
  const TA* X = x + (size_t)mat * sx;
  double* C = c + (size_t)mat * sc;
  if (item < n_off) {
    // ---- full off-diagonal tile (ti > tj)
    int ti = 1, e = item;
    while (e >= ti) { e -= ti; ++ti; }
    const int m0 = ti * BT, n0 = e * BT;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    typename Vec4<TA>::type ra = fetch4<TA, 0>(X, ldx, m0, n, 0, K, tid);
    typename Vec4<TA>::type rb = fetch4<TA, 0>(X, ldx, n0, n, 0, K, tid);
    park4<TA, 0>(ra, As0, tid);
    park4<TA, 0>(rb, Bs0, tid);
    __syncthreads();
    const int nchunk = (K + BK - 1) / BK;
    for (int ch = 0; ch < nchunk; ++ch) {
      const int cur = ch & 1;
      const bool more = ch + 1 < nchunk;
      if (more) {
        ra = fetch4<TA, 0>(X, ldx, m0, n, (ch + 1) * BK, K, tid);
        rb = fetch4<TA, 0>(X, ldx, n0, n, (ch + 1) * BK, K, tid);
      }
      const double* as = As0 + cur * (BK * BLD);
      const double* bs = Bs0 + cur * (BK * BLD);
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        const int kr = kk * 4 + (lane >> 4);
        double av[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = as[kr * BLD + wm + i * 16 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < 2; ++j) bv[j] = bs[kr * BLD + wn + j * 16 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
      if (more) {
        park4<TA, 0>(ra, As0 + (cur ^ 1) * (BK * BLD), tid);
        park4<TA, 0>(rb, Bs0 + (cur ^ 1) * (BK * BLD), tid);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r = m0 + wm + i * 16 + (lane >> 4) + 4 * reg;
          const int col = n0 + wn + j * 16 + (lane & 15);
          C[(size_t)r * ldc + col] = acc[i][j][reg];
          C[(size_t)col * ldc + r] = acc[i][j][reg];
        }
    return;
  }
  if (item < n_off + nt) {
    // ---- diagonal tile, 10 sub-tiles dealt 3 / 3 / 2 / 2 (every wave runs the same number of barriers)
    const int m0 = (item - n_off) * BT;
    if (wave == 0) gram_diag_tile<TA, 0>(X, ldx, m0, n, K, C, ldc, As0, tid);
    else if (wave == 1) gram_diag_tile<TA, 1>(X, ldx, m0, n, K, C, ldc, As0, tid);
    else if (wave == 2) gram_diag_tile<TA, 2>(X, ldx, m0, n, K, C, ldc, As0, tid);
    else gram_diag_tile<TA, 3>(X, ldx, m0, n, K, C, ldc, As0, tid);
    return;
  }
  // ---- the strip: rows s0 .. n-1 against every row j (one j per lane, 64 per wave, 256 per pass)
  const int s0 = nt * BT;
  double* S = gsm;                                   // [rs][K]
  for (int e = tid * 4; e < rs * K; e += 1024) {     // K % 4 == 0
    const int q = e / K, k = e - q * K;
    const typename Vec4<TA>::type v = *reinterpret_cast<const typename Vec4<TA>::type*>(X + (size_t)(s0 + q) * ldx + k);
    double2* d = reinterpret_cast<double2*>(S + q * K + k);
    d[0] = make_double2((double)v.x, (double)v.y);
    d[1] = make_double2((double)v.z, (double)v.w);
  }
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + tid;
    const bool valid = j < n;
    const TA* xr = X + (size_t)(valid ? j : 0) * ldx;
    double acc[GRAM_RMAX];
#pragma unroll
    for (int q = 0; q < GRAM_RMAX; ++q) acc[q] = 0.0;
    typename Vec4<TA>::type nx = *reinterpret_cast<const typename Vec4<TA>::type*>(xr);
    for (int k = 0; k < K; k += 4) {
      const typename Vec4<TA>::type v = nx;
      if (k + 4 < K) nx = *reinterpret_cast<const typename Vec4<TA>::type*>(xr + k + 4);
      const double v0 = (double)v.x, v1 = (double)v.y, v2 = (double)v.z, v3 = (double)v.w;
#pragma unroll
      for (int q = 0; q < GRAM_RMAX; ++q)
        if (q < rs) {
          const double2 sa = *reinterpret_cast<const double2*>(S + q * K + k);
          const double2 sb = *reinterpret_cast<const double2*>(S + q * K + k + 2);
          acc[q] = fma(v0, sa.x, acc[q]);
          acc[q] = fma(v1, sa.y, acc[q]);
          acc[q] = fma(v2, sb.x, acc[q]);
          acc[q] = fma(v3, sb.y, acc[q]);
        }
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < GRAM_RMAX; ++q)
        if (q < rs) {
          C[(size_t)(s0 + q) * ldc + j] = acc[q];
          if (j < s0) C[(size_t)j * ldc + s0 + q] = acc[q];
        }
    }
  }
}

template <typename TA>
static void launch_gram_rows(const void* a, int64_t sa, int lda, void* c, int64_t sc, int ldc, int batch, int n, int K,
                             hipStream_t st) {
  const int nt = n / BT, rs = n - nt * BT;
  const int items = nt * (nt - 1) / 2 + nt + (rs ? 1 : 0);
  size_t lds = (size_t)4 * BK * BLD * sizeof(double);
  const size_t strip = (size_t)rs * K * sizeof(double);
  if (strip > lds) lds = strip;
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * items));
  hipLaunchKernelGGL((gram_rows_f64_kernel<TA>), grid, dim3(256), lds, st, (const TA*)a, sa, lda, (double*)c, sc, ldc, n, K,
                     batch);
}

template <typename TA, typename TB, typename TC>
static void launch_bgemm(const void* a, int64_t sa, int lda, int ta, const void* b, int64_t sb, int ldb, int tb,
                         void* c, int64_t sc, int ldc, int batch, int M, int N, int K, int sym, const int32_t* skip,
                         hipStream_t st) {
  dim3 grid((unsigned)(((batch + 7) / 8) * 8 * ((N + BT - 1) / BT) * ((M + BT - 1) / BT)));
  hipLaunchKernelGGL((bgemm_f64_kernel<TA, TB, TC>), grid, dim3(256), 0, st, (const TA*)a, sa, lda, ta,
                     (const TB*)b, sb, ldb, tb, (TC*)c, sc, ldc, M, N, K, sym, batch, skip);
}

}  // namespace basd

extern "C" int basd_bgemm_f64(const void* a, int a_dtype, int64_t a_stride, int lda, int trans_a, const void* b,
                              int b_dtype, int64_t b_stride, int ldb, int trans_b, void* c, int c_dtype,
                              int64_t c_stride, int ldc, int batch, int M, int N, int K, int symmetric, void* stream) {
  return basd_bgemm_f64_masked(a, a_dtype, a_stride, lda, trans_a, b, b_dtype, b_stride, ldb, trans_b, c, c_dtype, c_stride,
                               ldc, batch, M, N, K, symmetric, nullptr, stream);
}

extern "C" int basd_bgemm_f64_masked(const void* a, int a_dtype, int64_t a_stride, int lda, int trans_a, const void* b,
                                     int b_dtype, int64_t b_stride, int ldb, int trans_b, void* c, int c_dtype,
                                     int64_t c_stride, int ldc, int batch, int M, int N, int K, int symmetric,
                                     const int32_t* skip, void* stream) {
  using namespace basd;
  if (batch <= 0 || M <= 0 || N <= 0) return BASD_OK;
  if (symmetric && M != N) return fail(BASD_ERR_SHAPE, "bgemm_f64: symmetric needs M == N");
  if (K <= 0) return fail(BASD_ERR_SHAPE, "bgemm_f64: bad shape batch=%d K=%d", batch, K);
  {
    const long long blocks = (long long)((batch + 7) / 8) * 8 * ((N + 63) / 64) * ((M + 63) / 64);
    if (blocks > 0x7fffffffLL) return fail(BASD_ERR_SHAPE, "bgemm_f64: %lld workgroups exceed the grid limit", blocks);
  }
  hipStream_t st = (hipStream_t)stream;
  const int key = a_dtype * 100 + b_dtype * 10 + c_dtype;
  const bool aligned = M % 4 == 0 && N % 4 == 0 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
                       a_stride % 4 == 0 && b_stride % 4 == 0 && ((uintptr_t)a & 31) == 0 && ((uintptr_t)b & 31) == 0;
  // C = X X^T of one row-major operand into fp64: the balanced Gram kernel (BASD_GRAM_ROWS=0: the generic tiles)
  if (symmetric && a == b && a_dtype == b_dtype && a_stride == b_stride && lda == ldb && !trans_a && trans_b &&
      c_dtype == BASD_DTYPE_F64 && aligned && skip == nullptr && M >= 64 && (M % 64) <= GRAM_RMAX &&
      (size_t)(M % 64) * K * 8 <= 65536) {
    const char* env = getenv("BASD_GRAM_ROWS");
    if (!(env && env[0] == '0')) {
      if (a_dtype == BASD_DTYPE_F32) launch_gram_rows<float>(a, a_stride, lda, c, c_stride, ldc, batch, M, K, st);
      else launch_gram_rows<double>(a, a_stride, lda, c, c_stride, ldc, batch, M, K, st);
      return check_launch("bgemm_f64 (symmetric Gram)");
    }
  }
#define BASD_BG(TA, TB, TC)                                                                                          \
  do {                                                                                                               \
    if (aligned) launch_bgemm_fast<TA, TB, TC>(a, a_stride, lda, trans_a, b, b_stride, ldb, trans_b, c, c_stride, ldc, batch, M, N, K, symmetric, skip, st); \
    else launch_bgemm<TA, TB, TC>(a, a_stride, lda, trans_a, b, b_stride, ldb, trans_b, c, c_stride, ldc, batch, M, N, K, symmetric, skip, st); \
  } while (0)
  switch (key) {
    case 2: BASD_BG(float, float, double); break;      // f32 x f32 -> f64
    case 0: BASD_BG(float, float, float); break;
    case 222: BASD_BG(double, double, double); break;
    case 220: BASD_BG(double, double, float); break;
    case 202: BASD_BG(double, float, double); break;
    case 200: BASD_BG(double, float, float); break;
    case 22: BASD_BG(float, double, double); break;
    case 20: BASD_BG(float, double, float); break;
    default: return fail(BASD_ERR_DTYPE, "bgemm_f64: dtype combination a=%d b=%d c=%d", a_dtype, b_dtype, c_dtype);
  }
#undef BASD_BG
  return check_launch("bgemm_f64");
}
