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
#include "basd_common.h"

namespace basd {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int BT = 64;      // tile
constexpr int BK = 16;      // K chunk
constexpr int BLD = BT + 2; // LDS row stride (doubles)

template <typename T>
__device__ __forceinline__ double ld_f64(const T* p) { return (double)(*p); }

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
                                                        int K, int sym) {
  __shared__ double As[BK * BLD];
  __shared__ double Bs[BK * BLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BT, n0 = blockIdx.x * BT;
  if (sym && n0 > m0) return;   // C = X X^T: lower tiles only, mirrored on store
  const TA* A = a + (size_t)blockIdx.z * sa;
  const TB* B = b + (size_t)blockIdx.z * sb;
  TC* C = c + (size_t)blockIdx.z * sc;
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

template <typename TA, typename TB, typename TC>
static void launch_bgemm(const void* a, int64_t sa, int lda, int ta, const void* b, int64_t sb, int ldb, int tb,
                         void* c, int64_t sc, int ldc, int batch, int M, int N, int K, int sym, hipStream_t st) {
  dim3 grid((N + BT - 1) / BT, (M + BT - 1) / BT, batch);
  hipLaunchKernelGGL((bgemm_f64_kernel<TA, TB, TC>), grid, dim3(256), 0, st, (const TA*)a, sa, lda, ta,
                     (const TB*)b, sb, ldb, tb, (TC*)c, sc, ldc, M, N, K, sym);
}

}  // namespace basd

extern "C" int basd_bgemm_f64(const void* a, int a_dtype, int64_t a_stride, int lda, int trans_a, const void* b,
                              int b_dtype, int64_t b_stride, int ldb, int trans_b, void* c, int c_dtype,
                              int64_t c_stride, int ldc, int batch, int M, int N, int K, int symmetric, void* stream) {
  using namespace basd;
  if (batch <= 0 || M <= 0 || N <= 0) return BASD_OK;
  if (symmetric && M != N) return fail(BASD_ERR_SHAPE, "bgemm_f64: symmetric needs M == N");
  if (K <= 0 || batch > 65535) return fail(BASD_ERR_SHAPE, "bgemm_f64: bad shape batch=%d K=%d", batch, K);
  hipStream_t st = (hipStream_t)stream;
  const int key = a_dtype * 100 + b_dtype * 10 + c_dtype;
#define BASD_BG(TA, TB, TC) launch_bgemm<TA, TB, TC>(a, a_stride, lda, trans_a, b, b_stride, ldb, trans_b, c, c_stride, ldc, batch, M, N, K, symmetric, st)
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
