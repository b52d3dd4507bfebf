// Diagonal-pivoted Cholesky of batched symmetric PSD fp64 matrices, plus the
// Marchenko-Pastur rank count.  The fp64 factor is what makes the fp32 Jacobi
// that follows (jacobi.hip) accurate: the Gram matrix is formed and factored
// in fp64, the factor has graded columns, and one-sided Jacobi on a
// column-graded factor keeps relative accuracy in the small singular values.
//
// One workgroup (1024 threads = 256 rows x 4 j-slices) per matrix, left-looking:
// step k picks the largest residual diagonal, forms the k-th column
//   v_r = A[r,piv] - sum_{j<k} L[r,j] L[piv,j]
// from the previous fp64 columns (column-major in `lwork`, L2-resident) and
// scales it.  Rows stay in ORIGINAL order (no physical swaps), so
// A = W0 W0^T with W0[:,k] the k-th column; the Jacobi does not care about the
// column order and a consumer that needs the triangular form gathers rows by
// `piv`.
#include "basd_common.h"

namespace basd {

__global__ __launch_bounds__(1024) void pchol_kernel(const double* __restrict__ a_all, int n,
                                                     double tol, float* __restrict__ w0_all, int ld,
                                                     double* __restrict__ lw_all,
                                                     int32_t* __restrict__ piv_all,
                                                     int32_t* __restrict__ rank_all) {
  __shared__ double s_d[256];
  __shared__ double s_prow[256];
  __shared__ double s_part[4][256];
  __shared__ double s_redv[16];
  __shared__ int s_redi[16];
  __shared__ int s_alive[256];
  __shared__ int s_piv;
  __shared__ double s_pivval;
  __shared__ double s_dmax0;

  const int tid = threadIdx.x;
  const int r = tid & 255;       // row
  const int part = tid >> 8;     // j-slice 0..3
  const double* A = a_all + (size_t)blockIdx.x * n * n;
  double* Lw = lw_all + (size_t)blockIdx.x * n * n;
  float* W0 = w0_all + (size_t)blockIdx.x * n * ld;
  int32_t* piv = piv_all + (size_t)blockIdx.x * n;

  if (part == 0) {
    s_d[r] = (r < n) ? A[(size_t)r * n + r] : -1.0;
    s_alive[r] = (r < n);
  }
  __syncthreads();
  int rank = n;
  for (int k = 0; k < n; ++k) {
    // ---- pivot search over alive rows (first 256 threads = 4 waves)
    if (part == 0) {
      double v = s_alive[r] ? s_d[r] : -1.0e300;
      int idx = r;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if ((tid & 63) == 0) { s_redv[tid >> 6] = v; s_redi[tid >> 6] = idx; }
    }
    __syncthreads();
    if (tid == 0) {
      double v = s_redv[0]; int idx = s_redi[0];
      for (int w = 1; w < 4; ++w)
        if (s_redv[w] > v || (s_redv[w] == v && s_redi[w] < idx)) { v = s_redv[w]; idx = s_redi[w]; }
      if (k == 0) s_dmax0 = v;
      s_piv = idx; s_pivval = v;
    }
    __syncthreads();
    const int pv = s_piv;
    if (!(s_pivval > tol * s_dmax0) || !(s_pivval > 0.0)) { rank = k; break; }
    // ---- pivot row of the previous columns
    if (tid < k) s_prow[tid] = Lw[(size_t)tid * n + pv];
    __syncthreads();
    // ---- partial dots over j = part, part+4, ...
    double acc = 0.0;
    if (r < n && s_alive[r]) {
      for (int j = part; j < k; j += 4) acc = fma(Lw[(size_t)j * n + r], s_prow[j], acc);
    }
    s_part[part][r] = acc;
    __syncthreads();
    double col = 0.0;
    if (part == 0 && r < n) {
      if (s_alive[r]) {
        const double v = A[(size_t)pv * n + r] - (s_part[0][r] + s_part[1][r] + s_part[2][r] + s_part[3][r]);
        s_part[0][r] = v;
      }
    }
    __syncthreads();
    if (part == 0 && r < n) {
      const double lkk = sqrt(fmax(s_part[0][pv], 0.0));
      if (s_alive[r]) col = (r == pv) ? lkk : s_part[0][r] / lkk;
      Lw[(size_t)k * n + r] = col;
      W0[(size_t)k * ld + r] = (float)col;
      s_d[r] -= col * col;
      if (r == pv) { s_alive[r] = 0; piv[k] = pv; }
    }
    if (part == 1 && r >= n && r < ld) W0[(size_t)k * ld + r] = 0.f;
    __syncthreads();
  }
  // ---- zero the columns beyond the numerical rank; complete the permutation
  for (int k = rank; k < n; ++k) {
    if (part == 0 && r < n) Lw[(size_t)k * n + r] = 0.0;
    if (part == 0 && r < ld) W0[(size_t)k * ld + r] = 0.f;
  }
  if (tid == 0) {
    rank_all[blockIdx.x] = rank;
    int k = rank;
    for (int i = 0; i < n && k < n; ++i)
      if (s_alive[i]) piv[k++] = i;
  }
}

// Marchenko-Pastur rank, reference src/losses/layer_selector.py:8-20, on device.
__global__ __launch_bounds__(256) void mp_rank_kernel(const float* __restrict__ evals, int n,
                                                      float scale, int64_t rows, int d, int cap,
                                                      int32_t* __restrict__ ranks) {
  __shared__ float s_v[256];
  __shared__ float s_sorted[256];
  __shared__ int s_count;
  const int tid = threadIdx.x;
  const float* e = evals + (size_t)blockIdx.x * n;
  s_v[tid] = (tid < n) ? e[tid] * scale : -1.f;
  if (tid == 0) s_count = 0;
  __syncthreads();
  if (tid < n) {
    const float mine = s_v[tid];
    int rk = 0;   // descending rank
    for (int c = 0; c < n; ++c) rk += (s_v[c] > mine) || (s_v[c] == mine && c < tid);
    s_sorted[rk] = mine;
  }
  __syncthreads();
  // the reference takes eigvalsh of the min(M, D) x min(M, D) Gram: its spectrum is the
  // n_eff largest eigenvalues of the D x D one
  const int n_eff = (int)((rows < (int64_t)n) ? rows : (int64_t)n);
  const int med_desc = n_eff - 1 - (n_eff - 1) / 2;   // lower median in ascending order
  const float sigma2 = s_sorted[med_desc];
  const float q = (float)((double)d / (double)rows);
  const float edge = sigma2 * (1.f + sqrtf(q)) * (1.f + sqrtf(q));
  if (tid < n_eff && s_sorted[tid] > edge) atomicAdd(&s_count, 1);
  __syncthreads();
  if (tid == 0) ranks[blockIdx.x] = s_count < cap ? s_count : cap;
}

}  // namespace basd

extern "C" int basd_pchol_f64(const double* a, int batch, int n, double tol, float* w0, int ld,
                              double* lwork, int32_t* piv, int32_t* rank, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n < 1 || n > 256 || ld < n || ld > 256 + 64)
    return fail(BASD_ERR_SHAPE, "pchol_f64: bad shape n=%d ld=%d", n, ld);
  hipLaunchKernelGGL(pchol_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, a, n, tol, w0,
                     ld, lwork, piv, rank);
  return check_launch("pchol_f64");
}

extern "C" int basd_mp_rank(const float* evals, int batch, int n, int64_t rows, int d, int cap,
                            int32_t* ranks, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n < 1 || n > 256 || rows < 1) return fail(BASD_ERR_SHAPE, "mp_rank: bad shape n=%d", n);
  // eigenvalues handed over are those of X^T X; the reference uses X^T X / M
  hipLaunchKernelGGL(mp_rank_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, evals, n,
                     1.0f / (float)rows, rows, d, cap, ranks);
  return check_launch("mp_rank");
}
