// Batched inverse of the pivoted Cholesky factor, fp64, one matrix per workgroup,
// packed lower triangle resident in LDS (n(n+1)/2 doubles <= 148 KiB at n = 192).
// Replaces torch.linalg.solve_triangular in the Procrustes core: rocBLAS' batched
// dtrsm runs as ~16 small GEMM launches PER MATRIX (16k launches per step in the
// round-1 profile).
//
// Input is the factor as basd_pchol_f64 leaves it (lwork column-major by step, rows
// in original order, plus piv / rank).  With P the pivot permutation, L_p = P L is
// lower triangular; the kernel forms X = L_p^-1 in place (LAPACK dtrti2 recurrence,
// last column first) and writes  out[k, piv[c]] = X[k, c],  i.e. out = L_p^-1 P, so
// that  out @ M  ==  L_p^-1 (P M)  for any M whose rows are in ORIGINAL order: no
// gather of the right-hand sides is needed.  Columns >= rank of L_p are replaced by
// unit columns (their rows of out are zeroed: pseudo-inverse semantics).
#include "basd_common.h"

namespace basd {

__device__ __forceinline__ int tri(int i, int k) { return i * (i + 1) / 2 + k; }

__global__ __launch_bounds__(768) void trinv_kernel(const double* __restrict__ lw_all,
                                                    const int32_t* __restrict__ piv_all,
                                                    const int32_t* __restrict__ rank_all, int n,
                                                    double* __restrict__ out_all) {
  extern __shared__ __align__(16) double X[];          // packed lower triangle
  int* s_piv = reinterpret_cast<int*>(X + (size_t)n * (n + 1) / 2);
  const int tid = threadIdx.x, nt = blockDim.x;
  const double* Lw = lw_all + (size_t)blockIdx.x * n * n;
  const int32_t* piv = piv_all + (size_t)blockIdx.x * n;
  const int rank = rank_all[blockIdx.x];
  double* out = out_all + (size_t)blockIdx.x * n * n;

  for (int i = tid; i < n; i += nt) s_piv[i] = piv[i];
  __syncthreads();
  // gather L_p[i, k] = Lw[k * n + piv[i]]  (k <= i); dead columns -> identity
  for (int e = tid; e < n * n; e += nt) {
    const int k = e / n, i = e - k * n;     // consecutive threads: consecutive i (reads of row k scattered by piv)
    if (k <= i) {
      double v;
      if (k < rank) v = Lw[(size_t)k * n + s_piv[i]];
      else v = (i == k) ? 1.0 : 0.0;
      X[tri(i, k)] = v;
    }
  }
  __syncthreads();
  const int row_l = tid >> 2, part = tid & 3;            // 4 lanes per row
  for (int j = n - 1; j >= 0; --j) {
    const double ajj = 1.0 / X[tri(j, j)];
    // v_i = sum_{k=j+1..i} X[i,k] * L[k,j]   for i = j+1 .. n-1   (rows strided over the workgroup)
    double v[2] = {0.0, 0.0};
    int cnt = 0;
    for (int i = j + 1 + row_l; i < n; i += (nt >> 2), ++cnt) {
      double acc = 0.0;
      const double* xi = X + tri(i, 0);
      for (int k = j + 1 + part; k <= i; k += 4) acc = fma(xi[k], X[tri(k, j)], acc);
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      v[cnt] = acc;
    }
    __syncthreads();                                      // column j is still the ORIGINAL L[:, j] until here
    cnt = 0;
    for (int i = j + 1 + row_l; i < n; i += (nt >> 2), ++cnt)
      if (part == 0) X[tri(i, j)] = -ajj * v[cnt];
    if (tid == 0) X[tri(j, j)] = ajj;
    __syncthreads();
  }
  // out[k, piv[c]] = X[k, c] (c <= k), zero elsewhere; rows k >= rank zeroed
  for (int e = tid; e < n * n; e += nt) {
    const int k = e / n, c = e - k * n;
    const double v = (c <= k && k < rank) ? X[tri(k, c)] : 0.0;
    out[(size_t)k * n + s_piv[c]] = v;
  }
}

}  // namespace basd

extern "C" int basd_trinv_f64(const double* lwork, const int32_t* piv, const int32_t* rank, int batch, int n,
                              double* out, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  const size_t lds = (size_t)n * (n + 1) / 2 * 8 + (size_t)n * 4 + 64;
  if (n < 1 || lds > 160 * 1024)
    return fail(BASD_ERR_SHAPE, "trinv_f64: n=%d does not fit the LDS-resident packed triangle", n);
  // 4 lanes per row, rows strided by 192 per pass: two passes cover n <= 384 (v[2])
  allow_full_lds((const void*)trinv_kernel);
  hipLaunchKernelGGL(trinv_kernel, dim3(batch), dim3(768), lds, (hipStream_t)stream, lwork, piv, rank, n, out);
  return check_launch("trinv_f64");
}
