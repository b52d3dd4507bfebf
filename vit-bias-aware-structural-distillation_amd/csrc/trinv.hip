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
#include <stdlib.h>
#include "basd_common.h"

namespace basd {

__device__ __forceinline__ int tri(int i, int k) { return i * (i + 1) / 2 + k; }

__global__ __launch_bounds__(768) void trinv_kernel(const double* __restrict__ lw_all,
                                                    const int32_t* __restrict__ piv_all,
                                                    const int32_t* __restrict__ rank_all, int n,
                                                    double* __restrict__ out_all, const int32_t* __restrict__ skip) {
  if (skip != nullptr && skip[blockIdx.x] != 0) return;       // masked problem: output left untouched
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

// ---------------------------------------------------------------------------------------------
// Blocked version on the fp64 matrix cores (default).  The kernel above walks the 192 columns one
// by one (two barriers and an LDS-latency-bound dot product per column: 0.4 ms per matrix).  Here
// L_p is held as 16 x 16 blocks and inverted block column by block column:
//   1. X_jj = L_jj^-1 for the 12 diagonal blocks at once (one thread per column, registers),
//   2. W_kj = L_kj X_jj for every sub-diagonal block (independent 16^3 products, one wave each),
//   3. for j = nb-2 .. 0:   X_ij = - sum_{k=j+1..i} X_ik W_kj   (i > j; one wave per row block,
//      v_mfma_f64_16x16x4_f64 chains), which only reads finished block columns > j of X and
//      column j of W: 11 levels with two barriers each instead of 192 steps.
// Same input / output contract as trinv_kernel.
typedef double f64x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int blk_off(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * 256; }

// C (16x16, registers) += A (block at a_off) * B (block at b_off), blocks row-major [16][16] in LDS
__device__ __forceinline__ f64x4t blk_mma(const double* __restrict__ X, int a_off, int b_off, f64x4t acc, int lane) {
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const double av = X[a_off + lr * 16 + 4 * kk + lq];       // A[row = lane & 15][k = 4 kk + (lane >> 4)]
    const double bv = X[b_off + (4 * kk + lq) * 16 + lr];     // B[k][col = lane & 15]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}

__global__ __launch_bounds__(768) void trinv_blocked_kernel(const double* __restrict__ lw_all,
                                                            const int32_t* __restrict__ piv_all,
                                                            const int32_t* __restrict__ rank_all, int n, int ld,
                                                            double* __restrict__ out_all, const int32_t* __restrict__ skip) {
  // n = size of the (leading) block that is inverted here, ld = size / row pitch of the whole matrix (ld > n: the
  // remaining rows are the border, trinv_border_kernel)
  if (skip != nullptr && skip[blockIdx.x] != 0) return;       // masked problem: output left untouched
  extern __shared__ __align__(16) double X[];          // lower block triangle, 16 x 16 row-major blocks
  const int nb = (n + 15) >> 4;
  double* s_rd = X + (size_t)nb * (nb + 1) / 2 * 256;  // [16 nb] reciprocal diagonal
  int* s_piv = reinterpret_cast<int*>(s_rd + 16 * nb); // [16 nb]
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const double* Lw = lw_all + (size_t)blockIdx.x * ld * ld;
  const int32_t* piv = piv_all + (size_t)blockIdx.x * ld;
  const int rank = rank_all[blockIdx.x] < n ? rank_all[blockIdx.x] : n;
  double* out = out_all + (size_t)blockIdx.x * ld * ld;

  for (int i = tid; i < 16 * nb; i += nt) s_piv[i] = (i < n) ? piv[i] : 0;
  __syncthreads();
  // ---- gather L_p[i, k] = Lw[k * n + piv[i]] (k <= i); dead / padding columns -> identity
  // (The scattered 8-byte reads fetch 849 MB per 1024 x 192^2 launch for 151 MB of lower-triangle data, PMC.  Measured
  // in round 3 and rejected: row-contiguous global passes with the permutation done on the LDS side -- inverse
  // permutation in s_piv, rows of Lw read front to back, rows of the result written front to back -- 0.411 vs 0.365 ms:
  // with one workgroup per CU (the 156 KB block triangle) the kernel is bound by its own load -> 11 levels -> store
  // chain, not by the bytes, and the extra LDS passes cost more than the saved fetches bring.)
  const int nblk = nb * (nb + 1) / 2;
  for (int e = tid; e < nblk * 256; e += nt) {
    const int b = e >> 8, r = e & 15, c = (e >> 4) & 15;   // consecutive threads: consecutive rows of a column
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
    const int bj = b - bi * (bi + 1) / 2;
    const int i = 16 * bi + r, k = 16 * bj + c;
    double v = (i == k) ? 1.0 : 0.0;
    if (k < rank && i < n && k <= i) v = Lw[(size_t)k * ld + s_piv[i]];
    X[b * 256 + r * 16 + c] = v;
  }
  __syncthreads();
  if (tid < 16 * nb) s_rd[tid] = 1.0 / X[blk_off(tid >> 4, tid >> 4) + (tid & 15) * 17];
  __syncthreads();
  // ---- 1. diagonal blocks: thread (block j, column c) solves L_jj x = e_c in registers
  double x[16];
  if (tid < 16 * nb) {
    const int j = tid >> 4, c = tid & 15;
    const double* D = X + blk_off(j, j);
    const double* rd = s_rd + 16 * j;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double acc = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < i; ++m) acc = fma(-D[i * 16 + m], x[m], acc);
      x[i] = (i >= c) ? acc * rd[i] : 0.0;
    }
  }
  __syncthreads();
  if (tid < 16 * nb) {
    double* D = X + blk_off(tid >> 4, tid >> 4);
    const int c = tid & 15;
#pragma unroll
    for (int i = 0; i < 16; ++i) D[i * 16 + c] = x[i];
  }
  __syncthreads();
  // ---- 2. W_kj = L_kj X_jj, in place, one wave per block
  const int lr = lane & 15, lq = lane >> 4;
  for (int b = wave; b < nblk; b += nw) {
    int bi = 0;
    while ((bi + 1) * (bi + 2) / 2 <= b) ++bi;
    const int bj = b - bi * (bi + 1) / 2;
    if (bi == bj) continue;                                  // uniform per wave
    f64x4t acc = {0.0, 0.0, 0.0, 0.0};
    acc = blk_mma(X, b * 256, blk_off(bj, bj), acc, lane);   // all operand reads precede the stores below
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) X[b * 256 + (lq + 4 * reg) * 16 + lr] = acc[reg];
  }
  __syncthreads();
  // ---- 3. block columns from the right: X_ij = - sum_k X_ik W_kj
  for (int j = nb - 2; j >= 0; --j) {
    const int i = j + 1 + wave;
    f64x4t acc = {0.0, 0.0, 0.0, 0.0};
    if (i < nb) {
      for (int k = j + 1; k <= i; ++k) acc = blk_mma(X, blk_off(i, k), blk_off(k, j), acc, lane);
    }
    __syncthreads();                                          // every W_kj of this column has been read
    if (i < nb) {
      double* D = X + blk_off(i, j);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) D[(lq + 4 * reg) * 16 + lr] = -acc[reg];
    }
    __syncthreads();
  }
  // ---- out[k, piv[c]] = X[k, c] (c <= k < rank), zero elsewhere
  for (int e = tid; e < n * n; e += nt) {
    const int k = e / n, c = e - k * n;
    double v = 0.0;
    if (c <= k && k < rank) v = X[blk_off(k >> 4, c >> 4) + (k & 15) * 16 + (c & 15)];
    out[(size_t)k * ld + s_piv[c]] = v;
  }
  // the columns of the border pivots are zero in the rows of the leading block (the inverse is lower triangular)
  for (int e = tid; e < n * (ld - n); e += nt) {
    const int k = e / (ld - n), c = n + e - k * (ld - n);
    out[(size_t)k * ld + piv[c]] = 0.0;
  }
}

// Border rows of L_p^-1 P for n_lead < ld <= n_lead + 16 (the 196-token matrices of the wide students: 192 + 4):
// with L_p = [[L11, 0], [L21, L22]],  rows i >= n_lead of the inverse are  (e_i^T - L_p[i, :i] X[:i, :]) / L_p[i, i],
// computed column by column of the OUTPUT layout (out[k, piv[c]] = X[k, c]: the leading rows written by
// trinv_blocked_kernel are read back row-contiguously, one thread per output column).  The unblocked kernel above
// needs 0.84 ms per 512 matrices of 196 x 196; blocked leading part + this border: see DESIGN.md section 5.
__global__ __launch_bounds__(256) void trinv_border_kernel(const double* __restrict__ lw_all,
                                                           const int32_t* __restrict__ piv_all,
                                                           const int32_t* __restrict__ rank_all, int n_lead, int ld,
                                                           double* __restrict__ out_all, const int32_t* __restrict__ skip) {
  if (skip != nullptr && skip[blockIdx.x] != 0) return;
  __shared__ double s_l[16][224];                       // s_l[r][k] = L_p[n_lead + r, k], k <= n_lead + r
  __shared__ int s_pr[16];
  const int tid = threadIdx.x, nbord = ld - n_lead;
  const double* Lw = lw_all + (size_t)blockIdx.x * ld * ld;
  const int32_t* piv = piv_all + (size_t)blockIdx.x * ld;
  const int rank = rank_all[blockIdx.x];
  double* out = out_all + (size_t)blockIdx.x * ld * ld;
  if (tid < nbord) s_pr[tid] = piv[n_lead + tid];
  __syncthreads();
  for (int e = tid; e < nbord * ld; e += 256) {
    const int r = e / ld, k = e - r * ld;
    s_l[r][k] = (k <= n_lead + r) ? Lw[(size_t)k * ld + s_pr[r]] : 0.0;
  }
  __syncthreads();
  const int j = tid;                                     // output column
  if (j >= ld) return;
  double acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = (r < nbord && s_pr[r] == j) ? 1.0 : 0.0;
  for (int k = 0; k < n_lead; ++k) {
    const double x = out[(size_t)k * ld + j];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r < nbord) acc[r] = fma(-s_l[r][k], x, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < nbord) {
      const int i = n_lead + r;
      double v = 0.0;
      if (i < rank) {
        v = acc[r];
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (q < r) v = fma(-s_l[r][n_lead + q], acc[q], v);
        v /= s_l[r][i];
      }
      acc[r] = v;                                        // from here on acc[r] = the finished row's entry of column j
      out[(size_t)i * ld + j] = v;
    }
  }
}

}  // namespace basd

extern "C" int basd_trinv_f64(const double* lwork, const int32_t* piv, const int32_t* rank, int batch, int n,
                              double* out, void* stream) {
  return basd_trinv_f64_masked(lwork, piv, rank, batch, n, out, nullptr, stream);
}

extern "C" int basd_trinv_f64_masked(const double* lwork, const int32_t* piv, const int32_t* rank, int batch, int n,
                                     double* out, const int32_t* skip, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  const size_t lds = (size_t)n * (n + 1) / 2 * 8 + (size_t)n * 4 + 64;
  if (n < 1 || (lds > 160 * 1024 && n > 208))
    return fail(BASD_ERR_SHAPE, "trinv_f64: n=%d does not fit the LDS-resident forms (n <= 208)", n);
  const int nb = (n + 15) / 16;
  const size_t lds_blk = (size_t)nb * (nb + 1) / 2 * 256 * 8 + (size_t)16 * nb * (8 + 4);
  if (nb <= 12 && lds_blk <= 160 * 1024) {
    allow_full_lds((const void*)trinv_blocked_kernel);
    hipLaunchKernelGGL(trinv_blocked_kernel, dim3(batch), dim3(768), lds_blk, (hipStream_t)stream, lwork, piv, rank, n,
                       n, out, skip);
    return check_launch("trinv_f64 (blocked)");
  }
  if (n > 192 && n <= 208) {     // 192 leading rows on the blocked kernel, up to 16 border rows behind it
    const size_t lds12 = (size_t)12 * 13 / 2 * 256 * 8 + (size_t)16 * 12 * (8 + 4);
    allow_full_lds((const void*)trinv_blocked_kernel);
    hipLaunchKernelGGL(trinv_blocked_kernel, dim3(batch), dim3(768), lds12, (hipStream_t)stream, lwork, piv, rank, 192,
                       n, out, skip);
    hipLaunchKernelGGL(trinv_border_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, lwork, piv, rank, 192, n, out,
                       skip);
    return check_launch("trinv_f64 (blocked + border)");
  }
  // 4 lanes per row, rows strided by 192 per pass: two passes cover n <= 384 (v[2])
  allow_full_lds((const void*)trinv_kernel);
  hipLaunchKernelGGL(trinv_kernel, dim3(batch), dim3(768), lds, (hipStream_t)stream, lwork, piv, rank, n, out, skip);
  return check_launch("trinv_f64");
}
