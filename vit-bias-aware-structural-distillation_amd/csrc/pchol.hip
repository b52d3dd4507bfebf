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
#include <stdlib.h>
#include "basd_common.h"

namespace basd {

// Register-resident version (default for n <= 192): right-looking, the whole residual matrix
// lives in VGPRs -- 768 threads as 24 x 32 tiles of 8 rows x 6 columns (48 doubles per thread) --
// and LDS only carries the scaled pivot row (1.5 KiB) from its owners to everybody.
// Step k:  (A) everyone reads the pivot (p, residual d_p) that wave 0 selected; the 32 threads that
// own row p publish  c_j = R[p][j] / sqrt(d_p)  (exact zeros at positions pivoted before, sqrt(d_p)
// at p);  (B) every thread applies  R[i][j] -= c_i c_j  to its tile (7 LDS reads, 48 fp64 FMAs),
// threads 64..255 store the column (c IS column k of the factor in ORIGINAL row order: no swaps,
// no final scatter) and wave 0 updates the residual diagonal (same fma as the tile update, so it
// stays bit-identical to R[i][i]) and picks the next pivot with DPP row reductions.
// The left-looking LDS kernel above spends its step in a latency-bound dot product of length k
// and a symmetric swap; this one has a fixed, short step (~0.5 us instead of 3.6 us).
__device__ __forceinline__ void dpp_argmax_step(double& v, int& idx, int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v), oi;
  int olo, ohi;
  switch (ctrl_sel) {   // dpp_ctrl must be an immediate
    case 0: olo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); oi = __builtin_amdgcn_update_dpp(idx, idx, 0xB1, 0xF, 0xF, false); break;
    case 1: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); oi = __builtin_amdgcn_update_dpp(idx, idx, 0x4E, 0xF, 0xF, false); break;
    case 2: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); oi = __builtin_amdgcn_update_dpp(idx, idx, 0x141, 0xF, 0xF, false); break;
    default: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); oi = __builtin_amdgcn_update_dpp(idx, idx, 0x140, 0xF, 0xF, false); break;
  }
  const double ov = __hiloint2double(ohi, olo);
  if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
}

// argmax over the 64 lanes of a wave (ties: lowest index); result uniform
__device__ __forceinline__ void wave_argmax(double& v, int& idx) {
  dpp_argmax_step(v, idx, 0);      // quad_perm [1,0,3,2]
  dpp_argmax_step(v, idx, 1);      // quad_perm [2,3,0,1]
  dpp_argmax_step(v, idx, 2);      // row_half_mirror
  dpp_argmax_step(v, idx, 3);      // row_mirror: every lane of a 16-lane row holds the row's winner
  double bv = 0.0; int bi = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), r * 16);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), r * 16);
    const int oi = __builtin_amdgcn_readlane(idx, r * 16);
    const double ov = __hiloint2double(hi, lo);
    if (r == 0 || ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  v = bv; idx = bi;
}

// TC = columns per thread, NTC = column groups: <6, 32> covers 192 x 192 with 24 x 32 threads, <7, 28> covers 196 columns
// (and up to 208 rows) with 26 x 28 = 728 of the 768 threads: the 196-token Gram matrices and cores of the wide students
// (BASELINE c4 / c5), which otherwise take the global-memory kernel below at 4 x the time.
template <int TC, int NTC>
__global__ __launch_bounds__(768) void pchol_reg_kernel(const double* __restrict__ a_all, int n, double tol,
                                                        const double* __restrict__ dmax_ref,
                                                        float* __restrict__ w0_all, int ld,
                                                        double* __restrict__ lw_all,
                                                        int32_t* __restrict__ piv_all,
                                                        int32_t* __restrict__ rank_all, const int32_t* __restrict__ skip) {
  if (skip != nullptr && skip[blockIdx.x] != 0) return;       // masked problem: outputs left untouched
  constexpr int NMAX = (TC * NTC + 15) / 16 * 16;    // 192 | 208: rows / columns the thread grid covers
  constexpr int NQ = (NMAX + 63) / 64;               // diagonal entries per lane of wave 0
  __shared__ __align__(16) double s_c[NMAX];
  __shared__ double s_pv;
  __shared__ int s_pi;
  const int tid = threadIdx.x, lane = tid & 63;
  const int tr = tid / NTC, tc = tid - tr * NTC;   // tile row / tile column (threads beyond the grid hold zero tiles)
  const double* A = a_all + (size_t)blockIdx.x * n * n;
  double* Lw = lw_all + (size_t)blockIdx.x * n * n;
  float* W0 = w0_all + (size_t)blockIdx.x * n * ld;
  int32_t* piv = piv_all + (size_t)blockIdx.x * n;

  double R[8][TC];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = 8 * tr + r;
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const int j = TC * tc + c;
      // only the lower triangle of A is read (row >= column): producers that accumulate lower tiles only
      // (basd_token_gram) need no mirroring pass
      R[r][c] = (i < n && j < n) ? (i >= j ? A[(size_t)i * n + j] : A[(size_t)j * n + i]) : 0.0;
    }
  }
  unsigned cdone = 0;                               // bit c: column 6 tc + c was pivoted (or is padding)
#pragma unroll
  for (int c = 0; c < TC; ++c)
    if (TC * tc + c >= n) cdone |= 1u << c;
  // wave 0: residual diagonal of rows lane, lane + 64, lane + 128
  double d[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) d[q] = 0.0;
  unsigned alive = 0;
  if (tid < 64) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int t = lane + 64 * q;
      if (t < n) { d[q] = A[(size_t)t * n + t]; alive |= 1u << q; }
    }
    double v = -1.0e300; int idx = n;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (((alive >> q) & 1) && d[q] > v) { v = d[q]; idx = lane + 64 * q; }
    wave_argmax(v, idx);
    if (tid == 0) { s_pv = v; s_pi = idx; }
  }
  int rank = n;
  double dmax0 = 0.0;
#pragma unroll 1
  for (int k = 0; k < n; ++k) {
    lds_barrier();                                              // (A) pivot visible, s_c free
    const double pval = s_pv;
    const int p = s_pi;
    if (k == 0) dmax0 = dmax_ref ? dmax_ref[blockIdx.x] : pval;
    if (!(pval > tol * dmax0) || !(pval > 0.0)) { rank = k; break; }
    if (tr == (p >> 3)) {                                       // owners of row p publish it
      const int pr = p & 7;
      double v[TC];
#pragma unroll
      for (int c = 0; c < TC; ++c) v[c] = R[0][c];
#pragma unroll
      for (int r = 1; r < 8; ++r)
        if (pr == r) {
#pragma unroll
          for (int c = 0; c < TC; ++c) v[c] = R[r][c];
        }
      const double lkk = sqrt(pval);
      const double rinv = 1.0 / lkk;
#pragma unroll
      for (int c = 0; c < TC; ++c) {
        double cj = ((cdone >> c) & 1) ? 0.0 : v[c] * rinv;
        if (TC * tc + c == p) cj = lkk;
        if (TC * tc + c < NMAX) s_c[TC * tc + c] = cj;
      }
    }
    {
      const int q = p / TC;
      if (tc == q) cdone |= 1u << (p - TC * q);
    }
    lds_barrier();                                              // (B) column k visible
    double cr[8], cc[TC];
    const int r0 = 8 * tr < NMAX - 8 ? 8 * tr : NMAX - 8;
    {
      const double2* pr2 = reinterpret_cast<const double2*>(s_c + r0);
      const int c0 = (TC * tc < NMAX - TC) ? TC * tc : NMAX - TC;      // (threads beyond the grid: any valid address)
      if (TC % 2 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const double2 t2 = pr2[r]; cr[2 * r] = t2.x; cr[2 * r + 1] = t2.y; }
      }
      if (TC % 2 == 0) {
        const double2* pc2 = reinterpret_cast<const double2*>(s_c + c0);
#pragma unroll
        for (int c = 0; c < TC / 2; ++c) { const double2 t2 = pc2[c]; cc[2 * c] = t2.x; cc[2 * c + 1] = t2.y; }
      } else {
#pragma unroll
        for (int c = 0; c < TC; ++c) cc[c] = s_c[c0 + c];
      }
    }
    if (tid < 64) {
      double v = -1.0e300; int idx = n;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int t = lane + 64 * q;
        const double c = s_c[t < NMAX ? t : 0];
        d[q] = fma(-c, c, d[q]);
        if (t == p) alive &= ~(1u << q);
        if (((alive >> q) & 1) && d[q] > v) { v = d[q]; idx = t; }
      }
      wave_argmax(v, idx);
      if (tid == 0) { s_pv = v; s_pi = idx; piv[k] = p; }
    } else if (tid < 64 + NMAX) {
      const int t = tid - 64;
      if (t < n) {
        const double c = s_c[t];
        Lw[(size_t)k * n + t] = c;
        W0[(size_t)k * ld + t] = (float)c;
      }
    }
    if (TC % 2 == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int c = 0; c < TC; ++c) R[r][c] = fma(-cr[r], cc[c], R[r][c]);
    } else {
      // 56 residual + 7 column values per thread leave no room for the eight row values: they are read pair by pair
      // (the scheduling barrier keeps the compiler from hoisting all four reads)
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2) {
        const double2 t2 = *reinterpret_cast<const double2*>(s_c + r0 + 2 * r2);
#pragma unroll
        for (int c = 0; c < TC; ++c) {
          R[2 * r2][c] = fma(-t2.x, cc[c], R[2 * r2][c]);
          R[2 * r2 + 1][c] = fma(-t2.y, cc[c], R[2 * r2 + 1][c]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // ---- zero the steps beyond the numerical rank and the padding columns; complete the permutation
  for (int e = tid; e < (n - rank) * n; e += blockDim.x) {
    const int kk = rank + e / n, t = e - (e / n) * n;
    Lw[(size_t)kk * n + t] = 0.0;
    W0[(size_t)kk * ld + t] = 0.f;
  }
  if (ld > n)
    for (int e = tid; e < n * (ld - n); e += blockDim.x) {
      const int kk = e / (ld - n), t = n + (e - kk * (ld - n));
      W0[(size_t)kk * ld + t] = 0.f;
    }
  if (tid < 64) {
    int base = rank;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool al = (alive >> q) & 1;
      const unsigned long long m = __ballot(al);
      if (al) piv[base + __popcll(m & ((1ull << lane) - 1ull))] = lane + 64 * q;
      base += __popcll(m);
    }
    if (tid == 0) rank_all[blockIdx.x] = rank;
  }
}

__global__ __launch_bounds__(1024) void pchol_kernel(const double* __restrict__ a_all, int n,
                                                     double tol, const double* __restrict__ dmax_ref,
                                                     float* __restrict__ w0_all, int ld,
                                                     double* __restrict__ lw_all,
                                                     int32_t* __restrict__ piv_all,
                                                     int32_t* __restrict__ rank_all, const int32_t* __restrict__ skip) {
  if (skip != nullptr && skip[blockIdx.x] != 0) return;       // masked problem: outputs left untouched
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
      if (k == 0) s_dmax0 = dmax_ref ? dmax_ref[blockIdx.x] : v;
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
        const double apr = (pv >= r) ? A[(size_t)pv * n + r] : A[(size_t)r * n + pv];     // lower triangle only
        const double v = apr - (s_part[0][r] + s_part[1][r] + s_part[2][r] + s_part[3][r]);
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
__global__ __launch_bounds__(1024) void mp_rank_kernel(const float* __restrict__ evals, int n,
                                                       float scale, int64_t rows, int d, int cap,
                                                       int32_t* __restrict__ ranks, int32_t* __restrict__ status) {
  __shared__ float s_v[1024];
  __shared__ float s_sorted[1024];
  __shared__ int s_count;
  const int tid = threadIdx.x;
  const float* e = evals + (size_t)blockIdx.x * n;
  s_v[tid] = (tid < n) ? e[tid] * scale : -1.f;
  s_sorted[tid] = -1.f;
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
  if (tid == 0) {
    ranks[blockIdx.x] = s_count < cap ? s_count : cap;
    // rank 0 makes the reference divide by sum(sw) = 0 (layer_selector.py:105, NaN weights); a NaN spectrum
    // compares false everywhere and also counts 0
    if (status && s_count == 0) atomicOr(status, (sigma2 == sigma2) ? BASD_STATUS_RANK0 : BASD_STATUS_NONFINITE);
  }
}

}  // namespace basd

extern "C" int basd_pchol_f64(const double* a, int batch, int n, double tol, const double* dmax_ref, float* w0, int ld,
                              double* lwork, int32_t* piv, int32_t* rank, void* stream) {
  return basd_pchol_f64_masked(a, batch, n, tol, dmax_ref, w0, ld, lwork, piv, rank, nullptr, stream);
}

extern "C" int basd_pchol_f64_masked(const double* a, int batch, int n, double tol, const double* dmax_ref, float* w0,
                                     int ld, double* lwork, int32_t* piv, int32_t* rank, const int32_t* skip,
                                     void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n < 1 || n > 256 || ld < n || ld > 256 + 64)
    return fail(BASD_ERR_SHAPE, "pchol_f64: bad shape n=%d ld=%d", n, ld);
  if (n <= 192) {
    hipLaunchKernelGGL((pchol_reg_kernel<6, 32>), dim3(batch), dim3(768), 0, (hipStream_t)stream, a, n, tol, dmax_ref, w0,
                       ld, lwork, piv, rank, skip);
  } else if (n <= 196) {   // the 196 tokens of the wide students
    hipLaunchKernelGGL((pchol_reg_kernel<7, 28>), dim3(batch), dim3(768), 0, (hipStream_t)stream, a, n, tol, dmax_ref, w0,
                       ld, lwork, piv, rank, skip);
  } else {   // global-memory (L2-resident) left-looking kernel for 196 < n <= 256
    hipLaunchKernelGGL(pchol_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, a, n, tol, dmax_ref, w0,
                       ld, lwork, piv, rank, skip);
  }
  return check_launch("pchol_f64");
}

extern "C" int basd_mp_rank(const float* evals, int batch, int n, int64_t rows, int d, int cap,
                            int32_t* ranks, int32_t* status, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n < 1 || n > 1024 || rows < 1) return fail(BASD_ERR_SHAPE, "mp_rank: bad shape n=%d", n);
  // eigenvalues handed over are those of X^T X; the reference uses X^T X / M
  hipLaunchKernelGGL(mp_rank_kernel, dim3(batch), dim3(n <= 256 ? 256 : 1024), 0, (hipStream_t)stream, evals, n,
                     1.0f / (float)rows, rows, d, cap, ranks, status);
  return check_launch("mp_rank");
}
