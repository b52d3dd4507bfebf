// Batched one-sided (Hestenes) Jacobi SVD, one matrix per workgroup, matrix
// resident in LDS (column-major, fp32).  Replaces the torch.linalg.svd /
// svdvals / eigvalsh / matrix_norm('nuc') calls of the reference loss path
// (src/losses/layer_selector.py:16,36,92,99; src/losses/relational.py:48).
//
// Layout: column c at W + c*ld, rows 0..m-1 significant, rows m..ld-1 zero.
// Work split: an aligned group of 8 lanes owns one column pair per step; each
// lane holds rows {4*sub + 32*ch + 0..3} of both columns in registers
// (ds_read_b128), the three dot products are reduced across the 8 lanes with
// DPP (no LDS traffic), every lane computes the rotation redundantly and
// writes its rows back.  Pairs of a step are disjoint (round-robin "circle"
// ordering), one workgroup barrier per step.
#include "basd_common.h"

namespace basd {

template <int MAXCH>
__global__ __launch_bounds__(1024) void jacobi_kernel(
    float* __restrict__ wg, int m, int n, int ld, int norm_rows, float tol,
    int max_sweeps, int sort, float* __restrict__ sigma, int32_t* __restrict__ sweeps_out,
    const int32_t* __restrict__ active, int active_rows) {
  extern __shared__ __align__(16) float lds[];
  float* W = lds;
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const size_t mat = (size_t)n * ld;
  float* s_sig = W + mat;                                  // [n]
  int* s_rank = reinterpret_cast<int*>(s_sig + 256);       // [n]
  int* s_flag = s_rank + 256;                              // [2]
  float* src = wg + (size_t)blockIdx.x * mat;

  for (size_t i = tid; i < mat / 4; i += nthreads)
    reinterpret_cast<float4*>(W)[i] = reinterpret_cast<const float4*>(src)[i];
  if (tid < 2) s_flag[tid] = 0;
  __syncthreads();

  // rows m..ld-1 are zero padding (kept zero by any rotation): never loaded / rotated / stored
  // `active` (optional, per matrix): only the leading n_act columns (and, with active_rows, rows)
  // are non-zero -- the rank-masked principal-angle blocks.  The tournament then runs over n_act
  // columns only; the remaining (zero) columns and rows are left untouched.
  int n_act = n;
  if (active) { n_act = active[blockIdx.x]; n_act = n_act < 2 ? 2 : (n_act > n ? n : n_act); }
  int mrows = (m + 3) & ~3;
  if (active && active_rows) { const int ma = (n_act + 3) & ~3; mrows = ma < mrows ? ma : mrows; }
  const int n_even = n_act + (n_act & 1);
  const int R = n_even - 1;
  const int npairs = n_even >> 1;
  const int g = tid >> 3, sub = tid & 7;
  const bool has_pair = g < npairs;
  int used_sweeps = 0;

  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    bool rotated = false;
    for (int t = 0; t < R; ++t) {
      int p, q;
      if (g == 0) { p = R; q = t; }
      else { p = t + g; if (p >= R) p -= R; q = t - g; if (q < 0) q += R; }
      if (has_pair && p < n_act && q < n_act) {
        float* cp = W + (size_t)p * ld + sub * 4;
        float* cq = W + (size_t)q * ld + sub * 4;
        float4 a[MAXCH], b[MAXCH];
        float alpha = 0.f, beta = 0.f, gamma = 0.f;
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
          if (sub * 4 + 32 * ch < mrows) {
            a[ch] = *reinterpret_cast<const float4*>(cp + 32 * ch);
            b[ch] = *reinterpret_cast<const float4*>(cq + 32 * ch);
            alpha = fmaf(a[ch].x, a[ch].x, fmaf(a[ch].y, a[ch].y, fmaf(a[ch].z, a[ch].z, fmaf(a[ch].w, a[ch].w, alpha))));
            beta = fmaf(b[ch].x, b[ch].x, fmaf(b[ch].y, b[ch].y, fmaf(b[ch].z, b[ch].z, fmaf(b[ch].w, b[ch].w, beta))));
            gamma = fmaf(a[ch].x, b[ch].x, fmaf(a[ch].y, b[ch].y, fmaf(a[ch].z, b[ch].z, fmaf(a[ch].w, b[ch].w, gamma))));
          }
        }
        alpha = group8_sum(alpha);
        beta = group8_sum(beta);
        gamma = group8_sum(gamma);
        // |gamma| > tol sqrt(alpha beta), without the sqrt
        if (gamma * gamma > tol * tol * alpha * beta && gamma != 0.f) {
          rotated = true;
          // The rotation ANGLE may be approximate (hardware rcp / sqrt, 1 ulp): any t gives an exact
          // plane rotation as long as (c, s) are consistent.  Only c = (1 + t^2)^(-1/2) is refined
          // (one Newton step on v_rsq_f32) so that c^2 + s^2 = 1 to rounding.  This removes four
          // IEEE divide / sqrt expansions from the per-step dependent chain.
          const float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.f * gamma);
          const float tt = copysignf(1.f, zeta) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.f)));
          const float w1 = fmaf(tt, tt, 1.f);
          float c = __builtin_amdgcn_rsqf(w1);
          c = c * fmaf(-0.5f * w1, c * c, 1.5f);
          const float s = c * tt;
          // Rutishauser form x' = x - s (y + tau x), y' = y + s (x - tau y), tau = s / (1 + c):
          // c = 1 - s*tau is never rounded to 1, so small-angle rotations (t^2 < eps) do not
          // inflate the column norms (a plain c*x - s*y update biased sigma by +2e-5 at n = 192)
          const float tau = s * __builtin_amdgcn_rcpf(1.0f + c);
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            if (sub * 4 + 32 * ch < mrows) {
              float4 na, nb;
              na.x = fmaf(-s, fmaf(tau, a[ch].x, b[ch].x), a[ch].x); nb.x = fmaf(s, fmaf(-tau, b[ch].x, a[ch].x), b[ch].x);
              na.y = fmaf(-s, fmaf(tau, a[ch].y, b[ch].y), a[ch].y); nb.y = fmaf(s, fmaf(-tau, b[ch].y, a[ch].y), b[ch].y);
              na.z = fmaf(-s, fmaf(tau, a[ch].z, b[ch].z), a[ch].z); nb.z = fmaf(s, fmaf(-tau, b[ch].z, a[ch].z), b[ch].z);
              na.w = fmaf(-s, fmaf(tau, a[ch].w, b[ch].w), a[ch].w); nb.w = fmaf(s, fmaf(-tau, b[ch].w, a[ch].w), b[ch].w);
              *reinterpret_cast<float4*>(cp + 32 * ch) = na;
              *reinterpret_cast<float4*>(cq + 32 * ch) = nb;
            }
          }
        }
      }
      __syncthreads();
    }
    used_sweeps = sweep + 1;
    if (rotated) s_flag[sweep & 1] = 1;
    __syncthreads();
    const int any = s_flag[sweep & 1];
    if (tid == 0) s_flag[(sweep + 1) & 1] = 0;
    __syncthreads();
    if (!any) break;
  }

  // column norms over the first norm_rows rows (one 8-lane group per column, strided)
  for (int c = g; c < n; c += (nthreads >> 3)) {
    const float* col = W + (size_t)c * ld;
    float acc = 0.f;
    for (int r = sub; r < norm_rows; r += 8) acc = fmaf(col[r], col[r], acc);
    acc = group8_sum(acc);
    if (sub == 0) s_sig[c] = sqrtf(acc);
  }
  __syncthreads();
  if (tid < n) {
    int rank = tid;
    if (sort) {
      const float mine = s_sig[tid];
      rank = 0;
      for (int c = 0; c < n; ++c) {
        const float o = s_sig[c];
        rank += (o > mine) || (o == mine && c < tid);
      }
    }
    s_rank[tid] = rank;
    sigma[(size_t)blockIdx.x * n + rank] = s_sig[tid];
  }
  __syncthreads();
  // write back (permuted) columns
  for (size_t i = tid; i < mat / 4; i += nthreads) {
    const int c = (int)((i * 4) / ld);
    const int r = (int)((i * 4) - (size_t)c * ld);
    const int dst = s_rank[c];
    *reinterpret_cast<float4*>(src + (size_t)dst * ld + r) = reinterpret_cast<const float4*>(W)[i];
  }
  if (sweeps_out && tid == 0) sweeps_out[blockIdx.x] = used_sweeps;
}

}  // namespace basd

extern "C" int basd_jacobi_svd(float* w, int batch, int m_rows, int n_cols, int ld, int norm_rows,
                               float tol, int max_sweeps, int sort, float* sigma,
                               int32_t* sweeps, const int32_t* active, int active_rows, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n_cols < 1 || n_cols > BASD_JACOBI_MAX_COLS || ld % 4 != 0 || m_rows > ld || m_rows < 1 ||
      norm_rows < 1 || norm_rows > m_rows)
    return fail(BASD_ERR_SHAPE, "jacobi_svd: bad shape m=%d n=%d ld=%d norm_rows=%d", m_rows, n_cols, ld, norm_rows);
  const size_t lds_bytes = (size_t)n_cols * ld * 4 + (256 + 256 + 8) * 4;
  if (lds_bytes > BASD_JACOBI_LDS_BYTES)
    return fail(BASD_ERR_SHAPE, "jacobi_svd: %d x %d (ld %d) needs %zu B of LDS > 160 KiB", m_rows, n_cols, ld, lds_bytes);
  const int npairs = (n_cols + 1) / 2;
  int threads = ((npairs * 8 + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  const int chunks = (((m_rows + 3) & ~3) + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
#define BASD_LAUNCH_JACOBI(MC)                                                                     \
  do {                                                                                             \
    hipFuncSetAttribute((const void*)jacobi_kernel<MC>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                        (int)lds_bytes);                                                           \
    hipLaunchKernelGGL(jacobi_kernel<MC>, dim3(batch), dim3(threads), lds_bytes, st, w, m_rows,    \
                       n_cols, ld, norm_rows, tol, max_sweeps, sort, sigma, sweeps, active,       \
                       active_rows);                                                              \
  } while (0)
  if (chunks <= 2) BASD_LAUNCH_JACOBI(2);
  else if (chunks <= 4) BASD_LAUNCH_JACOBI(4);
  else if (chunks <= 6) BASD_LAUNCH_JACOBI(6);
  else if (chunks <= 7) BASD_LAUNCH_JACOBI(7);
  else if (chunks <= 10) BASD_LAUNCH_JACOBI(10);
  else return fail(BASD_ERR_SHAPE, "jacobi_svd: ld %d > 320 rows unsupported", ld);
#undef BASD_LAUNCH_JACOBI
  return check_launch("jacobi_svd");
}
