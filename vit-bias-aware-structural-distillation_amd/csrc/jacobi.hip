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
#include <stdlib.h>
#include "basd_common.h"

namespace basd {

// floor of the squared-norm product in the rotation test (gamma / tol)^2 > max(alpha beta, TINY): below it the product
// is denormal or flushed (both columns under 1e-19: under-scaled input) and the "cosine" would be the quotient of two
// underflowed numbers
#define BASD_JACOBI_TINY 1.0e-37f
// Debris: a column whose squared norm has fallen below DEBRIS x the largest of its matrix (norm ratio 1e-10) is the
// rounding residue of a cancelled direction (rank-deficient input: the pivoted Cholesky factors these kernels are fed
// keep every real column above 3e-7 of the largest).  Its direction is noise, so its cosines with the real columns are
// O(1) again after every rotation that shrinks it: left in the game it asks for LARGE rotations sweep after sweep until
// it underflows (measured: 192 columns of rank 96 used all 40 sweeps).  Such columns are set to zero at the start of a
// sweep (the exact norms and the largest of them are at hand there): sigma = 0, never rotated again.
#define BASD_JACOBI_DEBRIS 1.0e-20f

// Health word (optional, one int32 per launch set, OR-ed with atomics): BASD_STATUS_NONCONVERGED when a matrix used
// all max_sweeps sweeps and was still rotating, BASD_STATUS_NONFINITE when a singular value is NaN / Inf (a NaN input
// makes every rotation test false, so such a matrix "converges" at once and would otherwise pass silently).
// Bits 8 .. 27 of the word are diagnostics of a non-converged solve (informational; OR-ed like the rest, so exact
// for the usual single offender): bits 8-11 the kernel variant (1 LDS-resident, 2 / 3 odd-even with a double / single
// mailbox, 4 two matrices per workgroup, 5 block ordering), bits 12-27 the matrix index (saturating at 65535).
__device__ __forceinline__ void report_status(int32_t* status, bool converged, const float* s_sig, int n, int tid,
                                              int nthreads, int kind, int mat) {
  if (!status) return;
  int st = 0;
  for (int c = tid; c < n; c += nthreads) {
    const float v = s_sig[c];
    if (!(v == v) || v > 3.0e38f) st |= BASD_STATUS_NONFINITE;
  }
  if (tid == 0 && !converged) st |= BASD_STATUS_NONCONVERGED | ((kind & 15) << 8) | ((mat > 65535 ? 65535 : mat) << 12);
  if (st) atomicOr(status, st);
}

template <int MAXCH>
__global__ __launch_bounds__(1024) void jacobi_kernel(
    float* __restrict__ wg, int m, int n, int ld, int norm_rows, float tol,
    int max_sweeps, int sort, float* __restrict__ sigma, int32_t* __restrict__ sweeps_out,
    const int32_t* __restrict__ active, int active_rows, int32_t* __restrict__ status) {
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
  bool converged = false;

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
        // (the cosine is scaled UP by 1 / tol before squaring: tol^2 alpha beta underflows for graded factors -- a
        // column of norm 1e-17 against one of norm 1 -- and the test then degenerates to gamma^2 > 0)
        const float gsc = gamma * (1.0f / tol);
        if (gsc * gsc > fmaxf(alpha * beta, BASD_JACOBI_TINY)) {
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
    if (!any) { converged = true; break; }
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
  if (sweeps_out && tid == 0) sweeps_out[blockIdx.x] = converged ? used_sweeps : -used_sweeps;
  report_status(status, converged, s_sig, n, tid, nthreads, 1, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Register-resident variant (the default when it fits): odd-even transposition ordering.
// The n columns form a line; steps alternate between pairing positions (2k, 2k+1) and
// (2k+1, 2k+2), every rotation is followed by a (logical) swap, and n consecutive steps visit
// every column pair exactly once.  An 8-lane slot keeps its two columns X, Y in VGPRs for the
// whole solve; the rotation itself touches no LDS.  Between steps each slot hands ONE column to a
// neighbour through an LDS mailbox (even->odd: Y goes to slot k-1, odd->even: X goes to slot k+1).
//
// The kernel is bound by the per-step dependent chain (LDS hand-over -> dot -> 8-lane reduction
// -> rotation parameters -> update -> hand-over), not by bandwidth, so NMAT = 2 lets every
// workgroup carry two INDEPENDENT matrices through the same steps: their instruction streams
// interleave in each wave (ILP) and they share the barriers.  NMAT = 1 double-buffers the mailbox
// (one barrier per step); NMAT = 2 uses one mailbox per matrix and two barriers per step.
typedef float v4f __attribute__((ext_vector_type(4)));
// Quadratic-convergence stop: a sweep in which every rotation was SMALL -- |cos| of the pair below QUAD and
// rotation tangent below QUAD_TAN -- is the last one: small-angle rotations at small cosines disturb the other
// pairs of their columns by (sine x cosine) <= 5e-6 each, the residual cosines stay at the tolerance and the next
// sweep would only verify that.  The tangent bound matters: inside a cluster of nearly equal singular values a
// tiny cosine still calls for a 45-degree rotation, and stopping on the cosine alone leaves 1e-4 residuals there
// (measured; scripts/time_jacobi.py has that stress case).  Saves 1-2 of ~9 sweeps at n = 192 on graded spectra
// with identical final orthogonality and singular values.
#define BASD_JACOBI_QUAD 5.0e-4f
#define BASD_JACOBI_QUAD_TAN 1.0e-2f

// x' = x - s (y + tau x),  y' = y + s (x - tau y)  on four rows of a column pair, IN PLACE.  Written as C++ inside the
// `if (rotate)` of the kernels below, the compiler computes the new columns into fresh registers and moves all of them
// back at the merge point (52 v_mov per rotation of 2 x 24 values: a quarter of a rotation's VALU issue cycles, and
// these kernels are VALU-issue-bound); with the outputs tied to the inputs there is nothing to move.  One register
// pair {tau, s} feeds every FMA: op_sel broadcasts one half, neg_lo / neg_hi negate it.
typedef float v2f_rot __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rotate_in_place(v4f& xa, v4f& ya, float tau, float s) {
  const v2f_rot ts = {tau, s};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    v2f_rot x = h ? (v2f_rot){xa.z, xa.w} : (v2f_rot){xa.x, xa.y};
    v2f_rot y = h ? (v2f_rot){ya.z, ya.w} : (v2f_rot){ya.x, ya.y};
    v2f_rot t1, t2;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t1) : "v"(ts), "v"(x), "v"(y));                  // y + tau x
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
        : "=v"(t2) : "v"(ts), "v"(y), "v"(x));                                                                   // x - tau y
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
        : "+v"(x) : "v"(ts), "v"(t1));                                                                           // x - s t1
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(y) : "v"(ts), "v"(t2));          // y + s t2
    if (h) { xa.z = x.x; xa.w = x.y; ya.z = y.x; ya.w = y.y; }
    else { xa.x = x.x; xa.y = x.y; ya.x = y.x; ya.y = y.y; }
  }
}

// NBUF = 2 double-buffers the mailbox (one barrier per step); NBUF = 1 halves its LDS footprint for a second
// barrier per step: what lets columns of up to 384 rows (MAXCH = 12: the block pairs of the D_s = 384 eigensolver,
// 96 slots x 1.5 KiB = 147 KiB) stay register-resident.
template <int MAXCH, int NMAT, int NBUF>
__global__ __launch_bounds__((NMAT == 2 || MAXCH > 7) ? 768 : 1024)
__attribute__((amdgpu_waves_per_eu((NMAT == 2 || MAXCH > 7) ? 3 : 4, (NMAT == 2 || MAXCH > 7) ? 3 : 4))) void jacobi_oe_kernel(
    float* __restrict__ wg, int batch, int m, int n, int ld, int norm_rows, float tol, int max_sweeps, int sort,
    float* __restrict__ sigma, int32_t* __restrict__ sweeps_out, const int32_t* __restrict__ active,
    int active_rows, int32_t* __restrict__ status) {
  extern __shared__ __align__(16) float lds[];
  static_assert(NMAT == 1 || NBUF == 1, "two matrices per workgroup use one mailbox each");
  constexpr int LDM = 32 * MAXCH;                  // mailbox column stride (all MAXCH chunks, no row guards)
  const int tid = threadIdx.x;
  const int k = tid >> 3, sub = tid & 7, roff = sub * 4;
  const int S_all = (n + 1) >> 1;                 // slots that own real columns
  float* mbox = lds;                               // [NMAT][NBUF][S_all][LDM]
  float* s_sig = mbox + (size_t)NMAT * NBUF * S_all * LDM;   // [n + 1]
  int* s_rank = reinterpret_cast<int*>(s_sig + 260);          // [n + 1]
  int* s_id = s_rank + 260;                        // [NMAT][NBUF][132] column ids travelling with the mailbox
  float* s_nrm = reinterpret_cast<float*>(s_id + 2 * 2 * 132);   // [NMAT][NBUF][132] squared norms travelling along
  int* s_flag = reinterpret_cast<int*>(s_nrm + 2 * 2 * 132);     // [2] any rotation, [2] any LARGE rotation

  if (NMAT == 1 && active != nullptr && active[blockIdx.x] < 0) {
    // skipped matrix (a block pair of a tournament whose matrix has already converged): nothing is read or written
    if (sweeps_out && threadIdx.x == 0) sweeps_out[blockIdx.x] = 0;
    return;
  }
  float* src[NMAT];
  int n_act[NMAT], mrows[NMAT], n_e[NMAT], S[NMAT], idX[NMAT], idY[NMAT];
  bool live[NMAT];
  v4f X[NMAT][MAXCH], Y[NMAT][MAXCH];
  float nX[NMAT], nY[NMAT];                        // squared column norms, updated by a' = a - t g, b' = b + t g
  int n_loop = 2;
#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const int mat = blockIdx.x * NMAT + mi;
    const bool exists = mat < batch;
    src[mi] = wg + (size_t)(exists ? mat : 0) * n * ld;
    int na = n;
    if (active && exists) { na = active[mat]; na = na < 2 ? 2 : (na > n ? n : na); }
    n_act[mi] = na;
    int mr = (m + 3) & ~3;
    if (active && active_rows) { const int ma = (na + 3) & ~3; mr = ma < mr ? ma : mr; }
    mrows[mi] = mr;
    n_e[mi] = na + (na & 1);                       // line length (a zero phantom column pads odd n)
    S[mi] = n_e[mi] >> 1;
    live[mi] = exists && (k < S[mi]);
    if (exists && n_e[mi] > n_loop) n_loop = n_e[mi];
    idX[mi] = 2 * k; idY[mi] = 2 * k + 1;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      const int r = roff + 32 * ch;
      const bool in = live[mi] && r < mr;
      X[mi][ch] = (in && idX[mi] < na) ? *reinterpret_cast<const v4f*>(src[mi] + (size_t)idX[mi] * ld + r) : (v4f){0.f, 0.f, 0.f, 0.f};
      Y[mi][ch] = (in && idY[mi] < na) ? *reinterpret_cast<const v4f*>(src[mi] + (size_t)idY[mi] * ld + r) : (v4f){0.f, 0.f, 0.f, 0.f};
    }
  }
  if (tid < 8) s_flag[tid] = 0;                    // [4 + 2 mi + parity]: largest squared column norm of the sweep
  __syncthreads();

  // rotation test |gamma| > tol sqrt(alpha beta) as (gamma / tol)^2 > alpha beta: with the tolerance on the right-hand
  // side (tol^2 alpha beta ~ 7e-13 alpha beta) the product underflows for the graded factors of rank-deficient Gram
  // matrices (a column of norm 1e-17 next to one of norm 1: the debris of cancelled directions) and the test
  // degenerates to gamma^2 > 0 -- such pairs then rotate, and count as LARGE rotations, on noise, sweep after sweep
  // (seen at BASELINE c5: a principal-angle problem of the 768-wide selector used all 40 sweeps once in ~30 steps)
  const float inv_tol = 1.0f / tol;
  int used_sweeps = 0;
  bool converged = false;
  int step = 0;                                    // global step counter: even = (2k, 2k+1) view
#pragma unroll 1
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    bool rotated = false, bigrot = false;
    // exact squared norms once per sweep; inside the sweep they follow the rotation identities
    // (drift ~ n eps only perturbs the rotation angle and the skip test, never the columns)
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {
      float ax = 0.f, ay = 0.f;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        const v4f x = X[mi][ch], y = Y[mi][ch];
        ax = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, ax))));
        ay = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, fmaf(y.w, y.w, ay))));
      }
      nX[mi] = group8_sum(ax);
      nY[mi] = group8_sum(ay);
      if (live[mi] && sub == 0) atomicMax(&s_flag[4 + 2 * mi + (sweep & 1)], __float_as_int(fmaxf(nX[mi], nY[mi])));
    }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < NMAT; ++mi) {            // debris columns become exact zero columns (gamma = 0: never rotated)
      const float debris = __int_as_float(s_flag[4 + 2 * mi + (sweep & 1)]) * BASD_JACOBI_DEBRIS;
      const bool zx = nX[mi] < debris, zy = nY[mi] < debris;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        if (zx) X[mi][ch] = (v4f){0.f, 0.f, 0.f, 0.f};
        if (zy) Y[mi][ch] = (v4f){0.f, 0.f, 0.f, 0.f};
      }
      if (zx) nX[mi] = 0.f;
      if (zy) nY[mi] = 0.f;
    }
#pragma unroll 1
    for (int t = 0; t < n_loop; ++t, ++step) {
      const bool even_view = (step & 1) == 0;
      const int buf = (NBUF == 2) ? (step & 1) : 0;
#pragma unroll
      for (int mi = 0; mi < NMAT; ++mi) {
        // in the odd view the last slot owns a single column (X) and slot 0's old first column idles
        const bool pair_ok = live[mi] && (even_view || k < S[mi] - 1);
        if (pair_ok) {
          const float alpha = nX[mi], beta = nY[mi];
          float gamma = 0.f;
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            const v4f x = X[mi][ch], y = Y[mi][ch];
            gamma = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, gamma))));
          }
          gamma = group8_sum(gamma);
          const float gsc = gamma * inv_tol;
          if (gsc * gsc > fmaxf(alpha * beta, BASD_JACOBI_TINY)) {
            rotated = true;
            const float gq = gamma * (1.0f / BASD_JACOBI_QUAD);
            const bool big_cos = gq * gq > alpha * beta;
            const float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.f * gamma);
            const float tt = copysignf(1.f, zeta) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(fmaf(zeta, zeta, 1.f)));
            bigrot = bigrot || big_cos || fabsf(tt) > BASD_JACOBI_QUAD_TAN;
            const float w1 = fmaf(tt, tt, 1.f);
            float c = __builtin_amdgcn_rsqf(w1);
            c = c * fmaf(-0.5f * w1, c * c, 1.5f);
            const float s = c * tt;
            const float tau = s * __builtin_amdgcn_rcpf(1.0f + c);
            nX[mi] = fmaf(-tt, gamma, alpha);
            nY[mi] = fmaf(tt, gamma, beta);
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) rotate_in_place(X[mi][ch], Y[mi][ch], tau, s);
          }
        }
      }
      // ---- hand one column over.  After the (logical) swap the pair is stored as (lo = Y, hi = X).
      // even -> odd view: position 2k (held in Y) goes to slot k-1; slot 0's copy stays parked in
      //   box[0] until the next odd -> even hand-over; the new Y is slot k+1's old position 2k+2.
      // odd -> even view: position 2k+2 (held in X after the swap; the lone last slot did not swap)
      //   goes to slot k+1; the new X is slot k-1's old position 2k (slot 0: the parked column).
#pragma unroll
      for (int mi = 0; mi < NMAT; ++mi) {
        float* box = mbox + ((size_t)mi * NBUF + buf) * S_all * LDM;
        int* idbox = s_id + (mi * 2 + buf) * 132;
        float* nbox = s_nrm + (mi * 2 + buf) * 132;
        if (even_view) {
          if (live[mi]) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(box + (size_t)k * LDM + roff + 32 * ch) = Y[mi][ch];
            if (sub == 0) { idbox[k] = idY[mi]; nbox[k] = nY[mi]; }
          }
        } else if (live[mi] && k < S[mi] - 1) {
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(box + (size_t)(k + 1) * LDM + roff + 32 * ch) = X[mi][ch];
          if (sub == 0) { idbox[k + 1] = idX[mi]; nbox[k + 1] = nX[mi]; }
        }
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < NMAT; ++mi) {
        const float* box = mbox + ((size_t)mi * NBUF + buf) * S_all * LDM;
        const int* idbox = s_id + (mi * 2 + buf) * 132;
        const float* nbox = s_nrm + (mi * 2 + buf) * 132;
        if (even_view) {
          if (live[mi] && k < S[mi] - 1) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) Y[mi][ch] = *reinterpret_cast<const v4f*>(box + (size_t)(k + 1) * LDM + roff + 32 * ch);
            idY[mi] = idbox[k + 1];
            nY[mi] = nbox[k + 1];
          }
        } else if (live[mi]) {
          const int pbuf = (NBUF == 2) ? (buf ^ 1) : 0;             // where slot 0 parked its column
          const float* rbox = (k == 0) ? mbox + ((size_t)mi * NBUF + pbuf) * S_all * LDM : box + (size_t)k * LDM;
          const int* ridbox = (k == 0) ? s_id + (mi * 2 + pbuf) * 132 : idbox + k;
          const float* rnbox = (k == 0) ? s_nrm + (mi * 2 + pbuf) * 132 : nbox + k;
          const bool lone = (k == S[mi] - 1);    // its single column stayed at position 2k+1 -> becomes Y
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            if (lone) Y[mi][ch] = X[mi][ch];
            X[mi][ch] = *reinterpret_cast<const v4f*>(rbox + roff + 32 * ch);
          }
          if (lone) { idY[mi] = idX[mi]; nY[mi] = nX[mi]; }
          idX[mi] = ridbox[0];
          nX[mi] = rnbox[0];
        }
      }
      if (NBUF == 1) __syncthreads();            // single mailbox: reads done before the next writes
    }
    used_sweeps = sweep + 1;
    if (rotated) s_flag[sweep & 1] = 1;
    if (bigrot) s_flag[2 + (sweep & 1)] = 1;
    __syncthreads();
    const int any = s_flag[sweep & 1], anybig = s_flag[2 + (sweep & 1)];
    if (tid == 0) {
      s_flag[(sweep + 1) & 1] = 0; s_flag[2 + ((sweep + 1) & 1)] = 0;
      s_flag[4 + ((sweep + 1) & 1)] = 0; s_flag[6 + ((sweep + 1) & 1)] = 0;
    }
    __syncthreads();
    if (!any || !anybig) { converged = true; break; }
  }
  // n_loop is even, so the solve ends in the even view: slot k holds positions 2k (X) and 2k+1 (Y)

#pragma unroll
  for (int mi = 0; mi < NMAT; ++mi) {
    const int mat = blockIdx.x * NMAT + mi;
    if (mat >= batch) break;                     // uniform across the workgroup
    // ---- norms over the first norm_rows rows; the phantom column (id == n_act, only when n_act is
    //      odd) gets -1 so that it ranks last and is dropped
    __syncthreads();
    for (int c = tid; c <= n; c += blockDim.x) s_sig[c] = 0.f;
    __syncthreads();
    if (live[mi]) {
      float ax = 0.f, ay = 0.f;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        const int r = roff + 32 * ch;
        const float mx = (r + 0 < norm_rows) ? 1.f : 0.f, my = (r + 1 < norm_rows) ? 1.f : 0.f;
        const float mz = (r + 2 < norm_rows) ? 1.f : 0.f, mw = (r + 3 < norm_rows) ? 1.f : 0.f;
        const v4f x = X[mi][ch], y = Y[mi][ch];
        ax = fmaf(mx * x.x, x.x, fmaf(my * x.y, x.y, fmaf(mz * x.z, x.z, fmaf(mw * x.w, x.w, ax))));
        ay = fmaf(mx * y.x, y.x, fmaf(my * y.y, y.y, fmaf(mz * y.z, y.z, fmaf(mw * y.w, y.w, ay))));
      }
      ax = group8_sum(ax);
      ay = group8_sum(ay);
      if (sub == 0) {
        s_sig[2 * k] = (idX[mi] >= n_act[mi]) ? -1.f : sqrtf(ax);
        s_sig[2 * k + 1] = (idY[mi] >= n_act[mi]) ? -1.f : sqrtf(ay);
      }
    }
    __syncthreads();
    // positions 0 .. n_e-1 hold the active columns (one may be the phantom), n_e .. n-1 are the
    // untouched zero columns of an `active` call; rank all of them (descending, ties by position)
    const int n_tot = (n_e[mi] > n) ? n_e[mi] : n;
    if (tid < n_tot) {
      int rank = tid;
      if (sort) {
        const float mine = s_sig[tid];
        rank = 0;
        for (int c = 0; c < n_tot; ++c) {
          const float o = s_sig[c];
          rank += (o > mine) || (o == mine && c < tid);
        }
      }
      s_rank[tid] = rank;
    }
    __syncthreads();
    {
      // every slot (live or not) writes the two positions it stands for
      const int p0 = 2 * k, p1 = 2 * k + 1;
      int d0 = (p0 < n_tot) ? s_rank[p0] : n, d1 = (p1 < n_tot) ? s_rank[p1] : n;
      if (!sort) { d0 = live[mi] ? idX[mi] : p0; d1 = live[mi] ? idY[mi] : p1; }
      const float sg0 = (p0 < n_tot) ? s_sig[p0] : -1.f, sg1 = (p1 < n_tot) ? s_sig[p1] : -1.f;
      if (k < ((n_tot + 1) >> 1)) {
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
          const int r = roff + 32 * ch;
          if (r < ld) {
            const v4f z4 = (v4f){0.f, 0.f, 0.f, 0.f};
            const bool inr = r < mrows[mi];
            if (d0 >= 0 && d0 < n && p0 < n_tot) *reinterpret_cast<v4f*>(src[mi] + (size_t)d0 * ld + r) = (live[mi] && inr) ? X[mi][ch] : z4;
            if (d1 >= 0 && d1 < n && p1 < n_tot) *reinterpret_cast<v4f*>(src[mi] + (size_t)d1 * ld + r) = (live[mi] && inr) ? Y[mi][ch] : z4;
          }
        }
        if (sub == 0) {
          if (d0 >= 0 && d0 < n && p0 < n_tot) sigma[(size_t)mat * n + d0] = sg0 < 0.f ? 0.f : sg0;
          if (d1 >= 0 && d1 < n && p1 < n_tot) sigma[(size_t)mat * n + d1] = sg1 < 0.f ? 0.f : sg1;
        }
      }
    }
    if (sweeps_out && tid == 0) sweeps_out[mat] = converged ? used_sweeps : -used_sweeps;
    __syncthreads();
    report_status(status, converged, s_sig, n_tot, tid, blockDim.x, NMAT == 2 ? 4 : (NBUF == 2 ? 2 : 3), mat);
  }
}

// ---------------------------------------------------------------------------------------------
// Block ordering (default for large batches): the hand-over chain (write -> wait -> barrier ->
// read -> wait -> barrier, walked in lockstep by every wave) is two thirds of a step of the kernel
// above and does not depend on the bytes moved (ablations).  Here a slot owns
// two BLOCKS of two columns, P = (c0, c1) and Q = (c2, c3); the blocks travel along the odd-even
// transposition line exactly like the single columns above, but every meeting of two blocks performs
// all four cross rotations -- (c0,c2),(c1,c3) then (c0,c3),(c1,c2), two independent rotations at a
// time -- before ONE hand-over of a whole block: four rotations per hand-over instead of one.  The
// pair inside a block is rotated once per sweep.  One matrix per 8 * S-thread workgroup (S = slots =
// a quarter of the columns: 384 threads at n = 192, 96 column VGPRs), single mailbox (74 KB): two
// independent workgroups per CU, so one's hand-over overlaps the other's rotations.
template <int MAXCH>
__global__ __launch_bounds__(MAXCH > 6 ? 448 : 384) __attribute__((amdgpu_waves_per_eu(MAXCH > 6 ? 2 : 3, MAXCH > 6 ? 2 : 3))) void jacobi_blk_kernel(
    float* __restrict__ wg, int batch, int m, int n, int ld, int norm_rows, float tol, int max_sweeps, int sort,
    float* __restrict__ sigma, int32_t* __restrict__ sweeps_out, int32_t* __restrict__ status,
    const int32_t* __restrict__ skip) {
  extern __shared__ __align__(16) float lds[];
  if (skip != nullptr && skip[blockIdx.x] < 0) {   // masked problem (active_rows == 2): nothing is read or written
    if (sweeps_out && threadIdx.x == 0) sweeps_out[blockIdx.x] = 0;
    return;
  }
  constexpr int LDC = 32 * MAXCH;                  // one column in the mailbox
  constexpr int LDB = 2 * LDC;                     // one block
  const int tid = threadIdx.x;
  const int k = tid >> 3, sub = tid & 7, roff = sub * 4;
  const int nb = (n + 1) >> 1;                     // blocks that hold real columns
  const int nbe = nb + (nb & 1);                   // line length in blocks (a zero phantom block pads odd nb)
  const int S = nbe >> 1;                          // slots
  float* mbox = lds;                               // [S + 1][LDB]
  float* s_sig = mbox + (size_t)(S + 1) * LDB;     // [4 S]
  int* s_rank = reinterpret_cast<int*>(s_sig + 264);      // [4 S]
  int* s_id = s_rank + 264;                        // [S][2] column ids travelling with the mailbox
  float* s_nrm = reinterpret_cast<float*>(s_id + 264);    // [S][2] squared norms travelling along
  int* s_flag = reinterpret_cast<int*>(s_nrm + 264);      // [2] any rotation, [2] any LARGE rotation

  const int mat = blockIdx.x;
  float* src = wg + (size_t)mat * n * ld;
  const int mr = (m + 3) & ~3;
  const bool live = k < S;
  v4f C[4][MAXCH];
  int id[4];
  float nr[4];                                     // squared column norms (rotation identities inside a sweep)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    id[c] = 4 * k + c;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      const int r = roff + 32 * ch;
      C[c][ch] = (live && id[c] < n && r < mr) ? *reinterpret_cast<const v4f*>(src + (size_t)id[c] * ld + r)
                                               : (v4f){0.f, 0.f, 0.f, 0.f};
    }
  }
  if (tid < 8) s_flag[tid] = 0;                    // [4 + parity]: largest squared column norm of the sweep
  __syncthreads();

  const float inv_tol = 1.0f / tol;
  bool rotated = false, bigrot = false;
  // one plane rotation of the column pair (A, B); the caller issues two independent ones back to back
  auto rot1 = [&](v4f (&A)[MAXCH], v4f (&B)[MAXCH], float& na, float& nb_, bool ok) {
    float ga = 0.f, gb = 0.f;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      const v4f x = A[ch], y = B[ch];
      ga = fmaf(x.x, y.x, fmaf(x.y, y.y, ga));
      gb = fmaf(x.z, y.z, fmaf(x.w, y.w, gb));
    }
    const float g = group8_sum(ga + gb);
    const float al = na, be = nb_;
    // |g| > tol sqrt(al be) as (g / tol)^2 > al be (see jacobi_oe_kernel); a product below TINY (under-scaled input:
    // both columns below 1e-19) is compared against TINY instead -- a coarser threshold there, never "any g != 0"
    const float gsc = g * inv_tol;
    if (ok && gsc * gsc > fmaxf(al * be, BASD_JACOBI_TINY)) {
      rotated = true;
      const float gq = g * (1.0f / BASD_JACOBI_QUAD);
      const bool big_cos = gq * gq > al * be;
      const float z = (be - al) * __builtin_amdgcn_rcpf(2.f * g);
      const float t = copysignf(1.f, z) * __builtin_amdgcn_rcpf(fabsf(z) + __builtin_amdgcn_sqrtf(fmaf(z, z, 1.f)));
      bigrot = bigrot || big_cos || fabsf(t) > BASD_JACOBI_QUAD_TAN;
      const float w = fmaf(t, t, 1.f);
      float c = __builtin_amdgcn_rsqf(w);
      c = c * fmaf(-0.5f * w, c * c, 1.5f);
      const float sn = c * t;
      const float u = sn * __builtin_amdgcn_rcpf(1.0f + c);
      na = fmaf(-t, g, al);
      nb_ = fmaf(t, g, be);
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) rotate_in_place(A[ch], B[ch], u, sn);
    }
  };
  auto rot2 = [&](v4f (&A0)[MAXCH], v4f (&B0)[MAXCH], float& na0, float& nb0, v4f (&A1)[MAXCH], v4f (&B1)[MAXCH],
                  float& na1, float& nb1, bool ok) {
    rot1(A0, B0, na0, nb0, ok);
    __builtin_amdgcn_sched_barrier(0);             // keep the register live ranges of the inlined copies apart
    rot1(A1, B1, na1, nb1, ok);
    __builtin_amdgcn_sched_barrier(0);
  };

  int used_sweeps = 0;
  bool converged = false;
  int step = 0;                                    // even = blocks (2k, 2k+1); nbe is even, sweeps start and end there
#pragma unroll 1
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    rotated = false;
    bigrot = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {                  // exact squared norms once per sweep
      float a = 0.f;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        const v4f x = C[c][ch];
        a = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, a))));
      }
      nr[c] = group8_sum(a);
    }
    if (live && sub == 0)
      atomicMax(&s_flag[4 + (sweep & 1)], __float_as_int(fmaxf(fmaxf(nr[0], nr[1]), fmaxf(nr[2], nr[3]))));
    __syncthreads();
    {
      const float debris = __int_as_float(s_flag[4 + (sweep & 1)]) * BASD_JACOBI_DEBRIS;
#pragma unroll
      for (int c = 0; c < 4; ++c) {                // debris columns become exact zero columns (never rotated again)
        const bool z = nr[c] < debris;
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch)
          if (z) C[c][ch] = (v4f){0.f, 0.f, 0.f, 0.f};
        if (z) nr[c] = 0.f;
      }
    }
    rot2(C[0], C[1], nr[0], nr[1], C[2], C[3], nr[2], nr[3], live);       // the pair inside each block
#pragma unroll 1
    for (int t = 0; t < nbe; ++t, ++step) {
      const bool even_view = (step & 1) == 0;
      const bool pair_ok = live && (even_view || k < S - 1);
      rot2(C[0], C[2], nr[0], nr[2], C[1], C[3], nr[1], nr[3], pair_ok);
      rot2(C[0], C[3], nr[0], nr[3], C[1], C[2], nr[1], nr[2], pair_ok);
      // ---- hand one block over (after the logical swap the pair is stored as (lo = Q, hi = P)):
      // even -> odd view: Q (block position 2k) goes to slot k-1, slot 0's copy stays parked in box 0;
      // odd -> even view: P (position 2k+2; the lone last slot did not swap) goes to slot k+1.
      // Register loads are the same for every live slot (a block copied or loaded under a per-slot condition
      // makes the register allocator spill hundreds of VGPRs); only the LDS writes are conditional.
      if (even_view) {
        // Q (block position 2k after the swap) goes to slot k-1 through box k (slot 0's stays parked there);
        // the last slot has no right neighbour: it parks its P in box S and takes it back as Q, which is
        // where its lone block of the odd view has to sit for the next even view
        if (live) {
          float* box = mbox + (size_t)k * LDB + roff;
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            *reinterpret_cast<v4f*>(box + 32 * ch) = C[2][ch];
            *reinterpret_cast<v4f*>(box + LDC + 32 * ch) = C[3][ch];
          }
          if (sub == 0) { s_id[2 * k] = id[2]; s_id[2 * k + 1] = id[3]; s_nrm[2 * k] = nr[2]; s_nrm[2 * k + 1] = nr[3]; }
          if (k == S - 1) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) {
              *reinterpret_cast<v4f*>(box + LDB + 32 * ch) = C[0][ch];
              *reinterpret_cast<v4f*>(box + LDB + LDC + 32 * ch) = C[1][ch];
            }
            if (sub == 0) { s_id[2 * k + 2] = id[0]; s_id[2 * k + 3] = id[1]; s_nrm[2 * k + 2] = nr[0]; s_nrm[2 * k + 3] = nr[1]; }
          }
        }
        __syncthreads();
        if (live) {
          const float* box = mbox + (size_t)(k + 1) * LDB + roff;
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            C[2][ch] = *reinterpret_cast<const v4f*>(box + 32 * ch);
            C[3][ch] = *reinterpret_cast<const v4f*>(box + LDC + 32 * ch);
          }
          id[2] = s_id[2 * k + 2]; id[3] = s_id[2 * k + 3];
          nr[2] = s_nrm[2 * k + 2]; nr[3] = s_nrm[2 * k + 3];
        }
      } else {
        // P (position 2k+2 after the swap) goes to slot k+1 through box k+1; slot 0 takes the parked block back
        if (live && k < S - 1) {
          float* box = mbox + (size_t)(k + 1) * LDB + roff;
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            *reinterpret_cast<v4f*>(box + 32 * ch) = C[0][ch];
            *reinterpret_cast<v4f*>(box + LDC + 32 * ch) = C[1][ch];
          }
          if (sub == 0) { s_id[2 * k + 2] = id[0]; s_id[2 * k + 3] = id[1]; s_nrm[2 * k + 2] = nr[0]; s_nrm[2 * k + 3] = nr[1]; }
        }
        __syncthreads();
        if (live) {
          const float* box = mbox + (size_t)k * LDB + roff;
#pragma unroll
          for (int ch = 0; ch < MAXCH; ++ch) {
            C[0][ch] = *reinterpret_cast<const v4f*>(box + 32 * ch);
            C[1][ch] = *reinterpret_cast<const v4f*>(box + LDC + 32 * ch);
          }
          id[0] = s_id[2 * k]; id[1] = s_id[2 * k + 1];
          nr[0] = s_nrm[2 * k]; nr[1] = s_nrm[2 * k + 1];
        }
      }
      __syncthreads();                             // single mailbox: reads done before the next writes
    }
    used_sweeps = sweep + 1;
    if (rotated) s_flag[sweep & 1] = 1;
    if (bigrot) s_flag[2 + (sweep & 1)] = 1;
    __syncthreads();
    const int any = s_flag[sweep & 1], anybig = s_flag[2 + (sweep & 1)];
    if (tid == 0) { s_flag[(sweep + 1) & 1] = 0; s_flag[2 + ((sweep + 1) & 1)] = 0; s_flag[4 + ((sweep + 1) & 1)] = 0; }
    __syncthreads();
    if (!any || !anybig) { converged = true; break; }
  }

  // ---- singular values = column norms over the first norm_rows rows; phantom columns (id >= n) rank last
  {
    float a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float acc = 0.f;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        const int r = roff + 32 * ch;
        const float mx = (r + 0 < norm_rows) ? 1.f : 0.f, my = (r + 1 < norm_rows) ? 1.f : 0.f;
        const float mz = (r + 2 < norm_rows) ? 1.f : 0.f, mw = (r + 3 < norm_rows) ? 1.f : 0.f;
        const v4f x = C[c][ch];
        acc = fmaf(mx * x.x, x.x, fmaf(my * x.y, x.y, fmaf(mz * x.z, x.z, fmaf(mw * x.w, x.w, acc))));
      }
      a[c] = group8_sum(acc);
    }
    if (live && sub == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s_sig[4 * k + c] = (id[c] >= n) ? -1.f : sqrtf(a[c]);
    }
  }
  __syncthreads();
  const int n_tot = 4 * S;                         // >= n; positions beyond the real columns hold -1
  for (int p = tid; p < n_tot; p += blockDim.x) {
    int rank = p;
    if (sort) {
      const float mine = s_sig[p];
      rank = 0;
      for (int c = 0; c < n_tot; ++c) {
        const float o = s_sig[c];
        rank += (o > mine) || (o == mine && c < p);
      }
    }
    s_rank[p] = rank;
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = sort ? s_rank[4 * k + c] : id[c];
      const float sg = s_sig[4 * k + c];
      if (id[c] < n && d < n) {
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
          const int r = roff + 32 * ch;
          if (r < ld) *reinterpret_cast<v4f*>(src + (size_t)d * ld + r) = (r < mr) ? C[c][ch] : (v4f){0.f, 0.f, 0.f, 0.f};
        }
        if (sub == 0) sigma[(size_t)mat * n + d] = sg < 0.f ? 0.f : sg;
      }
    }
  }
  if (sweeps_out && tid == 0) sweeps_out[mat] = converged ? used_sweeps : -used_sweeps;
  report_status(status, converged, s_sig, n_tot, tid, blockDim.x, 5, mat);
}


// ---------------------------------------------------------------------------------------------
// Quad-block ordering with scaled rotations (round 4; the default for batches of up to 192 x 192).
// What the block kernel above still pays per rotation and wave: the rotation parameters (five
// transcendentals, ~25 VALU instructions) are computed by all 8 lanes of a slot, i.e. 8 distinct
// parameter sets per wave instruction stream; the plane rotation is four FMAs per row pair; and the
// mailbox costs two barriers per hand-over.  Here
//   * a slot is 16 lanes (12 rows per lane at 192 rows) and owns two blocks of FOUR columns (the same 96
//     column VGPRs); a meeting of two blocks is 16 cross rotations in four rounds of four INDEPENDENT
//     rotations, one hand-over per 16 rotations (block kernel: per 4);
//   * the four dot products of a round are reduced with a reduce-scatter (11 DPP / select instructions for
//     all four), after which lane q of every quad holds rotation q's sum and computes rotation q's
//     parameters: ONE parameter instruction stream serves four rotations per slot, 16 per wave;
//   * the per-column bookkeeping (squared norm, scale, id) is DISTRIBUTED: lane q of a quad owns column q of
//     either block; a round fetches / returns the partner column's entry with one quad_perm each;
//   * rotations are applied as two shears on scaled columns (x = dx x~, y = dy y~):
//         x~' = x~ - a y~,  y~' = y~ + b x~',   a = t dy / dx,  b = s c dx / dy,  dx' = c dx,  dy' = dy / c
//     -- exactly the rotation x' = c x - s y, y' = s x + c y, two FMAs per row pair instead of four, in place.
//     The scales follow dx' = dx - dx h, dy' = dy + dy h / c with h = 1 - c = s^2 / (1 + c) computed without
//     cancellation: a c that rounds to 1 (t^2 < eps) must not inflate the norms (the bias the Rutishauser form
//     removed from the unscaled kernels); the scales are folded back into the columns once per sweep, where the
//     exact norms are recomputed anyway, so a scale is the product of at most ~190 factors in [0.70, 1.42].
//   * ONE barrier per hand-over: box j is written by slot j (even steps) or slot j - 1 (odd steps) and read by
//     the other one after the barrier; the next write to it comes from the slot that just read it (same lanes,
//     same addresses, LDS operations of a wave execute in order), so no second barrier is needed.
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ int dppi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__host__ __device__ constexpr int qperm(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }

// lane (l & 3) = q of every quad receives the sum over the LANES (16: one DPP row, 32: two) lanes of its slot of p_q
template <int LANES>
__device__ __forceinline__ float reduce_scatter4(float p0, float p1, float p2, float p3, bool bit0, bool bit1) {
  const float keepA = bit0 ? p1 : p0, giveA = bit0 ? p0 : p1;
  const float keepB = bit0 ? p3 : p2, giveB = bit0 ? p2 : p3;
  const float uA = keepA + dppf<qperm(1, 0, 3, 2)>(giveA);
  const float uB = keepB + dppf<qperm(1, 0, 3, 2)>(giveB);
  const float keep = bit1 ? uB : uA, give = bit1 ? uA : uB;
  float v = keep + dppf<qperm(2, 3, 0, 1)>(give);
  // across the four quads: row_ror:8 FIRST, then row_ror:4.  In this order every step adds the same two operands in
  // all lanes that end up with the same sum (l and l + 8, then l and l + 4), so the four quads hold BITWISE identical
  // copies.  (ror:4 then ror:8 pairs the partial sums differently in quads {0, 2} and {1, 3}: replicas that differ in
  // the last bit -- and inside a cluster of equal singular values the sign of beta - alpha, i.e. the direction of a
  // 45-degree rotation, then differs between the rows of one column.)
  v += dppf<0x128>(v);                              // row_ror:8
  v += dppf<0x124>(v);                              // row_ror:4
  if constexpr (LANES == 32) {
    // the slot's two DPP rows: v_permlane16_swap with the same register as both operands leaves (row 0's value, row
    // 1's value) in every lane of the row pair -- the same two operands in the same order in both rows
    typedef unsigned int pl_u32x2 __attribute__((ext_vector_type(2)));
    const pl_u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  return v;
}

// the two shears of a scaled rotation on four rows, in place: ab = {a, b}
__device__ __forceinline__ void shear_in_place(v4f& xa, v4f& ya, v2f_rot ab) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    v2f_rot x = h ? (v2f_rot){xa.z, xa.w} : (v2f_rot){xa.x, xa.y};
    v2f_rot y = h ? (v2f_rot){ya.z, ya.w} : (v2f_rot){ya.x, ya.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "+v"(x) : "v"(ab), "v"(y));   // x - a y
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(y) : "v"(ab), "v"(x));                // y + b x'
    if (h) { xa.z = x.x; xa.w = x.y; ya.z = y.x; ya.w = y.y; }
    else { xa.x = x.x; xa.y = x.y; ya.x = y.x; ya.y = y.y; }
  }
}

// partial dot products (over this lane's rows) of the four column pairs of a round, on the packed FMA, the four
// accumulation chains interleaved.  Written as asm: left to itself the SLP vectoriser packs the four dot products
// ACROSS columns and pays four v_mov per v_pk_fma.
#define BASD_LO2(v) __builtin_shufflevector(v, v, 0, 1)
#define BASD_HI2(v) __builtin_shufflevector(v, v, 2, 3)
template <int MAXCH>
__device__ __forceinline__ void dot4_cols(const v4f (&A0)[MAXCH], const v4f (&B0)[MAXCH], const v4f (&A1)[MAXCH],
                                          const v4f (&B1)[MAXCH], const v4f (&A2)[MAXCH], const v4f (&B2)[MAXCH],
                                          const v4f (&A3)[MAXCH], const v4f (&B3)[MAXCH], float (&p)[4]) {
  v2f_rot a0, a1, a2, a3;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(a0) : "v"(BASD_LO2(A0[0])), "v"(BASD_LO2(B0[0])));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(a1) : "v"(BASD_LO2(A1[0])), "v"(BASD_LO2(B1[0])));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(a2) : "v"(BASD_LO2(A2[0])), "v"(BASD_LO2(B2[0])));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(a3) : "v"(BASD_LO2(A3[0])), "v"(BASD_LO2(B3[0])));
  asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(BASD_HI2(A0[0])), "v"(BASD_HI2(B0[0])));
  asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(BASD_HI2(A1[0])), "v"(BASD_HI2(B1[0])));
  asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(BASD_HI2(A2[0])), "v"(BASD_HI2(B2[0])));
  asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(BASD_HI2(A3[0])), "v"(BASD_HI2(B3[0])));
#pragma unroll
  for (int ch = 1; ch < MAXCH; ++ch) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(BASD_LO2(A0[ch])), "v"(BASD_LO2(B0[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(BASD_LO2(A1[ch])), "v"(BASD_LO2(B1[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(BASD_LO2(A2[ch])), "v"(BASD_LO2(B2[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(BASD_LO2(A3[ch])), "v"(BASD_LO2(B3[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(BASD_HI2(A0[ch])), "v"(BASD_HI2(B0[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(BASD_HI2(A1[ch])), "v"(BASD_HI2(B1[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(BASD_HI2(A2[ch])), "v"(BASD_HI2(B2[ch])));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(BASD_HI2(A3[ch])), "v"(BASD_HI2(B3[ch])));
  }
  p[0] = a0.x + a0.y; p[1] = a1.x + a1.y; p[2] = a2.x + a2.y; p[3] = a3.x + a3.y;
}

// One round = four independent rotations (X_i, Y_i), i = 0..3, column indices 0..7 (0-3 block P, 4-7 block Q)
// covering all eight columns of the slot.  Lane q of a quad computes rotation q.
template <int X0, int Y0, int X1, int Y1, int X2, int Y2, int X3, int Y3>
struct RoundMap {
  __host__ __device__ static constexpr int X(int i) { return i == 0 ? X0 : (i == 1 ? X1 : (i == 2 ? X2 : X3)); }
  __host__ __device__ static constexpr int Y(int i) { return i == 0 ? Y0 : (i == 1 ? Y1 : (i == 2 ? Y2 : Y3)); }
  // gather: lane q takes the entry of column X[q] / Y[q] from its owner lane (column & 3) of register P (column < 4) or Q
  static constexpr int gx = qperm(X0 & 3, X1 & 3, X2 & 3, X3 & 3), gy = qperm(Y0 & 3, Y1 & 3, Y2 & 3, Y3 & 3);
  static constexpr int xq_mask = (X0 >> 2) | ((X1 >> 2) << 1) | ((X2 >> 2) << 2) | ((X3 >> 2) << 3);   // bit q: X[q] in block Q
  static constexpr int yq_mask = (Y0 >> 2) | ((Y1 >> 2) << 1) | ((Y2 >> 2) << 2) | ((Y3 >> 2) << 3);
  // scatter: owner lane j of P[j] (Q[j]) takes the new entry from the rotation that held the column
  static constexpr int src_of(int col) {
    for (int i = 0; i < 4; ++i) if (X(i) == col || Y(i) == col) return i;
    return 0;
  }
  static constexpr bool from_y(int col) {
    for (int i = 0; i < 4; ++i) if (Y(i) == col) return true;
    return false;
  }
  static constexpr int sp = qperm(src_of(0), src_of(1), src_of(2), src_of(3));
  static constexpr int sq = qperm(src_of(4), src_of(5), src_of(6), src_of(7));
  static constexpr int py_mask = (from_y(0) ? 1 : 0) | (from_y(1) ? 2 : 0) | (from_y(2) ? 4 : 0) | (from_y(3) ? 8 : 0);
  static constexpr int qy_mask = (from_y(4) ? 1 : 0) | (from_y(5) ? 2 : 0) | (from_y(6) ? 4 : 0) | (from_y(7) ? 8 : 0);
};
template <int CTRL> __device__ __forceinline__ float qgather(float v) {
  if constexpr (CTRL == qperm(0, 1, 2, 3)) return v;
  else return dppf<CTRL>(v);
}
template <int GCTRL, int QMASK> __device__ __forceinline__ float meta_gather(float mp, float mq, int q) {
  if constexpr (QMASK == 0) return qgather<GCTRL>(mp);
  else if constexpr (QMASK == 15) return qgather<GCTRL>(mq);
  else {
    const float a = qgather<GCTRL>(mp), b = qgather<GCTRL>(mq);
    return ((QMASK >> q) & 1) ? b : a;
  }
}
template <int SCTRL, int YMASK> __device__ __forceinline__ float meta_scatter(float xn, float yn, int q) {
  if constexpr (YMASK == 0) return qgather<SCTRL>(xn);
  else if constexpr (YMASK == 15) return qgather<SCTRL>(yn);
  else {
    const float a = qgather<SCTRL>(xn), b = qgather<SCTRL>(yn);
    return ((YMASK >> q) & 1) ? b : a;
  }
}

// LANES = lanes per slot: 16 (rows 4 sub + 64 ch) or 32 (rows 4 sub + 128 ch: columns of up to 384 rows at the same 96
// column VGPRs, 768 threads -- the block pairs of the D_s = 384 eigensolver).  MAXCH = 4 at 16 lanes (193 .. 256 rows:
// the 196 x 196 token-side Procrustes cores of the wide students) holds 128 column VGPRs: two waves per SIMD.
template <int MAXCH, int LANES>
__global__ __launch_bounds__(LANES == 32 ? 768 : (MAXCH > 3 ? 448 : 384))
__attribute__((amdgpu_waves_per_eu((LANES == 16 && MAXCH > 3) ? 2 : 3, (LANES == 16 && MAXCH > 3) ? 2 : 3))) void jacobi_b4_kernel(
    float* __restrict__ wg, int batch, int m, int n, int ld, int norm_rows, float tol, int max_sweeps, int sort,
    float* __restrict__ sigma, int32_t* __restrict__ sweeps_out, int32_t* __restrict__ status,
    const int32_t* __restrict__ active, int active_mode) {
  // active_mode 0: no `active`; 1: active[mat] = number of leading non-zero columns (the tournament runs over those,
  // the other columns are zero and stay zero); 2: the same and only the leading active[mat] ROWS are non-zero;
  // 3: `active` is a mask (< 0: skip the matrix, otherwise solve it completely)
  extern __shared__ __align__(16) float lds[];
  const int mat = blockIdx.x;
  int n_act = n;
  if (active_mode != 0) {
    const int av = active[mat];
    if (av < 0) {                                  // masked problem (any mode): nothing is read or written
      if (sweeps_out && threadIdx.x == 0) sweeps_out[mat] = 0;
      return;
    }
    if (active_mode != 3) n_act = av < 2 ? 2 : (av > n ? n : av);
  }
  constexpr int CHR = 4 * LANES;                   // rows of one chunk (4 per lane)
  constexpr int LDC = CHR * MAXCH;                 // one column in the mailbox
  constexpr int LDB = 4 * LDC;                     // one block
  const int tid = threadIdx.x;
  const int k = tid / LANES, sub = tid % LANES, roff = sub * 4, q = sub & 3;
  const bool bit0 = (sub & 1) != 0, bit1 = (sub & 2) != 0;
  const int nb_all = (n + 3) >> 2;
  const int S_all = (nb_all + 1) >> 1;             // slots the launch provides (threads / 16, rounded up to waves)
  const int nb = (n_act + 3) >> 2;                 // blocks that hold active columns
  const int nbe = nb + (nb & 1);                   // line length in blocks (a zero phantom block pads odd nb)
  const int S = nbe >> 1;                          // slots that take part
  float* mbox = lds;                               // [S_all + 1][LDB]
  float* s_sig = mbox + (size_t)(S_all + 1) * LDB; // [8 S_all] by position
  int* s_rank = reinterpret_cast<int*>(s_sig + 200);      // [8 S_all]
  float* s_nrm = reinterpret_cast<float*>(s_rank + 200);  // [S_all + 1][4] squared norms travelling with the mailbox
  float* s_scl = s_nrm + 104;                      // [S_all + 1][4] scales
  int* s_flag = reinterpret_cast<int*>(s_scl + 104);      // [2] any rotation, [2] any LARGE rotation, [2] max norm

  float* src = wg + (size_t)mat * n * ld;
  int mr = (m + 3) & ~3;
  if (active_mode == 2) { const int ma = (n_act + 3) & ~3; mr = ma < mr ? ma : mr; }
  const bool live = k < S;
  const bool last = k == S - 1;
  float* const mybox = mbox + (size_t)k * LDB + roff;          // box k, this lane's rows
  float* const mymeta = s_nrm + 4 * k + q;                      // box k, this lane's column: norm, +104 scale
  v4f C[8][MAXCH];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int idc = 8 * k + c;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      const int r = roff + CHR * ch;
      C[c][ch] = (live && idc < n_act && r < mr) ? *reinterpret_cast<const v4f*>(src + (size_t)idc * ld + r)
                                                 : (v4f){0.f, 0.f, 0.f, 0.f};
    }
  }
  // distributed bookkeeping: lane q of every quad owns column q of block P (C[q]) and of block Q (C[4 + q]).
  // Column ids are NOT carried: blocks move as wholes on a data-independent schedule -- a sweep of the odd-even
  // transposition line reverses the block order, so after s sweeps position p holds block (s odd ? nbe - 1 - p : p).
  float nP = 0.f, nQ = 0.f, dP = 1.f, dQ = 1.f;
  if (tid < 8) s_flag[tid] = 0;
  __syncthreads();

  const float inv_tol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / tol)));
  bool rotated = false, bigrot = false;

  auto round = [&](auto map, bool ok) {
    using M = decltype(map);
    float p[4];
    dot4_cols<MAXCH>(C[M::X(0)], C[M::Y(0)], C[M::X(1)], C[M::Y(1)], C[M::X(2)], C[M::Y(2)], C[M::X(3)], C[M::Y(3)], p);
    const float gt = reduce_scatter4<LANES>(p[0], p[1], p[2], p[3], bit0, bit1);
    const float al = meta_gather<M::gx, M::xq_mask>(nP, nQ, q), be = meta_gather<M::gy, M::yq_mask>(nP, nQ, q);
    const float dx = meta_gather<M::gx, M::xq_mask>(dP, dQ, q), dy = meta_gather<M::gy, M::yq_mask>(dP, dQ, q);
    const float g = gt * dx * dy;                  // dot product of the true columns
    const float gsc = g * inv_tol;
    const float ab_ = al * be;
    const bool rot = ok & (gsc * gsc > fmaxf(ab_, BASD_JACOBI_TINY));
    {
      // no branch around the rotation: a round in which all 16 rotations of the wave are below the threshold only
      // occurs on padded / rank-deficient columns (the solve stops after the first sweep of small rotations), and a
      // merge point after the in-place shears costs 48 register moves per round (the allocator does not coalesce the
      // halves of the column registers across it); converged pairs run the shears with a = b = 0, an exact no-op
      rotated |= rot;
      const float gq = g * (1.0f / BASD_JACOBI_QUAD);
      const bool big_cos = gq * gq > ab_;
      const float z = (be - al) * __builtin_amdgcn_rcpf(2.f * g);
      float t = copysignf(1.f, z) * __builtin_amdgcn_rcpf(fabsf(z) + __builtin_amdgcn_sqrtf(fmaf(z, z, 1.f)));
      t = rot ? t : 0.f;
      bigrot |= rot & (big_cos | (fabsf(t) > BASD_JACOBI_QUAD_TAN));
      const float w = fmaf(t, t, 1.f);
      float c = __builtin_amdgcn_rsqf(w);
      c = c * fmaf(-0.5f * w, c * c, 1.5f);
      const float sn = c * t;
      const float u = sn * __builtin_amdgcn_rcpf(1.0f + c);
      const float h = sn * u, hp = t * u;          // 1 - c and 1 / c - 1, no cancellation
      const float aln = fmaf(-t, g, al), ben = fmaf(t, g, be);
      const float a = t * (dy * __builtin_amdgcn_rcpf(dx));
      const float b = (sn * c) * (dx * __builtin_amdgcn_rcpf(dy));
      const float dxn = fmaf(-dx, h, dx), dyn = fmaf(dy, hp, dy);
      nP = meta_scatter<M::sp, M::py_mask>(aln, ben, q);
      nQ = meta_scatter<M::sq, M::qy_mask>(aln, ben, q);
      dP = meta_scatter<M::sp, M::py_mask>(dxn, dyn, q);
      dQ = meta_scatter<M::sq, M::qy_mask>(dxn, dyn, q);
      v2f_rot ab[4];
      ab[0] = (v2f_rot){dppf<qperm(0, 0, 0, 0)>(a), dppf<qperm(0, 0, 0, 0)>(b)};
      ab[1] = (v2f_rot){dppf<qperm(1, 1, 1, 1)>(a), dppf<qperm(1, 1, 1, 1)>(b)};
      ab[2] = (v2f_rot){dppf<qperm(2, 2, 2, 2)>(a), dppf<qperm(2, 2, 2, 2)>(b)};
      ab[3] = (v2f_rot){dppf<qperm(3, 3, 3, 3)>(a), dppf<qperm(3, 3, 3, 3)>(b)};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) shear_in_place(C[M::X(i)][ch], C[M::Y(i)][ch], ab[i]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // fold the scales into the columns (x <- d x~, d <- 1); `kill` zeroes a column instead
  auto fold = [&](bool killP, bool killQ) {
    const float fP = killP ? 0.f : dP, fQ = killQ ? 0.f : dQ;
    float f[8];
    f[0] = dppf<qperm(0, 0, 0, 0)>(fP); f[1] = dppf<qperm(1, 1, 1, 1)>(fP);
    f[2] = dppf<qperm(2, 2, 2, 2)>(fP); f[3] = dppf<qperm(3, 3, 3, 3)>(fP);
    f[4] = dppf<qperm(0, 0, 0, 0)>(fQ); f[5] = dppf<qperm(1, 1, 1, 1)>(fQ);
    f[6] = dppf<qperm(2, 2, 2, 2)>(fQ); f[7] = dppf<qperm(3, 3, 3, 3)>(fQ);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) C[c][ch] *= f[c];
    }
    dP = 1.f; dQ = 1.f;
  };
  auto exact_norms = [&](int rows) {
    float p[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = 0.f;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        const int r = roff + CHR * ch;
        const v4f x = C[c][ch];
        const float mx = (r + 0 < rows) ? 1.f : 0.f, my = (r + 1 < rows) ? 1.f : 0.f;
        const float mz = (r + 2 < rows) ? 1.f : 0.f, mw = (r + 3 < rows) ? 1.f : 0.f;
        a = fmaf(mx * x.x, x.x, fmaf(my * x.y, x.y, fmaf(mz * x.z, x.z, fmaf(mw * x.w, x.w, a))));
      }
      p[c] = a;
    }
    nP = reduce_scatter4<LANES>(p[0], p[1], p[2], p[3], bit0, bit1);
    nQ = reduce_scatter4<LANES>(p[4], p[5], p[6], p[7], bit0, bit1);
  };

  int used_sweeps = 0;
  bool converged = false;
  int step = 0;                                    // even = blocks (2k, 2k+1); nbe is even, sweeps start and end there
#pragma unroll 1
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    rotated = false;
    bigrot = false;
    fold(false, false);
    exact_norms(1 << 30);
    if (live && sub < 4) atomicMax(&s_flag[4 + (sweep & 1)], __float_as_int(fmaxf(nP, nQ)));
    lds_barrier();
    {
      const float debris = __int_as_float(s_flag[4 + (sweep & 1)]) * BASD_JACOBI_DEBRIS;
      const bool zP = nP < debris, zQ = nQ < debris;       // debris columns become exact zero columns
      if (__builtin_amdgcn_ballot_w64(zP || zQ) != 0) {
        fold(zP, zQ);
        if (zP) nP = 0.f;
        if (zQ) nQ = 0.f;
      }
    }
    // the six pairs inside each block, once per sweep: three rounds of two pairs per block
    round(RoundMap<0, 1, 2, 3, 4, 5, 6, 7>{}, live);
    round(RoundMap<0, 2, 1, 3, 4, 6, 5, 7>{}, live);
    round(RoundMap<0, 3, 1, 2, 4, 7, 5, 6>{}, live);
#pragma unroll 1
    for (int t = 0; t < nbe; ++t, ++step) {
      const bool even_view = (step & 1) == 0;
      const bool pair_ok = live & (even_view | !last);
      round(RoundMap<0, 4, 1, 5, 2, 6, 3, 7>{}, pair_ok);
      round(RoundMap<0, 5, 1, 6, 2, 7, 3, 4>{}, pair_ok);
      round(RoundMap<0, 6, 1, 7, 2, 4, 3, 5>{}, pair_ok);
      round(RoundMap<0, 7, 1, 4, 2, 5, 3, 6>{}, pair_ok);
      // ---- hand one block over (positions swap after a meeting: the pair is then stored as lo = Q, hi = P).
      // ONE base address per lane for the boxes and one for their bookkeeping entries; everything else is an
      // immediate offset (box k at 0, box k + 1 at LDB; norms / scales 104 entries apart)
      if (even_view) {
        // Q (block position 2k) goes to slot k-1 through box k (slot 0's stays parked there); the last slot has no
        // right neighbour: it parks its P in box S and takes it back as Q (its lone block of the odd view)
        if (live) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(mybox + c * LDC + CHR * ch) = C[4 + c][ch];
          }
          if (sub < 4) { mymeta[0] = nQ; mymeta[104] = dQ; }
          if (last) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
              for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(mybox + LDB + c * LDC + CHR * ch) = C[c][ch];
            }
            if (sub < 4) { mymeta[4] = nP; mymeta[104 + 4] = dP; }
          }
        }
        lds_barrier();
        if (live) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) C[4 + c][ch] = *reinterpret_cast<const v4f*>(mybox + LDB + c * LDC + CHR * ch);
          }
          nQ = mymeta[4]; dQ = mymeta[104 + 4];
        }
      } else {
        // P (position 2k+2 after the swap) goes to slot k+1 through box k+1; slot 0 takes the parked block back
        if (live && !last) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(mybox + LDB + c * LDC + CHR * ch) = C[c][ch];
          }
          if (sub < 4) { mymeta[4] = nP; mymeta[104 + 4] = dP; }
        }
        lds_barrier();
        if (live) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) C[c][ch] = *reinterpret_cast<const v4f*>(mybox + c * LDC + CHR * ch);
          }
          nP = mymeta[0]; dP = mymeta[104];
        }
      }
      // no second barrier: the next write to a box comes from the slot that has just read it (see the header)
    }
    used_sweeps = sweep + 1;
    if (rotated) s_flag[sweep & 1] = 1;
    if (bigrot) s_flag[2 + (sweep & 1)] = 1;
    lds_barrier();
    const int any = s_flag[sweep & 1], anybig = s_flag[2 + (sweep & 1)];
    if (tid == 0) { s_flag[(sweep + 1) & 1] = 0; s_flag[2 + ((sweep + 1) & 1)] = 0; s_flag[4 + ((sweep + 1) & 1)] = 0; }
    lds_barrier();
    if (!any || !anybig) { converged = true; break; }
  }

  // ---- singular values = norms of the true columns over the first norm_rows rows
  fold(false, false);
  exact_norms(norm_rows);
  lds_barrier();                                   // every wave is out of the sweep loop (mailbox reads done)
  // The thread-derived indices are re-derived here from an opaque copy of the thread id: kept live across the sweep
  // loop they cost registers the rotation rounds need (the allocator spills them into the inner loop otherwise).
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int k2 = tid2 / LANES, sub2 = tid2 % LANES, q2 = sub2 & 3, roff2 = sub2 * 4;
  const bool live2 = k2 < S;
  // column ids by position: live slots hold block (sweeps odd ? nbe - 1 - p : p) at block position p; slots beyond
  // the tournament never moved.  Ranking levels: active columns >= 0, inactive real columns -1, phantoms (id >= n) -2.
  const bool odd = (used_sweeps & 1) != 0;
  const int bP = live2 ? (odd ? nbe - 1 - 2 * k2 : 2 * k2) : 2 * k2;
  const int bQ = live2 ? (odd ? nbe - 2 - 2 * k2 : 2 * k2 + 1) : 2 * k2 + 1;
  if (k2 < S_all && sub2 < 4) {
    const int idP = 4 * bP + q2, idQ = 4 * bQ + q2;
    s_sig[8 * k2 + q2] = idP >= n ? -2.f : ((live2 && idP < n_act) ? sqrtf(nP) : -1.f);
    s_sig[8 * k2 + 4 + q2] = idQ >= n ? -2.f : ((live2 && idQ < n_act) ? sqrtf(nQ) : -1.f);
  }
  lds_barrier();
  const int n_tot = 8 * S_all;                     // >= n
  for (int p = tid2; p < n_tot; p += blockDim.x) {
    int rank = p;
    if (sort) {
      const float mine = s_sig[p];
      rank = 0;
      for (int c = 0; c < n_tot; ++c) {
        const float o = s_sig[c];
        rank += (o > mine) || (o == mine && c < p);
      }
    }
    s_rank[p] = rank;
  }
  lds_barrier();
  if (k2 < S_all) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int idc = 4 * (c < 4 ? bP : bQ) + (c & 3);
      const int d = sort ? s_rank[8 * k2 + c] : idc;
      const float sg = s_sig[8 * k2 + c];
      if (idc < n && d < n) {
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
          const int r = roff2 + CHR * ch;
          if (r < ld) *reinterpret_cast<v4f*>(src + (size_t)d * ld + r) = (live2 && r < mr) ? C[c][ch] : (v4f){0.f, 0.f, 0.f, 0.f};
        }
        if (sub2 == 0) sigma[(size_t)mat * n + d] = sg < 0.f ? 0.f : sg;
      }
    }
  }
  if (sweeps_out && tid2 == 0) sweeps_out[mat] = converged ? used_sweeps : -used_sweeps;
  report_status(status, converged, s_sig, n_tot, tid2, blockDim.x, 6, mat);
}

// ---------------------------------------------------------------------------------------------
// Hex-block ordering (round 4, second step): blocks of SIX columns, 16 slots of 16 lanes for 192 columns -- FOUR
// waves, one per SIMD.  What the quad-block kernel leaves on the table: its 24 slots are six waves on four SIMDs
// (two SIMDs carry two waves, two carry one: the workgroup runs at the pace of the loaded pair), and one
// parameter stream serves four rotations.  Here a meeting of two blocks is 36 cross rotations in six rounds of SIX
// independent rotations (one hand-over per 36), lane q (q = lane & 7 < 6) of each half of a slot computes rotation
// q, and a sweep is 32 steps instead of 48.  The price is 144 column VGPRs (two waves per SIMD at most) and that
// the per-column bookkeeping no longer moves with quad permutes: partner entries travel by ds_bpermute, rotation
// parameters are broadcast with ds_swizzle (lane i of every group of 8) -- 16 LDS-crossbar operations per round,
// no LDS memory.  Everything else (scaled rotations as two shears, scales folded once per sweep, one barrier per
// hand-over, analytic column ids) is the quad-block kernel's.
__device__ __forceinline__ float bperm(int idx4, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(idx4, __float_as_int(v)));
}
template <int I> __device__ __forceinline__ float bcast8(float v) {          // lane I of this lane's group of 8
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (I << 5) | 0x18));
}

// lane (l & 7) = q < 6 of the slot's 16 lanes receives the sum over the 16 lanes of p_q (lanes 6, 7: unspecified)
template <int LANES>
__device__ __forceinline__ float reduce_scatter6(const float (&p)[6], bool bit0, bool bit1, bool bit2) {
  const float k0 = bit0 ? p[1] : p[0], g0 = bit0 ? p[0] : p[1];
  const float k1 = bit0 ? p[3] : p[2], g1 = bit0 ? p[2] : p[3];
  const float k2 = bit0 ? p[5] : p[4], g2 = bit0 ? p[4] : p[5];
  const float u0 = k0 + dppf<qperm(1, 0, 3, 2)>(g0);           // index 0 + bit0
  const float u1 = k1 + dppf<qperm(1, 0, 3, 2)>(g1);           // index 2 + bit0
  const float u2 = k2 + dppf<qperm(1, 0, 3, 2)>(g2);           // index 4 + bit0
  const float ka = bit1 ? u1 : u0, ga = bit1 ? u0 : u1;
  const float kb = bit1 ? 0.f : u2, gb = bit1 ? u2 : 0.f;
  float v0 = ka + dppf<qperm(2, 3, 0, 1)>(ga);                 // index (l & 3), sum over the quad
  float v1 = kb + dppf<qperm(2, 3, 0, 1)>(gb);                 // index 4 + (l & 3) (only 4, 5 exist)
  // all four quads receive the sum over the quads of both (row_ror:8 first: bitwise identical replicas, see
  // reduce_scatter4), then each lane keeps the one its (l & 7) names
  v0 += dppf<0x128>(v0);
  v1 += dppf<0x128>(v1);
  v0 += dppf<0x124>(v0);
  v1 += dppf<0x124>(v1);
  float v = bit2 ? v1 : v0;
  if constexpr (LANES == 32) {                                  // the slot's two DPP rows (see reduce_scatter4)
    typedef unsigned int pl_u32x2 __attribute__((ext_vector_type(2)));
    const pl_u32x2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  return v;
}

template <int MAXCH>
__device__ __forceinline__ void dot6_cols(const v4f (&A0)[MAXCH], const v4f (&B0)[MAXCH], const v4f (&A1)[MAXCH],
                                          const v4f (&B1)[MAXCH], const v4f (&A2)[MAXCH], const v4f (&B2)[MAXCH],
                                          const v4f (&A3)[MAXCH], const v4f (&B3)[MAXCH], const v4f (&A4)[MAXCH],
                                          const v4f (&B4)[MAXCH], const v4f (&A5)[MAXCH], const v4f (&B5)[MAXCH],
                                          float (&p)[6]) {
  v2f_rot a0, a1, a2, a3, a4, a5;
#define BASD_D6_MUL(acc, A, B) asm("v_pk_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(BASD_LO2(A[0])), "v"(BASD_LO2(B[0])))
#define BASD_D6_FMA(acc, a, b) asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
  BASD_D6_MUL(a0, A0, B0); BASD_D6_MUL(a1, A1, B1); BASD_D6_MUL(a2, A2, B2);
  BASD_D6_MUL(a3, A3, B3); BASD_D6_MUL(a4, A4, B4); BASD_D6_MUL(a5, A5, B5);
  BASD_D6_FMA(a0, BASD_HI2(A0[0]), BASD_HI2(B0[0])); BASD_D6_FMA(a1, BASD_HI2(A1[0]), BASD_HI2(B1[0]));
  BASD_D6_FMA(a2, BASD_HI2(A2[0]), BASD_HI2(B2[0])); BASD_D6_FMA(a3, BASD_HI2(A3[0]), BASD_HI2(B3[0]));
  BASD_D6_FMA(a4, BASD_HI2(A4[0]), BASD_HI2(B4[0])); BASD_D6_FMA(a5, BASD_HI2(A5[0]), BASD_HI2(B5[0]));
#pragma unroll
  for (int ch = 1; ch < MAXCH; ++ch) {
    BASD_D6_FMA(a0, BASD_LO2(A0[ch]), BASD_LO2(B0[ch])); BASD_D6_FMA(a1, BASD_LO2(A1[ch]), BASD_LO2(B1[ch]));
    BASD_D6_FMA(a2, BASD_LO2(A2[ch]), BASD_LO2(B2[ch])); BASD_D6_FMA(a3, BASD_LO2(A3[ch]), BASD_LO2(B3[ch]));
    BASD_D6_FMA(a4, BASD_LO2(A4[ch]), BASD_LO2(B4[ch])); BASD_D6_FMA(a5, BASD_LO2(A5[ch]), BASD_LO2(B5[ch]));
    BASD_D6_FMA(a0, BASD_HI2(A0[ch]), BASD_HI2(B0[ch])); BASD_D6_FMA(a1, BASD_HI2(A1[ch]), BASD_HI2(B1[ch]));
    BASD_D6_FMA(a2, BASD_HI2(A2[ch]), BASD_HI2(B2[ch])); BASD_D6_FMA(a3, BASD_HI2(A3[ch]), BASD_HI2(B3[ch]));
    BASD_D6_FMA(a4, BASD_HI2(A4[ch]), BASD_HI2(B4[ch])); BASD_D6_FMA(a5, BASD_HI2(A5[ch]), BASD_HI2(B5[ch]));
  }
#undef BASD_D6_MUL
#undef BASD_D6_FMA
  p[0] = a0.x + a0.y; p[1] = a1.x + a1.y; p[2] = a2.x + a2.y;
  p[3] = a3.x + a3.y; p[4] = a4.x + a4.y; p[5] = a5.x + a5.y;
}

// One round = six independent rotations (X_i, Y_i), column indices 0..11 (0-5 block P, 6-11 block Q) covering all
// twelve columns of the slot; lane i (of each group of 8) computes rotation i.  The tables are 3 bits per lane:
// which lane of the group owns the entry a lane needs (gather) or produced the entry a lane owns (scatter).
template <int X0, int Y0, int X1, int Y1, int X2, int Y2, int X3, int Y3, int X4, int Y4, int X5, int Y5>
struct Round6 {
  __host__ __device__ static constexpr int X(int i) {
    return i == 0 ? X0 : (i == 1 ? X1 : (i == 2 ? X2 : (i == 3 ? X3 : (i == 4 ? X4 : X5))));
  }
  __host__ __device__ static constexpr int Y(int i) {
    return i == 0 ? Y0 : (i == 1 ? Y1 : (i == 2 ? Y2 : (i == 3 ? Y3 : (i == 4 ? Y4 : Y5))));
  }
  static constexpr int src_of(int col) {
    for (int i = 0; i < 6; ++i) if (X(i) == col || Y(i) == col) return i;
    return 0;
  }
  static constexpr bool from_y(int col) {
    for (int i = 0; i < 6; ++i) if (Y(i) == col) return true;
    return false;
  }
  static constexpr unsigned IDENT = 0 | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12) | (5u << 15) | (6u << 18) | (7u << 21);
  static constexpr unsigned tail = (6u << 18) | (7u << 21);    // lanes 6, 7 of a group point at themselves
  static constexpr unsigned gx_tbl() { unsigned t = tail; for (int i = 0; i < 6; ++i) t |= (unsigned)(X(i) % 6) << (3 * i); return t; }
  static constexpr unsigned gy_tbl() { unsigned t = tail; for (int i = 0; i < 6; ++i) t |= (unsigned)(Y(i) % 6) << (3 * i); return t; }
  static constexpr unsigned xq_msk() { unsigned t = 0; for (int i = 0; i < 6; ++i) t |= (unsigned)(X(i) / 6) << i; return t; }
  static constexpr unsigned yq_msk() { unsigned t = 0; for (int i = 0; i < 6; ++i) t |= (unsigned)(Y(i) / 6) << i; return t; }
  static constexpr unsigned sp_tbl() { unsigned t = tail; for (int j = 0; j < 6; ++j) t |= (unsigned)src_of(j) << (3 * j); return t; }
  static constexpr unsigned sq_tbl() { unsigned t = tail; for (int j = 0; j < 6; ++j) t |= (unsigned)src_of(6 + j) << (3 * j); return t; }
  static constexpr unsigned py_msk() { unsigned t = 0; for (int j = 0; j < 6; ++j) t |= (from_y(j) ? 1u : 0u) << j; return t; }
  static constexpr unsigned qy_msk() { unsigned t = 0; for (int j = 0; j < 6; ++j) t |= (from_y(6 + j) ? 1u : 0u) << j; return t; }
  static constexpr unsigned gx = gx_tbl(), gy = gy_tbl(), xq = xq_msk(), yq = yq_msk();
  static constexpr unsigned sp = sp_tbl(), sq = sq_tbl(), py = py_msk(), qy = qy_msk();
};
// entry of the lane the table names: register mp where the mask bit of this lane is 0, mq where it is 1
template <unsigned TBL, unsigned MASK, unsigned IDENT>
__device__ __forceinline__ float meta_move6(float mp, float mq, int q8, int idx4) {
  if constexpr (TBL == IDENT) {
    if constexpr (MASK == 0) return mp;
    else if constexpr (MASK == 0x3F) return mq;
    else return ((MASK >> q8) & 1) ? mq : mp;
  } else {
    if constexpr (MASK == 0) return bperm(idx4, mp);
    else if constexpr (MASK == 0x3F) return bperm(idx4, mq);
    else {
      const float a = bperm(idx4, mp), b = bperm(idx4, mq);
      return ((MASK >> q8) & 1) ? b : a;
    }
  }
}

// OCC = waves per SIMD the register allocation aims at: 1 (a batch that leaves every CU at most one matrix at a time:
// no spills, the latency of the parameter stream is all that counts) or 2 (two matrices per CU hide each other's
// parameter stream)
// LANES = 32: columns of up to 384 rows at the same 144 column VGPRs, eight waves (the block pairs of the D_s = 384
// eigensolver).
template <int MAXCH, int OCC, int LANES>
__global__ __launch_bounds__(16 * LANES) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void jacobi_b6_kernel(
    float* __restrict__ wg, int batch, int m, int n, int ld, int norm_rows, float tol, int max_sweeps, int sort,
    float* __restrict__ sigma, int32_t* __restrict__ sweeps_out, int32_t* __restrict__ status,
    const int32_t* __restrict__ active, int active_mode) {
  // active_mode as in jacobi_b4_kernel
  extern __shared__ __align__(16) float lds[];
  const int mat = blockIdx.x;
  int n_act = n;
  if (active_mode != 0) {
    const int av = active[mat];
    if (av < 0) {
      if (sweeps_out && threadIdx.x == 0) sweeps_out[mat] = 0;
      return;
    }
    if (active_mode != 3) n_act = av < 2 ? 2 : (av > n ? n : av);
  }
  constexpr int CHR = 4 * LANES;
  constexpr int LDC = CHR * MAXCH;
  constexpr int LDB = 6 * LDC;
  const int tid = threadIdx.x;
  const int k = tid / LANES, sub = tid % LANES, roff = sub * 4, q8 = sub & 7;
  const bool bit0 = (sub & 1) != 0, bit1 = (sub & 2) != 0, bit2 = (sub & 4) != 0;
  const bool rlane = q8 < 6;                       // this lane computes a rotation / owns a column entry
  const int grp4 = ((tid & 63) & ~7) << 2;         // byte index of lane 0 of this lane's group of 8 (ds_bpermute)
  const int nb_all = (n + 5) / 6;
  const int S_all = (nb_all + 1) >> 1;
  const int nb = (n_act + 5) / 6;
  const int nbe = nb + (nb & 1);
  const int S = nbe >> 1;
  float* mbox = lds;                               // [S_all + 1][LDB]
  float* s_sig = mbox + (size_t)(S_all + 1) * LDB; // [12 S_all] by position
  int* s_rank = reinterpret_cast<int*>(s_sig + 200);
  float* s_nrm = reinterpret_cast<float*>(s_rank + 200);  // [S_all + 1][6]
  float* s_scl = s_nrm + 104;
  int* s_flag = reinterpret_cast<int*>(s_scl + 104);

  float* src = wg + (size_t)mat * n * ld;
  int mr = (m + 3) & ~3;
  if (active_mode == 2) { const int ma = (n_act + 3) & ~3; mr = ma < mr ? ma : mr; }
  const bool live = k < S;
  const bool last = k == S - 1;
  float* const mybox = mbox + (size_t)k * LDB + roff;
  float* const mymeta = s_nrm + 6 * k + (rlane ? q8 : 5);
  v4f C[12][MAXCH];
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const int idc = 12 * k + c;
#pragma unroll
    for (int ch = 0; ch < MAXCH; ++ch) {
      const int r = roff + CHR * ch;
      C[c][ch] = (live && idc < n_act && r < mr) ? *reinterpret_cast<const v4f*>(src + (size_t)idc * ld + r)
                                                 : (v4f){0.f, 0.f, 0.f, 0.f};
    }
  }
  float nP = 0.f, nQ = 0.f, dP = 1.f, dQ = 1.f;
  if (tid < 8) s_flag[tid] = 0;
  __syncthreads();

  const float inv_tol = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(1.0f / tol)));
  bool rotated = false, bigrot = false;

  auto round = [&](auto map, bool ok) {
    using M = decltype(map);
    // partner entries first: the crossbar latency hides under the dot products
    const int ixx = grp4 | (((M::gx >> (3 * q8)) & 7) << 2), ixy = grp4 | (((M::gy >> (3 * q8)) & 7) << 2);
    const float al = meta_move6<M::gx, M::xq, M::IDENT>(nP, nQ, q8, ixx), be = meta_move6<M::gy, M::yq, M::IDENT>(nP, nQ, q8, ixy);
    const float dx = meta_move6<M::gx, M::xq, M::IDENT>(dP, dQ, q8, ixx), dy = meta_move6<M::gy, M::yq, M::IDENT>(dP, dQ, q8, ixy);
    float p[6];
    dot6_cols<MAXCH>(C[M::X(0)], C[M::Y(0)], C[M::X(1)], C[M::Y(1)], C[M::X(2)], C[M::Y(2)], C[M::X(3)], C[M::Y(3)],
                     C[M::X(4)], C[M::Y(4)], C[M::X(5)], C[M::Y(5)], p);
    const float gt = reduce_scatter6<LANES>(p, bit0, bit1, bit2);
    const float g = gt * dx * dy;
    const float gsc = g * inv_tol;
    const float ab_ = al * be;
    const bool rot = ok & rlane & (gsc * gsc > fmaxf(ab_, BASD_JACOBI_TINY));
    rotated |= rot;
    const float gq = g * (1.0f / BASD_JACOBI_QUAD);
    const bool big_cos = gq * gq > ab_;
    const float z = (be - al) * __builtin_amdgcn_rcpf(2.f * g);
    float t = copysignf(1.f, z) * __builtin_amdgcn_rcpf(fabsf(z) + __builtin_amdgcn_sqrtf(fmaf(z, z, 1.f)));
    t = rot ? t : 0.f;
    bigrot |= rot & (big_cos | (fabsf(t) > BASD_JACOBI_QUAD_TAN));
    const float w = fmaf(t, t, 1.f);
    float c = __builtin_amdgcn_rsqf(w);
    c = c * fmaf(-0.5f * w, c * c, 1.5f);
    const float sn = c * t;
    const float u = sn * __builtin_amdgcn_rcpf(1.0f + c);
    const float h = sn * u, hp = t * u;
    const float aln = fmaf(-t, g, al), ben = fmaf(t, g, be);
    const float a = t * (dy * __builtin_amdgcn_rcpf(dx));
    const float b = (sn * c) * (dx * __builtin_amdgcn_rcpf(dy));
    const float dxn = fmaf(-dx, h, dx), dyn = fmaf(dy, hp, dy);
    // (measured: the six (a, b) pairs through 96 bytes of LDS per slot -- one 8-byte write by the parameter lanes, three
    // 16-byte broadcast reads by all lanes -- instead of these twelve swizzles: 3.335 vs 3.32 ms per 1024 x 192^2,
    // 0.97 vs 0.98 for a single matrix: the crossbar operations are not what a round waits for; not kept)
    v2f_rot ab[6];
    ab[0] = (v2f_rot){bcast8<0>(a), bcast8<0>(b)};
    ab[1] = (v2f_rot){bcast8<1>(a), bcast8<1>(b)};
    ab[2] = (v2f_rot){bcast8<2>(a), bcast8<2>(b)};
    ab[3] = (v2f_rot){bcast8<3>(a), bcast8<3>(b)};
    ab[4] = (v2f_rot){bcast8<4>(a), bcast8<4>(b)};
    ab[5] = (v2f_rot){bcast8<5>(a), bcast8<5>(b)};
    const int isp = grp4 | (((M::sp >> (3 * q8)) & 7) << 2), isq = grp4 | (((M::sq >> (3 * q8)) & 7) << 2);
    nP = meta_move6<M::sp, M::py, M::IDENT>(aln, ben, q8, isp);
    nQ = meta_move6<M::sq, M::qy, M::IDENT>(aln, ben, q8, isq);
    dP = meta_move6<M::sp, M::py, M::IDENT>(dxn, dyn, q8, isp);
    dQ = meta_move6<M::sq, M::qy, M::IDENT>(dxn, dyn, q8, isq);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) shear_in_place(C[M::X(i)][ch], C[M::Y(i)][ch], ab[i]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto fold = [&](bool killP, bool killQ) {
    const float fP = killP ? 0.f : dP, fQ = killQ ? 0.f : dQ;
    float f[12];
    f[0] = bcast8<0>(fP); f[1] = bcast8<1>(fP); f[2] = bcast8<2>(fP);
    f[3] = bcast8<3>(fP); f[4] = bcast8<4>(fP); f[5] = bcast8<5>(fP);
    f[6] = bcast8<0>(fQ); f[7] = bcast8<1>(fQ); f[8] = bcast8<2>(fQ);
    f[9] = bcast8<3>(fQ); f[10] = bcast8<4>(fQ); f[11] = bcast8<5>(fQ);
#pragma unroll
    for (int c = 0; c < 12; ++c) {
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) C[c][ch] *= f[c];
    }
    dP = 1.f; dQ = 1.f;
  };
  auto exact_norms = [&](int rows) {
    float p[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      float a = 0.f;
#pragma unroll
      for (int ch = 0; ch < MAXCH; ++ch) {
        const int r = roff + CHR * ch;
        const v4f x = C[c][ch];
        const float mx = (r + 0 < rows) ? 1.f : 0.f, my = (r + 1 < rows) ? 1.f : 0.f;
        const float mz = (r + 2 < rows) ? 1.f : 0.f, mw = (r + 3 < rows) ? 1.f : 0.f;
        a = fmaf(mx * x.x, x.x, fmaf(my * x.y, x.y, fmaf(mz * x.z, x.z, fmaf(mw * x.w, x.w, a))));
      }
      p[c] = a;
    }
    const float pp[6] = {p[0], p[1], p[2], p[3], p[4], p[5]}, pq[6] = {p[6], p[7], p[8], p[9], p[10], p[11]};
    nP = reduce_scatter6<LANES>(pp, bit0, bit1, bit2);
    nQ = reduce_scatter6<LANES>(pq, bit0, bit1, bit2);
    if (!rlane) { nP = 0.f; nQ = 0.f; }
  };

  int used_sweeps = 0;
  bool converged = false;
  int step = 0;
#pragma unroll 1
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    rotated = false;
    bigrot = false;
    fold(false, false);
    exact_norms(1 << 30);
    if (live && rlane) atomicMax(&s_flag[4 + (sweep & 1)], __float_as_int(fmaxf(nP, nQ)));
    lds_barrier();
    {
      const float debris = __int_as_float(s_flag[4 + (sweep & 1)]) * BASD_JACOBI_DEBRIS;
      const bool zP = rlane & (nP < debris), zQ = rlane & (nQ < debris);
      if (__builtin_amdgcn_ballot_w64(zP || zQ) != 0) {
        fold(zP, zQ);
        if (zP) nP = 0.f;
        if (zQ) nQ = 0.f;
      }
    }
    // the fifteen pairs inside each block, once per sweep: five rounds of three pairs per block
    round(Round6<0, 5, 1, 4, 2, 3, 6, 11, 7, 10, 8, 9>{}, live);
    round(Round6<0, 4, 5, 3, 1, 2, 6, 10, 11, 9, 7, 8>{}, live);
    round(Round6<0, 3, 4, 2, 5, 1, 6, 9, 10, 8, 11, 7>{}, live);
    round(Round6<0, 2, 3, 1, 4, 5, 6, 8, 9, 7, 10, 11>{}, live);
    round(Round6<0, 1, 2, 5, 3, 4, 6, 7, 8, 11, 9, 10>{}, live);
#pragma unroll 1
    for (int t = 0; t < nbe; ++t, ++step) {
      const bool even_view = (step & 1) == 0;
      const bool pair_ok = live & (even_view | !last);
      round(Round6<0, 6, 1, 7, 2, 8, 3, 9, 4, 10, 5, 11>{}, pair_ok);
      round(Round6<0, 7, 1, 8, 2, 9, 3, 10, 4, 11, 5, 6>{}, pair_ok);
      round(Round6<0, 8, 1, 9, 2, 10, 3, 11, 4, 6, 5, 7>{}, pair_ok);
      round(Round6<0, 9, 1, 10, 2, 11, 3, 6, 4, 7, 5, 8>{}, pair_ok);
      round(Round6<0, 10, 1, 11, 2, 6, 3, 7, 4, 8, 5, 9>{}, pair_ok);
      round(Round6<0, 11, 1, 6, 2, 7, 3, 8, 4, 9, 5, 10>{}, pair_ok);
      if (even_view) {
        if (live) {
#pragma unroll
          for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(mybox + c * LDC + CHR * ch) = C[6 + c][ch];
          }
          if (sub < 6) { mymeta[0] = nQ; mymeta[104] = dQ; }
          if (last) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
#pragma unroll
              for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(mybox + LDB + c * LDC + CHR * ch) = C[c][ch];
            }
            if (sub < 6) { mymeta[6] = nP; mymeta[104 + 6] = dP; }
          }
        }
        lds_barrier();
        if (live) {
#pragma unroll
          for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) C[6 + c][ch] = *reinterpret_cast<const v4f*>(mybox + LDB + c * LDC + CHR * ch);
          }
          nQ = mymeta[6]; dQ = mymeta[104 + 6];
        }
      } else {
        if (live && !last) {
#pragma unroll
          for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) *reinterpret_cast<v4f*>(mybox + LDB + c * LDC + CHR * ch) = C[c][ch];
          }
          if (sub < 6) { mymeta[6] = nP; mymeta[104 + 6] = dP; }
        }
        lds_barrier();
        if (live) {
#pragma unroll
          for (int c = 0; c < 6; ++c) {
#pragma unroll
            for (int ch = 0; ch < MAXCH; ++ch) C[c][ch] = *reinterpret_cast<const v4f*>(mybox + c * LDC + CHR * ch);
          }
          nP = mymeta[0]; dP = mymeta[104];
        }
      }
    }
    used_sweeps = sweep + 1;
    if (rotated) s_flag[sweep & 1] = 1;
    if (bigrot) s_flag[2 + (sweep & 1)] = 1;
    lds_barrier();
    const int any = s_flag[sweep & 1], anybig = s_flag[2 + (sweep & 1)];
    if (tid == 0) { s_flag[(sweep + 1) & 1] = 0; s_flag[2 + ((sweep + 1) & 1)] = 0; s_flag[4 + ((sweep + 1) & 1)] = 0; }
    lds_barrier();
    if (!any || !anybig) { converged = true; break; }
  }

  fold(false, false);
  exact_norms(norm_rows);
  lds_barrier();
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int k2 = tid2 / LANES, sub2 = tid2 % LANES, roff2 = sub2 * 4;
  const bool live2 = k2 < S;
  const bool odd = (used_sweeps & 1) != 0;
  const int bP = live2 ? (odd ? nbe - 1 - 2 * k2 : 2 * k2) : 2 * k2;
  const int bQ = live2 ? (odd ? nbe - 2 - 2 * k2 : 2 * k2 + 1) : 2 * k2 + 1;
  if (k2 < S_all && sub2 < 6) {
    const int idP = 6 * bP + sub2, idQ = 6 * bQ + sub2;
    s_sig[12 * k2 + sub2] = idP >= n ? -2.f : ((live2 && idP < n_act) ? sqrtf(nP) : -1.f);
    s_sig[12 * k2 + 6 + sub2] = idQ >= n ? -2.f : ((live2 && idQ < n_act) ? sqrtf(nQ) : -1.f);
  }
  lds_barrier();
  const int n_tot = 12 * S_all;
  for (int p = tid2; p < n_tot; p += blockDim.x) {
    int rank = p;
    if (sort) {
      const float mine = s_sig[p];
      rank = 0;
      for (int c = 0; c < n_tot; ++c) {
        const float o = s_sig[c];
        rank += (o > mine) || (o == mine && c < p);
      }
    }
    s_rank[p] = rank;
  }
  lds_barrier();
  if (k2 < S_all) {
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      const int idc = 6 * (c < 6 ? bP : bQ) + (c % 6);
      const int d = sort ? s_rank[12 * k2 + c] : idc;
      const float sg = s_sig[12 * k2 + c];
      if (idc < n && d < n) {
#pragma unroll
        for (int ch = 0; ch < MAXCH; ++ch) {
          const int r = roff2 + CHR * ch;
          if (r < ld) *reinterpret_cast<v4f*>(src + (size_t)d * ld + r) = (live2 && r < mr) ? C[c][ch] : (v4f){0.f, 0.f, 0.f, 0.f};
        }
        if (sub2 == 0) sigma[(size_t)mat * n + d] = sg < 0.f ? 0.f : sg;
      }
    }
  }
  if (sweeps_out && tid2 == 0) sweeps_out[mat] = converged ? used_sweeps : -used_sweeps;
  report_status(status, converged, s_sig, n_tot, tid2, blockDim.x, 7, mat);
}

}  // namespace basd

extern "C" int basd_jacobi_svd(float* w, int batch, int m_rows, int n_cols, int ld, int norm_rows,
                               float tol, int max_sweeps, int sort, float* sigma,
                               int32_t* sweeps, const int32_t* active, int active_rows, int32_t* status,
                               void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n_cols < 1 || n_cols > BASD_JACOBI_MAX_COLS || ld % 4 != 0 || m_rows > ld || m_rows < 1 ||
      norm_rows < 1 || norm_rows > m_rows)
    return fail(BASD_ERR_SHAPE, "jacobi_svd: bad shape m=%d n=%d ld=%d norm_rows=%d", m_rows, n_cols, ld, norm_rows);
  const int npairs = (n_cols + 1) / 2;
  int threads = ((npairs * 8 + 63) / 64) * 64;
  if (threads < 64) threads = 64;
  const int chunks = (((m_rows + 3) & ~3) + 31) / 32;
  hipStream_t st = (hipStream_t)stream;
  // register-resident odd-even kernel: LDS only holds the double-buffered mailbox
  const int oe_ch = chunks <= 2 ? 2 : (chunks <= 4 ? 4 : (chunks <= 6 ? 6 : 7));
  const size_t oe_scratch = (260 + 260 + 2 * 2 * 132 * 2 + 8) * 4;
  // two matrices per workgroup (shared steps, ILP) once every CU would get at least two anyway
  const size_t lds_oe2 = (size_t)2 * npairs * 32 * oe_ch * 4 + oe_scratch;       // NMAT 2, single mailbox each
  const size_t lds_oe1 = lds_oe2;                                                // NMAT 1, double buffered
  const bool fits = lds_oe1 <= BASD_JACOBI_LDS_BYTES && chunks <= 7 && npairs <= 128;
#define BASD_LAUNCH_OE(MC, NM, GRID, LDSB)                                                            \
  do {                                                                                               \
    allow_full_lds((const void*)jacobi_oe_kernel<MC, NM, (NM == 1 ? 2 : 1)>);                         \
    hipLaunchKernelGGL((jacobi_oe_kernel<MC, NM, (NM == 1 ? 2 : 1)>), dim3(GRID), dim3(threads), (LDSB), st, w, batch, m_rows, \
                       n_cols, ld, norm_rows, tol, max_sweeps, sort, sigma, sweeps, active, active_rows, status); \
  } while (0)
  // active_rows == 2: `active` is a MASK -- entries are either < 0 (skip the matrix) or n_cols (solve it completely)
  const bool mask_only = active != nullptr && active_rows == 2;
  if (mask_only) active_rows = 0;
  {
    // quad-block ordering with scaled rotations: every launch of up to 192 x 192 (BASD_JACOBI_B4=0 falls back to the
    // kernels below: A/B timing).  BASD_JACOBI_B4_MIN sets the smallest batch it takes (default 1).
    const char* b4env = getenv("BASD_JACOBI_B4");
    const bool b4 = !(b4env && b4env[0] == '0');
    const char* b4min = getenv("BASD_JACOBI_B4_MIN");
    const int min_batch = b4min ? atoi(b4min) : 1;
    const int rows4 = (m_rows + 3) & ~3;
    const bool b4_16 = n_cols <= 196 && rows4 <= 256 && (rows4 <= 192 ? n_cols <= 192 : true);
    const bool b4_32 = n_cols <= 192 && rows4 > 256 && rows4 <= 384;
    const char* b6env = getenv("BASD_JACOBI_B6");
    // (from 129 columns: below that the quad-block kernel is four waves or fewer itself and its shorter rounds win --
    // 48 x 64^2: 0.136 vs 0.210 ms)
    const bool b6_on = !(b6env && b6env[0] == '0') && n_cols > 128 && n_cols <= 192;
    const bool b6_16 = b6_on && rows4 <= 192, b6_32 = b6_on && rows4 > 256 && rows4 <= 384;
    if (b4 && (b6_16 || b6_32) && batch >= min_batch) {
      // hex-block ordering: four waves for 192 columns (BASD_JACOBI_B6=0: the quad-block kernel, A/B timing)
      const int mode = active == nullptr ? 0 : (mask_only ? 3 : (active_rows ? 2 : 1));
      const int lanes = b6_32 ? 32 : 16;
      const int nb6 = (n_cols + 5) / 6, slots6 = (nb6 + 1) / 2;
      const int threads6 = ((slots6 * lanes + 63) / 64) * 64;
      const int chn = (rows4 + 4 * lanes - 1) / (4 * lanes);
      const size_t lds6 = ((size_t)(slots6 + 1) * 6 * 4 * lanes * chn + 200 * 2 + 104 * 2 + 8) * 4;
      if (lds6 > BASD_JACOBI_LDS_BYTES)
        return fail(BASD_ERR_SHAPE, "jacobi_svd: %d x %d needs %zu B of LDS", m_rows, n_cols, lds6);
#define BASD_LAUNCH_B6(MC, OC, LN)                                                                   \
  do {                                                                                               \
    allow_full_lds((const void*)jacobi_b6_kernel<MC, OC, LN>);                                       \
    hipLaunchKernelGGL((jacobi_b6_kernel<MC, OC, LN>), dim3(batch), dim3(threads6), lds6, st, w, batch, m_rows, n_cols, \
                       ld, norm_rows, tol, max_sweeps, sort, sigma, sweeps, status, active, mode);   \
  } while (0)
      const char* occenv = getenv("BASD_JACOBI_B6_OCC");
      const bool two = occenv ? occenv[0] == '2' : batch > 256;
      if (lanes == 32) BASD_LAUNCH_B6(3, 2, 32);
      else if (chn == 1) BASD_LAUNCH_B6(1, 2, 16);
      else if (chn == 2) BASD_LAUNCH_B6(2, 2, 16);
      else if (two) BASD_LAUNCH_B6(3, 2, 16);
      else BASD_LAUNCH_B6(3, 1, 16);
#undef BASD_LAUNCH_B6
      return check_launch("jacobi_svd (hex-block, scaled rotations)");
    }
    if (b4 && batch >= min_batch && n_cols >= 8 && (b4_16 || b4_32)) {
      const int mode = active == nullptr ? 0 : (mask_only ? 3 : (active_rows ? 2 : 1));
      const int lanes = b4_32 ? 32 : 16;
      const int nb4 = (n_cols + 3) / 4, slots4 = (nb4 + 1) / 2;
      const int threads4 = ((slots4 * lanes + 63) / 64) * 64;
      const int chn = (rows4 + 4 * lanes - 1) / (4 * lanes);
      const size_t lds4 = ((size_t)(slots4 + 1) * 4 * 4 * lanes * chn + 200 * 2 + 104 * 2 + 8) * 4;
      if (lds4 > BASD_JACOBI_LDS_BYTES)
        return fail(BASD_ERR_SHAPE, "jacobi_svd: %d x %d needs %zu B of LDS", m_rows, n_cols, lds4);
#define BASD_LAUNCH_B4(MC, LN)                                                                       \
  do {                                                                                               \
    allow_full_lds((const void*)jacobi_b4_kernel<MC, LN>);                                           \
    hipLaunchKernelGGL((jacobi_b4_kernel<MC, LN>), dim3(batch), dim3(threads4), lds4, st, w, batch, m_rows, n_cols, \
                       ld, norm_rows, tol, max_sweeps, sort, sigma, sweeps, status, active, mode);   \
  } while (0)
      if (lanes == 32) BASD_LAUNCH_B4(3, 32);
      else if (chn == 1) BASD_LAUNCH_B4(1, 16);
      else if (chn == 2) BASD_LAUNCH_B4(2, 16);
      else if (chn == 3) BASD_LAUNCH_B4(3, 16);
      else BASD_LAUNCH_B4(4, 16);
#undef BASD_LAUNCH_B4
      return check_launch("jacobi_svd (quad-block, scaled rotations)");
    }
  }
  if (chunks > 7 && chunks <= 12 && npairs <= 96) {
    // tall columns (up to 384 rows, at most 192 of them): single mailbox, 96 slots, 3 waves per SIMD
    const size_t lds_tall = (size_t)npairs * 32 * 12 * 4 + oe_scratch;
    allow_full_lds((const void*)jacobi_oe_kernel<12, 1, 1>);
    hipLaunchKernelGGL((jacobi_oe_kernel<12, 1, 1>), dim3(batch), dim3(threads), lds_tall, st, w, batch, m_rows, n_cols, ld,
                       norm_rows, tol, max_sweeps, sort, sigma, sweeps, active, active_rows, status);
    return check_launch("jacobi_svd (odd-even, tall columns)");
  }
  // (the block ordering also wins on small batches: 1.06 vs 1.27 ms at 48 matrices of 192^2, 1.27 vs 1.39 at 256,
  // equal at 4 - 24: measured in round 3; before, it was only taken from 512 matrices up)
  // MAXCH = 7 (193 .. 224 rows, up to 196 columns: the 196-token Procrustes cores of the wide students): 112 column
  // VGPRs (202 in all, no spills), seven waves, one workgroup per CU: 4.02 vs 5.00 ms per 512 matrices of 196 x 196
  // with the single-mailbox odd-even kernel
  const bool blk7 = oe_ch == 7 && n_cols <= 196;
  if ((active == nullptr || mask_only) && batch >= 32 && ((n_cols <= 192 && oe_ch <= 6) || blk7) && n_cols >= 8) {
    // block ordering: one matrix per workgroup, slots = ceil(ceil(n / 2) / 2)
    const int nbk = (n_cols + 1) / 2, slots = (nbk + 1) / 2;
    const int threads_b = ((slots * 8 + 63) / 64) * 64;
    const size_t lds_b = ((size_t)(slots + 1) * 2 * 32 * oe_ch + 264 * 4 + 8) * 4;
#define BASD_LAUNCH_BLK(MC)                                                                          \
  do {                                                                                               \
    allow_full_lds((const void*)jacobi_blk_kernel<MC>);                                              \
    hipLaunchKernelGGL((jacobi_blk_kernel<MC>), dim3(batch), dim3(threads_b), lds_b, st, w, batch, m_rows, n_cols, \
                       ld, norm_rows, tol, max_sweeps, sort, sigma, sweeps, status, active);         \
  } while (0)
    if (oe_ch == 2) BASD_LAUNCH_BLK(2);
    else if (oe_ch == 4) BASD_LAUNCH_BLK(4);
    else if (oe_ch == 6) BASD_LAUNCH_BLK(6);
    else BASD_LAUNCH_BLK(7);
#undef BASD_LAUNCH_BLK
    return check_launch("jacobi_svd (block odd-even)");
  }
  if (fits && batch >= 512 && npairs <= 96 && oe_ch <= 6) {
    const int grid2 = (batch + 1) / 2;
    if (oe_ch == 2) BASD_LAUNCH_OE(2, 2, grid2, lds_oe2);
    else if (oe_ch == 4) BASD_LAUNCH_OE(4, 2, grid2, lds_oe2);
    else BASD_LAUNCH_OE(6, 2, grid2, lds_oe2);
    return check_launch("jacobi_svd (odd-even x2)");
  }
  const size_t lds_single = (size_t)npairs * 32 * oe_ch * 4 + oe_scratch;            // NMAT 1, ONE mailbox
  if (!fits && chunks <= 7 && npairs <= 128 && lds_single <= BASD_JACOBI_LDS_BYTES) {
    // 193 .. 224 rows x up to 256 columns (the 196 x 196 token-side Procrustes cores of the wide students): the
    // double-buffered mailbox does not fit, a single one does; still register-resident (the LDS-resident kernel below
    // takes 7.7 ms for 512 such matrices)
    allow_full_lds((const void*)jacobi_oe_kernel<7, 1, 1>);
    hipLaunchKernelGGL((jacobi_oe_kernel<7, 1, 1>), dim3(batch), dim3(threads), lds_single, st, w, batch, m_rows, n_cols, ld,
                       norm_rows, tol, max_sweeps, sort, sigma, sweeps, active, active_rows, status);
    return check_launch("jacobi_svd (odd-even, single mailbox)");
  }
  if (fits) {
    if (oe_ch == 2) BASD_LAUNCH_OE(2, 1, batch, lds_oe1);
    else if (oe_ch == 4) BASD_LAUNCH_OE(4, 1, batch, lds_oe1);
    else if (oe_ch == 6) BASD_LAUNCH_OE(6, 1, batch, lds_oe1);
    else BASD_LAUNCH_OE(7, 1, batch, lds_oe1);
    return check_launch("jacobi_svd (odd-even)");
  }
#undef BASD_LAUNCH_OE
  const size_t lds_bytes = (size_t)n_cols * ld * 4 + (256 + 256 + 8) * 4;
  if (lds_bytes > BASD_JACOBI_LDS_BYTES)
    return fail(BASD_ERR_SHAPE, "jacobi_svd: %d x %d (ld %d) needs %zu B of LDS > 160 KiB", m_rows, n_cols, ld, lds_bytes);
#define BASD_LAUNCH_JACOBI(MC)                                                                     \
  do {                                                                                             \
    allow_full_lds((const void*)jacobi_kernel<MC>);                                                  \
    hipLaunchKernelGGL(jacobi_kernel<MC>, dim3(batch), dim3(threads), lds_bytes, st, w, m_rows,    \
                       n_cols, ld, norm_rows, tol, max_sweeps, sort, sigma, sweeps, active,       \
                       active_rows, status);                                                      \
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
