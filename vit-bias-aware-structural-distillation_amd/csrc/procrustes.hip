// Forward of the attention-weighted Procrustes term as ONE C entry: everything between basd_procrustes_prep and the
// loss value (reference src/losses/relational.py:47-48: cross-covariance, torch.linalg.matrix_norm(ord="nuc"), and
// what svd_backward needs, U V^T) as a chain of the library's own launches on a caller-provided workspace -- no
// allocation, no host synchronisation, no torch glue between the launches.  DESIGN.md sections 4 and 4c derive the
// two forms; in short, with s_w [n, d_s], t_w [n, d_t] (weighted, centred tokens) and cross = s_w^T t_w (never formed):
//   feature side (n > d_s):  Gt = t_w t_w^T,  cross cross^T = s_w^T Gt s_w,  polar core m of that Gram matrix,
//                            fac_s = t_w G^T = Gt (m s_w^T)^T  [n, d_s],  a_t = s_w m s_w^T  [n, n]  (s_w G = a_t t_w)
//   token side  (n <= d_s):  Gs = R_s^T R_s, Gt = R_t^T R_t (pivoted Cholesky), core C = R_s R_t^T, polar(C) = m C,
//                            fac_s = a_s = (polar(C) R_t)^T W_s,  a_t = (R_s^T polar(C)) W_t   (W = L^-1 P), both [n, n]
// and nuc = sum of the singular values of the core.  polar core (shared): Gram -> pivoted Cholesky L (fp64) -> fp32
// Jacobi L J = U Sigma -> J = L^-1 (U Sigma) (explicit fp64 inverse) -> m = U J^T L^-1.
#include "basd_common.h"
#include "../../include/basd_hip.h"

namespace basd {

// u[b, i, :] = (sigma_i > 0) ? w0[b, i, :r] / sigma_i : 0   (w0 rows are sigma_i u_i, stride ld);  nuc[b] += sigma_i
__global__ __launch_bounds__(256) void polar_unit_rows_kernel(const float* __restrict__ w0, int ld, int r,
                                                              const float* __restrict__ sigma, float* __restrict__ u,
                                                              float* __restrict__ nuc) {
  const int b = blockIdx.x;
  const float* sg = sigma + (size_t)b * r;
  if ((r & 3) == 0) {                                 // 16 bytes per lane (ld is a multiple of 4)
    const int r4 = r >> 2;
    for (int idx = threadIdx.x; idx < r * r4; idx += 256) {
      const int i = idx / r4, c4 = idx - i * r4;
      const float s = sg[i];
      float4 v = *reinterpret_cast<const float4*>(w0 + ((size_t)b * r + i) * ld + 4 * c4);
      v.x = s > 0.f ? v.x / fmaxf(s, 1e-30f) : 0.f;
      v.y = s > 0.f ? v.y / fmaxf(s, 1e-30f) : 0.f;
      v.z = s > 0.f ? v.z / fmaxf(s, 1e-30f) : 0.f;
      v.w = s > 0.f ? v.w / fmaxf(s, 1e-30f) : 0.f;
      *reinterpret_cast<float4*>(u + ((size_t)b * r + i) * r + 4 * c4) = v;
    }
  } else {
    for (int idx = threadIdx.x; idx < r * r; idx += 256) {
      const int i = idx / r, c = idx - i * r;
      const float s = sg[i];
      u[((size_t)b * r + i) * r + c] = s > 0.f ? w0[((size_t)b * r + i) * ld + c] / fmaxf(s, 1e-30f) : 0.f;
    }
  }
  if (nuc != nullptr) {
    __shared__ float red[4];
    float t = 0.f;
    for (int i = threadIdx.x; i < r; i += 256) t += sg[i];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) nuc[b] = red[0] + red[1] + red[2] + red[3];
  }
}

static inline int jacobi_ld_of(int m_rows) {
  int ld = (m_rows + 3) / 4 * 4;
  if (ld % 32 == 0) ld += 4;
  return ld;
}

struct Bump {                       // carve 256-byte aligned pieces out of the workspace; also used to size it
  char* base;
  int64_t used = 0;
  template <typename T> T* take(int64_t count) {
    T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
    used += (count * (int64_t)sizeof(T) + 255) / 256 * 256;
    return p;
  }
};

struct ProcrustesPlan {
  bool token_side;
  int r;                            // order of the polar core
  double* slot[10];                 // fp64 [batch, n_max, n_max] scratch matrices (5 feature side, 10 token side)
  float *w0, *u, *sigma;
  int32_t *piv, *rank, *piv2, *rank2;
};

static int64_t plan(ProcrustesPlan& p, char* base, int batch, int n, int d_s) {
  Bump w{base};
  p.token_side = n <= d_s;
  p.r = p.token_side ? n : d_s;
  const int64_t big = (int64_t)batch * (n > d_s ? n : d_s) * n;       // holds [n, n], [n, d_s <= n] and [r, r]
  for (int i = 0; i < 10; ++i) p.slot[i] = (i < (p.token_side ? 10 : 5)) ? w.take<double>(big) : nullptr;
  p.w0 = w.take<float>((int64_t)batch * p.r * jacobi_ld_of(p.r));
  p.u = w.take<float>((int64_t)batch * p.r * p.r);
  p.sigma = w.take<float>((int64_t)batch * p.r);
  p.piv = w.take<int32_t>((int64_t)batch * n);
  p.rank = w.take<int32_t>(batch);
  p.piv2 = w.take<int32_t>((int64_t)batch * n);
  p.rank2 = w.take<int32_t>(batch);
  return w.used;
}

#define BASD_TRY(call)            \
  do {                            \
    const int rc_ = (call);       \
    if (rc_ != BASD_OK) return rc_; \
  } while (0)

// C = op(A) op(B), all [batch] matrices contiguous with the given leading dimensions
static int mm(const void* a, int adt, int a_rows, int lda, int ta, const void* b, int bdt, int b_rows, int ldb, int tb,
              void* c, int cdt, int ldc, int batch, int M, int N, int K, int sym, void* st) {
  return basd_bgemm_f64(a, adt, (int64_t)a_rows * lda, lda, ta, b, bdt, (int64_t)b_rows * ldb, ldb, tb, c, cdt,
                        (int64_t)M * ldc, ldc, batch, M, N, K, sym, st);
}

// gram [batch, r, r] fp64 (lower triangle) -> sigma, nuc, m [batch, r, r] fp64 (polar(X) = m X for gram = X X^T).
// lwork / l_inv / j1 / theta are fp64 [batch, r, r] scratch; gram itself is dead after the factorisation.
static int polar_core(const double* gram, int batch, int r, double tol, ProcrustesPlan& p, double* lwork, double* l_inv,
                      double* j1, double* theta, double* m, float* nuc, int32_t* status, void* st) {
  const int ld = jacobi_ld_of(r);
  const int F32 = BASD_DTYPE_F32, F64 = BASD_DTYPE_F64;
  BASD_TRY(basd_pchol_f64(gram, batch, r, tol, nullptr, p.w0, ld, lwork, p.piv, p.rank, st));
  BASD_TRY(basd_jacobi_svd(p.w0, batch, r, r, ld, r, sqrtf((float)r) * 5.96e-8f, 60, 1, p.sigma, nullptr, nullptr, 0,
                           status, st));
  BASD_TRY(basd_trinv_f64(lwork, p.piv, p.rank, batch, r, l_inv, st));
  // J1 = L^-1 (U Sigma): l_inv [k, r] x (w0 [i, r])^T -> [k, i]; w0 is read in place through its leading dimension
  BASD_TRY(mm(l_inv, F64, r, r, 0, p.w0, F32, r, ld, 1, j1, F64, r, batch, r, r, r, 0, st));
  hipLaunchKernelGGL(polar_unit_rows_kernel, dim3(batch), dim3(256), 0, (hipStream_t)st, p.w0, ld, r, p.sigma, p.u, nuc);
  // theta = U J1^T: (u [i, r])^T x (j1 [k, i])^T -> [r, k]
  BASD_TRY(mm(p.u, F32, r, r, 1, j1, F64, r, r, 1, theta, F64, r, batch, r, r, r, 0, st));
  // m = theta L^-1 (the [k, c] factor Q2 = L^-1 X is never rounded to fp32)
  BASD_TRY(mm(theta, F64, r, r, 0, l_inv, F64, r, r, 0, m, F64, r, batch, r, r, r, 0, st));
  return check_launch("procrustes_fwd");
}

}  // namespace basd

extern "C" int64_t basd_procrustes_workspace_bytes(int batch, int n, int d_s, int d_t) {
  using namespace basd;
  (void)d_t;
  if (batch <= 0 || n <= 0 || d_s <= 0) return 0;
  ProcrustesPlan p;
  return plan(p, nullptr, batch, n, d_s);
}

extern "C" int basd_procrustes_fwd(const float* s_w, const float* t_w, int batch, int n, int d_s, int d_t, double tol,
                                   float* nuc, float* fac_s, float* a_t, int32_t* status, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n <= 0 || d_s <= 0 || d_t <= 0 || (n < d_s ? n : d_s) > BASD_JACOBI_MAX_COLS)
    return fail(BASD_ERR_SHAPE, "procrustes_fwd: n=%d d_s=%d d_t=%d (the core min(n, d_s) must be <= %d)", n, d_s, d_t,
                BASD_JACOBI_MAX_COLS);
  ProcrustesPlan p;
  const int64_t need = plan(p, nullptr, batch, n, d_s);
  if (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 255))
    return fail(BASD_ERR_WORKSPACE, "procrustes_fwd: workspace of %lld bytes (256-byte aligned) required, got %lld",
                (long long)need, (long long)workspace_bytes);
  plan(p, (char*)workspace, batch, n, d_s);
  const int F32 = BASD_DTYPE_F32, F64 = BASD_DTYPE_F64;
  void* st = stream;
  if (!p.token_side) {
    double *gt = p.slot[0], *h = p.slot[1], *gram = p.slot[2], *lwork = p.slot[3], *l_inv = p.slot[4];
    BASD_TRY(mm(t_w, F32, n, d_t, 0, t_w, F32, n, d_t, 1, gt, F64, n, batch, n, n, d_t, 1, st));        // Gt = t_w t_w^T
    BASD_TRY(mm(gt, F64, n, n, 0, s_w, F32, n, d_s, 0, h, F64, d_s, batch, n, d_s, n, 0, st));          // h = Gt s_w
    // (a symmetric result of two different operands: lower tiles only, 6 of 9; the factorisation reads the lower triangle)
    BASD_TRY(mm(s_w, F32, n, d_s, 1, h, F64, n, d_s, 0, gram, F64, d_s, batch, d_s, d_s, n, 1, st));    // s_w^T Gt s_w
    double *j1 = h, *theta = gram, *m = lwork;       // h dead after gram; gram dead after pchol; lwork dead after trinv
    // (theta may not alias j1 / l_inv, m may not alias theta / l_inv: j1 = slot 1, theta = 2, l_inv = 4, m = 3)
    BASD_TRY(polar_core(gram, batch, d_s, tol, p, lwork, l_inv, j1, theta, m, nuc, status, st));
    double* pm = p.slot[2];                          // P = m s_w^T  [d_s, n]  (theta is dead)
    BASD_TRY(mm(m, F64, d_s, d_s, 0, s_w, F32, n, d_s, 1, pm, F64, n, batch, d_s, n, d_s, 0, st));
    BASD_TRY(mm(gt, F64, n, n, 0, pm, F64, d_s, n, 1, fac_s, F32, d_s, batch, n, d_s, n, 0, st));       // Gt P^T = t_w G^T
    // (a_t = s_w m s_w^T is symmetric in exact arithmetic; computing its lower tiles only and mirroring them was
    // measured: -0.19 ms per 1024 matrices, but the polar factor a_t t_w loses accuracy -- 2.06e-5 against < 2e-5
    // with all tiles, m carries the 1e-5 asymmetry of the fp32 Jacobi -- so every tile is computed)
    BASD_TRY(mm(s_w, F32, n, d_s, 0, pm, F64, d_s, n, 0, a_t, F32, n, batch, n, n, d_s, 0, st));        // s_w P
    return check_launch("procrustes_fwd");
  }
  // ---- token side
  double *gs = p.slot[0], *gt = p.slot[1], *r_s = p.slot[2], *r_t = p.slot[3], *w_s = p.slot[4], *w_t = p.slot[5];
  const int ld = jacobi_ld_of(n);
  BASD_TRY(mm(s_w, F32, n, d_s, 0, s_w, F32, n, d_s, 1, gs, F64, n, batch, n, n, d_s, 1, st));
  BASD_TRY(mm(t_w, F32, n, d_t, 0, t_w, F32, n, d_t, 1, gt, F64, n, batch, n, n, d_t, 1, st));
  // r_x[b, k, :] = row k of R_x (zero rows beyond the rank); the fp32 copies the factorisation also writes are unused
  BASD_TRY(basd_pchol_f64(gs, batch, n, tol, nullptr, p.w0, ld, r_s, p.piv, p.rank, st));
  BASD_TRY(basd_pchol_f64(gt, batch, n, tol, nullptr, p.w0, ld, r_t, p.piv2, p.rank2, st));
  BASD_TRY(basd_trinv_f64(r_s, p.piv, p.rank, batch, n, w_s, st));
  BASD_TRY(basd_trinv_f64(r_t, p.piv2, p.rank2, batch, n, w_t, st));
  double *core = gs, *gram = gt;                     // the token Gram matrices are dead now
  BASD_TRY(mm(r_s, F64, n, n, 0, r_t, F64, n, n, 1, core, F64, n, batch, n, n, n, 0, st));              // C = R_s R_t^T
  BASD_TRY(mm(core, F64, n, n, 0, core, F64, n, n, 1, gram, F64, n, batch, n, n, n, 1, st));            // C C^T
  double *lwork = p.slot[6], *l_inv = p.slot[7], *j1 = p.slot[8], *m = p.slot[9], *theta = gram;
  BASD_TRY(polar_core(gram, batch, n, tol, p, lwork, l_inv, j1, theta, m, nuc, status, st));
  double *pc = lwork, *x1 = j1, *x2 = l_inv;         // dead scratch of the core
  BASD_TRY(mm(m, F64, n, n, 0, core, F64, n, n, 0, pc, F64, n, batch, n, n, n, 0, st));                 // polar(C) = m C
  BASD_TRY(mm(pc, F64, n, n, 0, r_t, F64, n, n, 0, x1, F64, n, batch, n, n, n, 0, st));                 // polar(C) R_t
  BASD_TRY(mm(x1, F64, n, n, 1, w_s, F64, n, n, 0, fac_s, F32, n, batch, n, n, n, 0, st));              // a_s = x1^T W_s
  BASD_TRY(mm(r_s, F64, n, n, 1, pc, F64, n, n, 0, x2, F64, n, batch, n, n, n, 0, st));                 // R_s^T polar(C)
  BASD_TRY(mm(x2, F64, n, n, 0, w_t, F64, n, n, 0, a_t, F32, n, batch, n, n, n, 0, st));                // a_t = x2 W_t
  return check_launch("procrustes_fwd");
}
