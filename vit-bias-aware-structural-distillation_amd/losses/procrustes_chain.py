"""The forward of the Procrustes term launched entry by entry (pivoted Cholesky, Jacobi, triangular inverse, fp64
batched GEMMs): the SAME kernels, in the same order, that ``basd_procrustes_fwd`` chains inside the library.  The
training path calls the composite entry; this form exists so that ``bench.py`` can put device events around the
individual launches in its instrumented eager steps (nothing can be timed inside one C call), and as the reference the
tests compare the composite entry with (``ops`` = the provider whose kernels are used)."""
import torch


def _polar_core_gram(ops, mx: torch.Tensor, tol: float):
    """mx = X X^T [B, r, r] fp64 (lower triangle meaningful) of some X [B, r, c] with r <= 196 ->
    (sigma [B, r] fp32 = singular values of X, m [B, r, r] fp64) with polar(X) = U V^T = m @ X.

    The pivoted Cholesky factor L of the Gram (X = L Q2, Q2 = L^-1 X with orthonormal rows) is diagonalised by the
    fp32 Jacobi: L J1 = U Sigma.  The right factor J1 = L^-1 (U Sigma) uses the explicit fp64 inverse of the graded L
    (never a division by sigma), so (U, J1) is a consistent pair and U V^T = U J1^T Q2 = (U J1^T L^-1) X is
    orthonormal to working precision.
    """
    r = mx.shape[1]
    w0, lwork, piv, rank = ops.pchol(mx, tol)
    sigma, _ = ops.jacobi_svd(w0, r)                                    # w0[:, i, :r] = sigma_i u_i
    l_inv = ops.trinv(lwork, piv, rank)                                 # L_p^-1 P  [B, k, r] fp64
    wf = w0[:, :, :r].contiguous()                                      # [B, i, r] = sigma_i u_i[r]
    j1 = ops.bgemm_f64(l_inv, wf, trans_b=True)                         # [B, k, i]:  L J1 = U Sigma
    u = torch.where(sigma.unsqueeze(-1) > 0, wf / sigma.clamp_min(1e-30).unsqueeze(-1),
                    torch.zeros(1, device=mx.device))                   # [B, i, r]
    theta = ops.bgemm_f64(u, j1, trans_a=True, trans_b=True)            # [B, r, k] = polar(L), fp64
    # polar(X) = theta Q2 with Q2 = L^-1 X: associate as (theta L^-1) X so that the [k, c] factor is never rounded
    # to fp32 (all on the fp64 MFMA)
    return sigma, ops.bgemm_f64(theta, l_inv)                           # [B, r, r'] fp64


def _polar_feature_side(ops, s_w: torch.Tensor, t_w: torch.Tensor, tol: float):
    """Feature-side form (D_s <= 192 <= N_s - 1, e.g. BASELINE c2): nuclear norm [B] of cross = s_w^T t_w and the two
    factors of the backward,  t_w G^T = p_s [B, N, D_s]  and  s_w G = a_t t_w  (a_t [B, N, N]),  G = U V^T = m cross.

    cross [D_s, D_t] itself is never formed: with G_t = t_w t_w^T (N x N, ONE pass over t_w, fp64)
        cross cross^T = s_w^T G_t s_w,   t_w G^T = G_t (m s_w^T)^T,   s_w G = (s_w m s_w^T) t_w,
    all N x N / N x D_s products.  (Round 1 materialised cross in fp64 -- 1.2 GB at c2 -- and streamed it three
    times, then wrote G in fp32: 8 GB of algorithmic traffic, 18.5 GB counted; this form moves the 0.6 GB of t_w
    once here and once in the backward.)"""
    gt = ops.bgemm_f64(t_w, t_w, trans_b=True, symmetric=True)          # [B, N, N] fp64
    h = ops.bgemm_f64(gt, s_w)                                          # [B, N, D_s] fp64
    # (a symmetric result of two different operands: lower tiles only, mirrored)
    sigma, m = _polar_core_gram(ops, ops.bgemm_f64(s_w, h, trans_a=True, symmetric=True), tol)    # Gram of cross, [B, D_s, D_s]
    p = ops.bgemm_f64(m, s_w, trans_b=True)                             # m s_w^T  [B, D_s, N] fp64
    p_s = ops.bgemm_f64(gt, p, trans_b=True, out_dtype=torch.float32)   # G_t P^T = t_w G^T  [B, N, D_s]
    a_t = ops.bgemm_f64(s_w, p, out_dtype=torch.float32)                # s_w P  [B, N, N]
    return sigma.sum(dim=-1), p_s, a_t


def _polar_token_side(ops, s_w: torch.Tensor, t_w: torch.Tensor, tol: float):
    """Token-side form for N_s <= D_s (every wide student: rank(cross) <= N_s - 1 = 195 for all BASELINE configs,
    SURVEY section 7): nuclear norm [B] and the two N x N matrices of the backward,
        t_w G^T = A_s s_w,   s_w G = A_t t_w      (G = U V^T of cross = s_w^T t_w, never formed).

    With the pivoted Cholesky factors of the token Gram matrices, s_w s_w^T = R_s^T R_s and t_w t_w^T = R_t^T R_t
    (R = the factor's columns, rows in original token order), s_w^T = Q_s R_s and t_w^T = Q_t R_t with orthonormal
    Q = (W s_w)^T, W = L^-1 P.  Hence cross = Q_s (R_s R_t^T) Q_t^T: its singular values are those of the
    rank x rank core C = R_s R_t^T, G = Q_s polar(C) Q_t^T, and
        A_s = R_t^T polar(C)^T W_s,   A_t = R_s^T polar(C) W_t.
    """
    gs = ops.bgemm_f64(s_w, s_w, trans_b=True, symmetric=True)          # [B, N, N] fp64
    gt = ops.bgemm_f64(t_w, t_w, trans_b=True, symmetric=True)
    _, r_s, piv_s, rk_s = ops.pchol(gs, tol)                       # r_s[b, k, n] = R_s (zero rows beyond the rank)
    _, r_t, piv_t, rk_t = ops.pchol(gt, tol)
    w_s = ops.trinv(r_s, piv_s, rk_s)                                    # [B, k, n]
    w_t = ops.trinv(r_t, piv_t, rk_t)
    core = ops.bgemm_f64(r_s, r_t, trans_b=True)                         # C [B, k_s, k_t] fp64
    sigma, m = _polar_core_gram(ops, ops.bgemm_f64(core, core, trans_b=True, symmetric=True), tol)
    pc = ops.bgemm_f64(m, core)                                          # polar(C) [B, k_s, k_t] fp64
    a_s = ops.bgemm_f64(ops.bgemm_f64(pc, r_t), w_s, trans_a=True, out_dtype=torch.float32)    # (pc R_t)^T W_s
    a_t = ops.bgemm_f64(ops.bgemm_f64(r_s, pc, trans_a=True), w_t, out_dtype=torch.float32)    # (R_s^T pc) W_t
    return sigma.sum(dim=-1), a_s, a_t




def procrustes_fwd_chain(ops, s_w: torch.Tensor, t_w: torch.Tensor, tol: float = 1e-13):
    """(nuc [B], fac_s, a_t): token side for N <= D_s, feature side otherwise -- the dispatch of basd_procrustes_fwd"""
    if s_w.shape[1] <= s_w.shape[2]:
        return _polar_token_side(ops, s_w, t_w, tol)
    return _polar_feature_side(ops, s_w, t_w, tol)
