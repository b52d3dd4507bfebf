"""Autograd functions of the BASD loss path on top of the HIP kernels.

Three differentiable pieces, each a ``torch.autograd.Function`` whose forward
and backward call the C-ABI kernels (through ``_ops.get_ops()``) plus small
dense ``torch.bmm`` / triangular-solve calls (plain library GEMMs):

``selector_weights``   per-layer Gram statistics -> MP ranks (on device, no
                       ``.item()``) -> PCA frames -> masked principal angles
                       -> softmax mixing weights [E, L]
                       (reference src/losses/layer_selector.py:69-108, 133-138)
``mix_layers``         all E weighted mixes of the teacher layers in one pass
                       (layer_selector.py:110-112)
``procrustes``         attention-weighted Procrustes value per sample
                       (src/losses/relational.py:29-50)

Numerical design: every Gram matrix is accumulated in
fp64, factored by a pivoted fp64 Cholesky (a column-graded factor) and the
factor is diagonalised by an fp32 one-sided Jacobi in LDS; singular-vector
pairs that have to be consistent (principal angles, polar factor) are never
recovered through a division by a small singular value of an un-graded matrix.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ._ops import get_ops, note_library_gemm

_EPS32 = float(torch.finfo(torch.float32).eps)
PCHOL_TOL = 1e-13


class BasdShapeError(NotImplementedError):
    pass


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
# BASD_PROCRUSTES_BWD=lib: the Procrustes backward through the library fp32 bmm + basd_procrustes_bwd_rows (rounds 1 - 3)
# instead of the one-call basd_procrustes_bwd (A/B timing)
PROCRUSTES_BWD_FUSED = __import__("os").environ.get("BASD_PROCRUSTES_BWD", "fused") != "lib"
WIDE_PANEL = 192          # widest eigenproblem of the LDS-resident Jacobi = panel of the blocked Cholesky
WIDE_BLOCK = 96           # column block of the blocked Jacobi: a PAIR of blocks is one LDS-resident problem
WIDE_COS_TOL = 1e-4      # columns of the blocked factor must end up orthogonal to this (else: NONCONVERGED flag)
WIDE_DIRECT_SWEEPS = 8    # outer sweeps when a block pair fits the tall-column Jacobi kernel (n <= 384)
_ROUND_CACHE: dict = {}


def _wide_sweeps(nblk: int) -> int:
    """Outer sweeps of the blocked Jacobi: fixed per width (no host sync).  Measured on graded random-basis spectra
    (condition 1e3 .. 4e3, emulated fp32-Jacobi rounding): column cosines reach their floor after 3 sweeps at 4 blocks
    (D_s = 384) and after 6 at 8 blocks (D_s = 768); one sweep of margin."""
    return 3 + (nblk + 1) // 2 if nblk > 4 else 4


def _tournament(nblk: int, device):
    """Round-robin schedule over ``nblk`` (even) column blocks: nblk - 1 rounds of nblk / 2 disjoint pairs, as
    device index tensors [nblk] (pair p = entries 2p, 2p+1).  Cached: building them during a stream capture would be
    an illegal host-to-device copy."""
    key = (nblk, str(device))
    hit = _ROUND_CACHE.get(key)
    if hit is None:
        ring = list(range(nblk))
        rounds = []
        for _ in range(nblk - 1):
            pairs = []
            for k in range(nblk // 2):
                a, b = ring[k], ring[nblk - 1 - k]
                pairs += [min(a, b), max(a, b)]
            rounds.append(torch.tensor(pairs, dtype=torch.long, device=device))
            ring = [ring[0]] + [ring[-1]] + ring[1:-1]
        hit = _ROUND_CACHE[key] = rounds
    return hit


def _tournament_perms(nblk: int, device):
    """The same schedule for blocks kept in PAIR order: perms[r] takes the order of round r - 1 (natural order before
    round 0; the last round's order when a sweep wraps around) to round r's, ``restore`` takes the last round's order
    back to the natural one.  Cached device tensors."""
    key = ("perm", nblk, str(device))
    hit = _ROUND_CACHE.get(key)
    if hit is None:
        rounds = [r.tolist() for r in _tournament(nblk, device)]
        def inverse(order):
            inv = [0] * nblk
            for pos, blk in enumerate(order):
                inv[blk] = pos
            return inv
        first = torch.tensor(rounds[0], dtype=torch.long, device=device)                       # natural -> round 0
        steps = [torch.tensor([inverse(rounds[r - 1])[blk] for blk in rounds[r]], dtype=torch.long, device=device)
                 for r in range(1, len(rounds))]
        wrap = torch.tensor([inverse(rounds[-1])[blk] for blk in rounds[0]], dtype=torch.long, device=device)
        restore = torch.tensor(inverse(rounds[-1]), dtype=torch.long, device=device)
        # pair p of round r holds the blocks rounds[r][2p], rounds[r][2p+1]; "siblings" = the blocks {2m, 2m+1}
        sib = [torch.tensor([(r[2 * p] ^ 1) == r[2 * p + 1] for p in range(nblk // 2)], dtype=torch.bool, device=device)
               for r in rounds]
        hit = _ROUND_CACHE[key] = (first, steps, wrap, restore, sib)
    first, steps, wrap, restore, sib = hit

    class _Perms:
        """iterating yields (the permutation to apply before the round, [npair] bool: the pair's blocks are siblings
        {2m, 2m + 1}); the first sweep starts from the natural order, later sweeps from the last round's order"""
        def __init__(self):
            self.started = False

        def __iter__(self):
            yield (wrap if self.started else first), sib[0]
            self.started = True
            yield from zip(steps, sib[1:])

    return _Perms(), restore


def _blocked_pchol(a64: torch.Tensor, n_pad: int) -> torch.Tensor:
    """Blocked Cholesky of PSD fp64 matrices wider than one panel: A = X X^T with X returned in the Jacobi layout
    [b, column (= elimination step), row] fp32.  Panels of WIDE_PANEL columns; inside a panel the register-resident
    diagonal-pivoted kernel (basd_pchol_f64) factors the current Schur complement block -- pivoting is local to the
    panel, the stop test is relative to the largest diagonal entry of the WHOLE matrix -- the rows below follow from
    the explicit inverse of the panel factor (basd_trinv_f64) and the trailing update is one fp64 MFMA GEMM
    (basd_bgemm_f64).  Only lower blocks of A are read."""
    ops = get_ops()
    b, n, _ = a64.shape
    a = a64.new_zeros(b, n_pad, n_pad)
    a[:, :n, :n] = a64
    dref = a64.diagonal(dim1=-2, dim2=-1).amax(dim=-1).contiguous()
    x = torch.zeros(b, n_pad, n_pad, dtype=torch.float32, device=a64.device)
    for p0 in range(0, n_pad, WIDE_PANEL):
        p1 = p0 + WIDE_PANEL
        _, lw, piv, rank = ops.pchol(a[:, p0:p1, p0:p1].contiguous(), PCHOL_TOL, dmax_ref=dref)
        x[:, p0:p1, p0:p1] = lw                                   # lw[b, step, row]: rows in original order
        if p1 < n_pad:
            l_inv = ops.trinv(lw, piv, rank)                      # [b, step, col] = L^-1 P (zero rows beyond the rank)
            lb = ops.bgemm_f64(l_inv, a[:, p1:, p0:p1].contiguous(), trans_b=True)    # [b, step, rows below]
            x[:, p0:p1, p1:] = lb
            a[:, p1:, p1:] -= ops.bgemm_f64(lb, lb, trans_a=True)                     # Schur complement
    return x


def _tall_tournament(ops, xv: torch.Tensor, done: torch.Tensor) -> torch.Tensor:
    """Block-Jacobi tournament with DIRECT visits: xv [b, nblk, 96, rows] fp32 (column blocks of X, rows <= 384) -> the
    same array with mutually orthogonal columns; matrices with ``done[b]`` set are left alone.

    A block pair (192 columns x up to 384 rows) is register-resident in the tall-column Jacobi kernel, so a visit is
    ONE launch that rotates the actual columns (one inner sweep, no Gram, no pair rotation matrix): the graded accuracy
    is the kernel's own.  Simulated on graded random-basis spectra: cosines < 1e-7 after 7 outer sweeps of single inner
    sweeps (21 visits at 4 blocks; the Gram form needs 5 x 3 visits of ~6 launches each).
    The blocks are kept in PAIR order (the kernel wants the two blocks of a pair contiguous): one block permutation per
    round takes round r's order to round r + 1's.  A matrix whose every visit of one whole outer sweep reported "no
    large rotation" (sweeps > 0) is finished: its pairs are skipped from then on (active = -1 costs a 5 us launch slot
    instead of a 0.3 ms sweep), all on the device -- no host sync.
    A pair with an all-zero block (columns beyond the numerical rank: the rank-masked principal-angle problems) is
    skipped too: rotating against zero columns is a no-op, and a round in which every pair of every matrix is skipped
    costs no sweep at all.  A matrix with ONE non-zero block still needs that block's own columns rotated: it is
    visited with its sibling."""
    b, nblk, _, rows = xv.shape
    perms, restore = _tournament_perms(nblk, xv.device)
    npair = nblk // 2
    cur = xv
    done = done.clone()
    nz = (xv.abs().amax(dim=(2, 3)) > 0)                             # [b, nblk]: block holds a non-zero column
    lone = (nz.sum(dim=1) == 1).unsqueeze(1)                          # [b, 1]
    full_cols = torch.full((b, npair), 2 * WIDE_BLOCK, dtype=torch.int32, device=xv.device)
    for _ in range(WIDE_DIRECT_SWEEPS):
        quiet = torch.ones(b, dtype=torch.bool, device=xv.device)
        for perm, siblings in perms:
            cur = cur[:, perm].contiguous()                          # [b, nblk, 96, rows], pair p = blocks 2p, 2p + 1
            nz = nz[:, perm]
            nzp = nz.view(b, npair, 2)
            work = (nzp.all(dim=2) | (lone & nzp.any(dim=2) & siblings.unsqueeze(0))) & ~done.unsqueeze(1)
            act = torch.where(work, full_cols, -1).reshape(-1).contiguous()
            _, sw = ops.jacobi_svd(cur.view(b * npair, 2 * WIDE_BLOCK, rows), rows, max_sweeps=1, sort=False,
                                   flag_status=False, active=act)
            quiet &= (sw.view(b, npair) >= 0).all(dim=1)
        done |= quiet
    return cur[:, restore]


def _gram_tournament(ops, xv: torch.Tensor, zero_blocks: bool, skip_matrix: torch.Tensor | None) -> torch.Tensor:
    """Block-Jacobi tournament in the Gram form (pairs taller than 384 rows): per visit G = Xp^T Xp -> J -> Xp J.
    Blocks kept in PAIR order: one block permutation per round (the gather that forms the pairs) instead of a gather
    and a scatter back into the natural order.  ``zero_blocks``: rank-masked problems -- a pair with an all-zero block
    has nothing to rotate against (22 of the 28 pairs of a sweep for the principal-angle problems of the 768-wide
    selector, rank ~350 of 768; known on the DEVICE only): such pairs, and every pair of a matrix in ``skip_matrix``,
    are masked out of every launch of the visit (`skip`: the workgroups of a masked problem return at once, no host
    sync, no re-packing) and keep their blocks; a matrix with ONE non-zero block is visited with its sibling."""
    b, nblk, _, n_pad = xv.shape
    perms, restore = _tournament_perms(nblk, xv.device)
    npair = nblk // 2
    cur = xv
    masked = zero_blocks or skip_matrix is not None
    if masked:
        nz = (xv.abs().amax(dim=(2, 3)) > 0) if zero_blocks else torch.ones(b, nblk, dtype=torch.bool, device=xv.device)
        lone = (nz.sum(dim=1) == 1).unsqueeze(1)
        live = torch.ones(b, 1, dtype=torch.bool, device=xv.device) if skip_matrix is None else ~skip_matrix.unsqueeze(1)
    for _ in range(_wide_sweeps(nblk)):
        for perm, siblings in perms:
            xp = cur[:, perm].reshape(b * npair, 2 * WIDE_BLOCK, n_pad)
            skip = None
            if masked:
                nz = nz[:, perm]
                nzp = nz.view(b, npair, 2)
                work = (nzp.all(dim=2) | (lone & nzp.any(dim=2) & siblings.unsqueeze(0))) & live        # [b, npair]
                skip = (~work).reshape(-1).to(torch.int32)
            rot = _pair_rotation(ops.bgemm_f64(xp, xp, trans_b=True, symmetric=True, skip=skip), skip=skip)
            new = ops.bgemm_f64(rot, xp, out_dtype=torch.float32, skip=skip)
            if masked:
                new = torch.where(work.reshape(-1, 1, 1), new, xp)
            cur = new.view(b, nblk, WIDE_BLOCK, n_pad)
    return cur[:, restore]


def _psd_eig_blocked(a64: torch.Tensor, zero_blocks: bool = False):
    """Eigen-decomposition of PSD fp64 matrices [b, n, n] with n > 192 (student widths 384 / 768: BASELINE c4 / c5).

    One-sided BLOCK Jacobi on the blocked Cholesky factor X (A = X X^T): the columns form n_pad / 96 blocks; a round
    of the tournament takes nblk / 2 disjoint block pairs of every matrix.  Up to 384 rows a visit rotates the columns
    of a pair directly (``_tall_tournament``); beyond that, for each pair
        G = Xp^T Xp (192 x 192, fp64 MFMA)  ->  J = _pair_rotation(G)  (pivoted Cholesky + Jacobi)
        Xp <- Xp J                          (fp64-accumulated MFMA GEMM, stored fp32)
    (``_gram_tournament``).  The accuracy is that of one-sided Jacobi (relative per column).  The number of outer
    sweeps is fixed per width: there is no convergence test on the host.

    ``zero_blocks``: the input is rank-masked (the principal-angle Gram matrices A_bar A_bar^T are non-zero in their
    leading k_j x k_j corner only, k_j = the teacher layer's Marchenko-Pastur rank: ~160 of 384 at c4, ~350 of 768 at
    c5).  X is then zero outside a corner of the same size.  Which matrices fit which corner is known on the DEVICE only,
    so both paths are enqueued and each matrix takes exactly one of them through the problem masks of the kernels:
    matrices whose X fits the SMALL corner (192: one complete Jacobi solve of the 192 x 192 corner; 384 for the
    768-wide problems: the direct tournament over four blocks, 21 launches) take the small path, the others the general
    one, whose launches return at once for masked problems (a fully masked launch costs ~5 us).  No host sync, any rank.
    Returns (sigma [b, n] descending, u [b, n, n] rows = eigenvectors)."""
    ops = get_ops()
    b, n, _ = a64.shape
    n_pad = -(-n // WIDE_PANEL) * WIDE_PANEL
    x = _blocked_pchol(a64, n_pad)                                       # [b, n_pad(col), n_pad(row)]
    nblk = n_pad // WIDE_BLOCK
    xv = x.view(b, nblk, WIDE_BLOCK, n_pad)
    tall = n_pad <= ops.JACOBI_TALL_ROWS
    small = xs_new = None
    if zero_blocks:
        corner = 2 * WIDE_BLOCK if tall else ops.JACOBI_TALL_ROWS
        outside = torch.maximum(x[:, corner:, :].abs().amax(dim=(1, 2)), x[:, :corner, corner:].abs().amax(dim=(1, 2)))
        small = outside == 0                                             # [b]: X lives in its leading corner
        xs = x[:, :corner, :corner].contiguous()                         # [b, corner (col), corner (row)]
        if tall:
            w = torch.zeros(b, corner, ops.jacobi_ld(corner), dtype=torch.float32, device=x.device)
            w[:, :, :corner] = xs
            ops.jacobi_svd(w, corner, sort=False, active=torch.where(small, corner, -1).to(torch.int32), active_rows=2)
            xs_new = w[:, :, :corner]
        else:
            xs_new = _tall_tournament(ops, xs.view(b, corner // WIDE_BLOCK, WIDE_BLOCK, corner), ~small).reshape(b, corner, corner)
    if tall:
        xv.copy_(_tall_tournament(ops, xv, small if small is not None else torch.zeros(b, dtype=torch.bool, device=x.device)))
    else:
        xv.copy_(_gram_tournament(ops, xv, zero_blocks, small))
    if small is not None:
        x[:, :corner, :corner] = torch.where(small.view(-1, 1, 1), xs_new, x[:, :corner, :corner])
    # the sweep counts are fixed (no host sync): verify on the device that the columns ARE orthogonal and raise the
    # NONCONVERGED bit of the health word otherwise (surfaces as BasdLinAlgError at the next status check)
    gram = ops.bgemm_f64(x, x, trans_b=True, symmetric=True)             # [b, n_pad, n_pad] = column Gram of X
    nrm2 = torch.diagonal(gram, dim1=-2, dim2=-1)
    live = (nrm2 > 0).to(gram.dtype)                                      # columns dropped at the rank are exact zeros
    scale = torch.rsqrt(nrm2.clamp_min(1e-300)) * live
    cosmax = (gram * scale.unsqueeze(-1) * scale.unsqueeze(-2) - torch.diag_embed(live)).abs().amax(dim=(-2, -1))
    ops.flag_if_exceeds(cosmax, WIDE_COS_TOL, ops.STATUS_NONCONVERGED)        # atomic OR; a NaN raises the flag too
    nrm = nrm2.sqrt()                                                     # [b, n_pad] singular values = column norms
    order = torch.argsort(nrm, dim=-1, descending=True, stable=True)[:, :n]
    sigma = torch.gather(nrm, 1, order).float()
    cols = torch.gather(x, 1, order.unsqueeze(-1).expand(b, n, n_pad))[:, :, :n]
    u = torch.where(sigma.unsqueeze(-1) > 0, cols / sigma.clamp_min(1e-30).unsqueeze(-1), torch.zeros_like(cols))
    return sigma, u


def _pair_rotation(g: torch.Tensor, skip: torch.Tensor | None = None) -> torch.Tensor:
    """g = Xp^T Xp [b, k, k] fp64 (Gram of a column-block pair) -> J^T [b, k, k] fp64: row i holds the coefficients of
    output column i over the input columns; Xp J has mutually orthogonal columns sorted by norm, J is orthogonal.

    J has to be accurate in the GRADED sense (a big column may leak into a small one only by tol x sigma_small /
    sigma_big), which the normalised left vectors U of the LDS Jacobi are not: they are orthonormal to 1e-6 in
    absolute terms, and Xp U then has column cosines of 1e-6 x sigma_big / sigma_small (measured on the GPU: 2.5e-4 at
    D_s = 384, a 1e-2 error of the student gradient; an fp64 Newton refinement of U diverges once that product
    approaches 1).  Instead, with the pivoted Cholesky factor g = L L^T, the Jacobi orthogonalises the columns of
    F = L^T (F^T F = g, so its RIGHT singular vectors are the eigenvectors of g): F J = W, and J = L^-T W with the
    explicit fp64 inverse.  Then Xp J = (Xp L^-T) W = Q W with Q orthonormal: every output column inherits exactly the
    relative rounding of its own W column and cos(x'_i, x'_j) = cos(w_i, w_j) <= the Jacobi tolerance, for any grading.
    Output columns beyond the numerical rank of the pair (cancellation noise of dependent columns) are zeroed."""
    ops = get_ops()
    b, k, _ = g.shape
    if skip is None:
        _, lw, piv, rank = ops.pchol(g, PCHOL_TOL)                         # lw[b, step, row] = L[row, step]
        wf = torch.zeros(b, k, ops.jacobi_ld(k), dtype=torch.float32, device=g.device)
        wf[:, :, :k] = lw.transpose(1, 2)                                  # column r of F = row r of L (entries over steps)
        ops.jacobi_svd(wf, k)                                              # F J = W in place, columns sorted by norm
        l_inv = ops.trinv(lw, piv, rank)                                   # [b, step, row] = L^-1 (rows >= rank zero)
        rot = ops.bgemm_f64(wf[:, :, :k].contiguous(), l_inv)              # J^T[i, r] = sum_step W[i, step] L^-1[step, r]
    else:
        # ``skip`` (int32 [b], device): masked problems run through none of the kernels; their rows of the result are
        # unspecified (the caller keeps the old blocks)
        _, lw, piv, rank = ops.pchol(g, PCHOL_TOL, skip=skip)
        wf = torch.zeros(b, k, ops.jacobi_ld(k), dtype=torch.float32, device=g.device)
        wf[:, :, :k] = lw.transpose(1, 2)
        mask = torch.where(skip != 0, -1, k).to(torch.int32)               # < 0: skip, k: solve completely
        ops.jacobi_svd(wf, k, active=mask, active_rows=2)
        l_inv = ops.trinv(lw, piv, rank, skip=skip)
        rot = ops.bgemm_f64(wf[:, :, :k].contiguous(), l_inv, skip=skip)
    keep = torch.arange(k, device=g.device).unsqueeze(0) < rank.unsqueeze(1)
    return rot * keep.unsqueeze(-1)


def psd_eig(a64: torch.Tensor, lower_only: bool = False, zero_blocks: bool = False):
    """Batched eigen-decomposition of symmetric PSD fp64 matrices [b, n, n]
    (``lower_only``: only the lower triangles are meaningful, e.g. ``token_gram(..., mirror=False)``).

    Returns (sigma [b, n] fp32 descending = sqrt(eigenvalues), u [b, n, n] fp32 with
    ROW i the unit eigenvector i (zero rows beyond the numerical rank), aux).
    """
    ops = get_ops()
    n = a64.shape[-1]
    if n > WIDE_PANEL:
        sigma, u = _psd_eig_blocked(a64, zero_blocks=zero_blocks)   # zero_blocks: rank-masked input (a hint, not a promise)
        return sigma, u, None
    if not lower_only:
        a64 = 0.5 * (a64 + a64.transpose(-1, -2))
    w0, lwork, piv, rank = ops.pchol(a64, PCHOL_TOL)             # reads the lower triangle only
    sigma, _ = ops.jacobi_svd(w0, n)
    safe = sigma.clamp_min(1e-30).unsqueeze(-1)
    u = torch.where(sigma.unsqueeze(-1) > 0, w0[:, :, :n] / safe, torch.zeros_like(w0[:, :, :n]))
    return sigma, u, (w0, lwork, piv, rank)


def resample_matrix(n_in: int, n_out: int, device, dtype=torch.float32) -> torch.Tensor:
    """[n_out, n_in] matrix of F.interpolate(mode='linear', align_corners=False)."""
    pos = (torch.arange(n_out, device=device, dtype=torch.float64) + 0.5) * (n_in / n_out) - 0.5
    pos = pos.clamp(min=0.0)
    lo = pos.floor().long().clamp(max=n_in - 1)
    hi = (lo + 1).clamp(max=n_in - 1)
    fr = (pos - lo.double())
    r = torch.zeros(n_out, n_in, device=device, dtype=torch.float64)
    rows = torch.arange(n_out, device=device)
    r.index_put_((rows, lo), 1.0 - fr, accumulate=True)
    r.index_put_((rows, hi), fr, accumulate=True)
    return r.to(dtype)


def importance_from_attention(attn: torch.Tensor, has_cls_token: bool) -> torch.Tensor:
    """[B,H,T,T] (or compact [B,H,1,T]) attention -> [B,N] importance, relational.py:22-27."""
    if attn.dim() == 2:
        return attn.float()
    if has_cls_token:
        return attn[:, :, 0, 1:].float().mean(dim=1)
    return attn.float().mean(dim=(1, 2))


# --------------------------------------------------------------------------- #
# selector weights
# --------------------------------------------------------------------------- #
@torch.no_grad()
def teacher_gram(tokens, proj_t):
    """(uncentred Gram [D, D] fp64 lower triangle, column sums [D] fp64) of one teacher layer's projected tokens:
    the per-layer piece of ``teacher_frames`` (can be launched as soon as the layer's block has run)."""
    return get_ops().token_gram(tokens, proj_t, mirror=False)


def _layer_grams(tokens, proj, grams=None):
    """Uncentred Gram [n, D, D] and column sums [n, D] (fp64, lower triangles) of n token tensors: ONE zeroed
    allocation for all layers, the per-layer kernels accumulate into its slices (no per-layer fills / stacks)."""
    ops = get_ops()
    if grams is not None:
        return torch.stack([g for g, _ in grams]), torch.stack([c for _, c in grams])
    n, d = len(tokens), proj.shape[0]
    unc = torch.zeros(n, d, d, dtype=torch.float64, device=proj.device)
    csum = torch.zeros(n, d, dtype=torch.float64, device=proj.device)
    for i, x in enumerate(tokens):
        ops.token_gram(x, proj, mirror=False, out=(unc[i], csum[i]))
    return unc, csum


def teacher_frames(teacher_tokens, proj_t, grams=None):
    """Teacher half of the selector (no gradient): per-layer Gram statistics -> MP ranks (device
    int32, no host sync) and rank-masked PCA frames.  Reference layer_selector.py:69-74, 133-138.

    Returned dict can be handed to ``selector_weights(..., frames=...)``; the trainer computes it on
    a side stream while the student forward runs (only 2 L small eigenproblems: 24 workgroups).
    """
    ops = get_ops()
    L = len(teacher_tokens)
    D = proj_t.shape[0]
    m_t = teacher_tokens[0].shape[0] * teacher_tokens[0].shape[1]
    unc, csum = _layer_grams(teacher_tokens, proj_t, grams)              # layer_selector.py:71-73, :134-136
    cen = unc - csum.unsqueeze(2) * csum.unsqueeze(1) / m_t              # centred Gram of every layer at once
    sigma, u, _ = psd_eig(torch.cat([unc, cen]), lower_only=True)
    ranks = ops.mp_rank(sigma[:L] ** 2, m_t, D, D - 1)  # int32 [L], stays on device
    v_t, s_t = u[L:], sigma[L:]
    idx = torch.arange(D, device=proj_t.device)
    keep = (idx.unsqueeze(0) < ranks.unsqueeze(1)).float()      # [L, D]  (index < k_j)
    return {"ranks": ranks, "keep": keep, "vm_t": v_t * keep.unsqueeze(-1), "sw": s_t * keep}


@torch.no_grad()
def teacher_ranks(teacher_tokens, proj_t) -> torch.Tensor:
    """MP ranks only (device int32 [L]): all a single-layer teacher needs -- with one teacher layer the softmax over
    layers is the constant 1 (reference layer_selector.py:108), so neither PCA frames nor principal angles nor the
    student's eigen-decompositions influence value or gradient ("c3 degenerates, exactly")."""
    ops = get_ops()
    D = proj_t.shape[0]
    m_t = teacher_tokens[0].shape[0] * teacher_tokens[0].shape[1]
    unc, _ = _layer_grams(teacher_tokens, proj_t)
    sigma, _, _ = psd_eig(unc, lower_only=True)
    return ops.mp_rank(sigma ** 2, m_t, D, D - 1)


@torch.no_grad()
def student_frames(student_tokens, proj_s):
    """Student half of the selector statistics (centred Gram eigen-decomposition per extraction point):
    (sigma [E, D], v [E, D, D]).  No gradient flows through these tensors themselves -- the selector backward
    differentiates the eigen-decomposition analytically from them -- so a trainer may compute them ahead of the
    loss (reference layer_selector.py:84-92)."""
    m_s = student_tokens[0].shape[0] * student_tokens[0].shape[1]
    unc, csum = _layer_grams([s.detach() for s in student_tokens], proj_s)
    sigma_s, v_s, _ = psd_eig(unc - csum.unsqueeze(2) * csum.unsqueeze(1) / m_s, lower_only=True)
    return sigma_s, v_s


class _SelectorWeightsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_temp, proj_s, ranks, keep, vm_t, sw, ready, pre_s, *student):
        ops = get_ops()
        E, L = len(student), vm_t.shape[0]
        D = proj_s.shape[0]
        dev = proj_s.device

        sigma_s, v_s = pre_s if pre_s is not None else student_frames(student, proj_s)
        lam_s = sigma_s.double() ** 2
        if ready is not None:
            ready()           # join the stream that produced the teacher frames only now: the student's own
                              # (latency-bound, 8-workgroup) eigen-solve above overlaps the teacher's

        # The small products of the selector (E L matrices of D x D, plus four [M, D_s] x [D_s, D_s] products in the
        # backward) stay on torch.matmul: moved to the own fp64-accumulated batched GEMM they cost +2.4 ms per c2 step
        # (bgemm_f64 7.0 vs 4.6 ms in the instrumented step, same box) -- measured in round 3 and reverted.
        a_full = torch.einsum("ibd,jcd->ijbc", v_s, vm_t)           # [E, L, D(b), D(c)]
        a_bar = a_full * keep.view(1, L, D, 1)                      # rows b < k_j

        if D <= WIDE_PANEL:
            ld = ops.jacobi_ld(D)
            w = torch.zeros(E * L, D, ld, device=dev, dtype=torch.float32)
            w[:, :, :D] = a_bar.reshape(E * L, D, D).transpose(1, 2)    # column c contiguous
            # only the leading k_j x k_j block of pair (i, j) is non-zero: sweep just that block
            sig, _ = ops.jacobi_svd(w, D, active=ranks.repeat(E), active_rows=True)   # cosines, descending
            sig = sig.view(E, L, D)
            vec = w[:, :, :D].view(E, L, D, D)                          # [E, L, m, b]: sigma_m u_m (un-normalised)
            unnormalised = True
        else:
            # wide students (D_s = 384 / 768): cosines and left singular vectors from the blocked eigen-solver on
            # the fp64 Gram A_bar A_bar^T (zero rows / columns beyond k_j: the blocked Cholesky stops at the rank)
            ab = a_bar.reshape(E * L, D, D)
            sig, vec, _ = psd_eig(ops.bgemm_f64(ab, ab, trans_b=True, symmetric=True), zero_blocks=True)
            sig, vec = sig.view(E, L, D), vec.view(E, L, D, D)
            unnormalised = False
        # acos / spectral weighting / softmax over layers and the diagonal of the backward seed in ONE kernel
        # (layer_selector.py:100-108): d2 [E, L], pre = -d2 / tau, weights, coef [E, L, D]
        d2, pre, wts, coef = ops.angle_weights(sig, sw, log_temp, unnormalised)
        tau = F.softplus(log_temp.float())

        # seeds of the backward, per unit d(d2_ij): Phi = U diag(gsig / sig) U^T = vec^T diag(coef) vec,
        # T = A_full A_bar^T Phi
        phi = torch.matmul((vec * coef.unsqueeze(-1)).transpose(-1, -2), vec)         # [E, L, b, c]
        t_seed = a_full @ a_bar.transpose(-1, -2) @ phi              # [E, L, D(b), D(a)]
        t_seed = t_seed * (1.0 - keep).view(1, L, D, 1)              # only b >= k_j (cross-subspace terms)

        ctx.save_for_backward(log_temp, proj_s, wts, d2, tau, t_seed, v_s, lam_s, *student)
        ctx.n_student = E
        return wts, pre

    @staticmethod
    def backward(ctx, g_w, g_pre_out):
        log_temp, proj_s, wts, d2, tau, t_seed, v_s, lam_s, *student = ctx.saved_tensors
        E = ctx.n_student
        ops = get_ops()
        if getattr(ops, "angle_weights_bwd", None) is not None and wts.shape[1] <= 64 and ops.handles(t_seed):
            # one C entry (basd_angle_weights_bwd): reduction over the teacher layers, eigenvalue-gap division and the
            # four small fp64 products in six launches on a workspace
            g_lt, w_tok = ops.angle_weights_bwd(g_w, g_pre_out, wts, d2, log_temp, t_seed, v_s, lam_s, proj_s)
            g_lt = g_lt.to(log_temp.dtype)
            grads = []
            for i in range(E):
                s = student[i]
                # (s - mean) W = s W - mean W: one fp32 copy of the tokens, the centring as the GEMM's bias row (the
                # explicit centred copy was two more passes over the [B N, D] tokens per extraction point)
                sf = s.float().reshape(-1, s.shape[-1])
                row = -(sf.mean(dim=0, keepdim=True) @ w_tok[i])
                grads.append(torch.addmm(row, sf, w_tok[i]).reshape(s.shape).to(s.dtype))
            return (g_lt, None, None, None, None, None, None, None, *grads)
        g_pre = wts * (g_w - (wts * g_w).sum(dim=1, keepdim=True))
        if g_pre_out is not None:
            g_pre = g_pre + g_pre_out
        g_d2 = -g_pre / tau.unsqueeze(1)
        g_tau = (g_pre * d2).sum(dim=1) / (tau * tau)
        g_lt = (g_tau * torch.sigmoid(log_temp.float())).to(log_temp.dtype)

        c = (g_d2.unsqueeze(-1).unsqueeze(-1) * t_seed).sum(dim=1).double()       # [E, D(b), D(a)]
        gap = lam_s.unsqueeze(1) - lam_s.unsqueeze(2)                            # [E, b, a] = lam_a - lam_b
        k = torch.where(gap.abs() > 0, c / torch.where(gap.abs() > 0, gap, torch.ones_like(gap)),
                        torch.zeros_like(c))
        v64 = v_s.double()
        g_gram = v64.transpose(1, 2) @ k @ v64                                   # [E, D, D]
        p64 = proj_s.double()
        w_tok = (p64.t() @ (g_gram + g_gram.transpose(1, 2)) @ p64).float()      # [E, D_s, D_s]
        grads = []
        for i in range(E):
            s = student[i]
            # centring z = s P^T over rows == centring s (linear map), so d loss / d s = (s - mean) W
            centred = s.float() - s.float().mean(dim=(0, 1), keepdim=True)
            grads.append((centred.reshape(-1, s.shape[-1]) @ w_tok[i]).reshape(s.shape).to(s.dtype))
        return (g_lt, None, None, None, None, None, None, None, *grads)


def selector_weights(student_tokens, teacher_tokens, proj_s, proj_t, log_temperatures, frames=None, pre_student=None):
    """-> (weights [E, L] with grad, ranks int32 [L] on device, pre_softmax [E, L]).

    ``frames`` = a ``teacher_frames`` result computed earlier (e.g. on a side stream);
    ``pre_student`` = a ``student_frames`` result for exactly these student tokens."""
    if frames is None:
        frames = teacher_frames([t.detach() for t in teacher_tokens], proj_t)
    # batch-strided views (CLS-stripped block outputs) are consumed in place
    wts, pre = _SelectorWeightsFn.apply(log_temperatures, proj_s, frames["ranks"], frames["keep"], frames["vm_t"],
                                        frames["sw"], frames.get("ready"), pre_student, *student_tokens)
    return wts, frames["ranks"], pre


# --------------------------------------------------------------------------- #
# mixing
# --------------------------------------------------------------------------- #
class _MixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, *layers):
        ops = get_ops()
        ctx.layers = layers
        return ops.mix_tokens(list(layers), w.detach().float())

    @staticmethod
    def backward(ctx, g):
        ops = get_ops()
        dots = ops.mix_grad_dots(list(ctx.layers), g.contiguous().float())
        return (dots.float(), *([None] * len(ctx.layers)))


def mix_layers(w: torch.Tensor, layers) -> torch.Tensor:
    """w [E, L], layers: L same-shape tensors -> [E, *shape] fp32 (grad flows to w only)."""
    return _MixFn.apply(w, *[t.detach() for t in layers])


# --------------------------------------------------------------------------- #
# Procrustes
# --------------------------------------------------------------------------- #
class _ProcrustesFn(torch.autograd.Function):
    """All extraction points at once: inputs are E student tensors [B,N_s,D_s], the mixed teacher
    tokens [E,B,N_t,D_t] and the mixed importance [E,B,N_t]; the E*B cross-covariances go through
    ONE chain of batched launches (fp64 GEMMs, pivoted Cholesky, Jacobi, triangular inverse)."""

    @staticmethod
    def forward(ctx, t_all, imp_all, *students):
        ops = get_ops()
        E = len(students)
        n_s, n_t = students[0].shape[1], t_all.shape[2]
        b, d_s, d_t, dev = students[0].shape[0], students[0].shape[2], t_all.shape[3], t_all.device
        s_w = torch.empty(E * b, n_s, d_s, dtype=torch.float32, device=dev)                    # [E*B, ...]
        t_w = torch.empty(E * b, n_s, d_t, dtype=torch.float32, device=dev)
        a = torch.empty(E * b, n_s, dtype=torch.float32, device=dev)
        tr = torch.empty(E * b, 2, dtype=torch.float32, device=dev)
        for i in range(E):
            sl = slice(i * b, (i + 1) * b)
            ops.procrustes_prep(students[i], t_all[i], imp_all[i], out=(s_w[sl], t_w[sl], a[sl], tr[sl]))
        ctx.token_side = n_s <= d_s
        if not ops.jacobi_fits(min(n_s, d_s), min(n_s, d_s)):
            raise BasdShapeError(f"Procrustes core min(N_s, D_s) = {min(n_s, d_s)} does not fit the LDS-resident Jacobi")
        # nuclear norm of cross = s_w^T t_w and the two factors of the backward (cross itself is never formed):
        #   token side  (N_s <= D_s):  t_w G^T = fac_s s_w  with fac_s = A_s [N, N];   feature side: fac_s = t_w G^T [N, D_s];
        #   both:                      s_w G = a_t t_w                                  (G = U V^T)
        nuc, fac_s, a_t = ops.procrustes_fwd(s_w, t_w, PCHOL_TOL)
        ctx.save_for_backward(s_w, t_w, a, imp_all, fac_s, a_t)
        ctx.n_t, ctx.E = n_t, E
        ctx.s_dtype = students[0].dtype
        return (tr[:, 0] + tr[:, 1] - 2.0 * nuc).view(E, -1)

    @staticmethod
    def backward(ctx, g_loss):
        s_w, t_w, a, imp_all, *polar = ctx.saved_tensors
        E = ctx.E
        n_s, n_t = s_w.shape[1], ctx.n_t
        ops = get_ops()
        gl = g_loss.float().reshape(-1).contiguous()
        # residuals R = W - (other side) G^T, their scaling by 2 gl sqrt(a) and the row dots <R, W> that make up
        # d loss / d a.  ONE C entry (basd_procrustes_bwd): the batched product a_t t_w (E B matrices [N, N] x [N, D_t],
        # 60 GF at c2) runs as a bf16 three-product split on the matrix cores with the residual in its epilogue --
        # rounds 1 - 3 used a library fp32 bmm (0.62 ms) plus a separate row pass here.
        if PROCRUSTES_BWD_FUSED and getattr(ops, "procrustes_bwd_supported", None) is not None and \
                ops.procrustes_bwd_supported(n_s, s_w.shape[2], t_w.shape[2]) and ctx.s_dtype in (torch.float32, torch.bfloat16):
            g_s, g_t, g_a = ops.procrustes_bwd(s_w, t_w, a, gl, polar[0].contiguous(), polar[1].contiguous(), ctx.s_dtype)
        else:
            note_library_gemm("Procrustes backward: fp32 bmm (A_t t_w)")
            if ctx.token_side:
                p_s, p_t = polar[0] @ s_w, polar[1] @ t_w                # t_w G^T = A_s s_w, s_w G = A_t t_w
            else:
                # t_w G^T was formed in the forward (the fp32 row kernel below works in place: keep the saved copy
                # intact for a second backward); s_w G = A_t t_w
                p_s = polar[0].clone() if ctx.s_dtype == torch.float32 else polar[0]
                p_t = polar[1] @ t_w
            g_s, dot_s = ops.procrustes_bwd_rows(p_s, s_w, a, gl, out_dtype=ctx.s_dtype)
            g_t, dot_t = ops.procrustes_bwd_rows(p_t, t_w, a, gl, out_dtype=torch.float32)
            g_a = (dot_s + dot_t) / (2.0 * a)
        imp = imp_all.float().reshape(-1, n_t)
        if n_t != n_s:
            r = resample_matrix(n_t, n_s, s_w.device)                  # [n_s, n_t]
            tot = (imp @ r.t()).sum(-1, keepdim=True)
            g_raw = (g_a - (a * g_a).sum(-1, keepdim=True)) / tot
            g_imp = g_raw @ r
            g_t = torch.matmul(r.t(), g_t)
        else:
            tot = imp.sum(-1, keepdim=True)
            g_imp = (g_a - (a * g_a).sum(-1, keepdim=True)) / tot
        b = g_s.shape[0] // E
        return (g_t.view(E, b, n_t, -1), g_imp.view(E, b, n_t), *g_s.view(E, b, n_s, -1).unbind(0))


def procrustes_all(students, t_all: torch.Tensor, imp_all: torch.Tensor) -> torch.Tensor:
    """students: E tensors [B,N_s,D_s]; t_all [E,B,N_t,D_t]; imp_all [E,B,N_t] -> values [E, B]."""
    return _ProcrustesFn.apply(t_all.contiguous().float(), imp_all.contiguous().float(), *students)


def procrustes(s: torch.Tensor, t: torch.Tensor, imp: torch.Tensor) -> torch.Tensor:
    """Per-sample attention-weighted Procrustes value [B]; differentiable in s, t, imp."""
    return procrustes_all([s], t.unsqueeze(0), imp.unsqueeze(0))[0]
