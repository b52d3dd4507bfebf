"""TEST-ONLY torch/CPU emulation of the C-ABI kernels (same call surface as
``basd_amd._native``).  It lets the CPU suite check the host logic -- autograd
formulas, masking, module plumbing -- against the golden vectors.  It is never
imported by the package; the product path is the HIP library only."""
from __future__ import annotations

import math

import torch

JACOBI_LDS_BYTES = 163840
# Emulated rounding of the fp32 Jacobi: the real kernel stops at pair cosines of ~1e-6, so its singular vectors are
# orthonormal to ~1e-6 only (absolute; JACOBI_NOISE is per ENTRY, 1e-7 gives that level at n = 192).  Host logic that is sensitive to this (the blocked eigensolver applies them to
# graded factors) is tested with this switched on.
JACOBI_NOISE = 0.0


def handles(t) -> bool:
    return True


STATUS_NONCONVERGED, STATUS_NONFINITE, STATUS_RANK0 = 1, 2, 4
_STATUS = torch.zeros(1, dtype=torch.int32)


def status_word(device=None):
    return _STATUS


def flag_if_exceeds(values, tol, bit):
    if bool((~(values.double() <= tol)).any()):
        _STATUS.bitwise_or_(torch.tensor([bit], dtype=torch.int32))


def raise_for_status(value):
    from basd_amd._native import raise_for_status as real
    real(value)


def check_status(device=None):
    value = int(_STATUS[0])
    _STATUS.zero_()
    raise_for_status(value)


def jacobi_ld(m_rows: int) -> int:
    ld = (m_rows + 3) // 4 * 4
    if ld % 32 == 0:
        ld += 4
    return ld


def jacobi_fits(n_cols: int, m_rows: int) -> bool:
    return n_cols <= 256 and n_cols * jacobi_ld(m_rows) * 4 + 520 * 4 <= JACOBI_LDS_BYTES


def token_gram(x, proj, mirror=True, out=None):
    x = x.reshape(-1, x.shape[-1])
    z = (x.float() @ proj.float().t()).double()
    g = z.t() @ z
    if not mirror:                                   # like the kernels: the strict upper triangle is not meaningful
        g = torch.tril(g) + torch.triu(torch.full_like(g, float("nan")), 1)
    if out is not None:
        out[0].copy_(g)
        out[1].copy_(z.sum(0))
        return out
    return g, z.sum(0)


def _poison(skip, *outs):
    """masked problems (``skip[b] != 0``) leave their outputs unspecified on the device: NaN / garbage here, so that host
    logic which reads them fails loudly"""
    if skip is not None:
        m = skip != 0
        for o in outs:
            if o.is_floating_point():
                o[m] = float("nan")
            else:
                o[m] = -12345
    return outs[0] if len(outs) == 1 else outs


def pchol(a, tol=1e-13, dmax_ref=None, skip=None):
    """Batched (vectorised over the batch) diagonal-pivoted Cholesky with the kernel's output contract."""
    if skip is not None:
        a = torch.where((skip != 0).view(-1, 1, 1), torch.eye(a.shape[-1], dtype=a.dtype).expand_as(a), a)
        return _poison(skip, *pchol(a, tol, dmax_ref))
    a = a.double()
    a = torch.tril(a) + torch.tril(a, -1).transpose(-1, -2)     # only the lower triangle is read
    batch, n, _ = a.shape
    ld = jacobi_ld(n)
    lwork = torch.zeros(batch, n, n, dtype=torch.float64)
    piv = torch.zeros(batch, n, dtype=torch.long)
    rank = torch.full((batch,), n, dtype=torch.long)
    d = torch.diagonal(a, dim1=-2, dim2=-1).clone()
    alive = torch.ones(batch, n, dtype=torch.bool)
    running = torch.ones(batch, dtype=torch.bool)
    dmax0 = d.max(dim=1).values if dmax_ref is None else dmax_ref.double().reshape(batch)
    ar = torch.arange(batch)
    for k in range(n):
        cand = torch.where(alive, d, torch.full_like(d, -1e300))
        pval, p = cand.max(dim=1)                                 # first index on ties, like the kernel
        ok = running & (pval > tol * dmax0) & (pval > 0)
        newly = running & ~ok
        rank[newly] = k
        running = ok
        if not bool(running.any()):
            break
        row_p = a[ar, p]                                          # [batch, n] = A[p, :]
        lp = lwork[ar, :, p]                                      # [batch, steps] = L[p, :k]
        v = row_p - torch.einsum("bkr,bk->br", lwork, lp)
        v = torch.where(alive, v, torch.zeros_like(v))
        lkk = pval.clamp_min(0).sqrt()
        col = v / lkk.clamp_min(1e-300).unsqueeze(1)
        col[ar, p] = lkk
        col = torch.where(running.unsqueeze(1), col, torch.zeros_like(col))
        lwork[:, k] = col
        d = torch.where(running.unsqueeze(1), d - col * col, d)
        upd = alive.clone()
        upd[ar, p] = False
        alive = torch.where(running.unsqueeze(1), upd, alive)
        piv[:, k] = torch.where(running, p, piv[:, k])
    # complete the permutation with the rows never pivoted (ascending), per matrix
    for b_ in range(batch):
        r = int(rank[b_])
        rest = torch.nonzero(alive[b_]).flatten()
        piv[b_, r:] = rest[: n - r]
    w0 = torch.zeros(batch, n, ld, dtype=torch.float32)
    w0[:, :, :n] = lwork.float()
    return w0, lwork, piv.int(), rank.int()


JACOBI_TALL_ROWS = 384


def jacobi_svd(w, m_rows, norm_rows=None, *, tol=None, max_sweeps=40, sort=True, active=None, active_rows=False,
               flag_status=True):
    batch, n_cols, ld = w.shape
    if norm_rows is None:
        norm_rows = m_rows
    sigma = torch.zeros(batch, n_cols, dtype=torch.float32)
    live = torch.ones(batch, dtype=torch.bool) if active is None else (active.view(-1) >= 0)
    if active is not None and int(active_rows) == 2:
        assert bool(((active.view(-1) < 0) | (active.view(-1) == n_cols)).all()), "active_rows == 2: a pure mask"
        active_rows = False
    if not bool(torch.isfinite(w[live]).all()):
        _STATUS[0] |= 2
        w.copy_(torch.nan_to_num(w, nan=0.0, posinf=0.0, neginf=0.0))
        sigma.fill_(float("nan"))
        return sigma, torch.full((batch,), 1, dtype=torch.int32)
    sweeps = torch.ones(batch, dtype=torch.int32)
    for b in range(batch):
        if active is not None and int(active[b]) < 0:  # skipped matrix: untouched
            sweeps[b] = 0
            continue
        full = w[b, :, :m_rows].double().t()           # [m, n]
        top = full[:norm_rows]
        # sweeps < 0: "still made a large rotation" = the input columns were not yet orthogonal to the kernel's
        # quadratic-convergence threshold (pair |cos| 5e-4)
        nrm_in = top.norm(dim=0)
        if int((nrm_in > 0).sum()) > 1:
            cn_in = top[:, nrm_in > 0] / nrm_in[nrm_in > 0]
            off = (cn_in.t() @ cn_in - torch.eye(cn_in.shape[1], dtype=torch.float64)).abs().max()
            if float(off) > 5e-4 and max_sweeps < 3:
                sweeps[b] = -1
        u, s, vh = torch.linalg.svd(top, full_matrices=False)
        r = s.shape[0]
        out = torch.zeros(m_rows, n_cols, dtype=torch.float64)
        small = s < 1e-13 * max(float(s[0]), 1e-300)
        s = torch.where(small, torch.zeros_like(s), s)
        # a converged input stays put in the real kernel (no rotation fires): only perturb when there was work to do
        nz = top.norm(dim=0) > 0
        gcos = torch.zeros(())
        if JACOBI_NOISE and int(nz.sum()) > 1:
            cn = top[:, nz] / top[:, nz].norm(dim=0, keepdim=True)
            gcos = (cn.t() @ cn - torch.eye(int(nz.sum()), dtype=torch.float64)).abs().max()
        if JACOBI_NOISE and float(gcos) > 1e-5:
            gen = torch.Generator().manual_seed(b)
            u = u + JACOBI_NOISE * torch.randn(u.shape, generator=gen, dtype=torch.float64)
            u = u / u.norm(dim=0, keepdim=True)
        out[:norm_rows, :r] = u * s
        if m_rows > norm_rows:                          # stacked [A; B]: B V
            out[norm_rows:, :r] = full[norm_rows:] @ vh.t()
        w[b, :, :m_rows] = out.t().float()
        sigma[b, :r] = s.float()
    return sigma, sweeps


def mp_rank(evals, rows, d, cap):
    out = []
    for e in evals:
        lam = (e.double() / rows)
        srt, _ = torch.sort(lam, descending=True)
        n_eff = min(rows, lam.numel())
        srt = srt[:n_eff]
        asc = srt.flip(0)
        sigma2 = asc[(n_eff - 1) // 2]
        edge = float(sigma2.float()) * (1.0 + math.sqrt(d / rows)) ** 2
        out.append(min(int((srt.float() > edge).sum()), cap))
        if out[-1] == 0:
            _STATUS[0] |= 4 if math.isfinite(edge) else 2
    return torch.tensor(out, dtype=torch.int32)


def angle_weights(sigma, sw, log_temp, unnormalised):
    eps = float(torch.finfo(torch.float32).eps)
    sig = sigma.float()
    sig_c = sig.clamp(max=1.0 - eps)
    theta = torch.acos(sig_c)
    den = sw.sum(-1)
    d2 = (sw.unsqueeze(0) * theta * theta).sum(-1) / den.unsqueeze(0)
    tau = torch.nn.functional.softplus(log_temp.detach().float())
    pre = -d2 / tau.unsqueeze(1)
    gsig = sw.unsqueeze(0) * 2.0 * theta * (-1.0 / torch.sqrt(1.0 - sig_c * sig_c)) / den.view(1, -1, 1)
    floor = 1e-9 if unnormalised else 1e-12
    ok = (sig <= 1.0 - eps) & (sig > floor)
    safe = sig.clamp_min(floor)
    coef = torch.where(ok, gsig / (safe ** 3 if unnormalised else safe), torch.zeros_like(gsig))
    return d2, pre, torch.softmax(pre, dim=1), coef


def mix_tokens(layers, w):
    st = torch.stack([t.float() for t in layers])
    return torch.einsum("el,l...->e...", w.float(), st).contiguous()


def mix_grad_dots(layers, g):
    st = torch.stack([t.double() for t in layers])
    return torch.einsum("e...,l...->el", g.double(), st)


def procrustes_prep(s, t, imp, out=None):
    from oracle.basd_oracle import resample_linear
    s, t, imp = s.float(), t.float(), imp.float()
    n_s = s.shape[1]
    t = resample_linear(t, n_s)
    if imp.shape[1] != n_s:
        imp = resample_linear(imp.unsqueeze(-1), n_s).squeeze(-1)
    a = imp / imp.sum(-1, keepdim=True)
    wc = a.unsqueeze(-1)
    s_w = wc.sqrt() * (s - (wc * s).sum(1, keepdim=True))
    t_w = wc.sqrt() * (t - (wc * t).sum(1, keepdim=True))
    tr = torch.stack([(s_w.double() ** 2).sum((1, 2)), (t_w.double() ** 2).sum((1, 2))], dim=1).float()
    if out is not None:
        for o, v in zip(out, (s_w, t_w, a, tr)):
            o.copy_(v)
        return out
    return s_w.contiguous(), t_w.contiguous(), a, tr


def procrustes_bwd_rows(r, w, a, gl, out_dtype=torch.float32):
    c2 = 2.0 * gl.reshape(-1, 1)
    res = w - r
    rowdot = c2 * (res * w).sum(-1)
    scaled = (c2 * a.sqrt()).unsqueeze(-1) * res
    if out_dtype == torch.float32:
        r.copy_(scaled)
        return r, rowdot
    return scaled.to(out_dtype), rowdot


def sf_adamw_step(y, g, z, v, *, lr, beta1, beta2, eps, weight_decay, ckp1, bias_correction2):
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (v / bias_correction2).sqrt_().add_(eps)
    gn = g / denom + weight_decay * y
    y.lerp_(z, ckp1)
    y.add_(gn, alpha=lr * (beta1 * (1 - ckp1) - 1))
    z.sub_(gn, alpha=lr)


def lerp_(y, z, w):
    y.add_(z - y, alpha=w)


def bgemm_f64(a, b, *, trans_a=False, trans_b=False, out_dtype=torch.float64, symmetric=False, skip=None):
    a, b = a.double(), b.double()
    if skip is not None:            # masked problems may hold unspecified inputs
        a, b = torch.nan_to_num(a), torch.nan_to_num(b)
    if trans_a:
        a = a.transpose(1, 2)
    if trans_b:
        b = b.transpose(1, 2)
    return _poison(skip, (a @ b).to(out_dtype))


def trinv(lwork, piv, rank, skip=None):
    if skip is not None:
        m = skip != 0
        lwork, piv, rank = lwork.clone(), piv.clone(), rank.clone()
        lwork[m] = torch.eye(lwork.shape[-1], dtype=lwork.dtype)
        piv[m] = torch.arange(piv.shape[-1], dtype=piv.dtype)
        rank[m] = 0
        return _poison(skip, trinv(lwork, piv, rank))
    batch, n, _ = lwork.shape
    out = torch.zeros(batch, n, n, dtype=torch.float64)
    for b in range(batch):
        r = int(rank[b])
        pv = piv[b].long()
        lp = lwork[b].t()[pv]                  # [r', k] lower triangular in pivot order
        lp = lp.clone()
        lp[:, r:] = 0.0
        lp[range(r, n), range(r, n)] = 1.0
        x = torch.linalg.solve_triangular(lp, torch.eye(n, dtype=torch.float64), upper=False)
        x[r:] = 0.0
        out[b][:, pv] = x
    return out


def gemm_supported(n, k):
    return k % 64 == 0 and k >= 64 and n >= 128 and (n % 256 == 0 or n % 192 == 0 or n % 128 == 0)


def gemm_bf16(x, w, bias=None, gelu=False):
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    if gelu:
        y = torch.nn.functional.gelu(y)
    return y.to(torch.bfloat16)


def procrustes_fwd(s_w, t_w, tol=1e-13):
    import sys
    from basd_amd.losses.procrustes_chain import procrustes_fwd_chain
    return procrustes_fwd_chain(sys.modules[__name__], s_w, t_w, tol)


def ce_uwso(logits, targets, smoothing, geo):
    z = logits.detach().double()
    b, c = z.shape
    t = targets.double() if targets.dim() == 2 else torch.nn.functional.one_hot(targets, c).double()
    t = (1.0 - smoothing) * t + smoothing / c
    logp = torch.log_softmax(z, dim=-1)
    ce = -(t * logp).sum(-1).mean()
    eps = torch.finfo(torch.float32).eps
    if geo is None:
        w_ce, w_geo, g = 1.0, 0.0, 0.0
    else:
        g = geo.detach().double().reshape(())
        ic, ig = 1.0 / ce.clamp_min(eps), 1.0 / g.clamp_min(eps)
        w_ce, w_geo = ic / (ic + ig), ig / (ic + ig)
    dl = (logp.exp() * t.sum(-1, keepdim=True) - t) * (w_ce / b)
    out4 = torch.stack([torch.as_tensor(w_ce * ce + w_geo * g), ce, torch.as_tensor(w_ce).double(),
                        torch.as_tensor(w_geo).double()]).float()
    return out4, dl.float()


def transpose_table(master, out, table):
    for src, dst, rows, cols in table:
        out[dst:dst + rows * cols].view(cols, rows).copy_(master[src:src + rows * cols].view(rows, cols).t())


def gemm_gelu_fwd(x, w, bias):
    pre = gemm_bf16(x, w, bias)
    return pre, torch.nn.functional.gelu(pre.float()).to(torch.bfloat16)


def gemm_gelu_bwd(dy, wt, pre):
    p = pre.float()
    cdf = 0.5 * (1.0 + torch.erf(p * 0.7071067811865476))
    pdf = torch.exp(-0.5 * p * p) * 0.3989422804014327
    return ((dy.float() @ wt.float().t()) * (cdf + p * pdf)).to(torch.bfloat16)


def wgrad_supported(n, k):
    return n % 64 == 0 and k % 64 == 0 and n >= 64 and k >= 64


def wgrad_bf16(dy, x, need_bias=True, out_w=None, out_b=None):
    dw = dy.float().t() @ x.float()
    db = dy.float().sum(0) if (need_bias or out_b is not None) else None
    if out_w is not None:
        out_w.add_(dw)
        dw = out_w
    if out_b is not None:
        out_b.add_(db)
        db = out_b
    return dw, db


def cls_importance_supported(t, hd):
    return 2 <= t <= 256 and hd in (32, 64)


def cls_importance(qkv, heads, head_dim, scale):
    b, t, _ = qkv.shape
    x = qkv.reshape(b, t, 3, heads, head_dim)
    q = x[:, 0, 0].float()                                  # [B, H, hd]  CLS query
    k = x[:, :, 1].float()                                  # [B, T, H, hd]
    logits = torch.einsum("bhd,bthd->bht", q, k).to(torch.bfloat16).float() * scale
    return logits.softmax(dim=-1)[:, :, 1:].mean(dim=1)


def add_layernorm_fwd(x, residual, gamma, beta, eps, row_scale=None, want_stats=False):
    xf = x.float()
    if row_scale is not None:
        xf = xf * row_scale.float().view(-1, *([1] * (x.dim() - 1)))
    s = (xf + residual.float()).to(torch.bfloat16)
    y, mean, rstd = layernorm_fwd(s, gamma, beta, eps)
    return (s, y, mean, rstd) if want_stats else (s, y)


def attention_fwd_supported(t, hd):
    return hd == 64 and 1 <= t <= 272


def attention_fwd(qkv, heads, head_dim, scale, want_importance=False, want_lse=False, query_mean=False):
    b, t, _ = qkv.shape
    x = qkv.reshape(b, t, 3, heads, head_dim).permute(2, 0, 3, 1, 4).float()
    q, k, v = x[0], x[1], x[2]
    logits = (q @ k.transpose(-1, -2)) * scale
    p = logits.softmax(dim=-1)
    out = (p.to(torch.bfloat16).float() @ v).transpose(1, 2).reshape(b, t, heads * head_dim).to(torch.bfloat16)
    if want_importance and query_mean:           # teachers without a CLS token: the map averaged over heads and queries
        return out, p.mean(dim=(1, 2))
    imp = cls_importance(qkv, heads, head_dim, scale) if want_importance else None
    if want_lse:
        return out, imp, torch.logsumexp(logits, dim=-1)
    return out, imp


def attention_bwd_supported(t, hd):
    return hd == 64 and 1 <= t <= 224


def attention_bwd(qkv, out, dout, lse, heads, head_dim, scale):
    b, t, _ = qkv.shape
    with torch.enable_grad():
        x = qkv.detach().float().reshape(b, t, 3, heads, head_dim).requires_grad_(True)
        q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
        p = ((q @ k.transpose(-1, -2)) * scale).softmax(dim=-1)
        o = (p @ v).transpose(1, 2).reshape(b, t, heads * head_dim)
        o.backward(dout.float())
    return x.grad.reshape(b, t, -1).to(torch.bfloat16)


def layernorm_supported(d):
    return d % 8 == 0 and 8 <= d <= 2048


def layernorm_fwd(x, gamma, beta, eps):
    xf = x.float()
    mean = xf.mean(-1)
    var = ((xf - mean.unsqueeze(-1)) ** 2).mean(-1)
    rstd = torch.rsqrt(var + eps)
    y = ((xf - mean.unsqueeze(-1)) * rstd.unsqueeze(-1) * gamma + beta).to(torch.bfloat16)
    return y, mean.reshape(-1), rstd.reshape(-1)


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, row_scale=None, want_branch=False):
    d = x.shape[-1]
    xf, dyf = x.float().reshape(-1, d), dy.float().reshape(-1, d)
    xh = (xf - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
    g = dyf * gamma
    dx = rstd.unsqueeze(-1) * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if dgamma is not None:
        dgamma.add_((dyf * xh).sum(0))
        dbeta.add_(dyf.sum(0))
    if dres is not None:
        dx = dx + dres.float().reshape(-1, d)
    out = dx.to(torch.bfloat16).reshape(x.shape)
    if not want_branch:
        return out
    sc = 1.0 if row_scale is None else row_scale.float().view(-1, *([1] * (x.dim() - 1)))
    return out, (dx.reshape(x.shape) * sc).to(torch.bfloat16)
