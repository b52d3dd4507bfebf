"""Kernel-level parity of the C-ABI HIP kernels against fp64 torch references
(GPU box only).  Tolerances are stated per test."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    import basd_amd._native as native
    assert torch.cuda.is_available(), "needs an MI355X"
    native.lib()
    return native


@pytest.fixture(autouse=True)
def _clean_status_word(nat):
    """the device health word is sticky by design: start every test from a clean one"""
    nat.status_word("cuda").zero_()
    yield


def _colmajor(a: torch.Tensor, ld: int) -> torch.Tensor:
    """[batch, m, n] row-major matrices -> [batch, n, ld] column-major with zero padded rows."""
    b, m, n = a.shape
    w = torch.zeros(b, n, ld, dtype=torch.float32, device=a.device)
    w[:, :, :m] = a.transpose(1, 2)
    return w.contiguous()


@pytest.mark.parametrize("m,n,batch", [(40, 32, 5), (33, 17, 3), (192, 192, 4), (196, 192, 2), (64, 7, 2),
                                       # the hex-block kernel (129 .. 192 columns, up to 192 rows; 32-lane slots for 257 .. 384
                                       # rows): ragged last blocks (n % 6, n % 12), a phantom block, one / two / three row chunks,
                                       # two matrices per CU (batch > 256)
                                       (192, 130, 3), (150, 150, 2), (176, 191, 2), (64, 180, 2), (120, 133, 300),
                                       (384, 192, 3), (300, 170, 2)])
def test_jacobi_singular_values_and_invariants(nat, m, n, batch):
    g = torch.Generator().manual_seed(m * 1000 + n)
    a = torch.randn(batch, m, n, generator=g)
    # graded columns: singular values span ~5 decades
    a = a * torch.logspace(0, -5, n).view(1, 1, n)
    ad = a.double()
    ld = nat.jacobi_ld(m)
    w = _colmajor(a.cuda(), ld)
    sigma, sweeps = nat.jacobi_svd(w, m)
    torch.cuda.synchronize()
    ref = torch.linalg.svdvals(ad)
    if m < n:
        ref = torch.cat([ref, torch.zeros(batch, n - m, dtype=ref.dtype)], dim=1)
    sig = sigma.cpu().double()
    # column-graded input: one-sided Jacobi keeps RELATIVE accuracy of every singular value
    err = float((sig[:, :min(m, n)] / ref[:, :min(m, n)] - 1).abs().max())
    print(f"jacobi {m}x{n}: max rel sigma error {err:.2e}, sweeps {sweeps.tolist()}")
    # relative accuracy is eps * kappa(B) for A = B D; a square Gaussian B (192x192) has kappa ~ 1e3
    if m < n:                                                 # wide: n - m exact zeros, the rest to relative accuracy
        assert float(sig[:, m:].abs().max()) < 1e-6 * float(ref.max())
        err = float((sig[:, :m] / ref[:, :m] - 1).abs().max())
    assert err < (5e-5 if m <= n else 5e-6)       # square Gaussian B: kappa ~ 1e3 (192 x 192: 1e-5 .. 3e-5 by seed)
    wf = w.cpu().double()[:, :, :m].transpose(1, 2)          # [batch, m, n] = U Sigma
    gram = wf.transpose(1, 2) @ wf
    off = gram - torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2))
    scale = torch.sqrt(torch.diagonal(gram, dim1=1, dim2=2))
    if m < n:                                                 # the null columns have no direction
        live = (scale > 1e-6 * scale.amax(dim=1, keepdim=True)).double()
        off = off * live.unsqueeze(1) * live.unsqueeze(2)
        scale = scale.clamp_min(1e-30)
    assert (off / (scale.unsqueeze(1) * scale.unsqueeze(2))).abs().max() < 5e-6
    # right rotations leave A A^T unchanged
    aat = ad @ ad.transpose(1, 2)
    assert torch.allclose(wf @ wf.transpose(1, 2), aat, rtol=0, atol=3e-6 * float(aat.abs().max()))
    assert int(sweeps.max()) < 40
    # sorted descending
    assert bool((sig[:, :-1] >= sig[:, 1:]).all())


def test_jacobi_stacked_gives_consistent_right_vectors(nat):
    k = 24
    g = torch.Generator().manual_seed(3)
    q1, _ = torch.linalg.qr(torch.randn(4, 64, k, generator=g))
    q2, _ = torch.linalg.qr(torch.randn(4, 64, k, generator=g))
    a = q1.transpose(1, 2) @ q2                      # cosines of principal angles
    stacked = torch.cat([a, torch.eye(k).expand(4, k, k)], dim=1)   # [4, 2k, k]
    ld = nat.jacobi_ld(2 * k)
    w = _colmajor(stacked.cuda(), ld)
    sigma, _ = nat.jacobi_svd(w, 2 * k, norm_rows=k)
    out = w.cpu().double()[:, :, :2 * k].transpose(1, 2)
    us, v = out[:, :k], out[:, k:]
    sig = sigma.cpu().double()
    assert torch.allclose(sig, torch.linalg.svdvals(a.double()), rtol=1e-5, atol=1e-7)
    # A V = U Sigma with V orthogonal
    assert torch.allclose(a.double() @ v, us, atol=2e-6)
    assert torch.allclose(v.transpose(1, 2) @ v, torch.eye(k, dtype=torch.float64).expand(4, k, k), atol=2e-6)


@pytest.mark.parametrize("n,rank", [(32, 32), (32, 11), (192, 192), (192, 63), (196, 195), (196, 60), (194, 194)])
def test_pchol_then_jacobi_is_an_eigensolver(nat, n, rank):
    g = torch.Generator().manual_seed(n + rank)
    x = torch.randn(3, 4 * n, rank, dtype=torch.float64, generator=g) * torch.logspace(0, -3, rank, dtype=torch.float64)
    mix = torch.randn(3, rank, n, dtype=torch.float64, generator=g)
    z = x @ mix
    a = z.transpose(1, 2) @ z
    w0, lwork, piv, rk = nat.pchol(a.cuda())
    torch.cuda.synchronize()
    assert rk.tolist() == [rank] * 3
    lw = lwork.cpu().transpose(1, 2)                 # [batch, n(rows), n(steps)]
    assert torch.allclose(lw @ lw.transpose(1, 2), a, rtol=0, atol=1e-10 * float(a.abs().max()))
    for b in range(3):
        assert sorted(piv[b].tolist()) == list(range(n))
    sigma, _ = nat.jacobi_svd(w0, n)
    ev = sigma.cpu().double() ** 2
    ref = torch.linalg.eigvalsh(a).flip(-1)
    # eigenvalues below 1e-9 * max are beyond what an fp64 Gram matrix resolves to 1e-5
    ok = ref[:, :rank] > 1e-9 * ref[:, :1]
    err = float(((ev[:, :rank] / ref[:, :rank] - 1).abs() * ok).max())
    print(f"pchol+jacobi n={n} rank={rank}: max rel eigenvalue error {err:.2e}")
    assert err < 1e-5
    tail = float(ev[:, rank:].abs().max()) if rank < n else 0.0
    assert tail <= 1e-12 * float(ref.max())
    # eigenvectors: A u = lambda u
    ld = w0.shape[2]
    u = (w0.cpu().double()[:, :, :n] / sigma.cpu().double().clamp(min=1e-300).unsqueeze(-1))[:, :rank]
    resid = u @ a - ev[:, :rank].unsqueeze(-1) * u
    assert float(resid.abs().max()) < 2e-5 * float(ref.max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d_in,d_out", [(240, 64, 32), (4096, 384, 192), (1000, 768, 192)])
def test_token_gram(nat, rows, d_in, d_out, dtype):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, d_in, generator=g).to(dtype)
    p = torch.linalg.qr(torch.randn(d_in, d_out, generator=g))[0].T.contiguous()
    gram, colsum = nat.token_gram(x.cuda(), p.cuda())
    z = x.double() @ p.double().T
    ref = z.T @ z
    assert torch.allclose(gram.cpu(), ref, rtol=0, atol=3e-6 * float(ref.abs().max()))
    assert torch.allclose(colsum.cpu(), z.sum(0), rtol=0, atol=3e-6 * float(z.abs().sum(0).max()))
    assert torch.allclose(gram, gram.T, rtol=0, atol=1e-12 * float(ref.abs().max()))


@pytest.mark.parametrize("rows,d_in,d_out", [(50176, 768, 192), (3000, 384, 192), (1000, 64, 32)])
def test_token_gram_centred_statistics_survive_a_dominant_mean(nat, rows, d_in, d_out):
    """bf16 tokens whose column means are 30 x their spread (a structured data set; block outputs of a ViT carry such
    offsets): the CENTRED Gram the selector forms as unc - s s^T / m (layer_selector.py:35) must keep fp32-level
    accuracy relative to ITSELF, not relative to the 900 x larger uncentred Gram -- the kernel centres every 128-row
    tile on its own mean before the fp32 MFMA and carries the means in fp64."""
    g = torch.Generator().manual_seed(rows + d_out)
    x = (torch.randn(rows, d_in, generator=g) + 30.0 * torch.randn(1, d_in, generator=g)).bfloat16()
    p = torch.linalg.qr(torch.randn(d_in, d_out, generator=g))[0].T.contiguous()
    gram, colsum = nat.token_gram(x.cuda(), p.cuda())
    z = x.double() @ p.double().T
    ref_unc, ref_sum = z.T @ z, z.sum(0)
    ref_cen = ref_unc - torch.outer(ref_sum, ref_sum) / rows
    cen = gram.cpu() - torch.outer(colsum.cpu(), colsum.cpu()) / rows
    assert torch.allclose(gram.cpu(), ref_unc, rtol=0, atol=1e-6 * float(ref_unc.abs().max()))
    assert torch.allclose(colsum.cpu(), ref_sum, rtol=0, atol=1e-6 * float(ref_sum.abs().max()))
    assert float(ref_unc.abs().max()) > 100 * float(ref_cen.abs().max())          # the regime this test is about
    # what is left is the fp32 accumulation of z = x P^T itself (values ~30, 768 terms: ~5e-5 absolute per entry, summed
    # over 50 176 rows against centred partners of unit spread); an fp32 Gram of the uncentred z would be off by 1e-3
    err = float((cen - ref_cen).abs().max()) / float(ref_cen.abs().max())
    assert err < 2e-5, err
    ev, ev_ref = torch.linalg.eigvalsh(cen), torch.linalg.eigvalsh(ref_cen)
    assert float(ev.min()) > 0.5 * float(ev_ref.min()) > 0
    assert torch.allclose(ev, ev_ref, rtol=0, atol=2e-5 * float(ev_ref.max()))


@pytest.mark.parametrize("rows,d_in,d_out", [(25216, 1024, 384), (2000, 128, 8), (4133, 192, 264), (65792, 1280, 768)])
def test_gemm_bf16x3_f32(nat, rows, d_in, d_out):
    """x P^T with x bf16, P fp32 as one bf16-MFMA GEMM over the three bf16 splits of P: fp32-GEMM accuracy (the splits
    carry P to 2^-24, the accumulation is fp32), ragged last row block, column tiles wider than the output"""
    assert nat.gemm_bf16x3_f32_supported(rows, d_out, d_in)
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, d_in, generator=g) * (1 + torch.rand(1, d_in, generator=g))).bfloat16()
    p = torch.linalg.qr(torch.randn(d_in, max(d_out, 1), generator=g))[0].T.contiguous() if d_out <= d_in else \
        torch.randn(d_out, d_in, generator=g) / d_in ** 0.5
    guard = torch.full((rows * d_out + 4096,), 7.0, device="cuda")
    z = nat.gemm_bf16x3_f32(x.cuda(), p.cuda())
    assert z.shape == (rows, d_out) and z.dtype == torch.float32
    ref = (x.cuda().double() @ p.cuda().double().T)
    scale = float(ref.abs().max())
    err = float((z.double() - ref).abs().max())
    assert err < 4e-6 * scale, (err, scale)
    assert float(guard.min()) == 7.0 and float(guard.max()) == 7.0


@pytest.mark.parametrize("rows,d_in,d_out", [(25216, 1024, 384), (65792, 1280, 768), (3000, 1024, 384)])
def test_token_gram_wide_at_training_sizes(nat, rows, d_in, d_out):
    """student widths 384 / 768 (BASELINE c4 / c5): projection on the bf16x3 GEMM (first two cases) or the fp64-MFMA
    batched GEMM (small M), Gram on the fp64 matrix cores"""
    g = torch.Generator().manual_seed(rows + 1)
    x = (torch.randn(rows, d_in, generator=g) + 2.0 * torch.randn(1, d_in, generator=g)).bfloat16().cuda()
    p = torch.linalg.qr(torch.randn(d_in, d_out, generator=g))[0].T.contiguous().cuda()
    gram, colsum = nat.token_gram(x, p, mirror=False)
    z = x.double() @ p.double().T
    ref, ref_sum = z.T @ z, z.sum(0)
    low = torch.tril(torch.ones(d_out, d_out, dtype=torch.bool, device="cuda"))
    assert torch.isfinite(gram[low]).all()
    assert float((gram - ref)[low].abs().max()) < 3e-6 * float(ref.abs().max())
    assert torch.allclose(colsum, ref_sum, rtol=0, atol=3e-6 * float(z.abs().sum(0).max()))


def test_mp_rank_matches_oracle(nat):
    from oracle import basd_oracle as O
    g = torch.Generator().manual_seed(0)
    mats, want = [], []
    for r in (3, 9, 0, 31):
        z = torch.randn(400, 32, generator=g)
        if r:
            z = z + 3.0 * (torch.randn(400, r, generator=g) @ torch.randn(r, 32, generator=g)) / r ** 0.5
        mats.append(torch.linalg.eigvalsh((z.T @ z).double()).float())
        want.append(min(O.mp_rank(z), 31))
    ranks = nat.mp_rank(torch.stack(mats).cuda(), 400, 32, 31)
    assert ranks.tolist() == want
    # M < D branch of the reference (layer_selector.py:14-15)
    z = torch.randn(20, 32, generator=g) + 3.0 * torch.randn(20, 2, generator=g) @ torch.randn(2, 32, generator=g)
    ev = torch.linalg.eigvalsh((z.T @ z).double()).float().unsqueeze(0)
    assert nat.mp_rank(ev.cuda(), 20, 32, 31).tolist() == [min(O.mp_rank(z), 31)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mix_and_dots(nat, dtype):
    g = torch.Generator().manual_seed(1)
    L, E = 5, 4
    layers = [torch.randn(3, 40, 64, generator=g).to(dtype) for _ in range(L)]
    w = torch.softmax(torch.randn(E, L, generator=g), dim=1)
    out = nat.mix_tokens([t.cuda() for t in layers], w.cuda())
    ref = torch.einsum("el,lbnd->ebnd", w.double(), torch.stack(layers).double())
    assert torch.allclose(out.cpu().double(), ref, rtol=1e-6, atol=1e-6)
    gr = torch.randn(E, 3, 40, 64, generator=g)
    dots = nat.mix_grad_dots([t.cuda() for t in layers], gr.cuda())
    refd = torch.einsum("ebnd,lbnd->el", gr.double(), torch.stack(layers).double())
    assert torch.allclose(dots.cpu(), refd, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("n_t", [40, 49, 9])
def test_procrustes_prep(nat, n_t):
    from oracle import basd_oracle as O
    g = torch.Generator().manual_seed(n_t)
    B, n_s, d_s, d_t = 3, 40, 32, 64
    s = torch.randn(B, n_s, d_s, generator=g)
    t = torch.randn(B, n_t, d_t, generator=g)
    imp = torch.rand(B, n_t, generator=g) + 0.1
    s_w, t_w, a, tr = nat.procrustes_prep(s.cuda(), t.cuda(), imp.cuda())
    sd, td, impd = s.double(), O.resample_linear(t.double(), n_s), imp.double()
    if n_t != n_s:
        impd = O.resample_linear(impd.unsqueeze(-1), n_s).squeeze(-1)
    w = impd / impd.sum(-1, keepdim=True)
    wc = w.unsqueeze(-1)
    s_ref = wc.sqrt() * (sd - (wc * sd).sum(1, keepdim=True))
    t_ref = wc.sqrt() * (td - (wc * td).sum(1, keepdim=True))
    assert torch.allclose(a.cpu().double(), w, rtol=1e-5, atol=1e-8)
    assert torch.allclose(s_w.cpu().double(), s_ref, atol=2e-6)
    assert torch.allclose(t_w.cpu().double(), t_ref, atol=2e-6)
    assert torch.allclose(tr[:, 0].cpu().double(), (s_ref ** 2).sum((1, 2)), rtol=1e-5)
    assert torch.allclose(tr[:, 1].cpu().double(), (t_ref ** 2).sum((1, 2)), rtol=1e-5)


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("dt_a,dt_b,dt_c", [(torch.float32, torch.float32, torch.float64),
                                            (torch.float64, torch.float64, torch.float64),
                                            (torch.float64, torch.float64, torch.float32)])
def test_bgemm_f64(nat, ta, tb, dt_a, dt_b, dt_c):
    g = torch.Generator().manual_seed(7)
    batch, M, N, K = 3, 70, 130, 53        # ragged vs the 64x64x16 tiling
    a = torch.randn(batch, *((K, M) if ta else (M, K)), generator=g, dtype=torch.float64).to(dt_a)
    b = torch.randn(batch, *((N, K) if tb else (K, N)), generator=g, dtype=torch.float64).to(dt_b)
    c = nat.bgemm_f64(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb, out_dtype=dt_c)
    ref = (a.double().transpose(1, 2) if ta else a.double()) @ (b.double().transpose(1, 2) if tb else b.double())
    tol = 1e-12 if dt_c == torch.float64 else 1e-6
    assert c.dtype == dt_c
    assert torch.allclose(c.cpu().double(), ref, rtol=0, atol=tol * float(ref.abs().max()) * K ** 0.5)


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("dt_a,dt_b,dt_c", [(torch.float32, torch.float32, torch.float64),
                                            (torch.float32, torch.float32, torch.float32),
                                            (torch.float64, torch.float64, torch.float64),
                                            (torch.float64, torch.float64, torch.float32),
                                            (torch.float64, torch.float32, torch.float64),
                                            (torch.float64, torch.float32, torch.float32),
                                            (torch.float32, torch.float64, torch.float64),
                                            (torch.float32, torch.float64, torch.float32)])
def test_bgemm_f64_aligned_fast_kernel(nat, ta, tb, dt_a, dt_b, dt_c):
    """64-aligned shapes dispatch bgemm_f64_fast_kernel (register prefetch, double-buffered LDS): the kernel the
    benchmark's Procrustes chain runs -- every dtype combination and transpose against fp64 torch"""
    g = torch.Generator().manual_seed(17)
    batch, M, N, K = 5, 192, 256, 196
    a = torch.randn(batch, *((K, M) if ta else (M, K)), generator=g, dtype=torch.float64).to(dt_a)
    b = torch.randn(batch, *((N, K) if tb else (K, N)), generator=g, dtype=torch.float64).to(dt_b)
    c = nat.bgemm_f64(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb, out_dtype=dt_c)
    ref = (a.double().transpose(1, 2) if ta else a.double()) @ (b.double().transpose(1, 2) if tb else b.double())
    tol = 1e-12 if dt_c == torch.float64 else 1e-6
    assert c.dtype == dt_c
    assert torch.allclose(c.cpu().double(), ref, rtol=0, atol=tol * float(ref.abs().max()) * K ** 0.5)


def test_bgemm_f64_aligned_symmetric_and_split_k(nat):
    """symmetric tile skipping on aligned shapes + the split-K Gram of the wide token statistics (_token_gram_wide)"""
    g = torch.Generator().manual_seed(19)
    x = torch.randn(4, 192, 320, generator=g, dtype=torch.float64)
    c = nat.bgemm_f64(x.cuda(), x.cuda(), trans_b=True, symmetric=True).cpu()
    ref = x @ x.transpose(1, 2)
    assert torch.allclose(c, ref, rtol=0, atol=1e-12 * float(ref.abs().max()))
    assert torch.equal(c, c.transpose(1, 2))
    tok = torch.randn(2048, 256, generator=g)
    proj = torch.randn(384, 256, generator=g) / 16
    gram, colsum = nat.token_gram(tok.cuda(), proj.cuda())
    z = (tok @ proj.t()).double()
    assert torch.allclose(gram.cpu(), z.t() @ z, rtol=0, atol=2e-6 * float((z.t() @ z).abs().max()))
    assert torch.allclose(colsum.cpu(), z.sum(0), rtol=0, atol=1e-4 * float(z.sum(0).abs().max()))


@pytest.mark.parametrize("n,decay", [(384, 0.985), (768, 0.99), (384, 0.97)])
def test_blocked_eigensolver_graded_spectra(nat, n, decay):
    """psd_eig for n > 192 (blocked Cholesky + block Jacobi with right-vector pair rotations) on random-basis graded
    spectra up to condition 1e5 in sigma: eigenvalues RELATIVE 2e-5 over the whole spectrum (the graded accuracy of
    one-sided Jacobi), orthonormal vectors, small residual.  A left-vector (U) pair rotation fails this at 2.5e-4 .. divergence."""
    from basd_amd.losses import functional as BF, _ops
    _ops.set_ops(None)
    g = torch.Generator().manual_seed(n)
    q, _ = torch.linalg.qr(torch.randn(2, n, n, generator=g, dtype=torch.float64))
    sig = decay ** torch.arange(n, dtype=torch.float64)
    a = ((q * sig ** 2) @ q.transpose(1, 2)).cuda()
    s, u, _ = BF.psd_eig(a)
    nat.check_status()
    assert float(((s.double().cpu() - sig) / sig).abs().max()) < 2e-5
    ud = u.double()
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    assert float((ud @ ud.transpose(1, 2) - eye).abs().max()) < 5e-6
    res = ud @ a - (s.double() ** 2).unsqueeze(-1) * ud
    assert float(res.norm(dim=-1).max()) < 2e-5           # lambda_max = 1; eigenvalues themselves are good to 2e-5 relative


def test_blocked_eigensolver_flags_missing_convergence(nat, monkeypatch):
    """with too few outer sweeps the device-side orthogonality check raises the NONCONVERGED bit"""
    from basd_amd.losses import functional as BF, _ops
    _ops.set_ops(None)
    g = torch.Generator().manual_seed(1)
    q, _ = torch.linalg.qr(torch.randn(1, 384, 384, generator=g, dtype=torch.float64))
    a = ((q * (0.985 ** torch.arange(384, dtype=torch.float64)) ** 2) @ q.transpose(1, 2)).cuda()
    monkeypatch.setattr(BF, "WIDE_DIRECT_SWEEPS", 2)
    BF.psd_eig(a)
    with pytest.raises(torch.linalg.LinAlgError, match="without converging"):
        nat.check_status()
    monkeypatch.setattr(BF, "WIDE_DIRECT_SWEEPS", 8)
    BF.psd_eig(a)
    nat.check_status()


@pytest.mark.parametrize("n,ranks", [(768, (120, 350, 384, 385, 500, 768)), (384, (60, 160, 192, 193, 300, 384))])
def test_blocked_eigensolver_rank_masked_matrices_take_the_right_path(nat, n, ranks):
    """zero_blocks=True: A is non-zero in its leading k x k corner only (the principal-angle Gram matrices of the wide
    students).  Matrices whose factor fits the small corner (384 of 768, 192 of 384) take the small path, the others the
    general tournament, selected per matrix ON THE DEVICE through the kernels' problem masks: one batch with ranks on
    both sides of the boundary, eigenvalues / vectors of every matrix against eigh, and against the unmasked solver"""
    from basd_amd.losses import functional as BF, _ops
    _ops.set_ops(None)
    g = torch.Generator().manual_seed(n)
    a = torch.zeros(len(ranks), n, n, dtype=torch.float64)
    for i, k in enumerate(ranks):
        c = torch.linalg.qr(torch.randn(k, k, generator=g, dtype=torch.float64))[0]
        cos = torch.cos(torch.rand(k, generator=g, dtype=torch.float64) * 1.5)          # cosines of principal angles
        a[i, :k, :k] = (c * cos ** 2) @ c.t()
    a = a.cuda()
    s, u, _ = BF.psd_eig(a, zero_blocks=True)
    nat.check_status()
    s0, u0, _ = BF.psd_eig(a)                                                            # every matrix through the general path
    nat.check_status()
    ref = torch.linalg.eigvalsh(a).flip(-1).clamp_min(0).sqrt()
    for i, k in enumerate(ranks):
        top = float(ref[i, 0])
        assert torch.allclose(s[i, :k].double(), ref[i, :k], rtol=0, atol=1e-5 * top), (k, float((s[i, :k].double() - ref[i, :k]).abs().max()))
        assert float(s[i, k:].abs().max()) <= 1e-6 * top if k < n else True
        assert torch.allclose(s[i], s0[i], rtol=0, atol=1e-5 * top)
        ud = u[i, :k].double()
        assert float((ud @ ud.t() - torch.eye(k, dtype=torch.float64, device="cuda")).abs().max()) < 1e-5
        res = ud @ a[i] - (s[i, :k].double() ** 2).unsqueeze(-1) * ud
        assert float(res.norm(dim=-1).max()) < 3e-5 * top ** 2
        assert float(u[i, :, k:].abs().max()) == 0.0 if k < n else True                  # eigenvectors stay in the corner


def test_blocked_eigensolver_rank_deficient(nat):
    """rank 150 in 384 dimensions: the blocked Cholesky stops at the rank, null directions come back as zero rows"""
    from basd_amd.losses import functional as BF, _ops
    _ops.set_ops(None)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 600, 150, generator=g, dtype=torch.float64) @ torch.randn(2, 150, 384, generator=g, dtype=torch.float64)
    a = (z.transpose(1, 2) @ z).cuda()
    s, u, _ = BF.psd_eig(a)
    ref = torch.linalg.svdvals(z)
    assert torch.allclose(s[:, :150].double().cpu(), ref[:, :150], rtol=1e-5, atol=0)
    assert float(s[:, 150:].abs().max()) <= 1e-6 * float(ref.max())
    live = u.abs().amax(-1) > 0
    assert live.sum(1).tolist() == [150, 150]
    nat.check_status()                                  # zero columns must not trip the orthogonality check


def test_status_word_flags_nonfinite_nonconverged_and_rank0(nat):
    """data-dependent failures are OR-ed into the device health word and surface as a LinAlgError on check"""
    nat.check_status()                                        # clean slate
    g = torch.Generator().manual_seed(5)
    a = torch.randn(4, 48, 32, generator=g).cuda()
    w = _colmajor(a, nat.jacobi_ld(48))
    nat.jacobi_svd(w.clone(), 48)
    nat.check_status()                                        # healthy input: no flag
    bad = w.clone()
    bad[1, 3, 5] = float("nan")
    nat.jacobi_svd(bad, 48)
    with pytest.raises(torch.linalg.LinAlgError, match="non-finite"):
        nat.check_status()
    nat.check_status()                                        # cleared by the failed check
    nat.jacobi_svd(w.clone(), 48, max_sweeps=1)
    with pytest.raises(torch.linalg.LinAlgError, match="without converging"):
        nat.check_status()
    ev = torch.ones(3, 64, device="cuda")                     # flat spectrum: nothing above the MP edge
    assert nat.mp_rank(ev, 4096, 64, 63).tolist() == [0, 0, 0]
    with pytest.raises(torch.linalg.LinAlgError, match="rank 0"):
        nat.check_status()


@pytest.mark.parametrize("n,rank", [(32, 32), (32, 20), (192, 192), (192, 100), (196, 196), (196, 195), (196, 193), (196, 120),
                                    (208, 208), (200, 197)])
def test_trinv_matches_triangular_solve(nat, n, rank):
    """n <= 192: blocked kernel; 193 .. 208 (the 196-token matrices of the wide students): blocked leading 192 rows +
    border rows"""
    g = torch.Generator().manual_seed(n * 7 + rank)
    z = torch.randn(3, 3 * n, rank, dtype=torch.float64, generator=g) @ torch.randn(3, rank, n, dtype=torch.float64, generator=g)
    a = z.transpose(1, 2) @ z
    w0, lwork, piv, rk = nat.pchol(a.cuda())
    out = nat.trinv(lwork, piv, rk).cpu()
    assert rk.tolist() == [rank] * 3
    for b in range(3):
        pv = piv[b].cpu().long()
        lp = lwork[b].cpu().t()[pv][:rank, :rank]              # live block, pivot order
        x = out[b][:, pv]                                       # undo the column scatter
        assert torch.allclose(x[:rank, :rank] @ lp, torch.eye(rank, dtype=torch.float64), atol=1e-9)
        assert float(x[rank:].abs().max()) == 0.0 if rank < n else True
        assert float(torch.triu(x[:rank, :rank], 1).abs().max()) == 0.0


def test_bgemm_f64_symmetric_mode(nat):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 150, 70, generator=g, dtype=torch.float64)
    c = nat.bgemm_f64(x.cuda(), x.cuda(), trans_b=True, symmetric=True).cpu()
    ref = x @ x.transpose(1, 2)
    assert torch.allclose(c, ref, rtol=0, atol=1e-12 * float(ref.abs().max()))
    assert torch.equal(c, c.transpose(1, 2))


@pytest.mark.parametrize("batch,n,k,dt", [(9, 196, 768, torch.float32), (3, 192, 320, torch.float64),
                                          (10, 200, 100, torch.float32), (2, 64, 16, torch.float32),
                                          (3, 132, 52, torch.float32), (17, 256, 64, torch.float64)])
def test_bgemm_f64_balanced_gram(nat, batch, n, k, dt):
    """C = X X^T of ONE device operand (same pointer on both sides) into fp64 dispatches gram_rows_f64_kernel: full
    off-diagonal tiles, diagonal tiles with the ten sub-tiles dealt over the waves, the n % 64 <= 8 trailing rows on
    the fp64 VALU -- the Gt = t_w t_w^T of the feature-side Procrustes chain is the (196, 768) case"""
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(batch, n, k, generator=g, dtype=torch.float64).to(dt)
    xd = x.cuda()
    c = nat.bgemm_f64(xd, xd, trans_b=True, symmetric=True).cpu()
    ref = x.double() @ x.double().transpose(1, 2)
    assert c.dtype == torch.float64
    assert torch.allclose(c, ref, rtol=0, atol=1e-12 * float(ref.abs().max()) * k ** 0.5)
    assert torch.equal(c, c.transpose(1, 2))


@pytest.mark.parametrize("n,ks", [(64, [5, 17, 40, 64]), (192, [7, 84, 131, 192, 2, 97])])
def test_jacobi_active_block_matches_full_run(nat, n, ks):
    """rank-masked input: sweeping only the leading active block gives the same singular values (n = 192: the hex-block
    kernel's active-columns / active-rows modes, the principal-angle launch of a c2 step)."""
    g = torch.Generator().manual_seed(5)
    a = torch.zeros(len(ks), n, n)
    for b, k in enumerate(ks):
        a[b, :k, :k] = torch.randn(k, k, generator=g)
    ld = nat.jacobi_ld(n)
    w1 = _colmajor(a.cuda(), ld)
    w2 = w1.clone()
    s1, _ = nat.jacobi_svd(w1, n)
    s2, sw = nat.jacobi_svd(w2, n, active=torch.tensor(ks, dtype=torch.int32, device="cuda"), active_rows=True)
    ref = torch.linalg.svdvals(a.double())
    assert torch.allclose(s2.cpu().double(), ref, rtol=2e-5, atol=1e-6)
    assert torch.allclose(s1, s2, rtol=2e-5, atol=1e-6)
    for b, k in enumerate(ks):            # columns beyond the block stay exactly zero
        assert float(s2[b, k:].abs().max()) == 0.0 if k < n else True
    # the mask mode: negative entries skip a matrix altogether (outputs untouched), the others are solved completely
    w3 = _colmajor(a.cuda(), ld)
    keep = w3.clone()
    mask = torch.tensor([(-1 if b % 2 else n) for b in range(len(ks))], dtype=torch.int32, device="cuda")
    s3, sw3 = nat.jacobi_svd(w3, n, active=mask, active_rows=2)
    for b in range(len(ks)):
        if b % 2:
            assert torch.equal(w3[b], keep[b]) and int(sw3[b]) == 0
        else:
            assert torch.allclose(s3[b], s1[b], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,K,gelu,bias", [(50432, 3072, 768, True, True), (50432, 768, 3072, False, True),
                                            (50432, 576, 192, False, True), (50432, 192, 768, False, False),
                                            (25216, 384, 1536, False, True), (1000, 768, 768, False, True),
                                            (300, 128, 64, True, True), (517, 1280, 320, True, False)])
def test_gemm_bf16_with_fused_epilogue(nat, M, N, K, gelu, bias):
    """basd_gemm_bf16 (ring kernel for 256-wide tiles, two-stage kernel for 192 / 128; ragged M; bias / exact-erf GELU
    epilogue) against the fp32 product of the SAME bf16 inputs: the only difference allowed is the final bf16
    rounding (2^-8 relative) plus the fp32 accumulation order"""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda() if bias else None
    y = nat.gemm_bf16(x, w, b, gelu=gelu)
    ref = x.float() @ w.float().t()
    if b is not None:
        ref = ref + b.float()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    assert y.dtype == torch.bfloat16 and y.shape == (M, N)
    err = (y.float() - ref).abs()
    assert float((err / (ref.abs() + 1e-2 * float(ref.abs().max()))).max()) < 1.2e-2
    assert float(err.max()) <= 6e-3 * float(ref.abs().max())
    # batched input view [B, T, K] -> [B, T, N]
    if M % 4 == 0:
        y3 = nat.gemm_bf16(x.view(4, M // 4, K), w, b, gelu=gelu)
        assert y3.shape == (4, M // 4, N) and torch.equal(y3.reshape(M, N), y)


@pytest.mark.parametrize("M,hidden,d", [(50432, 768, 192), (25216, 1536, 384), (1000, 3072, 768), (300, 128, 64),
                                        (517, 384, 192)])
def test_gemm_bf16_gelu_training_pair(nat, M, hidden, d):
    """fc1 + GELU in one launch (pre-activation saved) and fc2's input gradient times gelu'(pre) in one launch, against
    the unfused fp32 arithmetic on the same bf16 inputs (ring kernel at hidden % 256 == 0, two-stage kernel otherwise;
    ragged M)"""
    g = torch.Generator().manual_seed(M + hidden + d)
    x = torch.randn(M, d, generator=g).bfloat16().cuda()
    w1 = (torch.randn(hidden, d, generator=g) / d ** 0.5).bfloat16().cuda()
    b1 = torch.randn(hidden, generator=g).bfloat16().cuda()
    pre, act = nat.gemm_gelu_fwd(x, w1, b1)
    ref_pre = x.float() @ w1.float().t() + b1.float()
    assert pre.shape == act.shape == (M, hidden) and pre.dtype == act.dtype == torch.bfloat16
    assert float((pre.float() - ref_pre).abs().max()) <= 6e-3 * float(ref_pre.abs().max())
    # the activation is the GELU of the ROUNDED pre-activation (what nn.GELU sees behind a bf16 nn.Linear)
    ref_act = torch.nn.functional.gelu(pre.float())
    assert float((act.float() - ref_act).abs().max()) <= 4.1e-3 * float(ref_act.abs().max())
    assert torch.equal(pre, nat.gemm_bf16(x, w1, b1))                 # same accumulation, same rounding
    # backward: dY [M, d] through fc2 (weight [d, hidden]) and the GELU
    dy = (torch.randn(M, d, generator=g) * 0.1).bfloat16().cuda()
    w2 = (torch.randn(d, hidden, generator=g) / hidden ** 0.5).bfloat16().cuda()
    dpre = nat.gemm_gelu_bwd(dy, w2.t().contiguous(), pre)
    p = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(p).backward(dy.float() @ w2.float())
    ref = p.grad
    err = (dpre.float() - ref).abs()
    assert dpre.shape == (M, hidden)
    assert float(err.max()) <= 6e-3 * float(ref.abs().max())
    assert float((err / (ref.abs() + 1e-2 * float(ref.abs().max()))).max()) < 1.2e-2


@pytest.mark.parametrize("B,C,soft,smoothing,with_geo", [(256, 1000, True, 0.1, True), (256, 1000, False, 0.1, True),
                                                          (7, 10, True, 0.0, True), (33, 100, False, 0.05, False)])
def test_ce_uwso_matches_torch(nat, B, C, soft, smoothing, with_geo):
    """basd_ce_uwso: CE(label smoothing; soft or hard targets) + UW-SO weights + d total / d logits in two launches,
    against nn.CrossEntropyLoss and the reference's weighting arithmetic through autograd"""
    g = torch.Generator().manual_seed(B + C)
    logits = (torch.randn(B, C, generator=g) * 3).cuda()
    if soft:
        t = torch.rand(B, C, generator=g)
        targets = (t / t.sum(-1, keepdim=True)).cuda()
    else:
        targets = torch.randint(C, (B,), generator=g).cuda()
    geo = torch.tensor(3.7, device="cuda") if with_geo else None
    out4, dl = nat.ce_uwso(logits, targets, smoothing, geo)
    z = logits.double().requires_grad_(True)
    ce = torch.nn.CrossEntropyLoss(label_smoothing=smoothing)(z, targets.double() if soft else targets)
    if with_geo:
        inv = torch.stack([1.0 / ce.detach(), 1.0 / geo.double()])
        w = inv / inv.sum()
        total = w[0] * ce + w[1] * geo.double()
    else:
        w = torch.tensor([1.0, 0.0], device="cuda", dtype=torch.float64)
        total = ce
    total.backward()
    want4 = torch.stack([total.detach(), ce.detach(), w[0], w[1]])
    torch.testing.assert_close(out4.double(), want4, rtol=3e-6, atol=1e-7)
    assert float((dl.double() - z.grad).norm() / z.grad.norm()) < 3e-6


@pytest.mark.parametrize("batch,n,d_s,d_t", [(6, 196, 192, 768), (5, 64, 192, 384), (4, 196, 384, 1024), (3, 40, 24, 56)])
def test_procrustes_fwd_composite_equals_the_kernel_chain(nat, batch, n, d_s, d_t):
    """basd_procrustes_fwd (one C call on a workspace) against the same chain built from the individually exported
    entries (losses/procrustes_chain.py): feature side (n > d_s) and token side (n <= d_s); then against torch's SVD"""
    from basd_amd.losses.procrustes_chain import procrustes_fwd_chain
    g = torch.Generator().manual_seed(batch * n + d_s)
    s_w = (torch.randn(batch, n, d_s, generator=g) / n ** 0.5).cuda()
    t_w = (torch.randn(batch, n, d_t, generator=g) / n ** 0.5).cuda()
    s_w -= s_w.mean(1, keepdim=True)
    t_w -= t_w.mean(1, keepdim=True)
    nuc, fac_s, a_t = nat.procrustes_fwd(s_w, t_w, 1e-13)
    nat.check_status()
    nuc2, fac2, at2 = procrustes_fwd_chain(nat, s_w, t_w, 1e-13)
    torch.testing.assert_close(nuc, nuc2, rtol=1e-6, atol=0)
    torch.testing.assert_close(fac_s, fac2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a_t, at2, rtol=1e-5, atol=1e-6)
    cross = s_w.double().transpose(1, 2) @ t_w.double()
    u, sv, vh = torch.linalg.svd(cross, full_matrices=False)
    torch.testing.assert_close(nuc.double(), sv.sum(-1), rtol=2e-6, atol=0)
    gpol = u @ vh                                                   # [d_s, d_t]; full rank here (n - 1 >= d_s) or not
    want_pt = s_w.double() @ gpol                                   # s_w G = a_t t_w
    got_pt = a_t.double() @ t_w.double()
    want_ps = t_w.double() @ gpol.transpose(1, 2)                   # t_w G^T
    got_ps = fac_s.double() if n > d_s else fac_s.double() @ s_w.double()
    if n - 1 >= d_s:                                                # G unique
        assert float((got_pt - want_pt).norm() / want_pt.norm()) < 2e-5
        assert float((got_ps - want_ps).norm() / want_ps.norm()) < 2e-5
    else:                                                           # the null-space part of G cancels in these products
        assert float((got_pt - want_pt).norm() / want_pt.norm()) < 2e-5
        assert float((got_ps - want_ps).norm() / want_ps.norm()) < 2e-5


def test_procrustes_fwd_rejects_a_short_workspace(nat):
    import ctypes
    lib = nat.lib()
    need = lib.basd_procrustes_workspace_bytes(4, 196, 192, 768)
    assert need > 4 * 5 * 196 * 196 * 8
    s_w = torch.zeros(4, 196, 192, device="cuda")
    t_w = torch.zeros(4, 196, 768, device="cuda")
    out = torch.zeros(4, 196, 196, device="cuda")
    ws = torch.empty(need // 2, dtype=torch.uint8, device="cuda")
    rc = lib.basd_procrustes_fwd(s_w.data_ptr(), t_w.data_ptr(), 4, 196, 192, 768, ctypes.c_double(1e-13), out.data_ptr(),
                                 out.data_ptr(), out.data_ptr(), None, ws.data_ptr(), ctypes.c_int64(ws.numel()), None)
    assert rc == 3 and b"workspace" in lib.basd_last_error()          # BASD_ERR_WORKSPACE


def test_transpose_table_all_weights_in_one_launch(nat):
    """bf16 W^T images of many matrices of the flat fp32 master buffer (ragged shapes, > 64 entries = two launches)"""
    g = torch.Generator().manual_seed(5)
    shapes = [(192, 576), (576, 192), (768, 192), (192, 768), (33, 70), (1, 5), (64, 64)] * 10
    table, src, dst = [], 0, 0
    for r, c in shapes:
        table.append((src, dst, r, c))
        src += (r * c + 63) // 64 * 64
        dst += (r * c + 63) // 64 * 64
    master = torch.randn(src, generator=g).cuda()
    out = torch.zeros(dst, dtype=torch.bfloat16, device="cuda")
    nat.transpose_table(master, out, table)
    for s0, d0, r, c in table:
        want = master[s0:s0 + r * c].view(r, c).t().to(torch.bfloat16)
        assert torch.equal(out[d0:d0 + r * c].view(c, r), want)


def test_fused_mlp_matches_the_unfused_layers():
    """timm Mlp of a trained block: fused path (GELU inside the GEMM epilogues) against fc1 -> nn.GELU -> fc2 on the
    per-layer kernels: outputs and all five gradients"""
    from basd_amd.models.vit import Mlp
    from basd_amd.models import linear as lin_mod
    torch.manual_seed(3)
    mlp = Mlp(192, 768).cuda()
    x = torch.randn(8, 197, 192, device="cuda").bfloat16().requires_grad_(True)
    g = torch.randn(8, 197, 192, device="cuda").bfloat16() * 0.1
    assert lin_mod.fused_mlp_ok(x, mlp.fc1, mlp.fc2)
    y = mlp(x)
    y.backward(g)
    got = [y.detach().float(), x.grad.float()] + [p.grad.clone() for p in mlp.parameters()]
    x.grad = None
    for p in mlp.parameters():
        p.grad = None
    y2 = mlp.fc2(mlp.act(mlp.fc1(x)))
    y2.backward(g)
    want = [y2.detach().float(), x.grad.float()] + [p.grad for p in mlp.parameters()]
    for a, b in zip(got, want):
        assert float((a - b).norm() / b.norm()) < 6e-3, float((a - b).norm() / b.norm())


@pytest.mark.parametrize("M,N,K", [(50432, 576, 192), (1000, 192, 768), (333, 64, 64), (4096, 768, 192), (777, 128, 128),
                                   (50432, 192, 768), (50432, 192, 192), (50433, 768, 192), (4100, 384, 1536),
                                   (12345, 192, 384)])
def test_wgrad_bf16(nat, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    dy = (torch.randn(M, N, generator=g) * 0.1).bfloat16()
    x = torch.randn(M, K, generator=g).bfloat16()
    dw, db = nat.wgrad_bf16(dy.cuda(), x.cuda())
    ref_w = dy.double().t() @ x.double()
    ref_b = dy.double().sum(0)
    scale = float((dy.double().abs().t() @ x.double().abs()).max())
    assert torch.allclose(dw.cpu().double(), ref_w, rtol=0, atol=3e-6 * scale)
    assert torch.allclose(db.cpu().double(), ref_b, rtol=0, atol=3e-6 * float(dy.double().abs().sum(0).max()))


def test_basd_linear_matches_nn_linear_backward():
    from basd_amd.models.linear import BasdLinear
    torch.manual_seed(0)
    lin = BasdLinear(192, 576).cuda()
    ref = torch.nn.Linear(192, 576).cuda()
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(8, 197, 192, device="cuda", requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = lin(x)
        y2 = ref(x2)
    gy = torch.randn_like(y)
    y.backward(gy)
    y2.backward(gy)
    torch.testing.assert_close(y, y2, atol=0, rtol=0)
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad, atol=2e-2, rtol=2e-2)   # ref accumulates to bf16
    torch.testing.assert_close(lin.bias.grad, ref.bias.grad, atol=5e-2, rtol=2e-2)
    torch.testing.assert_close(x.grad, x2.grad, atol=1e-3, rtol=1e-2)
    # against an fp64 reference the fp32-accumulated kernel is the more accurate of the two
    w_true = gy.bfloat16().double().reshape(-1, 576).t() @ x.detach().bfloat16().double().reshape(-1, 192)
    err_mine = float((lin.weight.grad.double() - w_true).norm() / w_true.norm())
    err_ref = float((ref.weight.grad.double() - w_true).norm() / w_true.norm())
    assert err_mine < 1e-5 and err_mine <= err_ref


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batch_strided_views_are_consumed_without_copy(nat, dtype):
    """out[:, 1:, :] views (CLS stripped) through token_gram / mix / dots / prep == contiguous copies."""
    g = torch.Generator().manual_seed(9)
    B, T, D = 5, 41, 64
    blocks = [torch.randn(B, T, D, generator=g).to(dtype).cuda() for _ in range(3)]
    views = [b[:, 1:, :] for b in blocks]
    dense = [v.contiguous() for v in views]
    p = torch.linalg.qr(torch.randn(D, 32, generator=g))[0].T.contiguous().cuda()
    g1, c1 = nat.token_gram(views[0], p)
    g2, c2 = nat.token_gram(dense[0].reshape(-1, D), p)
    torch.testing.assert_close(g1, g2, rtol=1e-12, atol=1e-9)
    torch.testing.assert_close(c1, c2, rtol=1e-12, atol=1e-9)
    w = torch.softmax(torch.randn(2, 3, generator=g), 1).cuda()
    torch.testing.assert_close(nat.mix_tokens(views, w), nat.mix_tokens(dense, w), rtol=0, atol=0)
    gr = torch.randn(2, B, T - 1, D, generator=g).cuda()
    torch.testing.assert_close(nat.mix_grad_dots(views, gr), nat.mix_grad_dots(dense, gr), rtol=1e-12, atol=1e-9)
    t = torch.randn(B, T - 1, 96, generator=g).cuda()
    imp = (torch.rand(B, T - 1, generator=g) + 0.1).cuda()
    for a, b in zip(nat.procrustes_prep(views[1], t, imp), nat.procrustes_prep(dense[1], t, imp)):
        torch.testing.assert_close(a, b, rtol=0, atol=0)


def test_jacobi_two_matrices_per_workgroup_path(nat):
    """batch >= 512 takes the NMAT = 2 kernel (odd batch: the last workgroup carries one matrix)."""
    g = torch.Generator().manual_seed(21)
    batch, m, n = 513, 40, 33
    a = torch.randn(batch, m, n, generator=g) * torch.logspace(0, -3, n).view(1, 1, n)
    ld = nat.jacobi_ld(m)
    w = _colmajor(a.cuda(), ld)
    sigma, sweeps = nat.jacobi_svd(w, m)
    ref = torch.linalg.svdvals(a.double())
    assert float((sigma.cpu().double() / ref - 1).abs().max()) < 5e-6
    wf = w.cpu().double()[:, :, :m].transpose(1, 2)
    aat = a.double() @ a.double().transpose(1, 2)
    assert torch.allclose(wf @ wf.transpose(1, 2), aat, rtol=0, atol=3e-6 * float(aat.abs().max()))
    assert int(sweeps.max()) < 40 and int(sweeps.min()) >= 1
    # rank-masked blocks of different sizes sharing a workgroup
    ks = torch.randint(2, n + 1, (batch,), generator=g)
    b = torch.zeros(batch, n, n)
    for i in range(batch):
        b[i, :ks[i], :ks[i]] = torch.randn(int(ks[i]), int(ks[i]), generator=g)
    w2 = _colmajor(b.cuda(), nat.jacobi_ld(n))
    s2, _ = nat.jacobi_svd(w2, n, active=ks.int().cuda(), active_rows=True)
    assert torch.allclose(s2.cpu().double(), torch.linalg.svdvals(b.double()), rtol=3e-5, atol=2e-6)


@pytest.mark.parametrize("rows,D", [(50432, 192), (1000, 768), (37, 64), (513, 1280)])
def test_layernorm_fwd_bwd(nat, rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2.0 + 0.5).bfloat16()
    gamma = torch.randn(D, generator=g) * 0.5 + 1.0
    beta = torch.randn(D, generator=g) * 0.1
    dy = torch.randn(rows, D, generator=g).bfloat16()
    y, mean, rstd = nat.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-6)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    yr.backward(dy.double())
    assert torch.allclose(y.cpu().double(), yr.detach(), rtol=8e-3, atol=8e-3)          # bf16 output rounding
    assert torch.allclose(mean.cpu().double(), x.double().mean(-1), rtol=0, atol=1e-5)
    dgamma = torch.zeros(D, device="cuda")
    dbeta = torch.zeros(D, device="cuda")
    dx = nat.layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, rstd, dgamma, dbeta)
    assert torch.allclose(dx.cpu().double(), xr.grad, rtol=1e-2, atol=1e-2 * float(xr.grad.abs().max()))
    scale_g = float((dy.double().abs() * ((x.double() - x.double().mean(-1, keepdim=True)).abs())).sum(0).max()) + 1.0
    assert torch.allclose(dgamma.cpu().double(), gr.grad, rtol=1e-4, atol=1e-5 * scale_g)
    assert torch.allclose(dbeta.cpu().double(), br.grad, rtol=1e-4, atol=1e-5 * float(dy.double().abs().sum(0).max()))
    # frozen layer: no parameter gradients requested
    dx2 = nat.layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, rstd, None, None)
    assert torch.equal(dx, dx2)


@pytest.mark.parametrize("rows,D", [(1001, 768), (77, 192), (300, 1280)])
def test_add_layernorm_equals_add_then_layernorm(nat, rows, D):
    g = torch.Generator().manual_seed(rows * 3 + D)
    x = (torch.randn(rows, D, generator=g) * 2.0).bfloat16().cuda()
    r = torch.randn(rows, D, generator=g).bfloat16().cuda()
    gamma = (torch.randn(D, generator=g) * 0.5 + 1.0).cuda()
    beta = (torch.randn(D, generator=g) * 0.1).cuda()
    s, y = nat.add_layernorm_fwd(x, r, gamma, beta, 1e-6)
    s_ref = x + r                                            # bf16 add, rounded like torch's
    y_ref, _, _ = nat.layernorm_fwd(s_ref, gamma, beta, 1e-6)
    assert torch.equal(s, s_ref)
    assert torch.equal(y, y_ref)


@pytest.mark.parametrize("B,T,D", [(6, 197, 192), (3, 65, 384), (2, 50, 768)])
def test_add_layernorm_with_drop_path_scale_and_fused_backward(nat, B, T, D):
    """trained-block variant: s = residual + scale[sample] * branch with saved statistics; backward adds the residual
    gradient and emits the scaled branch gradient -- against fp32 torch on the same bf16 inputs"""
    g = torch.Generator().manual_seed(B * 100 + D)
    x = (torch.randn(B, T, D, generator=g) * 2.0).bfloat16().cuda()
    r = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    scale = (torch.bernoulli(torch.full((B,), 0.7), generator=g) / 0.7).cuda()
    gamma = (torch.randn(D, generator=g) * 0.5 + 1.0).cuda()
    beta = (torch.randn(D, generator=g) * 0.1).cuda()
    s, y, mean, rstd = nat.add_layernorm_fwd(x, r, gamma, beta, 1e-6, row_scale=scale, want_stats=True)
    s_ref = (r.double() + scale.double().view(-1, 1, 1) * x.double()).float().bfloat16()   # one rounding, like the fma
    assert float((s != s_ref).float().mean()) < 1e-4 and torch.allclose(s.float(), s_ref.float(), rtol=2 ** -7, atol=1e-3)
    y_ref, m_ref, r_ref = nat.layernorm_fwd(s, gamma, beta, 1e-6)
    assert torch.equal(y, y_ref) and torch.allclose(mean, m_ref) and torch.allclose(rstd, r_ref)
    dy = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    dres = torch.randn(B, T, D, generator=g).bfloat16().cuda()
    dgam, dbet = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    dx, dbr = nat.layernorm_bwd(dy, s, gamma, mean, rstd, dgam, dbet, dres=dres, row_scale=scale, want_branch=True)
    sf = s.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(sf, (D,), gm, bt, 1e-6).backward(dy.float())
    want = sf.grad + dres.float()
    assert float((dx.float() - want).abs().max()) < 2e-2 * float(want.abs().max())
    assert float((dbr.float() - scale.view(-1, 1, 1) * want).abs().max()) < 2e-2 * float(want.abs().max()) * float(scale.max())
    assert torch.allclose(dgam, gm.grad, rtol=2e-3, atol=2e-2) and torch.allclose(dbet, bt.grad, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("B,T,H,hd", [(5, 197, 12, 64), (3, 50, 3, 64), (2, 256, 6, 32), (4, 2, 1, 64), (3, 257, 16, 80),
                                      (2, 320, 2, 80)])
def test_cls_importance_matches_the_attention_map(nat, B, T, H, hd):
    """head-averaged CLS row of softmax(QK^T/sqrt(hd)) without the CLS column (teacher.py:33-37,
    relational.py:22-27), from the packed qkv projection."""
    g = torch.Generator().manual_seed(T + H)
    qkv = (torch.randn(B, T, 3 * H * hd, generator=g) * 1.5).to(torch.bfloat16)
    scale = hd ** -0.5
    out = nat.cls_importance(qkv.cuda(), H, hd, scale).cpu()
    x = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).float()
    q, k = x[0], x[1]
    logits = (q[:, :, :1] @ k.transpose(-2, -1))                     # fp32 accumulate
    ref = (logits.to(torch.bfloat16).float() * scale).softmax(dim=-1)[:, :, 0, 1:].mean(dim=1)
    assert out.shape == (B, T - 1)
    # a logit that lands on a bf16 rounding boundary may round the other way (different summation order)
    assert torch.allclose(out, ref, rtol=2e-2, atol=1e-6)
    exact = (logits * scale).softmax(dim=-1)[:, :, 0, 1:].mean(dim=1)
    assert float((out - exact).abs().max()) <= 1.5 * float((ref - exact).abs().max()) + 1e-6


@pytest.mark.parametrize("B,N,D,dt", [(6, 196, 768, torch.float32), (5, 49, 192, torch.bfloat16), (3, 7, 4, torch.float32)])
def test_procrustes_bwd_rows(nat, B, N, D, dt):
    g = torch.Generator().manual_seed(B * N + D)
    r = torch.randn(B, N, D, generator=g)
    w = torch.randn(B, N, D, generator=g)
    a = torch.rand(B, N, generator=g) + 0.01
    gl = torch.randn(B, generator=g)
    res = w.double() - r.double()
    want = (2.0 * gl.double().view(B, 1, 1) * a.double().sqrt().unsqueeze(-1) * res)
    want_dot = 2.0 * gl.double().view(B, 1) * (res * w.double()).sum(-1)
    out, dot = nat.procrustes_bwd_rows(r.cuda(), w.cuda(), a.cuda(), gl.cuda(), out_dtype=dt)
    tol = 1e-6 if dt == torch.float32 else 8e-3
    assert out.dtype == dt
    assert torch.allclose(out.cpu().double(), want, rtol=tol, atol=tol * float(want.abs().max()))
    assert torch.allclose(dot.cpu().double(), want_dot, rtol=1e-5, atol=1e-5 * float(want_dot.abs().max()))


@pytest.mark.parametrize("B,T,H,hd", [(3, 197, 12, 64), (2, 50, 3, 64), (2, 257, 2, 64), (1, 1, 1, 64), (2, 208, 1, 64),
                                      (2, 257, 16, 80), (3, 197, 2, 80), (2, 209, 1, 80), (1, 33, 3, 80)])
def test_attention_fwd_matches_sdpa(nat, B, T, H, hd):
    """fused teacher attention (+ tap) vs softmax(QK^T/sqrt(hd)) V in fp64 on the same bf16 inputs; head dim 80 and
    T = 257 are ViT-H/14, the teacher of BASELINE configs[4] (reference hook: src/models/teacher.py:27-39, any head dim)"""
    g = torch.Generator().manual_seed(T * 7 + H + hd)
    qkv = (torch.randn(B, T, 3 * H * hd, generator=g) * 1.2).to(torch.bfloat16)
    scale = hd ** -0.5
    out, imp = nat.attention_fwd(qkv.cuda(), H, hd, scale, want_importance=T >= 2)
    x = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).double()
    q, k, v = x[0], x[1], x[2]
    p = ((q @ k.transpose(-1, -2)) * scale).softmax(dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B, T, H * hd)
    err = float((out.cpu().double() - ref).abs().max())
    assert out.shape == (B, T, H * hd) and out.dtype == torch.bfloat16
    assert err < 2e-2 * float(ref.abs().max()) + 1e-3, err          # bf16 P and bf16 output rounding
    if T >= 2:
        logits = (q[:, :, :1] @ k.transpose(-2, -1)).float()
        want = (logits.to(torch.bfloat16).float() * scale).softmax(dim=-1)[:, :, 0, 1:].mean(dim=1)
        assert torch.allclose(imp.cpu(), want, rtol=2e-2, atol=1e-6)
        assert torch.allclose(imp, nat.cls_importance(qkv.cuda(), H, hd, scale), rtol=2e-2, atol=1e-6)


@pytest.mark.parametrize("B,T,H,hd", [(3, 196, 12, 64), (2, 50, 3, 64), (2, 256, 2, 64), (1, 1, 1, 64), (2, 208, 1, 64),
                                      (2, 256, 4, 80), (2, 209, 1, 80)])
def test_attention_fwd_query_mean_tap(nat, B, T, H, hd):
    """the tap of a teacher WITHOUT a CLS token (reference src/losses/relational.py:25-27: attention map averaged over
    heads and queries) as a by-product of the fused attention forward, vs the map built in fp64 from the same bf16 qkv"""
    g = torch.Generator().manual_seed(T * 5 + H + hd)
    qkv = (torch.randn(B, T, 3 * H * hd, generator=g) * 1.2).to(torch.bfloat16)
    scale = hd ** -0.5
    out, imp = nat.attention_fwd(qkv.cuda(), H, hd, scale, want_importance=True, query_mean=True)
    x = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).double()
    q, k, v = x[0], x[1], x[2]
    p = ((q @ k.transpose(-1, -2)) * scale).softmax(dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B, T, H * hd)
    assert float((out.cpu().double() - ref).abs().max()) < 2e-2 * float(ref.abs().max()) + 1e-3
    want = p.mean(dim=(1, 2))
    assert imp.shape == (B, T) and imp.dtype == torch.float32
    assert torch.allclose(imp.cpu().double(), want, rtol=2e-5, atol=1e-8)
    assert torch.allclose(imp.sum(-1).cpu(), torch.ones(B), atol=1e-5)
    out2, _ = nat.attention_fwd(qkv.cuda(), H, hd, scale)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("m,n", [(192, 192), (100, 100), (60, 50), (40, 10), (96, 21)])
def test_jacobi_block_ordering_large_batch(nat, m, n):
    """batches >= 512 take the block-ordering kernel (two columns per side and slot): singular values,
    orthogonality of the rotated columns and the invariance of A A^T, incl. an odd number of blocks"""
    batch = 512
    g = torch.Generator().manual_seed(m * 31 + n)
    a = torch.randn(batch, m, n, generator=g) * torch.logspace(0, -3, n).unsqueeze(0).unsqueeze(0)
    ld = nat.jacobi_ld(m)
    w = _colmajor(a.cuda(), ld)
    sigma, sweeps = nat.jacobi_svd(w, m)
    ref = torch.linalg.svdvals(a[:16].double())
    assert torch.allclose(sigma[:16].cpu().double(), ref, rtol=3e-5, atol=1e-6 * float(ref.max()))
    assert int(sweeps.max()) < 30
    cols = w[:16, :, :m].cpu().double()                       # [b, n, m] rotated columns, sorted by norm
    gram = cols @ cols.transpose(1, 2)
    off = gram - torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2))
    scale = torch.sqrt(torch.diagonal(gram, dim1=1, dim2=2).unsqueeze(2) * torch.diagonal(gram, dim1=1, dim2=2).unsqueeze(1))
    assert float((off.abs() / scale.clamp_min(1e-30)).max()) < 2e-5
    aat = a[:16].double() @ a[:16].double().transpose(1, 2)
    assert torch.allclose(cols.transpose(1, 2) @ cols, aat, rtol=0, atol=2e-5 * float(aat.abs().max()))
    assert float(w[:, :, m:].abs().max()) == 0.0 if ld > m else True


@pytest.mark.parametrize("batch", [4, 512])
def test_jacobi_converges_on_clustered_spectra(nat, batch):
    """groups of nearly equal singular values: tiny cosines still call for 45-degree rotations, so the
    quadratic-convergence stop must not end the solve on the cosine alone (both the small-batch and the
    block-ordering kernel)"""
    n = 96
    g = torch.Generator().manual_seed(7 + batch)
    sv = torch.logspace(0, -2, n // 8, dtype=torch.float64).repeat_interleave(8)
    sv = sv * (1 + 1e-6 * torch.randn(n, dtype=torch.float64, generator=g))
    q1 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    q2 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    a = (q1 * sv) @ q2.transpose(1, 2)
    w = _colmajor(a.float().cuda(), nat.jacobi_ld(n))
    sigma, sweeps = nat.jacobi_svd(w, n)
    cols = w[:4, :, :n].double()
    gram = cols @ cols.transpose(1, 2)
    d = torch.diagonal(gram, dim1=1, dim2=2).sqrt()
    cos = (gram / (d.unsqueeze(2) * d.unsqueeze(1))).abs()
    cos = cos - torch.diag_embed(torch.diagonal(cos, dim1=1, dim2=2))
    assert float(cos.max()) < 3e-6, float(cos.max())
    ref = torch.linalg.svdvals(a[:4])
    assert torch.allclose(sigma[:4].cpu().double(), ref, rtol=2e-5, atol=0)


@pytest.mark.parametrize("batch", [16, 512])
@pytest.mark.parametrize("case", ["rank 96 of 192", "graded to 1e-16", "graded to 1e-20, scaled 1e+6"])
def test_jacobi_on_rank_deficient_cholesky_factors(nat, batch, case):
    """the Jacobi input of the blocked eigensolver (functional._pair_rotation): F = L^T of the pivoted Cholesky factor
    of a rank-deficient fp64 Gram matrix.  The cancelled directions leave debris columns (norm 1e-10 .. 1e-23 of the
    largest) whose cosines with the real columns are noise: the rotation test must neither underflow on them
    (tol^2 alpha beta ~ 1e-47) nor chase them sweep after sweep -- round 3 found a BASELINE c5 step in ~30 raising
    NONCONVERGED from exactly such a matrix.  Both the odd-even (batch 16) and the block-ordering kernel (batch 512)."""
    nat.check_status()
    k = 192
    g = torch.Generator().manual_seed(batch)
    if case == "rank 96 of 192":
        lam = torch.cat([torch.cos(torch.rand(96, dtype=torch.float64, generator=g) * 1.5707) ** 2, torch.zeros(96, dtype=torch.float64)])
        scale = 1.0
    elif case == "graded to 1e-16":
        lam, scale = torch.logspace(0, -16, k, dtype=torch.float64), 1.0
    else:
        lam, scale = torch.logspace(0, -20, k, dtype=torch.float64), 1.0e6
    q = torch.linalg.qr(torch.randn(batch, k, k, dtype=torch.float64, generator=g))[0]
    gram = ((q * lam) @ q.transpose(1, 2) * scale).cuda()
    _, lw, piv, rank = nat.pchol(gram, 1e-13)
    wf = torch.zeros(batch, k, nat.jacobi_ld(k), dtype=torch.float32, device="cuda")
    wf[:, :, :k] = lw.transpose(1, 2)
    sigma, sweeps = nat.jacobi_svd(wf, k)
    nat.check_status()                                        # raises if a matrix ran out of sweeps
    assert int(sweeps.min()) > 0 and int(sweeps.max()) <= 34, (int(sweeps.min()), int(sweeps.max()))
    # the singular values above the Cholesky cut are the square roots of the Gram eigenvalues, to fp32 RELATIVE accuracy
    ref = (lam * scale).sqrt().sort(descending=True)[0]
    r = int(rank.min())
    keep = ref[:r] > 3e-5 * ref[0]                            # (the cut itself perturbs the values next to it)
    got = sigma[:, :r].cpu().double()
    assert torch.allclose(got[:, keep], ref[:r][keep].expand(batch, -1), rtol=3e-4, atol=0)


@pytest.mark.parametrize("B,T,H", [(3, 197, 3), (2, 65, 3), (2, 224, 6), (5, 17, 2), (2, 96, 12)])
def test_attention_bwd_matches_autograd(nat, B, T, H):
    """fused attention forward (+ LSE) and backward kernels against fp32 autograd through softmax(Q K^T / sqrt(hd)) V
    on the same bf16 inputs: outputs / gradients differ by the bf16 rounding of P, dS and of the results only"""
    hd = 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B, T, 3 * H * hd, generator=g) * 0.8).bfloat16().cuda()
    dout = torch.randn(B, T, H * hd, generator=g).bfloat16().cuda()
    scale = hd ** -0.5
    out, _, lse = nat.attention_fwd(qkv, H, hd, scale, want_lse=True)
    dqkv = nat.attention_bwd(qkv, out, dout, lse, H, hd, scale)
    x = qkv.float().reshape(B, T, 3, H, hd).requires_grad_(True)
    q, k, v = (x[:, :, i].transpose(1, 2) for i in range(3))
    logits = (q @ k.transpose(-1, -2)) * scale
    o = (logits.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, H * hd)
    o.backward(dout.float())
    ref = x.grad.reshape(B, T, -1)
    assert torch.allclose(lse, torch.logsumexp(logits, dim=-1).detach(), rtol=1e-5, atol=1e-4)
    assert float((out.float() - o.detach()).abs().max()) < 2e-2
    parts = dqkv.float().reshape(B, T, 3, H * hd), ref.reshape(B, T, 3, H * hd)
    for i, name in enumerate("qkv"):
        got, want = parts[0][:, :, i], parts[1][:, :, i]
        rel = float((got - want).norm() / want.norm())
        assert rel < 1.5e-2, (name, rel)
        assert float((got - want).abs().max()) < 4e-2 * float(want.abs().max()), name


@pytest.mark.parametrize("E,L,D,unnorm", [(4, 12, 192, True), (4, 32, 768, False), (1, 1, 32, True)])
def test_angle_weights_matches_the_torch_chain(nat, E, L, D, unnorm):
    """fused acos / spectral weighting / softmax / backward-seed diagonal (layer_selector.py:100-108) against the torch
    chain of the emulation, incl. clamped cosines (>= 1 - eps), rank-masked directions (sw = 0) and tiny cosines"""
    from tests import _emul
    g = torch.Generator().manual_seed(E * 100 + L)
    sig = torch.rand(E, L, D, generator=g).sort(dim=-1, descending=True).values
    sig[:, :, 0] = 1.0                                   # clamped: d(d2)/d(sigma) = 0 there
    sig[:, :, 1] = 1.0 - 3e-8
    sig[:, :, -3:] = torch.tensor([1e-7, 1e-11, 0.0])
    sw = torch.rand(L, D, generator=g).sort(dim=-1, descending=True).values + 0.1
    ranks = torch.randint(4, D, (L,), generator=g)
    sw = sw * (torch.arange(D).unsqueeze(0) < ranks.unsqueeze(1))
    lt = torch.linspace(-0.5, 0.8, E)
    d2, pre, w, coef = nat.angle_weights(sig.cuda(), sw.cuda(), lt.cuda(), unnorm)
    rd2, rpre, rw, rcoef = _emul.angle_weights(sig, sw, lt, unnorm)
    assert torch.allclose(d2.cpu(), rd2, rtol=2e-6, atol=1e-7) and torch.allclose(pre.cpu(), rpre, rtol=2e-6, atol=1e-7)
    assert torch.allclose(w.cpu(), rw, rtol=1e-5, atol=1e-7) and abs(float(w.sum(1).mean()) - 1.0) < 1e-6
    assert torch.allclose(coef.cpu(), rcoef, rtol=5e-5, atol=1e-6 * float(rcoef.abs().max()))
    assert float(coef[:, :, 0].abs().max()) == 0.0 and float(coef[:, :, -1].abs().max()) == 0.0


def test_empty_batches_are_noops(nat):
    """zero-sized batches return without launching (every C-ABI entry checks for them first)"""
    dev = "cuda"
    ld = nat.jacobi_ld(32)
    sigma, sweeps = nat.jacobi_svd(torch.zeros(0, 32, ld, device=dev), 32)
    assert sigma.shape == (0, 32) and sweeps.shape == (0,)
    w0, lwork, piv, rk = nat.pchol(torch.zeros(0, 32, 32, dtype=torch.float64, device=dev))
    assert w0.shape[0] == 0 and rk.shape == (0,)
    assert nat.trinv(lwork, piv, rk).shape == (0, 32, 32)
    c = nat.bgemm_f64(torch.zeros(0, 64, 16, dtype=torch.float64, device=dev), torch.zeros(0, 16, 64, dtype=torch.float64, device=dev))
    assert c.shape == (0, 64, 64)
    y, mean, rstd = nat.layernorm_fwd(torch.zeros(0, 192, dtype=torch.bfloat16, device=dev), torch.ones(192, device=dev),
                                      torch.zeros(192, device=dev), 1e-6)
    assert y.shape == (0, 192) and mean.shape == (0,)
    out, imp = nat.attention_fwd(torch.zeros(0, 197, 3 * 64, dtype=torch.bfloat16, device=dev), 1, 64, 0.125, True)
    assert out.shape == (0, 197, 64) and imp.shape == (0, 196)
    torch.cuda.synchronize()


@pytest.mark.parametrize("batch,n,d_s,d_t,s_dtype", [(6, 196, 192, 768, torch.bfloat16), (5, 196, 384, 1024, torch.float32),
                                                     (7, 64, 192, 384, torch.bfloat16), (3, 16, 32, 64, torch.float32),
                                                     (2, 52, 48, 80, torch.float32)])
def test_procrustes_bwd_entry_matches_the_unfused_chain(nat, batch, n, d_s, d_t, s_dtype):
    """basd_procrustes_bwd (one C call: the a_t t_w product as a bf16 three-product split with the residual / scaling /
    row dots in its epilogue; feature side (n > d_s) and token side) against the fp64 arithmetic of
    functional._ProcrustesFn.backward on the same factors"""
    g = torch.Generator().manual_seed(batch * 1000 + n)
    s_w = torch.randn(batch, n, d_s, generator=g).cuda()
    t_w = torch.randn(batch, n, d_t, generator=g).cuda()
    a = (torch.rand(batch, n, generator=g) + 0.1).cuda()
    a = (a / a.sum(-1, keepdim=True)).contiguous()
    gl = torch.randn(batch, generator=g).cuda()
    a_t = (torch.randn(batch, n, n, generator=g) / n ** 0.5).cuda()
    token_side = n <= d_s
    fac_s = ((torch.randn(batch, n, n, generator=g) / n ** 0.5) if token_side else torch.randn(batch, n, d_s, generator=g)).cuda()
    g_s, g_t, g_a = nat.procrustes_bwd(s_w, t_w, a, gl, fac_s, a_t, s_dtype)
    torch.cuda.synchronize()
    sw, tw, a64, gl64 = s_w.double(), t_w.double(), a.double(), gl.double()
    p_t = a_t.double() @ tw
    p_s = fac_s.double() @ sw if token_side else fac_s.double()
    c = (2.0 * gl64).view(-1, 1, 1) * a64.sqrt().unsqueeze(-1)
    want_t, want_s = c * (tw - p_t), c * (sw - p_s)
    dot = (2.0 * gl64).view(-1, 1) * (((tw - p_t) * tw).sum(-1) + ((sw - p_s) * sw).sum(-1))
    want_a = dot / (2.0 * a64)
    assert g_s.dtype == s_dtype and g_t.dtype == torch.float32 and g_s.shape == s_w.shape and g_t.shape == t_w.shape
    rel_t = float((g_t.double() - want_t).norm() / want_t.norm())
    rel_s = float((g_s.double() - want_s).norm() / want_s.norm())
    rel_a = float((g_a.double() - want_a).norm() / want_a.norm())
    assert rel_t < 3e-5, rel_t                           # three-product bf16 split: 2^-16 of |A| |W|
    assert rel_s < (5e-3 if s_dtype == torch.bfloat16 else 3e-5), rel_s
    assert rel_a < 1e-4, rel_a
    # element-wise: no entry is off by more than the split error of the product plus fp32 rounding
    assert float((g_t.double() - want_t).abs().max()) <= 1e-4 * float(want_t.abs().max())


@pytest.mark.parametrize("E,L,D,D_s,with_pre", [(4, 12, 192, 192, False), (2, 5, 64, 64, True), (3, 7, 48, 80, True)])
def test_angle_weights_bwd_entry_matches_the_torch_chain(nat, E, L, D, D_s, with_pre):
    """basd_angle_weights_bwd (one C call: layer reduction, eigenvalue-gap division, four fp64 products) against the
    torch arithmetic of functional._SelectorWeightsFn.backward on the same saved tensors"""
    g = torch.Generator().manual_seed(E * 100 + L)
    g_w = torch.randn(E, L, generator=g).cuda()
    g_pre_out = torch.randn(E, L, generator=g).cuda() if with_pre else None
    wts = torch.softmax(torch.randn(E, L, generator=g), dim=1).cuda()
    d2 = torch.rand(E, L, generator=g).cuda()
    log_temp = torch.randn(E, generator=g).cuda()
    t_seed = torch.randn(E, L, D, D, generator=g).cuda()
    v_s = torch.linalg.qr(torch.randn(E, D, D, generator=g))[0].cuda()
    lam_s = torch.sort(torch.rand(E, D, generator=g, dtype=torch.float64) + 0.01, descending=True).values.cuda()
    lam_s[:, 3] = lam_s[:, 2]                                      # an exact tie: that entry of K is defined as 0
    proj_s = torch.randn(D, D_s, generator=g).cuda()
    g_lt, w_tok = nat.angle_weights_bwd(g_w, g_pre_out, wts, d2, log_temp, t_seed, v_s, lam_s, proj_s)
    tau = torch.nn.functional.softplus(log_temp)
    g_pre = wts * (g_w - (wts * g_w).sum(dim=1, keepdim=True))
    if g_pre_out is not None:
        g_pre = g_pre + g_pre_out
    g_d2 = -g_pre / tau.unsqueeze(1)
    want_lt = (g_pre * d2).sum(dim=1) / (tau * tau) * torch.sigmoid(log_temp)
    c = (g_d2.unsqueeze(-1).unsqueeze(-1) * t_seed).sum(dim=1).double()
    gap = lam_s.unsqueeze(1) - lam_s.unsqueeze(2)
    k = torch.where(gap.abs() > 0, c / torch.where(gap.abs() > 0, gap, torch.ones_like(gap)), torch.zeros_like(c))
    v64, p64 = v_s.double(), proj_s.double()
    gg = v64.transpose(1, 2) @ k @ v64
    want = (p64.t() @ (gg + gg.transpose(1, 2)) @ p64)
    torch.testing.assert_close(g_lt, want_lt, rtol=2e-5, atol=1e-6)
    rel = float((w_tok.double() - want).norm() / want.norm())
    assert rel < 1e-5, rel


@pytest.mark.parametrize("rows,d,mean_scale", [(25088, 384, 0.0), (4100, 768, 30.0), (777, 400, 3.0), (64, 16, 0.0)])
def test_gram_f32_centred_matches_fp64(nat, rows, d, mean_scale):
    """basd_gram_f32_centred (tile-centred Gram of a tall fp32 matrix on the fp32 matrix cores, fp64 across the 64-row
    tiles; the wide students' token statistics) against the fp64 product: lower triangle and column sums of the
    UNCENTRED statistics, and the centred Gram formed from them the way the selector does -- also when the column means
    are 30 x the spread (the case the tile centring exists for) and for a ragged last tile"""
    import ctypes
    g = torch.Generator().manual_seed(rows + d)
    z = (torch.randn(rows, d, generator=g) + mean_scale * torch.randn(1, d, generator=g)).cuda()
    gram = torch.zeros(d, d, dtype=torch.float64, device="cuda")
    cs = torch.zeros(d, dtype=torch.float64, device="cuda")
    rc = nat.lib().basd_gram_f32_centred(nat._ptr(z), ctypes.c_int64(rows), d, nat._ptr(gram), nat._ptr(cs), nat._stream())
    assert rc == 0, nat.lib().basd_last_error()
    z64 = z.double()
    ref, refc = z64.t() @ z64, z64.sum(0)
    low, rlow = torch.tril(gram), torch.tril(ref)
    assert float((cs - refc).abs().max()) <= 1e-9 * float(refc.abs().max() + 1.0)
    assert float((low - rlow).abs().max()) <= 2e-7 * float(ref.abs().max())
    # the centred Gram (unc - s s^T / m): fp32-level accuracy relative to ITS scale, not the uncentred one's
    full = low + torch.tril(gram, -1).t()
    cen = full - cs.unsqueeze(1) * cs.unsqueeze(0) / rows
    zc = z64 - z64.mean(0, keepdim=True)
    rcen = zc.t() @ zc
    assert float((cen - rcen).abs().max()) <= 5e-6 * float(rcen.abs().max())
