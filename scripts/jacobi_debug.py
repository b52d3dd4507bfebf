"""Debug view of the Jacobi kernels on clustered spectra: per-matrix sweeps, sigma error, residual cosines."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat


def run(n, batch, group, spread, seed=7, tag=""):
    g = torch.Generator().manual_seed(seed)
    sv = torch.logspace(0, -2, n // group, dtype=torch.float64).repeat_interleave(group)
    sv = sv * (1 + spread * torch.randn(n, dtype=torch.float64, generator=g))
    q1 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    q2 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    a = (q1 * sv) @ q2.transpose(1, 2)
    ld = nat.jacobi_ld(n)
    w = torch.zeros(batch, n, ld, dtype=torch.float32, device="cuda")
    w[:, :, :n] = a.transpose(1, 2).float().cuda()
    sigma, sweeps = nat.jacobi_svd(w, n, flag_status=False)
    torch.cuda.synchronize()
    cols = w[:, :, :n].double().cpu()
    gram = cols @ cols.transpose(1, 2)
    dg = torch.diagonal(gram, dim1=1, dim2=2).clamp_min(1e-300).sqrt()
    cosm = (gram / (dg.unsqueeze(2) * dg.unsqueeze(1))).abs()
    cosm = cosm - torch.diag_embed(torch.diagonal(cosm, dim1=1, dim2=2))
    res = cosm.flatten(1).max(1).values
    ref = torch.linalg.svdvals(a)
    err = (sigma.cpu().double() / ref - 1).abs().max(1).values
    sw = sweeps.cpu()
    print(f"{tag} n {n} batch {batch} group {group} spread {spread:g}: sweeps min/mean/max {int(sw.min())}/{float(sw.float().mean()):.2f}/{int(sw.max())}"
          f"  nonconv {int((sw < 0).sum())}  sigma err max {float(err.max()):.2e} median {float(err.median()):.2e}"
          f"  resid max {float(res.max()):.2e} median {float(res.median()):.2e}  finite {bool(torch.isfinite(sigma).all())}")
    bad = torch.nonzero(err > 1e-4).flatten()[:3]
    for b in bad.tolist():
        e = (sigma[b].cpu().double() / ref[b] - 1).abs()
        print(f"   matrix {b}: sweeps {int(sw[b])}, worst sigma indices {torch.topk(e, 4).indices.tolist()} errs {torch.topk(e, 4).values.tolist()}")


tag = "B4" if os.environ.get("BASD_JACOBI_B4", "1") != "0" else "BLK"
for (n, batch, group, spread) in [(96, 64, 8, 1e-6), (96, 64, 8, 1e-3), (96, 64, 2, 1e-6), (96, 64, 8, 0.0), (192, 64, 8, 1e-6),
                                  (192, 64, 1, 0.0), (96, 64, 96, 1e-6), (64, 64, 8, 1e-6)]:
    run(n, batch, group, spread, tag=tag)
