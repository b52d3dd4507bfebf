"""Jacobi timing on Procrustes-like factors (GPU).  BASD_JACOBI_MODE selects an experimental variant."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat

def timeit(f, it=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3

for batch, n in [(1024, 192), (12, 192), (48, 64)]:
    g = torch.Generator().manual_seed(n)
    # cross-covariance-like: graded spectrum, condition ~1e5
    z = torch.randn(batch, 2 * n, n, dtype=torch.float64, generator=g) * torch.logspace(0, -2.5, n, dtype=torch.float64)
    z = z @ torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, generator=g))[0]
    a = (z.transpose(1, 2) @ z).cuda()
    w0, lw, piv, rk = nat.pchol(a)
    ref = torch.linalg.eigvalsh(a[:4].cpu()).flip(-1)
    keep = w0.clone()
    def run():
        w = keep.clone()
        return nat.jacobi_svd(w, n)
    sigma, sweeps = run()
    torch.cuda.synchronize()
    err = float(((sigma[:4].cpu().double() ** 2) / ref - 1).abs().max())
    t_clone = timeit(lambda: keep.clone())
    print(f"batch {batch} n {n}: jacobi {timeit(run) - t_clone:.3f} ms  sweeps min/mean/max {int(sweeps.min())}/{float(sweeps.float().mean()):.2f}/{int(sweeps.max())}  eig rel err {err:.2e}")
