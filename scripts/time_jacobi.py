"""Jacobi timing on Procrustes-like factors (GPU).  BASD_JACOBI_MODE selects an experimental variant."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat

def timeit(f, it=7):
    """median over `it` device-event brackets (a host-side stall of the allocator / runtime once showed up as a 22 ms
    'launch' in the mean of three wall-clock iterations)"""
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

for batch, n in [(1024, 192), (4, 192), (24, 192), (48, 192), (48, 64)]:
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
    wout = keep.clone()
    nat.jacobi_svd(wout, n)
    cols = wout[:32, :, :n].double()                       # rotated columns of 32 matrices
    gram = cols @ cols.transpose(1, 2)
    dg = torch.diagonal(gram, dim1=1, dim2=2).clamp_min(1e-300).sqrt()
    cosm = (gram / (dg.unsqueeze(2) * dg.unsqueeze(1))).abs()
    cosm = cosm - torch.diag_embed(torch.diagonal(cosm, dim1=1, dim2=2))
    resid = float(cosm.max())
    t_clone = timeit(lambda: keep.clone())
    print(f"batch {batch} n {n}: jacobi {timeit(run) - t_clone:.3f} ms  sweeps min/mean/max {int(sweeps.min())}/{float(sweeps.float().mean()):.2f}/{int(sweeps.max())}  eig rel err {err:.2e}  max residual |cos| {resid:.2e}")

# clustered spectrum (groups of nearly equal singular values): large-angle rotations at tiny cosines
batch, n = 512, 192
g = torch.Generator().manual_seed(7)
sv = torch.logspace(0, -2, 24, dtype=torch.float64).repeat_interleave(8) * (1 + 1e-6 * torch.randn(192, dtype=torch.float64, generator=g))
q1 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
q2 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
a = (q1 * sv) @ q2.transpose(1, 2)
ld = nat.jacobi_ld(n)
w = torch.zeros(batch, n, ld, dtype=torch.float32, device="cuda")
w[:, :, :n] = a.transpose(1, 2).float().cuda()
sigma, sweeps = nat.jacobi_svd(w, n)
cols = w[:32, :, :n].double()
gram = cols @ cols.transpose(1, 2)
dg = torch.diagonal(gram, dim1=1, dim2=2).clamp_min(1e-300).sqrt()
cosm = (gram / (dg.unsqueeze(2) * dg.unsqueeze(1))).abs()
cosm = cosm - torch.diag_embed(torch.diagonal(cosm, dim1=1, dim2=2))
ref = torch.linalg.svdvals(a[:8])
print(f"clustered spectrum batch {batch}: sweeps min/mean/max {int(sweeps.min())}/{float(sweeps.float().mean()):.2f}/{int(sweeps.max())} "
      f"sigma rel err {float((sigma[:8].cpu().double() / ref - 1).abs().max()):.2e}  max residual |cos| {float(cosm.max()):.2e}")
