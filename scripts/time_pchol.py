"""Timing + old/new comparison of the pivoted Cholesky and triangular inverse kernels (GPU)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat

def timeit(f, it=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3

for batch, n, rank in [(1024, 192, 192), (1024, 192, 120), (12, 192, 192), (4, 64, 64), (48, 100, 37), (512, 196, 195),
                       (512, 196, 120)]:
    g = torch.Generator().manual_seed(n + rank)
    z = torch.randn(batch, 2 * n, rank, dtype=torch.float64, generator=g) @ torch.randn(batch, rank, n, dtype=torch.float64, generator=g)
    a = (z.transpose(1, 2) @ z).cuda()
    w0, lw, piv, rk = nat.pchol(a)
    torch.cuda.synchronize()
    rec = lw.transpose(1, 2) @ lw
    err = float((rec - a).abs().max() / a.abs().max())
    perm_ok = all(sorted(piv[b].tolist()) == list(range(n)) for b in range(min(batch, 8)))
    print(f"batch {batch} n {n} rank {rank}: ranks ok {bool((rk == rank).all())} recon err {err:.2e} perm ok {perm_ok} "
          f"w0 pad zero {bool((w0[:, :, n:] == 0).all())} pchol {timeit(lambda: nat.pchol(a)):.3f} ms "
          f"trinv {timeit(lambda: nat.trinv(lw, piv, rk)):.3f} ms")
    out = nat.trinv(lw, piv, rk)
    b = 0
    pv = piv[b].long()
    lp = lw[b].t()[pv][:rank, :rank]
    x = out[b][:, pv]
    print("   trinv resid", float((x[:rank, :rank] @ lp - torch.eye(rank, dtype=torch.float64, device='cuda')).abs().max()))
