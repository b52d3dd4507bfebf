"""Jacobi stress cases (GPU): factors of rank-deficient / graded Gram matrices and clustered spectra."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import __graft_entry__  # noqa
import basd_amd._native as nat
torch.manual_seed(0)
k, b = 192, 512
def trial(name, lam, scale=1.0):
    q = torch.linalg.qr(torch.randn(b, k, k, dtype=torch.float64, device="cuda"))[0]
    g = (q * lam.cuda()) @ q.transpose(1, 2) * scale
    _, lw, piv, rank = nat.pchol(g, 1e-13)
    wf = torch.zeros(b, k, nat.jacobi_ld(k), dtype=torch.float32, device="cuda")
    wf[:, :, :k] = lw.transpose(1, 2)
    sigma, sw = nat.jacobi_svd(wf, k)
    st = int(nat.status_word("cuda")); nat.status_word("cuda").zero_()
    print(f"{name}: rank {int(rank.min())}..{int(rank.max())} sweeps {int(sw.abs().min())}..{int(sw.abs().max())} nonconverged {int((sw < 0).sum())} status {st & 255} "
          f"sigma min>0 {float(sigma[sigma > 0].min()):.2e} max {float(sigma.max()):.2e}", flush=True)
trial("logspace 0..-12", torch.logspace(0, -12, k, dtype=torch.float64))
trial("logspace 0..-16 (rank cut)", torch.logspace(0, -16, k, dtype=torch.float64))
trial("logspace 0..-13.5", torch.logspace(0, -13.5, k, dtype=torch.float64))
lam = torch.cat([torch.rand(100, dtype=torch.float64) * 0.9 + 0.1, torch.logspace(-9, -14, 92, dtype=torch.float64)])
trial("100 O(1) + 92 tiny 1e-9..1e-14", lam)
lam = torch.cat([torch.ones(60, dtype=torch.float64), torch.full((132,), 3e-13, dtype=torch.float64)])
trial("60 ones + 132 at 3e-13", lam)
trial("same, scale 1e-4", lam, 1e-4)
trial("same, scale 1e-8", lam, 1e-8)
lam = torch.cat([torch.ones(60, dtype=torch.float64), torch.logspace(-11, -13, 132, dtype=torch.float64)])
trial("60 ones + tail 1e-11..1e-13", lam)
trial("cos^2 of random angles", torch.cos(torch.rand(k, dtype=torch.float64) * 1.5707) ** 2)
trial("cos^4", torch.cos(torch.rand(k, dtype=torch.float64) * 1.5707) ** 4 * 1e-3)
trial("logspace 0..-20 (rank cut)", torch.logspace(0, -20, k, dtype=torch.float64))
trial("logspace 0..-16 scale 1e-6", torch.logspace(0, -16, k, dtype=torch.float64), 1e-6)
trial("logspace 0..-16 scale 1e+6", torch.logspace(0, -16, k, dtype=torch.float64), 1e+6)
lam = torch.cat([torch.cos(torch.rand(96, dtype=torch.float64) * 1.5707) ** 2, torch.zeros(96, dtype=torch.float64)])
trial("rank 96 exactly", lam)

# ---- clustered spectra
def colmajor(a, ld):
    b, m, n = a.shape
    w = torch.zeros(b, n, ld, dtype=torch.float32, device=a.device); w[:, :, :m] = a.transpose(1, 2); return w.contiguous()
def run(name, a, n):
    w = colmajor(a.float().cuda(), nat.jacobi_ld(n))
    sigma, sweeps = nat.jacobi_svd(w, n)
    st = int(nat.status_word("cuda")); nat.status_word("cuda").zero_()
    cols = w[:8, :, :n].double(); gram = cols @ cols.transpose(1, 2)
    d = torch.diagonal(gram, dim1=1, dim2=2).sqrt(); cos = (gram / (d.unsqueeze(2) * d.unsqueeze(1))).abs()
    cos = cos - torch.diag_embed(torch.diagonal(cos, dim1=1, dim2=2))
    print(f"{name}: sweeps {int(sweeps.abs().min())}..{int(sweeps.abs().max())} status {st & 255} max cos {float(cos.max()):.2e} (tol {n ** 0.5 * 5.96e-8:.2e})", flush=True)
for batch in (4, 512):
    n = 96; g = torch.Generator().manual_seed(7 + batch)
    sv = torch.logspace(0, -2, n // 8, dtype=torch.float64).repeat_interleave(8)
    sv = sv * (1 + 1e-6 * torch.randn(n, dtype=torch.float64, generator=g))
    q1 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    q2 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    run(f"clusters of 8, n 96, batch {batch}", (q1 * sv) @ q2.transpose(1, 2), n)
for batch, n, cluster in [(512, 192, 120), (16, 192, 120), (64, 196, 150), (512, 96, 96)]:
    g = torch.Generator().manual_seed(batch + n)
    sv = torch.cat([torch.ones(cluster, dtype=torch.float64), torch.logspace(-0.3, -3, n - cluster, dtype=torch.float64)])
    q1 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    q2 = torch.linalg.qr(torch.randn(batch, n, n, dtype=torch.float64, generator=g))[0]
    run(f"cluster {cluster} of {n}, batch {batch}", (q1 * sv) @ q2.transpose(1, 2), n)
g = torch.Generator().manual_seed(1)
a = torch.randn(512, 192, 192, generator=g) * torch.logspace(0, -3, 192).unsqueeze(0).unsqueeze(0)
run("graded 192, batch 512", a.double(), 192)
