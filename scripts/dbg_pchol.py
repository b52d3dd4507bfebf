import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat
n = rank = 192
g = torch.Generator().manual_seed(n + rank)
x = torch.randn(3, 4 * n, rank, dtype=torch.float64, generator=g) * torch.logspace(0, -3, rank, dtype=torch.float64)
mix = torch.randn(3, rank, n, dtype=torch.float64, generator=g)
z = x @ mix
a = z.transpose(1, 2) @ z
ref = torch.linalg.eigvalsh(a).flip(-1)
w0, lwork, piv, rk = nat.pchol(a.cuda())
torch.cuda.synchronize()
lw = lwork.cpu()
sv_l64 = torch.linalg.svdvals(lw) ** 2
print("eig from fp64 factor (cpu svd):", float((sv_l64 / ref - 1).abs().max()))
w0c = w0.cpu().double()[:, :, :n]
sv_w0 = torch.linalg.svdvals(w0c) ** 2
print("eig from fp32 factor (cpu svd):", float((sv_w0 / ref - 1).abs().max()))
sigma, sweeps = nat.jacobi_svd(w0, n)
ev = sigma.cpu().double() ** 2
e = (ev / ref - 1).abs()
print("eig from gpu jacobi:", float(e.max()), "argmax", e.argmax(dim=1).tolist(), "sweeps", sweeps.tolist())
print("vs svd of same fp32 factor:", float((ev / sv_w0 - 1).abs().max()))
print("smallest ref", ref[:, -3:].tolist(), "largest", ref[:, 0].tolist())
diag = torch.diagonal(lw, dim1=1, dim2=2)
