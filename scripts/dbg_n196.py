"""Debug aid: the n = 196 kernel variants (pchol global fallback, column-wise trinv, oe7 Jacobi) against fp64 torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basd_amd._native as nat
torch.manual_seed(0)
dev = "cuda"
for n, m in ((196, 384), (192, 384), (64, 192), (196, 1024)):
    b = 16
    x = torch.randn(b, n, m, device=dev, dtype=torch.float64)
    x = x - x.mean(1, keepdim=True)          # rank n - 1
    a = x @ x.transpose(1, 2)
    w0, lw, piv, rank = nat.pchol(a, 1e-13)
    rec = lw.transpose(1, 2) @ lw             # sum_k lw[k, r] lw[k, r']
    print(n, m, "rank", rank.tolist()[:4], "pchol recon", float((rec - a).abs().max() / a.abs().max()))
    linv = nat.trinv(lw, piv, rank)           # [b, k, r]
    ident = linv @ lw.transpose(1, 2)         # L^-1 P (P^T L) -> I_rank
    r = int(rank[0])
    eye = torch.zeros(n, n, device=dev, dtype=torch.float64); eye[:r, :r] = torch.eye(r, device=dev, dtype=torch.float64)
    print("   trinv |W L - I|", float((ident - eye).abs().max()))
    sig, sw = nat.jacobi_svd(w0, n)
    ref = torch.linalg.svdvals(x)
    print("   jacobi sigma rel err", float(((sig.double() - ref).abs() / ref[:, :1]).max()), "sweeps", sw.tolist()[:4])
    cols = w0[:, :, :n].double()
    gram = cols @ cols.transpose(1, 2)
    off = gram - torch.diag_embed(torch.diagonal(gram, dim1=1, dim2=2))
    print("   jacobi max offdiag / sigma0^2", float(off.abs().max() / gram.abs().max()))
    g2 = nat.bgemm_f64(x.float(), x.float(), trans_b=True, symmetric=True)
    print("   bgemm sym err", float((g2 - x.float().double() @ x.float().double().transpose(1, 2)).abs().max() / a.abs().max()))
