import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat
torch.manual_seed(0)
for rows, D in [(10, 192), (10240, 192), (10250, 192), (20480, 192), (50432, 192), (513, 1280), (2, 1280), (1, 1280), (300, 1024), (300, 2048)]:
    x = (torch.randn(rows, D) * 2 + 0.5).bfloat16().cuda()
    dy = torch.randn(rows, D).bfloat16().cuda()
    gamma = (torch.randn(D) * 0.5 + 1).cuda(); beta = torch.randn(D).cuda() * 0.1
    y, mean, rstd = nat.layernorm_fwd(x, gamma, beta, 1e-6)
    xf = x.double(); mu = xf.mean(-1, keepdim=True); rs = torch.rsqrt(((xf - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
    yref = (xf - mu) * rs * gamma.double() + beta.double()
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dx = nat.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db)
    xh = (xf - mu) * rs
    dgr = (dy.double() * xh).sum(0); dbr = dy.double().sum(0)
    print(rows, D, "y err", float((y.double() - yref).abs().max()), "mean err", float((mean.double() - mu.squeeze(-1)).abs().max()),
          "dgamma relerr", float((dg.double() - dgr).norm() / dgr.norm()), "dbeta relerr", float((db.double() - dbr).norm() / dbr.norm()))
