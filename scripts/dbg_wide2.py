"""Debug aid: check every wide psd_eig call of a full loss evaluation against fp64 eigh on the GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from basd_amd.losses import functional as BF, _ops
from tests._golden import load
from tests._run_loss import run_basd_loss
_ops.set_ops(None)
name = sys.argv[1] if len(sys.argv) > 1 else "c5_b4"
shape, inputs, gold = load(name)
calls = []
real = BF.psd_eig
def spy(a64, lower_only=False):
    out = real(a64, lower_only)
    if a64.shape[-1] > 192:
        calls.append((a64.detach().clone(), out[0].clone(), out[1].clone()))
    return out
BF.psd_eig = spy
res = run_basd_loss(shape, inputs, gold, "hard", device="cuda")
BF.psd_eig = real
for ci, (a, sig, u) in enumerate(calls):
    a = torch.tril(a) + torch.tril(a, -1).transpose(-1, -2)
    a = torch.nan_to_num(a)
    lam = torch.linalg.eigvalsh(a).flip(-1).clamp_min(0)
    l0 = lam[:, :1]
    ud = u.double()
    n = a.shape[-1]
    live = ud.abs().amax(-1) > 0
    eye = torch.diag_embed(live.double())
    orth = (ud @ ud.transpose(1, 2) - eye).abs().amax((1, 2))
    resid = (ud @ a - (sig.double() ** 2).unsqueeze(-1) * ud).abs().amax((1, 2)) / l0[:, 0]
    eerr = ((sig.double() ** 2 - lam).abs() / l0).amax(1)
    bad = torch.nonzero((orth > 1e-5) | (resid > 1e-5) | (eerr > 1e-5) | ~torch.isfinite(orth)).flatten().tolist()
    print(f"call {ci}: batch {a.shape[0]} n {n} orth max {float(orth.max()):.2e} resid max {float(resid.max()):.2e} "
          f"eig err max {float(eerr.max()):.2e} live min {int(live.sum(1).min())} max {int(live.sum(1).max())} bad {bad[:20]}")
    for bi in bad[:3]:
        print("    mat", bi, "orth", float(orth[bi]), "resid", float(resid[bi]), "eerr", float(eerr[bi]), "live", int(live[bi].sum()),
              "lam head", lam[bi, :3].tolist(), "lam@live", float(lam[bi, int(live[bi].sum()) - 1]))
