"""Debug aid: where does the wide path (D_s > 192) lose accuracy on the GPU?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import basd_amd._native as nat
from basd_amd.losses import functional as BF, _ops
from oracle import basd_oracle as O
from tests._golden import load, rel_l2
_ops.set_ops(None)
name = sys.argv[1] if len(sys.argv) > 1 else "c4_b8"
shape, inputs, gold = load(name)
dev = "cuda"
layers = inputs["token_layers"]
# ---- 1. Procrustes alone (token side) vs the oracle, on teacher layer 0 as "mixed" tokens
s = [inputs["student_tokens"][l].to(dev).requires_grad_(True) for l in layers]
t = torch.stack([inputs["teacher_tokens"][j] for j in range(len(layers))]).to(dev)
imp = torch.stack([O.importance_from_attention(inputs["teacher_attns"][j], shape.has_cls) for j in range(len(layers))]).to(dev)
val = BF.procrustes_all(s, t, imp)
val.mean().backward()
for i, l in enumerate(layers):
    sc = inputs["student_tokens"][l].clone().requires_grad_(True)
    v = O.procrustes(sc, inputs["teacher_tokens"][i], imp[i].cpu())
    v.mean().backward()
    print("procrustes", l, "value rel", float(((val[i].cpu() - v) / v).abs().max()), "grad rel-L2", rel_l2(s[i].grad.cpu(), sc.grad))
# ---- 2. blocked eigensolver vs fp64 eigh on the student centred Gram matrices
proj_s = gold["proj_s"].to(dev)
mats = []
for l in layers:
    st = inputs["student_tokens"][l].to(dev)
    g, c = nat.token_gram(st, proj_s, mirror=True)
    z = st.reshape(-1, shape.D_s).double() @ proj_s.double().t()
    gref = z.t() @ z
    print("gram rel err", float((g - gref).abs().max() / gref.abs().max()), "colsum", float((c - z.sum(0)).abs().max()))
    m = st.shape[0] * st.shape[1]
    mats.append(g - torch.outer(c, c) / m)
a = torch.stack(mats)
for sw in (3, 5, 8):
    BF.WIDE_SWEEPS = sw
    sig, u, _ = BF.psd_eig(a)
    lam, vec = torch.linalg.eigh(a)
    lam = lam.flip(-1)
    print(sw, "eig rel err (top 100)", float(((sig.double() ** 2 - lam) / lam)[:, :100].abs().max()),
          "all", float(((sig.double() ** 2 - lam) / lam).abs().max()))
    ud = u.double()
    orth = ud @ ud.transpose(1, 2) - torch.eye(u.shape[-1], device=dev, dtype=torch.float64)
    res = (ud @ a - (sig.double() ** 2).unsqueeze(-1) * ud)
    print("   orth", float(orth.abs().max()), "residual / lam0", float(res.abs().max() / lam[:, :1].max()))
    # subspace angle of top-50 block
    for k in (24, 100):
        pr = ud[:, :k] @ vec.flip(-1)[:, :, :k]      # [b, k, k]
        print("   top", k, "subspace cos min", float(torch.linalg.svdvals(pr).min()))
