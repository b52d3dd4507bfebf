"""basd_token_gram_bf16x3 at the c2 shapes; accuracy against an fp64 torch product on the same bf16 tokens.  (Round 4
timed a workspace + reduce form of the final accumulation against the 256-way fp64 atomics with this script: 221 vs
174 us -- rejected, see csrc/token_gram.hip.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat


def timeit(f, it=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


torch.manual_seed(0)
for (b, t, d_in, d_out) in [(256, 197, 768, 192), (256, 197, 192, 192), (64, 65, 384, 192)]:
    blk = (torch.randn(b, t, d_in, device="cuda") * 0.7 + 0.3).to(torch.bfloat16)
    x = blk[:, 1:, :]                                   # CLS-stripped view, consumed in place
    proj = torch.linalg.qr(torch.randn(d_in, d_out, device="cuda"))[0].t().contiguous()
    z = x.double().reshape(-1, d_in) @ proj.double().t()
    ref, refc = z.t() @ z, z.sum(0)
    for ws in (False,):
        gram = torch.zeros(d_out, d_out, dtype=torch.float64, device="cuda")
        cs = torch.zeros(d_out, dtype=torch.float64, device="cuda")
        nat.token_gram(x, proj, mirror=False, out=(gram, cs))
        low = torch.tril(gram)
        err = float((low - torch.tril(ref)).abs().max() / ref.abs().max())
        errc = float((cs - refc).abs().max() / refc.abs().max())
        def run():
            gram.zero_(); cs.zero_()
            nat.token_gram(x, proj, mirror=False, out=(gram, cs))
        t_zero = timeit(lambda: (gram.zero_(), cs.zero_()))
        print(f"rows {b * (t - 1)} d_in {d_in} d_out {d_out} atomics: {timeit(run) - t_zero:7.1f} us  "
              f"gram err {err:.1e} colsum err {errc:.1e}")
