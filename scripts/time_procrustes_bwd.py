"""basd_procrustes_bwd (fused bf16 three-product split + residual epilogue) against the library fp32 bmm + the row
kernels, at the c2 shapes (1024 x [196, 196] x [196, 768])."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat


def timeit(f, it=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for (batch, n, d_s, d_t) in [(1024, 196, 192, 768), (512, 196, 384, 1024)]:
    g = torch.Generator().manual_seed(0)
    s_w = torch.randn(batch, n, d_s, generator=g).cuda(); t_w = torch.randn(batch, n, d_t, generator=g).cuda()
    a = torch.rand(batch, n, generator=g).cuda() + 0.1; a = (a / a.sum(-1, keepdim=True)).contiguous()
    gl = torch.randn(batch, generator=g).cuda()
    a_t = (torch.randn(batch, n, n, generator=g) / n ** 0.5).cuda()
    token = n <= d_s
    fac_s = ((torch.randn(batch, n, n, generator=g) / n ** 0.5) if token else torch.randn(batch, n, d_s, generator=g)).cuda()

    def fused():
        return nat.procrustes_bwd(s_w, t_w, a, gl, fac_s, a_t, torch.bfloat16)

    def unfused():
        p_t = a_t @ t_w
        p_s = fac_s @ s_w if token else fac_s
        g_s, dot_s = nat.procrustes_bwd_rows(p_s, s_w, a, gl, out_dtype=torch.bfloat16)
        g_t, dot_t = nat.procrustes_bwd_rows(p_t, t_w, a, gl, out_dtype=torch.float32)
        return g_s, g_t, (dot_s + dot_t) / (2.0 * a)

    f, u = fused(), unfused()
    err = float((f[1] - u[1]).norm() / u[1].norm())
    print(f"batch {batch} n {n} d_s {d_s} d_t {d_t}: fused {timeit(fused):.3f} ms, library bmm + rows {timeit(unfused):.3f} ms, "
          f"g_t rel diff {err:.1e}")
