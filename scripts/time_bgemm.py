"""Timing + check of the fp64 batched GEMM on the Procrustes-core shapes (GPU)."""
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

B = 1024
g = torch.Generator().manual_seed(0)
f32, f64 = torch.float32, torch.float64
cases = [  # name, a shape, a dtype, b shape, b dtype, trans_a, trans_b, out dtype, symmetric
    ("cross   s_w^T t_w", (B, 196, 192), f32, (B, 196, 768), f32, True, False, f64, False),
    ("mx      c c^T sym", (B, 192, 768), f64, (B, 192, 768), f64, False, True, f64, True),
    ("j1      linv wf^T", (B, 192, 192), f64, (B, 192, 192), f32, False, True, f64, False),
    ("theta   u^T j1^T ", (B, 192, 192), f32, (B, 192, 192), f64, True, True, f32, False),
    ("q2      linv cross", (B, 192, 192), f64, (B, 192, 768), f64, False, False, f32, False),
    # the feature-side chain of basd_procrustes_fwd at c2 (round 4: 196 = 12.25 sub-tiles of 16)
    ("Gt      t_w t_w^T sym", (B, 196, 768), f32, (B, 196, 768), f32, False, True, f64, True),
    ("h       Gt s_w     ", (B, 196, 196), f64, (B, 196, 192), f32, False, False, f64, False),
    ("gram    s_w^T h    ", (B, 196, 192), f32, (B, 196, 192), f64, True, False, f64, False),
    ("fac_s   Gt pm^T    ", (B, 196, 196), f64, (B, 192, 196), f64, False, True, f32, False),
    ("a_t     s_w pm     ", (B, 196, 192), f32, (B, 192, 196), f64, False, False, f32, False),
]
tot = 0.0
for name, sa, da, sb, db, ta, tb, do, sym in cases:
    a = torch.randn(*sa, generator=g, dtype=f64).to(da).cuda()
    b = a if sym else torch.randn(*sb, generator=g, dtype=f64).to(db).cuda()
    c = nat.bgemm_f64(a, b, trans_a=ta, trans_b=tb, out_dtype=do, symmetric=sym)
    aa = a[:8].double(); bb = b[:8].double()
    ref = (aa.transpose(1, 2) if ta else aa) @ (bb.transpose(1, 2) if tb else bb)
    err = float((c[:8].double() - ref).abs().max() / ref.abs().max())
    ms = timeit(lambda: nat.bgemm_f64(a, b, trans_a=ta, trans_b=tb, out_dtype=do, symmetric=sym))
    M, N, K = ref.shape[1], ref.shape[2], (sa[1] if ta else sa[2])
    fl = 2.0 * B * M * N * K * (2 / 3 if sym else 1.0)
    tot += ms
    print(f"{name}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s computed  rel err {err:.1e}")
print(f"total {tot:.3f} ms")
