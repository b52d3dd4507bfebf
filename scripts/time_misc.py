"""Timing of the streaming kernels (GPU): LayerNorm fwd/bwd, procrustes_prep, bwd rows, Jacobi tolerance."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat

def timeit(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6

for rows, D in [(50432, 192), (50432, 768)]:
    x = torch.randn(rows, D, device="cuda").bfloat16()
    dy = torch.randn(rows, D, device="cuda").bfloat16()
    g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    y, mean, rstd = nat.layernorm_fwd(x, g, b, 1e-6)
    dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    tf = timeit(lambda: nat.layernorm_fwd(x, g, b, 1e-6))
    tb = timeit(lambda: nat.layernorm_bwd(dy, x, g, mean, rstd, dg, db))
    tb0 = timeit(lambda: nat.layernorm_bwd(dy, x, g, mean, rstd, None, None))
    mb = rows * D * 2 / 1e6
    print(f"LN rows {rows} D {D}: fwd {tf:.1f} us ({2*mb/tf/1e0:.0f} GB/s... {2*mb/tf:.2f} MB/us)  bwd {tb:.1f} us ({3*mb/tb:.2f} MB/us)  bwd frozen {tb0:.1f} us")
B = 256
s = torch.randn(B, 197, 192, device="cuda").bfloat16()[:, 1:]
t = torch.randn(B, 196, 768, device="cuda")
imp = torch.rand(B, 196, device="cuda") + 0.1
tp = timeit(lambda: nat.procrustes_prep(s, t, imp))
print(f"procrustes_prep: {tp:.1f} us")
M = 50432
tot = 0
for name, n, k in [("qkv", 576, 192), ("proj", 192, 192), ("fc1", 768, 192), ("fc2", 192, 768)]:
    dy = torch.randn(M, n, device="cuda").bfloat16()
    x = torch.randn(M, k, device="cuda").bfloat16()
    dw = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
    tw = timeit(lambda: nat.wgrad_bf16(dy, x, True, dw, db))
    tot += tw
    print(f"wgrad {name}: {tw:.1f} us  {2*M*n*k/tw/1e6:.0f} TF/s  {(M*(n+k)*2)/tw/1e6:.2f} TB/s")
print(f"wgrad total {tot:.1f} us")
