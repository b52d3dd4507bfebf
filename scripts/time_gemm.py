"""basd_gemm_bf16 against torch (fp32 reference on the same bf16 inputs) and timed against the library F.linear."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import basd_amd._native as nat

torch.manual_seed(0)
dev = "cuda"


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [(50432, 3072, 768, True), (50432, 768, 3072, False), (50432, 2304, 768, False), (50432, 768, 768, False),
          (50432, 768, 192, False), (50432, 192, 192, False), (50432, 576, 192, False), (25216, 4096, 1024, True),
          (1000, 768, 768, False), (300, 192, 64, True)]
for (M, N, K, gelu) in shapes:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    y = nat.gemm_bf16(x, w, b, gelu=gelu)
    ref = x.float() @ w.float().t() + b.float()
    if gelu:
        ref = F.gelu(ref)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    bad = int(((y.float() - ref).abs() > 2e-2 * ref.abs().max()).sum())
    t_own = bench(lambda: nat.gemm_bf16(x, w, b, gelu=gelu))
    if gelu:
        t_lib = bench(lambda: F.gelu(F.linear(x, w, b)))
    else:
        t_lib = bench(lambda: F.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} gelu={gelu}: max rel err {err:.2e} (bad {bad}); own {t_own:.0f} us = {fl / t_own / 1e6:.0f} TF/s; "
          f"library{' + gelu' if gelu else ''} {t_lib:.0f} us = {fl / t_lib / 1e6:.0f} TF/s", flush=True)
