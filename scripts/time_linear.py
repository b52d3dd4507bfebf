"""Library GEMM rates for the teacher / student linear layers, with and without TunableOp (GPU)."""
import os, sys, time, torch
import torch.nn.functional as F

def timeit(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3

M = 256 * 197
shapes = [("t.qkv", 768, 2304), ("t.proj", 768, 768), ("t.fc1", 768, 3072), ("t.fc2", 3072, 768),
          ("s.qkv", 192, 576), ("s.proj", 192, 192), ("s.fc1", 192, 768), ("s.fc2", 768, 192)]
print("TUNABLEOP", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"), flush=True)
tot = 0
for name, k, n in shapes:
    x = torch.randn(M, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: F.linear(x, w, b))
    g = torch.randn(M, n, device="cuda", dtype=torch.bfloat16)
    ms_d = timeit(lambda: g @ w)                       # dgrad
    print(f"{name}: fwd {ms*1e3:7.1f} us {2*M*k*n/ms/1e9:7.1f} TF/s   dgrad {ms_d*1e3:7.1f} us {2*M*k*n/ms_d/1e9:7.1f} TF/s", flush=True)
    tot += ms * (12 if name[0] == "t" else 12) + (ms_d * 12 if name[0] == "s" else 0)
print(f"per-step estimate {tot:.2f} ms")
