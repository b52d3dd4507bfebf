"""Device-event timing of basd_wgrad_bf16 on the student's shapes (DeiT-T at 256 images: M = 50432)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import basd_amd._native as nat  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 50432
for N, K in [(768, 192), (192, 768), (576, 192), (192, 192), (1536, 384), (384, 1536)]:
    dy = (torch.randn(M, N, device="cuda") * 0.1).bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    for _ in range(3):
        nat.wgrad_bf16(dy, x, out_w=dw, out_b=db)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    reps = 20
    for _ in range(reps):
        nat.wgrad_bf16(dy, x, out_w=dw, out_b=db)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    flop = 2.0 * M * N * K
    byts = 2.0 * M * (N + K)
    print(f"wgrad M={M} N={N} K={K}: {us:7.1f} us  {flop / us / 1e6:7.1f} TF/s  {byts / us / 1e6:6.2f} TB/s of operands")
