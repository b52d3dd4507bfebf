"""Teacher attention forward: fused kernel (+tap) vs library SDPA + separate tap (GPU)."""
import sys, os, time, torch
import torch.nn.functional as F
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

for B, T, H in [(256, 197, 12), (256, 197, 3), (256, 257, 16)]:
    hd = 64
    qkv = torch.randn(B, T, 3 * H * hd, device="cuda").bfloat16()
    scale = hd ** -0.5
    def lib():
        q, k, v = qkv.reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
        o = F.scaled_dot_product_attention(q, k, v)
        return o.transpose(1, 2).reshape(B, T, H * hd)
    with torch.no_grad():
        t_lib = timeit(lib)
        t_tap = timeit(lambda: nat.cls_importance(qkv, H, hd, scale)) if T <= 256 else float("nan")
        t_fused = timeit(lambda: nat.attention_fwd(qkv, H, hd, scale, True))
        t_fused_notap = timeit(lambda: nat.attention_fwd(qkv, H, hd, scale, False))
        err = float((lib().float() - nat.attention_fwd(qkv, H, hd, scale, False)[0].float()).abs().max())
    mb = B * T * H * hd * 2 * 4 / 1e6
    print(f"B {B} T {T} H {H}: library sdpa {t_lib:.1f} us + tap {t_tap:.1f} us | fused {t_fused:.1f} us (no tap {t_fused_notap:.1f})"
          f"  {mb / t_fused:.2f} MB/us  max |diff| vs library {err:.3e}")
