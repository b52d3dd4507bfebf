"""Operator-level profile of one eager train step (torch.profiler, GPU): which aten ops / shapes the
remaining elementwise time comes from."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import basd_amd._native as native
from basd_amd.config import load_config
from basd_amd.train import SyntheticLoader, build
from torch.profiler import profile, ProfilerActivity

native.lib()
dev = torch.device("cuda", 0)
CFG = os.path.join(ROOT, "vit-bias-aware-structural-distillation_amd", "configs")
import bench
cfg = load_config(bench.CFG, None, ["data.batch_size=256", "data.dataset=synthetic", "model.grad_checkpointing=false"])
trainer, info = build(cfg, device=dev)
loader = SyntheticLoader(256, cfg.model.vit.img_size, cfg.model.num_classes, 1, dev, seed=1234)
batch = next(iter(loader))
trainer.optimizer.train(); trainer.model.train()
for _ in range(3):
    trainer.train_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(2):
        trainer.train_step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t / 2e3, e.count // 2, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total self device time {tot:.2f} ms/step")
for t, n, k, sh in rows[:70]:
    print(f"{t:8.3f} ms {n:5d}x  {k[:48]:48s} {sh}")
