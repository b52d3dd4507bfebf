"""Which Python lines launch the torch / library kernels that are still inside a c2 step (GPU): one eager step under
torch.profiler with stacks, every non-basd device kernel attributed to the innermost frame inside this package."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import basd_amd._native as native
from basd_amd.config import load_config
from basd_amd.train import SyntheticLoader, build

cfg = load_config(os.path.join(ROOT, "vit-bias-aware-structural-distillation_amd", "configs", "config.yaml"), None,
                  ["data.batch_size=256", "data.dataset=synthetic", "model.student_preset=deit_tiny_patch16_224",
                   "basd.teacher_model_name=vit_base_patch16_224"])
dev = torch.device("cuda", 0)
native.lib()
trainer, _ = build(cfg, device=dev)
batch = next(iter(SyntheticLoader(256, cfg.model.vit.img_size, cfg.model.num_classes, 1, dev, seed=1)))
trainer.optimizer.train(); trainer.model.train()
trainer.overlap_teacher_stats = trainer.overlap_teacher_forward = False
for _ in range(2):
    trainer.train_step(batch); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    trainer.train_step(batch); torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_stack_n=12, group_by_input_shape=True):
    t = getattr(ka, "self_device_time_total", 0) or getattr(ka, "self_cuda_time_total", 0)
    if t <= 0:
        continue
    frame = next((f for f in ka.stack if "distillation_amd" in f or "basd_amd" in f), ka.stack[0] if ka.stack else "?")
    rows.append((t, ka.count, ka.key[:44], str(ka.input_shapes)[:64], frame.strip()[-100:]))
rows.sort(key=lambda r: -r[0])
print(f"total self device time: {sum(r[0] for r in rows) / 1e3:.2f} ms")
for t, n, name, shapes, frame in rows[:90]:
    print(f"{t / 1e3:7.3f} ms  x{n:3d}  {name:44s} {shapes:64s} {frame}")
