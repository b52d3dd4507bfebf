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
from torch.profiler import profile, ProfilerActivity, record_function
import functools
import basd_amd.losses.functional as BF
import basd_amd.losses.combined as BC


def label(obj, name, tag):
    fn = getattr(obj, name)
    raw = fn.__func__ if isinstance(fn, staticmethod) else fn
    @functools.wraps(raw)
    def wrapped(*a, **k):
        with record_function("L:" + tag):
            return raw(*a, **k)
    setattr(obj, name, staticmethod(wrapped) if isinstance(obj, type) and name in ("forward", "backward") else wrapped)


for cls, tag in ((BF._SelectorWeightsFn, "selector"), (BF._MixFn, "mix"), (BF._ProcrustesFn, "procrustes")):
    label(cls, "forward", tag + ".fwd")
    label(cls, "backward", tag + ".bwd")
for fname in ("student_frames", "teacher_frames", "selector_weights", "mix_layers", "procrustes_all"):
    if hasattr(BF, fname):
        label(BF, fname, fname)
label(trainer.model, "forward", "student.forward")
label(trainer.basd_loss, "forward", "basd_loss.forward")
label(trainer, "_teacher_branch", "teacher_branch")

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function("L:step"):
        trainer.train_step(batch); torch.cuda.synchronize()

import collections
agg = collections.defaultdict(lambda: [0, 0.0])
per_label = collections.defaultdict(float)
for ev in prof.events():
    ks = getattr(ev, "kernels", None)
    if not ks:
        continue
    lab, q = "?", ev
    while q is not None:
        if q.name.startswith("L:") and q.name != "L:step":
            lab = q.name[2:]
            break
        q = q.cpu_parent
    if lab == "?":
        q = ev
        while q is not None:
            if q.name.endswith("Backward") or q.name.startswith("autograd::engine"):
                lab = "autograd:" + q.name[:40]
                break
            q = q.cpu_parent
    for kq in ks:
        if "basd::" in kq.name:
            continue
        agg[(lab, ev.name[:36], kq.name[:60])][0] += 1
        agg[(lab, ev.name[:36], kq.name[:60])][1] += kq.duration
        per_label[lab] += kq.duration
print("non-basd device time by label (us):")
for lab, t in sorted(per_label.items(), key=lambda kv: -kv[1]):
    print(f"  {t:9.1f}  {lab}")
print()
for (lab, op, kn), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
    print(f"{t:8.1f} us x{n:3d}  [{lab:28s}] {op:36s} {kn}")
