"""Where do the library / torch elementwise launches of one train step come from?

Runs a few eager c2 steps, then one under torch.profiler with Python stacks and prints, per aten op that launched a
device kernel NOT written in this repo, the device time and the innermost frames inside the package.  Used to hunt the
cast / copy / add kernels that ride along the hand-written ones ("library kernels still on the path").

    python scripts/prof_small_kernels.py [batch]
"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import basd_amd._native as native  # noqa: E402
from basd_amd.config import load_config  # noqa: E402
from basd_amd.train import SyntheticLoader, build  # noqa: E402


def _instrument():
    """wrap every function / method / autograd.Function.{forward,backward} of the package in a record_function scope
    (this build's profiler returns no Python stacks), so that an aten op can be attributed to its innermost scope"""
    import functools
    import importlib
    import inspect
    import types
    from torch.profiler import record_function
    mods = ["losses.functional", "losses.combined", "losses.layer_selector", "losses.relational", "training.trainer",
            "training.optim", "training.mixup", "training.data_parallel", "models.vit", "models.linear", "models.teacher"]

    def wrap(fn, label):
        @functools.wraps(fn)
        def inner(*a, **k):
            with record_function(label):
                return fn(*a, **k)
        return inner

    for m in mods:
        mod = importlib.import_module("basd_amd." + m)
        for name, obj in list(vars(mod).items()):
            if isinstance(obj, types.FunctionType) and obj.__module__ == mod.__name__:
                setattr(mod, name, wrap(obj, f"basd:{m}.{name}"))
            elif inspect.isclass(obj) and obj.__module__ == mod.__name__:
                for an, av in list(vars(obj).items()):
                    if an.startswith("__") and an != "__call__":
                        continue
                    if isinstance(av, staticmethod):
                        setattr(obj, an, staticmethod(wrap(av.__func__, f"basd:{m}.{name}.{an}")))
                    elif isinstance(av, types.FunctionType):
                        setattr(obj, an, wrap(av, f"basd:{m}.{name}.{an}"))


def _scope_of(ev):
    p, chain = ev.cpu_parent, []
    while p is not None and len(chain) < 2:
        if p.name.startswith("basd:"):
            chain.append(p.name[5:])
        p = p.cpu_parent
    return " <- ".join(chain) or "(no scope)"


def main():
    _instrument()
    batch_size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda", 0)
    native.lib()
    cfg = load_config(os.path.join(ROOT, "vit-bias-aware-structural-distillation_amd", "configs", "config.yaml"), None,
                      [f"data.batch_size={batch_size}", "data.dataset=synthetic", "model.student_preset=deit_tiny_patch16_224",
                       "basd.teacher_model_name=vit_base_patch16_224", "model.vit.img_size=224", "model.vit.patch_size=16",
                       "model.grad_checkpointing=false"])
    trainer, _ = build(cfg, device=dev)
    batch = next(iter(SyntheticLoader(batch_size, 224, cfg.model.num_classes, 1, dev, seed=1234)))
    trainer.optimizer.train()
    trainer.model.train()
    for _ in range(3):
        trainer.train_step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        trainer.train_step(batch)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "device_time_total", 0) or 0
        if dt <= 0 or not ev.kernels:
            continue
        names = [k.name for k in ev.kernels]
        if all("basd::" in n for n in names):
            continue
        key = (ev.name, names[0][:48], _scope_of(ev))
        agg[key][0] += 1
        agg[key][1] += sum(k.duration for k in ev.kernels)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    print(f"non-own device kernels of one eager step: {sum(v[0] for _, v in rows)} launches, {total / 1e3:.2f} ms")
    for (op, kern, where), (n, us) in rows[:110]:
        print(f"{us:9.1f} us {n:4d} x {op:22s} {kern:48s} {where}")


if __name__ == "__main__":
    main()
