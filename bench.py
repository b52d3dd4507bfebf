#!/usr/bin/env python3
"""BASD train-step benchmark (BASELINE.json metric): images/sec of one full
distillation step -- MixUp/CutMix, student fwd (token taps), frozen teacher fwd
(token + importance taps), BASD loss (selector, mixing, Procrustes, CE, UW-SO),
backward (+ RCCL gradient averaging), fused Schedule-Free AdamW, zero_grad -- on
synthetic device-resident 224x224 batches, DeiT-Tiny student / ViT-Base/16
teacher, 256 images per GPU (weak scaling: the reference's batch_size is per
process).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description), with
``roofline`` for the dominant hand-written kernel (timed with device events on
the launch stream inside the timed region) and ``cpu_baseline`` (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = os.path.join(ROOT, "vit-bias-aware-structural-distillation_amd", "configs", "config.yaml")

# BASELINE.json configs: (student preset, teacher preset, image size, student patch, per-GPU batch,
# fwd FLOPs per image of student / teacher (2*MAC), description).  c2 is the configuration the metric is
# quoted on and the default; the others are run with --config.
CONFIGS = {
    "c1": ("deit_tiny_patch16_224", "vit_small_patch16_224", 32, 4, 64, 0.73e9, 2.84e9,
           "BASELINE configs[0]: DeiT-Tiny student, ViT-Small teacher, 32x32 images, patch 4"),
    "c2": ("deit_tiny_patch16_224", "vit_base_patch16_224", 224, 16, 256, 2.51e9, 35.13e9,
           "BASELINE configs[1]: DeiT-Tiny/16 student, ViT-Base/16 teacher"),
    "c3": ("deit_tiny_patch16_224", "resnet50", 224, 16, 256, 2.51e9, 8.2e9,
           "BASELINE configs[2]: DeiT-Tiny/16 student, ResNet-50 teacher (one layer of 49 tokens x 2048 channels, uniform "
           "importance; the frozen conv trunk runs through MIOpen as a black box)"),
    "c4": ("deit_small_patch16_224", "vit_large_patch16_224", 224, 16, 128, 9.20e9, 123.1e9,
           "BASELINE configs[3]: DeiT-Small/16 student, ViT-Large/16 teacher (128 images per GPU of the global 1024)"),
    "c5": ("vit_base_patch16_224", "vit_huge_patch14_224", 224, 16, 256, 35.13e9, 334.6e9,
           "BASELINE configs[4]: ViT-Base/16 student, ViT-Huge/14 teacher (256 images per GPU of the global 2048)"),
}


def _nbytes(t):
    return t.numel() * t.element_size()


# ALGORITHMIC bytes of one launch of the streaming kernels (every captured teacher token read once, every
# tapped student token once, the E mixed outputs written once), from the arguments / results of the call:
#   mix_tokens      reads the L teacher layers once, writes the E mixed [B, N, D_t] fp32 tensors
#   mix_grad_dots   reads the L teacher layers once and the E gradient tensors once (the [E, L] dots are negligible)
#   procrustes_prep reads the student tokens, the mixed teacher tokens and the importance once, writes s_w and t_w
#   add_layernorm_fwd (the frozen teacher's fused residual add + LayerNorm) reads x and the residual, writes s and y
STREAMING = {
    "mix_tokens": lambda a, k, out: sum(_nbytes(t) for t in a[0]) + _nbytes(out),
    "mix_grad_dots": lambda a, k, out: sum(_nbytes(t) for t in a[0]) + _nbytes(a[1]),
    "procrustes_prep": lambda a, k, out: _nbytes(a[0]) + a[1].numel() * 4 + _nbytes(a[2]) + _nbytes(out[0]) + _nbytes(out[1]),
    "add_layernorm_fwd": lambda a, k, out: 2 * _nbytes(a[0]) + _nbytes(out[0]) + _nbytes(out[1]),
}


class KernelTimer:
    """Device-event timing of the C-ABI kernels, on the stream they are launched on."""

    def __init__(self, native, names):
        self.native, self.names = native, names
        self.records = {n: [] for n in names}
        self.meta = {n: [] for n in names}
        self._orig = {}
        self.active = False

    def install(self):
        for name in self.names:
            fn = getattr(self.native, name)
            self._orig[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **k):
                if not self.active:
                    return _fn(*a, **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = _fn(*a, **k)
                e.record()
                self.records[_name].append((s, e))
                if _name in ("gemm_bf16", "gemm_gelu_fwd", "gemm_gelu_bwd"):
                    # (rows, N, K) of y = x w^T: a[0] [..., K], a[1] [N, K]
                    self.meta[_name].append((a[0].numel() // a[0].shape[-1], a[1].shape[0], a[1].shape[1], s, e))
                if _name in STREAMING:
                    label = f"{_name}[D={a[0].shape[-1]}]" if _name == "add_layernorm_fwd" else _name
                    self.meta[_name].append((STREAMING[_name](a, k, out), s, e, label))
                if _name == "jacobi_svd":
                    # (batch, n_cols, m_rows, rank-masked?) -- masked launches sweep a smaller block
                    self.meta[_name].append((a[0].shape[0], a[0].shape[1], a[1], k.get("active") is not None, s, e, out[1]))
                return out
            setattr(self.native, name, wrapped)

    def launch_procrustes_entry_by_entry(self, on: bool):
        """the Procrustes forward is ONE C call (basd_procrustes_fwd) in the product path: nothing inside it can be
        bracketed by events.  For the instrumented eager steps the same kernels are launched entry by entry
        (losses/procrustes_chain.py: identical launches, identical order), so that the wrappers above see them."""
        from basd_amd.losses.procrustes_chain import procrustes_fwd_chain
        if on:
            self._orig["procrustes_fwd"] = self.native.procrustes_fwd
            self.native.procrustes_fwd = lambda s_w, t_w, tol=1e-13: procrustes_fwd_chain(self.native, s_w, t_w, tol)
        elif "procrustes_fwd" in self._orig:
            self.native.procrustes_fwd = self._orig.pop("procrustes_fwd")

    def summary(self):
        out = {}
        for name, evs in self.records.items():
            if evs:
                ms = [s.elapsed_time(e) for s, e in evs]
                out[name] = {"launches": len(ms), "total_ms": sum(ms), "avg_ms": sum(ms) / len(ms)}
        return out


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh child processes of this script, one rank per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would: the reference gets its ranks from
    ``accelerate launch``, src/training/trainer.py:80-82), relay rank 0's stdout (the JSON line) and return non-zero if
    any rank failed.  The parent never touches the GPU (no torch.cuda call before this point), so nothing that holds a
    device context is ever re-executed."""
    import subprocess
    import tempfile
    port = _free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BASD_BENCH_LAUNCHER="self")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        rc = 0
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:                                  # a rank died: the others would wait in a collective for ever --
                rc = bad[0] if bad[0] > 0 else 1     # stop exactly the children started here
                for pr in procs:
                    if pr.poll() is None:
                        pr.kill()
                for pr in procs:
                    pr.wait()
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.2)
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    return rc


def launch_check() -> None:
    """--launch-check: what a rank does when only the rendezvous is under test (CPU, gloo): initialise the process
    group the launcher's environment describes, all-reduce one value, rank 0 prints a JSON line with the world size
    torch.distributed actually sees (tests/test_bench_launcher_cpu.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen, total = 1, 1.0
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen, total = dist.get_world_size(), float(t)
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "rccl_ranks": seen, "backend": "gloo",
                          "all_reduce_of_ones": total, "launcher": os.environ.get("BASD_BENCH_LAUNCHER", "external")}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launch-check", action="store_true", help="rendezvous rehearsal on CPU (gloo): no GPU work")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2", help="BASELINE.json configuration (default: c2, "
                    "the one the metric is quoted on)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=256, help="batch of the CPU baseline sample (default: the metric's 256 images: one warm-up + one timed CPU step, ~40 s each on 16 cores)")
    ap.add_argument("--global-batch", type=int, default=None, help="STRONG scaling: fix the global batch (e.g. 256) and give every rank global / world images; default: weak scaling, --batch images per GPU")
    ap.add_argument("--grad-checkpointing", action="store_true")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE", help="further config overrides (A/B "
                    "experiments, e.g. --set basd.gemm_tile_run=0); the default line uses none")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into a hipGraph")
    ap.add_argument("--no-ab", action="store_true", help="skip the second, unpipelined measurement (profiling runs)")
    ap.add_argument("--no-pipeline", action="store_true", help="do not overlap the teacher forward of batch k + 1 with "
                    "loss / backward of batch k (every step then starts with its own teacher forward)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (before any torch.cuda call)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} ranks")
    if args.launch_check:
        return launch_check()
    student_preset, teacher_preset, img_size, patch, cfg_batch, F_STUDENT, F_TEACHER, workload = CONFIGS[args.config]
    if args.batch is None:
        args.batch = cfg_batch
    scaling = "weak"
    if args.global_batch is not None:
        w_ = int(os.environ.get("WORLD_SIZE", "1"))
        if args.global_batch % w_:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by the world size {w_}")
        args.batch = args.global_batch // w_
        scaling = "strong"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path for the product kernels)")
    # BASD_DIST_BACKEND=gloo + several ranks on one GPU is a rehearsal mode for boxes with a single
    # device (the driver's multi-GPU runs use nccl = RCCL, one rank per GPU)
    backend = os.environ.get("BASD_DIST_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_dev)

    import basd_amd._native as native
    from basd_amd.config import load_config
    from basd_amd.train import SyntheticLoader, build
    native.lib()

    cfg = load_config(CFG, None, [f"data.batch_size={args.batch}", "data.dataset=synthetic",
                                  f"model.student_preset={student_preset}", f"basd.teacher_model_name={teacher_preset}",
                                  f"model.vit.img_size={img_size}", f"model.vit.patch_size={patch}",
                                  f"model.grad_checkpointing={'true' if args.grad_checkpointing else 'false'}"]
                      + ([f"basd.teacher_patch_size={patch}"] if args.config == "c1" else []) + list(args.set))
    def progress(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    trainer, info = build(cfg, device=dev)
    from basd_amd.losses import _ops as basd_ops
    basd_ops.FALLBACKS.clear()          # the start-up probes (one dummy image through each model, fp32) do not count
    progress(f"built {args.config}: {student_preset} / {teacher_preset}, {args.batch} images per GPU")
    # four different synthetic batches, cycled through the warm-up and timed steps: no step sees the data of the
    # previous one (nothing could be reused across steps, and the data-dependent Jacobi sweep counts vary); the
    # instrumented probe steps use the first
    batches = [next(iter(SyntheticLoader(args.batch, cfg.model.vit.img_size, cfg.model.num_classes, 1, dev,
                                         seed=1234 + rank + 1000 * i))) for i in range(4)]
    batch = batches[0]
    trainer.optimizer.train()
    trainer.model.train()

    timer = KernelTimer(native, ["jacobi_svd", "pchol", "trinv", "bgemm_f64", "token_gram", "mix_tokens",
                                 "mix_grad_dots", "procrustes_prep", "wgrad_bf16", "sf_adamw_step", "mp_rank",
                                 "gemm_bf16", "gemm_gelu_fwd", "gemm_gelu_bwd", "add_layernorm_fwd"])
    timer.install()

    # a few eager steps first: rank sanity check + (events cannot be recorded inside a captured graph)
    # the per-kernel device-event timings used for the roofline object
    eager_probe = 3
    timer.active = True
    timer.launch_procrustes_entry_by_entry(True)
    for i in range(eager_probe):
        loss, _ = trainer.train_step(batch)
        torch.cuda.synchronize()
        progress(f"eager step {i}: loss {float(loss):.4f}")
        if i == 0:
            timer.active = False
            # the instrumented steps run on ONE stream (the product schedule puts the teacher branch on a side stream:
            # every kernel then shares the GPU and a device-event bracket measures the sharing, not the kernel)
            stream_flags = (trainer.overlap_teacher_stats, trainer.overlap_teacher_forward)
            trainer.overlap_teacher_stats = trainer.overlap_teacher_forward = False
            ranks = trainer.basd_loss.layer_selector.subspace_ranks
            if min(ranks.values()) < 1 or not torch.isfinite(loss):
                raise SystemExit(f"synthetic batch gives a rank-0 teacher layer / non-finite loss: {ranks} {loss}")
            for n_ in timer.records:          # drop the first (cold) step from the statistics
                timer.records[n_].clear()
                timer.meta[n_].clear()
            timer.active = True
    torch.cuda.synchronize()
    timer.active = False
    trainer.overlap_teacher_stats, trainer.overlap_teacher_forward = stream_flags       # product schedule from here on
    timer.launch_procrustes_entry_by_entry(False)      # the captured / timed steps use the composite entry
    probe_steps = eager_probe - 1
    graphed = False
    if not args.eager:
        graphed = trainer.enable_graph(batch, pipeline=False if args.no_pipeline else None)
        progress(f"hipGraph capture: {'ok' if graphed else 'failed: ' + str(trainer.graph_error)}")

    def batch_at(i):
        return batches[i % len(batches)]

    # train_step(batch, next_batch): with the pipelined captured step the teacher forward + statistics of the NEXT batch
    # run on the side stream under loss / backward of this one (software pipelining; the warm-up steps fill the pipe)
    for i in range(args.warmup):
        loss, _ = trainer.train_step(batch_at(i), batch_at(i + 1))
    torch.cuda.synchronize()
    progress(f"{args.warmup} warm-up steps done; timing {args.steps} steps")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    # the timed steps run the product path untouched (hipGraph replay, or eager with the composite C entries): the
    # per-kernel event timings of the roofline object come from the instrumented probe steps above
    timer.active = False
    t0 = time.perf_counter()
    host_s = 0.0
    for i in range(args.steps):
        th = time.perf_counter()
        loss, _ = trainer.train_step(batch_at(args.warmup + i), batch_at(args.warmup + i + 1))
        host_s += time.perf_counter() - th      # time to ENQUEUE a step (no device sync inside)
    barrier()
    elapsed = time.perf_counter() - t0
    timer.active = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = 1e3 * elapsed / args.steps
    global_batch = args.batch * world
    value = global_batch * args.steps / elapsed
    pipelined = trainer._pipe is not None
    unpipelined = None
    if pipelined and world == 1 and not args.no_ab:
        # the same K steps with every step waiting for its own teacher forward (the schedule of rounds 1 - 2), for
        # reference: re-capture without the pipeline, re-warm, time
        if trainer.enable_graph(batch, pipeline=False):
            for i in range(max(args.warmup, 2)):
                trainer.train_step(batch_at(i), None)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                trainer.train_step(batch_at(args.warmup + i), None)
            torch.cuda.synchronize()
            e1 = time.perf_counter() - t1
            unpipelined = {"ms_per_step": 1e3 * e1 / args.steps, "value": global_batch * args.steps / e1}

    if rank == 0:
        ks = timer.summary()
        # ---- roofline of the dominant hand-written kernel: the register-resident one-sided Jacobi (block ordering).
        # It is neither HBM- nor MFMA-bound (VALU-issue-bound); it is priced against the
        # fp32 vector/matrix peak (157.3 TF, equal on gfx950) with ALGORITHMIC flops =
        # sweeps * n(n-1)/2 pairs * 14 m flops (3 dots of length m + a 4-FMA rotation of two
        # columns), sweeps = what every matrix of the launch actually ran.
        roof = None
        if "jacobi_svd" in ks:
            flops, tot_ms, launches, sweep_sum, mats = 0.0, 0.0, 0, 0.0, 0
            big = max(b for (b, *_rest) in timer.meta["jacobi_svd"])
            big_n = max(n for (b, n, *_rest) in timer.meta["jacobi_svd"] if b == big)
            for (b, n, m, masked, ev_s, ev_e, sweeps_t) in timer.meta["jacobi_svd"]:
                if masked or b != big or n != big_n:
                    continue          # dominant launch only: the E*B Procrustes cores (the small selector launches
                                      # run on a side stream / sweep data-dependent blocks)
                # sweeps actually run by every matrix of the launch (the kernel returns them): the work DONE,
                # not a nominal count -- the convergence test decides how many sweeps the algorithm needs
                sw = float(sweeps_t.float().abs().sum())
                flops += sw * (n * (n - 1) / 2) * 14.0 * m
                sweep_sum += sw
                mats += b
                tot_ms += ev_s.elapsed_time(ev_e)
                launches += 1
            achieved = flops / (tot_ms / 1e3) / 1e12
            # HBM bytes per launch of the biggest Jacobi launch (E*B matrices) from the committed PMC passes
            # (profiles/r01_pmc_hbm_traffic.json; bench.py cannot read PMC counters itself)
            traffic = None
            if big == 1024 and big_n == 192:      # the committed PMC passes were taken at this launch shape
                try:
                    with open(os.path.join(ROOT, "profiles", "r04_pmc_hbm_traffic.json")) as f:
                        for key, v in json.load(f)["kernels"].items():
                            if key.startswith("basd::jacobi_b6_kernel<3") and key.endswith(f"grid={big * 256}"):
                                traffic = v["hbm_bytes_per_launch"]
                except OSError:
                    pass
            kname = ("basd::jacobi_b6_kernel<3, 2> (register-resident one-sided Jacobi: hex-block odd-even ordering, scaled "
                     "rotations" if big_n <= 192 else
                     "basd::jacobi_b4_kernel<4, 16> (register-resident one-sided Jacobi: quad-block odd-even ordering, scaled "
                     "rotations")
            roof = {"kernel": f"{kname}, the E*B = {big} Procrustes cores of a step, {big_n}x{big_n} each)",
                    # this kernel is VALU-issue-bound (neither of the contract's "hbm" | "mfma"): priced against the fp32
                    # vector peak, which equals the fp32 MFMA peak on gfx950 (157.3 TF)
                    "bound": "valu", "bound_detail": "fp32 VALU kernel priced against the fp32 vector peak (= fp32 MFMA peak, 157.3 TF/s); "
                    "per round of six independent rotations and wave: 108 v_pk_fma/mul_f32 (36 dot-product + 72 shear updates) + "
                    "~115 other instructions (reduce-scatter of the six dot products, ONE rotation-parameter stream for six "
                    "rotations per slot, 16 LDS-crossbar moves), one LDS hand-over and one barrier per 36 rotations; four "
                    "waves per matrix, two matrices per CU",
                    "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s",
                    "frac": achieved / 157.3, "traffic": traffic,
                    "avg_launch_ms": tot_ms / launches, "launches_per_step": launches / probe_steps,
                    "ms_per_step": tot_ms / probe_steps,
                    "measured_in": (f"{probe_steps} instrumented eager steps of the same process before the timed region "
                                    "(the timed steps replay one hipGraph; device events cannot be recorded inside it, nor inside "
                                    "basd_procrustes_fwd: the probe steps launch that entry's kernels one by one; "
                                    "rocprofv3 of the same command sees the graph-launched kernels: profiles/)"),
                    "mean_sweeps": sweep_sum / max(mats, 1),
                    "traffic_source": "constant from the committed PMC passes (profiles/r04_pmc_hbm_traffic.json), not "
                                      "measured in this run" if traffic is not None else None,
                    "note": "VALU kernel priced against the fp32 vector = matrix peak; algorithmic (textbook) "
                            "flops = sweeps actually run (returned per matrix by the kernel) x n(n-1)/2 pairs x 14 m (three "
                            "dot products of length m + a four-FMA rotation of two columns); the kernel EXECUTES ~6 m per "
                            "pair (one dot product by the incremental norms, two shears of 2 m instead of the 4 m rotation), "
                            "i.e. 0.43 x this figure (DESIGN.md section 5)"}
        # ---- the kernel family with the largest TOTAL time: the bf16 GEMM of the ViT blocks (teacher forward, student
        # forward + input gradients), algorithmic flops 2 M N K over the device-event time of the same probe steps
        gemm = None
        g_flops, g_ms, g_n = 0.0, 0.0, 0
        by_shape = {}
        for name in ("gemm_bf16", "gemm_gelu_fwd", "gemm_gelu_bwd"):
            for (rows, n_, k_, ev_s, ev_e) in timer.meta.get(name, []):
                ms = ev_s.elapsed_time(ev_e)
                g_flops += 2.0 * rows * n_ * k_
                g_ms += ms
                g_n += 1
                rec = by_shape.setdefault(f"{rows}x{n_}x{k_}", [0, 0.0, 2.0 * rows * n_ * k_])
                rec[0] += 1
                rec[1] += ms
        if g_n:
            ach = g_flops / (g_ms / 1e3) / 1e12
            top = sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:4]
            gemm = {"kernel": "basd::gemm_bf16_pring_kernel / gemm_bf16_nt_kernel (bf16 MFMA 16x16x32, LDS-DMA operand ring)",
                    "bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                    "launches_per_step": g_n / probe_steps, "ms_per_step": g_ms / probe_steps,
                    "largest_shapes": {k_: {"launches_per_step": v_[0] / probe_steps, "avg_us": 1e3 * v_[1] / v_[0],
                                            "tflops": v_[2] * v_[0] / (v_[1] / 1e3) / 1e12} for k_, v_ in top},
                    "measured_in": "device events around every GEMM entry in the instrumented eager probe steps; those "
                                   "steps run teacher and student on ONE stream (overlap switched off for them), so each "
                                   "GEMM has the GPU to itself; eager launches still carry a few us of launch latency each"}
        # ---- streaming stages (>= 60 % of the 8 TB/s HBM peak is the target): algorithmic bytes of the
        # launch / device-event time, single stream
        streaming = {}
        by_label = {}
        for name in STREAMING:
            for r in timer.meta.get(name, []):
                by_label.setdefault(r[3], []).append(r)
        for name, recs in by_label.items():
            if recs:
                by, ms = sum(r[0] for r in recs), sum(r[1].elapsed_time(r[2]) for r in recs)
                streaming[name] = {"launches_per_step": len(recs) / probe_steps, "algorithmic_bytes_per_launch": by / len(recs),
                                   "avg_us": 1e3 * ms / len(recs), "achieved": by / (ms / 1e3) / 1e9, "peak": 8000.0,
                                   "unit": "GB/s", "frac": by / (ms / 1e3) / 1e9 / 8000.0}
        if streaming:
            tot_b = sum(v["algorithmic_bytes_per_launch"] * v["launches_per_step"] for v in streaming.values())
            tot_s = sum(v["avg_us"] * v["launches_per_step"] for v in streaming.values()) / 1e6
            streaming["all"] = {"bound": "hbm", "achieved": tot_b / tot_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                                "frac": tot_b / tot_s / 1e9 / 8000.0, "bytes_per_step": tot_b,
                                "measured_in": "device events in the single-stream instrumented probe steps"}
        vit_flops = global_batch * ((4 if args.grad_checkpointing else 3) * F_STUDENT + F_TEACHER)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_step import cpu_step_images_per_sec
            print("[bench] GPU timing done; timing the CPU baseline sample", file=sys.stderr, flush=True)
            # the metric's own batch (256): one warm-up step + one timed step (~2 x 40 s of CPU work on the 16-core share)
            small = args.cpu_batch < 128
            cpu = cpu_step_images_per_sec(batch=args.cpu_batch, timed_steps=3 if small else 1, warmup=1)
        line = {
            "metric": ("images/sec BASD train step, DeiT-T student / ViT-B teacher bs=256" if args.config == "c2" else
                       f"images/sec BASD train step, {student_preset} student / {teacher_preset} teacher"),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{workload} (random init), {img_size}x{img_size}, E=4 extraction points, "
                                   "loss linalg fp32/fp64",
                       "global_batch": global_batch, "per_gpu_batch": args.batch, "parallelism": f"dp{world}",
                       "grad_checkpointing": bool(args.grad_checkpointing)},
            "roofline": roof,
            "roofline_top_by_total_time": gemm,
            "roofline_streaming": streaming,
            "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "launcher": os.environ.get("BASD_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "none"),
            "cpu_baseline": cpu,
            "library_fallbacks": dict(__import__("basd_amd.losses._ops", fromlist=["FALLBACKS"]).FALLBACKS),
            "vit_gemm": {"algorithmic_tflop_per_step": vit_flops / 1e12,
                         "tflops_if_whole_step": vit_flops / (ms_per_step / 1e3) / 1e12, "peak_bf16": 2500.0},
            "kernel_ms_per_step": {k: v["total_ms"] / probe_steps for k, v in ks.items()},
            "hip_graph": graphed, "hip_graph_error": trainer.graph_error, "pipeline_error": trainer.pipeline_error,
            "schedule": ("teacher forward + selector statistics of batch k+1 on a side stream under loss / backward of "
                         "batch k (software pipelining across steps; every timed step still runs one teacher forward, one "
                         "student forward / backward, one loss, one optimizer step; results equal the unpipelined "
                         "schedule: tests/test_train_step_gpu.py)" if pipelined else
                         "every step: [teacher forward || student forward] -> loss -> backward -> optimizer"),
            "unpipelined": unpipelined,
            "peak_hbm_gb": torch.cuda.max_memory_allocated(dev) / 2**30,
            "host_enqueue_ms_per_step": 1e3 * host_s / args.steps,
            "loss": float(loss),
            "teacher_ranks": list(trainer.basd_loss.layer_selector.subspace_ranks.values()),
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
