"""BASD trainer: the train step of reference ``src/training/trainer.py:133-164``
on the MI355X path (bf16 ViTs, HIP loss kernels, fused Schedule-Free AdamW,
bucketed RCCL gradient averaging overlapped with backward).

Operator surface kept from the reference: ``_extract_student`` (:16-37) and
``Trainer(student_model, config, accelerator, teacher, *, student_info)`` with
``.model``, ``.optimizer``, ``.basd_loss``, ``.train``, ``.save_checkpoint``,
``.save_weights``, ``.load_checkpoint`` (:41-211).  ``accelerator`` is accepted
for signature compatibility and may be None: mixed precision is
``torch.autocast(bf16)`` and data parallelism is ``GradientReducer``.
"""
from __future__ import annotations

import collections
import contextlib

import os

from collections import defaultdict
from pathlib import Path

import torch
import torch.nn as nn

from ..losses._ops import get_ops
from ..losses.combined import BASDLoss
from ..models.teacher import TeacherModel, extract_intermediates
from .data_parallel import GradientReducer
from .mixup import mixup_cutmix
from .optim import AdamWScheduleFree, FlatParams


def _extract_student(model: nn.Module, x: torch.Tensor, layer_indices, *, layer_paths, has_cls_token: bool):
    """Student forward with token taps at ``layer_indices`` (CLS stripped), reference :16-37."""
    hooks, captured = [], {}
    for idx in layer_indices:
        block = model.get_submodule(layer_paths[idx])

        def make_hook(i):
            def hook(mod, inp, out):
                captured[i] = out[:, 1:, :] if has_cls_token else out
            return hook
        hooks.append(block.register_forward_hook(make_hook(idx)))
    try:
        logits = model(x)
    finally:
        for h in hooks:
            h.remove()
    return logits, captured


class Trainer:
    def __init__(self, student_model: nn.Module, config, accelerator=None, teacher: TeacherModel = None, *,
                 student_info: dict, bucket_bytes: int = 32 << 20):
        self.accelerator = accelerator
        self.config = config
        self.device = next(student_model.parameters()).device
        self.criterion = nn.CrossEntropyLoss(label_smoothing=config.training.label_smoothing)
        self._teacher = teacher
        self._student_layer_paths = student_info["layer_paths"]
        self._student_has_cls = student_info["has_cls_token"]
        self.basd_loss = BASDLoss(
            base_criterion=self.criterion, student_dim=student_info["embed_dim"],
            teacher_dim=teacher.embed_dim, student_depth=student_info["depth"],
            num_student_tokens=student_info["num_tokens"], config=config.basd,
            teacher_has_cls_token=teacher.has_cls_token).to(self.device)
        self.model = student_model
        # one flat buffer: student parameters first, then the selector temperatures
        # (the reference adds them as an extra param group inheriting lr / weight decay, :74-76)
        self.flat = FlatParams(list(student_model.parameters()) + list(self.basd_loss.parameters()))
        self.optimizer = AdamWScheduleFree(self.flat, lr=config.training.learning_rate,
                                           weight_decay=config.training.weight_decay)
        from ..models.linear import BasdLinear
        from ..models.vit import MixedLayerNorm, PatchEmbed
        lin_params = [q for m in student_model.modules() if isinstance(m, (BasdLinear, MixedLayerNorm, PatchEmbed))
                      for q in m.parameters()]
        lin_weights = [m.weight for m in student_model.modules() if isinstance(m, BasdLinear)]
        self.flat.enable_bf16_shadow(lin_params, transposed=lin_weights)
        if config.basd.get("bucket_mb") is not None:          # optional key: all-reduce bucket size in MiB
            bucket_bytes = int(float(config.basd.bucket_mb) * (1 << 20))
        self.reducer = GradientReducer(self.flat, bucket_bytes=bucket_bytes)
        self.reducer.broadcast_parameters()
        self._force_segmented = str(config.basd.get("segmented_backward", "false")).lower() in ("1", "true")
        self.optimizer.z.copy_(self.flat.data)
        self.best_val_acc = 0.0
        self.metrics_history = defaultdict(list)
        self.num_classes = config.model.num_classes
        self.use_mixup = True
        self.autocast_dtype = torch.bfloat16
        self.overlap_teacher_stats = True
        # Two streams.  The teacher branch (forward + its selector statistics) runs on a side stream next to the student
        # forward (per-step schedule) or, pipelined across steps, next to loss / backward of the previous batch.  Round 2
        # saw the GPU DEADLOCK at ViT-B / ViT-H shapes when both streams carried library GEMMs (three runs; no trace of
        # them survives, so the cause -- persistent library kernels spinning on peers that get no CU -- is a hypothesis,
        # not a finding).  The rule that follows from it is structural, not a width heuristic: the main stream always
        # has library GEMMs (the 1000-class head, the fp32 products of the loss backward), so the side stream may exist
        # only if the teacher branch enqueues NONE.  ``_ensure_stream_policy`` runs that branch once on the main stream
        # under a recorder before any two-stream step; with the hand-written GEMM / attention / patch-embedding / wide
        # token-Gram kernels every BASELINE configuration passes.  basd.overlap_teacher_forward / BASD_OVERLAP_TEACHER =
        # false turns the forward overlap off; forcing it on while the branch does use library GEMMs is refused.
        forced = config.basd.get("overlap_teacher_forward", os.environ.get("BASD_OVERLAP_TEACHER"))
        self._overlap_forced = forced is not None and str(forced).lower() in ("1", "true")
        self.overlap_teacher_forward = forced is None or str(forced).lower() in ("auto", "1", "true")
        self._stream_policy_done = False
        self.two_stream_refused = None          # library GEMM call sites of the teacher branch, if any were found
        self._seg_spec = None
        self._seg_pending = None
        self._seg_late = None
        # backward in two stages with the late gradients all-reduced under the second one: OPT-IN
        # (basd.segmented_backward: true).  The path has only ever run with gloo on CPU and on a single GPU rank, where
        # the range all-reduce is a no-op: until one multi-GPU run has compared its gradients with the eager hook path,
        # several ranks use the schedule that HAS been validated -- one captured graph, then one all-reduce of the flat
        # gradient buffer (23 MB for the benchmarked student: ~0.15 ms on xGMI next to a ~36 ms step).
        self.segmented = self._force_segmented
        # debug check of the invariant the two-stage path rests on (stage 2 never writes the late slice of the flat
        # gradient buffer): BASD_CHECK_SEGMENTS=1 compares the slice before / after the second replay
        self._check_segments = os.environ.get("BASD_CHECK_SEGMENTS", "0") == "1"
        # Software pipelining of the frozen teacher ACROSS steps (captured steps only): while loss / backward of batch k
        # run, the side stream computes the teacher forward + selector statistics of batch k + 1 into the other of two
        # held sets, so the step never waits for the teacher and the latency-bound kernels of the loss no longer leave
        # the GPU idle (c2: 45.1 -> 39.5 ms).  Needs the NEXT batch at train_step (``next_batch``); every step still
        # runs one teacher forward.  Multi-layer (ViT) teachers only: a single-layer teacher has no frames to hold.
        self.pipeline_teacher = str(config.basd.get("pipeline_teacher", "true")).lower() in ("1", "true")
        # basd.capture_step (default on for devices): Trainer.train captures the step into a hipGraph on its first batch
        self.capture_step = str(config.basd.get("capture_step", "true")).lower() in ("1", "true")
        self._capture_tried = False
        self._pipe = None
        self.pipeline_error = None
        # The step runs two streams (teacher branch / student + loss).  A fully persistent GEMM launch holds every CU until
        # it ends, and the other stream's short kernels queue behind it (c2, same box: 41.3 ms per step against 39.5 with
        # workgroups that retire every two tiles; alone on the GPU the persistent form is the fastest: fc1 + GELU 288 vs
        # 327 us).  basd.gemm_tile_run overrides (0 = fully persistent).  Round 4, with the shorter loss chain: one tile
        # per workgroup is better again for the pipelined c2 step (32.53 / 32.59 vs 32.89 / 33.00 ms with two, 33.1 with
        # three, same box; per-step schedule 35.89 vs 35.99, c3 equal, c4 54.3 vs 54.0).
        # Scoped (``_tile_scope``): only steps that really run two streams see it; evaluation, inference and steps whose
        # side stream was refused keep the fully persistent default.
        self._gemm_tile_run = int(config.basd.get("gemm_tile_run", 1))
        self._graph_pool = None
        self._side = None
        self._graph = None
        self._graph_tail = None
        self.graph_error = None
        # pinned copies of the kernels' health word still in flight, oldest first: (event, host tensor); read late
        self._status_pending = collections.deque()
        self._status_free = []

    def _tile_scope(self):
        """context of a training step / a capture: the GEMM tile_run of the two-stream schedule while a side stream is in
        use, the fully persistent kernels otherwise"""
        if self.device.type != "cuda":
            return contextlib.nullcontext()
        from .. import _native
        two_streams = self.overlap_teacher_stats and (self.overlap_teacher_forward or self.pipeline_teacher)
        return _native.gemm_tile_run(self._gemm_tile_run if two_streams else 0)

    def _side_stream(self):
        if self.device.type != "cuda" or not self.overlap_teacher_stats:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _ensure_stream_policy(self, clean) -> None:
        """Decide ONCE, before the first two-stream step, whether the teacher branch may have a stream of its own: it is
        run on the current stream under a recorder of library GEMM call sites (losses/_ops.py); any hit serialises the
        step (no side stream, no cross-step pipelining) -- or raises, if the overlap was forced by the configuration."""
        if self._stream_policy_done or self.device.type != "cuda":
            return
        self._stream_policy_done = True
        from ..losses._ops import record_library_gemms
        sel = self.basd_loss.layer_selector
        with record_library_gemms() as seen, torch.no_grad():
            self._teacher_branch(clean)
        sel._frames = None
        torch.cuda.current_stream().synchronize()
        if seen:
            self.two_stream_refused = sorted(seen)
            if self._overlap_forced:
                raise ValueError("basd.overlap_teacher_forward is forced on, but the teacher branch enqueues library GEMMs "
                                 f"({', '.join(self.two_stream_refused)}): two streams of library GEMMs deadlocked the GPU "
                                 "in round 2; refusing")
            self.overlap_teacher_stats = False
            self.overlap_teacher_forward = False
            self.pipeline_teacher = False

    # ------------------------------------------------- backward in two stages
    # With several ranks the gradients are all-reduced (reference: DDP's bucket overlap, src/training/trainer.py:80-82).
    # A captured step cannot run hooks, and a collective inside a hipGraph cannot be rehearsed on a one-GPU box, so the
    # captured backward is CUT instead: stage 1 = loss -> head -> student blocks [cut, depth) (graph A), stage 2 = blocks
    # [0, cut) + patch embedding (graph B).  The late parameters are a contiguous suffix of the flat gradient buffer: their
    # all-reduce is launched between the two replays and flies under stage 2; the rest follows.  The cut sits at the
    # second extraction point, so stage 2 is about a third of the student backward.  Autograd interface of the cut: the
    # output of block cut - 1 (+ the fused pre-normalised tensor that travels with it) and the token taps below the cut.
    def _segment_spec(self):
        if self._seg_spec is None:
            self._seg_spec = False
            paths = self._student_layer_paths
            layers = sorted(self.basd_loss.token_layers)
            cut = layers[1] if len(layers) > 1 else len(paths) // 3
            if 1 <= cut < len(paths) and not getattr(self.model, "grad_checkpointing", False):
                blk = self.model.get_submodule(paths[cut])
                # block cut's first norm is evaluated by the fused add + norm of block cut - 1 (stage 2): the late range
                # starts behind it
                named = [q for name, q in blk.named_parameters() if not name.startswith("norm1.")]
                ids = {id(q): i for i, q in enumerate(self.flat.params)}
                if named and id(named[0]) in ids:
                    first = min(ids[id(q)] for q in named)
                    norm1 = [q for name, q in blk.named_parameters() if name.startswith("norm1.") and id(q) in ids]
                    self._seg_spec = {"cut": cut, "offset": self.flat.offsets[first], "late": self.flat.params[first:],
                                      "norm1": norm1}
        return self._seg_spec or None

    def _student_forward(self, student_imgs, seg):
        """student forward with the token taps; with ``seg`` also the autograd interface of the backward cut"""
        holder, hook = {}, None
        if seg is not None:
            prev = self.model.get_submodule(self._student_layer_paths[seg["cut"] - 1])
            hook = prev.register_forward_hook(lambda m, i, o: holder.__setitem__("out", o))
        try:
            with torch.autocast(device_type=self.device.type, dtype=self.autocast_dtype):
                logits, s_tokens = _extract_student(self.model, student_imgs, self.basd_loss.token_layers,
                                                    layer_paths=self._student_layer_paths,
                                                    has_cls_token=self._student_has_cls)
        finally:
            if hook is not None:
                hook.remove()
        iface = None
        if seg is not None:
            out = holder["out"]
            pre = getattr(out, "_basd_prenorm", None)
            iface = [out] + ([pre[1]] if pre is not None else []) + [s_tokens[l] for l in sorted(s_tokens) if l < seg["cut"]]
            # unfused path: block cut applies its own first norm, i.e. in stage 1
            self._seg_late = (seg["late"] if pre is not None else seg["norm1"] + seg["late"])
        return logits, s_tokens, iface

    def _backward_stage1(self, loss, iface):
        late = self._seg_late
        grads = torch.autograd.grad([loss], late + iface, allow_unused=True)
        for q, g in zip(late, grads[:len(late)]):
            if g is not None:                     # parameters without a gradient sink (head, temperatures, ...)
                q.grad.add_(g.to(q.grad.dtype))
        self._seg_pending = [(t, g) for t, g in zip(iface, grads[len(late):]) if g is not None]

    def _backward_stage2(self):
        tensors, grads = zip(*self._seg_pending)
        self._seg_pending = None
        torch.autograd.backward(list(tensors), list(grads))

    # ------------------------------------------------------------------ step
    def _forward_backward(self, clean, student_imgs, mixed_targets, split=False):
        """teacher fwd -> (side stream) teacher statistics || student fwd -> loss -> backward (``split``: stage 1 only,
        ``_backward_stage2`` finishes it)."""
        self._ensure_stream_policy(clean)
        seg = self._segment_spec() if split else None
        self.flat.refresh_bf16()          # one cast kernel for every Linear weight of the student
        # The frozen teacher and its selector statistics (12 Gram passes + 24 small eigenproblems that
        # occupy 24 of the 256 CUs) are independent of the student forward.  overlap_teacher_forward:
        # the whole teacher branch runs on the side stream next to the student forward (the teacher's
        # MFMA-bound GEMMs and the student's small, memory-bound kernels fill each other's gaps);
        # otherwise only the statistics do.
        side = self._side_stream()
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        selector = self.basd_loss.layer_selector

        def student_forward():
            return self._student_forward(student_imgs, seg)

        if side is not None and self.overlap_teacher_forward:
            # Measured and rejected inside the captured graph (same-box A/B, ms per step): the teacher's Gram passes
            # layer by layer on a third stream +0.9; the same on THIS stream behind the student forward, one event per
            # layer, +3.7.  The graph executor does best with two coarse branches.
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                t_tokens, t_importance = extract_intermediates(self._teacher, clean)
                selector.precompute_teacher(t_tokens)
            logits, s_tokens, iface = student_forward()
        else:
            t_tokens, t_importance = extract_intermediates(self._teacher, clean)
            if side is not None:
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    selector.precompute_teacher(t_tokens)
            logits, s_tokens, iface = student_forward()
        # no join here: the selector waits for the side stream (event recorded by precompute_teacher) only
        # where it consumes the teacher frames, so the student's statistics overlap the teacher's
        loss = self.basd_loss(logits.float(), mixed_targets, s_tokens, t_tokens, t_importance)
        if side is not None and self.basd_loss.layer_selector._frames is not None:
            main.wait_stream(side)               # the loss did not consume the precomputation: join anyway
            self.basd_loss.layer_selector._frames = None
        if side is not None and self.overlap_teacher_forward and not capturing:
            # eager mode: the teacher outputs were allocated on the side stream and are read on this one
            for t in list(t_tokens.values()) + list(t_importance.values()):
                if t is not None:
                    t.record_stream(main)
        if seg is not None:
            self._backward_stage1(loss, iface)
        else:
            loss.backward()
        return loss.detach(), logits.detach()

    # ------------------------------------------------------- teacher pipeline
    def _teacher_branch(self, clean):
        """teacher forward + the teacher half of the selector statistics -> (tokens, importance, (layer indices, frames))"""
        sel = self.basd_loss.layer_selector
        t_tokens, t_importance = extract_intermediates(self._teacher, clean)
        sel.precompute_teacher(t_tokens)
        idx, frames = sel._frames
        sel._frames = None
        return t_tokens, t_importance, (idx, {k: v for k, v in frames.items() if k != "ready"})

    @staticmethod
    def _held_like(src):
        tokens, imp, (idx, frames) = src
        def like(v):
            return torch.empty_like(v) if isinstance(v, torch.Tensor) else v
        return ({k: like(v) for k, v in tokens.items()}, {k: like(v) for k, v in imp.items()},
                (idx, {k: like(v) for k, v in frames.items()}))

    @staticmethod
    def _held_copy(dst, src) -> None:
        for d, s_ in ((dst[0], src[0]), (dst[1], src[1]), (dst[2][1], src[2][1])):
            for k, v in s_.items():
                if isinstance(v, torch.Tensor):
                    d[k].copy_(v)

    def _piped_forward_backward(self, held, out_held, clean_next, student_imgs, mixed_targets, split=False):
        """student fwd -> loss -> backward of the CURRENT batch on the held teacher outputs, while the side stream runs
        the teacher branch of the NEXT batch into ``out_held`` (``split``: backward stage 1 only; the side stream is
        joined here, stage 2 -- a third of the student backward -- runs alone)"""
        seg = self._segment_spec() if split else None
        self.flat.refresh_bf16()
        main, side = torch.cuda.current_stream(), self._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._held_copy(out_held, self._teacher_branch(clean_next))
        tokens, importance, (idx, frames) = held
        self.basd_loss.layer_selector._frames = (idx, dict(frames))
        logits, s_tokens, iface = self._student_forward(student_imgs, seg)
        loss = self.basd_loss(logits.float(), mixed_targets, s_tokens, tokens, importance)
        if seg is not None:
            self._backward_stage1(loss, iface)
        else:
            loss.backward()
        main.wait_stream(side)
        return loss.detach(), logits.detach()

    def _capture(self, fn):
        """capture fn() into a hipGraph; "global" capture mode first, then thread-local (a process group's watchdog
        thread can invalidate a global-mode capture)"""
        last = None
        pool = None if self._graph_pool is None else self._graph_pool
        for mode in ("global", "thread_local"):
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, pool=pool, capture_error_mode=mode):
                    out = fn()
                if self._graph_pool is None:
                    self._graph_pool = graph.pool()
                return graph, out
            except Exception as exc:
                last = exc
                self.basd_loss.layer_selector._frames = None
                torch.cuda.synchronize()
                self.flat.zero_grad()
        raise last

    def _capture_step(self, fn):
        """capture one step: fn(split) -> (loss, logits).  Segmented: graph A = fn(True) (everything up to backward stage
        1), graph B = backward stage 2; otherwise one graph.  -> (graph A, graph B | None, outputs)"""
        if self.segmented and self._segment_spec() is not None:
            ga, out = self._capture(lambda: fn(True))
            gb, _ = self._capture(lambda: (self._backward_stage2(), None)[1])
            return ga, gb, out
        g, out = self._capture(lambda: fn(False))
        return g, None, out

    def _run_step_fn(self, fn):
        """eager execution of a step function with the schedule the capture will have (warm-up passes)"""
        split = self.segmented and self._segment_spec() is not None
        out = fn(split)
        if split:
            self._backward_stage2()
        return out

    def _enable_pipeline(self, warmup: int) -> None:
        """two captured steps that ping-pong between two held sets of teacher outputs"""
        with torch.no_grad():
            held0 = self._teacher_branch(self._g_clean)
        held = [held0, self._held_like(held0)]
        self._g_clean_next = torch.empty_like(self._g_clean)
        self._g_clean_next.copy_(self._g_clean)
        warm = torch.cuda.Stream(device=self.device)
        warm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(warm):
            for i in range(max(warmup, 2)):
                self._run_step_fn(lambda split, i=i: self._piped_forward_backward(
                    held[i & 1], held[1 - (i & 1)], self._g_clean_next, self._g_imgs, self._g_targets, split))
                self.flat.zero_grad()
        torch.cuda.current_stream().wait_stream(warm)
        torch.cuda.synchronize()
        graphs, tails, outs = [], [], []
        for p_ in (0, 1):
            g, gb, out = self._capture_step(lambda split, p_=p_: self._piped_forward_backward(
                held[p_], held[1 - p_], self._g_clean_next, self._g_imgs, self._g_targets, split))
            self.flat.zero_grad()
            graphs.append(g)
            tails.append(gb)
            outs.append(out)
        # held[0] was overwritten by the warm-up / capture passes: nothing is valid until a step fills it
        self._pipe = {"graphs": graphs, "tails": tails, "out": outs, "held": held, "cur": 0, "valid_for": None,
                      "blind": 0, "announced": None}

    def enable_graph(self, batch: dict, warmup: int = 3, pipeline: bool | None = None) -> bool:
        with self._tile_scope():
            return self._enable_graph(batch, warmup, pipeline)

    def _enable_graph(self, batch: dict, warmup: int = 3, pipeline: bool | None = None) -> bool:
        """Capture teacher fwd + student fwd + loss + backward into ONE hipGraph (static input
        buffers).  A step then costs the host one graph launch instead of ~1 300 kernel launches:
        on a shared host the eager step (26 ms of Python/launch work on an idle CPU) becomes
        host-bound as soon as the CPU is contended.  MixUp/CutMix (host RNG), the gradient
        all-reduce (one collective over the flat buffer) and the optimizer step (per-step scalars)
        stay outside the graph.  Returns False (and stays eager) if capture is not possible."""
        if self.device.type != "cuda":
            return False
        try:
            b, c = batch["label"].shape[0], self.num_classes
            self._g_clean = torch.empty_like(batch["clean"])
            self._g_imgs = torch.empty_like(batch["augmented"])
            self._g_targets = torch.empty(b, c, device=self.device)
            self._g_clean.copy_(batch["clean"])
            self._g_imgs.copy_(batch["augmented"])
            self._g_targets.copy_(torch.nn.functional.one_hot(batch["label"], c).float())
            self.reducer.paused = True
            warm = torch.cuda.Stream(device=self.device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):
                for _ in range(warmup):
                    self._run_step_fn(lambda split: self._forward_backward(self._g_clean, self._g_imgs, self._g_targets,
                                                                           split))
                    self.flat.zero_grad()
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            self._graph_pool = None
            self._pipe = None
            want_pipe = (self.pipeline_teacher if pipeline is None else pipeline) and self.two_stream_refused is None
            if want_pipe and len(self._teacher.layer_paths) > 1:
                try:
                    self._enable_pipeline(warmup)
                except Exception as exc:       # the pipelined capture is an optimisation of an optimisation
                    self._pipe, self._graph_pool = None, None
                    self.pipeline_error = f"{type(exc).__name__}: {exc}"
                    self.basd_loss.layer_selector._frames = None
                    torch.cuda.synchronize()
                    self.flat.zero_grad()
            if self._pipe is not None:
                self._graph = self._pipe["graphs"][0]            # "a captured step exists" for the code below
                self._graph_tail = None
                self._g_loss, self._g_logits = self._pipe["out"][0]
            else:
                graph, tail, (self._g_loss, self._g_logits) = self._capture_step(
                    lambda split: self._forward_backward(self._g_clean, self._g_imgs, self._g_targets, split))
                self.flat.zero_grad()
                self._graph, self._graph_tail = graph, tail
            return self._ranks_agree_on_capture()
        except Exception as exc:      # capture is an optimisation: never lose the run over it
            self._graph = None
            self._graph_tail = None
            self._pipe = None
            self.reducer.paused = False
            self.graph_error = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize()
            self.flat.zero_grad()
            if self.reducer.enabled:              # vote "no capture" so that the other ranks drop theirs too
                import torch.distributed as dist
                vote = torch.zeros(3, dtype=torch.int64, device=self.device)
                for op in (dist.ReduceOp.MIN, dist.ReduceOp.MAX):
                    dist.all_reduce(vote.clone(), op=op, group=self.reducer.group)
            return False

    def _ranks_agree_on_capture(self) -> bool:
        """every rank must replay the same schedule (the all-reduce calls of the two-stage replay have to match in number
        and size): if the captures differ between ranks, all of them drop to the eager step"""
        if not self.reducer.enabled:
            return True
        import torch.distributed as dist
        tail = self._pipe["tails"][0] if self._pipe is not None else self._graph_tail
        mine = torch.tensor([1, int(self._pipe is not None), int(tail is not None)], dtype=torch.int64, device=self.device)
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.reducer.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.reducer.group)
        if bool((lo == hi).all()):
            return True
        self._graph = self._graph_tail = self._pipe = None
        self.reducer.paused = False
        self.graph_error = "the ranks captured different schedules: all of them run the eager step"
        return False

    # ---------------------------------------------------------------- health
    def _poll_status(self, drain: bool = False) -> None:
        """Raise (BasdLinAlgError, a torch.linalg.LinAlgError) if a kernel of an EARLIER step flagged non-finite input, a
        Jacobi solve without convergence or a rank-0 teacher layer.  The reference raises from inside torch.linalg with
        a host sync per call; here the flags are OR-ed into one device word which is copied to pinned memory after every
        step.  Copies that have landed are looked at when a step starts, WITHOUT waiting (waiting for the previous
        step's copy would stop the host from enqueueing ahead: 0.6 ms of idle GPU per step at c2); only when two are
        still in flight does the host wait for the older one, so a flag surfaces at most two steps late."""
        while self._status_pending:
            event, host = self._status_pending[0]
            if not (drain or len(self._status_pending) > 2 or event is None or event.query()):
                break
            if event is not None:
                event.synchronize()
            self._status_pending.popleft()
            self._status_free.append(host)
            get_ops().raise_for_status(int(host[0]))

    def check_health(self) -> None:
        """wait for every health word still in flight and raise if any kernel flagged a failure (called at the end of
        an epoch and before a checkpoint is written)"""
        self._poll_status(drain=True)

    def _post_status(self) -> None:
        ops = get_ops()
        word = ops.status_word(self.device)
        if self._status_free:
            host = self._status_free.pop()
        else:
            host = torch.zeros(1, dtype=torch.int32)
            if self.device.type == "cuda":
                host = host.pin_memory()
        host.copy_(word, non_blocking=True)
        word.zero_()
        event = None
        if self.device.type == "cuda":
            event = torch.cuda.Event()
            event.record()
        self._status_pending.append((event, host))
        if event is None:
            self._poll_status()

    def _replay(self, graph, tail) -> None:
        """replay a captured step and average the gradients over the ranks: with a two-stage capture the late parameters'
        all-reduce runs under the second graph, otherwise one all-reduce follows the replay"""
        graph.replay()
        if tail is None:
            self.reducer.reduce_all()
            return
        off = self._segment_spec()["offset"]
        self.reducer.reduce_range_async(off, self.flat.numel)
        late_before = self.flat.grad[off:].clone() if self._check_segments and not self.reducer.enabled else None
        tail.replay()
        if late_before is not None and not torch.equal(late_before, self.flat.grad[off:]):
            raise RuntimeError("two-stage captured backward: stage 2 wrote into the late gradient slice "
                               f"[{off}, {self.flat.numel}) that is being all-reduced under it")
        self.reducer.reduce_range_async(0, off)
        self.reducer.wait_ranges()

    def train_step(self, batch: dict, next_batch: dict | None = None):
        with self._tile_scope():
            return self._train_step(batch, next_batch)

    def _train_step(self, batch: dict, next_batch: dict | None = None):
        """One optimisation step on a device-resident batch {"clean","augmented","label"}.  ``next_batch`` (optional):
        the batch the NEXT call will get -- with a captured, pipelined step its teacher forward runs under this step's
        loss and backward (its "clean" tensor must be the very object passed next time)."""
        self._poll_status()
        clean, student_imgs, targets = batch["clean"], batch["augmented"], batch["label"]
        graph_fits = self._graph is not None and clean.shape == self._g_clean.shape and \
            student_imgs.shape == self._g_imgs.shape
        if self.use_mixup:
            # with a captured step the blend is written straight into the graph's static input buffers
            student_imgs, mixed_targets = mixup_cutmix(student_imgs, targets, self.num_classes,
                                                       out=self._g_imgs if graph_fits else None,
                                                       out_targets=self._g_targets if graph_fits else None)
        else:
            mixed_targets = targets
        if self._graph is not None and (clean.shape != self._g_clean.shape or student_imgs.shape != self._g_imgs.shape):
            # a ragged last batch (or any other shape): the captured step only fits the shapes it was captured with;
            # copy_() would raise or, worse, broadcast a batch of one -- run this step eagerly instead
            hooks_were_paused, self.reducer.paused = self.reducer.paused, False
            try:
                loss, logits = self._forward_backward(clean, student_imgs, mixed_targets)
                self.reducer.finish()
            finally:
                self.reducer.paused = hooks_were_paused
            self._post_status()
            self.optimizer.step()
            self.optimizer.zero_grad()
            return loss, logits
        def held_is_for(t):
            """the held teacher outputs were computed from exactly this tensor: same object AND same version counter (a
            loader that refills one static device buffer in place passes an identity test with different contents)"""
            v = self._pipe["valid_for"]
            return v is not None and v[0] is t and v[1] == t._version

        if self._pipe is not None and not held_is_for(clean) and self._pipe["blind"] >= 2:
            # a caller that never announces the next batch and never repeats one pays a second, unpipelined teacher
            # pass per step: give the pipeline up and capture the per-step schedule instead
            self.enable_graph({"clean": clean, "augmented": batch["augmented"], "label": targets}, pipeline=False)
        if self._pipe is not None:
            pipe = self._pipe
            cur = pipe["cur"]
            if not held_is_for(clean):
                # first step, or the sequence was broken: the teacher branch of THIS batch runs now, unpipelined
                with torch.no_grad():
                    self._held_copy(pipe["held"][cur], self._teacher_branch(clean))
                pipe["blind"] += int(pipe["announced"] is False)
            else:
                pipe["blind"] = 0
            upcoming = clean                      # nothing announced: bet on the same batch object coming again
            pipe["announced"] = next_batch is not None
            if next_batch is not None and next_batch["clean"].shape == self._g_clean_next.shape:
                upcoming = next_batch["clean"]
            self._g_clean_next.copy_(upcoming)
            if student_imgs is not self._g_imgs:
                self._g_imgs.copy_(student_imgs)
            if mixed_targets is not self._g_targets:
                if mixed_targets.dim() == 1:
                    mixed_targets = torch.nn.functional.one_hot(mixed_targets, self.num_classes).float()
                self._g_targets.copy_(mixed_targets)
            self._replay(pipe["graphs"][cur], pipe["tails"][cur])
            loss, logits = pipe["out"][cur]
            pipe["valid_for"], pipe["cur"] = (upcoming, upcoming._version), cur ^ 1
        elif self._graph is not None:
            self._g_clean.copy_(clean)
            if student_imgs is not self._g_imgs:
                self._g_imgs.copy_(student_imgs)
            if mixed_targets is not self._g_targets:
                if mixed_targets.dim() == 1:
                    mixed_targets = torch.nn.functional.one_hot(mixed_targets, self.num_classes).float()
                self._g_targets.copy_(mixed_targets)
            self._replay(self._graph, self._graph_tail)
            loss, logits = self._g_loss, self._g_logits
        elif self._force_segmented and self._segment_spec() is not None:
            # eager two-stage backward (tests; the eager default overlaps per bucket from the gradient hooks instead)
            hooks_were_paused, self.reducer.paused = self.reducer.paused, True
            try:
                loss, logits = self._forward_backward(clean, student_imgs, mixed_targets, split=True)
                off = self._segment_spec()["offset"]
                self.reducer.reduce_range_async(off, self.flat.numel)
                self._backward_stage2()
                self.reducer.reduce_range_async(0, off)
                self.reducer.wait_ranges()
            finally:
                self.reducer.paused = hooks_were_paused
        else:
            loss, logits = self._forward_backward(clean, student_imgs, mixed_targets)
            self.reducer.finish()
        self._post_status()
        self.optimizer.step()
        self.optimizer.zero_grad()
        return loss, logits

    def _train_epoch(self, train_loader):
        total_loss = torch.tensor(0.0, device=self.device)
        correct = torch.tensor(0, device=self.device, dtype=torch.long)
        total = 0
        def on_device(b):
            return None if b is None else {k: v.to(self.device, non_blocking=True) for k, v in b.items()}

        it = iter(train_loader)
        upcoming = on_device(next(it, None))
        while upcoming is not None:
            batch, upcoming = upcoming, on_device(next(it, None))     # one batch of lookahead for the teacher pipeline
            if self._graph is None and self.capture_step and not self._capture_tried and self.device.type == "cuda":
                # the product loop runs the captured (and, for ViT teachers, cross-step pipelined) step: first
                # full-size batch; falls back to eager if the capture fails (graph_error says why)
                self._capture_tried = True
                self.enable_graph(batch)
            loss, logits = self.train_step(batch, upcoming)
            n = batch["label"].size(0)
            total_loss += loss * n
            correct += logits.argmax(1).eq(batch["label"]).sum()
            total += n
        self.check_health()
        return {"train_loss": (total_loss / total).item(), "train_acc": 100.0 * (correct / total).item()}

    def train(self, train_loader, val_loader=None, start_epoch: int = 0, evaluate_fn=None):
        num_epochs = self.config.training.num_epochs
        for epoch in range(start_epoch, num_epochs):
            self.optimizer.train()
            self.model.train()
            # fresh augmentations every epoch: the dual-view dataset seeds sample i's RNG from (seed, epoch, i)
            ds = getattr(train_loader, "dataset", None)
            if hasattr(ds, "set_epoch"):
                ds.set_epoch(epoch)
            metrics = self._train_epoch(train_loader)
            self.optimizer.eval()
            if evaluate_fn is not None and val_loader is not None:
                metrics.update(evaluate_fn(self.model, val_loader))
            print(" ".join([f"epoch {epoch + 1}/{num_epochs}"] + [f"{k}={v:.6f}" for k, v in metrics.items()]))
            for k, v in metrics.items():
                self.metrics_history[k].append(v)
            if metrics.get("val_acc", -1.0) > self.best_val_acc:
                self.best_val_acc = metrics["val_acc"]
                self.save_checkpoint("best_model", epoch)
                self.save_weights("best_model.pth", epoch)
            self.save_checkpoint("latest", epoch)
        self.save_weights("final_model.pth", num_epochs - 1)
        return self.metrics_history

    # ----------------------------------------------------------- checkpoints
    def _ckpt_dir(self) -> Path:
        d = Path(self.config.run.output_dir) / self.config.run.name / "checkpoints"
        d.mkdir(parents=True, exist_ok=True)
        return d

    def save_checkpoint(self, name: str, epoch: int) -> None:
        """``checkpoints/<name>/`` in the layout ``accelerator.save_state`` writes for the reference
        (src/training/trainer.py:94-103 with ``register_for_checkpointing(basd_loss)`` :84; file names of
        accelerate 1.x: ``model.safetensors``, ``optimizer.bin``, ``custom_checkpoint_0.pkl`` = the registered
        BASDLoss state dict, ``random_states_0.pkl``) plus the reference's own ``custom_state.pth``."""
        import random

        self.check_health()
        import numpy as np
        from safetensors.torch import save_file
        d = self._ckpt_dir() / name
        d.mkdir(parents=True, exist_ok=True)
        save_file({k: v.detach().contiguous().cpu() for k, v in self.model.state_dict().items()}, str(d / "model.safetensors"))
        torch.save({k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in self.optimizer.state_dict().items()},
                   d / "optimizer.bin")
        torch.save({k: v.detach().cpu() for k, v in self.basd_loss.state_dict().items()}, d / "custom_checkpoint_0.pkl")
        rng = {"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(),
               "torch_manual_seed": torch.get_rng_state()}
        if self.device.type == "cuda":
            rng["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
        torch.save(rng, d / "random_states_0.pkl")
        torch.save({"epoch": epoch, "best_val_acc": self.best_val_acc,
                    "metrics_history": dict(self.metrics_history)}, d / "custom_state.pth")

    def save_weights(self, filename: str, epoch: int) -> None:
        torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict()}, self._ckpt_dir() / filename)

    def load_checkpoint(self, checkpoint_path: str) -> int:
        """Restore EVERYTHING ``save_checkpoint`` wrote (reference :113-123 -> ``accelerator.load_state``): student
        weights, the loss module's state (``log_temperatures`` and the random-orthogonal ``proj_s`` / ``proj_t``
        buffers -- a resume with another seed must not change the selector's projections), optimizer state, RNG
        streams; then the bf16 shadow of the weights is refreshed.  Tensors are loaded with ``weights_only=True``."""
        import random

        import numpy as np
        from safetensors.torch import load_file
        d = Path(checkpoint_path)
        self.model.load_state_dict(load_file(str(d / "model.safetensors"), device=str(self.device)))
        self.basd_loss.load_state_dict(torch.load(d / "custom_checkpoint_0.pkl", map_location=self.device,
                                                  weights_only=True))
        # parameters are views of the flat buffer: the two loads above already wrote into it; the optimizer state
        # carries the same buffer (train-mode "y" point) plus z, v and the step counters
        self.optimizer.load_state_dict(torch.load(d / "optimizer.bin", map_location=self.device, weights_only=True))
        self.flat.refresh_bf16()
        rng_file = d / "random_states_0.pkl"
        if rng_file.exists():
            rng = torch.load(rng_file, map_location="cpu", weights_only=False)   # python / numpy state tuples
            random.setstate(rng["random_state"])
            np.random.set_state(rng["numpy_random_seed"])
            torch.set_rng_state(rng["torch_manual_seed"])
            if self.device.type == "cuda" and "torch_cuda_manual_seed" in rng:
                torch.cuda.set_rng_state_all(rng["torch_cuda_manual_seed"])
        custom = torch.load(d / "custom_state.pth", map_location=self.device, weights_only=True)
        self.best_val_acc = custom["best_val_acc"]
        self.metrics_history = defaultdict(list, custom["metrics_history"])
        return custom["epoch"] + 1
