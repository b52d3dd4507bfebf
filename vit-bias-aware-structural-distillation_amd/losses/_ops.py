"""Kernel provider used by the loss modules.

The product provider is ``basd_amd._native`` (ctypes over the C-ABI HIP library)
and it is the ONLY provider this package ever selects: there is no CPU
fallback.  ``set_ops`` is the injection point the CPU test-suite uses to exercise
the host logic (autograd formulas, masking, module plumbing) with its own provider
(``tests/_emul.py``); nothing in the package calls it or knows about that provider.
A provider answers ``handles(tensor)``: whether its kernels take this tensor (the HIP
library: device tensors only).
"""
from __future__ import annotations

_ops = None


def get_ops():
    global _ops
    if _ops is None:
        from .. import _native
        _native.lib()        # raises if the HIP library is missing
        _ops = _native
    return _ops


def set_ops(provider) -> None:
    """Tests only."""
    global _ops
    _ops = provider


# ---------------------------------------------------------------------------------------------------------------------
# Strict mode.  A layer whose shape the hand-written kernels do not tile falls back to a library kernel (hipBLASLt /
# aten): correct, but not the path BASELINE.json's north_star asks for, and silent.  Every such fallback on DEVICE
# tensors goes through ``library_fallback``: it is counted (``FALLBACKS``, reported by bench.py) and, with BASD_STRICT=1
# in the environment (or ``set_strict(True)``), it raises instead.  Declared library calls do not count: the 1000-class
# head is a plain ``nn.Linear`` (N = 1000 on 256 CLS rows, 0.05 ms; it never passes through ``BasdLinear``), the fp32
# products of the loss backward report through ``note_library_gemm`` only.
import collections
import os

STRICT = os.environ.get("BASD_STRICT", "0") == "1"
FALLBACKS: "collections.Counter[str]" = collections.Counter()


class StrictModeError(RuntimeError):
    pass


def set_strict(on: bool) -> None:
    global STRICT
    STRICT = bool(on)


_GEMM_RECORDERS: list = []


def note_library_gemm(what: str) -> None:
    """A library GEMM (hipBLASLt / rocBLAS behind torch.matmul, F.linear, bmm) is about to be enqueued on the current
    stream.  Declared library calls (the head, the fp32 products of the loss backward) report here without counting as
    fallbacks; the trainer uses the record to decide whether two streams may run side by side (trainer.py,
    ``_ensure_stream_policy``)."""
    for rec in _GEMM_RECORDERS:
        rec.add(what)


# aten operators that end in a library GEMM / fused-attention kernel on device tensors (matmul, einsum, F.linear and
# the @ operator decompose into these below the dispatcher)
_LIBRARY_GEMM_OPS = frozenset({
    "mm", "bmm", "addmm", "baddbmm", "addbmm", "addmv", "mv", "dot", "linear", "matmul", "_addmm_activation",
    "_scaled_mm", "_scaled_dot_product_flash_attention", "_scaled_dot_product_efficient_attention",
    "_scaled_dot_product_attention_math", "_flash_attention_forward", "_efficient_attention_forward",
})


def _device_gemm_probe(seen: set):
    """A TorchDispatchMode that records every aten GEMM-class operator that runs on a device tensor: the STRUCTURAL
    part of the two-stream gate (an un-annotated ``torch.matmul`` added to the teacher branch later is seen as well;
    the hand annotations only give the records readable names)."""
    import torch
    from torch.utils._python_dispatch import TorchDispatchMode

    def on_device(x):
        if isinstance(x, torch.Tensor):
            return x.device.type != "cpu"
        if isinstance(x, (list, tuple)):
            return any(on_device(y) for y in x)
        return False

    class _Probe(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = func._schema.name.split("::")[-1]
            if name in _LIBRARY_GEMM_OPS and (on_device(args) or on_device(tuple((kwargs or {}).values()))):
                seen.add(f"aten::{name}")
            return func(*args, **(kwargs or {}))

    return _Probe()


class record_library_gemms:
    """``with record_library_gemms() as seen:`` -> set of the library GEMM call sites reached inside the block: the
    annotated ones by name (``note_library_gemm`` / ``library_fallback``) plus every aten GEMM-class operator the
    dispatcher sees on a device tensor (``aten::mm`` ...), annotated or not."""

    def __enter__(self):
        self.seen = set()
        _GEMM_RECORDERS.append(self.seen)
        self._probe = _device_gemm_probe(self.seen)
        self._probe.__enter__()
        return self.seen

    def __exit__(self, *exc):
        self._probe.__exit__(*exc)
        _GEMM_RECORDERS.remove(self.seen)
        return False


def library_fallback(what: str, detail: str = "") -> None:
    """Call right before running a library kernel in place of a hand-written one (device tensors only)."""
    FALLBACKS[what] += 1
    note_library_gemm(what)
    if STRICT:
        raise StrictModeError(f"BASD_STRICT: {what} would run on a library kernel instead of the HIP path ({detail})")
