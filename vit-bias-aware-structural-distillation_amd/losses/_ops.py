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
# in the environment (or ``set_strict(True)``), it raises instead.  Layers that are declared library calls (the
# 1000-class head: N = 1000 on 256 CLS rows, 0.05 ms) carry ``library_ok = True`` and do not count.
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


class record_library_gemms:
    """``with record_library_gemms() as seen:`` -> set of the library GEMM call sites reached inside the block"""

    def __enter__(self):
        self.seen = set()
        _GEMM_RECORDERS.append(self.seen)
        return self.seen

    def __exit__(self, *exc):
        _GEMM_RECORDERS.remove(self.seen)
        return False


def library_fallback(what: str, detail: str = "") -> None:
    """Call right before running a library kernel in place of a hand-written one (device tensors only)."""
    FALLBACKS[what] += 1
    note_library_gemm(what)
    if STRICT:
        raise StrictModeError(f"BASD_STRICT: {what} would run on a library kernel instead of the HIP path ({detail})")
