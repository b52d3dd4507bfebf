"""Kernel provider used by the loss modules.

The product provider is ``basd_amd._native`` (ctypes over the C-ABI HIP library)
and it is the ONLY provider this package ever selects: there is no CPU
fallback.  ``set_ops`` exists so that the CPU test-suite can exercise the host
logic (autograd formulas, masking, module plumbing) with a test-only emulation
that lives under ``tests/``; nothing in the package calls it.
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


def is_emulated() -> bool:
    """True only inside the CPU test-suite (a test provider was installed with set_ops)."""
    return _ops is not None and getattr(_ops, "__name__", "").endswith("_emul")
