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
