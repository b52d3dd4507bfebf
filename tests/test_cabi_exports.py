"""The C-ABI library loads on a CPU-only machine and exports every symbol that
include/basd_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_agree():
    import basd_amd._native as native
    if not os.path.exists(native.LIB_PATH):
        native.build()
    lib = ctypes.CDLL(native.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "basd_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(basd_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in basd_hip.h but not exported"
    assert declared == set(native.EXPORTS)
    lib.basd_version.restype = ctypes.c_int
    assert lib.basd_version() >= 100
    lib.basd_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.basd_last_error(), bytes)


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    import basd_amd._native as native
    with pytest.raises(native.BasdNativeError):
        native.token_gram(torch.zeros(64, 32), torch.eye(32))
