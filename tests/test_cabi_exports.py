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
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(basd_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in basd_hip.h but not exported"
    assert declared == set(native.EXPORTS)
    lib.basd_version.restype = ctypes.c_int
    assert lib.basd_version() >= 100
    lib.basd_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.basd_last_error(), bytes)


def test_ctypes_signatures_match_the_header():
    """every export carries explicit argtypes (no default int conversion of 64-bit sizes / pointers) and they agree,
    argument by argument, with the declarations of include/basd_hip.h"""
    import basd_amd._native as native
    header = open(os.path.join(ROOT, "include", "basd_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    seen = 0
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(basd_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        want = []
        for a in ([] if args in ("void", "") else args.split(",")):
            a = a.strip()
            if "*" in a:
                want.append(ctypes.c_void_p)
            elif a.startswith("int64_t"):
                want.append(ctypes.c_int64)
            elif a.startswith("float"):
                want.append(ctypes.c_float)
            elif a.startswith("double"):
                want.append(ctypes.c_double)
            else:
                assert a.startswith("int "), (name, a)
                want.append(ctypes.c_int)
        assert tuple(want) == tuple(native._SIGNATURES[name]), name
        fn = getattr(native.lib(), name)
        assert list(fn.argtypes) == want and fn.restype is not None, name
        seen += 1
    assert seen == len(native.EXPORTS)


def test_status_codes_map_to_a_linalg_error():
    import pytest
    import torch
    import basd_amd._native as native
    native.raise_for_status(0)
    for bit in (native.STATUS_NONCONVERGED, native.STATUS_NONFINITE, native.STATUS_RANK0):
        with pytest.raises(torch.linalg.LinAlgError):
            native.raise_for_status(bit)
    header = open(os.path.join(ROOT, "include", "basd_hip.h")).read()
    for name, val in (("NONCONVERGED", native.STATUS_NONCONVERGED), ("NONFINITE", native.STATUS_NONFINITE),
                      ("RANK0", native.STATUS_RANK0)):
        assert re.search(rf"#define BASD_STATUS_{name} {val}\b", header)


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    import basd_amd._native as native
    with pytest.raises(native.BasdNativeError):
        native.token_gram(torch.zeros(64, 32), torch.eye(32))
