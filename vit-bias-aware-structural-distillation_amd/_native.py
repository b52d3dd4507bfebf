"""ctypes binding of ``libbasd_hip.so`` (C-ABI declared in include/basd_hip.h).

There is NO CPU fallback: if the library is missing or a call fails this
module raises.  torch is used only for device memory and streams.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BASD_LIB: an alternative build of the same library (A/B timing of kernel variants from scripts/); still no CPU path
LIB_PATH = os.environ.get("BASD_LIB") or os.path.join(_HERE, "libbasd_hip.so")
CSRC = os.path.join(_HERE, "csrc")

DTYPE_F32, DTYPE_BF16, DTYPE_F64 = 0, 1, 2
JACOBI_LDS_BYTES = 163840

EXPORTS = (
    "basd_version", "basd_last_error", "basd_token_gram", "basd_token_gram_bf16x3", "basd_gram_f32_centred", "basd_pchol_f64", "basd_jacobi_svd",
    "basd_mp_rank", "basd_flag_if_exceeds_f64", "basd_angle_weights", "basd_ce_uwso",
    "basd_procrustes_workspace_bytes", "basd_procrustes_fwd", "basd_procrustes_bwd_side",
    "basd_procrustes_bwd_workspace_bytes", "basd_procrustes_bwd", "basd_angle_weights_bwd_workspace_bytes",
    "basd_angle_weights_bwd", "basd_mix_tokens", "basd_procrustes_prep", "basd_mix_grad_dots",
    "basd_sf_adamw_step", "basd_lerp", "basd_transpose_bf16_table", "basd_bgemm_f64", "basd_trinv_f64", "basd_bgemm_f64_masked", "basd_trinv_f64_masked", "basd_pchol_f64_masked", "basd_wgrad_bf16", "basd_wgrad_workspace_bytes", "basd_wgrad_bf16_ws", "basd_gemm_bf16",
    "basd_gemm_bf16_gelu_fwd", "basd_gemm_bf16_gelu_bwd", "basd_gemm_bf16x3_f32", "basd_layernorm_fwd_bf16", "basd_layernorm_bwd_bf16",
    "basd_cls_importance_bf16", "basd_add_layernorm_fwd_bf16", "basd_procrustes_bwd_rows", "basd_attention_fwd_bf16", "basd_attention_fwd_qmean_bf16", "basd_attention_bwd_bf16",
)


_P, _I, _I64, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
# argument types of every export, in the order of include/basd_hip.h
_SIGNATURES = {
    "basd_version": (),
    "basd_last_error": (),
    "basd_token_gram": (_P, _I, _I64, _I, _I, _I64, _P, _I, _P, _P, _P),
    "basd_token_gram_bf16x3": (_P, _I64, _I, _I, _I64, _P, _I, _P, _P, _P),
    "basd_gram_f32_centred": (_P, _I64, _I, _P, _P, _P),
    "basd_pchol_f64": (_P, _I, _I, _D, _P, _P, _I, _P, _P, _P, _P),
    "basd_pchol_f64_masked": (_P, _I, _I, _D, _P, _P, _I, _P, _P, _P, _P, _P),
    "basd_trinv_f64": (_P, _P, _P, _I, _I, _P, _P),
    "basd_trinv_f64_masked": (_P, _P, _P, _I, _I, _P, _P, _P),
    "basd_jacobi_svd": (_P, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P, _I, _P, _P),
    "basd_mp_rank": (_P, _I, _I, _I64, _I, _I, _P, _P, _P),
    "basd_flag_if_exceeds_f64": (_P, _I64, _D, _I, _P, _P),
    "basd_angle_weights": (_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P),
    "basd_mix_tokens": (_P, _I, _I, _I, _P, _I64, _I64, _I64, _P, _P),
    "basd_procrustes_prep": (_P, _I, _I64, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P),
    "basd_mix_grad_dots": (_P, _I, _I, _I, _P, _I64, _I64, _I64, _P, _P),
    "basd_procrustes_workspace_bytes": (_I, _I, _I, _I),
    "basd_procrustes_fwd": (_P, _P, _I, _I, _I, _I, _D, _P, _P, _P, _P, _P, _I64, _P),
    "basd_procrustes_bwd_side": (_P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P),
    "basd_procrustes_bwd_workspace_bytes": (_I, _I),
    "basd_angle_weights_bwd_workspace_bytes": (_I, _I, _I),
    "basd_angle_weights_bwd": (_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I64, _P),
    "basd_procrustes_bwd": (_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I64, _P),
    "basd_ce_uwso": (_P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P),
    "basd_transpose_bf16_table": (_P, _P, _P, _I, _P),
    "basd_bgemm_f64": (_P, _I, _I64, _I, _I, _P, _I, _I64, _I, _I, _P, _I, _I64, _I, _I, _I, _I, _I, _I, _P),
    "basd_bgemm_f64_masked": (_P, _I, _I64, _I, _I, _P, _I, _I64, _I, _I, _P, _I, _I64, _I, _I, _I, _I, _I, _I, _P, _P),
    "basd_wgrad_bf16": (_P, _P, _I64, _I, _I, _P, _P, _P),
    "basd_wgrad_workspace_bytes": (_I64, _I, _I),
    "basd_wgrad_bf16_ws": (_P, _P, _I64, _I, _I, _P, _P, _P, _I64, _P),
    "basd_gemm_bf16": (_P, _P, _P, _P, _I64, _I, _I, _I, _I, _P),
    "basd_gemm_bf16_gelu_fwd": (_P, _P, _P, _P, _P, _I64, _I, _I, _I, _P),
    "basd_gemm_bf16x3_f32": (_P, _P, _P, _I64, _I, _I, _P),
    "basd_gemm_bf16_gelu_bwd": (_P, _P, _P, _P, _I64, _I, _I, _I, _P),
    "basd_layernorm_fwd_bf16": (_P, _P, _P, _I64, _I, _F, _P, _P, _P, _P),
    "basd_add_layernorm_fwd_bf16": (_P, _P, _P, _P, _I64, _I, _F, _P, _P, _P, _P, _P, _I, _P),
    "basd_layernorm_bwd_bf16": (_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _I, _P),
    "basd_procrustes_bwd_rows": (_P, _P, _P, _P, _I64, _I, _I, _P, _I, _P, _P),
    "basd_cls_importance_bf16": (_P, _I, _I, _I, _I, _F, _P, _P),
    "basd_attention_fwd_bf16": (_P, _I, _I, _I, _I, _F, _P, _P, _P, _P),
    "basd_attention_fwd_qmean_bf16": (_P, _I, _I, _I, _I, _F, _P, _P, _P),
    "basd_attention_bwd_bf16": (_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P),
    "basd_sf_adamw_step": (_P, _P, _P, _P, _I64, _D, _D, _D, _D, _D, _D, _D, _P),
    "basd_lerp": (_P, _P, _I64, _F, _P),
}


class BasdNativeError(RuntimeError):
    pass


class BasdLinAlgError(torch.linalg.LinAlgError):
    """Data-dependent failure of a linear-algebra kernel (non-finite input, Jacobi without convergence, a rank-0
    teacher layer): the counterpart of the ``torch._C._LinAlgError`` the reference's ``torch.linalg`` calls raise."""


STATUS_NONCONVERGED, STATUS_NONFINITE, STATUS_RANK0 = 1, 2, 4
_STATUS_TEXT = {
    STATUS_NONCONVERGED: "a Jacobi SVD used all its sweeps without converging",
    STATUS_NONFINITE: "non-finite values reached a Jacobi SVD / the Marchenko-Pastur rank (NaN or Inf in the tokens)",
    STATUS_RANK0: "a teacher layer has Marchenko-Pastur rank 0 (pure-noise tokens): the reference's weights are NaN here",
}


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into the in-tree shared library."""
    res = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if res.returncode != 0:
        raise BasdNativeError(f"building libbasd_hip.so failed:\n{res.stdout}\n{res.stderr}")
    if verbose:
        print(res.stdout)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BasdNativeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the BASD kernels)")
        L = ctypes.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise BasdNativeError(f"{LIB_PATH} does not export {name}")
            fn = getattr(L, name)
            fn.argtypes = list(_SIGNATURES[name])      # explicit: no default int conversion of 64-bit sizes / pointers
            fn.restype = (ctypes.c_char_p if name == "basd_last_error" else
                          ctypes.c_int64 if name.endswith("_workspace_bytes") else ctypes.c_int)
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().basd_last_error().decode()
        raise BasdNativeError(f"{what} failed with status {rc}: {msg}")


def _ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DTYPE_F32
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    raise BasdNativeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def handles(t: torch.Tensor) -> bool:
    """The kernels of this provider take device tensors only (module code asks before choosing the fused path)."""
    return t.is_cuda


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise BasdNativeError("BASD kernels need tensors on an MI355X device (no CPU path)")


_STATUS: dict = {}


def status_word(device) -> torch.Tensor:
    """The per-device int32 health word the data-dependent kernels OR their flags into (allocated once; kernels
    captured in a hipGraph keep writing to it)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    buf = _STATUS.get(key)
    if buf is None:
        buf = _STATUS[key] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
    return buf


def raise_for_status(value: int) -> None:
    if value:
        what = "; ".join(text for bit, text in _STATUS_TEXT.items() if value & bit)
        if value >> 8:        # diagnostics of a non-converged Jacobi solve (csrc/jacobi.hip, report_status)
            what += f" [kernel variant {(value >> 8) & 15}, matrix {(value >> 12) & 0xffff}]"
        raise BasdLinAlgError(f"BASD kernels reported status {value & 255}: {what}")


def check_status(device=None) -> None:
    """Read (synchronising) and clear the health word of ``device``; raises BasdLinAlgError if a kernel flagged a
    failure since the last check."""
    buf = status_word(device if device is not None else torch.cuda.current_device())
    value = int(buf.item())
    if value:
        buf.zero_()
    raise_for_status(value)


def jacobi_ld(m_rows: int) -> int:
    """Column stride used for LDS-resident Jacobi inputs (multiple of 4, not of 32)."""
    ld = (m_rows + 3) // 4 * 4
    if ld % 32 == 0:
        ld += 4
    return ld


# --------------------------------------------------------------------------- #
_SPLIT_CACHE: dict = {}


def split_bf16x3(proj: torch.Tensor) -> torch.Tensor:
    """fp32 [d_out, d_in] -> bf16 [3, d_out, d_in] with proj == hi + mid + lo to 2^-24 (cached per buffer)."""
    key = (proj.data_ptr(), proj._version, tuple(proj.shape))
    hit = _SPLIT_CACHE.get(key, (None, None))[0]
    if hit is None:
        p = proj.detach().float()
        hi = p.bfloat16()
        r1 = p - hi.float()
        mid = r1.bfloat16()
        lo = (r1 - mid.float()).bfloat16()
        hit = torch.stack([hi, mid, lo]).contiguous()
        if len(_SPLIT_CACHE) > 16:
            _SPLIT_CACHE.clear()
        _SPLIT_CACHE[key] = (hit, proj)            # holds proj: its address cannot be handed to another tensor meanwhile
    return hit


def split_bf16x3_rows(proj: torch.Tensor) -> torch.Tensor:
    """fp32 [d_out, d_in] -> bf16 [ceil(d_out / 256) * 256, 3 * d_in]: row n = (hi | mid | lo) of proj[n] side by side,
    zero rows behind d_out (operand layout of basd_gemm_bf16x3_f32; cached per buffer)."""
    key = ("rows", proj.data_ptr(), proj._version, tuple(proj.shape))
    hit = _SPLIT_CACHE.get(key, (None, None))[0]
    if hit is None:
        d_out, d_in = proj.shape
        s3 = split_bf16x3(proj)                                                     # [3, d_out, d_in]
        hit = torch.zeros((d_out + 255) // 256 * 256, 3 * d_in, dtype=torch.bfloat16, device=proj.device)
        hit[:d_out] = s3.permute(1, 0, 2).reshape(d_out, 3 * d_in)
        _SPLIT_CACHE[key] = (hit, proj)
    return hit


def gemm_bf16x3_f32_supported(m: int, n: int, k: int) -> bool:
    return m > 7 * 256 and k % 64 == 0 and k >= 128 and n % 8 == 0


def gemm_bf16x3_f32(x: torch.Tensor, proj: torch.Tensor) -> torch.Tensor:
    """x [M, K] bf16, proj [N, K] fp32 -> x proj^T [M, N] fp32: ONE persistent bf16-MFMA GEMM of depth 3 K over the
    (hi | mid | lo) bf16 splits of proj, fp32 accumulation across all three (x is exact in bf16, proj to 2^-24)."""
    _need_cuda(x, proj)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and proj.dim() == 2 and x.shape[1] == proj.shape[1]
    x = x.contiguous()
    m, k = x.shape
    n = proj.shape[0]
    w3 = split_bf16x3_rows(proj.contiguous().float())
    z = torch.empty(m, n, dtype=torch.float32, device=x.device)
    _check(lib().basd_gemm_bf16x3_f32(_ptr(x), _ptr(w3), _ptr(z), ctypes.c_int64(m), n, k, _stream()),
           "basd_gemm_bf16x3_f32")
    return z


def _token_view(x: torch.Tensor):
    """[B, N, D] (possibly a batch-strided view such as out[:, 1:, :]) or [M, D] ->
    (tensor to take the pointer from, rows, d, rows_per_batch, batch_stride) without copying when possible."""
    if x.dim() == 3:
        b, n, d = x.shape
        if x.stride(2) == 1 and x.stride(1) == d and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0:
            return x, b * n, d, n, x.stride(0)
        x = x.contiguous()
        return x, b * n, d, n, n * d
    x = x.contiguous()
    return x, x.shape[0], x.shape[1], x.shape[0], 0


def _mirror_lower(gram: torch.Tensor) -> torch.Tensor:
    """The kernels accumulate the lower 16x16 tiles only (diagonal tiles are complete)."""
    low = torch.tril(gram)
    return low + torch.tril(gram, -1).t()


def _split_rows(m: int) -> int:
    """Number of K-slabs for the split-K Gram of a tall [m, d] matrix (slab rows stay a multiple of 4)."""
    for s in (64, 32, 16, 8, 4, 2):
        if m % (4 * s) == 0 and m // s >= 256:
            return s
    return 1


# BASD_WIDE_GRAM=f64: the split-K fp64-MFMA Gram of rounds 2 - 3 instead of basd_gram_f32_centred (A/B timing)
WIDE_GRAM_FP32 = os.environ.get("BASD_WIDE_GRAM", "f32c") != "f64"


def _token_gram_wide(x: torch.Tensor, proj: torch.Tensor):
    """d_out > 256 (student widths 384 / 768, BASELINE c4 / c5): the fused one-pass kernel keeps a [128, d_out] z tile
    and d_out^2 / 256 fp64 accumulator tiles on chip, which stops at 256 columns.  Here z = X P^T is materialised in
    fp32 and z^T z runs on the fp64 matrix cores as a split-K batched GEMM (basd_bgemm_f64, symmetric tiles only)
    followed by the slab sum.  z = X P^T is an own kernel either way (a library GEMM here would be the only one of the
    teacher branch, i.e. the only reason not to run that branch on its own stream: trainer.py, _ensure_stream_policy):
    bf16 tokens at training sizes take the bf16x3 persistent GEMM (basd_gemm_bf16x3_f32: 24 x 0.1 ms at BASELINE c4;
    the fp64-MFMA batched GEMM at batch 1 needed 2.6 ms per layer there), anything else the fp64-MFMA one."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype == torch.bfloat16 and gemm_bf16x3_f32_supported(x2.shape[0], proj.shape[0], x2.shape[1]):
        z = gemm_bf16x3_f32(x2, proj)
    else:
        z = bgemm_f64(x2.float().unsqueeze(0), proj.unsqueeze(0), trans_b=True, out_dtype=torch.float32)[0]
    m, d = z.shape
    if WIDE_GRAM_FP32 and d % 16 == 0:
        # round 4: tile-centred Gram on the fp32 matrix cores, fp64 across the 64-row tiles (basd_gram_f32_centred)
        z = z.contiguous()
        gram = torch.zeros(d, d, dtype=torch.float64, device=z.device)
        colsum = torch.zeros(d, dtype=torch.float64, device=z.device)
        _check(lib().basd_gram_f32_centred(_ptr(z), ctypes.c_int64(m), d, _ptr(gram), _ptr(colsum), _stream()),
               "basd_gram_f32_centred")
        return _mirror_lower(gram), colsum
    s = _split_rows(m)
    zs = z.view(s, m // s, d)
    gram = bgemm_f64(zs, zs, trans_a=True, symmetric=True).sum(dim=0)              # [d, d] fp64
    return gram, z.sum(dim=0, dtype=torch.float64)


def token_gram(x: torch.Tensor, proj: torch.Tensor, mirror: bool = True, out=None):
    """x [M, d_in] or [B, N, d_in] view (f32/bf16), proj [d_out, d_in] f32 ->
    gram [d_out, d_out] f64, colsum [d_out] f64.  ``mirror=False`` leaves the strict upper triangle
    unspecified (the kernels fill lower tiles only; ``pchol`` reads nothing else).
    ``out`` = (gram, colsum) ZEROED fp64 buffers to accumulate into (e.g. slices of one per-step allocation: a
    16-layer step otherwise spends 32 launches on zero-fills); requires ``mirror=False``."""
    _need_cuda(x, proj)
    if proj.shape[0] > 256 or proj.shape[0] % 16 or x.shape[-1] % 32:
        g, c = _token_gram_wide(x, proj.contiguous().float())
        if out is not None:
            out[0].copy_(g)
            out[1].copy_(c)
            return out
        return g, c
    x, m, d_in, rpb, bstride = _token_view(x)
    proj = proj.contiguous().float()
    d_out = proj.shape[0]
    if out is not None:
        assert not mirror
        gram, colsum = out
        assert gram.shape == (d_out, d_out) and colsum.shape == (d_out,) and gram.dtype == torch.float64
        assert gram.is_contiguous() and colsum.is_contiguous()
    else:
        gram = torch.zeros(d_out, d_out, dtype=torch.float64, device=x.device)
        colsum = torch.zeros(d_out, dtype=torch.float64, device=x.device)
    i64 = ctypes.c_int64
    if x.dtype == torch.bfloat16 and d_out in (32, 64, 128, 192) and d_in % 32 == 0:
        ps = split_bf16x3(proj)
        _check(lib().basd_token_gram_bf16x3(_ptr(x), i64(m), d_in, rpb, i64(bstride), _ptr(ps), d_out, _ptr(gram),
                                            _ptr(colsum), _stream()), "basd_token_gram_bf16x3")
        return (_mirror_lower(gram) if mirror else gram), colsum
    _check(lib().basd_token_gram(_ptr(x), _dtype_code(x), i64(m), d_in, rpb, i64(bstride), _ptr(proj), d_out,
                                 _ptr(gram), _ptr(colsum), _stream()), "basd_token_gram")
    return (_mirror_lower(gram) if mirror else gram), colsum


def _skip_mask(skip, batch: int):
    """int32 [batch] device mask of the *_masked entries (non-zero = leave the problem's outputs untouched) | None"""
    if skip is None:
        return None
    assert skip.dtype == torch.int32 and skip.numel() == batch and skip.is_contiguous()
    return skip


def pchol(a: torch.Tensor, tol: float = 1e-13, dmax_ref: torch.Tensor | None = None, skip: torch.Tensor | None = None):
    """a [batch, n, n] f64 PSD -> (w0 [batch, n, ld] f32, lwork [batch, n, n] f64, piv, rank).
    ``dmax_ref`` (fp64 [batch], optional): the stop test is ``pivot > tol * dmax_ref[b]`` instead of relative to the
    matrix's own largest diagonal entry (panels of a blocked factorisation).  ``skip`` (int32 [batch] on the device,
    optional): the outputs of problems with a non-zero entry are left uninitialised (basd_pchol_f64_masked)."""
    _need_cuda(a)
    a = a.contiguous()
    if dmax_ref is not None:
        dmax_ref = dmax_ref.contiguous()
        assert dmax_ref.dtype == torch.float64 and dmax_ref.numel() == a.shape[0]
    batch, n, _ = a.shape
    ld = jacobi_ld(n)
    w0 = torch.empty(batch, n, ld, dtype=torch.float32, device=a.device)
    lwork = torch.empty(batch, n, n, dtype=torch.float64, device=a.device)
    piv = torch.empty(batch, n, dtype=torch.int32, device=a.device)
    rank = torch.empty(batch, dtype=torch.int32, device=a.device)
    _check(lib().basd_pchol_f64_masked(_ptr(a), batch, n, ctypes.c_double(tol), _ptr(dmax_ref), _ptr(w0), ld, _ptr(lwork),
                                       _ptr(piv), _ptr(rank), _ptr(_skip_mask(skip, batch)), _stream()),
           "basd_pchol_f64_masked")
    return w0, lwork, piv, rank


def jacobi_svd(w: torch.Tensor, m_rows: int, norm_rows: int | None = None, *, tol: float | None = None,
               max_sweeps: int = 60, sort: bool = True, active: torch.Tensor | None = None,
               active_rows: bool = False, flag_status: bool = True):
    """In-place one-sided Jacobi on w [batch, n_cols, ld] (column-major matrices).

    Returns (sigma [batch, n_cols], sweeps [batch]); w's columns become sigma_c * u_c.
    ``flag_status=False``: do not report non-convergence / non-finite values into the health word (single-sweep
    visits of a block tournament stop early on purpose).
    """
    _need_cuda(w)
    assert w.is_contiguous() and w.dtype == torch.float32
    batch, n_cols, ld = w.shape
    if norm_rows is None:
        norm_rows = m_rows
    if tol is None:
        tol = (m_rows ** 0.5) * 5.96e-8
    sigma = torch.empty(batch, n_cols, dtype=torch.float32, device=w.device)
    sweeps = torch.empty(batch, dtype=torch.int32, device=w.device)
    _check(lib().basd_jacobi_svd(_ptr(w), batch, m_rows, n_cols, ld, norm_rows, ctypes.c_float(tol),
                                 max_sweeps, int(sort), _ptr(sigma), _ptr(sweeps),
                                 _ptr(None if active is None else active.contiguous().int()), int(active_rows),
                                 _ptr(status_word(w.device) if flag_status else None), _stream()),
           "basd_jacobi_svd")
    return sigma, sweeps


def jacobi_fits(n_cols: int, m_rows: int) -> bool:
    return n_cols <= 256 and n_cols * jacobi_ld(m_rows) * 4 + 520 * 4 <= JACOBI_LDS_BYTES


JACOBI_TALL_ROWS = 384       # register-resident variant for block pairs: n_cols <= 192 columns of up to 384 rows


def mp_rank(evals: torch.Tensor, rows: int, d: int, cap: int) -> torch.Tensor:
    """evals [batch, n] (eigenvalues of X^T X) -> int32 ranks [batch], on device."""
    _need_cuda(evals)
    evals = evals.contiguous().float()
    batch, n = evals.shape
    ranks = torch.empty(batch, dtype=torch.int32, device=evals.device)
    _check(lib().basd_mp_rank(_ptr(evals), batch, n, ctypes.c_int64(rows), d, cap, _ptr(ranks),
                              _ptr(status_word(evals.device)), _stream()), "basd_mp_rank")
    return ranks


def flag_if_exceeds(values: torch.Tensor, tol: float, bit: int) -> None:
    """OR ``bit`` into the device health word if any entry of ``values`` (fp64) is not <= tol (NaN included): atomic,
    no host sync."""
    _need_cuda(values)
    v = values.detach().reshape(-1).to(torch.float64).contiguous()
    _check(lib().basd_flag_if_exceeds_f64(_ptr(v), ctypes.c_int64(v.numel()), ctypes.c_double(tol), int(bit),
                                          _ptr(status_word(v.device)), _stream()), "basd_flag_if_exceeds_f64")


def angle_weights(sigma: torch.Tensor, sw: torch.Tensor, log_temp: torch.Tensor, unnormalised: bool):
    """sigma [E, L, D] cosines, sw [L, D] masked teacher singular values, log_temp [E] ->
    (d2 [E, L], pre [E, L], weights [E, L], coef [E, L, D]) fp32; see basd_angle_weights."""
    _need_cuda(sigma, sw, log_temp)
    sigma, sw, log_temp = sigma.contiguous().float(), sw.contiguous().float(), log_temp.detach().contiguous().float()
    e, l, d = sigma.shape
    assert sw.shape == (l, d) and log_temp.numel() == e
    d2 = torch.empty(e, l, dtype=torch.float32, device=sigma.device)
    pre, wts = torch.empty_like(d2), torch.empty_like(d2)
    coef = torch.empty(e, l, d, dtype=torch.float32, device=sigma.device)
    _check(lib().basd_angle_weights(_ptr(sigma), _ptr(sw), _ptr(log_temp), e, l, d, int(unnormalised), _ptr(d2), _ptr(pre),
                                    _ptr(wts), _ptr(coef), _stream()), "basd_angle_weights")
    return d2, pre, wts, coef


def angle_weights_bwd(g_w, g_pre_out, wts, d2, log_temp, t_seed, v_s, lam_s, proj_s):
    """Backward of the selector weights as ONE C call (basd_angle_weights_bwd): -> (g_log_temp [E] fp32,
    w_tok [E, D_s, D_s] fp32 with d loss / d s_i = (s_i - column mean) w_tok[i])."""
    _need_cuda(g_w, wts, d2, log_temp, t_seed, v_s, lam_s, proj_s)
    f32 = lambda t_: t_.detach().contiguous().float()
    g_w, wts, d2, log_temp, t_seed, v_s, proj_s = map(f32, (g_w, wts, d2, log_temp, t_seed, v_s, proj_s))
    g_pre_out = None if g_pre_out is None else f32(g_pre_out)
    lam_s = lam_s.detach().contiguous().double()
    e, l, d, _ = t_seed.shape
    d_s = proj_s.shape[1]
    assert proj_s.shape[0] == d and v_s.shape == (e, d, d) and lam_s.shape == (e, d) and g_w.shape == (e, l)
    g_lt = torch.empty(e, dtype=torch.float32, device=g_w.device)
    w_tok = torch.empty(e, d_s, d_s, dtype=torch.float32, device=g_w.device)
    ws = torch.empty(int(lib().basd_angle_weights_bwd_workspace_bytes(e, d, d_s)), dtype=torch.uint8, device=g_w.device)
    _check(lib().basd_angle_weights_bwd(_ptr(g_w), _ptr(g_pre_out), _ptr(wts), _ptr(d2), _ptr(log_temp), _ptr(t_seed),
                                        _ptr(v_s), _ptr(lam_s), _ptr(proj_s), e, l, d, d_s, _ptr(g_lt), _ptr(w_tok),
                                        _ptr(ws), ctypes.c_int64(ws.numel()), _stream()), "basd_angle_weights_bwd")
    return g_lt, w_tok


def _ptr_table(layers: list[torch.Tensor]):
    """Host array of device pointers (handed to the kernel by value: no H2D copy, graph-capturable)."""
    return (ctypes.c_void_p * len(layers))(*[t.data_ptr() for t in layers])


def _layer_views(layers):
    """Common (per_batch, batch_stride) of same-shape layers; copies only layers that are not
    dense-per-sample views."""
    shape = layers[0].shape
    assert all(t.shape == shape and t.dtype == layers[0].dtype for t in layers)
    per_batch = layers[0][0].numel()
    def dense_per_sample(t):
        return t[0].is_contiguous() and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
    strides = {t.stride(0) for t in layers}
    if len(strides) == 1 and all(dense_per_sample(t) for t in layers):
        return list(layers), per_batch, layers[0].stride(0)
    return [t.contiguous() for t in layers], per_batch, per_batch


def mix_tokens(layers: list[torch.Tensor], w: torch.Tensor) -> torch.Tensor:
    """layers: L same-shape [B, ...] tensors (batch-strided views allowed); w [E, L] f32 -> [E, *shape] f32."""
    _need_cuda(*layers, w)
    code = _dtype_code(layers[0])
    layers, per_batch, bstride = _layer_views(layers)
    E, L = w.shape
    assert L == len(layers)
    elems = layers[0].numel()
    out = torch.empty((E,) + tuple(layers[0].shape), dtype=torch.float32, device=w.device)
    table = _ptr_table(layers)
    i64 = ctypes.c_int64
    _check(lib().basd_mix_tokens(table, code, L, E, _ptr(w.contiguous().float()), i64(elems), i64(per_batch),
                                 i64(bstride), _ptr(out), _stream()), "basd_mix_tokens")
    return out


def mix_grad_dots(layers: list[torch.Tensor], g: torch.Tensor) -> torch.Tensor:
    """dots[i, j] = <g[i], layers[j]>; g [E, *shape] f32 -> [E, L] f64."""
    _need_cuda(*layers, g)
    code = _dtype_code(layers[0])
    layers, per_batch, bstride = _layer_views(layers)
    E, L = g.shape[0], len(layers)
    elems = layers[0].numel()
    assert g.numel() == E * elems and g.dtype == torch.float32
    dots = torch.zeros(E, L, dtype=torch.float64, device=g.device)
    table = _ptr_table(layers)
    i64 = ctypes.c_int64
    _check(lib().basd_mix_grad_dots(table, code, L, E, _ptr(g.contiguous()), i64(elems), i64(per_batch),
                                    i64(bstride), _ptr(dots), _stream()), "basd_mix_grad_dots")
    return dots


def procrustes_prep(s: torch.Tensor, t: torch.Tensor, imp: torch.Tensor, out=None):
    """s [B,N_s,D_s] (f32/bf16, batch-strided view allowed), t [B,N_t,D_t] f32, imp [B,N_t] f32
    -> s_w, t_w, a, tr[B,2]  (written into ``out`` = (s_w, t_w, a, tr) when given: contiguous fp32 views)."""
    _need_cuda(s, t, imp)
    if not (s[0].is_contiguous() and s.data_ptr() % 16 == 0):
        s = s.contiguous()
    t = t.contiguous().float()
    imp = imp.contiguous().float()
    B, N_s, D_s = s.shape
    _, N_t, D_t = t.shape
    dev = s.device
    if out is None:
        s_w = torch.empty(B, N_s, D_s, dtype=torch.float32, device=dev)
        t_w = torch.empty(B, N_s, D_t, dtype=torch.float32, device=dev)
        a = torch.empty(B, N_s, dtype=torch.float32, device=dev)
        tr = torch.empty(B, 2, dtype=torch.float32, device=dev)
    else:
        s_w, t_w, a, tr = out
        assert s_w.shape == (B, N_s, D_s) and t_w.shape == (B, N_s, D_t) and a.shape == (B, N_s) and tr.shape == (B, 2)
        for o in out:
            assert o.dtype == torch.float32 and o.is_contiguous()
    _check(lib().basd_procrustes_prep(_ptr(s), _dtype_code(s), ctypes.c_int64(s.stride(0)), _ptr(t), _ptr(imp), B, N_s,
                                      N_t, D_s, D_t, _ptr(s_w), _ptr(t_w), _ptr(a), _ptr(tr), _stream()),
           "basd_procrustes_prep")
    return s_w, t_w, a, tr


def procrustes_bwd_rows(r: torch.Tensor, w: torch.Tensor, a: torch.Tensor, gl: torch.Tensor, out_dtype=torch.float32):
    """r = (other side) G^T, w [B, N, D] fp32, a [B, N], gl [B] -> with R = w - r:
    (2 gl sqrt(a) R  [B, N, D] in out_dtype, 2 gl <R, w>_d  [B, N]).  With out_dtype fp32 the result overwrites ``r``."""
    _need_cuda(r, w, a, gl)
    for t_ in (r, w, a, gl):
        assert t_.dtype == torch.float32 and t_.is_contiguous()
    B, N, D = r.shape
    assert w.shape == r.shape and a.shape == (B, N) and gl.numel() == B
    out = r if out_dtype == torch.float32 else torch.empty(B, N, D, dtype=out_dtype, device=r.device)
    rowdot = torch.empty(B, N, dtype=torch.float32, device=r.device)
    code = DTYPE_F32 if out_dtype == torch.float32 else DTYPE_BF16
    _check(lib().basd_procrustes_bwd_rows(_ptr(r), _ptr(w), _ptr(a), _ptr(gl), ctypes.c_int64(B * N), N, D, _ptr(out),
                                          code, _ptr(rowdot), _stream()), "basd_procrustes_bwd_rows")
    return out, rowdot


def sf_adamw_step(y, g, z, v, *, lr, beta1, beta2, eps, weight_decay, ckp1, bias_correction2) -> None:
    """In-place fused Schedule-Free AdamW step on flat fp32 buffers (all same length)."""
    _need_cuda(y, g, z, v)
    n = y.numel()
    assert g.numel() == n and z.numel() == n and v.numel() == n
    for t in (y, g, z, v):
        assert t.dtype == torch.float32 and t.is_contiguous()
    f = ctypes.c_double
    _check(lib().basd_sf_adamw_step(_ptr(y), _ptr(g), _ptr(z), _ptr(v), ctypes.c_int64(n), f(lr), f(beta1),
                                    f(beta2), f(eps), f(weight_decay), f(ckp1), f(bias_correction2), _stream()),
           "basd_sf_adamw_step")


def lerp_(y, z, w: float) -> None:
    _need_cuda(y, z)
    assert y.dtype == torch.float32 and z.dtype == torch.float32 and y.numel() == z.numel()
    _check(lib().basd_lerp(_ptr(y), _ptr(z), ctypes.c_int64(y.numel()), ctypes.c_float(w), _stream()), "basd_lerp")


# Scratch buffers of the workspace entries (basd_procrustes_fwd, basd_wgrad_bf16_ws): one per (kind, device), grown on
# demand.  A buffer that has been handed out is NEVER freed -- a captured hipGraph keeps replaying into the address it
# was captured with, so an outgrown buffer is parked in ``_RETIRED`` instead of being released -- and a call from a
# stream other than the buffer's last user is ordered behind that user (event wait), so two streams (or two trainers)
# sharing a device never write the same scratch concurrently.
_SCRATCH: dict = {}
_RETIRED: list = []


class _Scratch:
    __slots__ = ("buf", "stream", "event", "shared")

    def __init__(self, buf, stream):
        self.buf, self.stream, self.event, self.shared = buf, stream, None, False


def _scratch(kind: str, device: torch.device, nbytes: int, min_bytes: int = 0) -> "_Scratch":
    key = (kind, device.index)
    cur = torch.cuda.current_stream(device)
    rec = _SCRATCH.get(key)
    if rec is None or rec.buf.numel() < nbytes:
        if rec is not None:
            _RETIRED.append(rec.buf)
        rec = _SCRATCH[key] = _Scratch(torch.empty(max(nbytes, min_bytes), dtype=torch.uint8, device=device), cur)
    elif rec.stream.cuda_stream != cur.cuda_stream:
        if rec.event is not None:
            cur.wait_event(rec.event)
        else:
            cur.wait_stream(rec.stream)
        rec.shared = True                       # from now on every use leaves an event behind
        rec.stream = cur
    return rec


def _scratch_used(rec: "_Scratch") -> None:
    if rec.shared:
        rec.event = torch.cuda.Event()
        rec.event.record(rec.stream)



def procrustes_fwd(s_w: torch.Tensor, t_w: torch.Tensor, tol: float = 1e-13):
    """s_w [batch, n, d_s], t_w [batch, n, d_t] fp32 (weighted, centred tokens from ``procrustes_prep``) ->
    (nuc [batch], fac_s, a_t [batch, n, n]) fp32; fac_s = t_w G^T [batch, n, d_s] if n > d_s else a_s [batch, n, n]
    (see basd_procrustes_fwd).  One C call; the scratch comes from ``_scratch`` (per device, never freed once used)."""
    _need_cuda(s_w, t_w)
    assert s_w.dtype == torch.float32 and t_w.dtype == torch.float32 and s_w.shape[:2] == t_w.shape[:2]
    s_w, t_w = s_w.contiguous(), t_w.contiguous()
    batch, n, d_s = s_w.shape
    d_t = t_w.shape[2]
    need = int(lib().basd_procrustes_workspace_bytes(batch, n, d_s, d_t))
    rec = _scratch("procrustes", s_w.device, need)
    ws = rec.buf
    nuc = torch.empty(batch, dtype=torch.float32, device=s_w.device)
    fac_s = torch.empty(batch, n, d_s if n > d_s else n, dtype=torch.float32, device=s_w.device)
    a_t = torch.empty(batch, n, n, dtype=torch.float32, device=s_w.device)
    _check(lib().basd_procrustes_fwd(_ptr(s_w), _ptr(t_w), batch, n, d_s, d_t, ctypes.c_double(tol), _ptr(nuc),
                                     _ptr(fac_s), _ptr(a_t), _ptr(status_word(s_w.device)), _ptr(ws),
                                     ctypes.c_int64(ws.numel()), _stream()), "basd_procrustes_fwd")
    _scratch_used(rec)
    return nuc, fac_s, a_t


def procrustes_bwd_supported(n: int, d_s: int, d_t: int) -> bool:
    return 4 <= n <= 256 and n % 4 == 0 and d_t % 16 == 0 and d_t >= 16 and (n > d_s or (d_s % 16 == 0 and d_s >= 16)) and d_s % 4 == 0


def procrustes_bwd(s_w: torch.Tensor, t_w: torch.Tensor, a: torch.Tensor, gl: torch.Tensor, fac_s: torch.Tensor,
                   a_t: torch.Tensor, s_dtype=torch.float32):
    """Backward of ``procrustes_fwd`` as ONE C call (basd_procrustes_bwd): -> (g_s [batch, n, d_s] in ``s_dtype``,
    g_t [batch, n, d_t] fp32, g_a [batch, n] fp32).  The big product a_t t_w runs as a bf16 three-product split with the
    residual / scaling / row dots in its epilogue."""
    _need_cuda(s_w, t_w, a, gl, fac_s, a_t)
    for t_ in (s_w, t_w, a, gl, fac_s, a_t):
        assert t_.dtype == torch.float32 and t_.is_contiguous()
    batch, n, d_s = s_w.shape
    d_t = t_w.shape[2]
    assert a_t.shape == (batch, n, n) and fac_s.shape == (batch, n, n if n <= d_s else d_s) and gl.numel() == batch
    g_s = torch.empty(batch, n, d_s, dtype=s_dtype, device=s_w.device)
    g_t = torch.empty(batch, n, d_t, dtype=torch.float32, device=s_w.device)
    g_a = torch.empty(batch, n, dtype=torch.float32, device=s_w.device)
    ws = torch.empty(int(lib().basd_procrustes_bwd_workspace_bytes(batch, n)), dtype=torch.uint8, device=s_w.device)
    code = DTYPE_F32 if s_dtype == torch.float32 else DTYPE_BF16
    _check(lib().basd_procrustes_bwd(_ptr(s_w), _ptr(t_w), _ptr(a), _ptr(gl), _ptr(fac_s), _ptr(a_t), batch, n, d_s, d_t,
                                     _ptr(g_s), code, _ptr(g_t), _ptr(g_a), _ptr(ws), ctypes.c_int64(ws.numel()),
                                     _stream()), "basd_procrustes_bwd")
    return g_s, g_t, g_a


def ce_uwso(logits: torch.Tensor, targets: torch.Tensor, smoothing: float, geo: torch.Tensor | None):
    """logits [B, C] fp32; targets [B, C] float (soft) or [B] int64; geo: 0-d fp32 device tensor | None ->
    (out4 = [total, ce, w_ce, w_geo] fp32 device, dlogits [B, C] = d total / d logits)."""
    _need_cuda(logits, targets, geo)
    assert logits.dtype == torch.float32 and logits.dim() == 2
    logits = logits.contiguous()
    b, c = logits.shape
    soft = labels = None
    if targets.dim() == 2:
        assert targets.shape == (b, c)
        soft = targets.float().contiguous()
    else:
        assert targets.shape == (b,)
        labels = targets.to(torch.int64).contiguous()
    if geo is not None:
        geo = geo.detach().float().reshape(1).contiguous()
    row = torch.empty(b, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits)
    out4 = torch.empty(4, dtype=torch.float32, device=logits.device)
    _check(lib().basd_ce_uwso(_ptr(logits), _ptr(soft), _ptr(labels), b, c, ctypes.c_float(smoothing), _ptr(geo), _ptr(row),
                              _ptr(dl), _ptr(out4), _stream()), "basd_ce_uwso")
    return out4, dl


def transpose_table(master: torch.Tensor, out: torch.Tensor, table) -> None:
    """table: list of (src offset, dst offset, rows, cols); out[dst + c rows + r] = bf16(master[src + r cols + c]) for
    every entry, one launch (weights^T for the input-gradient GEMMs, refreshed once per step)."""
    _need_cuda(master, out)
    assert master.dtype == torch.float32 and out.dtype == torch.bfloat16
    if not table:
        return
    for src, dst, rows, cols in table:
        assert 0 <= src and src + rows * cols <= master.numel() and 0 <= dst and dst + rows * cols <= out.numel()
    flat = (ctypes.c_int64 * (4 * len(table)))(*[int(v) for e in table for v in e])
    _check(lib().basd_transpose_bf16_table(_ptr(master), _ptr(out), ctypes.cast(flat, ctypes.c_void_p), len(table),
                                           _stream()), "basd_transpose_bf16_table")


def bgemm_f64(a: torch.Tensor, b: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
              out_dtype=torch.float64, symmetric: bool = False, skip: torch.Tensor | None = None) -> torch.Tensor:
    """Batched op(a) @ op(b) with fp64 accumulation; a, b [batch, r, c] fp32/fp64 contiguous.  An operand with batch
    1 is broadcast over the other one's batch (batch stride 0: no copies).  ``skip``: see ``pchol``."""
    _need_cuda(a, b)
    a, b = a.contiguous(), b.contiguous()
    code = {torch.float32: DTYPE_F32, torch.float64: DTYPE_F64}
    batch = max(a.shape[0], b.shape[0])
    M, K = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
    K2, N = (b.shape[2], b.shape[1]) if trans_b else (b.shape[1], b.shape[2])
    assert K == K2 and a.shape[0] in (1, batch) and b.shape[0] in (1, batch), (a.shape, b.shape, trans_a, trans_b)
    c = torch.empty(batch, M, N, dtype=out_dtype, device=a.device)
    i64 = ctypes.c_int64
    sa = a.shape[1] * a.shape[2] if a.shape[0] == batch else 0
    sb = b.shape[1] * b.shape[2] if b.shape[0] == batch else 0
    _check(lib().basd_bgemm_f64_masked(_ptr(a), code[a.dtype], i64(sa), a.shape[2], int(trans_a),
                                       _ptr(b), code[b.dtype], i64(sb), b.shape[2], int(trans_b),
                                       _ptr(c), code[out_dtype], i64(M * N), N, batch, M, N, K, int(symmetric),
                                       _ptr(_skip_mask(skip, batch)), _stream()),
           "basd_bgemm_f64_masked")
    return c


def trinv(lwork: torch.Tensor, piv: torch.Tensor, rank: torch.Tensor, skip: torch.Tensor | None = None) -> torch.Tensor:
    """(lwork, piv, rank) from pchol -> L_p^-1 P  [batch, n, n] fp64 (see basd_trinv_f64).  ``skip``: see ``pchol``."""
    _need_cuda(lwork, piv, rank)
    batch, n, _ = lwork.shape
    out = torch.empty(batch, n, n, dtype=torch.float64, device=lwork.device)
    _check(lib().basd_trinv_f64_masked(_ptr(lwork.contiguous()), _ptr(piv.contiguous()), _ptr(rank.contiguous()), batch, n,
                                       _ptr(out), _ptr(_skip_mask(skip, batch)), _stream()), "basd_trinv_f64_masked")
    return out


# Tiles a workgroup of the persistent GEMM kernel multiplies before it retires (basd_gemm_bf16's ``tile_run``): 0 = its
# whole share, the fastest form when the GEMM has the GPU to itself (the default: inference, eager / single-stream steps);
# the Trainer sets 2 while its two-stream pipelined step is in use (a persistent launch holds every CU until it ends and
# the other stream's short kernels queue behind it: measured 41.3 vs 39.5 ms per c2 step).
GEMM_TILE_RUN = 0
# The same for the GEMMs of the TRAINED model that take the persistent kernel (fc1 + GELU forward, fc2 input gradient +
# GELU backward: basd_gemm_bf16_gelu_fwd / _bwd): they are on the step's own chain, not on the side stream.
# Fully persistent (0) whatever the scope above says: pipelined c2 step 32.54 / 32.54 ms against 32.77 / 32.82 with one
# tile per workgroup and 32.65 / 32.60 with two (same box).  BASD_GEMM_TILE_RUN_STUDENT overrides (A/B runs).
GEMM_TILE_RUN_STUDENT = int(os.environ.get("BASD_GEMM_TILE_RUN_STUDENT", "0"))


def _student_tile_run() -> int:
    return GEMM_TILE_RUN_STUDENT


@contextlib.contextmanager
def gemm_tile_run(k: int):
    """``with gemm_tile_run(2):`` -- the GEMM entries called inside retire their workgroups every k tiles (0: fully
    persistent, the fastest form when the launch has the GPU to itself).  Scoped: the Trainer wraps only the steps that
    really run two streams; evaluation, inference and any other model in the process keep the default."""
    global GEMM_TILE_RUN
    prev, GEMM_TILE_RUN = GEMM_TILE_RUN, int(k)
    try:
        yield
    finally:
        GEMM_TILE_RUN = prev


def gemm_supported(n: int, k: int) -> bool:
    return k % 64 == 0 and k >= 64 and n >= 128 and (n % 256 == 0 or n % 192 == 0 or n % 128 == 0)


def gemm_bf16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, gelu: bool = False) -> torch.Tensor:
    """x [..., K], w [N, K], bias [N] | None (all bf16) -> epi(x w^T + bias) [..., N] bf16; epi = exact-erf GELU if
    ``gelu``.  Inference / explicit-backward building block (no autograd)."""
    _need_cuda(x, w, bias)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and (bias is None or bias.dtype == torch.bfloat16)
    k = x.shape[-1]
    n = w.shape[0]
    assert w.shape[1] == k and gemm_supported(n, k), (tuple(w.shape), k)
    x2 = x.reshape(-1, k)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    w = w.contiguous()
    y = torch.empty(x2.shape[0], n, dtype=torch.bfloat16, device=x.device)
    epi = 2 if gelu else (1 if bias is not None else 0)
    _check(lib().basd_gemm_bf16(_ptr(x2), _ptr(w), _ptr(None if bias is None else bias.contiguous()), _ptr(y),
                                ctypes.c_int64(x2.shape[0]), n, k, epi, GEMM_TILE_RUN, _stream()), "basd_gemm_bf16")
    return y.view(*x.shape[:-1], n)


def gemm_gelu_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None):
    """x [..., K], w [N, K], bias [N] | None (bf16) -> (pre, act) [..., N] bf16: pre = x w^T + bias rounded to bf16,
    act = gelu(pre) -- fc1 + nn.GELU of a trained block in one launch, ``pre`` saved for ``gemm_gelu_bwd``."""
    _need_cuda(x, w, bias)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and (bias is None or bias.dtype == torch.bfloat16)
    k, n = x.shape[-1], w.shape[0]
    assert w.shape[1] == k and gemm_supported(n, k), (tuple(w.shape), k)
    x2 = x.reshape(-1, k)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    pre = torch.empty(x2.shape[0], n, dtype=torch.bfloat16, device=x.device)
    act = torch.empty_like(pre)
    _check(lib().basd_gemm_bf16_gelu_fwd(_ptr(x2), _ptr(w.contiguous()), _ptr(None if bias is None else bias.contiguous()),
                                         _ptr(pre), _ptr(act), ctypes.c_int64(x2.shape[0]), n, k, _student_tile_run(),
                                         _stream()),
           "basd_gemm_bf16_gelu_fwd")
    return pre.view(*x.shape[:-1], n), act.view(*x.shape[:-1], n)


def gemm_gelu_bwd(dy: torch.Tensor, wt: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    """dy [..., K], wt [N, K] (= fc2.weight^T, contiguous), pre [..., N] (bf16) -> (dy wt^T) * gelu'(pre) [..., N] bf16:
    the input gradient of fc2 and the GELU backward in one launch."""
    _need_cuda(dy, wt, pre)
    assert dy.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and pre.dtype == torch.bfloat16
    k, n = dy.shape[-1], wt.shape[0]
    assert wt.shape[1] == k and pre.shape[-1] == n and gemm_supported(n, k), (tuple(wt.shape), tuple(pre.shape), k)
    dy2 = dy.reshape(-1, k)
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    pre2 = pre.reshape(-1, n)
    assert pre2.is_contiguous() and pre2.shape[0] == dy2.shape[0]
    out = torch.empty_like(pre2)
    _check(lib().basd_gemm_bf16_gelu_bwd(_ptr(dy2), _ptr(wt.contiguous()), _ptr(pre2), _ptr(out),
                                         ctypes.c_int64(dy2.shape[0]), n, k, _student_tile_run(), _stream()),
           "basd_gemm_bf16_gelu_bwd")
    return out.view(*pre.shape)


def wgrad_supported(n: int, k: int) -> bool:
    return n % 64 == 0 and k % 64 == 0 and n >= 64 and k >= 64


def wgrad_bf16(dy: torch.Tensor, x: torch.Tensor, need_bias: bool = True, out_w: torch.Tensor | None = None,
               out_b: torch.Tensor | None = None):
    """dy [M, N], x [M, K] bf16 -> (dw [N, K] fp32, db [N] fp32 | None).

    With ``out_w`` / ``out_b`` (fp32, contiguous, e.g. views of the flat gradient buffer) the kernel
    ACCUMULATES into them and they are returned as is."""
    _need_cuda(dy, x)
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
    dy, x = dy.contiguous(), x.contiguous()
    m, n = dy.shape
    k = x.shape[1]
    dw = out_w if out_w is not None else torch.zeros(n, k, dtype=torch.float32, device=dy.device)
    db = out_b if out_b is not None else (torch.zeros(n, dtype=torch.float32, device=dy.device) if need_bias else None)
    assert dw.is_contiguous() and dw.dtype == torch.float32 and dw.shape == (n, k)
    need = int(lib().basd_wgrad_workspace_bytes(ctypes.c_int64(m), n, k))
    if need > 0:
        rec = _scratch("wgrad", dy.device, need, min_bytes=256 * 192 * 192 * 4)
        _check(lib().basd_wgrad_bf16_ws(_ptr(dy), _ptr(x), ctypes.c_int64(m), n, k, _ptr(dw), _ptr(db), _ptr(rec.buf),
                                        ctypes.c_int64(rec.buf.numel()), _stream()), "basd_wgrad_bf16_ws")
        _scratch_used(rec)
    else:
        _check(lib().basd_wgrad_bf16(_ptr(dy), _ptr(x), ctypes.c_int64(m), n, k, _ptr(dw), _ptr(db), _stream()),
               "basd_wgrad_bf16")
    return dw, db


def cls_importance_supported(t: int, hd: int) -> bool:
    return 2 <= t <= 320 and hd in (32, 64, 80)


def cls_importance(qkv: torch.Tensor, heads: int, head_dim: int, scale: float) -> torch.Tensor:
    """qkv [B, T, 3 * heads * head_dim] bf16 (CLS token first) -> head-averaged CLS attention [B, T-1] fp32."""
    _need_cuda(qkv)
    assert qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 3 * heads * head_dim
    qkv = qkv.contiguous()
    b, t = qkv.shape[0], qkv.shape[1]
    out = torch.empty(b, t - 1, dtype=torch.float32, device=qkv.device)
    _check(lib().basd_cls_importance_bf16(_ptr(qkv), b, t, heads, head_dim, ctypes.c_float(scale), _ptr(out), _stream()),
           "basd_cls_importance_bf16")
    return out


def attention_fwd_supported(t: int, hd: int) -> bool:
    return hd in (64, 80) and 1 <= t <= 272


def attention_fwd(qkv: torch.Tensor, heads: int, head_dim: int, scale: float, want_importance: bool = False,
                  want_lse: bool = False, query_mean: bool = False):
    """qkv [B, T, 3 * heads * head_dim] bf16 -> (out [B, T, heads * head_dim] bf16, importance | None) and, with
    ``want_lse``, additionally the log-sum-exp [B, heads, T] fp32 the backward kernel needs.  importance: the CLS row of
    the head-averaged attention map without its first entry, [B, T-1] fp32 -- or, with ``query_mean`` (teachers without a
    CLS token), that map averaged over the queries, [B, T].  No autograd here (the student wraps forward +
    ``attention_bwd`` in an autograd.Function)."""
    _need_cuda(qkv)
    assert qkv.dtype == torch.bfloat16 and qkv.shape[-1] == 3 * heads * head_dim
    qkv = qkv.contiguous()
    b, t = qkv.shape[0], qkv.shape[1]
    out = torch.empty(b, t, heads * head_dim, dtype=torch.bfloat16, device=qkv.device)
    if want_importance and query_mean:
        assert not want_lse
        imp = torch.empty(b, heads, t, dtype=torch.float32, device=qkv.device)
        _check(lib().basd_attention_fwd_qmean_bf16(_ptr(qkv), b, t, heads, head_dim, ctypes.c_float(scale), _ptr(out),
                                                   _ptr(imp), _stream()), "basd_attention_fwd_qmean_bf16")
        return out, imp.sum(dim=1)
    imp = torch.empty(b, heads, t - 1, dtype=torch.float32, device=qkv.device) if want_importance else None
    lse = torch.empty(b, heads, t, dtype=torch.float32, device=qkv.device) if want_lse else None
    _check(lib().basd_attention_fwd_bf16(_ptr(qkv), b, t, heads, head_dim, ctypes.c_float(scale), _ptr(out), _ptr(imp),
                                         _ptr(lse), _stream()), "basd_attention_fwd_bf16")
    imp = imp.sum(dim=1) if want_importance else None
    return (out, imp, lse) if want_lse else (out, imp)


def attention_bwd_supported(t: int, hd: int) -> bool:
    return hd == 64 and 1 <= t <= 224


def attention_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, heads: int,
                  head_dim: int, scale: float) -> torch.Tensor:
    """-> dqkv [B, T, 3 * heads * head_dim] bf16 (gradient of the packed projection)."""
    _need_cuda(qkv, out, dout, lse)
    b, t = qkv.shape[0], qkv.shape[1]
    assert qkv.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and lse.dtype == torch.float32
    assert qkv.is_contiguous() and out.is_contiguous() and lse.is_contiguous() and lse.shape == (b, heads, t)
    dout = dout.to(torch.bfloat16).contiguous()
    dqkv = torch.empty_like(qkv)
    _check(lib().basd_attention_bwd_bf16(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), b, t, heads, head_dim,
                                         ctypes.c_float(scale), _ptr(dqkv), _stream()), "basd_attention_bwd_bf16")
    return dqkv


def layernorm_supported(d: int) -> bool:
    return d % 8 == 0 and 8 <= d <= 2048


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float):
    """x [..., D] bf16 contiguous, gamma/beta fp32 -> (y bf16, mean [rows] fp32, rstd [rows] fp32)."""
    _need_cuda(x, gamma, beta)
    assert x.dtype == torch.bfloat16 and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    x = x.contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(lib().basd_layernorm_fwd_bf16(_ptr(x), _ptr(gamma.contiguous()), _ptr(beta.contiguous()), ctypes.c_int64(rows),
                                         d, ctypes.c_float(eps), _ptr(y), _ptr(mean), _ptr(rstd), _stream()),
           "basd_layernorm_fwd_bf16")
    return y, mean, rstd


def add_layernorm_fwd(x: torch.Tensor, residual: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                      row_scale: torch.Tensor | None = None, want_stats: bool = False):
    """s = bf16(residual + row_scale[sample] * x) and y = LayerNorm(s): -> (s, y) or, with ``want_stats`` (a trained
    block: backward needs them), (s, y, mean, rstd).  ``row_scale`` fp32 [B] for x [B, T, D] (stochastic depth)."""
    _need_cuda(x, residual, gamma, beta)
    assert x.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16 and x.shape == residual.shape
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    x, residual = x.contiguous(), residual.contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    s = torch.empty_like(x)
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if want_stats else None
    rps = 1
    if row_scale is not None:
        row_scale = row_scale.contiguous().float()
        assert rows % row_scale.numel() == 0
        rps = rows // row_scale.numel()
    _check(lib().basd_add_layernorm_fwd_bf16(_ptr(x), _ptr(residual), _ptr(gamma.contiguous()), _ptr(beta.contiguous()),
                                             ctypes.c_int64(rows), d, ctypes.c_float(eps), _ptr(s), _ptr(y), _ptr(mean),
                                             _ptr(rstd), _ptr(row_scale), rps, _stream()), "basd_add_layernorm_fwd_bf16")
    return (s, y, mean, rstd) if want_stats else (s, y)


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                  dgamma: torch.Tensor | None, dbeta: torch.Tensor | None, dres: torch.Tensor | None = None,
                  row_scale: torch.Tensor | None = None, want_branch: bool = False):
    """-> dx bf16 (= LayerNorm backward + ``dres``); with ``want_branch`` -> (dx, row_scale[sample] * dx).
    dgamma / dbeta (fp32, may be None together) are accumulated into."""
    _need_cuda(dy, x, gamma)
    dy = dy.contiguous()
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dbranch = torch.empty_like(x) if want_branch else None
    rps = 1
    if dres is not None:
        dres = dres.to(torch.bfloat16).contiguous()
    if row_scale is not None:
        row_scale = row_scale.contiguous().float()
        rps = rows // row_scale.numel()
    _check(lib().basd_layernorm_bwd_bf16(_ptr(dy), _ptr(x), _ptr(gamma.contiguous()), _ptr(mean), _ptr(rstd),
                                         ctypes.c_int64(rows), d, _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(dres),
                                         _ptr(dbranch), _ptr(row_scale), rps, _stream()), "basd_layernorm_bwd_bf16")
    return (dx, dbranch) if want_branch else dx
