"""ctypes binding of libmvdetr_ops.so (C ABI in include/mvdetr_ops.h).

There is NO fallback: if the shared library is missing or does not export the expected symbols
the import of the op layer raises.  GPU tensors only ever reach the HIP kernels; calls whose tensors are ALL on
the CPU run the same library's own host path (csrc/host_path.cpp -- where the reference extension raises
"Not implemented on the CPU", ms_deform_attn.h:38); mixed devices raise.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime this library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# MVDETR_OPS_LIB: another build of the same sources (the phase-stamp build libmvdetr_ops_trace.so of tools/experiments)
LIB_PATH = os.environ.get("MVDETR_OPS_LIB") or os.path.join(CSRC, "libmvdetr_ops.so")
ABI_VERSION = 14

_vp, _i = ctypes.c_void_p, ctypes.c_int
_MSDA_FWD = [_vp] * 6 + [_i] * 7 + [_vp]
_MSDA_BWD = [_vp] * 7 + [_i] * 7 + [_vp] * 3
_WARP = [_vp] * 3 + [_i] * 7 + [_vp]
_MSDA_FWD_HOST = [_vp] * 5 + [_i] * 7 + [_vp]
_MSDA_BWD_HOST = [_vp] * 6 + [_i] * 7 + [_vp] * 3
_WARP_HOST = [_vp] * 2 + [_i] * 8 + [_vp]
_MSDA_FUSED = [_vp] * 5 + [ctypes.c_int64] + [_vp] * 2 + [_i] * 10 + [_vp]
_MSDA_FUSED_LEVELS = [_vp] * 5 + [ctypes.c_int64] + [_vp] * 2 + [_i] * 12 + [_vp]

SIGNATURES = {
    "mvdetr_ops_abi_version": ([], _i),
    "mvdetr_msda_last_forward_impl": ([], ctypes.c_char_p),
    "mvdetr_msda_last_forward_kernel": ([], ctypes.c_char_p),
    "mvdetr_msda_last_forward_resources": ([ctypes.POINTER(ctypes.c_int)] * 3, _i),
    "mvdetr_msda_set_forward_impl": ([_i], _i),
    "mvdetr_msda_set_backward_deterministic": ([_i], _i),
    "mvdetr_msda_get_backward_deterministic": ([], _i),
    "mvdetr_msda_release_scratch": ([], _i),
    "mvdetr_warp_last_kernel": ([], ctypes.c_char_p),
    "mvdetr_warp_release_scratch": ([], _i),
    "mvdetr_msda_forward_f32": (_MSDA_FWD, _i),
    "mvdetr_msda_forward_f64": (_MSDA_FWD, _i),
    "mvdetr_msda_forward_f16": (_MSDA_FWD, _i),
    "mvdetr_msda_forward_bf16": (_MSDA_FWD, _i),
    "mvdetr_msda_fused_supported": ([_i] * 7, _i),
    "mvdetr_msda_forward_fused_f32": (_MSDA_FUSED, _i),
    "mvdetr_msda_fused_levels_supported": ([_i] * 9, _i),
    "mvdetr_msda_forward_fused_levels_f32": (_MSDA_FUSED_LEVELS, _i),
    "mvdetr_msda_fused_train_supported": ([_i] * 7, _i),
    "mvdetr_msda_forward_fused_train_f32": ([_vp] * 5 + [ctypes.c_int64, _vp] + [_i] * 7 + [_vp, _vp], _i),
    "mvdetr_msda_backward_fused_f32": ([_vp] * 6 + [ctypes.c_int64, _vp, _i, _vp, _vp] + [_i] * 6 + [_vp, _vp], _i),
    "mvdetr_msda_backward_f32": (_MSDA_BWD, _i),
    "mvdetr_msda_backward_f64": (_MSDA_BWD, _i),
    "mvdetr_add_layernorm_f32": ([_vp] * 5 + [ctypes.c_int64, _i, ctypes.c_float, _vp], _i),
    "mvdetr_add_layernorm_add_f32": ([_vp] * 6 + [ctypes.c_int64, ctypes.c_int64, _i, ctypes.c_float, _vp, _vp], _i),
    "mvdetr_warp_perspective_forward_f32": (_WARP, _i),
    "mvdetr_warp_perspective_forward_f64": (_WARP, _i),
    "mvdetr_warp_perspective_backward_f32": (_WARP, _i),
    "mvdetr_warp_perspective_backward_f64": (_WARP, _i),
    "mvdetr_warp_perspective_backward_tagged_f32": ([_vp] * 3 + [_i] * 7 + [ctypes.c_uint64, _vp], _i),
    "mvdetr_warp_perspective_backward_tagged_f64": ([_vp] * 3 + [_i] * 7 + [ctypes.c_uint64, _vp], _i),
    "mvdetr_warp_backward_plan_bytes": ([_i] * 7, ctypes.c_int64),
    "mvdetr_warp_backward_plan_f32": ([_vp, _vp] + [_i] * 6 + [_vp], _i),
    "mvdetr_warp_backward_plan_f64": ([_vp, _vp] + [_i] * 6 + [_vp], _i),
    "mvdetr_warp_perspective_backward_planned_f32": ([_vp] * 4 + [_i] * 7 + [_vp], _i),
    "mvdetr_warp_perspective_backward_planned_f64": ([_vp] * 4 + [_i] * 7 + [_vp], _i),
    "mvdetr_transpose_f32": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "mvdetr_transpose_f64": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "mvdetr_msda_forward_host_f32": (_MSDA_FWD_HOST, _i),
    "mvdetr_msda_forward_host_f64": (_MSDA_FWD_HOST, _i),
    "mvdetr_msda_backward_host_f32": (_MSDA_BWD_HOST, _i),
    "mvdetr_msda_backward_host_f64": (_MSDA_BWD_HOST, _i),
    "mvdetr_warp_perspective_forward_host_f32": (_WARP_HOST, _i),
    "mvdetr_warp_perspective_forward_host_f64": (_WARP_HOST, _i),
    "mvdetr_warp_perspective_backward_host_f32": (_WARP_HOST, _i),
    "mvdetr_warp_perspective_backward_host_f64": (_WARP_HOST, _i),
}


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libmvdetr_ops.so failed:\n" + res.stdout)
    if verbose:
        print(res.stdout)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C mvdetr_amd/csrc`). "
                "There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise ImportError(f"{LIB_PATH} does not export {name}") from e
            fn.argtypes, fn.restype = argtypes, restype
        got = handle.mvdetr_ops_abi_version()
        if got != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {got}, expected {ABI_VERSION} (stale build?)")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    """Launch failures raise (the reference only printf's them, cuh:948-952)."""
    if rc != 0:
        name = "hipErrorInvalidValue" if rc == 1 else f"hipError {rc}"
        raise RuntimeError(f"{what} failed: {name}")


def current_stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def suffix(dtype, half_ok=False) -> str:
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    if half_ok and dtype == torch.float16:
        return "f16"
    if half_ok and dtype == torch.bfloat16:
        return "bf16"
    # AT_DISPATCH_FLOATING_TYPES in the reference: float and double only (ms_deform_attn_cuda.cu:64)
    raise RuntimeError(f'"mvdetr_ops" not implemented for \'{dtype}\'')
