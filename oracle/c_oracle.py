"""ctypes front-end of oracle/liboracle.so (the plain-C restatement in oracle/oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/oracle.c for scope and parity status.
Takes and returns CPU torch tensors (fp32 or fp64, made contiguous here).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc via oracle/Makefile (no GPU toolchain involved)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_impl.h")]
    stale = not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _sfx(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"oracle supports fp32/fp64, got {t.dtype}")


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _i64(*xs):
    return [ctypes.c_int64(int(x)) for x in xs]


def msda_forward(value, shapes, level_start_index, loc, aw):
    value, loc, aw = value.contiguous(), loc.contiguous(), aw.contiguous()
    shapes = shapes.to(torch.int64).contiguous()
    lsi = level_start_index.to(torch.int64).contiguous()
    B, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = torch.empty(B, Lq, M * D, dtype=value.dtype)
    fn = getattr(_lib(), f"oracle_msda_forward_{_sfx(value)}")
    rc = fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), *_i64(B, S, M, D, L, Lq, P), _p(out))
    assert rc == 0
    return out


def msda_backward(value, shapes, level_start_index, loc, aw, grad_out):
    value, loc, aw, grad_out = value.contiguous(), loc.contiguous(), aw.contiguous(), grad_out.contiguous()
    shapes = shapes.to(torch.int64).contiguous()
    lsi = level_start_index.to(torch.int64).contiguous()
    B, S, M, D = value.shape
    Lq, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(aw)
    fn = getattr(_lib(), f"oracle_msda_backward_{_sfx(value)}")
    rc = fn(_p(grad_out), _p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), *_i64(B, S, M, D, L, Lq, P),
            _p(gv), _p(gl), _p(ga))
    assert rc == 0
    return gv, gl, ga


def warp_perspective(src, M, dsize):
    src = src.contiguous()
    M = M.to(src.dtype).contiguous()
    N, C, h, w = src.shape
    H, W = int(dsize[0]), int(dsize[1])
    out = torch.empty(N, C, H, W, dtype=src.dtype)
    fn = getattr(_lib(), f"oracle_warp_perspective_{_sfx(src)}")
    rc = fn(_p(src), _p(M), *_i64(N, C, h, w, H, W), _p(out))
    assert rc == 0
    return out


def warp_perspective_backward(grad_out, M, src_hw):
    grad_out = grad_out.contiguous()
    M = M.to(grad_out.dtype).contiguous()
    N, C, H, W = grad_out.shape
    h, w = int(src_hw[0]), int(src_hw[1])
    gs = torch.empty(N, C, h, w, dtype=grad_out.dtype)
    fn = getattr(_lib(), f"oracle_warp_perspective_backward_{_sfx(grad_out)}")
    rc = fn(_p(grad_out), _p(M), *_i64(N, C, h, w, H, W), _p(gs))
    assert rc == 0
    return gs
