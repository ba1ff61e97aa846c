"""Python face of the native extension, with the reference pybind module's name and signatures.

Reference: multiview_detector/models/ops/src/vision.cpp:13-16 (module def),
ms_deform_attn.h:20-61 (device dispatch), cuda/ms_deform_attn_cuda.cu:20-153 (argument checks,
output allocation, im2col_step chunking).  The kernels are reached through the C ABI of
include/mvdetr_ops.h (ctypes) on torch's current HIP stream.

Differences from the reference, all deliberate:
  * kernel launch failures raise RuntimeError instead of being printf'd (cuh:948-952);
  * the batch is not chunked by im2col_step -- the kernels take the whole batch in one launch --
    but the divisibility check (cu:52) is kept so the same misuse raises the same way;
  * forward output is allocated with empty() (the kernel writes every element) instead of zeros();
  * CPU tensors WORK: the reference raises "Not implemented on the CPU" (ms_deform_attn.h:38,60 -- its
    ms_deform_attn_cpu.cpp:17-41 are stubs); here they run on the library's own host implementation
    (csrc/host_path.cpp, std::thread), same contract, deterministic.  All tensors of a call must live on one device.
"""
from __future__ import annotations

import torch

from .. import _lib


def _check_inputs(named, allow_host=False):
    """Contiguity and device checks of ms_deform_attn_cuda.cu:28-38.  Returns True when the call is a host call (every
    tensor on the CPU; only where ``allow_host``), False for a device call; mixed devices raise."""
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
    if allow_host and not any(t.is_cuda for _, t in named):
        return True
    for name, t in named:
        if not t.is_cuda:
            # ms_deform_attn.h:38,60
            raise RuntimeError("Not implemented on the CPU" if name == "value" and not allow_host
                               else f"{name} must be a CUDA tensor")
    return False


def _dims(value, spatial_shapes, sampling_loc, im2col_step):
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(batch, int(im2col_step))
    if batch > 0 and (step <= 0 or batch % step != 0):
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")
    return batch, spatial_size, num_heads, channels, num_levels, num_query, num_point


def _meta(t, device):
    if t.dtype != torch.int64:
        raise RuntimeError("spatial_shapes / level_start_index must be int64 (torch.long)")
    return t


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """-> Tensor[batch, num_query, num_heads*channels]  (vision.cpp:14; ms_deform_attn_cuda.cu:20-80)"""
    host = _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                          ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                          ("attn_weight", attn_weight)], allow_host=True)
    B, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    # float16 / bfloat16: device forward only (16-bit storage, fp32 arithmetic; include/mvdetr_ops.h)
    sfx = _lib.suffix(value.dtype, half_ok=not host)
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("value, sampling_loc and attn_weight must have the same dtype")
    _meta(spatial_shapes, value.device), _meta(level_start_index, value.device)
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    if host:
        rc = getattr(_lib.lib(), f"mvdetr_msda_forward_host_{sfx}")(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), B, S, M, D, L, Lq, P, out.data_ptr())
        _lib.check(rc, "ms_deform_attn_forward (host)")
        return out
    with torch.cuda.device(value.device):
        rc = getattr(_lib.lib(), f"mvdetr_msda_forward_{sfx}")(
            _lib.current_stream_ptr(value.device), value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            B, S, M, D, L, Lq, P, out.data_ptr())
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]  (vision.cpp:15; cu:83-153)"""
    host = _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                          ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                          ("attn_weight", attn_weight), ("grad_output", grad_output)], allow_host=True)
    B, S, M, D, L, Lq, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    sfx = _lib.suffix(value.dtype)
    if any(t.dtype != value.dtype for t in (sampling_loc, attn_weight, grad_output)):
        raise RuntimeError("value, sampling_loc, attn_weight and grad_output must have the same dtype")
    _meta(spatial_shapes, value.device), _meta(level_start_index, value.device)
    grad_value = torch.zeros_like(value)              # accumulated with atomics
    grad_loc = torch.empty_like(sampling_loc)         # fully written
    grad_aw = torch.empty_like(attn_weight)           # fully written
    if host:
        rc = getattr(_lib.lib(), f"mvdetr_msda_backward_host_{sfx}")(
            grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Lq, P, grad_value.data_ptr(),
            grad_loc.data_ptr(), grad_aw.data_ptr())
        _lib.check(rc, "ms_deform_attn_backward (host)")
        return [grad_value, grad_loc, grad_aw]
    with torch.cuda.device(value.device):
        rc = getattr(_lib.lib(), f"mvdetr_msda_backward_{sfx}")(
            _lib.current_stream_ptr(value.device), grad_output.data_ptr(), value.data_ptr(),
            spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), B, S, M, D, L, Lq, P, grad_value.data_ptr(), grad_loc.data_ptr(),
            grad_aw.data_ptr())
    _lib.check(rc, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_aw]


def fused_supported(value, num_levels, num_query, num_point, query_levels=None) -> bool:
    """True when ms_deform_attn_forward_fused takes this call (deformable-encoder shapes, fp32)."""
    if not value.is_cuda or value.dtype != torch.float32:
        return False
    return fused_supported_dims(*value.shape, num_levels, num_query, num_point, query_levels)


def fused_supported_dims(B, S, M, D, num_levels, num_query, num_point, query_levels=None) -> bool:
    """The same from the dimensions alone (CUDA fp32 tensors assumed): for callers whose value tensor is still in
    flight (a pending all-gather, mvdetr_amd/dist.py)."""
    l0, l1 = (0, num_levels) if query_levels is None else query_levels
    return bool(_lib.lib().mvdetr_msda_fused_levels_supported(B, S, M, D, num_levels, num_query, num_point, l0, l1))


def ms_deform_attn_forward_fused(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                                 attn_logits, level_major=False, query_levels=None, *, raw=None, ref_level_major=False,
                                 raw_level_outer=False):
    """Core + the module arithmetic around it (ms_deform_attn.py:100-107) in one kernel (inference):
    value [B,S,M,D]; reference_points [B or 1, Lq, L, P, 2] (may be a batch-expanded view), or [B or 1, Lq, L, 2]
    when the P sampling points of a (query, level) share one reference point (MVDeTr's map: P identical copies) --
    that form may also be level-major, [B or 1, L, Lq, 2], with ``ref_level_major``;
    sampling_offsets [B,Lq,M,L,P,2] and attn_logits [B,Lq,M,L,P] are the raw Linear outputs
    ([B,Lq,L,M,P,2] / [B,Lq,L,M,P] with ``level_major``); each may be a column block of a wider GEMM
    output (dense per query, arbitrary query stride).  Alternatively ``raw`` [B, Lq, M/g * L * 12g] holds both,
    slice-interleaved (g = 32/D heads per 128-byte slice of the token row; per (query, slice, level): g*P*2 offsets
    then g*P logits -- see slice_major_rows()); sampling_offsets / attn_logits are then None; ``raw_level_outer``: the same
    runs ordered [B, Lq, L, M/g, 12g] (level outermost).  -> [B, Lq, M*D].
    ``query_levels=(l0, l1)``: the Lq queries are the tokens of levels l0..l1-1 only (one rank's cameras in a
    query-sharded encoder, mvdetr_amd/dist.py); value still holds every level.
    No extension counterpart in the reference: this is SURVEY row f1."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index)])
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    flags = 0
    if raw is not None:
        if sampling_offsets is not None or attn_logits is not None or level_major:
            raise RuntimeError("raw (slice-interleaved) excludes sampling_offsets / attn_logits / level_major")
        if D not in (16, 32):
            raise RuntimeError("the slice-interleaved layout exists for 16 or 32 channels per head")
        P = 4
        Lq = raw.shape[1]
        if (raw.dim() != 3 or raw.shape[0] != B or raw.shape[2] < M * L * P * 3 or not raw.is_cuda or raw.dtype != value.dtype
                or raw.stride(2) != 1 or (B > 1 and raw.stride(0) != raw.stride(1) * Lq)):
            raise RuntimeError("raw must be a CUDA tensor [B, Lq, >= M*L*P*3] of value's dtype, dense per query")
        g = 32 // D
        off_ptr, logit_ptr = raw.data_ptr(), raw.data_ptr() + g * P * 2 * raw.element_size()
        qstrides = [raw.stride(1), raw.stride(1)]
        flags |= 4 | (16 if raw_level_outer else 0)
    else:
        if raw_level_outer:
            raise RuntimeError("raw_level_outer describes the slice-interleaved raw tensor")
        Lq, P = sampling_offsets.shape[1], sampling_offsets.shape[4]
        if tuple(sampling_offsets.shape[2:4]) != ((L, M) if level_major else (M, L)):
            raise RuntimeError("sampling_offsets layout does not match level_major")
        qstrides = []
        for name, t, inner in (("sampling_offsets", sampling_offsets, M * L * P * 2), ("attn_logits", attn_logits, M * L * P)):
            if not t.is_cuda or t.dtype != value.dtype:
                raise RuntimeError(f"{name} must be a CUDA tensor of value's dtype")
            if not t.is_contiguous():
                # accept a column block [..., a:b] of a [B, Lq, K] GEMM output, viewed as [B, Lq, ., ., .(, 2)]
                st = t.stride()
                dense_inner = all(st[i] == st[i + 1] * t.shape[i + 1] for i in range(2, t.dim() - 1)) and st[-1] == 1
                if not dense_inner or (B > 1 and st[0] != st[1] * Lq):
                    raise RuntimeError(f"{name} tensor has to be contiguous per query")
            qstrides.append(t.stride(1))
        off_ptr, logit_ptr = sampling_offsets.data_ptr(), attn_logits.data_ptr()
        flags |= 1 if level_major else 0
    shared_ref = reference_points.dim() == 4
    if ref_level_major and not shared_ref:
        raise RuntimeError("ref_level_major needs one reference point per (query, level)")
    want = ((L, Lq, 2) if ref_level_major else (Lq, L, 2)) if shared_ref else (Lq, L, P, 2)
    if (tuple(reference_points.shape[1:]) != want or not reference_points.is_cuda
            or reference_points.dtype != value.dtype or reference_points.shape[0] not in (1, B)):
        raise RuntimeError("reference_points must be a CUDA tensor of shape [B or 1, Lq, L, P, 2], [B or 1, Lq, L, 2] "
                           "or (ref_level_major) [B or 1, L, Lq, 2]")
    if not reference_points[0].is_contiguous():
        reference_points = reference_points.contiguous()
    rstride = reference_points.stride(0) if reference_points.shape[0] > 1 else 0
    flags |= (2 if shared_ref else 0) | (8 if ref_level_major else 0)
    _meta(spatial_shapes, value.device), _meta(level_start_index, value.device)
    l0, l1 = (0, L) if query_levels is None else (int(query_levels[0]), int(query_levels[1]))
    out = torch.empty((B, Lq, M * D), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        rc = _lib.lib().mvdetr_msda_forward_fused_levels_f32(
            _lib.current_stream_ptr(value.device), value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), reference_points.data_ptr(), rstride, off_ptr, logit_ptr, flags,
            qstrides[0], qstrides[1], l0, l1, B, S, M, D, L, Lq, P, out.data_ptr())
    _lib.check(rc, "ms_deform_attn_forward_fused")
    return out


def fused_train_supported(B, S, M, D, num_levels, num_query, num_point) -> bool:
    """True when the fused TRAINING pair takes a call of these dimensions (CUDA fp32 tensors, equal level shapes and one
    reference point per (query, level) are the caller's to check): include/mvdetr_ops.h."""
    return bool(_lib.lib().mvdetr_msda_fused_train_supported(B, S, M, D, num_levels, num_query, num_point))


def _train_args(value, spatial_shapes, level_start_index, reference_points, raw):
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("reference_points", reference_points)])
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    if value.dtype != torch.float32 or raw.dtype != torch.float32 or reference_points.dtype != torch.float32:
        raise RuntimeError("the fused training pair is float32")
    if tuple(reference_points.shape[1:]) != (L, S, 2) or reference_points.shape[0] not in (1, B):
        raise RuntimeError("reference_points must be [B or 1, L, Lq, 2]: one point per (query, level), level-major")
    if (raw.dim() != 3 or raw.shape[0] != B or raw.shape[1] != S or raw.shape[2] < M * L * 12 or not raw.is_cuda
            or raw.stride(2) != 1 or (B > 1 and raw.stride(0) != raw.stride(1) * S)):
        raise RuntimeError("raw must be a CUDA tensor [B, Lq, >= M*L*P*3], dense per query (slice_major_rows(level_outer=True))")
    _meta(spatial_shapes, value.device), _meta(level_start_index, value.device)
    rstride = reference_points.stride(0) if reference_points.shape[0] > 1 else 0
    return B, S, M, D, L, rstride


def ms_deform_attn_forward_fused_train(value, spatial_shapes, level_start_index, reference_points, raw):
    """Forward of the fused training pair: -> (out [B, Lq, M*D], stats [B, Lq, M, 2]).  ``raw`` [B, Lq, M*L*12] is the
    module's ONE GEMM output in the slice-interleaved, level-outermost layout; ``reference_points`` [B or 1, L, Lq, 2].
    ``out`` equals ms_deform_attn_forward_fused(..., raw=raw, ref_level_major=True, raw_level_outer=True) bit for bit (the same
    kernel); ``stats`` = (maximum logit, 1 / sum exp(logit - maximum)) per (query, head), for the backward."""
    B, S, M, D, L, rstride = _train_args(value, spatial_shapes, level_start_index, reference_points, raw)
    out = torch.empty((B, S, M * D), dtype=value.dtype, device=value.device)
    stats = torch.empty((B, S, M, 2), dtype=value.dtype, device=value.device)
    with torch.cuda.device(value.device):
        rc = _lib.lib().mvdetr_msda_forward_fused_train_f32(
            _lib.current_stream_ptr(value.device), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            reference_points.data_ptr(), rstride, raw.data_ptr(), raw.stride(1), B, S, M, D, L, 4, out.data_ptr(), stats.data_ptr())
    _lib.check(rc, "ms_deform_attn_forward_fused_train")
    return out, stats


def ms_deform_attn_backward_fused(grad_output, value, spatial_shapes, level_start_index, reference_points, raw, stats, out):
    """Backward of the fused training pair: -> (grad_value [B, S, M, D], grad_raw like ``raw``).  Softmax and location
    arithmetic are differentiated inside the kernels; no sampling_locations / attention_weights tensor exists."""
    B, S, M, D, L, rstride = _train_args(value, spatial_shapes, level_start_index, reference_points, raw)
    for name, t_ in (("grad_output", grad_output), ("stats", stats), ("out", out)):
        if not t_.is_cuda or t_.dtype != torch.float32 or not t_.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous CUDA float32 tensor")
    grad_value = torch.zeros_like(value)                       # accumulated (LDS fixed point + fp32 atomics at the flush)
    grad_raw = torch.empty_like(raw) if raw.shape[2] == M * L * 12 else torch.zeros_like(raw)
    with torch.cuda.device(value.device):
        rc = _lib.lib().mvdetr_msda_backward_fused_f32(
            _lib.current_stream_ptr(value.device), grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), reference_points.data_ptr(), rstride, raw.data_ptr(), raw.stride(1), stats.data_ptr(),
            out.data_ptr(), B, S, M, D, L, 4, grad_value.data_ptr(), grad_raw.data_ptr())
    _lib.check(rc, "ms_deform_attn_backward_fused")
    return grad_value, grad_raw


def slice_major_rows(M, L, P, D, level_outer=False):
    """Row order of the ONE Linear that produces the slice-interleaved ``raw`` tensor from the reference module's two:
    (rows of sampling_offsets.weight, rows of attention_weights.weight shifted by M*L*P*2), i.e. indices into
    cat([sampling_offsets.weight, attention_weights.weight]).  Reference row orders: offsets (m, l, p, xy),
    logits (m, l, p) -- ms_deform_attn.py:99-101.  Runs (one per slice and level: g heads' offsets, then their logits)
    ordered slice-major [M/g, L] or, ``level_outer``, level-major [L, M/g]."""
    g = 32 // D
    n_off = M * L * P * 2

    def run(s, l):
        r = []
        for h in range(g):
            m = s * g + h
            r += [((m * L + l) * P + p) * 2 + xy for p in range(P) for xy in range(2)]
        for h in range(g):
            m = s * g + h
            r += [n_off + (m * L + l) * P + p for p in range(P)]
        return r

    pairs = [(s, l) for l in range(L) for s in range(M // g)] if level_outer else \
            [(s, l) for s in range(M // g) for l in range(L)]
    rows = []
    for s, l in pairs:
        rows += run(s, l)
    return rows


def last_forward_impl() -> str:
    """Which kernel variant the last forward on this thread dispatched to (bench/tests only)."""
    return _lib.lib().mvdetr_msda_last_forward_impl().decode()


def last_forward_kernel() -> str:
    """Name of the kernel the last forward on this thread launched (bench/tests only)."""
    return _lib.lib().mvdetr_msda_last_forward_kernel().decode()


def last_forward_resources():
    """What the code object records for the kernel instantiation the last forward on this thread launched:
    ``{"num_regs", "scratch_bytes_per_lane", "static_lds_bytes"}`` (hipFuncGetAttributes; scratch > 0 = it spills), or
    None when the launcher does not report it (bench/tests only)."""
    import ctypes
    r, s_, l_ = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
    if not _lib.lib().mvdetr_msda_last_forward_resources(ctypes.byref(r), ctypes.byref(s_), ctypes.byref(l_)):
        return None
    return {"num_regs": r.value, "scratch_bytes_per_lane": s_.value, "static_lds_bytes": l_.value}


_IMPLS = {"auto": 0, "gather": 1, "tile": 2}


def set_forward_impl(name: str) -> str:
    """Tuning/testing knob: 'auto' | 'gather' | 'tile'.  Returns the previous setting."""
    prev = _lib.lib().mvdetr_msda_set_forward_impl(_IMPLS[name])
    return {v: k for k, v in _IMPLS.items()}[prev]



def set_backward_deterministic(on: bool) -> bool:
    """Opt into the bit-reproducible backward (``mvdetr_msda_set_backward_deterministic``, include/mvdetr_ops.h): grad_value is
    summed in 64-bit fixed point instead of with fp32 atomics (the reference's atomicAdd, cuh:125-152, is not reproducible
    either).  Deformable-encoder calls only; any other backward raises while the mode is on.  Returns the previous state."""
    return bool(_lib.lib().mvdetr_msda_set_backward_deterministic(1 if on else 0))


def backward_deterministic() -> bool:
    """Whether the bit-reproducible backward is on (the library holds the state; MVDETR_MSDA_BWD_DETERMINISTIC=1 sets its
    initial value)."""
    return bool(_lib.lib().mvdetr_msda_get_backward_deterministic())


def backward_deterministic_supported(B: int, S: int, M: int, D: int, L: int, Lq: int, P: int) -> bool:
    """The calls the deterministic mode serves (csrc/msda_backward_onepass.hip: msda_backward_deterministic_supported): fp32
    deformable-encoder calls with 16-channel heads and 4 points, at most 16 levels, one batch element's tensors below 2 GiB and
    S * L * P < 2^24 (the 64-bit sums' headroom).  Everything else is refused with hipErrorNotSupported while the mode is on."""
    return (D == 16 and P == 4 and Lq == S and 1 <= L <= 16 and S * M * D * 4 < 2 ** 31 and S * M * L * P * 2 * 4 < 2 ** 31
            and S * L * P < 2 ** 24)


def release_scratch() -> None:
    """Hand back the deterministic mode's cached accumulators (``mvdetr_msda_release_scratch``); call it when no backward is
    in flight."""
    _lib.check(_lib.lib().mvdetr_msda_release_scratch(), "release_scratch")
