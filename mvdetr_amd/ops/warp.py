"""warp_perspective: the feature -> ground-plane homography warp of MVDeTr as one HIP kernel.

Drop-in for the reference's ``kornia.warp_perspective(src, M, dsize, mode='bilinear',
padding_mode='zeros', align_corners=False)`` call (multiview_detector/models/mvdetr.py:194-195;
also frameDataset.py:80, grid_visualize.py:19-21).  Differentiable w.r.t. ``src``.
"""
from __future__ import annotations

import os
import threading

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


DST_NHWC, SRC_NHWC, NEAREST = 1, 2, 4          # bits of the C ABI's layout_nhwc argument (include/mvdetr_ops.h)


def _launch(name, a, M, n, c, h, w, H, W, layout, out, plan=None, tag=0):
    """One C-ABI call: ``name`` = "forward" | "backward"; with ``plan`` (backward only) the two-step gradient's second step
    (mvdetr_warp_perspective_backward_planned_*); with ``tag`` != 0 (backward only) the one-call gradient with a version tag
    of the matrices (mvdetr_warp_perspective_backward_tagged_*: the library reuses its own plan while the tag stays)."""
    if tag and plan is None and a.is_cuda and name == "backward":
        with torch.cuda.device(a.device):
            rc = getattr(_lib.lib(), f"mvdetr_warp_perspective_backward_tagged_{_lib.suffix(a.dtype)}")(
                _lib.current_stream_ptr(a.device), a.data_ptr(), M.data_ptr(), n, c, h, w, H, W, layout, int(tag), out.data_ptr())
        _lib.check(rc, "warp_perspective_backward_tagged")
        return
    if plan is not None:
        with torch.cuda.device(a.device):
            rc = getattr(_lib.lib(), f"mvdetr_warp_perspective_backward_planned_{_lib.suffix(a.dtype)}")(
                _lib.current_stream_ptr(a.device), a.data_ptr(), M.data_ptr(), plan.data_ptr(), n, c, h, w, H, W, layout,
                out.data_ptr())
        _lib.check(rc, "warp_perspective_backward_planned")
        return
    if not a.is_cuda:
        # the library's own CPU path (csrc/host_path.cpp): same layouts; the interpolation mode is its own argument
        rc = getattr(_lib.lib(), f"mvdetr_warp_perspective_{name}_host_{_lib.suffix(a.dtype)}")(
            a.data_ptr(), M.data_ptr(), n, c, h, w, H, W, layout & 3, 1 if layout & NEAREST else 0, out.data_ptr())
        _lib.check(rc, f"warp_perspective_{name} (host)")
        return
    with torch.cuda.device(a.device):
        rc = getattr(_lib.lib(), f"mvdetr_warp_perspective_{name}_{_lib.suffix(a.dtype)}")(
            _lib.current_stream_ptr(a.device), a.data_ptr(), M.data_ptr(), n, c, h, w, H, W, layout,
            out.data_ptr())
    _lib.check(rc, f"warp_perspective_{name}")


def _transpose(x, n, rows, cols):
    """[n, rows, cols] -> [n, cols, rows] (contiguous CUDA tensor) with the library's tiled transpose (grid limits:
    n <= 65535 and rows <= 64 * 65535; beyond them a strided torch copy does the same job)."""
    if n > 65535 or (rows + 63) // 64 > 65535:
        return x.view(n, rows, cols).transpose(1, 2).contiguous()
    out = torch.empty((n, cols, rows), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        rc = getattr(_lib.lib(), f"mvdetr_transpose_{_lib.suffix(x.dtype)}")(
            _lib.current_stream_ptr(x.device), x.data_ptr(), n, rows, cols, out.data_ptr())
    _lib.check(rc, "transpose")
    return out


def last_kernel() -> str:
    """Name of the device kernel the last warp call of this PROCESS launched -- any thread: autograd runs backwards on its
    own thread (tests / bench introspection)."""
    return _lib.lib().mvdetr_warp_last_kernel().decode()


def release_scratch() -> None:
    """Drop the gather backward's cached geometry scratch of every (device, stream): include/mvdetr_ops.h,
    mvdetr_warp_release_scratch.  Call when no warp backward is in flight (e.g. after destroying streams)."""
    _lib.check(_lib.lib().mvdetr_warp_release_scratch(), "mvdetr_warp_release_scratch")


def _channel_last_source(src, channels_last_out):
    """True when ``src`` ([N,C,h,w] shape) already lies in memory as [N,h,w,C] and a channel-last kernel takes it
    (NHWC destination: any float dtype; NCHW destination: float32): then the warp reads it in place instead of
    copying it to NCHW first."""
    n, c, h, w = src.shape
    return ((channels_last_out or src.dtype == torch.float32) and src.is_cuda and c > 1 and h * w > 1
            and src.is_contiguous(memory_format=torch.channels_last)
            and not src.is_contiguous() and (c * src.element_size()) % 16 == 0 and src.data_ptr() % 16 == 0)


class _BackwardPlans:
    """Plans of the gather backward (csrc/warp_perspective.hip: which destination pixels can touch each 2x2 block of source
    texels) for the matrices seen lately.  A plan depends on M and the shapes only, so training without augmentation -- the
    same projection matrices every iteration -- builds it once and every backward is ONE kernel.  Entries hold the matrix
    tensor itself (its storage cannot be handed to another tensor while cached) and its version counter; writes through
    ``.data`` are invisible to it, like to autograd.

    Autograd runs backwards on one thread per device, so the cache is guarded by a lock and partitioned per (device, stream):
    a plan is only ever reused on the stream that built it (no cross-stream ordering to get wrong), every partition keeps its
    own ``KEEP`` most recent plans (a DataParallel run's replicas do not evict each other), and the library's geometry knobs
    (``MVDETR_WARP_BWD_GEOMETRY`` / ``MVDETR_WARP_BWD_HEAVY``, which change a plan's contents) are part of the key."""
    KEEP = 4
    MAX_PARTS = 8                  # (device, stream) partitions kept, least recently used first out: short-lived side streams
                                   # (and recycled stream handles) must not pin plans and matrices for the life of the process

    def __init__(self):
        self.lock = threading.Lock()
        self.parts = {}            # (device index, stream pointer) -> [(M, version, key, plan), ...], most recent first;
                                   # the dict itself is kept in order of last use (oldest partition first)

    def get(self, M, shapes):
        n, c, h, w, H, W = shapes
        stream = _lib.current_stream_ptr(M.device)
        part_key = (M.device.index, int(stream or 0))
        key = (shapes, os.environ.get("MVDETR_WARP_BWD_GEOMETRY"), os.environ.get("MVDETR_WARP_BWD_HEAVY"))
        with self.lock:
            entries = self.parts.pop(part_key, [])
            self.parts[part_key] = entries                  # (most recently used partition last)
            for i, (m, ver, k, plan) in enumerate(entries):
                if k == key and m.data_ptr() == M.data_ptr() and ver == M._version and m.dtype == M.dtype:
                    if i:
                        entries.insert(0, entries.pop(i))
                    return plan
        nbytes = int(_lib.lib().mvdetr_warp_backward_plan_bytes(n, c, h, w, H, W, M.element_size()))
        if nbytes <= 0:
            return None
        plan = torch.empty(nbytes, dtype=torch.uint8, device=M.device)
        with torch.cuda.device(M.device):
            rc = getattr(_lib.lib(), f"mvdetr_warp_backward_plan_{_lib.suffix(M.dtype)}")(
                stream, M.data_ptr(), n, c, h, w, H, W, plan.data_ptr())
        if rc == 801:                                        # hipErrorNotSupported: the scatter kernels take the call
            return None
        _lib.check(rc, "warp_backward_plan")
        with self.lock:
            entries = self.parts.pop(part_key, [])
            self.parts[part_key] = entries
            entries.insert(0, (M, M._version, key, plan))
            del entries[self.KEEP:]
            while len(self.parts) > self.MAX_PARTS:
                del self.parts[next(iter(self.parts))]
        return plan

    def clear(self):
        with self.lock:
            self.parts.clear()

    @property
    def entries(self):
        """Every cached (M, version, key, plan), most recent first within its (device, stream) partition (tests, bench)."""
        with self.lock:
            return [e for part in self.parts.values() for e in part]


_plans = _BackwardPlans()


class WarpPerspectiveFunction(Function):
    @staticmethod
    def forward(ctx, src, M, dsize, channels_last_out, nearest=False):
        n, c, h, w = src.shape
        H, W = int(dsize[0]), int(dsize[1])
        src_cl = _channel_last_source(src, channels_last_out)
        _channel_last_source_flag = src_cl                         # the caller's own layout (the gradient's layout)
        if not src_cl:
            src = src.contiguous()
            if (channels_last_out and src.is_cuda and c > 1 and h * w > 1
                    and (c * src.element_size()) % 16 == 0 and h * w * c < 2 ** 31 and n <= 65535):
                # an NCHW source for a channel-last destination: one tiled transpose (reads and writes in 256-byte runs)
                # and the channel-last kernel, instead of the NCHW kernel's 4-byte gathers (92 -> 56 us at Wildtrack
                # size).  NCHW -> NCHW keeps the gather kernel (90 us): transpose + warp_fwd_cl_nchw measured 142 us.
                src = _transpose(src, n, c, h * w)                 # memory is now [n, h, w, c]
                src_cl = True
        layout = (DST_NHWC if channels_last_out else 0) | (SRC_NHWC if src_cl else 0) | (NEAREST if nearest else 0)
        shape = (n, H, W, c) if channels_last_out else (n, c, H, W)
        out = torch.empty(shape, dtype=src.dtype, device=src.device)
        _launch("forward", src, M, n, c, h, w, H, W, layout, out)
        ctx.save_for_backward(M)
        ctx.geom = (n, c, h, w, H, W, layout)
        ctx.src_was_cl = _channel_last_source_flag
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (M,) = ctx.saved_tensors
        n, c, h, w, H, W, layout = ctx.geom
        near = layout & NEAREST
        if grad_out.is_cuda and (c * grad_out.element_size()) % 16 == 0 and h * w * c < 2 ** 31:
            # channel-last on both sides whatever the forward's layouts were: there the gradient is a GATHER over the
            # destination pixels whose footprint touches each source texel (csrc/warp_perspective.hip: no atomics,
            # deterministic, grad_src written once); NCHW gradients are transposed on the way in / out
            g = grad_out.contiguous() if layout & DST_NHWC else _transpose(grad_out.contiguous(), n, c, H * W)
            grad_src = torch.empty((n, c, h, w), dtype=grad_out.dtype, device=grad_out.device,
                                   memory_format=torch.channels_last)
            plan = _plans.get(M, (n, c, h, w, H, W)) if g.data_ptr() % 16 == 0 else None
            if plan is not None:
                _launch("backward", g, M, n, c, h, w, H, W, DST_NHWC | SRC_NHWC | near, grad_src, plan=plan)
            else:
                _launch("backward", g, M, n, c, h, w, H, W, DST_NHWC | SRC_NHWC | near, grad_src)
            if ctx.src_was_cl:
                return grad_src, None, None, None, None
            return _transpose(grad_src.permute(0, 2, 3, 1), n, h * w, c).view(n, c, h, w), None, None, None, None
        # the device entry overwrites grad_src (it zeroes it itself before scattering); the host entry accumulates
        alloc = torch.empty if grad_out.is_cuda else torch.zeros
        grad_src = alloc((n, c, h, w), dtype=grad_out.dtype, device=grad_out.device)
        _launch("backward", grad_out.contiguous(), M, n, c, h, w, H, W, (layout & DST_NHWC) | near, grad_src)
        return grad_src, None, None, None, None


def warp_perspective(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=False,
                     channels_last_out=False):
    """src [N,C,h,w] -> [N,C,H,W] (or [N,H,W,C] with ``channels_last_out``), sampling the source at
    the pre-image of every destination pixel under ``M [N,3,3]`` (destination pixel <- source
    pixel), with kornia's normalisation convention.

    A ``src`` in ``torch.channels_last`` memory format is read in place when ``channels_last_out`` is set
    (no NCHW copy; every bilinear corner is then one contiguous channel vector).

    ``M`` may live on the CPU (mvdetr.py:194 moves it each call); it is copied to ``src``'s device.
    Modes: 'bilinear' (the model, mvdetr.py:194) and 'nearest' (the dataset's ground-plane masks,
    frameDataset.py:80); zero padding and align_corners=False only, the configuration the reference uses.
    CPU tensors run on the library's own host implementation (the reference has none; kornia does).
    """
    if mode not in ("bilinear", "nearest") or padding_mode != "zeros" or align_corners not in (False, None):
        raise NotImplementedError(
            "warp_perspective: mode 'bilinear' or 'nearest', padding_mode='zeros', align_corners=False "
            "(what MVDeTr calls) are implemented")
    if src.dim() != 4 or M.shape[-2:] != (3, 3) or M.reshape(-1, 3, 3).shape[0] != src.shape[0]:
        raise ValueError(f"warp_perspective: expected src [N,C,h,w] and M [N,3,3], got "
                         f"{tuple(src.shape)} and {tuple(M.shape)}")
    if src.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("warp_perspective: float32 or float64 features")
    M = M.reshape(-1, 3, 3).to(device=src.device, dtype=src.dtype).contiguous()
    return WarpPerspectiveFunction.apply(src, M, tuple(dsize), bool(channels_last_out), mode == "nearest")
