"""norm(x + residual) in one HIP kernel (inference), for the tails of the shadow transformer's encoder layer
(multiview_detector/models/deformable_transformer.py:96-100).  Training keeps torch's differentiable ops."""
from __future__ import annotations

import torch

from .. import _lib

SUPPORTED_COLS = (64, 128, 256)


def fused_add_layer_norm_available(x: torch.Tensor, norm: torch.nn.LayerNorm) -> bool:
    """True when add_layer_norm() runs the HIP kernel for these arguments: CUDA fp32, LayerNorm over a last
    dimension of 64/128/256 channels, and nothing that needs autograd."""
    if not (x.is_cuda and x.dtype == torch.float32 and len(norm.normalized_shape) == 1
            and norm.normalized_shape[0] == x.shape[-1] and x.shape[-1] in SUPPORTED_COLS):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or (norm.weight is not None and norm.weight.requires_grad)):
        return False
    return (norm.weight is None) == (norm.bias is None)


def add_layer_norm(x: torch.Tensor, residual: torch.Tensor, norm: torch.nn.LayerNorm, then_add: torch.Tensor = None):
    """``norm(x + residual)`` (residual may be None).  One pass over the rows on the GPU when
    fused_add_layer_norm_available(); the module's own torch ops otherwise (training / other shapes).
    With ``then_add`` ([1 or B, n, C], e.g. the position embedding) returns ``(y, y + then_add)`` -- the second
    output comes out of the same pass."""
    fused = fused_add_layer_norm_available(x, norm) and (residual is None or residual.shape == x.shape)
    if then_add is not None:
        fused = fused and (then_add.is_cuda and then_add.dtype == torch.float32 and then_add.dim() == x.dim() == 3
                           and then_add.shape[1:] == x.shape[1:] and then_add.shape[0] in (1, x.shape[0]))
    if not fused:
        y = norm(x if residual is None else x + residual)
        return y if then_add is None else (y, y + then_add)
    xc = x.contiguous()
    rc = None if residual is None else residual.to(torch.float32).contiguous()
    out = torch.empty_like(xc)
    cols = xc.shape[-1]
    rows = xc.numel() // cols
    w = norm.weight.detach().contiguous() if norm.weight is not None else None
    b = norm.bias.detach().contiguous() if norm.bias is not None else None
    a2 = None if then_add is None else then_add.contiguous()
    out2 = None if then_add is None else torch.empty_like(xc)
    with torch.cuda.device(x.device):
        code = _lib.lib().mvdetr_add_layernorm_add_f32(
            _lib.current_stream_ptr(x.device), xc.data_ptr(), 0 if rc is None else rc.data_ptr(),
            0 if w is None else w.data_ptr(), 0 if b is None else b.data_ptr(),
            0 if a2 is None else a2.data_ptr(), 0 if a2 is None else a2.numel() // cols, rows, cols, float(norm.eps),
            out.data_ptr(), 0 if out2 is None else out2.data_ptr())
    _lib.check(code, "add_layer_norm")
    return out if then_add is None else (out, out2)
