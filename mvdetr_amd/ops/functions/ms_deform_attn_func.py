"""autograd glue for multi-scale deformable attention.

Mirrors multiview_detector/models/ops/functions/ms_deform_attn_func.py:21-38: saves the five
inputs, calls the extension's forward / backward, returns gradients for (value, sampling
locations, attention weights) only, once-differentiable.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        out = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                          sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, aw = ctx.saved_tensors
        g_value, g_loc, g_aw = MSDA.ms_deform_attn_backward(value, shapes, start, loc, aw,
                                                            grad_output.contiguous(), ctx.im2col_step)
        return g_value, None, None, g_loc, g_aw, None


class MSDeformAttnFusedFunction(Function):
    """The fused TRAINING pair (no counterpart in the reference; SURVEY row f1 for the training path): MSDeformAttn.forward
    from the module's raw offsets / logits, with softmax and location arithmetic (ms_deform_attn.py:100-107) inside the
    kernels in both directions.  Inputs: value [B, S, M, D]; ``raw`` [B, Lq, M*L*12] -- the ONE Linear output in the
    slice-interleaved, level-outermost layout (MSDA.slice_major_rows(level_outer=True)); ``reference_points`` [B or 1, L, Lq, 2].
    Saves value, raw, the output and the softmax statistics [B, Lq, M, 2] -- not the 203 MB of sampling locations and
    attention weights the unfused path keeps for its backward.  Gradients: value and raw; once-differentiable like the
    reference's function (func.py:29)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, reference_points, raw):
        out, stats = MSDA.ms_deform_attn_forward_fused_train(value, value_spatial_shapes, value_level_start_index,
                                                             reference_points, raw)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, reference_points, raw, stats, out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, ref, raw, stats, out = ctx.saved_tensors
        g_value, g_raw = MSDA.ms_deform_attn_backward_fused(grad_output.contiguous(), value, shapes, start, ref, raw, stats, out)
        return g_value, None, None, None, g_raw


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Debug-only torch formulation kept for API parity with the reference
    (ms_deform_attn_func.py:41-61, "for debug and test only").

    NEVER called by MSDeformAttnFunction, MSDeformAttn or anything else in this package: the
    product path is the HIP extension and raises without it.  (The test-suite's checker is the
    separate restatement under oracle/.)
    """
    B, S, M, D = value.shape
    Lq, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    out = value.new_zeros(B * M, D, Lq)
    start = 0
    for lvl in range(L):
        H, W = int(value_spatial_shapes[lvl][0]), int(value_spatial_shapes[lvl][1])
        feat = value[:, start:start + H * W].permute(0, 2, 3, 1).reshape(B * M, D, H, W)
        start += H * W
        grid = (2 * sampling_locations[:, :, :, lvl] - 1).transpose(1, 2).reshape(B * M, Lq, P, 2)
        taps = F.grid_sample(feat, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
        w = attention_weights[:, :, :, lvl].transpose(1, 2).reshape(B * M, 1, Lq, P)
        out = out + (taps * w).sum(-1)
    return out.view(B, M * D, Lq).transpose(1, 2).contiguous()
