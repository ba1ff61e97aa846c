"""CPU oracle of one whole frame (TEST INFRASTRUCTURE ONLY -- see oracle/torch_oracle.py).

Functional restatement, from a state dict, of MVDeTr.forward in eval mode
(multiview_detector/models/mvdetr.py:151-218) with the reference's CPU formulation of the two hot
ops: kornia-semantics warp (torch_oracle.warp_perspective) and the grid_sample-based
ms_deform_attn_core_pytorch inside DeformTransWorldFeat (torch_oracle.deform_trans_world_feat).
Used by tests (BEV parity, <= 1e-4) and as bench.py's cpu_baseline ("port").
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import torch_oracle


def _bn(x, p, k):
    return F.batch_norm(x, p[k + ".running_mean"], p[k + ".running_var"], p[k + ".weight"], p[k + ".bias"],
                        training=False, eps=1e-5)


def _block(x, p, k, stride, dilation):
    # BasicBlock of the reference's fork: only conv1 is dilated (resnet.py:46-51)
    out = F.relu(_bn(F.conv2d(x, p[k + ".conv1.weight"], None, stride, dilation, dilation), p, k + ".bn1"))
    out = _bn(F.conv2d(out, p[k + ".conv2.weight"], None, 1, 1), p, k + ".bn2")
    if k + ".downsample.0.weight" in p:
        x = _bn(F.conv2d(x, p[k + ".downsample.0.weight"], None, stride), p, k + ".downsample.1")
    return F.relu(out + x)


def resnet18_trunk(p, x, prefix="base."):
    """children()[:-2] of resnet18(replace_stride_with_dilation=[False, True, True])
    (mvdetr.py:103-105; resnet.py:137-148, 163-186)."""
    x = F.relu(_bn(F.conv2d(x, p[prefix + "0.weight"], None, 2, 3), p, prefix + "1"))
    x = F.max_pool2d(x, 3, 2, 1)
    # (layer index in the Sequential, stride of block 0, dilation of block 0 conv1, dilation of block 1 conv1)
    for idx, stride, d0, d1 in ((4, 1, 1, 1), (5, 2, 1, 1), (6, 1, 1, 2), (7, 1, 2, 4)):
        x = _block(x, p, f"{prefix}{idx}.0", stride, d0)
        x = _block(x, p, f"{prefix}{idx}.1", 1, d1)
    return x


def _head(p, k, x):
    if k + ".2.weight" in p:
        x = F.relu(F.conv2d(x, p[k + ".0.weight"], p[k + ".0.bias"], padding=1))
        return F.conv2d(x, p[k + ".2.weight"], p[k + ".2.bias"])
    return F.conv2d(x, p[k + ".0.weight"], p[k + ".0.bias"])


def features(p, imgs):
    B, N, C, H, W = imgs.shape
    feat = resnet18_trunk(p, imgs.reshape(B * N, C, H, W))
    if "bottleneck.0.weight" in p:
        feat = F.conv2d(feat, p["bottleneck.0.weight"], p["bottleneck.0.bias"])     # Dropout2d: eval
    return feat


def world_from_features(p, feat, proj, Rworld_shape, reference_points, num_cam, n_heads=8, n_points=4):
    """warp + shadow transformer (the hot path) on the CPU."""
    B = feat.shape[0] // num_cam
    world = torch_oracle.warp_perspective(feat, proj, Rworld_shape).view(B, num_cam, feat.shape[1], *Rworld_shape)
    wf = {k[len("world_feat."):]: v for k, v in p.items() if k.startswith("world_feat.")}
    return torch_oracle.deform_trans_world_feat(wf, world, reference_points, n_heads=n_heads, n_points=n_points)


def forward(p, imgs, proj, Rworld_shape, reference_points, num_cam):
    """-> (world_heatmap, world_offset), (imgs_heatmap, imgs_offset, imgs_wh)"""
    feat = features(p, imgs)
    img_out = (_head(p, "img_heatmap", feat), _head(p, "img_offset", feat), _head(p, "img_wh", feat))
    world = world_from_features(p, feat, proj, Rworld_shape, reference_points, num_cam)
    return (_head(p, "world_heatmap", world), _head(p, "world_offset", world)), img_out
