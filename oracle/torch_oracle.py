"""CPU oracle (torch formulation) for the multiview fusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mvdetr_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the timed CPU baseline.

It restates, with plain torch CPU ops, the algorithm the reference runs when no
GPU extension is involved:

  * ``msda_core``      <- ms_deform_attn_core_pytorch
                          (multiview_detector/models/ops/functions/ms_deform_attn_func.py:41-61)
  * ``msda_module``    <- MSDeformAttn.forward arithmetic around the core
                          (multiview_detector/models/ops/modules/ms_deform_attn.py:92-117)
  * ``warp_perspective`` <- kornia.warp_perspective as called at
                          multiview_detector/models/mvdetr.py:194-195

Parity status
-------------
* ``msda_core`` / ``msda_module`` are PINNED: tests/golden/make_golden.py imports the
  reference's own functions in the build container and stores their outputs
  (tests/golden/*.npz); tests/test_oracle.py checks this file against them bit-for-bit
  (fp64) / to 1e-6 (fp32).  The reference's CUDA sources are unbuildable here (CUDA-only,
  THC headers), so there is no ``oracle/_ref``.
* ``warp_perspective``: PARITY UNPINNED for the grid-generation half.  kornia is an
  un-pinned third-party dependency of the reference (README.md:42 lists "kornia" with no
  version; the repository dates to 2021-04, i.e. the kornia 0.5.x series), it is not vendored
  under /root/reference and not installed in this image, and the reference has no test or
  golden vector at that call site.  The function below restates kornia 0.5's published
  algorithm (normalize_homography -> inverse -> transform_points over a normalised meshgrid ->
  F.grid_sample); the sampler half is torch's own ``F.grid_sample`` and therefore exact.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """Multi-scale deformable attention, the reference's pure-PyTorch formulation.

    value [B,S,M,D]; spatial_shapes [L,2] (H,W); sampling_locations [B,Lq,M,L,P,2] as (x,y) in
    [0,1]; attention_weights [B,Lq,M,L,P]  ->  [B,Lq,M*D].
    (ms_deform_attn_func.py:41-61: per level, fold heads into the batch, sample with
    grid_sample(bilinear, zeros, align_corners=False) at 2*loc-1, weight and sum over L*P.)
    """
    B, S, M, D = value.shape
    Lq, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    sizes = [(int(h), int(w)) for h, w in spatial_shapes]
    grids = sampling_locations * 2 - 1
    start = 0
    per_level = []
    for lvl, (H, W) in enumerate(sizes):
        plane = value[:, start:start + H * W]                       # [B, HW, M, D]
        start += H * W
        plane = plane.permute(0, 2, 3, 1).reshape(B * M, D, H, W)   # heads into batch, NCHW
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(B * M, Lq, P, 2)
        per_level.append(F.grid_sample(plane, g, mode="bilinear", padding_mode="zeros",
                                       align_corners=False))      # [B*M, D, Lq, P]
    sampled = torch.stack(per_level, dim=3)                         # [B*M, D, Lq, L, P]
    w = attention_weights.permute(0, 2, 1, 3, 4).reshape(B * M, 1, Lq, L, P)
    out = (sampled * w).sum(dim=(3, 4))                             # [B*M, D, Lq]
    return out.reshape(B, M * D, Lq).transpose(1, 2).contiguous()


def msda_sampling_locations(reference_points, sampling_offsets, spatial_shapes):
    """loc = ref[:, :, None] + off / (W_l, H_l) with the MVDeTr 5-D reference points
    [B,Lq,L,P,2] (ms_deform_attn.py:104-107)."""
    normalizer = torch.stack([spatial_shapes[:, 1], spatial_shapes[:, 0]], -1).to(sampling_offsets.dtype)
    return reference_points[:, :, None] + sampling_offsets / normalizer[None, None, None, :, None, :]


def msda_module(params, query, reference_points, input_flatten, spatial_shapes, n_heads, n_points,
                padding_mask=None, return_intermediates=False):
    """MSDeformAttn.forward (ms_deform_attn.py:92-117) as a function of a parameter dict with the
    reference's names: sampling_offsets.{weight,bias}, attention_weights.{...}, value_proj.{...},
    output_proj.{...}."""
    B, Lq, C = query.shape
    S = input_flatten.shape[1]
    L = spatial_shapes.shape[0]
    value = F.linear(input_flatten, params["value_proj.weight"], params["value_proj.bias"])
    if padding_mask is not None:
        value = value.masked_fill(padding_mask[..., None], 0.0)
    value = value.view(B, S, n_heads, C // n_heads)
    off = F.linear(query, params["sampling_offsets.weight"], params["sampling_offsets.bias"]
                   ).view(B, Lq, n_heads, L, n_points, 2)
    aw = F.linear(query, params["attention_weights.weight"], params["attention_weights.bias"]
                  ).view(B, Lq, n_heads, L * n_points)
    aw = F.softmax(aw, -1).view(B, Lq, n_heads, L, n_points)
    loc = msda_sampling_locations(reference_points, off, spatial_shapes)
    core = msda_core(value, spatial_shapes, loc, aw)
    out = F.linear(core, params["output_proj.weight"], params["output_proj.bias"])
    if return_intermediates:
        return out, loc, aw, value
    return out


# ----------------------------------------------------------------------------------------------
# kornia.warp_perspective (kornia 0.5.x semantics; see the parity note in the module docstring)
# ----------------------------------------------------------------------------------------------

def _normal_transform_pixel(height, width, dtype, eps=1e-14):
    """Pixel -> [-1, 1] (corner-aligned) normalisation matrix, kornia normal_transform_pixel."""
    t = torch.tensor([[1.0, 0.0, -1.0], [0.0, 1.0, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)
    wd = eps if width == 1 else width - 1.0
    hd = eps if height == 1 else height - 1.0
    t[0, 0] = t[0, 0] * 2.0 / wd
    t[1, 1] = t[1, 1] * 2.0 / hd
    return t


def warp_grid(M, src_hw, dst_hw):
    """Sampling grid [N, H_out, W_out, 2] in grid_sample's normalised units.

    dst_norm<-src_norm = N_dst @ M @ inv(N_src); inverted; applied to a linspace(-1,1) meshgrid
    with a homogeneous divide that only divides where |z| > 1e-8 (kornia
    convert_points_from_homogeneous)."""
    dtype = M.dtype
    h, w = src_hw
    H, W = dst_hw
    n_src = _normal_transform_pixel(h, w, dtype)
    n_dst = _normal_transform_pixel(H, W, dtype)
    dst_from_src = n_dst @ (M @ torch.inverse(n_src))
    src_from_dst = torch.inverse(dst_from_src)                       # [N,3,3]
    xs = torch.linspace(-1, 1, W, dtype=dtype)
    ys = torch.linspace(-1, 1, H, dtype=dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    pts = torch.stack([gx, gy, torch.ones_like(gx)], -1)           # [H,W,3]
    hom = torch.einsum("hwk,njk->nhwj", pts, src_from_dst)        # [N,H,W,3]
    z = hom[..., 2:3]
    scale = torch.where(z.abs() > 1e-8, 1.0 / z, torch.ones_like(z))
    return hom[..., :2] * scale


def warp_perspective(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=False):
    """src [N,C,h,w], M [N,3,3] (dst pixel <- src pixel), dsize (H,W) -> [N,C,H,W]."""
    grid = warp_grid(M.to(src.dtype), src.shape[-2:], dsize)
    return F.grid_sample(src, grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


# ----------------------------------------------------------------------------------------------
# Callers either side of the kernels (kept here so full-frame parity has a CPU checker)
# ----------------------------------------------------------------------------------------------

def create_pos_embedding(img_size, num_pos_feats=64, temperature=10000, scale=None):
    """Normalised sine position embedding [1, 2*num_pos_feats, H, W]
    (multiview_detector/models/trans_world_feat.py:15-37)."""
    import math

    if scale is None:
        scale = 2 * math.pi
    H, W = int(img_size[0]), int(img_size[1])
    ones = torch.ones(1, H, W)
    y = ones.cumsum(1, dtype=torch.float32)
    x = ones.cumsum(2, dtype=torch.float32)
    y = y / (y[:, -1:, :] + 1e-6) * scale
    x = x / (x[:, :, -1:] + 1e-6) * scale
    k = torch.arange(num_pos_feats, dtype=torch.float32)
    freq = temperature ** (2 * (k // 2) / num_pos_feats)
    px = x[..., None] / freq
    py = y[..., None] / freq
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def encoder_layer(p, prefix, src, pos, ref, shapes, n_heads, n_points):
    """DeformableTransformerEncoderLayer.forward in eval mode
    (multiview_detector/models/deformable_transformer.py:75-85)."""
    sub = {k[len(prefix) + len("self_attn."):]: v for k, v in p.items()
           if k.startswith(prefix + "self_attn.")}
    a = msda_module(sub, src + pos, ref, src, shapes, n_heads, n_points)
    d = src.shape[-1]
    src = F.layer_norm(src + a, (d,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])
    f = F.linear(F.relu(F.linear(src, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"])),
                 p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])
    return F.layer_norm(src + f, (d,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"])


def deform_trans_world_feat(p, x, reference_points, n_heads=8, n_points=4, stride=2, n_layers=3):
    """DeformTransWorldFeat.forward in eval mode (trans_world_feat.py:87-110) for B == 1 (the only
    batch size the reference supports, trans_world_feat.py:94) and, by broadcasting the level
    embedding, for B > 1 as well.  ``p`` is the module's state dict."""
    B, N, C, H, W = x.shape
    y = F.relu(F.conv2d(x.view(B * N, C, H, W), p["downsample.0.weight"], p["downsample.0.bias"],
                        stride=stride, padding=1))
    Ch, h, w = y.shape[1:]
    src = y.view(B, N, Ch, h, w).permute(0, 1, 3, 4, 2).reshape(B, N * h * w, Ch)
    pos = create_pos_embedding((h, w), Ch // 2).flatten(2).transpose(1, 2).unsqueeze(1)   # [1,1,hw,C]
    pos = (pos + p["lvl_embedding"].view(1, N, 1, Ch)).reshape(1, N * h * w, Ch).expand(B, -1, -1)
    shapes = torch.tensor([[h, w]] * N, dtype=torch.long)
    ref = reference_points.unsqueeze(0).expand(B, -1, -1, -1, -1)
    for i in range(n_layers):
        src = encoder_layer(p, f"encoder.layers.{i}.", src, pos, ref, shapes, n_heads, n_points)
    mem = src.view(B, N, h, w, Ch).permute(0, 1, 4, 2, 3).reshape(B, N * Ch, h, w)
    m = F.relu(F.conv2d(mem, p["merge_linear.0.weight"], p["merge_linear.0.bias"]))
    m = F.interpolate(m, size=(H, W), mode="bilinear", align_corners=False)
    return F.relu(F.conv2d(m, p["upsample.1.weight"], p["upsample.1.bias"], padding=1))


def conv_world_feat(p, x):
    """ConvWorldFeat.forward (multiview_detector/models/conv_world_feat.py:9-14,38-52, reduction=None), functional:
    p = its state dict.  x [B, N, C, H, W] -> [B, C, H, W]."""
    B, N, C, H, W = x.shape
    y = F.relu(F.conv2d(x.reshape(B * N, C, H, W), p["downsample.0.weight"], p["downsample.0.bias"], stride=2, padding=1))
    h, w = y.shape[-2:]
    gx, gy = np.meshgrid(np.arange(w), np.arange(h))
    coord = torch.stack([torch.from_numpy(gx / (w - 1) * 2 - 1).float(), torch.from_numpy(gy / (h - 1) * 2 - 1).float()], 0)
    y = torch.cat([y.reshape(B, N * y.shape[1], h, w), coord.unsqueeze(0).repeat(B, 1, 1, 1)], 1)
    y = F.relu(F.conv2d(y, p["world_feat.0.weight"], p["world_feat.0.bias"], padding=1))
    y = F.relu(F.conv2d(y, p["world_feat.2.weight"], p["world_feat.2.bias"], padding=2, dilation=2))
    y = F.relu(F.conv2d(y, p["world_feat.4.weight"], p["world_feat.4.bias"], padding=4, dilation=4))
    y = F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False)
    return F.relu(F.conv2d(y, p["upsample.1.weight"], p["upsample.1.bias"], padding=1))
