"""The shadow transformer: the direct caller of MSDeformAttn on the fusion path.

Mirrors, with identical parameter names (reference checkpoints load unchanged):
  * create_pos_embedding                 multiview_detector/models/trans_world_feat.py:15-37
  * DeformTransWorldFeat                 multiview_detector/models/trans_world_feat.py:70-119
  * DeformableTransformerEncoder(Layer)  multiview_detector/models/deformable_transformer.py:22-85

Deliberate differences:
  * B > 1 works (the reference reshapes the level embedding with the batch size and fails,
    trans_world_feat.py:94);
  * reference points, position embedding, spatial shapes and level offsets are (non-persistent)
    buffers: they move with .to(device) once instead of being re-uploaded every forward
    (deformable_transformer.py:48, trans_world_feat.py:93,95-96) and stay out of the state dict,
    like the reference's plain attributes;
  * the input may arrive channel-last ([B,N,H,W,C], straight from warp_perspective(...,
    channels_last_out=True)); the token tensor is then produced without the NCHW->NHWC permute
    copy of trans_world_feat.py:92.
"""
from __future__ import annotations

import copy
import math

import torch
import torch.nn.functional as F
from torch import nn

from .ops.add_layernorm import add_layer_norm, fused_add_layer_norm_available
from .ops.modules import MSDeformAttn


def create_pos_embedding(img_size, num_pos_feats=64, temperature=10000, normalize=True, scale=None):
    """Sine position embedding ``[1, 2 * num_pos_feats, H, W]`` (same signature and values as trans_world_feat.py:15-37).

    Channel ``k < F`` (F = num_pos_feats) encodes the row, channel ``F + k`` the column of a cell:
    ``sin(c / T**(2 * (k // 2) / F))`` for even k and ``cos`` of the same angle for odd k, where c counts cells from 1
    and, when ``normalize``, is mapped to ``(0, scale]`` by ``c / (size + 1e-6) * scale`` (scale = 2 pi by default).
    Built from two small per-axis tables ([H, F] and [W, F]) that are broadcast over the other axis.
    """
    if scale is not None and not normalize:
        raise ValueError("normalize should be True if scale is passed")
    H, W = int(img_size[0]), int(img_size[1])
    F_ = int(num_pos_feats)
    k = torch.arange(F_, dtype=torch.float32)
    period = temperature ** (2 * (k // 2) / F_)                 # [F]; pairs (2m, 2m + 1) share a period
    odd = (torch.arange(F_) % 2).bool()

    def axis_table(size):
        c = torch.arange(1, size + 1, dtype=torch.float32)
        if normalize:
            c = c / (size + 1e-6) * (2 * math.pi if scale is None else scale)
        angle = c[:, None] / period                             # [size, F]
        return torch.where(odd, angle.cos(), angle.sin())

    rows = axis_table(H).t()[:, :, None].expand(F_, H, W)       # [F, H, W]: constant along x
    cols = axis_table(W).t()[:, None, :].expand(F_, H, W)       # [F, H, W]: constant along y
    return torch.cat((rows, cols), 0)[None].contiguous()


class DeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None, *,
                query_levels=None, projected_value=None, query=None, next_pos=None, shared_reference=None):
        """deformable_transformer.py:88-100.  With ``projected_value`` (self_attn.project_value of ALL tokens)
        ``src``/``pos``/``reference_points`` may hold only the queries of levels ``query_levels`` -- one rank's
        share of a query-sharded layer (mvdetr_amd/dist.py); the layer is per-token apart from the attention.
        ``query``: ``src + pos`` if the caller already has it; ``next_pos``: also return ``output + next_pos`` (the
        next layer's query) -- the encoder uses both to fold the position add into the LayerNorm pass."""
        attn = self.self_attn(self.with_pos_embed(src, pos) if query is None else query, reference_points, src, spatial_shapes,
                              level_start_index, padding_mask, query_levels=query_levels,
                              projected_value=projected_value, shared_reference=shared_reference)
        # eval mode on the GPU: residual add + LayerNorm in one HIP pass (dropout is the identity there);
        # training keeps the reference's differentiable torch ops
        if not self.training and fused_add_layer_norm_available(src, self.norm1):
            src = add_layer_norm(attn, src, self.norm1)
            # bias + ReLU in the GEMM's epilogue (hipBLASLt) instead of a separate pass over [tokens, d_ffn]
            hidden = torch._addmm_activation(self.linear1.bias, src.flatten(0, -2), self.linear1.weight.t(),
                                             use_gelu=False).view(*src.shape[:-1], -1) \
                if hasattr(torch, "_addmm_activation") else F.relu(self.linear1(src))
            return add_layer_norm(self.linear2(hidden), src, self.norm2, then_add=next_pos)
        src = self.norm1(src + self.dropout1(attn))
        ffn = self.linear2(self.dropout2(F.relu(self.linear1(src))))
        out = self.norm2(src + self.dropout3(ffn))
        return out if next_pos is None else (out, out + next_pos)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, reference_points=None):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers
        self.register_buffer("reference_shared", None, persistent=False)
        self._reference_key = None
        if reference_points is not None:
            self.register_buffer("reference_points", reference_points, persistent=False)
            self.shared_reference()
        else:
            self.reference_points = None

    def shared_reference(self):
        """``[L, Lq, 2]`` when every (query, level) holds ONE point repeated n_points times -- MVDeTr's map with all
        heights 0, mvdetr.py:49-58 -- else None.  Derived from ``reference_points`` and re-derived whenever that buffer is
        replaced or written in place (keyed by the tensor object and its ``_version``: a comparison per forward, no device
        sync unless the key moved), so the fused path can never sample a stale copy of the map."""
        rp = self.reference_points
        if rp is None:
            return None
        # (the source tensor itself is kept and compared with `is` -- an id() can be reused by a later tensor of the same shape)
        key = self._reference_key
        if key is None or key[0] is not rp or key[1] != rp._version:
            same = bool((rp == rp[..., :1, :]).all())
            self.reference_shared = rp[..., 0, :].transpose(0, 1).contiguous() if same else None
            self._reference_key = (rp, rp._version)
        return self.reference_shared

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios=None, pos=None, padding_mask=None):
        if self.reference_points is None:
            raise ValueError("MVDeTr's MSDeformAttn takes 5-D reference points [Lq, L, P, 2]; the 4-D "
                             "default of Deformable-DETR is not part of this contract "
                             "(ms_deform_attn.py:104-107)")
        ref = self.reference_points.unsqueeze(0).expand(src.shape[0], -1, -1, -1, -1)
        shared = self.shared_reference()
        shared = None if shared is None else shared.unsqueeze(0)
        out, query = src, None
        for i, layer in enumerate(self.layers):
            if pos is not None and i + 1 < self.num_layers:
                out, query = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask, query=query,
                                   next_pos=pos, shared_reference=shared)   # the next layer's src + pos comes with the LayerNorm
            else:
                out = layer(out, pos, ref, spatial_shapes, level_start_index, padding_mask, query=query,
                            shared_reference=shared)
        return out


class DeformTransWorldFeat(nn.Module):
    def __init__(self, num_cam, Rworld_shape, base_dim, hidden_dim=128, dropout=0.1, nhead=8,
                 dim_feedforward=512, n_points=4, stride=2, reference_points=None):
        super().__init__()
        self.num_cam, self.hidden_dim, self.stride = num_cam, hidden_dim, stride
        self.downsample = nn.Sequential(nn.Conv2d(base_dim, hidden_dim, 3, stride, 1), nn.ReLU())
        layer = DeformableTransformerEncoderLayer(hidden_dim, dim_feedforward, dropout, n_levels=num_cam,
                                                  n_heads=nhead, n_points=n_points)
        self.encoder = DeformableTransformerEncoder(layer, 3, reference_points)
        H, W = int(Rworld_shape[0]) // stride, int(Rworld_shape[1]) // stride
        self.register_buffer("pos_embedding", create_pos_embedding((H, W), hidden_dim // 2), persistent=False)
        shapes = torch.tensor([[H, W]] * num_cam, dtype=torch.long)
        self.register_buffer("spatial_shapes", shapes, persistent=False)
        self.register_buffer("level_start_index",
                             torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1])), persistent=False)
        self.lvl_embedding = nn.Parameter(torch.Tensor(num_cam, hidden_dim))
        self.merge_linear = nn.Sequential(nn.Conv2d(hidden_dim * num_cam, hidden_dim, 1), nn.ReLU())
        self.upsample = nn.Sequential(nn.Upsample(list(map(int, Rworld_shape)), mode="bilinear", align_corners=False),
                                      nn.Conv2d(hidden_dim, hidden_dim, 3, 1, 1), nn.ReLU())
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        nn.init.normal_(self.lvl_embedding)

    def tokens(self, x):
        """[B,N,C,H,W] (or channel-last [B,N,H,W,C]) world features -> ([B, N*h*w, C] tokens, h, w)."""
        if x.shape[2] != self.downsample[0].in_channels:          # channel-last input
            B, N, H, W, C = x.shape
            y = x.reshape(B * N, H, W, C).permute(0, 3, 1, 2)     # NCHW view over NHWC memory
        else:
            B, N, C, H, W = x.shape
            y = x.reshape(B * N, C, H, W)
        y = self.downsample(y)
        h, w = y.shape[-2:]
        tok = y.permute(0, 2, 3, 1).reshape(B, N * h * w, self.hidden_dim)   # free if y is channels_last
        return tok, h, w

    def level_pos(self, h, w):
        """Position + camera embedding of every token, [1, N*h*w, C] (trans_world_feat.py:95-98)."""
        N, C = self.num_cam, self.hidden_dim
        pos = self.pos_embedding.flatten(2).transpose(1, 2).unsqueeze(1)                 # [1,1,hw,C]
        return (pos + self.lvl_embedding.view(1, N, 1, C)).reshape(1, N * h * w, C)      # any B

    def fuse(self, src, B, h, w):
        """[B, N*h*w, C] tokens of ALL cameras -> merged BEV feature [B, C, H, W]: level/position
        embedding, 3 deformable encoder layers, per-camera 1x1 merge, upsample (trans_world_feat.py:93-110).
        Split from tokens() so a view-sharded run can all-gather between the two (mvdetr_amd/dist.py)."""
        N, C = self.num_cam, self.hidden_dim
        lvl_pos = self.level_pos(h, w)
        memory = self.encoder(src, self.spatial_shapes, self.level_start_index, None, lvl_pos)
        merged = memory.view(B, N, h, w, C).permute(0, 1, 4, 2, 3).reshape(B, N * C, h, w)
        return self.upsample(self.merge_linear(merged))

    def forward(self, x, visualize=False):
        src, h, w = self.tokens(x)
        return self.fuse(src, x.shape[0], h, w)


class ConvWorldFeat(nn.Module):
    """MVDet-style aggregation: stride-2 conv per view, the views' channels concatenated with a 2-channel
    coordinate map, three (dilated) 3x3 convolutions, bilinear upsampling back to the world grid and a 3x3 conv
    (multiview_detector/models/conv_world_feat.py:21-52, same parameter names).  No MSDeformAttn involved; this is
    the torch-only plumbing BASELINE.json's config 0 ('--world_feat conv') names.  Like the reference it needs
    hidden_dim == base_dim (it views the down-sampled [B*N, hidden, h, w] as [B, N*base_dim, h, w], l.44); the
    reference's reduction='sum' branch cannot run (it sums a 4-D tensor over its channel axis) and is not offered."""

    def __init__(self, num_cam, Rworld_shape, base_dim, hidden_dim=128, stride=2, reduction=None):
        super().__init__()
        if reduction is not None:
            raise ValueError("only reduction=None exists (the reference's 'sum' branch is broken)")
        if hidden_dim != base_dim:
            raise ValueError("ConvWorldFeat needs hidden_dim == base_dim (conv_world_feat.py:44)")
        H, W = int(Rworld_shape[0]) // stride, int(Rworld_shape[1]) // stride
        self.downsample = nn.Sequential(nn.Conv2d(base_dim, hidden_dim, 3, stride, 1), nn.ReLU())
        gx, gy = torch.meshgrid(torch.arange(W, dtype=torch.float64), torch.arange(H, dtype=torch.float64), indexing="xy")
        coord = torch.stack([gx / (W - 1) * 2 - 1, gy / (H - 1) * 2 - 1], 0).unsqueeze(0).float()   # conv_world_feat.py:9-14
        self.register_buffer("coord_map", coord, persistent=False)
        self.world_feat = nn.Sequential(nn.Conv2d(base_dim * num_cam + 2, hidden_dim, 3, padding=1), nn.ReLU(),
                                        nn.Conv2d(hidden_dim, hidden_dim, 3, padding=2, dilation=2), nn.ReLU(),
                                        nn.Conv2d(hidden_dim, hidden_dim, 3, padding=4, dilation=4), nn.ReLU())
        self.upsample = nn.Sequential(nn.Upsample(list(map(int, Rworld_shape)), mode="bilinear", align_corners=False),
                                      nn.Conv2d(hidden_dim, base_dim, 3, 1, 1), nn.ReLU())

    def forward(self, x, visualize=False):
        B, N, C, H, W = x.shape
        x = self.downsample(x.reshape(B * N, C, H, W))
        h, w = x.shape[-2:]
        x = torch.cat([x.reshape(B, N * C, h, w), self.coord_map.expand(B, -1, -1, -1)], 1)
        return self.upsample(self.world_feat(x))
