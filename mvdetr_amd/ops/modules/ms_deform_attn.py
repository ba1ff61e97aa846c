"""MSDeformAttn module with the MVDeTr contract (5-D reference points).

Mirrors multiview_detector/models/ops/modules/ms_deform_attn.py:30-117: same constructor,
parameter names (sampling_offsets, attention_weights, value_proj, output_proj -- reference
checkpoints load unchanged), initialisation, and forward arithmetic; the core runs on the HIP
extension through MSDeformAttnFunction.
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from .. import MultiScaleDeformableAttention as MSDA
from ..functions import MSDeformAttnFunction


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError(f"invalid input for _is_power_of_2: {n} (type: {type(n)})")
    return n != 0 and (n & (n - 1)) == 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        """d_model hidden size; n_levels feature levels (= cameras in MVDeTr); n_heads attention
        heads; n_points sampling points per head per level."""
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a power-of-two head dimension maps best onto the wave64 "
                          "lane groups of the HIP kernels.")
        self.im2col_step = 64          # kept for API parity; the HIP kernels do not chunk the batch
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._validated = None
        self._validated_q = None
        self._shared_ref_cache = None
        # inference calls of deformable-encoder shape run softmax + location arithmetic inside the kernel
        self.fused_inference = True
        self._fused_cache = None
        self._reset_parameters()

    def _reset_parameters(self):
        # ms_deform_attn.py:62-77: zero offset weights, bias = unit 8-neighbourhood directions
        # scaled by (point index + 1); zero attention logits (uniform weights); xavier projections
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        ang = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        dirs = torch.stack([ang.cos(), ang.sin()], -1)
        dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
        grid = dirs.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def _fused_projection(self):
        """Weights of ONE Linear producing [offsets | logits] with output rows reordered from the reference's
        (head, level, point) to (level, head, point): a free change of the output layout that lets the kernel
        read what one level iteration needs from the same cache lines.  Cached until a parameter changes."""
        ps = (self.sampling_offsets.weight, self.sampling_offsets.bias, self.attention_weights.weight,
              self.attention_weights.bias)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._fused_cache is None or self._fused_cache[0] != key:
            M, L, P, C = self.n_heads, self.n_levels, self.n_points, self.d_model
            with torch.no_grad():
                ow = ps[0].view(M, L, P * 2, C).transpose(0, 1).reshape(M * L * P * 2, C)
                ob = ps[1].view(M, L, P * 2).transpose(0, 1).reshape(-1)
                aw = ps[2].view(M, L, P, C).transpose(0, 1).reshape(M * L * P, C)
                ab = ps[3].view(M, L, P).transpose(0, 1).reshape(-1)
                self._fused_cache = (key, torch.cat([ow, aw], 0).contiguous(), torch.cat([ob, ab], 0).contiguous())
        return self._fused_cache[1], self._fused_cache[2]

    def _check_lengths(self, spatial_shapes, len_in):
        # the reference asserts sum(H*W) == Len_in on every call (ms_deform_attn.py:94), which is a
        # device->host sync per layer; validate each (tensor, length) pair once instead
        key = (spatial_shapes.data_ptr(), spatial_shapes._version, tuple(spatial_shapes.shape), len_in)
        if self._validated != key:
            assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == len_in
            self._validated = key

    def project_value(self, input_flatten, input_padding_mask=None):
        """value_proj + padding mask (ms_deform_attn.py:96-98) on any subset of the tokens: (N, n, C) ->
        (N, n, C).  Split out so that a query-sharded run can project its own tokens and all-gather the result
        (mvdetr_amd/dist.py) instead of projecting every token on every rank."""
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        return value

    def _shared_reference(self, reference_points):
        """[N, Lq, L, P, 2] -> contiguous [1 or N, Lq, L, 2] when the P points of every (query, level) are the same
        point (MVDeTr's reference map, mvdetr.py:49-58 with all heights 0), else None.  Checked once per tensor
        (one device sync) and cached: the fused kernel then loads a quarter of the reference bytes."""
        key = (reference_points.data_ptr(), reference_points._version, tuple(reference_points.shape),
               reference_points.stride())
        if self._shared_ref_cache is None or self._shared_ref_cache[0] != key:
            ref = reference_points[:1] if reference_points.stride(0) == 0 else reference_points
            first = ref[..., :1, :]
            same = bool((ref == first).all())
            self._shared_ref_cache = (key, first.squeeze(-2).contiguous() if same else None)
        return self._shared_ref_cache[1]

    def _check_query_levels(self, spatial_shapes, query_levels, len_q):
        key = (spatial_shapes.data_ptr(), spatial_shapes._version, tuple(query_levels), len_q)
        if self._validated_q != key:
            l0, l1 = query_levels
            assert 0 <= l0 < l1 <= spatial_shapes.shape[0]
            assert int((spatial_shapes[l0:l1, 0] * spatial_shapes[l0:l1, 1]).sum()) == len_q
            self._validated_q = key

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, *, query_levels=None, projected_value=None):
        """query (N, Lq, C); reference_points (N, Lq, n_levels, n_points, 2) in [0,1] -- MVDeTr's 5-D
        form (ms_deform_attn.py:104-107) -- or (..., 4) boxes; input_flatten (N, sum H_l*W_l, C);
        input_spatial_shapes (n_levels, 2); input_level_start_index (n_levels,);
        input_padding_mask (N, sum H_l*W_l) True = padding.  Returns (N, Lq, C).

        Two keyword extensions for the query-sharded encoder (not in the reference): ``projected_value``
        (N, sum H_l*W_l, C) = project_value() of all tokens, used instead of projecting ``input_flatten`` here;
        ``query_levels=(l0, l1)`` promises that the Lq queries are exactly the tokens of levels l0..l1-1, which
        lets the fused kernel take the call (it is only a hint: results do not depend on it)."""
        N, Len_q, _ = query.shape
        M, D = self.n_heads, self.d_model // self.n_heads
        # ``projected_value`` may also be a zero-argument callable returning the tensor, with a ``length`` attribute
        # (= sum H_l*W_l): a value that is still in flight (an asynchronous all-gather).  It is resolved as late as
        # possible -- after the offsets/logits GEMM has been enqueued -- so that the collective overlaps that GEMM.
        pending = projected_value if callable(projected_value) else None
        if pending is not None:
            value, Len_in = None, int(pending.length)
        elif projected_value is not None:
            value, Len_in = projected_value, projected_value.shape[1]
        else:
            Len_in = input_flatten.shape[1]
            value = self.project_value(input_flatten, input_padding_mask)
        self._check_lengths(input_spatial_shapes, Len_in)
        if query_levels is not None:
            self._check_query_levels(input_spatial_shapes, query_levels, Len_q)
        needs_grad = torch.is_grad_enabled() and ((value is not None and value.requires_grad) or query.requires_grad
                                                  or self.sampling_offsets.weight.requires_grad)
        if (self.fused_inference and reference_points.shape[-1] == 2 and reference_points.dim() == 5
                and not needs_grad and query.is_cuda and query.dtype == torch.float32
                and (value is None or (value.is_cuda and value.dtype == torch.float32))
                and MSDA.fused_supported_dims(N, Len_in, M, D, self.n_levels, Len_q, self.n_points, query_levels)):
            # one GEMM for offsets + logits, rows permuted to level-major (see _fused_projection)
            w, b = self._fused_projection()
            raw = F.linear(query, w, b)
            if pending is not None:
                value = pending()
            value = value.view(N, Len_in, M, D)
            n_off = self.n_heads * self.n_levels * self.n_points * 2
            L, M, P = self.n_levels, self.n_heads, self.n_points
            shared = self._shared_reference(reference_points)
            out = MSDA.ms_deform_attn_forward_fused(
                value.contiguous(), input_spatial_shapes, input_level_start_index,
                reference_points if shared is None else shared,
                raw[..., :n_off].unflatten(-1, (L, M, P, 2)), raw[..., n_off:].unflatten(-1, (L, M, P)),
                level_major=True, query_levels=query_levels)
            return self.output_proj(out)
        if pending is not None:
            value = pending()
        value = value.view(N, Len_in, M, D)
        offsets = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        weights = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
        weights = F.softmax(weights, -1).view(N, Len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, :, :] \
                + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead."
                             .format(reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                         locations.contiguous(), weights, self.im2col_step)
        return self.output_proj(out)
