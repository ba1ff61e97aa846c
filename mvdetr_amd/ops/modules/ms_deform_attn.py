"""MSDeformAttn module with the MVDeTr contract (5-D reference points).

Mirrors multiview_detector/models/ops/modules/ms_deform_attn.py:30-117: same constructor,
parameter names (sampling_offsets, attention_weights, value_proj, output_proj -- reference
checkpoints load unchanged), initialisation, and forward arithmetic; the core runs on the HIP
extension through MSDeformAttnFunction.
"""
from __future__ import annotations

import math
import os
import warnings
import weakref

import torch
import torch.nn.functional as F
from torch import nn

from .. import MultiScaleDeformableAttention as MSDA
from ..functions import MSDeformAttnFunction, MSDeformAttnFusedFunction


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
        # the permuted [offsets | logits] weight of the fused path is rebuilt on every call unless the owner declares
        # the parameters frozen (cache_fused_projection(True)): a cache cannot see writes through .data
        self._cache_projection = False
        self._fused_cache = None
        self._fused_rows = None
        # calls that need gradients take the fused TRAINING pair (raw offsets / logits in, their gradient out) where it
        # applies; MVDETR_MSDA_FUSED_TRAIN=0 keeps the reference's unfused arithmetic + MSDeformAttnFunction
        self.fused_training = os.environ.get("MVDETR_MSDA_FUSED_TRAIN", "1") != "0"
        self._equal_levels = None
        # order of the fused path's raw tensor: runs [L, M/g] (level outermost) or [M/g, L] -- see slice_major_rows
        self.raw_level_outer = os.environ.get("MVDETR_MSDA_RAW_LAYOUT", "level") != "slice"
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

    def cache_fused_projection(self, enable=True):
        """Inference deployments with frozen parameters: keep the fused path's permuted weight instead of rebuilding it
        (two small gathers) on every call.  Call again, or with False, after changing sampling_offsets /
        attention_weights in ANY way -- writes through ``.data`` (EMA swaps, constant_(w.data)) are invisible to a cache."""
        self._cache_projection = bool(enable)
        self._fused_cache = None
        return self

    def _fused_projection(self):
        """Weight and bias of ONE Linear producing the fused kernel's slice-interleaved raw tensor: the rows of the
        reference's two Linears (offsets (m, l, p, xy), logits (m, l, p); ms_deform_attn.py:55-56) reordered by
        MSDA.slice_major_rows -- a free change of the output layout that puts what one workgroup reads for a
        (query, level) into one contiguous run."""
        if self._cache_projection and self._fused_cache is not None:
            return self._fused_cache
        dev = self.sampling_offsets.weight.device
        if self._fused_rows is None or self._fused_rows.device != dev:
            rows = MSDA.slice_major_rows(self.n_heads, self.n_levels, self.n_points, self.d_model // self.n_heads,
                                         level_outer=self.raw_level_outer)
            self._fused_rows = torch.tensor(rows, dtype=torch.long, device=dev)
        with torch.no_grad():
            w = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0).index_select(0, self._fused_rows)
            b = torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0).index_select(0, self._fused_rows)
        if self._cache_projection:
            self._fused_cache = (w, b)
        return w, b

    def _levels_equal(self, spatial_shapes):
        """All levels of one shape (MVDeTr: levels = cameras)?  One device sync per shapes tensor, remembered for that
        tensor object and version."""
        c = self._equal_levels
        if c is None or c[0]() is not spatial_shapes or c[1] != spatial_shapes._version:
            eq = bool((spatial_shapes == spatial_shapes[:1]).all())
            self._equal_levels = c = (weakref.ref(spatial_shapes), spatial_shapes._version, eq)
        return c[2]

    def _fused_projection_train(self):
        """The same permuted [offsets | logits] Linear as _fused_projection, built under autograd: the row gather is
        differentiable, so grad_raw reaches sampling_offsets / attention_weights through it."""
        dev = self.sampling_offsets.weight.device
        if self._fused_rows is None or self._fused_rows.device != dev:
            rows = MSDA.slice_major_rows(self.n_heads, self.n_levels, self.n_points, self.d_model // self.n_heads,
                                         level_outer=self.raw_level_outer)
            self._fused_rows = torch.tensor(rows, dtype=torch.long, device=dev)
        w = torch.cat([self.sampling_offsets.weight, self.attention_weights.weight], 0).index_select(0, self._fused_rows)
        b = torch.cat([self.sampling_offsets.bias, self.attention_weights.bias], 0).index_select(0, self._fused_rows)
        return w, b

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
        """[N, Lq, L, P, 2] -> level-major [1 or N, L, Lq, 2] when the P points of every (query, level) are the same
        point (MVDeTr's reference map, mvdetr.py:49-58 with all heights 0), else None.  The test costs a device
        sync, so its verdict is remembered -- for that tensor OBJECT (weak reference + version counter), never for an
        address: the caching allocator hands a freed address to the next tensor of the same shape.  Callers that build
        a new view every forward pass ``shared_reference`` to forward() instead (mvdetr_amd/world_feat.py does)."""
        c = self._shared_ref_cache
        if c is not None and c[0]() is reference_points and c[1] == reference_points._version:
            same = c[2]
        else:
            ref = reference_points[:1] if reference_points.stride(0) == 0 else reference_points
            same = bool((ref == ref[..., :1, :]).all())
            self._shared_ref_cache = (weakref.ref(reference_points), reference_points._version, same)
        if not same:
            return None
        ref = reference_points[:1] if reference_points.stride(0) == 0 else reference_points
        return ref[..., 0, :].transpose(1, 2).contiguous()                   # rebuilt from the live tensor every time

    def _check_query_levels(self, spatial_shapes, query_levels, len_q):
        key = (spatial_shapes.data_ptr(), spatial_shapes._version, tuple(query_levels), len_q)
        if self._validated_q != key:
            l0, l1 = query_levels
            assert 0 <= l0 < l1 <= spatial_shapes.shape[0]
            assert int((spatial_shapes[l0:l1, 0] * spatial_shapes[l0:l1, 1]).sum()) == len_q
            self._validated_q = key

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None, *, query_levels=None, projected_value=None,
                shared_reference=None):
        """query (N, Lq, C); reference_points (N, Lq, n_levels, n_points, 2) in [0,1] -- MVDeTr's 5-D
        form (ms_deform_attn.py:104-107) -- or (..., 4) boxes; input_flatten (N, sum H_l*W_l, C);
        input_spatial_shapes (n_levels, 2); input_level_start_index (n_levels,);
        input_padding_mask (N, sum H_l*W_l) True = padding.  Returns (N, Lq, C).

        Two keyword extensions for the query-sharded encoder (not in the reference): ``projected_value``
        (N, sum H_l*W_l, C) = project_value() of all tokens, used instead of projecting ``input_flatten`` here;
        ``query_levels=(l0, l1)`` promises that the Lq queries are exactly the tokens of levels l0..l1-1, which
        lets the fused kernel take the call (it is only a hint: results do not depend on it).
        ``shared_reference`` (N or 1, n_levels, Lq, 2): the caller's promise that the n_points reference points of every
        (query, level) coincide, with that point level-major -- saves the module its own test (a device sync)."""
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
        needs_grad = torch.is_grad_enabled() and (query.requires_grad or (value is not None and value.requires_grad)
                                                  or any(p.requires_grad for p in self.parameters()))
        if needs_grad and query.is_cuda and MSDA.backward_deterministic() and not (
                query.dtype == torch.float32 and MSDA.backward_deterministic_supported(N, Len_in, M, D, self.n_levels, Len_q, self.n_points)):
            # (the library would refuse at BACKWARD time with hipErrorNotSupported -- after the whole forward: say it here)
            raise RuntimeError("MSDeformAttn: the deterministic backward (MSDA.set_backward_deterministic / MVDETR_MSDA_BWD_DETERMINISTIC) "
                               f"serves fp32 deformable-encoder calls with 16-channel heads only; this call has d_model / n_heads = {D}, "
                               f"{self.n_levels} levels, Len_q = {Len_q}, Len_in = {Len_in}, dtype {query.dtype}")
        if (self.fused_inference and reference_points.shape[-1] == 2 and reference_points.dim() == 5
                and not needs_grad and query.is_cuda and query.dtype == torch.float32 and D in (16, 32)
                and (value is None or (value.is_cuda and value.dtype == torch.float32))
                and (pending is None or getattr(pending, "dtype", torch.float32) == torch.float32)
                and reference_points.is_cuda and reference_points.dtype == torch.float32
                and MSDA.fused_supported_dims(N, Len_in, M, D, self.n_levels, Len_q, self.n_points, query_levels)):
            # one GEMM for offsets + logits, rows permuted to the slice-interleaved layout (see _fused_projection)
            w, b = self._fused_projection()
            raw = F.linear(query, w, b)
            if pending is not None:
                value = pending()
            value = value.view(N, Len_in, M, D).contiguous()
            shared = shared_reference if shared_reference is not None else self._shared_reference(reference_points)
            if value.data_ptr() % 16 == 0 and raw.data_ptr() % 16 == 0:      # (the kernel's alignment contract)
                out = MSDA.ms_deform_attn_forward_fused(
                    value, input_spatial_shapes, input_level_start_index,
                    reference_points if shared is None else shared, None, None, query_levels=query_levels, raw=raw,
                    ref_level_major=shared is not None, raw_level_outer=self.raw_level_outer)
                return self.output_proj(out)
            # odd storage offsets: the reference arithmetic on the same raw values
            n_off = self.n_heads * self.n_levels * self.n_points * 2
            inv = torch.empty_like(self._fused_rows)
            inv[self._fused_rows] = torch.arange(inv.numel(), device=inv.device)
            plain = raw.index_select(-1, inv)
            offsets = plain[..., :n_off].reshape(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
            weights = plain[..., n_off:].reshape(N, Len_q, self.n_heads, self.n_levels * self.n_points)
            return self._reference_core(value, offsets, weights, reference_points, input_spatial_shapes,
                                        input_level_start_index, N, Len_q)
        if pending is not None:
            value = pending()
        if (needs_grad and self.fused_training and self.raw_level_outer and query_levels is None and self.n_points == 4
                and reference_points.dim() == 5 and reference_points.shape[-1] == 2
                and query.is_cuda and query.dtype == torch.float32 and value.is_cuda and value.dtype == torch.float32
                and reference_points.is_cuda and reference_points.dtype == torch.float32
                # (the fused function returns no gradient for the reference points: learned / refined points keep the
                # differentiable path below)
                and not reference_points.requires_grad
                and (shared_reference is None or not shared_reference.requires_grad)
                and MSDA.fused_train_supported(N, Len_in, M, D, self.n_levels, Len_q, self.n_points)
                and self._levels_equal(input_spatial_shapes)):
            shared = shared_reference if shared_reference is not None else self._shared_reference(reference_points)
            if shared is not None:
                # training: ONE GEMM for offsets + logits, then the fused pair -- neither sampling locations nor attention
                # weights are materialised, forward or backward (ms_deform_attn.py:100-114 + func.py:21-38 in two kernels
                # each way)
                w, b = self._fused_projection_train()
                raw = F.linear(query, w, b)
                value4 = value.view(N, Len_in, M, D).contiguous()
                shared = shared.contiguous()
                # the kernels' contract: fp32 everywhere (autocast may have made `raw` 16-bit), 16-byte aligned tensors,
                # 8-byte aligned reference points; anything else takes the unfused path below instead of raising
                if (raw.dtype == torch.float32 and shared.dtype == torch.float32 and value4.data_ptr() % 16 == 0
                        and raw.data_ptr() % 16 == 0 and shared.data_ptr() % 8 == 0):
                    out = MSDeformAttnFusedFunction.apply(value4, input_spatial_shapes, input_level_start_index,
                                                          shared, raw)
                    return self.output_proj(out)
        value = value.view(N, Len_in, M, D)
        offsets = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        weights = self.attention_weights(query).view(N, Len_q, self.n_heads, self.n_levels * self.n_points)
        return self._reference_core(value, offsets, weights, reference_points, input_spatial_shapes,
                                    input_level_start_index, N, Len_q)

    def _reference_core(self, value, offsets, weights, reference_points, input_spatial_shapes,
                        input_level_start_index, N, Len_q):
        """ms_deform_attn.py:100-117 from the raw offsets / logits on: softmax, sampling locations, the extension's
        differentiable function, output projection."""
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
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         locations.contiguous(), weights, self.im2col_step)
        return self.output_proj(out)
