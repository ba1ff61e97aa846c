"""Multi-GPU execution of the fusion path: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" is RCCL on ROCm; "gloo" on CPU for the tests).

The reference has no distributed code at all (single process, one .cuda() device).
Two ways the path shards:

  * frames (batch) are independent -> replicas, NO data-path collective ("dp");
  * cameras are independent up to the shadow transformer: backbone, heads, warp and the stride-2
    token conv run per view; every rank then needs the tokens of ALL views as attention `value`
    -> exactly ONE all-gather of per-view world tokens per frame ("views").  The payload per view is
    h*w*C fp32 = 5.5 MB at Wildtrack size; over fully connected xGMI RCCL uses a direct exchange for
    that size.  The encoder is then either replicated (ViewShardedFrame(encoder="replicated")) or
  * query-sharded (SURVEY 8e option B / 8f row f3, the default): every MSDeformAttn query is independent and
    only `value` must be complete, so each rank keeps running ONLY its own cameras' tokens through the three
    encoder layers.  Per layer it projects its own tokens (value_proj), all-gathers the projected values
    (the same 5.5 MB per view -- this replaces the token all-gather above, which is no longer needed), and
    runs attention (fused kernel restricted to its query levels), norm and FFN on its own queries.  The
    per-camera 1x1 merge convolution (trans_world_feat.py:82,107-108) is a sum over cameras: each rank
    applies its cameras' weight columns and ONE all-reduce of the [C, h, w] partial sums finishes it.
    Collectives per frame: 3 all-gathers + 1 all-reduce; no rank ever touches another camera's queries.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default process
    group when WORLD_SIZE > 1 (nccl == RCCL when a GPU is present, gloo otherwise)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        # MVDETR_DIST_BACKEND=gloo: several ranks on ONE GPU for testing (RCCL wants a GPU per rank; gloo stages
        # CUDA tensors through the host)
        backend = os.environ.get("MVDETR_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def partition_views(num_views: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) view ranges per rank, sizes differing by at most one, larger shards
    first (7 views / 8 ranks -> seven ranks with one view, rank 7 idle; 16 / 8 -> two each)."""
    base, extra = divmod(num_views, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def all_gather_view_tokens(local_tokens: torch.Tensor, num_views: int, group=None, hw: int = None,
                           async_op: bool = False):
    """local_tokens [B, n_local*hw, C] (this rank's views, possibly none) -> [B, num_views*hw, C] on
    every rank, views in camera order (``async_op``: a zero-argument callable returning that tensor once the
    collective is done, with a ``length`` attribute = num_views*hw).  One all_gather_into_tensor; ragged shards are padded to the
    largest shard and the padding is dropped after the collective.  Inference only: the collective is
    not differentiable (the tokens are detached)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_tokens
    rank = dist.get_rank(group)
    parts = partition_views(num_views, world)
    B, _, C = local_tokens.shape
    n_max = max(e - s for s, e in parts)
    n_loc = parts[rank][1] - parts[rank][0]
    if hw is None:
        hw = local_tokens.shape[1] // n_loc if n_loc else None
        # every rank must agree on hw (tokens per view); ranks without a view learn it from the others via a
        # tiny all-reduce -- callers that know it pass `hw` and skip this host synchronisation
        hw_t = torch.tensor([hw or 0], device=local_tokens.device, dtype=torch.int64)
        if any(e == s for s, e in parts):
            dist.all_reduce(hw_t, op=dist.ReduceOp.MAX, group=group)
        hw = int(hw_t.item())
    with torch.no_grad():
        send = local_tokens.new_zeros(B, n_max * hw, C)
        if n_loc:
            send[:, :n_loc * hw] = local_tokens.detach()
        # concatenated-along-dim-0 form: accepted by both RCCL and gloo
        recv = local_tokens.new_empty(world * B, n_max * hw, C)
        work = dist.all_gather_into_tensor(recv, send.contiguous(), group=group, async_op=async_op)

    def assemble():
        if work is not None:
            work.wait()                    # the current stream waits for the collective; the host does not block
        r4 = recv.view(world, B, n_max * hw, C)
        pieces = [r4[r, :, :(e - s) * hw] for r, (s, e) in enumerate(parts) if e > s]
        return torch.cat(pieces, dim=1)

    if not async_op:
        return assemble()
    # the caller resolves it when it needs the tokens (MSDeformAttn does so after enqueueing its own GEMM, so the
    # collective -- on RCCL's stream -- overlaps that GEMM)
    assemble.length = num_views * hw
    return assemble


class QueryShardedFusion:
    """DeformTransWorldFeat.fuse() with the encoder's queries partitioned by camera over `world` ranks.

    Written as the per-rank steps (layer_value / layer_update / merge_partial / merge_finish) plus
    __call__, which strings them together with the collectives -- so a test can drive several emulated ranks
    in lock-step on one device and compare with the unsharded fuse()."""

    def __init__(self, world_feat, rank: int, world: int, group=None):
        self.wf, self.rank, self.world, self.group = world_feat, rank, world, group
        self.views = partition_views(world_feat.num_cam, world)[rank]

    def own_slice(self, h, w):
        s, e = self.views
        return slice(s * h * w, e * h * w)

    def layer_value(self, i, src_own):
        """This rank's share of layer i's `value`: value_proj of its own tokens, [B, n_own*h*w, C]."""
        return self.wf.encoder.layers[i].self_attn.project_value(src_own)

    def layer_update(self, i, src_own, value_all, h, w):
        """Encoder layer i on this rank's queries, given the projected values of ALL cameras."""
        s, e = self.views
        if e == s:
            if callable(value_all):
                value_all()                # an idle rank still completes the collective it took part in
            return src_own
        wf, own = self.wf, self.own_slice(h, w)
        B = src_own.shape[0]
        ref = wf.encoder.reference_points[own].unsqueeze(0).expand(B, -1, -1, -1, -1)
        shared = wf.encoder.shared_reference()
        if shared is not None:
            shared = shared[:, own].unsqueeze(0)                 # level-major [1, L, own queries, 2]
        return wf.encoder.layers[i](src_own, wf.level_pos(h, w)[:, own], ref, wf.spatial_shapes,
                                    wf.level_start_index, query_levels=(s, e), projected_value=value_all,
                                    shared_reference=shared)

    def merge_partial(self, src_own, B, h, w):
        """This rank's cameras' term of merge_linear's 1x1 convolution (no bias), [B, C, h, w]."""
        wf, (s, e), C = self.wf, self.views, self.wf.hidden_dim
        conv = wf.merge_linear[0]
        if e == s:
            return conv.weight.new_zeros(B, C, h, w)
        x = src_own.view(B, e - s, h, w, C).permute(0, 1, 4, 2, 3).reshape(B, (e - s) * C, h, w)
        return torch.nn.functional.conv2d(x, conv.weight[:, s * C:e * C])

    def merge_finish(self, partial_sum):
        wf = self.wf
        y = partial_sum + wf.merge_linear[0].bias.view(1, -1, 1, 1)
        return wf.upsample(wf.merge_linear[1](y))

    @torch.no_grad()
    def __call__(self, local_tokens, B, h, w):
        """local_tokens [B, n_own*h*w, C] -> merged BEV feature [B, C, H, W], identical on every rank (up to
        the summation order of the all-reduce)."""
        src = local_tokens
        for i in range(self.wf.encoder.num_layers):
            value = all_gather_view_tokens(self.layer_value(i, src), self.wf.num_cam, self.group, hw=h * w,
                                           async_op=True)          # resolved inside the attention module
            src = self.layer_update(i, src, value, h, w)
        part = self.merge_partial(src, B, h, w).contiguous()
        if self.world > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        return self.merge_finish(part)


class ViewShardedFrame:
    """Runs one frame with the cameras partitioned over the ranks of `group`.  encoder="sharded": the shadow
    transformer stays partitioned by camera (QueryShardedFusion); "replicated": one token all-gather, then
    every rank runs the whole encoder."""

    def __init__(self, model, group=None, encoder="sharded"):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.range = partition_views(model.num_cam, self.world)[self.rank]
        if encoder not in ("sharded", "replicated"):
            raise ValueError(f"encoder must be 'sharded' or 'replicated', got {encoder!r}")
        self.sharded = QueryShardedFusion(model.world_feat, self.rank, self.world, group) \
            if encoder == "sharded" else None

    @torch.no_grad()
    def __call__(self, imgs_local, M):
        """imgs_local [B, n_local, 3, H, W]: this rank's cameras only; M [B, N, 3, 3] for all cameras.
        Returns (world_heatmap, world_offset), identical on every rank."""
        m = self.model
        s, e = self.range
        B = M.shape[0]
        H, W = m.Rworld_shape
        wf = m.world_feat
        h, w = H // wf.stride, W // wf.stride
        if e > s:
            proj = m.frame_proj_mats(M).view(B, m.num_cam, 3, 3)[:, s:e].reshape(-1, 3, 3)
            feat = m.features(imgs_local)
            from .ops import warp_perspective
            world = warp_perspective(feat, proj.to(feat.device, non_blocking=True), (H, W), channels_last_out=True)
            local, h, w = wf.tokens(world.view(B, e - s, H, W, -1))
        else:
            dev = next(m.parameters()).device
            local = torch.zeros(B, 0, wf.hidden_dim, device=dev)
        if e == s:                                         # idle rank: learn the token grid from the model
            h, w = wf.pos_embedding.shape[-2:]
        if self.sharded is not None:
            fused = self.sharded(local, B, h, w)
        else:
            # (tokens per view passed explicitly: an idle rank -- 7 views over 8 GPUs -- would otherwise make every rank
            # agree on it through an extra all-reduce and a host read-back per frame)
            tokens = all_gather_view_tokens(local, m.num_cam, self.group, hw=h * w)
            fused = wf.fuse(tokens, B, h, w)
        return m.world_heatmap(fused), m.world_offset(fused)


def barrier_and_max(elapsed_s: float, device) -> float:
    """Max of a per-rank wall time over all ranks (the bench contract's timing rule)."""
    if not dist.is_initialized():
        return elapsed_s
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
