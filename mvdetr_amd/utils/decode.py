"""BEV heat-map -> ground-plane detections.

Mirrors ``mvdet_decode`` (multiview_detector/utils/decode.py:80-93) and the per-frame post-processing of
the reference's test loop (multiview_detector/trainer.py:133-149): every cell of the world heat-map is a
candidate at (cell + predicted offset) * world_reduce, candidates above ``cls_thres`` go through the
distance NMS, survivors are written as ``frame x y`` rows.  Device-agnostic torch; runs where its inputs live.
"""
from __future__ import annotations

import torch

from .nms import nms


def mvdet_decode(scoremap, offset=None, reduce=4):
    """scoremap [B,1,H,W] (already a probability), offset [B,2,H,W] or None -> [B, H*W, 3] rows
    (x, y, score) in row-major cell order, x = (column + dx) * reduce, y = (row + dy) * reduce; without an
    offset map the cell centre (+0.5) is used (decode.py:80-93)."""
    B, C, H, W = scoremap.shape
    ys, xs = torch.meshgrid(torch.arange(H, device=scoremap.device), torch.arange(W, device=scoremap.device),
                            indexing="ij")
    xy = torch.stack([xs, ys], -1).reshape(1, H * W, 2).float().expand(B, -1, -1)
    if offset is not None:
        xy = xy + offset.permute(0, 2, 3, 1).reshape(B, H * W, 2)
    else:
        xy = xy + 0.5
    xy = xy * reduce
    scores = scoremap.permute(0, 2, 3, 1).reshape(B, H * W, C)[..., :1]
    return torch.cat([xy, scores], dim=2)


def detections_from_heatmap(world_heatmap, world_offset, frames, world_reduce=4, cls_thres=0.4, indexing="xy",
                            dist_thres=20, top_k=float("inf")):
    """The reference test loop's result rows for one batch (trainer.py:133-149): raw heat-map logits
    [B,1,H,W] + offsets [B,2,H,W] -> float tensor [n, 3] of (frame, x, y) rows, frames in batch order,
    detections of a frame in descending score order.  ``indexing='ij'`` swaps the coordinates like
    trainer.py:139-142."""
    xys = mvdet_decode(torch.sigmoid(world_heatmap.detach()), None if world_offset is None else world_offset.detach(),
                       reduce=world_reduce)
    positions, scores = xys[:, :, :2], xys[:, :, 2]
    if indexing != "xy":
        positions = positions[:, :, [1, 0]]
    rows = []
    for b in range(xys.shape[0]):
        sel = scores[b] > cls_thres
        pos, s = positions[b, sel], scores[b, sel]
        keep, count = nms(pos, s, dist_thres, top_k)
        kept = pos[keep[:count]]
        rows.append(torch.cat([torch.full((count, 1), float(frames[b]), device=kept.device), kept], dim=1))
    return torch.cat(rows, dim=0) if rows else torch.empty(0, 3)
