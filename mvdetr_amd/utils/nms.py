"""Greedy distance-based non-maximum suppression on ground-plane points.

Same contract as ``nms`` of the reference (multiview_detector/utils/nms.py:7-44): visit the ``top_k``
highest-scoring points in descending score order, keep a point unless an already kept point lies within
``dist_thres`` of it (Euclidean, ``<=`` suppresses), return ``(keep, count)`` with ``keep`` a LongTensor of
``len(scores)`` whose first ``count`` entries are the kept indices (the rest is zero).

The reference re-gathers and re-measures the surviving candidates after every kept point; here the
candidates are sorted once and a suppression mask is swept, measuring each kept point against all
candidates in one vector operation -- same decisions, one gather.
"""
from __future__ import annotations

import torch


def nms(points, scores, dist_thres=50 / 2.5, top_k=50):
    assert points.shape[0] == scores.shape[0], "make sure same points and scores have the same size"
    n = scores.shape[0]
    keep = torch.zeros(n, dtype=torch.long, device=scores.device)
    if points.numel() == 0:
        return keep, 0
    # ascending sort, then take the tail and walk it backwards: ties come out in the reference's order
    order = scores.sort(0)[1]
    k = n if top_k == float("inf") else min(int(top_k), n)
    order = order[n - k:].flip(0)
    cand = points[order]
    alive = torch.ones(k, dtype=torch.bool, device=scores.device)
    count, i = 0, 0
    while i < k:
        keep[count] = order[i]
        count += 1
        # a kept point suppresses every later candidate within dist_thres (and itself)
        far = torch.linalg.vector_norm(cand[i] - cand, dim=1) > dist_thres
        alive &= far
        alive[: i + 1] = False
        nxt = torch.nonzero(alive[i + 1:])
        if nxt.numel() == 0:
            break
        i = i + 1 + int(nxt[0])
    return keep, count
