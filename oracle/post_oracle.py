"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain Python loops, small cases) of the reference's detection
post-processing and CLEAR-MOD metric, the checker for mvdetr_amd/utils and mvdetr_amd/evaluation
(SURVEY 8f row f4).  Nothing in the product imports this file.

Parity status: PINNED -- tests/test_postprocess.py checks every function here against tests/golden/post.npz,
which tests/golden/make_golden_post.py produced by running the reference's own code (nms.py, decode.py,
evaluateDetection.py + CLEAR_MOD_HUN.py), including the reference's demo pair with its known answer
(MODA 88.4454 / MODP 75.6048 / precision 93.5818 / recall 94.9580).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.optimize import linear_sum_assignment


def mvdet_decode(scoremap, offset, reduce):
    """multiview_detector/utils/decode.py:80-93 -- numpy, cell by cell."""
    B, _, H, W = scoremap.shape
    out = np.zeros((B, H * W, 3), dtype=np.float32)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                dx, dy = (offset[b, 0, y, x], offset[b, 1, y, x]) if offset is not None else (0.5, 0.5)
                out[b, y * W + x] = ((np.float32(x) + np.float32(dx)) * np.float32(reduce),
                                     (np.float32(y) + np.float32(dy)) * np.float32(reduce), scoremap[b, 0, y, x])
    return out


def nms(points, scores, dist_thres, top_k, order=None):
    """multiview_detector/utils/nms.py:7-44 -- greedy, on Python lists.  ``order``: candidate indices by
    ascending score (the caller supplies torch's sort so that ties break the same way)."""
    n = len(scores)
    keep = [0] * n
    if n == 0:
        return keep, 0
    if order is None:
        order = sorted(range(n), key=lambda i: scores[i])
    order = list(order)
    k = n if top_k == float("inf") else min(int(top_k), n)
    remaining = order[n - k:]
    count = 0
    while remaining:
        best = remaining.pop()
        keep[count] = best
        count += 1
        bx, by = np.float32(points[best][0]), np.float32(points[best][1])
        nxt = []
        for j in remaining:
            dx, dy = bx - np.float32(points[j][0]), by - np.float32(points[j][1])
            if np.sqrt(dx * dx + dy * dy) > np.float32(dist_thres):
                nxt.append(j)
        remaining = nxt
    return keep, count


def clear_mod(res_rows, gt_rows, td=50 / 2.5):
    """evaluateDetection.py:52-91 + CLEAR_MOD_HUN.py:30-98 on [frame, x, y] rows -> (recall, precision,
    MODA, MODP).  Only the frames of the result rows are scored; frames after the last one that has ground
    truth drop out."""
    res_rows, gt_rows = np.asarray(res_rows, dtype=np.float64), np.asarray(gt_rows, dtype=np.float64)
    if res_rows.size == 0:
        return 0, 0, 0, 0
    frames = sorted(set(res_rows[:, 0].tolist()))
    per_frame = []
    for f in frames:
        per_frame.append(([r[1:3] for r in gt_rows if r[0] == f], [r[1:3] for r in res_rows if r[0] == f]))
    last = max(i for i, (g, _) in enumerate(per_frame) if g)
    c = fp = m = g_total = 0
    modp_terms = []
    for g, d in per_frame[:last + 1]:
        matched = {}
        if g and d:
            cost = np.zeros((len(g), len(d)))
            for i, a in enumerate(g):
                for j, b in enumerate(d):
                    dist = math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)
                    cost[i, j] = 1e6 if dist > td else dist
            for i, j in zip(*linear_sum_assignment(cost)):
                if cost[i, j] < td:
                    matched[i] = cost[i, j]
        c += len(matched)
        fp += len(d) - len(matched)
        m += len(g) - len(matched)
        g_total += len(g)
        modp_terms += [1 - matched[i] / td for i in sorted(matched)]
    pos = lambda v: v if v > 0 else 0                       # noqa: E731
    modp = pos(sum(modp_terms) / c * 100) if c else 0
    return pos(c / g_total * 100), pos(c / (fp + c) * 100) if fp + c else 0, pos((1 - (m + fp) / g_total) * 100), modp
