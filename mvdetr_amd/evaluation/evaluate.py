"""CLEAR-MOD detection metrics on the ground plane.

Same contract as the reference's Python evaluation (used whenever the MATLAB engine is absent,
multiview_detector/evaluation/evaluate.py:21-33):
  * ``evaluateDetection_py(res_fpath, gt_fpath, dataset_name)``   pyeval/evaluateDetection.py:6-93
  * ``CLEAR_MOD_HUN(gt, det)``                                      pyeval/CLEAR_MOD_HUN.py:10-100
  * ``evaluate(res_fpath, gt_fpath, dataset)``                      evaluate.py:21-33
Both files hold ``frame x y`` rows (trainer.py:153-155).  Only the frames that appear in the RESULT file are
scored, renumbered 0.. in ascending order.  Per frame the ground-truth and detected points are matched by
a minimum-cost assignment (scipy.optimize.linear_sum_assignment, the reference's own solver) with pairs
farther than 20 grid cells (50 cm / 2.5 cm) priced out; a match counts when its distance is < 20.

Written on whole-frame arrays (one distance matrix per frame, no per-pair Python loops); the reference's
quirks that change numbers are kept and listed in ``CLEAR_MOD_HUN``.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

TD = 50 / 2.5                 # match distance threshold, grid cells
_FAR = 1e6                    # price of a forbidden pair (CLEAR_MOD_HUN.py:67-70)


def _rows(a):
    a = np.asarray(a, dtype=np.float64)
    return a.reshape(-1, a.shape[-1]) if a.size else a.reshape(0, 3)


def _frame_tables(gt_raw, det_raw):
    """[frame, x, y] rows -> [frame_ctr, index_in_frame, x, y] tables restricted to the frames present in
    ``det_raw`` (evaluateDetection.py:52-90)."""
    frames = np.unique(det_raw[:, 0])
    out = []
    for raw in (gt_raw, det_raw):
        ctr = np.searchsorted(frames, raw[:, 0])
        ctr_ok = np.minimum(ctr, len(frames) - 1)
        sel = frames[ctr_ok] == raw[:, 0]
        order = np.argsort(ctr_ok[sel], kind="stable")           # frame by frame, file order inside a frame
        f = ctr_ok[sel][order]
        first = np.searchsorted(f, f)                            # position of each frame's first row
        idx = np.arange(len(f)) - first
        out.append(np.column_stack([f, idx, raw[sel][order][:, 1], raw[sel][order][:, 2]]).astype(np.float64))
    return out


def CLEAR_MOD_HUN(gt, det):
    """gt, det: [frame, id, x, y] rows with frames numbered from 0.  Returns (recall, precision, MODA, MODP)
    in percent.  Behaviour kept from CLEAR_MOD_HUN.py:30-98:
      * the number of scored frames is max(gt frame) + 1 -- trailing frames without ground truth drop out,
        false positives included;
      * a pair at exactly the threshold distance is neither priced out (``> td``) nor matched (``< td``);
      * each metric is clamped below at 0; MODP sums 1 - d/td over matches in (frame, gt index) order."""
    gt, det = np.asarray(gt, dtype=np.float64), np.asarray(det, dtype=np.float64)
    F = int(gt[:, 0].max()) + 1
    matched = n_gt = n_det = 0
    terms = []
    for t in range(F):
        g, d = gt[gt[:, 0] == t][:, 2:4], det[det[:, 0] == t][:, 2:4]
        n_gt += len(g)
        n_det += len(d)
        if not len(g) or not len(d):
            continue
        diff = g[:, None, :] - d[None, :, :]
        dist = np.sqrt(diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1])
        cost = np.where(dist > TD, _FAR, dist)
        rows, cols = linear_sum_assignment(cost)
        ok = cost[rows, cols] < TD
        matched += int(ok.sum())
        by_gt = np.argsort(rows[ok], kind="stable")
        terms.extend((1 - dist[rows[ok], cols[ok]][by_gt] / TD).tolist())
    fp, miss = n_det - matched, n_gt - matched
    with np.errstate(divide="ignore", invalid="ignore"):
        total = 0.0
        for v in terms:
            total += v
        modp = np.float64(total) / np.float64(matched) * 100
        moda = (1 - (np.float64(miss) + fp) / np.float64(n_gt)) * 100
        recall = np.float64(matched) / np.float64(n_gt) * 100
        precision = np.float64(matched) / np.float64(fp + matched) * 100
    clamp = lambda v: v if v > 0 else 0                          # noqa: E731  (NaN -> 0 too)
    return clamp(recall), clamp(precision), clamp(moda), clamp(modp)


def evaluateDetection_py(res_fpath, gt_fpath, dataset_name=None):
    """Paths (or arrays) of ``frame x y`` rows -> (recall, precision, MODA, MODP).  An empty result file scores
    (0, 0, 0, 0) (evaluateDetection.py:61-63)."""
    gt_raw = _rows(np.loadtxt(gt_fpath) if isinstance(gt_fpath, (str, bytes)) or hasattr(gt_fpath, "__fspath__")
                   else gt_fpath)
    det_raw = _rows(np.loadtxt(res_fpath) if isinstance(res_fpath, (str, bytes)) or hasattr(res_fpath, "__fspath__")
                    else res_fpath)
    if det_raw.shape[0] == 0:
        return 0, 0, 0, 0
    gt_all, det_all = _frame_tables(gt_raw, det_raw)
    return CLEAR_MOD_HUN(gt_all, det_all)


def evaluate(res_fpath, gt_fpath, dataset="wildtrack"):
    """evaluate.py:21-33 without the MATLAB branch (no MATLAB engine in this environment; the reference
    falls back to exactly this Python path when the import fails)."""
    return evaluateDetection_py(res_fpath, gt_fpath, dataset)
