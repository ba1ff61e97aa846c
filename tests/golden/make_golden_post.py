#!/usr/bin/env python3
"""Golden vectors for the detection post-processing / metric row, made by RUNNING the
reference's own Python code (imported from /root/reference -- build container only; nothing is copied
except the two demo DATA files the reference ships for its evaluation self-test).

    python tests/golden/make_golden_post.py

Writes:
  post.npz        nms_* : points / scores / dist_thres / top_k -> keep, count   (utils/nms.py:7-44)
                  dec_* : scoremap / offset / reduce -> rows                     (utils/decode.py:80-93)
                  ev_*  : result rows / ground-truth rows -> recall, precision, MODA, MODP
                          (evaluation/pyeval/evaluateDetection.py:6-93, CLEAR_MOD_HUN.py:10-100)
                  demo  : the four numbers of the reference's demo pair (evaluate.py:36-52)
  gt-demo.txt, test-demo.txt   the demo pair itself (data files of the reference's evaluation folder)
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)

from multiview_detector.utils.nms import nms  # noqa: E402
from multiview_detector.utils.decode import mvdet_decode  # noqa: E402
from multiview_detector.evaluation.pyeval.evaluateDetection import evaluateDetection_py  # noqa: E402

EVAL = os.path.join(REF, "multiview_detector", "evaluation")


def main():
    out = {}
    g = torch.Generator().manual_seed(2024)
    cases = 0
    for trial in range(24):
        n = [0, 1, 2, 5, 17, 60, 150, 400][trial % 8]
        pts = torch.rand(n, 2, generator=g) * torch.tensor([1440.0, 480.0]) / (1 if trial % 2 else 8)
        if trial % 3 == 0:
            pts = pts.round()                                   # exact ties in distance
        sc = torch.rand(n, generator=g)
        if trial % 4 == 1:
            sc = (sc * 6).round() / 6                           # ties in score
        for thres, topk in ((20, np.inf), (50 / 2.5, 50), (7.5, 9)):
            keep, count = nms(pts, sc, thres, topk)
            out[f"nms_{cases}_points"], out[f"nms_{cases}_scores"] = pts.numpy(), sc.numpy()
            out[f"nms_{cases}_args"] = np.array([thres, topk], dtype=np.float64)
            out[f"nms_{cases}_keep"], out[f"nms_{cases}_count"] = keep.numpy(), np.array(count)
            cases += 1
    out["nms_cases"] = np.array(cases)

    for i, (B, H, W, red, with_off) in enumerate([(1, 6, 9, 4, True), (2, 5, 7, 2, False), (3, 1, 4, 4, True)]):
        hm = torch.rand(B, 1, H, W, generator=g)
        off = torch.randn(B, 2, H, W, generator=g) if with_off else None
        out[f"dec_{i}_scoremap"] = hm.numpy()
        if with_off:
            out[f"dec_{i}_offset"] = off.numpy()
        out[f"dec_{i}_reduce"] = np.array(red)
        out[f"dec_{i}_rows"] = mvdet_decode(hm, off, red).numpy()
    out["dec_cases"] = np.array(3)

    gt = np.loadtxt(os.path.join(EVAL, "gt-demo.txt"))
    det = np.loadtxt(os.path.join(EVAL, "test-demo.txt"))
    out["demo"] = np.array(evaluateDetection_py(os.path.join(EVAL, "test-demo.txt"),
                                                os.path.join(EVAL, "gt-demo.txt"), "Wildtrack"), dtype=np.float64)
    rng = np.random.default_rng(5)
    with tempfile.TemporaryDirectory() as d:
        for i in range(8):
            frames = np.unique(det[:, 0])
            frames = frames[rng.random(len(frames)) < 0.25]                       # keep the fixtures small
            dsel = det[np.isin(det[:, 0], frames) & (rng.random(len(det)) < rng.uniform(0.3, 1.0))]
            jitter = rng.integers(-14, 15, (len(dsel), 2)) if i % 2 else np.zeros((len(dsel), 2))
            dsel = dsel + np.concatenate([np.zeros((len(dsel), 1)), jitter], 1)
            if i % 3 == 0:
                dsel = dsel[rng.permutation(len(dsel))]                           # rows out of frame order
            gsel = gt[np.isin(gt[:, 0], frames) & (rng.random(len(gt)) < rng.uniform(0.6, 1.0))]
            if i == 5:                                                            # last scored frames lose their GT
                gsel = gsel[gsel[:, 0] < np.sort(frames)[-2]]
            if i == 6:                                                            # a pair at exactly td = 20
                f0 = dsel[0, 0]
                g0 = gsel[gsel[:, 0] == f0][0]
                dsel = np.concatenate([dsel, [[f0, g0[1] + 12, g0[2] + 16]]], 0)
            np.savetxt(os.path.join(d, "r.txt"), dsel, "%d")
            np.savetxt(os.path.join(d, "g.txt"), gsel, "%d")
            out[f"ev_{i}_res"], out[f"ev_{i}_gt"] = dsel.astype(np.int32), gsel.astype(np.int32)
            out[f"ev_{i}_metrics"] = np.array(evaluateDetection_py(os.path.join(d, "r.txt"), os.path.join(d, "g.txt"), "x"),
                                              dtype=np.float64)
    out["ev_cases"] = np.array(8)
    np.savez_compressed(os.path.join(HERE, "post.npz"), **out)
    for name in ("gt-demo.txt", "test-demo.txt"):
        shutil.copyfile(os.path.join(EVAL, name), os.path.join(HERE, name))
        os.chmod(os.path.join(HERE, name), 0o644)
    print("post.npz:", os.path.getsize(os.path.join(HERE, "post.npz")), "bytes; demo metrics", out["demo"])


if __name__ == "__main__":
    main()
