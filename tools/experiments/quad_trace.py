#!/usr/bin/env python3
"""Phase timeline of one workgroup of msda_fwd_quad (needs a -DMVDETR_QUAD_TRACE build of the library)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import encoder_msda_inputs  # noqa: E402
import mvdetr_amd.ops  # noqa: E402,F401
from mvdetr_amd import _lib  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402

L, H, W, M, D, P = 7, 60, 180, 8, 16, 4
value, shapes, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(L, H, W, M, D, P, seed=0, noise_px=1.0)]
MSDA.set_forward_impl("tile")
for _ in range(5):
    out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
torch.cuda.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 1024)()
lib.mvdetr_debug_quad_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.mvdetr_debug_quad_trace(buf, 1024)
t = list(buf)
t0 = min(x for x in t if x)
for l in range(L):
    print(f"level {l}")
    for w in range(12):
        row = t[w * 80 + l * 11: w * 80 + l * 11 + 11]
        if not row[0]:
            continue
        print(f"  wave {w:2d}: arrive {(row[0] - t0) / 100:7.2f}  bar1 +{(row[1] - row[0]) / 100:5.2f}  copy +{(row[2] - row[1]) / 100:5.2f}  bar2 +{(row[3] - row[2]) / 100:5.2f}  cams " + " ".join(f"{(row[4 + c] - row[3 + c]) / 100:5.2f}" for c in range(7)))
