#!/usr/bin/env python3
"""Phase timeline of one workgroup of msda_fwd_group (needs a -DMVDETR_GROUP_TRACE build of the library)."""
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
buf = (ctypes.c_ulonglong * 2048)()
lib.mvdetr_debug_group_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mvdetr_debug_group_trace(buf, 2048)
t = list(buf)
t0 = min(x for x in t if x)
for l in range(L):
    print(f"level {l}")
    for w in range(4):
        r = t[w * 128 + l * 16: w * 128 + l * 16 + 16]
        if not r[0]:
            continue
        d = lambda a, b: (r[b] - r[a]) / 100 if r[a] and r[b] else float("nan")  # noqa: E731
        print(f"  wave {w}: arrive {(r[0] - t0) / 100:7.2f}  bar1 +{d(0, 1):5.2f}  issue +{d(1, 2):5.2f}  landed +{d(2, 3):5.2f}  written +{d(3, 4):5.2f}  "
              f"bar2 +{d(4, 5):5.2f}  cams " + " ".join(f"{d(5 + c, 6 + c):5.2f}" for c in range(7)))
