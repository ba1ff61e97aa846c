#!/usr/bin/env python3
"""Phase timeline of one workgroup of msda_fwd_resident (needs a -DMVDETR_RF_TRACE build of the library, MVDETR_OPS_LIB)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import encoder_msda_inputs  # noqa: E402
import mvdetr_amd.ops  # noqa: E402,F401
from mvdetr_amd import _lib  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402
L, H, W, M, D, P = 7, 60, 180, 8, 16, 4
value, shapes, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(L, H, W, M, D, P, seed=0, noise_px=1.0)]
for _ in range(3):
    MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
torch.cuda.synchronize()
print(MSDA.last_forward_kernel())
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 2048)()
lib.mvdetr_debug_rf_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mvdetr_debug_rf_trace(buf, 2048)
t = list(buf)
t0 = min(x for x in t if x)
for j in range(12):
    for w in range(4):
        r = t[j * 64 + w * 16: j * 64 + w * 16 + 16]
        if not r[0]:
            continue
        d = lambda a, b: (r[b] - r[a]) / 100 if r[a] and r[b] else float("nan")  # noqa: E731
        print(f"job {j} wave {w}: start {(r[0] - t0) / 100:7.2f}  bar +{d(0, 1):5.2f}  dma issue +{d(1, 2):5.2f}  loads issue +{d(2, 3):5.2f}  "
              f"landed +{d(3, 4):5.2f}  taps +{d(4, 5):5.2f}  far +{d(5, 6):5.2f}")
