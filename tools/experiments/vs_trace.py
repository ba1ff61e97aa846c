#!/usr/bin/env python3
"""Phase timeline of one workgroup of msda_bwd_value_sort (needs a -DMVDETR_VS_TRACE build of msda_backward_sort.hip)."""
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
go = torch.randn(1, L * H * W, M * D, device="cuda")
for _ in range(3):
    MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)
torch.cuda.synchronize()
lib = _lib.lib()
buf = (ctypes.c_ulonglong * 256)()
lib.mvdetr_debug_vs_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.mvdetr_debug_vs_trace(buf, 256)
t = list(buf)
names = ["zero+sync", "A count", "sync", "scan", "B place", "gather", "sync+store", "flush"]
print(f"job start -> level 0: {(t[1] - t[0]) / 100:.2f} us (stage grad_out)")
for l in range(L):
    r = t[1 + l * 8: 10 + l * 8]
    print(f"level {l}: " + "  ".join(f"{n} {(r[i + 1] - r[i]) / 100:5.2f}" for i, n in enumerate(names[1:])) +
          (f"   | next zero+sync {(t[1 + (l + 1) * 8] - r[8]) / 100:5.2f}" if l + 1 < L else ""))
print(f"job total {(t[8 + (L - 1) * 8] - t[0]) / 100:.1f} us")
