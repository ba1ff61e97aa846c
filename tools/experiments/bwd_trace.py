#!/usr/bin/env python3
"""Phase timeline of one workgroup of msda_bwd_value_win (needs a -DMVDETR_BWD_TRACE build of the library), the
backward's timing, and its parity against the generic atomic kernel."""
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
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)
torch.cuda.synchronize()
a, b = torch.cuda.Event(True), torch.cuda.Event(True)
a.record()
for _ in range(10):
    MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)
b.record(); torch.cuda.synchronize()
print(f"backward {a.elapsed_time(b) * 100:.1f} us")
gv64, _, _ = MSDA.ms_deform_attn_backward(value.double(), shapes, lsi, loc.double(), aw.double(), go.double(), 64)
print("grad_value max err vs fp64 kernel", (gv.double() - gv64).abs().max().item(), "max |gv|", gv64.abs().max().item())
lib = _lib.lib()
if hasattr(lib, "mvdetr_debug_bwd_trace"):
    buf = (ctypes.c_ulonglong * 2048)()
    lib.mvdetr_debug_bwd_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mvdetr_debug_bwd_trace(buf, 2048)
    t = list(buf)
    t0 = min(x for x in t if x)
    for j in range(8):
        for w in range(4):
            r = t[j * 64 + w * 16: j * 64 + w * 16 + 16]
            if not r[0]:
                continue
            d = lambda a, b: (r[b] - r[a]) / 100 if r[a] and r[b] else float("nan")  # noqa: E731
            print(f"job {j} wave {w}: start {(r[0] - t0) / 100:7.2f}  loads+bounds +{d(0, 1):5.2f}  mass +{d(1, 2):5.2f}  accumulate +{d(2, 3):5.2f}  "
                  f"bar +{d(3, 4):5.2f}  flush +{d(4, 5):5.2f}   steps 0-3 (taps | adds): " +
                  "  ".join(f"{d(6 + 2 * s, 7 + 2 * s):.2f}|{d(7 + 2 * s, 8 + 2 * s):.2f}" for s in range(3)))
