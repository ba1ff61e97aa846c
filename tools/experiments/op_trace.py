#!/usr/bin/env python3
"""Phase timeline of one workgroup of msda_bwd_onepass (needs the -DMVDETR_BWD_TRACE build: make libmvdetr_ops_optrace.so,
MVDETR_OPS_LIB=.../libmvdetr_ops_optrace.so) and the backward's timing."""
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
print(f"backward {a.elapsed_time(b) * 100:.1f} us (incl. memset)")
lib = _lib.lib()
if hasattr(lib, "mvdetr_debug_onepass_trace"):
    buf = (ctypes.c_ulonglong * 4096)()
    lib.mvdetr_debug_onepass_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.mvdetr_debug_onepass_trace(buf, 4096)
    t = list(buf)
    t0 = min(x for x in t if x)
    for j in range(12):
        for w in range(4):
            r = t[j * 64 + w * 16: j * 64 + w * 16 + 16]
            if not r[0]:
                continue
            d = lambda a, b: (r[b] - r[a]) / 100 if r[a] and r[b] else float("nan")  # noqa: E731
            print(f"job {j} wave {w}: start {(r[0] - t0) / 100:7.2f}  sample+dma+loads +{d(0, 1):5.2f}  bounds+mass +{d(1, 2):5.2f}  pass1 +{d(2, 3):5.2f}  "
                  f"bar +{d(3, 4):5.2f}  flush +{d(4, 5):5.2f} | step0: build {d(6, 7):.2f} slots {d(7, 14):.2f} finalize {d(14, 15):.2f}"
                  f" | steps 1-3: " + " ".join(f"{d(6 + 2 * s, 8 + 2 * s):.2f}" for s in range(1, 3)))
