"""How the forward kernels degrade as the learned offsets grow: Wildtrack shape, offsets = init bias grid + N(0, s px)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mvdetr_amd.ops  # noqa
import MultiScaleDeformableAttention as MSDA
from helpers import encoder_msda_inputs
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for s in (0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 9.0):
    v, sh, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(7, 60, 180, 8, 16, 4, B=1, seed=0, noise_px=s)]
    row = [f"noise {s:4.1f} px"]
    for impl in ("tile", "gather", "auto"):
        MSDA.set_forward_impl(impl)
        row.append(f"{impl} {t(lambda: MSDA.ms_deform_attn_forward(v, sh, lsi, loc, aw, 64)):8.1f} us")
    MSDA.set_forward_impl("auto")
    go = torch.randn(1, 75600, 128, device="cuda")
    row.append(f"bwd {t(lambda: MSDA.ms_deform_attn_backward(v, sh, lsi, loc, aw, go, 64), 5):8.1f} us")
    print("  ".join(row), flush=True)
