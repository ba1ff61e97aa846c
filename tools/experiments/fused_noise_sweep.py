"""The fused training pair as the offsets spread: Wildtrack shape, raw offsets = bias grid + N(0, s px); HIP-event times."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mvdetr_amd.ops  # noqa
import MultiScaleDeformableAttention as MSDA
from helpers import fused_train_inputs
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
for s in (0.5, 1.0, 2.0, 3.0, 4.0):
    v, sh, lsi, ref, raw, _ = [x.cuda() for x in fused_train_inputs(7, 60, 180, 8, 16, 4, seed=0, noise_px=s)]
    go = torch.randn(1, 75600, 128, device="cuda")
    out, st = MSDA.ms_deform_attn_forward_fused_train(v, sh, lsi, ref, raw)
    f = t(lambda: MSDA.ms_deform_attn_forward_fused_train(v, sh, lsi, ref, raw))
    b = t(lambda: MSDA.ms_deform_attn_backward_fused(go, v, sh, lsi, ref, raw, st, out), 6)
    print(f"noise {s:4.1f} px  fused forward {f:8.1f} us  fused backward {b:8.1f} us", flush=True)
