"""Forward at Wildtrack size: fp32 (tile / gather) against the 16-bit storage gather kernels."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import mvdetr_amd.ops  # noqa
import MultiScaleDeformableAttention as MSDA
from helpers import encoder_msda_inputs

v, s, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(7, 60, 180, M=8, D=16, P=4, noise_px=1.0)]

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for impl in ("auto", "gather"):
    MSDA.set_forward_impl(impl)
    print("fp32 impl", impl, f"{timeit(lambda: MSDA.ms_deform_attn_forward(v, s, lsi, loc, aw, 64)):.1f} us",
          MSDA.last_forward_kernel())
for dt in (torch.float16, torch.bfloat16):
    a = (v.to(dt), s, lsi, loc.to(dt), aw.to(dt))
    print(dt, f"{timeit(lambda: MSDA.ms_deform_attn_forward(*a, 64)):.1f} us", MSDA.last_forward_kernel())
