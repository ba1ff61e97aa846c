"""add_layernorm at the encoder's size (75,600 x 128): HIP-event time of the fused op, with and without the second output."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvdetr_amd.ops.add_layernorm import add_layer_norm
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
x = torch.randn(1, 75600, 128, device="cuda"); r = torch.randn_like(x)
norm = torch.nn.LayerNorm(128).cuda().eval()
with torch.no_grad():
    us = t(lambda: add_layer_norm(x, r, norm))
print(f"add_layer_norm 75600 x 128: {us:.1f} us = {3 * x.numel() * 4 / us / 1e3:.0f} GB/s")
