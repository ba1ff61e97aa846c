"""add_layernorm at Wildtrack size: time and agreement with torch."""
import sys, torch
sys.path.insert(0, ".")
from mvdetr_amd.ops import add_layer_norm
rows, C = 75600, 128
x = torch.randn(1, rows, C, device="cuda"); r = torch.randn(1, rows, C, device="cuda")
ln = torch.nn.LayerNorm(C).cuda().eval()
with torch.no_grad():
    ln.weight.normal_(); ln.bias.normal_()
    ref = ln(x + r)
    out = add_layer_norm(x, r, ln)
    print("max err", (out - ref).abs().max().item())
    for _ in range(5): add_layer_norm(x, r, ln)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(50): add_layer_norm(x, r, ln)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print(f"add_layernorm {us:.1f} us  {3 * rows * C * 4 / us / 1e6:.0f} GB/s")
