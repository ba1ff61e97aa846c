"""Does hipBLASLt's bias+ReLU epilogue (torch._addmm_activation) beat linear + relu for the FFN's first GEMM?"""
import torch, torch.nn.functional as F
x = torch.randn(75600, 128, device="cuda"); w = torch.randn(512, 128, device="cuda") * 0.05; b = torch.randn(512, device="cuda")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
a = F.relu(F.linear(x, w, b)); c = torch._addmm_activation(b, x, w.t(), use_gelu=False)
print("max diff", (a - c).abs().max().item())
print("linear+relu  %.1f us" % t(lambda: F.relu(F.linear(x, w, b))))
print("linear+relu_ %.1f us" % t(lambda: F.linear(x, w, b).relu_()))
print("addmm_act    %.1f us" % t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)))
print("linear only  %.1f us" % t(lambda: F.linear(x, w, b)))
