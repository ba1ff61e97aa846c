"""fp32 GEMMs of the encoder layer: F.linear (weight [N, K]) against addmm with the weight stored [K, N], with and
without TunableOp."""
import os, sys, torch
import torch.nn.functional as F
tune = "--tune" in sys.argv
if tune:
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
    if hasattr(tun, "write_file_on_exit"): tun.write_file_on_exit(False)
M = 75600
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for K, N in ((128, 128), (128, 672), (128, 512), (512, 128)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    wt = w.t().contiguous()
    y0 = F.linear(x, w, b); y1 = torch.addmm(b, x, wt)
    print(f"K={K:4d} N={N:4d}  linear {t(lambda: F.linear(x, w, b)):7.1f} us   addmm[K,N] {t(lambda: torch.addmm(b, x, wt)):7.1f} us   "
          f"max diff {(y0 - y1).abs().max().item():.2e}  flops-floor {2 * M * K * N / 157e12 * 1e6:.0f} us  bytes-floor {4 * (M * K + M * N) / 5e12 * 1e6:.0f} us", flush=True)
