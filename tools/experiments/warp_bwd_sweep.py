#!/usr/bin/env python3
"""Kernel-only timing of the channel-last warp backward (C ABI, HIP events) over the MVDETR_WARP_BWD_HEAVY threshold."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from mvdetr_amd import geometry  # noqa: E402
from mvdetr_amd.ops import warp as warp_mod  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "wildtrack"
geom = geometry.GEOMETRIES[cfg]
L, C = geom.num_cam, geom.feat_channels
h, w = geom.Rimg_shape
H, W = geom.Rworld_shape
Ks, Rts = geometry.synthetic_rig(geom, seed=0)
pm = geometry.build_proj_mats(geom, Ks, Rts)
M = geometry.compose_frame_proj_mats(pm, torch.eye(3).repeat(1, L, 1, 1), geom.img_reduce).cuda().float().contiguous()
go = torch.randn(L, H, W, C, device="cuda")
gs = torch.empty(L, h, w, C, device="cuda")
nbytes = 4 * L * C * (h * w + H * W)


def run():
    warp_mod._launch("backward", go, M, L, C, h, w, H, W, 3, gs)


for heavy in sys.argv[2:] or ["128"]:
    os.environ["MVDETR_WARP_BWD_HEAVY"] = heavy
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 50
    print(f"{cfg} heavy_above={heavy:>6s}: {us:7.1f} us  {nbytes / us / 1e6 / 8 * 100:5.1f}% of 8 TB/s", flush=True)
