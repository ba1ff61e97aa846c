#!/usr/bin/env python3
"""Forward routes of the warp at one configuration, HIP events (us): NCHW->NCHW, transpose + channel-last -> NCHW, ..."""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from mvdetr_amd import geometry
from mvdetr_amd.ops import warp_perspective
from mvdetr_amd.ops import warp as warp_mod
cfg = sys.argv[1] if len(sys.argv) > 1 else "wildtrack"
geom = geometry.GEOMETRIES[cfg]
L, C = geom.num_cam, geom.feat_channels
h, w = geom.Rimg_shape; H, W = geom.Rworld_shape
Ks, Rts = geometry.synthetic_rig(geom, seed=0)
pm = geometry.build_proj_mats(geom, Ks, Rts)
M = geometry.compose_frame_proj_mats(pm, torch.eye(3).repeat(1, L, 1, 1), geom.img_reduce).cuda().float()
src = torch.randn(L, C, h, w, device="cuda"); src_cl = src.contiguous(memory_format=torch.channels_last)
nbytes = 4 * L * C * (h * w + H * W)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) * 1e3 / n
for name, fn in (("NCHW -> NCHW", lambda: warp_perspective(src, M, (H, W))),
                 ("NHWC -> NCHW", lambda: warp_perspective(src_cl, M, (H, W))),
                 ("NCHW -> (transpose) -> NHWC -> NCHW", lambda: warp_perspective(src.contiguous(memory_format=torch.channels_last), M, (H, W))),
                 ("NHWC -> NHWC", lambda: warp_perspective(src_cl, M, (H, W), channels_last_out=True))):
    us = t(fn); print(f"{cfg} {name:40s} {us:7.1f} us  {nbytes / us / 1e6 / 8 * 100:5.1f}%  [{warp_mod.last_kernel()}]", flush=True)
