#!/usr/bin/env python3
"""Fused forward at one configuration, one timing line per raw-tensor layout (slice-outer / level-outer runs); the job map
and the window shift are process-wide environment knobs (MVDETR_MSDA_JOBMAP=band|blocks, MVDETR_MSDA_WINDOW_SHIFT=0|1), so
run the script once per combination.  HIP events around every launch; parity of each variant against the gather kernel.

    python tools/experiments/fwd_variants.py [--config wildtrack] [--noise 1.0] [--iters 40] [--batch 1]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import encoder_msda_inputs  # noqa: E402
from mvdetr_amd import geometry  # noqa: E402
import mvdetr_amd.ops  # noqa: E402,F401
import MultiScaleDeformableAttention as MSDA  # noqa: E402


def time_us(fn, iters, warmup=8):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(iters))
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]


def inputs(config, noise, B):
    geom = geometry.GEOMETRIES[config]
    L = geom.num_cam
    H, W = geom.Rworld_shape[0] // 2, geom.Rworld_shape[1] // 2
    M, D, P = 8, geom.feat_channels // 8, 4
    S = L * H * W
    value, shapes, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(L, H, W, M, D, P, B=B, seed=0, noise_px=noise)]
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1).cuda()                 # [S,2]
    off = (loc - ref[None, :, None, None, None, :]) * torch.tensor([W, H], device="cuda", dtype=torch.float32)
    logit = aw.clamp_min(1e-30).log()
    plain = torch.cat([off.reshape(B, S, -1), logit.reshape(B, S, -1)], -1)
    ref_lm = ref.view(1, S, 1, 2).expand(1, S, L, 2).transpose(1, 2).contiguous()
    nbytes = 4 * B * (S * M * D + 3 * S * M * L * P + S * M * D)
    return dict(L=L, H=H, W=W, M=M, D=D, P=P, S=S, value=value, shapes=shapes, lsi=lsi, loc=loc, aw=aw, plain=plain,
                ref_lm=ref_lm, nbytes=nbytes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="wildtrack")
    ap.add_argument("--noise", type=float, nargs="+", default=[1.0])
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    tag = " ".join(f"{k}={os.environ.get(k, '-')}" for k in ("MVDETR_MSDA_JOBMAP", "MVDETR_MSDA_WINDOW_SHIFT", "MVDETR_OPS_LIB"))
    for noise in a.noise:
        d = inputs(a.config, noise, a.batch)
        MSDA.set_forward_impl("gather")
        want = MSDA.ms_deform_attn_forward(d["value"], d["shapes"], d["lsi"], d["loc"], d["aw"], 64)
        for level_outer in (False, True):
            rows = torch.tensor(MSDA.slice_major_rows(d["M"], d["L"], d["P"], d["D"], level_outer=level_outer), device="cuda")
            raw = d["plain"].index_select(-1, rows).contiguous()
            fn = lambda: MSDA.ms_deform_attn_forward_fused(d["value"], d["shapes"], d["lsi"], d["ref_lm"], None, None, raw=raw,  # noqa: E731
                                                           ref_level_major=True, raw_level_outer=level_outer)
            err = (fn() - want).abs().max().item()
            avg, med, mn = time_us(fn, a.iters)
            print(f"{a.config} B={a.batch} noise {noise:3.1f} raw={'level-outer' if level_outer else 'slice-outer'} [{tag}] "
                  f"avg {avg:7.1f} med {med:7.1f} min {mn:7.1f} us  {d['nbytes'] / med * 1e6 / 8e12 * 100:5.2f}% (median)  err {err:.1e}  "
                  f"{MSDA.last_forward_kernel()} {MSDA.last_forward_resources()}", flush=True)


if __name__ == "__main__":
    main()
