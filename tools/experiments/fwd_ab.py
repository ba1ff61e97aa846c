#!/usr/bin/env python3
"""A/B timing of the MSDA forward kernels at one configuration (HIP events), plus a parity check of what ran against
the gather kernel.  Run once per kernel family: MVDETR_MSDA_QUAD=0|1 (read once per process).

    python tools/experiments/fwd_ab.py [--config wildtrack] [--noise 1.0] [--iters 30]
"""
import argparse
import math
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


def time_us(fn, iters, warmup=5):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="wildtrack")
    ap.add_argument("--noise", type=float, nargs="+", default=[0.0, 1.0, 2.0])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    geom = geometry.GEOMETRIES[a.config]
    L = geom.num_cam
    H, W = geom.Rworld_shape[0] // 2, geom.Rworld_shape[1] // 2
    M, D, P, B = 8, geom.feat_channels // 8, 4, a.batch
    S = L * H * W
    nbytes = 4 * B * (S * M * D + 3 * S * M * L * P + S * M * D)
    tag = f"QUAD={os.environ.get('MVDETR_MSDA_QUAD', '1')} GROUP={os.environ.get('MVDETR_MSDA_GROUP', '1')}"
    print(f"# {a.config} L={L} {H}x{W} S={S} D={D} B={B} {tag}")
    for noise in a.noise:
        value, shapes, lsi, loc, aw = [x.cuda() for x in encoder_msda_inputs(L, H, W, M, D, P, B=B, seed=0, noise_px=noise)]
        MSDA.set_forward_impl("gather")
        ref_out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
        MSDA.set_forward_impl("tile")
        fn = lambda: MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)  # noqa: E731
        err = (fn() - ref_out).abs().max().item()
        avg, med, mn = time_us(fn, a.iters)
        print(f"unfused  noise {noise:3.1f}px  avg {avg:7.1f} med {med:7.1f} min {mn:7.1f} us  {nbytes / avg * 1e6 / 8e12 * 100:5.1f}%  err_vs_gather {err:.2e}", flush=True)
        # fused entry: raw offsets (pixels) / logits such that ref + off/size == loc and softmax(logit) == aw
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1).cuda()                 # [S,2]
        off = (loc - ref[None, :, None, None, None, :]) * torch.tensor([W, H], device="cuda", dtype=torch.float32)
        logit = aw.clamp_min(1e-30).log()
        # level-major column blocks of one [B,S,672]-like GEMM output
        raw = torch.cat([off.permute(0, 1, 3, 2, 4, 5).reshape(B, S, -1), logit.permute(0, 1, 3, 2, 4).reshape(B, S, -1)], -1).contiguous()
        n_off = L * M * P * 2
        o_lm, l_lm = raw[..., :n_off].unflatten(-1, (L, M, P, 2)), raw[..., n_off:].unflatten(-1, (L, M, P))
        ref4 = ref.view(1, S, 1, 2).expand(1, S, L, 2).contiguous()
        fn = lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, ref4, o_lm, l_lm, level_major=True)  # noqa: E731
        err = (fn() - ref_out).abs().max().item()
        avg, med, mn = time_us(fn, a.iters)
        print(f"fused    noise {noise:3.1f}px  avg {avg:7.1f} med {med:7.1f} min {mn:7.1f} us  {nbytes / avg * 1e6 / 8e12 * 100:5.1f}%  err_vs_gather {err:.2e}", flush=True)
        # slice-interleaved raw tensor + level-major reference points (what the module feeds the kernel)
        rows = torch.tensor(MSDA.slice_major_rows(M, L, P, D), device="cuda")
        plain = torch.cat([off.reshape(B, S, -1), logit.reshape(B, S, -1)], -1)
        raw_s = plain.index_select(-1, rows).contiguous()
        ref_lm = ref4.transpose(1, 2).contiguous()
        fn = lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, ref_lm, None, None, raw=raw_s, ref_level_major=True)  # noqa: E731
        err = (fn() - ref_out).abs().max().item()
        avg, med, mn = time_us(fn, a.iters)
        print(f"fused-sl noise {noise:3.1f}px  avg {avg:7.1f} med {med:7.1f} min {mn:7.1f} us  {nbytes / avg * 1e6 / 8e12 * 100:5.1f}%  err_vs_gather {err:.2e}", flush=True)
        fn = lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, ref4, None, None, raw=raw_s)  # noqa: E731
        err = (fn() - ref_out).abs().max().item()
        avg, med, mn = time_us(fn, a.iters)
        print(f"fused-sl(ref q-major) {noise:3.1f}px  avg {avg:7.1f} med {med:7.1f} min {mn:7.1f} us  {nbytes / avg * 1e6 / 8e12 * 100:5.1f}%  err_vs_gather {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
