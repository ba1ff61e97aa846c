#!/usr/bin/env python3
"""Kernel-level timings of the hot path on one GPU (HIP events on torch's current stream).

    python tools/microbench.py [--config wildtrack|multiviewx|stress16] [--iters 50]

Prints one line per kernel: average us, algorithmic GB/s (SURVEY 8d byte counts), fraction of the
8 TB/s HBM peak.  Used while tuning; bench.py is the contract-level benchmark.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

from helpers import encoder_msda_inputs, fused_train_inputs, random_msda_inputs  # noqa: E402
from mvdetr_amd import geometry  # noqa: E402
import mvdetr_amd.ops  # noqa: E402,F401
from mvdetr_amd.ops import warp_perspective  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402

PEAK = 8.0e12


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


def report(name, us, nbytes):
    avg, med, mn = us
    print(f"{name:34s} avg {avg:9.1f} us  med {med:9.1f}  min {mn:9.1f}   {nbytes / avg / 1e3:8.1f} GB/s (alg)  "
          f"{nbytes / avg * 1e6 / PEAK * 100:5.1f}% of 8 TB/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="wildtrack")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--skip-bwd", action="store_true")
    ap.add_argument("--only", choices=["msda", "warp"], default=None)
    a = ap.parse_args()
    geom = geometry.GEOMETRIES[a.config]
    L = geom.num_cam
    H, W = geom.Rworld_shape[0] // 2, geom.Rworld_shape[1] // 2
    C = geom.feat_channels
    M, D, P, B = 8, C // 8, 4, a.batch
    S = L * H * W
    print(f"# {a.config}: L={L} token map {H}x{W} S=Lq={S} M={M} D={D} P={P} B={B}  device={torch.cuda.get_device_name(0)}")
    fwd_bytes = 4 * B * (S * M * D + 3 * S * M * L * P + S * M * D)
    bwd_bytes = 4 * B * (S * M * D + 2 * S * M * D + 6 * S * M * L * P)

    if a.only != "warp":
        bench_msda(a, L, H, W, M, D, P, B, S, fwd_bytes, bwd_bytes)
    if a.only != "msda":
        bench_warp(a, geom, L, C)


def bench_msda(a, L, H, W, M, D, P, B, S, fwd_bytes, bwd_bytes):
    for tag, maker in (("realistic", lambda: encoder_msda_inputs(L, H, W, M, D, P, B=B, seed=0)),
                       ("uniform", lambda: random_msda_inputs(B, [(H, W)] * L, M, D, S, P, seed=1, lo=0, hi=1))):
        value, shapes, lsi, loc, aw = [x.cuda() for x in maker()]
        for impl in ("gather", "tile"):
            MSDA.set_forward_impl(impl)
            fn = lambda: MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)  # noqa: E731
            fn()
            used = MSDA.last_forward_impl()
            if used == impl:
                report(f"msda_fwd[{tag}] impl={used}", time_us(fn, a.iters), fwd_bytes)
        MSDA.set_forward_impl("auto")
        fn = lambda: MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)  # noqa: E731
        report(f"msda_fwd[{tag}] auto (+probe)", time_us(fn, a.iters), fwd_bytes)
        if not a.skip_bwd:
            go = torch.randn(B, S, M * D, device="cuda")
            fn = lambda: MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)  # noqa: E731
            report(f"msda_bwd[{tag}] (+memset)", time_us(fn, max(5, a.iters // 5)), bwd_bytes)

    # fused entry (what the model calls in inference), all query levels and one rank's share (query-sharded)
    g = torch.Generator().manual_seed(0)
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(1, H * W, 1, 1, 2).repeat(1, L, L, P, 1).cuda()
    raw = torch.randn(B, S, L * M * P * 3, generator=g)                  # [offsets | logits], level-major
    n_off = L * M * P * 2
    # SURVEY 8d's locality-realistic input for the fused entry too: offsets = the module's initial bias grid
    # (ms_deform_attn.py:64-69: head m's ray, point p at (p + 1) px) + N(0, 1 px); logits N(0, 1)
    import math
    ang = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    dirs = torch.stack([ang.cos(), ang.sin()], -1)
    dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
    bias = dirs.view(1, M, 1, 2) * torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)      # [1, M, P, 2]
    raw[..., :n_off] += bias.expand(L, M, P, 2).reshape(-1)
    raw = raw.cuda()
    off, logit = raw[..., :n_off].unflatten(-1, (L, M, P, 2)), raw[..., n_off:].unflatten(-1, (L, M, P))
    value, shapes, lsi = [x.cuda() for x in encoder_msda_inputs(L, H, W, M, D, P, B=B, seed=0)[:3]]
    fn = lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, ref, off, logit, level_major=True)  # noqa: E731
    report("msda_fwd_fused[all levels]", time_us(fn, a.iters), fwd_bytes)
    for n_own in (1, 2) if B == 1 else ():          # (a rank's share is a slice of the query axis: dense only for one frame)
        q = n_own * H * W
        own_bytes = 4 * B * (S * M * D + 3 * q * M * L * P + q * M * D)
        o2, l2, r2 = off[:, :q], logit[:, :q], ref[:, :q]
        fn = lambda: MSDA.ms_deform_attn_forward_fused(value, shapes, lsi, r2, o2, l2, level_major=True,  # noqa: E731
                                                       query_levels=(0, n_own))
        report(f"msda_fwd_fused[{n_own} of {L} levels]", time_us(fn, a.iters), own_bytes)

    # fused TRAINING pair: raw [B, Lq, M*L*12] in the level-outer slice layout in, gradient of the same tensor out; the
    # byte counts are SURVEY 8d's for the unfused kernels it replaces (so the fractions compare like for like)
    if not a.skip_bwd and MSDA.fused_train_supported(B, S, M, D, L, S, P):
        del raw, off, logit
        value, shapes, lsi, ref_lm, raw_t, rows = [x.cuda() for x in fused_train_inputs(L, H, W, M, D, P, B=B, seed=0)]
        assert rows.tolist() == MSDA.slice_major_rows(M, L, P, D, level_outer=True)
        fn = lambda: MSDA.ms_deform_attn_forward_fused_train(value, shapes, lsi, ref_lm, raw_t)  # noqa: E731
        report("msda_fwd_fused_train (+stats)", time_us(fn, a.iters), fwd_bytes)
        out_t, stats_t = fn()
        go = torch.randn(B, S, M * D, device="cuda")
        fn = lambda: MSDA.ms_deform_attn_backward_fused(go, value, shapes, lsi, ref_lm, raw_t, stats_t, out_t)  # noqa: E731
        report("msda_bwd_fused (+memset)", time_us(fn, max(5, a.iters // 3)), bwd_bytes)


def bench_warp(a, geom, L, C):
    h, w = geom.Rimg_shape
    Hw, Ww = geom.Rworld_shape
    Ks, Rts = geometry.synthetic_rig(geom, seed=0)
    pm = geometry.build_proj_mats(geom, Ks, Rts)
    Mx = geometry.compose_frame_proj_mats(pm, torch.eye(3).repeat(1, L, 1, 1), geom.img_reduce).cuda()
    src = torch.randn(L, C, h, w, device="cuda")
    wbytes = 4 * L * C * (h * w + Hw * Ww)
    report("warp_fwd NCHW", time_us(lambda: warp_perspective(src, Mx, (Hw, Ww)), a.iters), wbytes)
    report("warp_fwd NHWC", time_us(lambda: warp_perspective(src, Mx, (Hw, Ww), channels_last_out=True), a.iters), wbytes)
    src_cl = src.contiguous(memory_format=torch.channels_last)
    report("warp_fwd NHWC <- NHWC source", time_us(lambda: warp_perspective(src_cl, Mx, (Hw, Ww), channels_last_out=True), a.iters), wbytes)
    if not a.skip_bwd:
        from mvdetr_amd.ops.warp import WarpPerspectiveFunction
        for tag, s_in, nhwc in (("NCHW", src, False), ("NHWC <- NHWC source", src_cl, True)):
            leaf = s_in.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
            out = warp_perspective(leaf, Mx, (Hw, Ww), channels_last_out=nhwc)
            go = torch.randn_like(out)
            fn = lambda: torch.autograd.grad(out, leaf, go, retain_graph=True)  # noqa: E731
            report(f"warp_bwd {tag} (autograd)", time_us(fn, max(5, a.iters // 3)), wbytes)
    if not a.skip_bwd:
        # the channel-last backward at the C ABI (no autograd / Python in the timed region)
        from mvdetr_amd.ops import warp as warp_mod
        go = torch.randn(L, Hw, Ww, C, device="cuda")
        gs = torch.empty(L, h, w, C, device="cuda")
        Mc = Mx.reshape(-1, 3, 3).float().contiguous()
        report("warp_bwd NHWC, C ABI entry", time_us(lambda: warp_mod._launch("backward", go, Mc, L, C, h, w, Hw, Ww, 3, gs),
                                                        max(10, a.iters)), wbytes)
    # for scale: a plain device copy of the same number of bytes
    x = torch.empty(wbytes // 8, device="cuda")
    y = torch.empty_like(x)
    report("torch copy (same bytes as warp)", time_us(lambda: y.copy_(x), a.iters), wbytes)


if __name__ == "__main__":
    main()
