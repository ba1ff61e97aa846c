#!/usr/bin/env python3
"""Randomised parity run on the GPU box: encoder-shaped MSDA calls of random geometry through every dispatch route
(public forward, backward, fused forward in the plain and slice-interleaved layouts, the fused training pair, the host path)
against the oracle,
for a wall-clock budget.  The seeded 28-case sweep of tests/test_msda_gpu.py is the regression version of this.

    python tools/fuzz_parity.py --minutes 8 [--seed 7]
"""
import argparse
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import encoder_msda_inputs  # noqa: E402
from oracle import c_oracle, torch_oracle  # noqa: E402  (the checker)
import mvdetr_amd.ops  # noqa: E402,F401
import MultiScaleDeformableAttention as MSDA  # noqa: E402

TOL = 1e-4


def one_case(rnd, i):
    L = rnd.choice([6, 7, 7, 7, 6, 3, 5, 8, 12, 16])
    D = rnd.choice([16, 16, 16, 32])
    M = rnd.choice([2, 4, 8]) if D == 16 else rnd.choice([1, 2, 4])
    H, W = rnd.randint(1, 64), rnd.randint(1, 190)
    while L * H * W * M * D > 6_000_000:                     # keep the oracle's share of the time small
        H, W = max(1, H // 2), max(1, W * 2 // 3)
    B = rnd.choice([1, 1, 2, 3])
    noise = rnd.choice([0.0, 0.5, 1.0, 2.0, 5.0, 12.0])
    tag = f"#{i} L={L} D={D} M={M} {H}x{W} B={B} noise={noise}"
    value, shapes, lsi, loc, aw = encoder_msda_inputs(L, H, W, M=M, D=D, B=B, seed=1000 + i, noise_px=noise)
    d = [x.cuda() for x in (value, shapes, lsi, loc, aw)]
    want = c_oracle.msda_forward(value.double(), shapes, lsi, loc.double(), aw.double())
    got = MSDA.ms_deform_attn_forward(*d, 64).cpu().double()
    bad = []
    if (got - want).abs().max().item() >= TOL:
        bad.append(("forward", (got - want).abs().max().item(), MSDA.last_forward_kernel()))
    host = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64).double()
    if (host - want).abs().max().item() >= TOL:
        bad.append(("host forward", (host - want).abs().max().item(), ""))
    go = torch.randn(B, loc.shape[1], M * D, generator=torch.Generator().manual_seed(i))
    ref = c_oracle.msda_backward(value.double(), shapes, lsi, loc.double(), aw.double(), go.double())
    grads = [x.cpu().double() for x in MSDA.ms_deform_attn_backward(*d, go.cuda(), 64)]
    # grad_loc is discontinuous where a tap sits on a texel centre: compare away from those
    px = loc * torch.tensor([W, H], dtype=torch.float32) - 0.5
    frac = px - px.floor()
    smooth = ((frac > 2e-3) & (frac < 1 - 2e-3)).all(-1).double()
    # (grad_aw is a bilinear blend of four <grad_out, value> dots of magnitude ~4 (up to ~20) that may cancel; the kernel's fp32
    # pixel position -- loc * W - 0.5, as in the reference, cuh:285 -- is off by up to 1e-5 px, i.e. 1e-5 x the corner differences:
    # the error is measured against that scale.  Round 4's soak found 2.5e-4 against a scale of 1 on one element whose dots were
    # +19.9 / -18.9 -- fp32 arithmetic, not the kernel: tools/experiments notes in DESIGN 2.)
    for a, b, name, scale in zip(grads, ref, ("grad_value", "grad_loc", "grad_aw"), (1.0, float(max(W, H)), 4.0)):
        err = (a - b).abs() / (scale + b.abs())
        if name == "grad_loc":
            err = err * smooth[..., None]
        if err.max().item() >= 2e-4:
            bad.append((name, err.max().item(), ""))
    # the deterministic mode (ABI 13; deformable-encoder calls of 16-channel heads): bit-identical over two runs, same bars
    if D == 16:
        MSDA.set_backward_deterministic(True)
        try:
            det = [[x.clone() for x in MSDA.ms_deform_attn_backward(*d, go.cuda(), 64)] for _ in range(2)]
        finally:
            MSDA.set_backward_deterministic(False)
        if not all(torch.equal(a, b) for a, b in zip(*det)):
            bad.append(("deterministic backward differs between two runs", 0.0, ""))
        for a, b, name, scale in zip(det[0], ref, ("det grad_value", "det grad_loc", "det grad_aw"), (1.0, float(max(W, H)), 4.0)):
            err = (a.cpu().double() - b).abs() / (scale + b.abs())
            if name == "det grad_loc":
                err = err * smooth[..., None]
            if err.max().item() >= 2e-4:
                bad.append((name, err.max().item(), ""))
    if MSDA.fused_supported(d[0], L, value.shape[1], 4):
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        r3 = torch.stack([xs / W, ys / H], -1).reshape(1, H * W, 1, 2).repeat(B, L, L, 1)
        off = (loc - r3[:, :, None, :, None, :]) * torch.tensor([W, H], dtype=torch.float32)
        logit = torch.log(aw.clamp_min(1e-30))
        fused = MSDA.ms_deform_attn_forward_fused(d[0], d[1], d[2], r3.cuda(), off.cuda(), logit.cuda()).cpu().double()
        # (the fused entry recomputes locations from ref + off / size: positions differ by an ulp from `loc`)
        if (fused - want).abs().max().item() >= TOL:
            bad.append(("fused", (fused - want).abs().max().item(), MSDA.last_forward_kernel()))
        if D in (16, 32) and M % (32 // D) == 0:
            rows = torch.tensor(MSDA.slice_major_rows(M, L, 4, D))
            raw = torch.cat([off.reshape(B, loc.shape[1], -1), logit.reshape(B, loc.shape[1], -1)], -1).index_select(-1, rows)
            fs = MSDA.ms_deform_attn_forward_fused(d[0], d[1], d[2], r3.cuda(), None, None, raw=raw.contiguous().cuda()).cpu().double()
            if (fs - want).abs().max().item() >= TOL:
                bad.append(("fused slice layout", (fs - want).abs().max().item(), MSDA.last_forward_kernel()))
    # the fused TRAINING pair (6 / 7 equal levels of 16-channel heads): raw offsets / logits in, their gradient out, against the
    # C oracle's backward chained through loc = ref + off / (W, H) and the softmax by hand in fp64 (tests/test_fused_train_gpu.py)
    S = value.shape[1]
    if MSDA.fused_train_supported(B, S, M, D, L, S, 4):
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        cells = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1)                       # [S, 2]
        wh = torch.tensor([W, H], dtype=torch.float64)
        off_t = ((loc.double() - cells.double()[None, :, None, None, None, :]) * wh).float()         # [B, S, M, L, P, 2] pixels
        g = torch.Generator().manual_seed(7000 + i)
        logit_t = torch.randn(B, S, M, L, 4, generator=g) * 1.5
        rows = torch.tensor(MSDA.slice_major_rows(M, L, 4, D, level_outer=True))
        raw_t = torch.cat([off_t.reshape(B, S, -1), logit_t.reshape(B, S, -1)], -1).index_select(-1, rows).contiguous()
        ref_lm = cells[None, None].expand(1, L, S, 2).contiguous()
        out_t, stats_t = MSDA.ms_deform_attn_forward_fused_train(d[0], d[1], d[2], ref_lm.cuda(), raw_t.cuda())
        loc64 = cells.double()[None, :, None, None, None, :] + off_t.double() / wh
        aw64 = torch.softmax(logit_t.double().flatten(-2), -1).view(B, S, M, L, 4)
        want_t = c_oracle.msda_forward(value.double(), shapes, lsi, loc64.contiguous(), aw64.contiguous())
        if (out_t.cpu().double() - want_t).abs().max().item() >= TOL:
            bad.append(("fused train forward", (out_t.cpu().double() - want_t).abs().max().item(), MSDA.last_forward_kernel()))
        gv_r, gl_r, ga_r = c_oracle.msda_backward(value.double(), shapes, lsi, loc64.contiguous(), aw64.contiguous(), go.double())
        goff_r = gl_r / wh
        glogit_r = aw64 * (ga_r - (aw64 * ga_r).sum((-1, -2), keepdim=True))
        gv_t, graw_t = MSDA.ms_deform_attn_backward_fused(go.cuda(), d[0], d[1], d[2], ref_lm.cuda(), raw_t.cuda(), stats_t, out_t)
        inv = torch.empty_like(rows)
        inv[rows] = torch.arange(rows.numel())
        gplain = graw_t.cpu().double().index_select(-1, inv)
        n_off = M * L * 4 * 2
        px_t = loc64 * wh - 0.5
        smooth_t = ((px_t - px_t.round()).abs().amin(-1) > 1e-3).double()
        for a_, b_, name in ((gv_t.cpu().double(), gv_r, "fused grad_value"),
                             (gplain[..., :n_off].reshape(goff_r.shape) * smooth_t[..., None], goff_r * smooth_t[..., None], "fused grad_offsets"),
                             (gplain[..., n_off:].reshape(glogit_r.shape), glogit_r, "fused grad_logits")):
            err = ((a_ - b_).abs() / (1.0 + b_.abs())).max().item()
            if not err < 2e-4:
                bad.append((name, err, ""))
    return tag, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    t0, n, failures = time.time(), 0, []
    while time.time() - t0 < a.minutes * 60:
        tag, bad = one_case(rnd, n)
        n += 1
        if bad:
            failures.append((tag, bad))
            print("FAIL", tag, bad, flush=True)
    print(f"{n} cases in {time.time() - t0:.0f} s, {len(failures)} failing")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
