#!/usr/bin/env python3
"""Randomised parity run on the GPU box: encoder-shaped MSDA calls of random geometry through every dispatch route
(public forward, backward, fused forward in the plain and slice-interleaved layouts, the host path) against the oracle,
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
    for a, b, name, scale in zip(grads, ref, ("grad_value", "grad_loc", "grad_aw"), (1.0, float(max(W, H)), 1.0)):
        err = (a - b).abs() / (scale + b.abs())
        if name == "grad_loc":
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
