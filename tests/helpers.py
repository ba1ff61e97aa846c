"""Input builders shared by the GPU parity tests and the bench (synthetic, seeded)."""
import math

import torch


def level_start_index(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def random_msda_inputs(B, shapes_hw, M, D, Lq, P, seed=0, dtype=torch.float32, lo=-0.2, hi=1.2,
                       value_scale=1.0):
    """value ~ N(0,1)*scale, locations ~ U[lo, hi] (a good part outside [0,1]), weights normalised
    over L*P like ops/test.py:33-36."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shapes_hw, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = (torch.randn(B, S, M, D, generator=g) * value_scale).to(dtype)
    loc = (torch.rand(B, Lq, M, L, P, 2, generator=g) * (hi - lo) + lo).to(dtype)
    aw = torch.rand(B, Lq, M, L, P, generator=g) + 1e-5
    aw = (aw / aw.sum((-1, -2), keepdim=True)).to(dtype)
    return value, shapes, level_start_index(shapes), loc, aw


def encoder_msda_inputs(L, H, W, M=8, D=16, P=4, B=1, seed=0, noise_px=1.0, dtype=torch.float32):
    """Locality-realistic inputs of MVDeTr's shadow transformer: L equal H x W levels,
    Lq = S, reference = identity pixel-centre grid for every level, offsets = the module's initial
    bias grid (ms_deform_attn.py:64-69) + N(0, noise_px) pixels, weights = softmax(N(0,1))."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor([(H, W)] * L, dtype=torch.long)
    S = L * H * W
    value = torch.randn(B, S, M, D, generator=g).to(dtype)
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1)            # [S,2]
    ang = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    dirs = torch.stack([ang.cos(), ang.sin()], -1)
    dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
    bias = dirs.view(M, 1, 1, 2) * torch.arange(1, P + 1).view(1, 1, P, 1)         # [M,1,P,2]
    off = bias[None, None] + noise_px * torch.randn(B, S, M, L, P, 2, generator=g)
    loc = ref[None, :, None, None, None, :] + off / torch.tensor([W, H], dtype=torch.float32)
    aw = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    return value, shapes, level_start_index(shapes), loc.to(dtype).contiguous(), aw.to(dtype)


def fused_train_inputs(L, H, W, M=8, D=16, P=4, B=1, seed=0, noise_px=1.0):
    """The same realistic encoder input in the fused TRAINING pair's form (include/mvdetr_ops.h): -> value, shapes, lsi,
    reference points [1, L, Lq, 2] (one per (query, level): the query's own cell centre), raw [B, Lq, M*L*P*3] = the
    module's single GEMM output in the slice-interleaved, level-outermost layout (offsets in pixels = bias grid +
    N(0, noise_px); logits N(0, 1)), and the row permutation that produced it (slice_major_rows(level_outer=True))."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor([(H, W)] * L, dtype=torch.long)
    S = L * H * W
    value = torch.randn(B, S, M, D, generator=g)
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    cells = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1)           # [S, 2]
    ref_lm = cells[None, None].expand(1, L, S, 2).contiguous()
    ang = torch.arange(M, dtype=torch.float32) * (2.0 * math.pi / M)
    dirs = torch.stack([ang.cos(), ang.sin()], -1)
    dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
    bias = dirs.view(M, 1, 1, 2) * torch.arange(1, P + 1).view(1, 1, P, 1)          # [M, 1, P, 2]
    off = bias[None, None] + noise_px * torch.randn(B, S, M, L, P, 2, generator=g)  # reference order (m, l, p, xy)
    logit = torch.randn(B, S, M, L, P, generator=g)
    hps, n_off = 32 // D, M * L * P * 2
    rows = []
    for l in range(L):                                                              # runs: (level, slice) -> hps heads' offsets, then logits
        for s_ in range(M // hps):
            for h in range(hps):
                rows += [(((s_ * hps + h) * L + l) * P + p) * 2 + xy for p in range(P) for xy in range(2)]
            for h in range(hps):
                rows += [n_off + ((s_ * hps + h) * L + l) * P + p for p in range(P)]
    rows = torch.tensor(rows)
    plain = torch.cat([off.reshape(B, S, -1), logit.reshape(B, S, -1)], -1)
    return value, shapes, level_start_index(shapes), ref_lm, plain.index_select(-1, rows).contiguous(), rows


def smooth_features(n, c, h, w, seed=0, dtype=torch.float32):
    """O(1) band-limited feature maps (a few low spatial frequencies per channel)."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xs = torch.linspace(0, 1, w).view(1, 1, 1, w)
    out = torch.zeros(n, c, h, w)
    for _ in range(4):
        fy = torch.rand(n, c, 1, 1, generator=g) * 6
        fx = torch.rand(n, c, 1, 1, generator=g) * 6
        ph = torch.rand(n, c, 1, 1, generator=g) * 2 * math.pi
        out += torch.randn(n, c, 1, 1, generator=g) * torch.sin(2 * math.pi * (fy * ys + fx * xs) + ph)
    return out.to(dtype)


def pyramid_encoder_inputs(shapes_hw, M=8, D=32, P=4, B=1, seed=0, noise_px=1.5, dtype=torch.float32):
    """Deformable-DETR-style encoder inputs over levels of DIFFERENT sizes: Lq = S, each query's
    reference point is its own normalised cell centre in every level, offsets ~ N(0, noise_px) pixels
    of the sampled level."""
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor(shapes_hw, dtype=torch.long)
    L = shapes.shape[0]
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, M, D, generator=g).to(dtype)
    refs = []
    for H, W in shapes_hw:
        ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
        refs.append(torch.stack([xs / W, ys / H], -1).reshape(-1, 2))
    ref = torch.cat(refs, 0)                                                     # [S,2]
    wh = torch.tensor([[w, h] for h, w in shapes_hw], dtype=torch.float32)       # [L,2]
    off = noise_px * torch.randn(B, S, M, L, P, 2, generator=g) / wh[None, None, None, :, None, :]
    loc = ref[None, :, None, None, None, :] + off
    aw = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P)
    return value, shapes, level_start_index(shapes), loc.to(dtype).contiguous(), aw.to(dtype)
