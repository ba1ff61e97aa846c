"""The fused TRAINING pair (ABI 10): MSDeformAttn.forward / backward from the module's raw offsets / logits, softmax and
location arithmetic inside the kernels both ways (include/mvdetr_ops.h; replaces ms_deform_attn.py:100-114 + func.py:21-38
+ cuh:237-299, 956-1327 + torch's backward of the module arithmetic).

Oracle: the C restatement's backward (oracle/oracle.c, fp64) for grad_value / grad_sampling_loc / grad_attn_weight, chained
by hand (fp64 torch on the CPU) through loc = ref + off / (W, H) and the softmax -- the reference's own arithmetic -- and
fp64 autograd through oracle/torch_oracle.msda_module for the module-level test.  Tolerance: 2e-4 relative (fp32 kernels,
fixed-point grad_value windows), as for the unfused backward (tests/test_msda_gpu.py).
"""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import encoder_msda_inputs, level_start_index  # noqa: E402
from oracle import c_oracle, torch_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    return MSDA


def _raw_inputs(L, H, W, M, D, B, seed, noise_px):
    """value, shapes, lsi, per-(query, level) reference points [1, L, Lq, 2], raw offsets (pixels) [B, Lq, M, L, P, 2], raw
    logits [B, Lq, M, L, P] -- the realistic encoder input of SURVEY 8d in the module's raw form."""
    value, shapes, lsi, loc, aw = encoder_msda_inputs(L, H, W, M, D, 4, B=B, seed=seed, noise_px=noise_px)
    S = L * H * W
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    cells = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1)                       # [S, 2]
    g = torch.Generator().manual_seed(seed + 100)
    ref_ql = (cells[:, None, :] + 0.002 * torch.randn(S, L, 2, generator=g)).contiguous()        # [Lq, L, 2]
    off = (loc - ref_ql[None, :, None, :, None, :]) * torch.tensor([W, H], dtype=torch.float32)
    logit = torch.randn(B, S, M, L, 4, generator=g) * 1.5
    logit[:, ::7, :, 2, 1] += 12.0                                   # a late, much larger logit: the lazy softmax must rescale
    return value, shapes, lsi, ref_ql, off.contiguous(), logit


def _to_raw(MSDA, off, logit, M, L, D):
    B, Lq = off.shape[:2]
    rows = torch.tensor(MSDA.slice_major_rows(M, L, 4, D, level_outer=True))
    plain = torch.cat([off.reshape(B, Lq, -1), logit.reshape(B, Lq, -1)], -1)
    return plain.index_select(-1, rows).contiguous(), rows


def _reference_grads(value, shapes, lsi, ref_ql, off, logit, go):
    """fp64 chain on the CPU: C oracle's backward at (loc, aw), then the module arithmetic's backward by hand."""
    H, W = int(shapes[0, 0]), int(shapes[0, 1])
    wh = torch.tensor([W, H], dtype=torch.float64)
    off64, logit64 = off.double(), logit.double()
    loc = ref_ql.double()[None, :, None, :, None, :] + off64 / wh
    B, Lq, M, L, P = logit.shape
    aw = torch.softmax(logit64.flatten(-2), -1).view(B, Lq, M, L, P)
    gv, gl, ga = c_oracle.msda_backward(value.double(), shapes, lsi, loc.contiguous(), aw.contiguous(), go.double())
    g_off = gl / wh                                                  # loc = ref + off / (W, H)
    dsum = (aw * ga).sum((-1, -2), keepdim=True)
    g_logit = aw * (ga - dsum)                                       # softmax backward
    return gv, g_off, g_logit, loc, aw


@pytest.mark.parametrize("L,H,W,B,noise,M,D", [
    (7, 13, 21, 1, 1.0, 8, 16), (6, 12, 34, 2, 2.5, 8, 16), (7, 6, 16, 1, 0.0, 8, 16),
    # ABI 13: every encoder shape of the LDS-tiled kernels -- other level counts (16-channel heads: the one-pass backward; the
    # forward is the inference kernel of that shape + a statistics pass) ...
    (3, 12, 20, 1, 1.0, 8, 16), (5, 9, 17, 2, 1.0, 4, 16), (8, 10, 18, 1, 1.5, 8, 16), (12, 7, 19, 1, 1.0, 4, 16), (16, 6, 14, 1, 2.0, 2, 16),
    # ... and 32-channel heads (msda_bwd_value_tok<32, fused> + the level-groups sampling kernel on the raw tensor)
    (6, 12, 20, 1, 1.0, 4, 32), (7, 9, 21, 2, 2.0, 8, 32), (16, 6, 14, 1, 1.0, 2, 32), (3, 11, 13, 1, 0.0, 2, 32), (9, 5, 33, 1, 6.0, 1, 32)])
def test_fused_training_pair_vs_oracle(ops, L, H, W, B, noise, M, D):
    MSDA = ops
    value, shapes, lsi, ref_ql, off, logit = _raw_inputs(L, H, W, M, D, B, seed=3, noise_px=noise)
    raw, rows = _to_raw(MSDA, off, logit, M, L, D)
    ref_lm = ref_ql.transpose(0, 1).contiguous()[None]                # [1, L, Lq, 2]
    d = dict(value=value.cuda(), shapes=shapes.cuda(), lsi=lsi.cuda(), ref=ref_lm.cuda(), raw=raw.cuda())
    assert MSDA.fused_train_supported(B, value.shape[1], M, D, L, value.shape[1], 4)
    out, stats = MSDA.ms_deform_attn_forward_fused_train(d["value"], d["shapes"], d["lsi"], d["ref"], d["raw"])
    # the forward is the inference kernel: the same bits
    inf = MSDA.ms_deform_attn_forward_fused(d["value"], d["shapes"], d["lsi"], d["ref"], None, None, raw=d["raw"],
                                            ref_level_major=True, raw_level_outer=True)
    assert torch.equal(out, inf)
    # the statistics rebuild the softmax (the kernel's reference maximum is lazy: only the product matters)
    a_got = torch.exp(logit.cuda() - stats[..., 0][..., None, None]) * stats[..., 1][..., None, None]
    a_want = torch.softmax(logit.double().flatten(-2), -1).view_as(logit)
    assert (a_got.cpu().double() - a_want).abs().max().item() < 2e-6
    go = torch.randn(B, value.shape[1], M * D, generator=torch.Generator().manual_seed(5))
    gv_ref, goff_ref, glogit_ref, loc, aw = _reference_grads(value, shapes, lsi, ref_ql, off, logit, go)
    want_out = c_oracle.msda_forward(value.double(), shapes, lsi, loc.contiguous(), aw.contiguous())
    assert (out.cpu().double() - want_out).abs().max().item() < 1e-4
    gv, graw = MSDA.ms_deform_attn_backward_fused(go.cuda(), d["value"], d["shapes"], d["lsi"], d["ref"], d["raw"], stats, out)
    inv = torch.empty_like(rows)
    inv[rows] = torch.arange(rows.numel())
    gplain = graw.cpu().double().index_select(-1, inv)
    n_off = M * L * 4 * 2
    goff = gplain[..., :n_off].reshape(goff_ref.shape)
    glogit = gplain[..., n_off:].reshape(glogit_ref.shape)
    err_v = (gv.cpu().double() - gv_ref).abs() / (1.0 + gv_ref.abs())
    assert err_v.max().item() < 2e-4, "grad_value"
    # offsets: away from texel centres (the bilinear blend's derivative jumps there, test_msda_gpu.py)
    px = loc * torch.tensor([W, H], dtype=torch.float64) - 0.5
    smooth = ((px - px.round()).abs().amin(-1) > 1e-4).double()
    err_o = (goff - goff_ref).abs() / (1.0 + goff_ref.abs()) * smooth[..., None]
    assert err_o.max().item() < 2e-4, "grad of the raw offsets"
    err_l = (glogit - glogit_ref).abs() / (1.0 + glogit_ref.abs())
    assert err_l.max().item() < 2e-4, "grad of the raw logits"
    assert goff_ref.abs().max().item() > 0.05 and glogit_ref.abs().max().item() > 0.05 and gv_ref.abs().max().item() > 0.05


def test_module_trains_through_the_fused_pair(ops):
    """MSDeformAttn with gradients enabled takes the fused pair (ONE GEMM, no sampling_locations / attention_weights) and
    gives the gradients of the unfused path and of fp64 autograd through the oracle's module, for the tokens and every
    parameter."""
    from mvdetr_amd.ops.modules import MSDeformAttn
    from mvdetr_amd.ops.functions import ms_deform_attn_func as ff
    torch.manual_seed(0)
    C, L, M, P, H, W = 128, 7, 8, 4, 10, 22
    S = L * H * W
    attn = MSDeformAttn(C, L, M, P).cuda()
    with torch.no_grad():
        attn.sampling_offsets.weight.normal_(0, 0.02)
        attn.attention_weights.weight.normal_(0, 0.1)
    shapes = torch.tensor([[H, W]] * L).cuda()
    lsi = level_start_index(shapes.cpu()).cuda()
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 1, 1, 2).repeat(L, L, P, 1)[None].cuda()     # [1, S, L, P, 2]
    tok = torch.randn(1, S, C, generator=torch.Generator().manual_seed(1))
    gout = torch.randn(1, S, C, generator=torch.Generator().manual_seed(2)).cuda()

    calls = {"fused": 0, "unfused": 0}
    real_f, real_u = ff.MSDeformAttnFusedFunction.apply, ff.MSDeformAttnFunction.apply

    def run(fused):
        attn.fused_training = fused
        attn.zero_grad()
        x = tok.clone().cuda().requires_grad_(True)
        out = attn(x, ref, x, shapes, lsi)
        out.backward(gout)
        return out.detach(), x.grad.detach(), {k: p.grad.detach().clone() for k, p in attn.named_parameters()}

    import mvdetr_amd.ops.modules.ms_deform_attn as mod
    mod.MSDeformAttnFusedFunction = type("F", (), {"apply": staticmethod(lambda *a: (calls.__setitem__("fused", calls["fused"] + 1), real_f(*a))[1])})
    mod.MSDeformAttnFunction = type("U", (), {"apply": staticmethod(lambda *a: (calls.__setitem__("unfused", calls["unfused"] + 1), real_u(*a))[1])})
    try:
        o1, gx1, gp1 = run(True)
        o0, gx0, gp0 = run(False)
    finally:
        mod.MSDeformAttnFusedFunction, mod.MSDeformAttnFunction = ff.MSDeformAttnFusedFunction, ff.MSDeformAttnFunction
        attn.fused_training = True
    assert calls == {"fused": 1, "unfused": 1}
    # fp64 autograd through the oracle's restatement of the module
    params = {k: v.detach().cpu().double().requires_grad_(True) for k, v in attn.state_dict().items()}
    x64 = tok.double().requires_grad_(True)
    o64 = torch_oracle.msda_module(params, x64, ref.cpu().double(), x64, shapes.cpu(), M, P)
    o64.backward(gout.cpu().double())
    assert (o1.cpu().double() - o64.detach()).abs().max().item() < 1e-4

    def close(a, b, name):
        a, b = a.cpu().double(), b.cpu().double()
        assert (a - b).abs().max().item() < 3e-4 * (1.0 + b.abs().max().item()), name
    close(gx1, x64.grad, "tokens vs oracle")
    close(gx1, gx0, "tokens vs unfused")
    for k in gp1:
        close(gp1[k], params[k].grad, f"{k} vs oracle")
        close(gp1[k], gp0[k], f"{k} vs unfused")
        assert params[k].grad.abs().max().item() > 0


@pytest.mark.parametrize("name,L,H,W,B,D", [("wildtrack", 7, 60, 180, 1, 16), ("multiviewx_batch4", 6, 80, 125, 4, 16),
                                            ("stress16", 16, 60, 180, 1, 32)])
def test_fused_training_pair_at_wildtrack_size(ops, name, L, H, W, B, D):
    """Full Wildtrack shape (75,600 queries x 8 heads x 7 levels x 4 points), BASELINE configs[3]'s encoder shape (MultiviewX,
    6 cameras, 4 frames per step: 240,000 queries) and configs[4]'s (16 cameras, 32-channel heads: 172,800 queries x 8 heads x 16
    levels -- the general route: msda_fwd_group<SPLIT> + statistics, msda_bwd_value_tok<32, fused> + the level-groups sampling
    kernel): every element of grad_value and of the raw gradient against the fp64 C oracle chained through the module arithmetic."""
    MSDA = ops
    M = 8
    value, shapes, lsi, ref_ql, off, logit = _raw_inputs(L, H, W, M, D, B, seed=0, noise_px=1.0)
    raw, rows = _to_raw(MSDA, off, logit, M, L, D)
    ref_lm = ref_ql.transpose(0, 1).contiguous()[None]
    dv, ds, dl, dr, draw = value.cuda(), shapes.cuda(), lsi.cuda(), ref_lm.cuda(), raw.cuda()
    out, stats = MSDA.ms_deform_attn_forward_fused_train(dv, ds, dl, dr, draw)
    go = torch.randn(B, value.shape[1], M * D, generator=torch.Generator().manual_seed(9))
    gv, graw = MSDA.ms_deform_attn_backward_fused(go.cuda(), dv, ds, dl, dr, draw, stats, out)
    gv_ref, goff_ref, glogit_ref, loc, aw = _reference_grads(value, shapes, lsi, ref_ql, off, logit, go)
    inv = torch.empty_like(rows)
    inv[rows] = torch.arange(rows.numel())
    gplain = graw.cpu().double().index_select(-1, inv)
    n_off = M * L * 4 * 2
    goff, glogit = gplain[..., :n_off].reshape(goff_ref.shape), gplain[..., n_off:].reshape(glogit_ref.shape)
    assert ((gv.cpu().double() - gv_ref).abs() / (1.0 + gv_ref.abs())).max().item() < 2e-4
    px = loc * torch.tensor([W, H], dtype=torch.float64) - 0.5
    smooth = ((px - px.round()).abs().amin(-1) > 1e-4).double()
    assert ((goff - goff_ref).abs() / (1.0 + goff_ref.abs()) * smooth[..., None]).max().item() < 2e-4
    assert ((glogit - glogit_ref).abs() / (1.0 + glogit_ref.abs())).max().item() < 2e-4
    # deterministic part: the raw gradient has no atomics in it
    _, graw2 = MSDA.ms_deform_attn_backward_fused(go.cuda(), dv, ds, dl, dr, draw, stats, out)
    assert torch.equal(graw, graw2)
    # adjoint identity in value: <go, f(v)> = <grad_value, v>  (f is linear in value)
    lhs = (go.double() * out.cpu().double()).sum().item()
    rhs = (gv.cpu().double() * value.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2
    assert math.isfinite(lhs)


def _fused_sweep_cases(n=8, seed=77):
    import random
    rnd = random.Random(seed)
    cases = []
    for i in range(n):
        L = rnd.choice([6, 7])
        H, W = rnd.randint(1, 30), rnd.randint(1, 70)
        B = rnd.choice([1, 1, 2])
        noise = rnd.choice([0.0, 1.0, 3.0, 8.0])                  # 8 px: most taps leave their windows (the far paths)
        cases.append((i, L, H, W, B, noise, 8, 16))
    # the general routes (ABI 13): any level count, both head widths
    for i in range(n, 2 * n):
        L = rnd.choice([3, 4, 5, 8, 9, 12, 16, 6, 7])
        D = rnd.choice([16, 32, 32])
        M = rnd.choice([2, 4, 8]) if D == 16 else rnd.choice([1, 2, 4])
        H, W = rnd.randint(1, 24), rnd.randint(1, 60)
        while L * H * W * M * D * L > 60_000_000:                 # (the fp64 chain on the CPU)
            H, W = max(1, H // 2), max(1, W * 2 // 3)
        cases.append((i, L, H, W, rnd.choice([1, 1, 2]), rnd.choice([0.0, 1.0, 3.0, 8.0]), M, D))
    return cases


@pytest.mark.parametrize("i,L,H,W,B,noise,M,D", _fused_sweep_cases())
def test_fused_training_pair_seeded_sweep(ops, i, L, H, W, B, noise, M, D):
    """Random geometry (partial tiles, single rows / columns, two frames, offsets from 0 to 8 px) through the fused training pair
    against the fp64 chain; the regression form of tools/fuzz_parity.py's fused leg."""
    MSDA = ops
    value, shapes, lsi, ref_ql, off, logit = _raw_inputs(L, H, W, M, D, B, seed=200 + i, noise_px=noise)
    S = value.shape[1]
    if not MSDA.fused_train_supported(B, S, M, D, L, S, 4):
        pytest.skip("shape outside the fused pair")
    raw, rows = _to_raw(MSDA, off, logit, M, L, D)
    ref_lm = ref_ql.transpose(0, 1).contiguous()[None]
    dv = dict(value=value.cuda(), shapes=shapes.cuda(), lsi=lsi.cuda(), ref=ref_lm.cuda(), raw=raw.cuda())
    out, stats = MSDA.ms_deform_attn_forward_fused_train(dv["value"], dv["shapes"], dv["lsi"], dv["ref"], dv["raw"])
    go = torch.randn(B, S, M * D, generator=torch.Generator().manual_seed(300 + i))
    gv_ref, goff_ref, glogit_ref, loc, aw = _reference_grads(value, shapes, lsi, ref_ql, off, logit, go)
    want_out = c_oracle.msda_forward(value.double(), shapes, lsi, loc.contiguous(), aw.contiguous())
    assert (out.cpu().double() - want_out).abs().max().item() < 1e-4
    gv, graw = MSDA.ms_deform_attn_backward_fused(go.cuda(), dv["value"], dv["shapes"], dv["lsi"], dv["ref"], dv["raw"], stats, out)
    inv = torch.empty_like(rows)
    inv[rows] = torch.arange(rows.numel())
    gplain = graw.cpu().double().index_select(-1, inv)
    n_off = M * L * 4 * 2
    goff, glogit = gplain[..., :n_off].reshape(goff_ref.shape), gplain[..., n_off:].reshape(glogit_ref.shape)
    assert ((gv.cpu().double() - gv_ref).abs() / (1.0 + gv_ref.abs())).max().item() < 2e-4, "grad_value"
    px = loc * torch.tensor([W, H], dtype=torch.float64) - 0.5
    smooth = ((px - px.round()).abs().amin(-1) > 1e-4).double()
    assert ((goff - goff_ref).abs() / (1.0 + goff_ref.abs()) * smooth[..., None]).max().item() < 2e-4, "grad of the raw offsets"

    # (round 4 measured this against 4 + |ref|: the kernels formed the pixel position as ONE fp32 number, 8e-6 px off at x ~ 143,
    # which a blend of +-19 dots turns into 3e-4.  Round 5's kernels split the position -- common.h: fused_px -- and the bar is
    # the other gradients' again)
    assert ((glogit - glogit_ref).abs() / (1.0 + glogit_ref.abs())).max().item() < 2e-4, "grad of the raw logits"
