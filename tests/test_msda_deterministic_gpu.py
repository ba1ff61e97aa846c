"""The opt-in deterministic MSDeformAttn backward (mvdetr_msda_set_backward_deterministic, include/mvdetr_ops.h, ABI 13).

The reference adds grad_value with atomicAdd (ms_deform_im2col_cuda.cuh:125-152): its backward is not reproducible run to
run, and neither are this library's default kernels (fp32 atomics when LDS windows are flushed).  With the mode on,
grad_value is summed in 64-bit fixed point -- integer adds commute -- and must come out BIT-identical however the
workgroups are scheduled, while staying within the parity bars of the default path against the fp64 oracle."""
import pytest
import torch

from helpers import encoder_msda_inputs, level_start_index, random_msda_inputs
from oracle import c_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    return None, MSDA


@pytest.fixture
def deterministic(ops):
    _, MSDA = ops
    prev = MSDA.set_backward_deterministic(True)
    yield MSDA
    MSDA.set_backward_deterministic(prev)
    torch.cuda.synchronize()
    MSDA.release_scratch()                                    # (the mode's accumulators are cached per stream between calls)


def dev(*xs):
    return [x.cuda() for x in xs]


def _disturb():
    """Other work on the device between two runs: shifts which workgroup gets where first."""
    a = torch.randn(2048, 2048, device="cuda")
    (a @ a).sum().item()


CASES = {
    "mvdetr_like_partial_tiles": lambda: encoder_msda_inputs(7, 21, 43, seed=2, noise_px=1.0),
    "wide_offsets_many_misses": lambda: encoder_msda_inputs(3, 24, 40, seed=4, noise_px=6.0),
    "batch2_six_levels": lambda: encoder_msda_inputs(6, 17, 19, B=2, seed=5),
    "five_levels_any_count_kernel": lambda: encoder_msda_inputs(5, 12, 18, M=2, D=16, seed=8),
    # every tap far from its query: the default path stands such tiles down to the lane-group kernel (fp32 atomics); here the
    # window kernel's far path takes them, one 64-bit add per contribution
    "uniform_locations": lambda: random_msda_inputs(1, [(20, 33)] * 3, 4, 16, 3 * 20 * 33, 4, seed=9, lo=-0.1, hi=1.1),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_deterministic_backward_is_bit_reproducible_and_matches_the_oracle(deterministic, case):
    MSDA = deterministic
    value, shapes, lsi, loc, aw = CASES[case]()
    go = torch.randn(value.shape[0], loc.shape[1], value.shape[2] * value.shape[3], generator=torch.Generator().manual_seed(1))
    args = dev(value, shapes, lsi, loc, aw, go)
    runs = []
    for _ in range(4):
        runs.append([x.clone() for x in MSDA.ms_deform_attn_backward(*args, 64)])
        _disturb()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    ref = c_oracle.msda_backward(value.double(), shapes, lsi, loc.double(), aw.double(), go.double())
    W = float(shapes[:, 1].max())
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).double()
    px = loc.double() * wh[None, None, None, :, None, :] - 0.5
    smooth = ((px - px.round()).abs().amin(-1) > 1e-4).double()
    for a, b, name, scale in zip(runs[0], ref, ("grad_value", "grad_loc", "grad_aw"), (1.0, W, 1.0)):
        err = (a.cpu().double() - b).abs() / (scale + b.abs())
        if name == "grad_loc":
            err = err * smooth[..., None]
        assert err.max().item() < 2e-4, name


def test_deterministic_backward_at_wildtrack_size(deterministic):
    """Full size (75,600 tokens, 7 cameras): bit-identical over runs, and equal to the default path within fp32 summation noise."""
    MSDA = deterministic
    value, shapes, lsi, loc, aw = encoder_msda_inputs(7, 60, 180, seed=0, noise_px=1.0)
    go = torch.randn(1, loc.shape[1], 128, generator=torch.Generator().manual_seed(3))
    args = dev(value, shapes, lsi, loc, aw, go)
    first = [x.clone() for x in MSDA.ms_deform_attn_backward(*args, 64)]
    for _ in range(3):
        _disturb()
        again = MSDA.ms_deform_attn_backward(*args, 64)
        for a, b in zip(first, again):
            assert torch.equal(a, b)
    MSDA.set_backward_deterministic(False)
    default = MSDA.ms_deform_attn_backward(*args, 64)
    MSDA.set_backward_deterministic(True)
    scale = default[0].abs().max().item()
    assert (first[0] - default[0]).abs().max().item() < 2e-5 * scale
    for a, b in zip(first[1:], default[1:]):
        assert (a - b).abs().max().item() <= 1e-4 * (1.0 + b.abs().max().item())


@pytest.mark.parametrize("variant", ["tiny", "huge", "wild_weights", "mixed_magnitudes"])
def test_deterministic_binary_point_follows_the_data(deterministic, variant):
    """One binary point per call, 38 bits below max|grad_out| x max(1, max|aw|): tiny, huge and unnormalised inputs keep the
    accuracy of the default path relative to the gradient's own scale; queries nine decades below the call's maximum keep
    what 38 bits leave them."""
    MSDA = deterministic
    value, shapes, lsi, loc, aw = encoder_msda_inputs(7, 19, 37, M=8, D=16, seed=21, noise_px=1.5)
    go = torch.randn(1, loc.shape[1], 128, generator=torch.Generator().manual_seed(5))
    if variant == "tiny":
        go = go * 1e-30
    elif variant == "huge":
        go = go * 1e25
    elif variant == "wild_weights":
        aw = (aw - 0.02) * 300.0
    elif variant == "mixed_magnitudes":
        go = go * torch.logspace(-6, 3, go.shape[1]).view(1, -1, 1)
    ref = c_oracle.msda_backward(value.double(), shapes, lsi, loc.double(), aw.double(), go.double())[0]
    gv = MSDA.ms_deform_attn_backward(*dev(value, shapes, lsi, loc, aw, go), 64)[0].cpu().double()
    assert torch.isfinite(gv).all()
    if variant != "mixed_magnitudes":
        assert (gv - ref).abs().max().item() < 2e-5 * ref.abs().max().item()
    else:
        # absolute steps of 2^-38 of the call's maximum (1e3 x ~4): a token whose gradient is 1e-6 of that still has ~18 bits
        flat_ref, flat_gv = ref.flatten(2)[0], gv.flatten(2)[0]
        local = torch.nn.functional.max_pool1d(flat_ref.abs().amax(1)[None, None], 1201, 1, 600)[0, 0]
        floor = ref.abs().max().item() * 2.0 ** -30
        assert ((flat_gv - flat_ref).abs().amax(1) / (local + floor)).max().item() < 1e-3


def test_deterministic_backward_propagates_nonfinite_gradients(deterministic):
    MSDA = deterministic
    value, shapes, lsi, loc, aw = encoder_msda_inputs(7, 12, 20, M=8, D=16, seed=22)
    go = torch.randn(1, loc.shape[1], 128, generator=torch.Generator().manual_seed(6))
    go[0, 33, 5] = float("inf")
    go[0, 700, 17] = float("nan")
    args = dev(value, shapes, lsi, loc, aw, go)
    gv = MSDA.ms_deform_attn_backward(*args, 64)[0].cpu()
    again = MSDA.ms_deform_attn_backward(*args, 64)[0].cpu()
    assert torch.equal(torch.isfinite(gv), torch.isfinite(again))
    fin = torch.isfinite(gv)
    assert torch.equal(gv[fin], again[fin])
    ok = go.clone()
    ok[0, 33, 5] = 0
    ok[0, 700, 17] = 0
    ref = c_oracle.msda_backward(value.double(), shapes, lsi, loc.double(), aw.double(), ok.double())[0]
    gv4, ref4 = gv.view(1, -1, 8, 16), ref.view(1, -1, 8, 16)
    bad = ~torch.isfinite(gv4)
    assert bad[..., 0, 5].any() and bad[..., 1, 1].any()
    clean = torch.ones(8, 16, dtype=torch.bool)
    clean[0, 5] = clean[1, 1] = False
    assert torch.isfinite(gv4[..., clean]).all()
    assert (gv4[..., clean].double() - ref4[..., clean]).abs().max().item() < 2e-4


def test_deterministic_mode_refuses_what_it_cannot_serve(deterministic):
    """Decoder-like calls (num_query != spatial_size), other head widths and fp64 have no deterministic kernel: loud, not
    served by one that is not."""
    MSDA = deterministic
    value, shapes, lsi, loc, aw = random_msda_inputs(1, [(8, 8), (4, 4)], 2, 8, 11, 2, seed=3)
    go = torch.randn(1, 11, 16)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(*dev(value, shapes, lsi, loc, aw, go), 64)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(*dev(value.double(), shapes, lsi, loc.double(), aw.double(), go.double()), 64)
    # unequal level shapes are device data: the kernel answers with NaN
    from helpers import pyramid_encoder_inputs
    value, shapes, lsi, loc, aw = pyramid_encoder_inputs([(10, 37), (20, 11), (7, 7)], M=4, D=16, seed=10, noise_px=3.0)
    go = torch.randn(1, loc.shape[1], 64)
    gv = MSDA.ms_deform_attn_backward(*dev(value, shapes, lsi, loc, aw, go), 64)[0]
    assert torch.isnan(gv).all()
    MSDA.set_backward_deterministic(False)
    gv = MSDA.ms_deform_attn_backward(*dev(value, shapes, lsi, loc, aw, go), 64)[0]
    assert torch.isfinite(gv).all()


def test_module_refuses_early_what_the_deterministic_mode_cannot_serve(deterministic):
    """a model with 32-channel heads used to run its whole forward and fail at backward time with error 801.  The
    module says so at the first forward that needs gradients; inference is not affected, and 16-channel heads train."""
    MSDA = deterministic
    assert MSDA.backward_deterministic()
    from mvdetr_amd.ops.modules import MSDeformAttn
    L, H, W = 3, 8, 12
    shapes = torch.tensor([[H, W]] * L)
    S = L * H * W
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 1, 1, 2).repeat(L, L, 4, 1)[None].cuda()
    for d_model, ok in ((128, True), (256, False)):                  # 8 heads: 16- / 32-channel heads
        mod = MSDeformAttn(d_model, L, 8, 4).cuda()
        q = torch.randn(1, S, d_model, device="cuda")
        with torch.no_grad():
            mod(q, ref, q, shapes.cuda(), level_start_index(shapes).cuda())           # inference: always served
        if ok:
            mod(q, ref, q, shapes.cuda(), level_start_index(shapes).cuda()).sum().backward()
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in mod.parameters())
        else:
            with pytest.raises(RuntimeError, match="deterministic backward"):
                mod(q, ref, q, shapes.cuda(), level_start_index(shapes).cuda())
    MSDA.set_backward_deterministic(False)
    assert not MSDA.backward_deterministic()


def test_deterministic_fused_training_pair(deterministic):
    """The fused training backward (mvdetr_msda_backward_fused_f32) in the same mode: gradients of value and of the raw
    offsets / logits tensor bit-identical over runs and equal to the default pair within summation noise."""
    MSDA = deterministic
    from mvdetr_amd.ops.functions.ms_deform_attn_func import MSDeformAttnFusedFunction
    from helpers import fused_train_inputs
    value, shapes, lsi, ref, raw, _ = fused_train_inputs(7, 24, 44, M=8, D=16, B=2, seed=17, noise_px=1.0)
    go = torch.randn(2, value.shape[1], 128, generator=torch.Generator().manual_seed(18))

    def run():
        v = value.cuda().requires_grad_(True)
        r = raw.cuda().requires_grad_(True)
        out = MSDeformAttnFusedFunction.apply(v, shapes.cuda(), lsi.cuda(), ref.cuda(), r)
        out.backward(go.cuda())
        return v.grad.clone(), r.grad.clone()

    first = run()
    for _ in range(3):
        _disturb()
        for a, b in zip(first, run()):
            assert torch.equal(a, b)
    MSDA.set_backward_deterministic(False)
    default = run()
    MSDA.set_backward_deterministic(True)
    assert (first[0] - default[0]).abs().max().item() < 2e-5 * default[0].abs().max().item()
    assert (first[1] - default[1]).abs().max().item() < 1e-4 * (1.0 + default[1].abs().max().item())
