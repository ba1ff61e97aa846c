"""Parity at the sizes BASELINE.json names (VERDICT r01 "next round" item 1): the kernels that the benchmark actually
times -- the fused, slice-interleaved entry with a shared reference point -- and the whole hot path, at full Wildtrack
size; MultiviewX (L = 6, 80 x 125, B = 1 and 4) and the 16-camera stress configuration (L = 16, D = 32, S = 172,800)
against the C oracle, with adjoint identities for the backward.  The tests of ops/test.py:21-60, in spirit, at size."""
import pytest
import torch

from helpers import encoder_msda_inputs, smooth_features
from mvdetr_amd import geometry
from oracle import c_oracle, frame_oracle, torch_oracle

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4        # BASELINE.json north_star: within 1e-4 fp32 on O(1) features


@pytest.fixture(scope="module")
def ops():
    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    from mvdetr_amd.ops.functions import MSDeformAttnFunction
    from mvdetr_amd.ops.modules import MSDeformAttn
    return MSDA, MSDeformAttnFunction, MSDeformAttn


def _perturb(attn, seed):
    """Seeded stand-in for learned weights: the reference initialises the offset / attention projections to zero
    (ms_deform_attn.py:62-73), which would make every query sample one constant pattern."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        attn.sampling_offsets.weight.copy_(0.05 * torch.randn(attn.sampling_offsets.weight.shape, generator=g))
        attn.attention_weights.weight.copy_(0.1 * torch.randn(attn.attention_weights.weight.shape, generator=g))


def _identity_reference(L, H, W, P):
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    return torch.stack([xs / W, ys / H], -1).reshape(-1, 1, 1, 2).repeat(L, L, P, 1)[None]       # [1, L*H*W, L, P, 2]


# ---- (a) the benchmarked kernel at the benchmarked size ------------------------------------------------------------------
@pytest.mark.parametrize("config,batch", [("wildtrack", 1), ("multiviewx", 1), ("multiviewx", 4)])
def test_fused_module_at_full_size_vs_oracle(ops, config, batch):
    MSDA, _, MSDeformAttn = ops
    geom = geometry.GEOMETRIES[config]
    L, (H, W), C, M, P = geom.num_cam, (geom.Rworld_shape[0] // 2, geom.Rworld_shape[1] // 2), geom.feat_channels, 8, 4
    S = L * H * W
    torch.manual_seed(5)
    attn = MSDeformAttn(C, L, M, P).eval()
    _perturb(attn, 11)
    tokens = torch.randn(batch, S, C)
    query = tokens + 0.3 * torch.randn(batch, S, C)
    shapes = torch.tensor([[H, W]] * L)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    ref = _identity_reference(L, H, W, P).expand(batch, -1, -1, -1, -1)
    dev = torch.device("cuda")
    attn_d = attn.to(dev)
    with torch.no_grad():
        got = attn_d(query.to(dev), ref.to(dev), tokens.to(dev), shapes.to(dev), lsi.to(dev))
        assert MSDA.last_forward_impl() == "tile_fused"                    # the fused entry took the call
        attn_d.fused_inference = False
        unfused = attn_d(query.to(dev), ref.to(dev), tokens.to(dev), shapes.to(dev), lsi.to(dev))
        attn_d.fused_inference = True
    params = {k: v.detach().cpu() for k, v in attn.state_dict().items()}
    with torch.no_grad():
        want = torch.cat([torch_oracle.msda_module(params, query[b:b + 1], ref[b:b + 1], tokens[b:b + 1], shapes, M, P)
                          for b in range(batch)])
    assert want.abs().max().item() > 0.5
    assert (got.cpu() - want).abs().max().item() < FP32_TOL
    assert (unfused.cpu() - want).abs().max().item() < FP32_TOL


# ---- (b) north_star's BEV criterion at size -------------------------------------------------------------------------------
@pytest.mark.parametrize("config", ["wildtrack", "multiviewx"])
def test_hot_path_bev_within_1e4_at_full_size(config):
    from mvdetr_amd.model import build_model
    geom = geometry.GEOMETRIES[config]
    model = build_model(config, seed=0).eval()
    for i, layer in enumerate(model.world_feat.encoder.layers):
        _perturb(layer.self_attn, 20 + i)
    N, C, (h, w) = geom.num_cam, geom.feat_channels, geom.Rimg_shape
    # O(1) band-limited feature maps: on white noise the fp32 formulation of the warp itself is only good to ~2e-4 near
    # the horizon (tests/test_warp_gpu.py), which says nothing about the kernels
    feat = smooth_features(N, C, h, w, seed=4)
    Mx = geometry.random_affine_mats(1, N, geom.input_img_shape, seed=2, translate=0.05, scale=(0.9, 1.1))
    proj = model.frame_proj_mats(Mx)
    model = model.cuda()
    with torch.no_grad():
        got = model.hot_path(feat.cuda().contiguous(memory_format=torch.channels_last), proj.cuda()).cpu()
        got_nchw = model.hot_path(feat.cuda(), proj.cuda()).cpu()
        p = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        ref = model.world_feat.encoder.reference_points.detach().cpu()
        want = frame_oracle.world_from_features(p, feat, proj, model.Rworld_shape, ref, N, n_heads=8, n_points=4)
    assert got.shape == want.shape == (1, C, *geom.Rworld_shape)
    assert want.abs().max().item() > 0.1
    assert (got - want).abs().max().item() < FP32_TOL
    assert (got_nchw - want).abs().max().item() < FP32_TOL


# ---- (c) MultiviewX and the 16-camera stress configuration against the C oracle -----------------------------------------------
def _cfg(config):
    geom = geometry.GEOMETRIES[config]
    return geom.num_cam, geom.Rworld_shape[0] // 2, geom.Rworld_shape[1] // 2, geom.feat_channels // 8


@pytest.mark.parametrize("config,batch", [("multiviewx", 1), ("multiviewx", 4), ("stress16", 1)])
def test_forward_backward_at_full_size_vs_c_oracle(ops, config, batch):
    MSDA, F, _ = ops
    L, H, W, D = _cfg(config)
    value, shapes, lsi, loc, aw = encoder_msda_inputs(L, H, W, 8, D, 4, B=batch, seed=3)
    S = L * H * W
    dv = [x.cuda() for x in (value, shapes, lsi, loc, aw)]
    out = MSDA.ms_deform_attn_forward(*dv, 64).cpu()
    want = c_oracle.msda_forward(value, shapes, lsi, loc, aw)                # fp32 C oracle, every query
    assert (out - want).abs().max().item() < FP32_TOL
    sub = slice(0, S, 211)                                                   # fp64 truth on a strided subset
    want64 = c_oracle.msda_forward(value.double(), shapes, lsi, loc[:, sub].double().contiguous(),
                                   aw[:, sub].double().contiguous())
    assert (out[:, sub].double() - want64).abs().max().item() < FP32_TOL
    # backward: adjoint identities <g, f(v)> = <dv, v> = <dw, w> over the whole problem + a strided fp64 subset
    go = torch.randn(batch, S, 8 * D, generator=torch.Generator().manual_seed(9))
    gv, gl, ga = MSDA.ms_deform_attn_backward(*dv, go.cuda(), 64)
    lhs = (go.double() * out.double()).sum().item()
    assert abs(lhs - (gv.cpu().double() * value.double()).sum().item()) < 1e-6 * (abs(lhs) + 1e3)
    assert abs(lhs - (ga.cpu().double() * aw.double()).sum().item()) < 1e-6 * (abs(lhs) + 1e3)
    sub = slice(0, S, 1999)
    lo, awc, goc = [x[:, sub].double().contiguous() for x in (loc, aw, go)]
    _, rl, ra = c_oracle.msda_backward(value.double(), shapes, lsi, lo, awc, goc)
    err = (gl[:, sub].cpu().double() - rl).abs()
    assert (err / (50 + rl.abs())).max().item() < 1e-4
    erra = (ga[:, sub].cpu().double() - ra).abs()
    assert (erra / (1 + ra.abs())).max().item() < 5e-4


def test_stress16_fused_entry_vs_unfused_and_oracle(ops):
    """16 cameras, 32 channels per head: the many-camera kernel behind the fused entry at S = 172,800."""
    MSDA, _, MSDeformAttn = ops
    L, H, W, D = _cfg("stress16")
    C, M, P, S = 8 * D, 8, 4, L * H * W
    torch.manual_seed(6)
    attn = MSDeformAttn(C, L, M, P).eval()
    _perturb(attn, 31)
    tokens = torch.randn(1, S, C)
    query = tokens + 0.3 * torch.randn(1, S, C)
    shapes = torch.tensor([[H, W]] * L)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    ref = _identity_reference(L, H, W, P)
    dev = torch.device("cuda")
    attn_d = attn.to(dev)
    with torch.no_grad():
        got = attn_d(query.to(dev), ref.to(dev), tokens.to(dev), shapes.to(dev), lsi.to(dev)).cpu()
        attn_d.fused_inference = False
        unfused = attn_d(query.to(dev), ref.to(dev), tokens.to(dev), shapes.to(dev), lsi.to(dev)).cpu()
    assert (got - unfused).abs().max().item() < 5e-5
    sub = slice(0, S, 37)                                                    # oracle on a strided subset of the queries
    params = {k: v.detach().cpu() for k, v in attn.state_dict().items()}
    with torch.no_grad():
        want = torch_oracle.msda_module(params, query[:, sub], ref[:, sub], tokens, shapes, M, P)
    assert (got[:, sub] - want).abs().max().item() < FP32_TOL


# ---- (d) configs[2]'s per-rank kernel at Wildtrack size ---------------------------------------------------
@pytest.mark.parametrize("levels", [(0, 1), (6, 7), (0, 4), (4, 7)])
def test_fused_query_levels_at_wildtrack_size_vs_oracle(ops, levels):
    """mvdetr_msda_forward_fused_levels_f32 as ONE rank of a view-sharded Wildtrack frame calls it: value holds all 7
    cameras' 75,600 tokens, the queries are the tokens of levels [l0, l1) -- 1-of-7 (first and last camera) and the 4 + 3
    split of a 2-rank run.  Every query against the fp32 C oracle, a strided subset against fp64."""
    MSDA, _, _ = ops
    L, H, W, M, D, P = 7, 60, 180, 8, 16, 4
    S = L * H * W
    l0, l1 = levels
    q0, q1 = l0 * H * W, l1 * H * W
    value, shapes, lsi, loc, aw = encoder_msda_inputs(L, H, W, M, D, P, seed=13)
    # raw tensors whose module arithmetic reproduces (loc, aw) up to rounding: offsets in pixels around the identity
    # reference grid, logits = log(weights)
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref1 = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1)                            # [S, 2]
    off = (loc - ref1[None, :, None, None, None, :]) * torch.tensor([W, H], dtype=torch.float32)    # [1, S, M, L, P, 2]
    logit = aw.clamp_min(1e-30).log()
    ref = ref1.view(1, S, 1, 1, 2).expand(1, S, L, P, 2).contiguous()
    loc_k = torch_oracle.msda_sampling_locations(ref, off, shapes)
    aw_k = torch.softmax(logit.flatten(3), -1).view(1, S, M, L, P)
    dev = torch.device("cuda")
    got = MSDA.ms_deform_attn_forward_fused(
        value.to(dev), shapes.to(dev), lsi.to(dev), ref[:, q0:q1].contiguous().to(dev), off[:, q0:q1].contiguous().to(dev),
        logit[:, q0:q1].contiguous().to(dev), query_levels=(l0, l1))
    assert MSDA.last_forward_impl() == "tile_fused"
    assert got.shape == (1, q1 - q0, M * D)
    want = c_oracle.msda_forward(value, shapes, lsi, loc_k[:, q0:q1].contiguous(), aw_k[:, q0:q1].contiguous())
    assert want.abs().max().item() > 0.5
    assert (got.cpu() - want).abs().max().item() < FP32_TOL
    sub = slice(0, q1 - q0, 97)
    want64 = c_oracle.msda_forward(value.double(), shapes, lsi, loc_k[:, q0:q1][:, sub].double().contiguous(),
                                   aw_k[:, q0:q1][:, sub].double().contiguous())
    assert (got[:, sub].cpu().double() - want64).abs().max().item() < FP32_TOL
