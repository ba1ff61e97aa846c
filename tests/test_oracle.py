"""The oracle against the reference's golden vectors (CPU only).

These tests pin oracle/torch_oracle.py and oracle/oracle.c to outputs produced by the reference's
own Python code (tests/golden/make_golden.py).  Everything else in the suite trusts the oracle
only because these pass.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, t
from oracle import c_oracle, torch_oracle


# ---- (1) ops/test.py shapes -----------------------------------------------------------------------
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-15), ("f32", torch.float32, 1e-9)])
def test_msda_testpy_shapes(tag, dtype, tol):
    g = load_golden(f"msda_testpy_{tag}.npz")
    args = (t(g["value"]), t(g["shapes"]), t(g["loc"]), t(g["aw"]))
    out_t = torch_oracle.msda_core(*args)
    assert out_t.dtype == dtype
    assert torch.equal(out_t, t(g["out"]))           # same torch ops in the same order: bit-exact
    out_c = c_oracle.msda_forward(args[0], args[1], t(g["level_start_index"]), args[2], args[3])
    assert (out_c - t(g["out"])).abs().max().item() <= tol


# ---- (2) MVDeTr-mini forward, (4) backward --------------------------------------------------------
def test_msda_mini_forward_fp64_and_fp32():
    g = load_golden("msda_mini.npz")
    v, s, lsi, loc, aw = t(g["value"]), t(g["shapes"]), t(g["level_start_index"]), t(g["loc"]), t(g["aw"])
    out64 = torch_oracle.msda_core(v.double(), s, loc.double(), aw.double())
    assert torch.equal(out64, t(g["out"]))
    assert torch.equal(torch_oracle.msda_core(v, s, loc, aw), t(g["out_f32"]))
    c64 = c_oracle.msda_forward(v.double(), s, lsi, loc.double(), aw.double())
    assert (c64 - t(g["out"])).abs().max().item() < 1e-13
    c32 = c_oracle.msda_forward(v, s, lsi, loc, aw)
    # the C restatement runs grid_sample's fp32 arithmetic; it lands on the fp32 golden to rounding
    assert (c32 - t(g["out_f32"])).abs().max().item() < 2e-6
    assert (c32.double() - t(g["out"])).abs().max().item() < 1e-5


def test_msda_mini_backward():
    g = load_golden("msda_mini.npz")
    v, s, lsi = t(g["value"]).double(), t(g["shapes"]), t(g["level_start_index"])
    loc, aw, go = t(g["loc"]).double(), t(g["aw"]).double(), t(g["grad_out"]).double()
    v.requires_grad_(True), loc.requires_grad_(True), aw.requires_grad_(True)
    out = torch_oracle.msda_core(v, s, loc, aw)
    gv, gl, ga = torch.autograd.grad(out, (v, loc, aw), go)
    for mine, name in ((gv, "grad_value"), (gl, "grad_loc"), (ga, "grad_aw")):
        assert torch.equal(mine, t(g[name])), name
    cv, cl, ca = c_oracle.msda_backward(v.detach(), s, lsi, loc.detach(), aw.detach(), go)
    assert (cv - t(g["grad_value"])).abs().max().item() < 1e-12
    assert (cl - t(g["grad_loc"])).abs().max().item() < 1e-11
    assert (ca - t(g["grad_aw"])).abs().max().item() < 1e-12
    # fp32 entry point against the fp64 truth
    fv, fl, fa = c_oracle.msda_backward(v.detach().float(), s, lsi, loc.detach().float(),
                                        aw.detach().float(), go.float())
    assert (fv.double() - t(g["grad_value"])).abs().max().item() < 1e-4
    assert (fl.double() - t(g["grad_loc"])).abs().max().item() < 2e-3   # multiplied by W,H and D-summed
    assert (fa.double() - t(g["grad_aw"])).abs().max().item() < 1e-4


# ---- (3) borders ----------------------------------------------------------------------------------
def test_msda_edges():
    g = load_golden("msda_edges.npz")
    v, s, lsi, loc, aw = (t(g[k]) for k in ("value", "shapes", "level_start_index", "loc", "aw"))
    assert torch.equal(torch_oracle.msda_core(v, s, loc, aw), t(g["out"]))
    assert (c_oracle.msda_forward(v, s, lsi, loc, aw) - t(g["out"])).abs().max().item() < 1e-13


# ---- (5) module arithmetic --------------------------------------------------------------------------
def test_msda_module_restatement():
    g = load_golden("msda_module.npz")
    d_model, L, M, P = (int(x) for x in g["dims"])
    params = {k[2:]: t(v) for k, v in g.items() if k.startswith("p.")}
    out, loc, aw, value = torch_oracle.msda_module(params, t(g["query"]), t(g["ref"]), t(g["src"]),
                                                   t(g["shapes"]), M, P, return_intermediates=True)
    assert torch.equal(loc, t(g["loc"]))
    assert torch.equal(aw, t(g["aw"]))
    assert torch.equal(value, t(g["value"]))
    assert torch.equal(out, t(g["out"]))


# ---- (6) position embedding ------------------------------------------------------------------------
def test_product_pos_embedding_equals_the_reference_bit_for_bit():
    """mvdetr_amd.world_feat.create_pos_embedding (written from the formula) against the reference's output."""
    from mvdetr_amd.world_feat import create_pos_embedding
    g = load_golden("pos_embedding.npz")
    assert torch.equal(create_pos_embedding((6, 9), 8), t(g["small"]))
    assert torch.equal(create_pos_embedding((60, 180), 64), torch_oracle.create_pos_embedding((60, 180), 64))
    assert create_pos_embedding((5, 7), 6, normalize=False).shape == (1, 12, 5, 7)
    with pytest.raises(ValueError):
        create_pos_embedding((5, 7), 6, normalize=False, scale=1.0)


def test_pos_embedding():
    g = load_golden("pos_embedding.npz")
    assert torch.equal(torch_oracle.create_pos_embedding((6, 9), 8), t(g["small"]))
    big = torch_oracle.create_pos_embedding((60, 180), 64)
    assert list(big.shape) == list(g["big_shape"])
    assert torch.equal(big[0, :, ::20, ::45], t(g["big_rows"]))
    assert abs(float(big.double().sum()) - float(g["big_sum"])) < 1e-9
    assert abs(float(big.double().abs().sum()) - float(g["big_abs_sum"])) < 1e-9


# ---- (7) DeformTransWorldFeat mini -----------------------------------------------------------------
def test_world_feat_mini():
    g = load_golden("world_feat_mini.npz")
    num_cam, H, W, base_dim, hidden, nhead, P = (int(x) for x in g["dims"])
    params = {k[2:]: t(v) for k, v in g.items() if k.startswith("p.")}
    out = torch_oracle.deform_trans_world_feat(params, t(g["x"]), t(g["ref"]), n_heads=nhead, n_points=P)
    assert out.shape == t(g["out"]).shape
    assert (out - t(g["out"])).abs().max().item() < 1e-5


def test_conv_world_feat_mini():
    """BASELINE config 0's world-feature block: oracle restatement AND the product module (plain torch, runs on the
    CPU) against the reference's own output (conv_world_feat.py:21-52)."""
    from mvdetr_amd.world_feat import ConvWorldFeat
    g = load_golden("conv_world_feat_mini.npz")
    num_cam, H, W, base_dim = (int(x) for x in g["dims"])
    params = {k[2:]: t(v) for k, v in g.items() if k.startswith("p.")}
    want = t(g["out"])
    out = torch_oracle.conv_world_feat(params, t(g["x"]))
    assert out.shape == want.shape and (out - want).abs().max().item() < 1e-6
    mod = ConvWorldFeat(num_cam, (H, W), base_dim, hidden_dim=base_dim).eval()
    missing, unexpected = mod.load_state_dict(params, strict=True)        # same parameter names as the reference
    with torch.no_grad():
        got = mod(t(g["x"]))
    assert (got - want).abs().max().item() < 1e-6
    with pytest.raises(ValueError):
        ConvWorldFeat(num_cam, (H, W), base_dim, hidden_dim=base_dim * 2)


# ---- (8) warp: both restatements against the golden's INDEPENDENT fp64 closed form (tests/golden/make_golden.py,
# numpy only); none of it is pinned against kornia itself, which is absent -- see test_warp_convention_* for what the
# reference's own code does pin ------------------------------------------------------------------------------------
def test_warp_restatements_vs_independent_closed_form():
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]), t(g["M"])
    out64 = torch_oracle.warp_perspective(src, M, (12, 36))
    assert (out64 - t(g["out"])).abs().max().item() < 1e-12
    c64 = c_oracle.warp_perspective(src, M, (12, 36))
    assert (c64 - out64).abs().max().item() < 1e-10
    c32 = c_oracle.warp_perspective(src.float(), M.float(), (12, 36))
    assert (c32.double() - out64).abs().max().item() < 1e-4
    nz = (out64 != 0).double().mean().item()
    assert 0.05 < nz < 1.0        # the fixture exercises both in-view and out-of-view pixels


def test_warp_net_effect_formula():
    """The closed form quoted in SURVEY 8(a1): output (i,j) samples the source at
    p = M^-1 (j,i,1), x = p_x * w/(w-1) - 0.5, y = p_y * h/(h-1) - 0.5."""
    g = load_golden("warp_restatement.npz")
    M = t(g["M"])
    grid = torch_oracle.warp_grid(M, (9, 16), (12, 36))                  # normalised
    x = ((grid[..., 0] + 1) * 16 - 1) / 2
    y = ((grid[..., 1] + 1) * 9 - 1) / 2
    jj, ii = torch.meshgrid(torch.arange(36.0, dtype=torch.float64), torch.arange(12.0, dtype=torch.float64),
                            indexing="xy")
    pts = torch.stack([jj, ii, torch.ones_like(jj)], -1)
    p = torch.einsum("hwk,njk->nhwj", pts, torch.inverse(M))
    px, py = p[..., 0] / p[..., 2], p[..., 1] / p[..., 2]
    assert (x - (px * 16 / 15 - 0.5)).abs().max().item() < 1e-9
    assert (y - (py * 9 / 8 - 0.5)).abs().max().item() < 1e-9


def _blob_centroids(warp_fn, g):
    """Warp one Gaussian blob per pinned feature pixel and return the intensity centroid of its image, per point."""
    N, h, w, H, W = (int(x) for x in g["dims"])
    uv, M = g["src_uv"], t(g["M"])
    K = uv.shape[1]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64), torch.arange(w, dtype=torch.float64), indexing="ij")
    src = torch.stack([torch.stack([torch.exp(-((xs - uv[n, k, 0]) ** 2 + (ys - uv[n, k, 1]) ** 2) / (2 * 1.2 ** 2))
                                    for k in range(K)]) for n in range(N)])            # [N, K, h, w]: channel k = blob k
    out = warp_fn(src, M, (H, W)).double()
    Y, X = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    mass = out.sum((-1, -2))
    assert (mass > 0.5).all()
    return torch.stack([(out * X).sum((-1, -2)) / mass, (out * Y).sum((-1, -2)) / mass], -1)      # [N, K, 2] (x, y)


def check_warp_convention(warp_fn, g):
    """A feature-pixel blob must come out on the world grid where the REFERENCE's projection code puts that pixel
    (mvdetr.py:82-95,155-161 + utils/projection.py:4-14, recorded in warp_convention.npz): pins the direction of the
    homography (dst <- src), the (x, y) order and the composition.  Tolerance: the blob is stretched by the ground-plane
    magnification (its centroid moves by a pixel or two) and kornia's size/(size-1) quirk shifts it by < 1 px; a wrong
    convention is off by tens of pixels (asserted too)."""
    want = t(g["dst_xy"])
    got = _blob_centroids(warp_fn, g)
    dist = (got - want).norm(dim=-1)
    assert dist.max().item() < 3.0, dist
    # the transposed and the inverse convention would not pass
    assert (got.flip(-1) - want).norm(dim=-1).min().item() > 10
    return dist


def test_warp_convention_pinned_to_the_reference_projection_code():
    g = load_golden("warp_convention.npz")
    check_warp_convention(lambda s_, M_, d_: torch_oracle.warp_perspective(s_, M_, d_), g)
    check_warp_convention(lambda s_, M_, d_: c_oracle.warp_perspective(s_, M_, d_), g)


def test_warp_backward_c_vs_autograd():
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]).clone().requires_grad_(True), t(g["M"])
    out = torch_oracle.warp_perspective(src, M, (12, 36))
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (gs,) = torch.autograd.grad(out, src, go)
    cs = c_oracle.warp_perspective_backward(go, M, (9, 16))
    assert (cs - gs).abs().max().item() < 1e-10


# ---- property: the C forward is linear in value and in the weights ---------------------------------
def test_msda_linearity():
    g = load_golden("msda_mini.npz")
    v, s, lsi, loc, aw = (t(g[k]) for k in ("value", "shapes", "level_start_index", "loc", "aw"))
    v, loc, aw = v.double(), loc.double(), aw.double()
    v2 = torch.randn(v.shape, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
    a = c_oracle.msda_forward(v, s, lsi, loc, aw)
    b = c_oracle.msda_forward(v2, s, lsi, loc, aw)
    ab = c_oracle.msda_forward(2 * v - 3 * v2, s, lsi, loc, 0.5 * aw)
    assert (ab - 0.5 * (2 * a - 3 * b)).abs().max().item() < 1e-12
