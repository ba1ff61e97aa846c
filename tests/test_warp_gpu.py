"""GPU parity tests of the homography warp kernel against the oracle's restatement of
kornia.warp_perspective (see oracle/torch_oracle.py for the parity status of that restatement)."""
import pytest
import torch

from conftest import load_golden, t
from helpers import smooth_features
from mvdetr_amd import geometry
from oracle import c_oracle, torch_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def warp():
    from mvdetr_amd.ops import warp_perspective
    return warp_perspective


def wildtrack_mats(aug_seed=None, geom=geometry.WILDTRACK):
    Ks, Rts = geometry.synthetic_rig(geom, seed=0)
    pm = geometry.build_proj_mats(geom, Ks, Rts)
    M = torch.eye(3).repeat(1, geom.num_cam, 1, 1) if aug_seed is None else \
        geometry.random_affine_mats(1, geom.num_cam, geom.input_img_shape, seed=aug_seed)
    return geometry.compose_frame_proj_mats(pm, M, geom.img_reduce)


def test_golden_fixture(warp):
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]), t(g["M"])
    out64 = warp(src.cuda(), M, (12, 36)).cpu()
    assert (out64 - t(g["out"])).abs().max().item() < 1e-11
    out32 = warp(src.float().cuda(), M.float(), (12, 36)).cpu()
    assert (out32.double() - t(g["out"])).abs().max().item() < 1e-4


def test_convention_pinned_to_the_reference_projection_code(warp):
    """tests/test_oracle.py::check_warp_convention on the HIP kernel (fp32 and fp64, NCHW and channel-last output)."""
    from test_oracle import check_warp_convention
    g = load_golden("warp_convention.npz")
    check_warp_convention(lambda s_, M_, d_: warp(s_.cuda(), M_, d_).cpu(), g)
    check_warp_convention(lambda s_, M_, d_: warp(s_.float().cuda(), M_.float(), d_).cpu(), g)
    check_warp_convention(lambda s_, M_, d_: warp(s_.float().cuda(), M_.float(), d_, channels_last_out=True).permute(0, 3, 1, 2).cpu(), g)


def test_nearest_mode_equals_grid_sample_nearest(warp):
    """kornia.warp_perspective(..., 'nearest') of frameDataset.py:80 (ground-plane masks)."""
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]), t(g["M"])
    want = torch_oracle.warp_perspective(src, M, (12, 36), mode="nearest")
    got = warp(src.cuda(), M, (12, 36), "nearest").cpu()
    assert torch.equal(got, want)
    got32 = warp(src.float().cuda(), M.float(), (12, 36), mode="nearest", channels_last_out=True).cpu().permute(0, 3, 1, 2)
    assert (got32.double() - want).abs().max().item() < 1e-6          # same texels (fp64 geometry), fp32 values


def _report(row):
    """Measured deviations of the fp32 kernel from the two oracles, printed and appended to
    gpurun_out/warp_parity_stats.jsonl (copied to profiles/ and quoted in DESIGN.md section 2)."""
    import json
    import os
    print("warp parity:", json.dumps(row))
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "warp_parity_stats.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass


def _against_both_oracles(out, src, M, frac_within=0.99, tag=""):
    """The kernel evaluates the geometry in fp64, so it must sit on the fp64 oracle (<= 1e-5).  The fp32
    torch op chain -- what the reference actually runs -- is itself only accurate to a few 1e-4 where
    the homography is ill-conditioned (fp32 3x3 inverse, cancellation in z near the horizon; two
    independent fp32 evaluations of the same algorithm differ by ~4e-4, and the chain's result depends on
    the host's BLAS), so agreement with it is bounded pointwise by ITS OWN distance from the fp64
    evaluation, and must be within the 1e-4 bar on (almost) all pixels."""
    ref64 = c_oracle.warp_perspective(src.double(), M.double(), (120, 360))
    err64 = (out.double() - ref64).abs().max().item()
    assert err64 < 1e-5
    ref32 = torch_oracle.warp_perspective(src, M, (120, 360))
    oracle_own = (ref32.double() - ref64).abs()
    diff = (out - ref32).abs().double()
    _report({"test": tag, "max_abs_hip_minus_fp64_oracle": err64, "max_abs_hip_minus_fp32_chain": diff.max().item(),
             "fraction_of_pixels_beyond_1e-4_vs_fp32_chain": (diff >= 1e-4).double().mean().item(),
             "max_abs_fp32_chain_minus_fp64_oracle": oracle_own.max().item(),
             "fraction_of_pixels_where_fp32_chain_is_beyond_1e-4_of_fp64": (oracle_own >= 1e-4).double().mean().item()})
    assert (diff <= 1e-4 + 1.5 * oracle_own).all()
    assert (diff < 1e-4).double().mean().item() > frac_within
    # and the C restatement of the fp32 chain lands in the same place
    c32 = c_oracle.warp_perspective(src, M, (120, 360))
    assert ((out - c32).abs().double() <= 1e-4 + 1.5 * (c32.double() - ref64).abs()).all()


@pytest.mark.parametrize("aug", [None, 1])
def test_wildtrack_white_noise(warp, aug):
    """White-noise features are the adversarial case: O(1) change per source pixel."""
    M = wildtrack_mats(aug)
    src = torch.randn(7, 128, 90, 160, generator=torch.Generator().manual_seed(0))
    out = warp(src.cuda(), M, (120, 360)).cpu()
    _against_both_oracles(out, src, M, frac_within=0.99, tag=f"white noise, NCHW, augmentation seed {aug}")
    frac_zero = (out == 0).float().mean().item()
    assert 0.1 < frac_zero < 0.6           # a good part of the plane is outside each camera's view


@pytest.mark.parametrize("aug", [None, 2])
def test_wildtrack_smooth_features(warp, aug):
    M = wildtrack_mats(aug)
    src = smooth_features(7, 128, 90, 160, seed=3)
    out = warp(src.cuda(), M, (120, 360)).cpu()
    _against_both_oracles(out, src, M, frac_within=0.999, tag=f"band-limited features, NCHW, augmentation seed {aug}")


def test_channels_last_output_equals_permuted(warp):
    M = wildtrack_mats(3)
    src = torch.randn(7, 128, 90, 160, generator=torch.Generator().manual_seed(1)).cuda()
    a = warp(src, M, (120, 360))
    b = warp(src, M, (120, 360), channels_last_out=True)          # (tiled transpose + the channel-last kernel)
    assert b.shape == (7, 120, 360, 128)
    assert (a.permute(0, 2, 3, 1) - b).abs().max().item() < 2e-6   # same weights, possibly another fma contraction
    assert torch.equal(a.permute(0, 2, 3, 1) == 0, b == 0)
    # ragged channel / pixel counts (not multiples of 64)
    src = torch.randn(3, 37, 11, 13, generator=torch.Generator().manual_seed(2)).cuda()
    Ms = wildtrack_mats(None)[:3] @ torch.diag(torch.tensor([12.0, 12.0, 1.0]))
    Ms = torch.diag(torch.tensor([0.1, 0.1, 1.0])) @ Ms
    a = warp(src, Ms, (9, 21))
    b = warp(src, Ms, (9, 21), channels_last_out=True)
    assert (a.permute(0, 2, 3, 1) - b).abs().max().item() < 2e-6   # (two template instances: fma contraction may differ)
    assert torch.equal(a.permute(0, 2, 3, 1) == 0, b == 0)
    ref = c_oracle.warp_perspective(src.cpu().double(), Ms.double(), (9, 21))
    assert (a.cpu().double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("nhwc", [False, True])
def test_backward_vs_oracle(warp, nhwc):
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]).cuda().requires_grad_(True), t(g["M"])
    out = warp(src, M, (12, 36), channels_last_out=nhwc)
    go = torch.randn(2, 8, 12, 36, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    (gs,) = torch.autograd.grad(out, src, (go.permute(0, 2, 3, 1) if nhwc else go).contiguous().cuda())
    ref = c_oracle.warp_perspective_backward(go, M, (9, 16))
    assert (gs.cpu() - ref).abs().max().item() < 1e-10


def test_gradcheck_small(warp):
    g = load_golden("warp_restatement.npz")
    src = t(g["src"])[:, :3].contiguous().cuda().requires_grad_(True)
    M = t(g["M"])
    assert torch.autograd.gradcheck(lambda s: warp(s, M, (6, 10)), (src,))


def test_degenerate_and_misuse(warp):
    src = torch.randn(1, 4, 5, 6).cuda()
    # a homography that maps everything far outside: all zeros, no fault
    far = torch.tensor([[[1.0, 0, 1e6], [0, 1, 1e6], [0, 0, 1]]])
    assert float(warp(src, far, (7, 9)).abs().max()) == 0.0
    # singular matrix: kornia would produce inf/nan grids -> zeros after the bounds test; must not fault
    sing = torch.zeros(1, 3, 3)
    out = warp(src, sing, (7, 9))
    assert out.shape == (1, 4, 7, 9) and torch.isfinite(out).all()
    # empty
    assert warp(src[:0], far[:0], (7, 9)).shape == (0, 4, 7, 9)
    with pytest.raises(NotImplementedError):
        warp(src, far, (7, 9), "bicubic")
    with pytest.raises(NotImplementedError):
        warp(src, far, (7, 9), padding_mode="border")
    with pytest.raises(ValueError):
        warp(src, far.repeat(2, 1, 1), (7, 9))


def test_identity_homography_shows_kornia_normalisation_quirk(warp):
    """With M = I and equal sizes the result is NOT the identity: kornia normalises with the
    corner-aligned map but samples with align_corners=False (x = j*w/(w-1) - 0.5)."""
    src = torch.arange(8.0).view(1, 1, 1, 8).repeat(1, 1, 3, 1).cuda()
    out = warp(src, torch.eye(3)[None], (3, 8))[0, 0, 1].cpu()
    expect = torch.tensor([j * 8 / 7 - 0.5 for j in range(8)])
    expect[0] = 0.5 * 0.0 + 0.5 * 0.0      # x = -0.5: half of pixel 0 (value 0) and half padding
    expect[7] = 0.5 * 7.0                  # x = 7.5: half of pixel 7, half padding
    assert (out - expect).abs().max().item() < 1e-5


# ---- channel-last SOURCE (what a channels_last trunk hands over), read in place ------------------------
def _count_kernel_calls(monkeypatch):
    """Records the layout flag of every launch (1 = NHWC destination, 2 = NHWC source)."""
    from mvdetr_amd.ops import warp as warp_mod
    seen = []
    real = warp_mod._launch

    def spy(name, a, M, n, c, h, w, H, W, layout, out, **kw):
        seen.append((name, layout))
        return real(name, a, M, n, c, h, w, H, W, layout, out, **kw)
    monkeypatch.setattr(warp_mod, "_launch", spy)
    return seen


@pytest.mark.parametrize("aug", [None, 5])
def test_channels_last_source_wildtrack(warp, aug, monkeypatch):
    seen = _count_kernel_calls(monkeypatch)
    M = wildtrack_mats(aug)
    src = torch.randn(7, 128, 90, 160, generator=torch.Generator().manual_seed(6))
    src_cl = src.cuda().contiguous(memory_format=torch.channels_last)
    out = warp(src_cl, M, (120, 360), channels_last_out=True)
    assert seen[-1] == ("forward", 3)                               # the in-place kernel ran, not a copy + NCHW kernel
    assert out.shape == (7, 120, 360, 128) and out.is_contiguous()
    _against_both_oracles(out.permute(0, 3, 1, 2).cpu(), src, M, frac_within=0.99, tag=f"white noise, channel-last, augmentation seed {aug}")
    plain = warp(src.cuda(), M, (120, 360), channels_last_out=True)
    assert seen[-1] == ("forward", 3)                               # NCHW source: tiled transpose + the channel-last kernel
    assert torch.equal(out, plain)
    # ... and its gradient comes back in the caller's NCHW layout, equal to the channel-last call's
    leaf = src.cuda().requires_grad_(True)
    leaf_cl = src_cl.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
    go = torch.randn(7, 120, 360, 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    (g1,) = torch.autograd.grad(warp(leaf, M, (120, 360), channels_last_out=True), leaf, go)
    (g2,) = torch.autograd.grad(warp(leaf_cl, M, (120, 360), channels_last_out=True), leaf_cl, go)
    assert g1.is_contiguous() and g1.shape == (7, 128, 90, 160)
    assert (g1 - g2).abs().max().item() <= 1e-4 * (1 + g2.abs().max().item())      # (atomics: summation order)
    # an NCHW destination of a channel-last fp32 source: read in place, written through an LDS tile as whole runs
    nchw = warp(src_cl, M, (120, 360))
    assert seen[-1] == ("forward", 2)
    ref_nchw = warp(src.cuda(), M, (120, 360))
    assert seen[-1] == ("forward", 0)                               # NCHW -> NCHW keeps the gather kernel
    assert (nchw - ref_nchw).abs().max().item() < 2e-6 and torch.equal(nchw == 0, ref_nchw == 0)


@pytest.mark.parametrize("dtype,C", [(torch.float32, 8), (torch.float32, 36), (torch.float64, 6), (torch.float32, 6)])
def test_channels_last_source_small_and_ragged(warp, dtype, C, monkeypatch):
    seen = _count_kernel_calls(monkeypatch)
    g = torch.Generator().manual_seed(C)
    src = torch.randn(3, C, 11, 13, generator=g, dtype=dtype)
    Ms = wildtrack_mats(None)[:3] @ torch.diag(torch.tensor([12.0, 12.0, 1.0]))
    Ms = (torch.diag(torch.tensor([0.1, 0.1, 1.0])) @ Ms).to(dtype)
    src_cl = src.cuda().contiguous(memory_format=torch.channels_last)
    out = warp(src_cl, Ms, (9, 21), channels_last_out=True)
    in_place = (C * src.element_size()) % 16 == 0
    assert seen[-1] == ("forward", 3 if in_place else 1)            # C*size not a multiple of 16 bytes: copy path
    ref = c_oracle.warp_perspective(src.double(), Ms.double(), (9, 21))
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert (out.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_channels_last_source_backward(warp, dtype, monkeypatch):
    seen = _count_kernel_calls(monkeypatch)
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]).to(dtype), t(g["M"])
    src_cl = src.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = warp(src_cl, M, (12, 36), channels_last_out=True)
    go = torch.randn(2, 8, 12, 36, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    (gs,) = torch.autograd.grad(out, src_cl, go.permute(0, 2, 3, 1).contiguous().to(dtype).cuda())
    assert ("forward", 3) in seen and seen[-1] == ("backward", 3)
    assert gs.shape == src.shape and gs.is_contiguous(memory_format=torch.channels_last)
    ref = c_oracle.warp_perspective_backward(go, M, (9, 16))
    assert (gs.cpu().double() - ref).abs().max().item() < (1e-10 if dtype == torch.float64 else 1e-4)


def test_channels_last_source_gradcheck(warp):
    g = load_golden("warp_restatement.npz")
    src = t(g["src"])[:, :4].contiguous().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    M = t(g["M"])
    assert torch.autograd.gradcheck(lambda s: warp(s, M, (6, 10), channels_last_out=True), (src,))


# ---- which kernel runs per layout route; the gather backward --------------------------------------------------------
def _last_kernel():
    from mvdetr_amd.ops.warp import last_kernel
    return last_kernel()


def test_kernel_name_per_layout_route(warp):
    """an experiment once became the default fp32 NCHW -> NCHW forward unnoticed (5.5x slower).  Every layout
    route is pinned to its kernel here (mvdetr_warp_last_kernel)."""
    M = wildtrack_mats(None)
    src = torch.randn(7, 128, 90, 160, generator=torch.Generator().manual_seed(0)).cuda()
    src_cl = src.contiguous(memory_format=torch.channels_last)
    warp(src, M, (120, 360))
    assert _last_kernel() == "warp_fwd_nchw_patch"                  # the literal kornia contract, mvdetr.py:194-195 (fp32)
    warp(src.double(), M, (120, 360))
    assert _last_kernel() == "warp_fwd<NCHW>"
    warp(src[:, :, :, :158], M, (120, 360))                         # rows that are not whole 16-byte pieces
    assert _last_kernel() == "warp_fwd<NCHW>"
    warp(src, M, (120, 360), channels_last_out=True)
    assert _last_kernel() == "warp_fwd_cl"                          # after the tiled transpose
    warp(src_cl, M, (120, 360), channels_last_out=True)
    assert _last_kernel() == "warp_fwd_cl"
    warp(src_cl, M, (120, 360))
    assert _last_kernel() == "warp_fwd_cl_nchw"
    warp(src[:, :37], M, (120, 360), channels_last_out=True)        # 37 channels: no 16-byte chunks
    assert _last_kernel() == "warp_fwd<NHWC>"
    for s_in, nhwc in ((src, False), (src, True), (src_cl, True), (src_cl, False)):
        leaf = s_in.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        out = warp(leaf, M, (120, 360), channels_last_out=nhwc)
        out.backward(torch.ones_like(out))
        assert _last_kernel() == "warp_bwd_gather[planned]"         # autograd: plan kept per matrix tensor (ops/warp.py)
    leaf = src[:, :37].detach().clone().requires_grad_(True)
    warp(leaf, M, (120, 360)).sum().backward()
    assert _last_kernel() == "warp_bwd<NCHW>"


def _bwd_cl(go_nhwc, M, n, c, h, w, env=None, nearest=False):
    """The channel-last backward entry through the C ABI; `env` = extra environment for this call."""
    import os
    from mvdetr_amd.ops import warp as warp_mod
    H, W = go_nhwc.shape[1:3]
    gs = torch.full((n, h, w, c), float("nan"), dtype=go_nhwc.dtype, device="cuda")     # must be overwritten
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        warp_mod._launch("backward", go_nhwc, M.to(device="cuda", dtype=go_nhwc.dtype).contiguous(), n, c, h, w, H, W, 3 | (4 if nearest else 0), gs)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k)
            else:
                os.environ[k] = v
    return gs


@pytest.mark.parametrize("aug", [None, 4])
def test_backward_gather_wildtrack_size(warp, aug):
    """Gather backward at full Wildtrack size: vs the fp32 C oracle (scatter order differs: 1e-4 relative to the largest
    sum), vs the library's atomic scatter kernel, deterministic bit for bit, same result through the clipping geometry."""
    M = wildtrack_mats(aug)
    go = torch.randn(7, 120, 360, 128, generator=torch.Generator().manual_seed(2)).cuda()
    g1 = _bwd_cl(go, M, 7, 128, 90, 160)
    assert _last_kernel() == "warp_bwd_gather"
    assert torch.isfinite(g1).all()
    g2 = _bwd_cl(go, M, 7, 128, 90, 160)
    assert torch.equal(g1, g2)                                       # deterministic
    g3 = _bwd_cl(go, M, 7, 128, 90, 160, env={"MVDETR_WARP_BWD_GEOMETRY": "clip"})
    assert (g1 - g3).abs().max().item() <= 1e-5 * (1 + g1.abs().max().item())    # other candidate scans: another fixed order
    assert torch.equal(g1 == 0, g3 == 0)
    for heavy_above in ("0", "1000000000"):                          # every block on the workgroup path / on the wave path
        g4 = _bwd_cl(go, M, 7, 128, 90, 160, env={"MVDETR_WARP_BWD_HEAVY": heavy_above})
        assert (g1 - g4).abs().max().item() <= 1e-5 * (1 + g1.abs().max().item()) and torch.equal(g1 == 0, g4 == 0)
    gsc = _bwd_cl(go, M, 7, 128, 90, 160, env={"MVDETR_WARP_BWD_IMPL": "scatter"})
    assert _last_kernel() == "warp_bwd_cl"
    scale = 1 + gsc.abs().max().item()
    assert (g1 - gsc).abs().max().item() <= 2e-5 * scale
    assert torch.equal(g1 == 0, gsc == 0)                            # the same texels receive gradient
    ref = c_oracle.warp_perspective_backward(go.permute(0, 3, 1, 2).cpu().double(), M.float().double(), (90, 160))
    assert (g1.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() <= 2e-5 * scale
    # adjoint identity <go, warp(x)> == <bwd(go), x>
    x = torch.randn(7, 128, 90, 160, generator=torch.Generator().manual_seed(5)).cuda().contiguous(memory_format=torch.channels_last)
    y = warp(x, M, (120, 360), channels_last_out=True)
    lhs, rhs = (go.double() * y.double()).sum().item(), (g1.double() * x.permute(0, 2, 3, 1).double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0) + 1e-3


def _horizon_mats(dtype=torch.float64):
    """Homographies whose line at infinity crosses the SOURCE image (a horizon inside the view) and whose pre-image of the
    line at infinity crosses the destination grid: the gather's clipping path, both signs of the homogeneous coordinate."""
    Ms = []
    for k, (a, b, c) in enumerate([(0.0, 0.11, -0.5), (0.03, -0.09, 0.4), (-0.05, 0.0, 0.6), (0.02, 0.13, -1.0)]):
        A = torch.tensor([[2.1, 0.3 * k, 1.0], [-0.2, 2.4, 2.0 - k], [a, b, c]], dtype=torch.float64)
        Ms.append(A)
    return torch.stack(Ms).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("geometry_env", [None, "clip"])
@pytest.mark.parametrize("nearest", [False, True])
def test_backward_gather_horizon_inside_the_view(warp, dtype, geometry_env, nearest):
    M = _horizon_mats()
    n, c, h, w, H, W = 4, 8, 9, 16, 14, 33
    env = {"MVDETR_WARP_BWD_GEOMETRY": geometry_env} if geometry_env else None
    go = torch.randn(n, H, W, c, generator=torch.Generator().manual_seed(7), dtype=torch.float64)
    gs = _bwd_cl(go.to(dtype).cuda(), M.to(dtype), n, c, h, w, env=env, nearest=nearest)
    assert _last_kernel() == "warp_bwd_gather"
    # reference: autograd through the oracle's own forward (torch ops, fp64) on the matrices as the kernel sees them
    Mk = M.to(dtype).double()
    x = torch.zeros(n, c, h, w, dtype=torch.float64, requires_grad=True)
    y = torch_oracle.warp_perspective(x, Mk, (H, W), mode="nearest" if nearest else "bilinear")
    (ref,) = torch.autograd.grad(y, x, go.permute(0, 3, 1, 2))
    assert ref.abs().max().item() > 0.5
    tol = 1e-10 if dtype == torch.float64 else 1e-4
    assert (gs.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() < tol * (1 + ref.abs().max().item())


def test_backward_gather_pixels_kornia_does_not_divide(warp):
    """|z| <= 1e-8: kornia's convert_points_from_homogeneous leaves the point undivided, so that destination column samples a
    position unrelated to the projective map.  The gather's scans skip those pixels; warp_bwd_scans lists them and the
    gather takes the listed ones as extra candidates."""
    A = torch.tensor([[1.0, 0, 0], [0, 1, 0], [1, 0, -4]], dtype=torch.float64)        # M^-1: z = j - 4
    M = torch.linalg.inv(A)[None]
    n, c, h, w, H, W = 1, 4, 12, 16, 10, 9
    src = torch.randn(n, c, h, w, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    fwd = warp(src.cuda().contiguous(memory_format=torch.channels_last), M, (H, W), channels_last_out=True)
    want = c_oracle.warp_perspective(src, M, (H, W))
    assert (fwd.permute(0, 3, 1, 2).cpu() - want).abs().max().item() < 1e-12
    assert want[..., 4].abs().max().item() > 0.1                     # the undivided column does sample the image
    go = torch.randn(n, H, W, c, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    gs = _bwd_cl(go.cuda(), M, n, c, h, w)
    assert _last_kernel() == "warp_bwd_gather"
    ref = c_oracle.warp_perspective_backward(go.permute(0, 3, 1, 2).contiguous(), M, (h, w))
    assert (gs.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() < 1e-12
    only_col = torch.zeros_like(go)
    only_col[:, :, 4] = go[:, :, 4]
    assert _bwd_cl(only_col.cuda(), M, n, c, h, w).abs().max().item() > 0.1            # the listed pixels carry gradient
    assert torch.equal(_bwd_cl(go.cuda(), M, n, c, h, w), gs)                          # ... deterministically
    gh = _bwd_cl(go.cuda(), M, n, c, h, w, env={"MVDETR_WARP_BWD_HEAVY": "0"})         # and through the workgroup path
    assert (gh.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() < 1e-12


@pytest.mark.parametrize("C,dtype", [(4, torch.float32), (8, torch.float64), (36, torch.float32), (256, torch.float32),
                                     (320, torch.float32), (2, torch.float64)])
def test_backward_gather_channel_counts(warp, C, dtype):
    """Lane groups of 1 ... 64 lanes per 2x2 texel block, more than 64 chunks (channel groups), odd source sizes."""
    n, h, w, H, W = 3, 11, 13, 9, 21
    Ms = wildtrack_mats(None)[:3] @ torch.diag(torch.tensor([12.0, 12.0, 1.0]))
    Ms = (torch.diag(torch.tensor([0.1, 0.1, 1.0])) @ Ms).double()
    go = torch.randn(n, H, W, C, generator=torch.Generator().manual_seed(C), dtype=torch.float64)
    ref = c_oracle.warp_perspective_backward(go.permute(0, 3, 1, 2).contiguous(), Ms.to(dtype).double(), (h, w))
    assert ref.abs().max().item() > 0.5
    tol = 1e-11 if dtype == torch.float64 else 2e-5
    for env in (None, {"MVDETR_WARP_BWD_HEAVY": "0"}):
        gs = _bwd_cl(go.to(dtype).cuda(), Ms, n, C, h, w, env=env)
        assert _last_kernel() == "warp_bwd_gather"
        assert (gs.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() < tol * (1 + ref.abs().max().item())


def test_stress16_size_forward_and_adjoint(warp):
    """BASELINE configs[4] at full size: 16 cameras x 256 channels x 135 x 240 -> 120 x 360, both source layouts, against the
    fp64 C oracle (forward) and through the adjoint identity + a camera subset against the oracle (backward)."""
    geom = geometry.GEOMETRIES["stress16"]
    M = wildtrack_mats(None, geom)
    h, w = geom.Rimg_shape
    H, W = geom.Rworld_shape
    C = geom.feat_channels
    assert (geom.num_cam, C, h, w, H, W) == (16, 256, 135, 240, 120, 360)
    src = torch.randn(16, C, h, w, generator=torch.Generator().manual_seed(0))
    ref = c_oracle.warp_perspective(src.double(), M.float().double(), (H, W))
    a = warp(src.cuda(), M, (H, W))
    assert _last_kernel() == "warp_fwd_nchw_patch"
    assert (a.cpu().double() - ref).abs().max().item() < 1e-5
    src_cl = src.cuda().contiguous(memory_format=torch.channels_last)
    b = warp(src_cl, M, (H, W), channels_last_out=True)
    assert _last_kernel() == "warp_fwd_cl"
    assert (b.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item() < 1e-5
    del a, ref
    go = torch.randn(16, H, W, C, generator=torch.Generator().manual_seed(1)).cuda()
    for leaf in (src.cuda().requires_grad_(True), src_cl.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)):
        out = warp(leaf, M, (H, W), channels_last_out=True)
        (gs,) = torch.autograd.grad(out, leaf, go)
        assert _last_kernel() == "warp_bwd_gather[planned]"
        lhs = (go.double() * out.detach().double()).sum().item()
        rhs = (gs.double() * leaf.detach().double()).sum().item()
        assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0) + 1e-2
    cams = [0, 7, 15]
    refb = c_oracle.warp_perspective_backward(go[cams].permute(0, 3, 1, 2).cpu().double().contiguous(), M[cams].float().double(), (h, w))
    assert (gs[cams].cpu().double() - refb).abs().max().item() <= 2e-5 * (1 + refb.abs().max().item())


def _time_us(fn, iters=10, trials=5):
    """Best of `trials` averages over `iters` launches: a tripwire must not trip on a noisy neighbour or on allocator work
    left over from earlier tests."""
    for _ in range(3):
        fn()
    best = float("inf")
    for _ in range(trials):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best


def test_perf_guard_every_route_within_4x_of_a_copy(warp):
    """Not a benchmark: a tripwire.  Each forward route at Wildtrack size must stay within 4x (+ 60 us of slack) of a device copy
    of the same bytes (r02's accidental default took 16x; the slowest route takes 3x), the gather backward within 6x (it takes 2.5x).
    Best-of-5 timings after emptying the caching allocator: inside the whole suite a single average once tripped on noise."""
    torch.cuda.empty_cache()
    M = wildtrack_mats(None).cuda()
    src = torch.randn(7, 128, 90, 160, device="cuda")
    src_cl = src.contiguous(memory_format=torch.channels_last)
    nbytes = 4 * 7 * 128 * (90 * 160 + 120 * 360)
    x = torch.empty(nbytes // 8, device="cuda")
    y = torch.empty_like(x)
    copy_us = _time_us(lambda: y.copy_(x))
    routes = {"NCHW->NCHW": lambda: warp(src, M, (120, 360)),
              "NCHW->NHWC": lambda: warp(src, M, (120, 360), channels_last_out=True),
              "NHWC->NHWC": lambda: warp(src_cl, M, (120, 360), channels_last_out=True),
              "NHWC->NCHW": lambda: warp(src_cl, M, (120, 360))}
    for name, fn in routes.items():
        us = _time_us(fn)
        assert us <= 4.0 * copy_us + 60.0, f"{name}: {us:.0f} us against a {copy_us:.0f} us copy"
    go = torch.randn(7, 120, 360, 128, device="cuda")
    us = _time_us(lambda: _bwd_cl(go, M, 7, 128, 90, 160))
    assert us <= 6.0 * copy_us + 100.0, f"gather backward: {us:.0f} us against a {copy_us:.0f} us copy"


def test_planned_backward_equals_the_one_call_entry_and_the_plan_is_reused(warp):
    """ABI 10: mvdetr_warp_backward_plan_* + mvdetr_warp_perspective_backward_planned_* (what autograd calls) give the
    bits of mvdetr_warp_perspective_backward_*; the plan is built once per matrix tensor and survives until the matrices
    change (in-place write -> version counter -> a new plan)."""
    from mvdetr_amd.ops import warp as warp_mod
    M = wildtrack_mats(None).float().cuda()
    src = torch.randn(7, 128, 90, 160, generator=torch.Generator().manual_seed(5)).cuda().contiguous(memory_format=torch.channels_last)
    go = torch.randn(7, 120, 360, 128, generator=torch.Generator().manual_seed(6)).cuda()
    plain = _bwd_cl(go, M, 7, 128, 90, 160)
    assert _last_kernel() == "warp_bwd_gather"
    warp_mod._plans.clear()
    grads = []
    for _ in range(3):
        leaf = src.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
        warp(leaf, M, (120, 360), channels_last_out=True).backward(go)
        assert _last_kernel() == "warp_bwd_gather[planned]"
        grads.append(leaf.grad.permute(0, 2, 3, 1))
    assert len(warp_mod._plans.entries) == 1                          # one plan for the three calls
    for g in grads:
        assert torch.equal(g, plain)
    plan0 = warp_mod._plans.entries[0][3]
    M.mul_(1.0)                                                        # same values, new version: the plan is rebuilt
    leaf = src.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
    warp(leaf, M, (120, 360), channels_last_out=True).backward(go)
    assert len(warp_mod._plans.entries) == 2 and warp_mod._plans.entries[0][3] is not plan0
    assert torch.equal(leaf.grad.permute(0, 2, 3, 1), plain)
    # other matrices through the cache: their own gradient, not the cached one's
    M2 = wildtrack_mats(4).float().cuda()
    leaf = src.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
    warp(leaf, M2, (120, 360), channels_last_out=True).backward(go)
    assert torch.equal(leaf.grad.permute(0, 2, 3, 1), _bwd_cl(go, M2, 7, 128, 90, 160))


def test_tagged_backward_reuses_the_librarys_plan_and_forgets_it_when_it_must(warp):
    """ABI 12: mvdetr_warp_perspective_backward_tagged_* -- the one-call entry with a caller-supplied version tag of the
    matrices.  Same tag: the geometry in the per-(device, stream) scratch is reused (the gather alone is launched); another tag,
    other shapes or an untagged call in between: it is rebuilt.  Every result equals the untagged entry bit for bit."""
    from mvdetr_amd.ops import warp as warp_mod
    n, c, h, w, H, W = 7, 128, 90, 160, 120, 360
    M1 = wildtrack_mats(None).float().cuda().contiguous()
    M2 = wildtrack_mats(4).float().cuda().contiguous()
    go = torch.randn(n, H, W, c, generator=torch.Generator().manual_seed(8)).cuda()

    def run(M, tag):
        gs = torch.empty(n, h, w, c, device="cuda")
        warp_mod._launch("backward", go, M, n, c, h, w, H, W, 3, gs, tag=tag)
        return gs
    want1, want2 = run(M1, 0), run(M2, 0)
    assert torch.equal(run(M1, 5), want1)                      # builds the plan, remembers tag 5
    assert torch.equal(run(M1, 5), want1)                      # reuses it
    assert torch.equal(run(M1, 5), want1)
    assert torch.equal(run(M2, 6), want2)                      # another tag: rebuilt for the other matrices
    assert torch.equal(run(M2, 6), want2)
    assert torch.equal(run(M1, 0), want1)                      # an untagged call overwrites the scratch's plan ...
    assert torch.equal(run(M2, 6), want2)                      # ... so tag 6 must not be trusted any more
    # other shapes under the same tag: a different key, rebuilt
    go2 = torch.randn(n, 60, 180, c, generator=torch.Generator().manual_seed(9)).cuda()
    gs_a, gs_b = torch.empty(n, h, w, c, device="cuda"), torch.empty(n, h, w, c, device="cuda")
    warp_mod._launch("backward", go2, M2, n, c, h, w, 60, 180, 3, gs_a, tag=6)
    warp_mod._launch("backward", go2, M2, n, c, h, w, 60, 180, 3, gs_b)
    assert torch.equal(gs_a, gs_b)
    # and it is faster than rebuilding: the scans kernel is gone from the call (loose bound: any gain at all)
    t_tag = _time_us(lambda: run(M1, 77))
    t_plain = _time_us(lambda: run(M1, 0))
    assert t_tag < t_plain, (t_tag, t_plain)


def test_backward_cost_is_bounded_with_the_horizon_in_view(warp):
    """a source square just below the horizon has a far-end image millions of pixels long; the scans' fast path
    clamped its lines to the grid but not their length, so one block could walk up to 2e9 candidates (seconds).  Costs above
    H*W now go to the clipping path.  Guard: horizon-in-view matrices at Wildtrack size take no longer than a few times the
    ordinary geometry."""
    import time
    n, c, h, w, H, W = 4, 128, 90, 160, 120, 360
    go = torch.randn(n, H, W, c, generator=torch.Generator().manual_seed(3)).cuda()
    # the horizon (the pre-image of the line at infinity) runs through the source image a little above its lower edge
    Ms = []
    for k in range(n):
        A = torch.tensor([[2.0, 0.1 * k, 5.0], [0.05, 2.2, -3.0], [1e-4 * k, 0.012 + 0.001 * k, -1.0]], dtype=torch.float64)
        Ms.append(A)
    Mh = torch.stack(Ms)
    ordinary = wildtrack_mats(None)[:n]

    def run(M):
        _bwd_cl(go, M.float(), n, c, h, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g = _bwd_cl(go, M.float(), n, c, h, w)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3, g

    t_ord, _ = run(ordinary)
    t_hor, g = run(Mh)
    assert torch.isfinite(g).all()
    assert t_hor < max(20 * t_ord, 5e-3), (t_hor, t_ord)
    # and it is still the right gradient: adjoint identity against the forward on the same matrices
    x = torch.randn(n, h, w, c, generator=torch.Generator().manual_seed(4)).cuda()
    y = warp(x.permute(0, 3, 1, 2), Mh, (H, W), channels_last_out=True)
    lhs, rhs = (y.double() * go.double()).sum().item(), (x.double() * g.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), 1.0) + 1e-2


def test_release_scratch_between_backward_calls():
    """mvdetr_warp_release_scratch: the gather backward keeps its geometry scratch per (device, stream); dropping it
    between calls changes nothing in the next call's result, on the default stream and on a side stream."""
    from mvdetr_amd.ops import warp as warp_mod
    torch.manual_seed(0)
    N, C, h, w, H, W = 2, 32, 24, 40, 30, 50
    Mx = _mats(N, h, w, H, W, seed=5) if "_mats" in globals() else None
    if Mx is None:
        Mx = torch.eye(3).repeat(N, 1, 1)
        Mx[:, 0, 0], Mx[:, 1, 1], Mx[:, 0, 2], Mx[:, 1, 2] = 1.2, 1.1, 2.0, -1.5
    Mx = Mx.cuda().float().contiguous()
    go = torch.randn(N, H, W, C, device="cuda")

    def run():
        gs = torch.empty(N, h, w, C, device="cuda")
        warp_mod._launch("backward", go, Mx, N, C, h, w, H, W, 3, gs)
        return gs

    a = run()
    torch.cuda.synchronize()
    warp_mod.release_scratch()
    b = run()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = run()
    side.synchronize()
    torch.cuda.synchronize()
    warp_mod.release_scratch()
    assert torch.equal(a, b) and torch.equal(a, c)
    assert a.abs().max().item() > 0
