"""Every kernel route that is selected by an environment variable, run through the C ABI and checked against the oracle.

The library reads its `MVDETR_*` variables once per process, so every setting runs in a CHILD process (this file as a
script: `python test_knob_routes_gpu.py <bwd|fwd|warp> <out.pt>`, all cases of a family in one child) and the parent compares what the child stored with the fp64 C
oracle (oracle/oracle.c) -- the kernels a knob selects ship in the library and are tested like the default ones.  This is
the reference's own strategy for its backward variants: ops/test.py:63-86 runs gradcheck over every channel count, i.e.
over every one of the col2im kernels cuh:956-1327 dispatches to.

Routes (knob table): MVDETR_MSDA_BWD_IMPL = twopass (default) | split | onepass | atomic,
MVDETR_MSDA_BWD_ORDER = spread, MVDETR_MSDA_GROUP = 0, MVDETR_MSDA_FWD_IMPL = gather | tile, MVDETR_WARP_FWD_NCHW = gather.
(MVDETR_MSDA_WINDOW_SHIFT = 0 and the MVDETR_WARP_BWD_* knobs have their tests in test_msda_gpu.py / test_warp_gpu.py.)
"""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

BWD_ROUTES = [{}, {"MVDETR_MSDA_BWD_IMPL": "split"}, {"MVDETR_MSDA_BWD_IMPL": "twopass"}, {"MVDETR_MSDA_BWD_IMPL": "onepass"},
              {"MVDETR_MSDA_BWD_IMPL": "atomic"}, {"MVDETR_MSDA_BWD_IMPL": "split", "MVDETR_MSDA_BWD_ORDER": "spread"},
              {"MVDETR_MSDA_BWD_IMPL": "onepass", "MVDETR_MSDA_BWD_ORDER": "spread"}]
FWD_ROUTES = [{}, {"MVDETR_MSDA_GROUP": "0"}, {"MVDETR_MSDA_FWD_IMPL": "gather"}, {"MVDETR_MSDA_FWD_IMPL": "tile"}]


def _route_id(env):
    return ",".join(f"{k[7:].lower()}={v}" for k, v in env.items()) or "default"


# ---- the cases, built identically in parent and child (seeded) --------------------------------------------------------
def _bwd_case(name):
    sys.path.insert(0, HERE)
    from helpers import encoder_msda_inputs, random_msda_inputs
    if name == "bwd_encoder":            # MVDeTr's shape in small: 7 cameras, 16-channel heads, taps near their cells
        value, shapes, lsi, loc, aw = encoder_msda_inputs(7, 19, 37, M=8, D=16, seed=31, noise_px=1.5)
    elif name == "bwd_encoder_b2":       # two frames, six cameras (MultiviewX), a map that is not a multiple of the tiles
        value, shapes, lsi, loc, aw = encoder_msda_inputs(6, 13, 29, M=8, D=16, B=2, seed=32, noise_px=1.0)
    elif name == "bwd_encoder_d32":      # 32-channel heads (the one-pass kernels do not take them: every route must still answer)
        value, shapes, lsi, loc, aw = encoder_msda_inputs(5, 12, 20, M=4, D=32, seed=33, noise_px=1.5)
    elif name == "bwd_far":              # a fifth of the taps anywhere in (and outside) the map
        value, shapes, lsi, loc, aw = encoder_msda_inputs(7, 19, 37, M=8, D=16, seed=34, noise_px=1.0)
        g = torch.Generator().manual_seed(35)
        far = torch.rand(loc.shape[:-1], generator=g) < 0.2
        loc = torch.where(far[..., None], torch.rand(loc.shape, generator=g) * 1.4 - 0.2, loc)
    elif name == "bwd_uniform":          # locations anywhere: the stand-down paths
        value, shapes, lsi, loc, aw = random_msda_inputs(1, [(12, 20)] * 7, 8, 16, 7 * 12 * 20, 4, seed=36)
    elif name == "bwd_wildtrack":        # full Wildtrack size (checked through adjoint identities + a strided oracle subset)
        value, shapes, lsi, loc, aw = encoder_msda_inputs(7, 60, 180, M=8, D=16, seed=37, noise_px=1.0)
    else:
        raise KeyError(name)
    go = torch.randn(value.shape[0], loc.shape[1], value.shape[2] * value.shape[3], generator=torch.Generator().manual_seed(38))
    return value, shapes, lsi, loc.contiguous(), aw.contiguous(), go


def _fwd_case(name):
    sys.path.insert(0, HERE)
    from helpers import encoder_msda_inputs, random_msda_inputs
    if name == "fwd_encoder":
        return encoder_msda_inputs(7, 19, 37, M=8, D=16, seed=41, noise_px=1.5)
    if name == "fwd_encoder_d32":
        return encoder_msda_inputs(9, 12, 20, M=4, D=32, B=2, seed=42, noise_px=1.5)
    if name == "fwd_uniform":
        return random_msda_inputs(1, [(12, 20)] * 7, 8, 16, 7 * 12 * 20, 4, seed=43)
    raise KeyError(name)


def _warp_case():
    sys.path.insert(0, ROOT)
    from mvdetr_amd import geometry
    geom = geometry.GEOMETRIES["wildtrack"]
    Ks, Rts = geometry.synthetic_rig(geom, seed=3)
    pm = geometry.build_proj_mats(geom, Ks, Rts)
    M = geometry.compose_frame_proj_mats(pm, torch.eye(3).repeat(1, geom.num_cam, 1, 1), geom.img_reduce).reshape(-1, 3, 3).float()
    src = torch.randn(geom.num_cam, 16, 90, 160, generator=torch.Generator().manual_seed(44))
    return src, M[: geom.num_cam].contiguous(), (120, 360)


BWD_CASES = ["bwd_encoder", "bwd_encoder_b2", "bwd_encoder_d32", "bwd_far", "bwd_uniform", "bwd_wildtrack"]
FWD_CASES = ["fwd_encoder", "fwd_encoder_d32", "fwd_uniform"]


def _child(family, out_path):
    """Every case of one family under this process's environment -> {case: results} (one torch start-up per route)."""
    sys.path.insert(0, ROOT)
    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    results = {}
    if family == "bwd":
        for case in BWD_CASES:
            value, shapes, lsi, loc, aw, go = [x.cuda() for x in _bwd_case(case)]
            res = [x.cpu() for x in MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)]
            if case == "bwd_wildtrack":
                out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
                lhs = (go.double() * out.double()).sum().item()
                rhs_v = (res[0].cuda().double() * value.double()).sum().item()
                rhs_a = (res[2].cuda().double() * aw.double()).sum().item()
                sub = slice(0, loc.shape[1], 997)
                res = [torch.tensor([lhs, rhs_v, rhs_a], dtype=torch.float64), res[1][:, sub].contiguous(), res[2][:, sub].contiguous(),
                       res[0].double().abs().sum()]
            results[case] = res
    elif family == "fwd":
        for case in FWD_CASES:
            value, shapes, lsi, loc, aw = [x.cuda() for x in _fwd_case(case)]
            out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64).cpu()
            results[case] = [out, MSDA.last_forward_impl()]
    elif family == "warp":
        from mvdetr_amd.ops import warp_perspective
        from mvdetr_amd.ops import warp as warp_mod
        src, M, dsize = _warp_case()
        out = warp_perspective(src.cuda(), M.cuda(), dsize).cpu()
        results["warp"] = [out, warp_mod.last_kernel()]
    else:
        raise KeyError(family)
    torch.save(results, out_path)


_CACHE = {}


def _run_child(family, env):
    key = (family, tuple(sorted(env.items())))
    if key not in _CACHE:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "out.pt")
            subprocess.run([sys.executable, os.path.abspath(__file__), family, out], env=dict(os.environ, **env), check=True, timeout=900)
            _CACHE[key] = torch.load(out)
    return _CACHE[key]


pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", BWD_ROUTES, ids=_route_id)
@pytest.mark.parametrize("case", BWD_CASES[:-1])
def test_backward_routes_vs_oracle(case, env):
    """All three gradients of every backward route against the fp64 C oracle (bars of test_msda_gpu.py's encoder sweep)."""
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    value, shapes, lsi, loc, aw, go = _bwd_case(case)
    ref = c_oracle.msda_backward(value.double(), shapes, lsi, loc.double(), aw.double(), go.double())
    got = _run_child("bwd", env)[case]
    W = float(shapes[0, 1])
    # (grad_loc of a tap within fp32 rounding of a texel centre depends on which side the rounding falls: the blend's slope
    # jumps there -- such taps are excluded, as in test_msda_gpu.py)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).double()
    px = loc.double() * wh[None, None, None, :, None, :] - 0.5
    smooth = ((px - px.round()).abs().amin(-1) > 1e-4).double()
    for a, b, name, scale in zip(got, ref, ("grad_value", "grad_loc", "grad_aw"), (1.0, W, 1.0)):
        assert torch.isfinite(a).all(), name
        err = (a.double() - b).abs() / (scale + b.abs())
        if name == "grad_loc":
            err = err * smooth[..., None]
        assert err.max().item() < 2e-4, (name, err.max().item())
    assert ref[0].abs().max().item() > 0.05


@pytest.mark.parametrize("env", BWD_ROUTES, ids=_route_id)
def test_backward_routes_at_wildtrack_size(env):
    """Full Wildtrack size: <grad_out, f(value)> = <grad_value, value> = <grad_aw, aw> (the op is linear in value and in the
    weights) and a strided subset of grad_loc / grad_aw against the fp64 oracle."""
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    value, shapes, lsi, loc, aw, go = _bwd_case("bwd_wildtrack")
    sums, gl, ga, gv_abs = _run_child("bwd", env)["bwd_wildtrack"]
    lhs, rhs_v, rhs_a = sums.tolist()
    assert abs(lhs - rhs_v) < 1e-6 * (abs(lhs) + 1e3) and abs(lhs - rhs_a) < 1e-6 * (abs(lhs) + 1e3)
    assert gv_abs.item() > 1e3
    sub = slice(0, loc.shape[1], 997)
    lo, awc, goc = [x[:, sub].double().contiguous() for x in (loc, aw, go)]
    _, rl, ra = c_oracle.msda_backward(value.double(), shapes, lsi, lo, awc, goc)
    assert ((gl.double() - rl).abs() / (50 + rl.abs())).max().item() < 1e-4
    assert ((ga.double() - ra).abs() / (1 + ra.abs())).max().item() < 5e-4


@pytest.mark.parametrize("env", FWD_ROUTES, ids=_route_id)
@pytest.mark.parametrize("case", FWD_CASES)
def test_forward_routes_vs_oracle(case, env):
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    value, shapes, lsi, loc, aw = _fwd_case(case)
    want = c_oracle.msda_forward(value.double(), shapes, lsi, loc.double(), aw.double())
    out, impl = _run_child("fwd", env)[case]
    assert (out.double() - want.view_as(out)).abs().max().item() < 1e-4
    if env.get("MVDETR_MSDA_FWD_IMPL") == "gather":
        assert impl == "gather"
    if env.get("MVDETR_MSDA_FWD_IMPL") == "tile":
        assert impl == "tile"


@pytest.mark.parametrize("env", [{}, {"MVDETR_WARP_FWD_NCHW": "gather"}], ids=_route_id)
def test_warp_nchw_routes_vs_oracle(env):
    """The literal kornia layouts (NCHW -> NCHW): the LDS-patch kernel (default) and the gather kernel the knob selects."""
    sys.path.insert(0, ROOT)
    from oracle import c_oracle
    src, M, dsize = _warp_case()
    want = c_oracle.warp_perspective(src.double(), M.double(), dsize)
    out, kernel = _run_child("warp", env)["warp"]
    assert (out.double() - want).abs().max().item() < 1e-5
    assert ("patch" in kernel) == (not env), kernel


if __name__ == "__main__":
    _child(sys.argv[1], sys.argv[2])
