"""the library's own CPU path (csrc/host_path.cpp) behind the reference's Python face -- CPU tensors
work where the reference raises (ms_deform_attn_cpu.cpp:17-41 are stubs).  Checked against the oracle and the
reference-generated goldens; runs without a GPU."""
import pytest
import torch

from conftest import load_golden, t
from helpers import level_start_index, random_msda_inputs
from oracle import c_oracle, torch_oracle


@pytest.fixture(scope="module")
def ops():
    import mvdetr_amd.ops  # noqa: F401
    import MultiScaleDeformableAttention as MSDA
    from mvdetr_amd.ops.functions import MSDeformAttnFunction
    from mvdetr_amd.ops.modules import MSDeformAttn
    from mvdetr_amd.ops import warp_perspective
    return MSDA, MSDeformAttnFunction, MSDeformAttn, warp_perspective


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_forward_testpy_golden_on_cpu_tensors(ops, tag):
    MSDA = ops[0]
    g = load_golden(f"msda_testpy_{tag}.npz")
    out = MSDA.ms_deform_attn_forward(t(g["value"]), t(g["shapes"]), t(g["level_start_index"]), t(g["loc"]), t(g["aw"]), 2)
    assert out.device.type == "cpu"
    assert (out - t(g["out"])).abs().max().item() < (1e-12 if tag == "f64" else 1e-6)


def test_forward_and_backward_mini_golden_on_cpu_tensors(ops):
    MSDA = ops[0]
    g = load_golden("msda_mini.npz")
    value, shapes, lsi, loc, aw = (t(g[k]) for k in ("value", "shapes", "level_start_index", "loc", "aw"))
    out = MSDA.ms_deform_attn_forward(value.double(), shapes, lsi, loc.double(), aw.double(), 64)
    assert (out - t(g["out"]).double()).abs().max().item() < 1e-12
    out32 = MSDA.ms_deform_attn_forward(value.float(), shapes, lsi, loc.float(), aw.float(), 64)
    assert (out32 - t(g["out_f32"])).abs().max().item() < 1e-5
    go = t(g["grad_out"]).double()
    gv, gl, ga = MSDA.ms_deform_attn_backward(value.double(), shapes, lsi, loc.double(), aw.double(), go, 64)
    for got, key in ((gv, "grad_value"), (gl, "grad_loc"), (ga, "grad_aw")):
        want = t(g[key]).double()
        assert (got - want).abs().max().item() <= 1e-10 * (1 + want.abs().max().item()), key


@pytest.mark.parametrize("B,lv,M,D,Lq,P", [(2, [(6, 4), (3, 2)], 2, 2, 5, 2), (1, [(5, 7)] * 3, 4, 16, 105, 4), (1, [(9, 4), (2, 11)], 3, 30, 17, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_host_forward_backward_vs_oracle(ops, B, lv, M, D, Lq, P, dtype):
    MSDA = ops[0]
    value, shapes, lsi, loc, aw = random_msda_inputs(B, lv, M, D, Lq, P, seed=7, dtype=dtype)
    out = MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)
    want = c_oracle.msda_forward(value, shapes, lsi, loc, aw)
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    assert (out - want).abs().max().item() < tol
    go = torch.randn(B, Lq, M * D, dtype=dtype, generator=torch.Generator().manual_seed(1))
    got = MSDA.ms_deform_attn_backward(value, shapes, lsi, loc, aw, go, 64)
    ref = c_oracle.msda_backward(value, shapes, lsi, loc, aw, go)
    for a, b in zip(got, ref):
        assert (a - b).abs().max().item() <= (2e-4 if dtype == torch.float32 else 1e-10) * (1 + b.abs().max().item())


def test_gradcheck_through_the_function_on_cpu(ops):
    F = ops[1]
    value, shapes, lsi, loc, aw = random_msda_inputs(1, [(4, 5), (3, 3)], 2, 4, 6, 2, seed=3, dtype=torch.float64, lo=0.05, hi=0.95)
    value.requires_grad_(True), loc.requires_grad_(True), aw.requires_grad_(True)
    assert torch.autograd.gradcheck(lambda v, lo_, a: F.apply(v, shapes, lsi, lo_, a, 64), (value, loc, aw), eps=1e-6, atol=1e-6)


def test_module_on_cpu_equals_the_reference_golden(ops):
    MSDeformAttn = ops[2]
    g = load_golden("msda_module.npz")
    d_model, L, M, P = (int(x) for x in g["dims"])
    mod = MSDeformAttn(d_model, L, M, P)
    mod.load_state_dict({k[2:]: t(v) for k, v in g.items() if k.startswith("p.")})
    shapes = t(g["shapes"])
    out = mod(t(g["query"]), t(g["ref"]), t(g["src"]), shapes, level_start_index(shapes))
    assert (out - t(g["out"])).abs().max().item() < 1e-5
    out.sum().backward()                                           # training works on the CPU too
    assert mod.sampling_offsets.weight.grad is not None


def test_mixed_devices_and_misuse_still_raise(ops):
    MSDA = ops[0]
    value, shapes, lsi, loc, aw = random_msda_inputs(1, [(4, 4)], 2, 4, 3, 2, seed=0)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(value.transpose(2, 3), shapes, lsi, loc, aw, 64)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc.double(), aw, 64)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(value, shapes.int(), lsi, loc, aw, 64)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_host_warp_vs_oracle_and_restatement_golden(ops, dtype):
    warp = ops[3]
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]).to(dtype), t(g["M"]).to(dtype)
    out = warp(src, M, (12, 36))
    assert out.device.type == "cpu"
    assert (out.double() - t(g["out"])).abs().max().item() < (1e-4 if dtype == torch.float32 else 1e-10)
    nhwc = warp(src, M, (12, 36), channels_last_out=True)
    assert torch.equal(nhwc.permute(0, 3, 1, 2), out)
    # backward == the oracle's autograd through grid_sample
    s1 = src.double().clone().requires_grad_(True)
    go = torch.randn(out.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    (warp(s1, M.double(), (12, 36)) * go).sum().backward()
    s2 = src.double().clone().requires_grad_(True)
    (torch_oracle.warp_perspective(s2, M.double(), (12, 36)) * go).sum().backward()
    assert (s1.grad - s2.grad).abs().max().item() < 1e-10


def test_host_warp_nearest_equals_grid_sample_nearest(ops):
    warp = ops[3]
    g = load_golden("warp_restatement.npz")
    src, M = t(g["src"]), t(g["M"])
    got = warp(src, M, (12, 36), "nearest")                        # positional mode, like frameDataset.py:80
    want = torch_oracle.warp_perspective(src, M, (12, 36), mode="nearest")
    assert torch.equal(got, want)
