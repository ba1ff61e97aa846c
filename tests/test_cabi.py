"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/mvdetr_ops.h declares, and the Python face raises for misuse exactly where the
reference does.  No kernel is launched here."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from mvdetr_amd import _lib
    _lib.build()
    return _lib


def test_header_symbols_are_exported(built):
    hdr = open(os.path.join(ROOT, "include", "mvdetr_ops.h")).read()
    declared = set(re.findall(r"\b(mvdetr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 10
    assert declared == set(built.SIGNATURES), (declared ^ set(built.SIGNATURES))
    nm = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (mvdetr_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    lib = built.lib()
    assert lib.mvdetr_ops_abi_version() == built.ABI_VERSION
    assert lib.mvdetr_msda_last_forward_impl() == b"none"


def test_header_compiles_as_plain_c():
    src = '#include "mvdetr_ops.h"\nint main(void){return MVDETR_OPS_ABI_VERSION - 1;}\n'
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-", "-fsyntax-only"], input=src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_code_object_targets_gfx950(built):
    out = subprocess.run(["strings", built.LIB_PATH], capture_output=True, text=True).stdout
    assert "amdgcn-amd-amdhsa--gfx950" in out
    assert "gfx942" not in out and "gfx90a" not in out      # one target, no fat multi-arch build


def test_extension_module_name_and_signatures(built):
    import mvdetr_amd.ops  # noqa: F401  registers the shim
    import MultiScaleDeformableAttention as MSDA
    import inspect
    fwd = list(inspect.signature(MSDA.ms_deform_attn_forward).parameters)
    bwd = list(inspect.signature(MSDA.ms_deform_attn_backward).parameters)
    assert fwd == ["value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight", "im2col_step"]
    assert bwd == ["value", "spatial_shapes", "level_start_index", "sampling_loc", "attn_weight",
                   "grad_output", "im2col_step"]


def test_cpu_tensors_run_on_the_librarys_own_host_path_and_fused_entry_stays_device_only(built):
    """where the reference raises 'Not implemented on the CPU' (ms_deform_attn.h:38) CPU tensors run on
    csrc/host_path.cpp -- never on the oracle (tests/test_host_path.py checks the numbers).  Mixed devices and the
    fused inference entry (a device-only extension) still raise."""
    import MultiScaleDeformableAttention as MSDA
    from mvdetr_amd.ops.functions import MSDeformAttnFunction
    from mvdetr_amd.ops import warp_perspective
    v = torch.ones(1, 4, 2, 2)
    s = torch.tensor([[2, 2]])
    out = MSDeformAttnFunction.apply(v, s, torch.tensor([0]), torch.full((1, 1, 2, 1, 1, 2), 0.5),
                                     torch.ones(1, 1, 2, 1, 1), 64)
    assert out.device.type == "cpu" and torch.allclose(out, torch.ones(1, 1, 4))
    w = warp_perspective(torch.ones(1, 2, 4, 4), torch.eye(3)[None], (4, 4))
    assert w.device.type == "cpu" and w.shape == (1, 2, 4, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor|Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward_fused(v, s, torch.tensor([0]), torch.zeros(1, 1, 1, 1, 2), torch.zeros(1, 1, 2, 1, 1, 2),
                                          torch.zeros(1, 1, 2, 1, 1))


def test_non_contiguous_raises(built):
    import MultiScaleDeformableAttention as MSDA
    v = torch.zeros(1, 4, 2, 4)[..., ::2]
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        MSDA.ms_deform_attn_forward(v, torch.tensor([[2, 2]]), torch.tensor([0]),
                                    torch.zeros(1, 1, 2, 1, 1, 2), torch.zeros(1, 1, 2, 1, 1), 64)


def test_missing_library_fails_loudly(tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from mvdetr_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "try:\n"
        "    _lib.lib()\n"
        "except ImportError as e:\n"
        "    print('IMPORTERROR', 'no CPU fallback' in str(e).lower() or 'There is no CPU fallback' in str(e))\n"
    ) % (ROOT, str(tmp_path / "nope.so"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "IMPORTERROR True" in out.stdout, out.stdout + out.stderr


def test_module_parity_of_parameters_and_init():
    """Parameter names / shapes / initial values of MSDeformAttn equal the reference's
    (ms_deform_attn.py:55-77); golden from the reference module itself."""
    from conftest import load_golden
    from mvdetr_amd.ops.modules import MSDeformAttn
    g = load_golden("msda_module_init.npz")
    m = MSDeformAttn(128, 7, 8, 4)
    names = sorted(k for k, _ in m.named_parameters())
    assert names == sorted(["sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight",
                            "attention_weights.bias", "value_proj.weight", "value_proj.bias",
                            "output_proj.weight", "output_proj.bias"])
    assert torch.equal(m.sampling_offsets.bias.detach(), torch.from_numpy(g["offsets_bias"]))
    assert float(m.sampling_offsets.weight.abs().max()) == float(g["offsets_weight_absmax"]) == 0.0
    assert float(m.attention_weights.weight.abs().max()) == 0.0
    assert float(m.attention_weights.bias.abs().max()) == 0.0
    assert m.im2col_step == int(g["im2col_step"]) == 64
    assert float(m.value_proj.bias.abs().max()) == 0.0 and float(m.output_proj.bias.abs().max()) == 0.0
    with pytest.raises(ValueError):
        MSDeformAttn(130, 7, 8, 4)


def test_dropin_aliases_reference_import_paths():
    """Caller code written against the reference's package paths imports unchanged."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import mvdetr_amd.dropin\n"
        "from multiview_detector.models.ops.modules import MSDeformAttn\n"
        "from multiview_detector.models.ops.functions import MSDeformAttnFunction\n"
        "from multiview_detector.models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch\n"
        "from multiview_detector.models.ops import warp_perspective\n"
        "import MultiScaleDeformableAttention as MSDA\n"
        "import mvdetr_amd.ops.modules as m\n"
        "assert MSDeformAttn is m.MSDeformAttn and hasattr(MSDA, 'ms_deform_attn_forward')\n"
        "print('DROPIN OK')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "DROPIN OK" in out.stdout, out.stdout + out.stderr


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mvdetr_amd/ may reference it."""
    import pathlib
    for path in pathlib.Path(ROOT, "mvdetr_amd").rglob("*.py"):
        text = path.read_text()
        assert "import oracle" not in text and "from oracle" not in text, path
    for path in pathlib.Path(ROOT, "mvdetr_amd", "csrc").glob("*"):
        if path.suffix in (".hip", ".h"):
            assert "oracle" not in path.read_text(), path


def test_fused_train_support_bounds_the_whole_raw_tensor(built):
    """msda_fwd_group2 folds the batch index into a 32-bit scalar offset of ONE whole-tensor buffer descriptor, so the
    camera-grouped kernels (and the training pair built on them) only take calls whose WHOLE raw tensor stays below 4 GiB; larger
    batches must be refused here (and run the 64-bit tile kernel through the inference entry)."""
    lib = built.lib()
    S, M, D, L, P = 75600, 8, 16, 7, 4
    assert lib.mvdetr_msda_fused_train_supported(1, S, M, D, L, S, P) == 1
    assert lib.mvdetr_msda_fused_train_supported(21, S, M, D, L, S, P) == 1       # 21 x 203 MB = 4.27e9 B < 2^32
    assert lib.mvdetr_msda_fused_train_supported(22, S, M, D, L, S, P) == 0       # 4.47e9 B
