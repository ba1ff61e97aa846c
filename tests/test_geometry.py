"""Host-side geometry (proj matrices, reference points) against values produced by the reference's
own code on the same synthetic rig (tests/golden/geometry.npz)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from mvdetr_amd import geometry


@pytest.mark.parametrize("geom", [geometry.WILDTRACK, geometry.MULTIVIEWX], ids=lambda g: g.name)
def test_proj_mats_and_reference_map(geom):
    g = load_golden("geometry.npz")
    Ks, Rts = geometry.synthetic_rig(geom, seed=3)
    np.testing.assert_allclose(Ks[0], g[f"{geom.name}.K0"], rtol=0, atol=0)
    np.testing.assert_allclose(Rts[0], g[f"{geom.name}.Rt0"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(geometry.get_worldcoord_from_imgcoord_mat(Ks[0], Rts[0], 0.3),
                               g[f"{geom.name}.w_from_i0"], rtol=1e-12)
    pm = geometry.build_proj_mats(geom, Ks, Rts)
    np.testing.assert_allclose(pm, g[f"{geom.name}.proj_mats"], rtol=1e-12, atol=1e-15)
    M = torch.from_numpy(g[f"{geom.name}.affine"])
    fp = geometry.compose_frame_proj_mats(pm, M, geom.img_reduce)
    np.testing.assert_allclose(fp.numpy(), g[f"{geom.name}.frame_proj"], rtol=1e-6, atol=1e-9)
    for npts in (4, 8):
        ref = geometry.create_reference_map(geom, Ks, Rts, npts).numpy()[::97]
        np.testing.assert_allclose(ref, g[f"{geom.name}.ref{npts}"], rtol=0, atol=2e-6)


def test_reference_map_is_identity_for_four_points():
    geom = geometry.WILDTRACK
    Ks, Rts = geometry.synthetic_rig(geom, seed=3)
    ref = geometry.create_reference_map(geom, Ks, Rts, 4)
    H, W = geom.Rworld_shape[0] // 2, geom.Rworld_shape[1] // 2
    assert ref.shape == (H * W, 7, 4, 2)
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ident = torch.stack([xs / W, ys / H], -1).reshape(-1, 1, 1, 2)
    assert (ref - ident).abs().max().item() < 1e-5


def test_shapes_match_survey_table():
    assert geometry.WILDTRACK.Rworld_shape == (120, 360)
    assert geometry.WILDTRACK.Rimg_shape == (90, 160)
    assert geometry.WILDTRACK.input_img_shape == (720, 1280)
    assert geometry.MULTIVIEWX.Rworld_shape == (160, 250)
    assert geometry.STRESS16.Rimg_shape == (135, 240)
    assert geometry.STRESS16.input_img_shape == (1080, 1920)
