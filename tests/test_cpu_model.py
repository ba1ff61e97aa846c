"""The caller model on CPU tensors, through the product's own host path (no GPU): BASELINE.json configs[0]
(--world_feat conv, "PyTorch CPU-only"), the deform_trans path the reference cannot run without CUDA, and the
ResNet-50 trunk of configs[3]."""
import pytest
import torch

from mvdetr_amd import geometry
from mvdetr_amd.model import build_model
from oracle import frame_oracle, torch_oracle


def _inputs(seed=3):
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(1, 3, 3, *geometry.MINI.input_img_shape, generator=g)
    M = geometry.random_affine_mats(1, 3, geometry.MINI.input_img_shape, seed=2, translate=0.05, scale=(0.9, 1.1))
    return imgs, M


def test_config0_conv_world_feat_runs_on_the_cpu_and_matches_the_oracle():
    model = build_model("mini", seed=0, world_feat_arch="conv", channels_last=False).eval()
    imgs, M = _inputs()
    with torch.no_grad():
        (wh, wo), (ih, io, iw) = model(imgs, M)
        assert wh.shape == (1, 1, 24, 72) and wo.shape == (1, 2, 24, 72) and ih.shape == (3, 1, 18, 32)
        # same features through the oracle's warp + ConvWorldFeat restatement
        feat = model.features(imgs)
        proj = model.frame_proj_mats(M)
        world = torch_oracle.warp_perspective(feat, proj, model.Rworld_shape).view(1, 3, -1, *model.Rworld_shape)
        p = {k[len("world_feat."):]: v for k, v in model.state_dict().items() if k.startswith("world_feat.")}
        want = torch_oracle.conv_world_feat(p, world)
        got = model.hot_path(feat, proj)
    assert (got - want).abs().max().item() < 1e-4


def test_deform_trans_frame_on_the_cpu_matches_the_frame_oracle():
    """The reference's MSDeformAttn raises on CPU tensors; here the whole frame runs on csrc/host_path.cpp."""
    model = build_model("mini", seed=0, channels_last=False).eval()
    with torch.no_grad():
        for layer in model.world_feat.encoder.layers:
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.02)
            layer.self_attn.attention_weights.weight.normal_(0, 0.05)
    imgs, M = _inputs()
    with torch.no_grad():
        (wh, wo), _ = model(imgs, M)
        p = {k: v.detach() for k, v in model.state_dict().items()}
        ref = model.world_feat.encoder.reference_points
        (rwh, rwo), _ = frame_oracle.forward(p, imgs, model.frame_proj_mats(M), model.Rworld_shape, ref, 3)
    assert (wh - rwh).abs().max().item() < 1e-4 and (wo - rwo).abs().max().item() < 1e-4


def test_resnet50_trunk_is_wired():
    model = build_model("mini", seed=0, arch="resnet50", channels_last=False).eval()
    assert sum(p.numel() for p in model.base.parameters()) == 23_508_032          # torchvision's resnet50 without fc
    imgs, M = _inputs()
    with torch.no_grad():
        (wh, wo), (ih, _, _) = model(imgs, M)
    assert wh.shape == (1, 1, 24, 72) and ih.shape == (3, 1, 18, 32) and torch.isfinite(wh).all()
    with pytest.raises(ValueError):
        build_model("mini", arch="vgg11")


def test_shared_reference_follows_the_reference_points_buffer():
    """the encoder's one-point-per-(query, level) shortcut must be re-derived when reference_points is replaced
    or written in place -- the fused path would otherwise keep sampling from a stale map."""
    import torch
    from mvdetr_amd.world_feat import DeformableTransformerEncoder, DeformableTransformerEncoderLayer
    ref = torch.rand(12, 3, 1, 2).repeat(1, 1, 4, 1)                       # [Lq, L, P, 2], one point repeated P times
    enc = DeformableTransformerEncoder(DeformableTransformerEncoderLayer(16, 32, 0.0, 3, 2, 4), 1, ref.clone())
    a = enc.shared_reference()
    assert a.shape == (3, 12, 2) and torch.equal(a, ref[..., 0, :].transpose(0, 1))
    assert enc.shared_reference() is a                                     # cached while the buffer is untouched
    enc.reference_points.mul_(0.5)                                         # in-place edit: same tensor, new version
    assert torch.equal(enc.shared_reference(), 0.5 * ref[..., 0, :].transpose(0, 1))
    enc.reference_points = torch.rand(12, 3, 4, 2)                         # replaced by a map with P distinct points
    assert enc.shared_reference() is None


def test_config0_at_the_real_multiviewx_geometry_on_the_cpu():
    """BASELINE.json configs[0] at its own size (only a 3-camera mini geometry had run): MultiviewX, 6 cameras,
    3x720x1280 inputs, ResNet-18, --world_feat conv, CPU only.  The whole frame runs through the product (trunk: torch; warp:
    the library's host path; ConvWorldFeat: torch), and the path's output -- warp + world features from the model's own
    image features -- is checked against the oracle's kornia restatement + ConvWorldFeat restatement."""
    geom = geometry.MULTIVIEWX
    model = build_model("multiviewx", seed=0, world_feat_arch="conv", channels_last=False).eval()
    g = torch.Generator().manual_seed(5)
    imgs = torch.randn(1, geom.num_cam, 3, *geom.input_img_shape, generator=g)
    M = geometry.random_affine_mats(1, geom.num_cam, geom.input_img_shape, seed=4, translate=0.05, scale=(0.9, 1.1))
    assert imgs.shape == (1, 6, 3, 720, 1280) and model.Rworld_shape == (160, 250)
    with torch.no_grad():
        (wh, wo), (ih, io, iw) = model(imgs, M)
        assert wh.shape == (1, 1, 160, 250) and wo.shape == (1, 2, 160, 250) and ih.shape == (6, 1, 90, 160)
        feat = model.features(imgs)
        proj = model.frame_proj_mats(M)
        got = model.hot_path(feat, proj)
        world = torch_oracle.warp_perspective(feat, proj, model.Rworld_shape).view(1, 6, -1, *model.Rworld_shape)
        p = {k[len("world_feat."):]: v for k, v in model.state_dict().items() if k.startswith("world_feat.")}
        want = torch_oracle.conv_world_feat(p, world)
        # and the heads on top reproduce the frame's outputs
        assert (model.world_heatmap(got) - wh).abs().max().item() < 1e-5
    assert got.shape == want.shape == (1, 128, 160, 250)
    scale = want.abs().max().item()
    assert scale > 1e-3
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, scale)


def test_model_copies_and_pickles_without_its_upload_state():
    """the per-device staging buffers / events / cached uploads of MVDeTr.frame_proj_mats are run-time state,
    not part of the model: deepcopy and pickle leave them behind, and the host copy of proj_mats follows the buffer object
    (`is` + version), not its id()."""
    import copy
    import io
    from mvdetr_amd.model import build_model
    m = build_model("mini", seed=0).eval()
    M = torch.eye(3).repeat(1, m.num_cam, 1, 1)
    p0 = m.frame_proj_mats(M)
    m._transient[("cuda", 0, 0)] = object()                      # stands in for pinned buffers + events
    m2 = copy.deepcopy(m)
    assert m2._transient == {} and m2._proj_host_src is None
    assert torch.equal(m2.frame_proj_mats(M), p0)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3._transient == {} and torch.equal(m3.frame_proj_mats(M), p0)
    # replacing the buffer is seen even if the new tensor were to get the old one's id: the cache holds the object
    m.proj_mats = m.proj_mats.clone() * 2.0
    assert not torch.equal(m.frame_proj_mats(M), p0)
