"""Fused residual add + LayerNorm (mvdetr_add_layernorm_f32) against torch's own fp32 ops and an fp64 reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(75600, 128), (1, 128), (257, 64), (1000, 256), (3, 128)])
@pytest.mark.parametrize("affine,with_res", [(True, True), (False, True), (True, False)])
def test_add_layer_norm_matches_torch(rows, cols, affine, with_res):
    from mvdetr_amd.ops.add_layernorm import add_layer_norm, fused_add_layer_norm_available
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(2, rows, cols, generator=g) * 3 + 0.7).cuda()
    res = (torch.randn(2, rows, cols, generator=g) * 0.5).cuda() if with_res else None
    norm = torch.nn.LayerNorm(cols, elementwise_affine=affine).cuda()
    if affine:
        with torch.no_grad():
            norm.weight.normal_(1.0, 0.3, generator=None)
            norm.bias.normal_(0.0, 0.3)
    with torch.no_grad():
        assert fused_add_layer_norm_available(x, norm)
        got = add_layer_norm(x, res, norm)
        want32 = norm(x if res is None else x + res)
        s = (x if res is None else x + res).double()
        want64 = torch.nn.functional.layer_norm(s, (cols,), None if not affine else norm.weight.double(),
                                                None if not affine else norm.bias.double(), norm.eps)
    assert got.shape == x.shape and got.dtype == torch.float32
    assert (got.double() - want64).abs().max().item() < 5e-6
    assert (got - want32).abs().max().item() < 5e-6


def test_add_layer_norm_training_and_other_shapes_use_torch(monkeypatch):
    from mvdetr_amd import _lib
    from mvdetr_amd.ops.add_layernorm import add_layer_norm, fused_add_layer_norm_available
    norm = torch.nn.LayerNorm(128).cuda()
    x = torch.randn(4, 128, device="cuda", requires_grad=True)
    assert not fused_add_layer_norm_available(x, norm)                  # autograd needed
    y = add_layer_norm(x, x.detach(), norm)
    y.sum().backward()
    assert x.grad is not None
    odd = torch.nn.LayerNorm(96).cuda()
    with torch.no_grad():
        assert not fused_add_layer_norm_available(torch.randn(4, 96, device="cuda"), odd)
        assert add_layer_norm(torch.randn(4, 96, device="cuda"), None, odd).shape == (4, 96)
    # non-contiguous inputs are accepted (copied)
    with torch.no_grad():
        xt = torch.randn(128, 40, device="cuda").t()
        assert (add_layer_norm(xt, None, norm) - norm(xt)).abs().max().item() < 5e-6
    # unsupported widths are refused by the C entry itself
    lib = _lib.lib()
    buf = torch.zeros(96, device="cuda")
    assert lib.mvdetr_add_layernorm_f32(0, buf.data_ptr(), 0, 0, 0, 1, 96, 1e-5, buf.data_ptr()) == 801


def test_encoder_layer_eval_uses_the_fused_tail_and_matches_training_mode_ops():
    from mvdetr_amd.world_feat import DeformableTransformerEncoderLayer
    torch.manual_seed(0)
    L, H, W, C = 3, 8, 12, 128
    layer = DeformableTransformerEncoderLayer(C, 256, 0.1, n_levels=L, n_heads=8, n_points=4).cuda().eval()
    S = L * H * W
    src, pos = torch.randn(1, S, C, device="cuda"), torch.randn(1, S, C, device="cuda")
    ys, xs = torch.meshgrid(torch.arange(H) + 0.5, torch.arange(W) + 0.5, indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(1, H * W, 1, 1, 2).repeat(1, L, L, 4, 1).cuda()
    shapes = torch.tensor([[H, W]] * L, device="cuda")
    lsi = torch.arange(L, device="cuda") * H * W
    with torch.no_grad():
        fused = layer(src, pos, ref, shapes, lsi)
    src_g = src.clone().requires_grad_(True)                # grad mode: the torch formulation
    plain = layer(src_g, pos, ref, shapes, lsi)
    assert (fused - plain.detach()).abs().max().item() < 2e-5


@pytest.mark.parametrize("batch_pos", [False, True])
def test_add_layer_norm_second_output(batch_pos):
    from mvdetr_amd.ops.add_layernorm import add_layer_norm
    g = torch.Generator().manual_seed(3)
    x, res = torch.randn(2, 501, 128, generator=g).cuda(), torch.randn(2, 501, 128, generator=g).cuda()
    pos = torch.randn(2 if batch_pos else 1, 501, 128, generator=g).cuda()
    norm = torch.nn.LayerNorm(128).cuda()
    with torch.no_grad():
        y, y2 = add_layer_norm(x, res, norm, then_add=pos)
        want = norm(x + res)
    assert (y - want).abs().max().item() < 5e-6
    assert (y2 - (want + pos)).abs().max().item() < 5e-6
    # torch path (autograd needed) returns the same pair
    xg = x.clone().requires_grad_(True)
    z, z2 = add_layer_norm(xg, res, norm, then_add=pos)
    assert (z.detach() - y).abs().max().item() < 5e-6 and (z2.detach() - y2).abs().max().item() < 5e-6


def test_encoder_folds_position_add_and_matches_layer_by_layer():
    """DeformableTransformerEncoder passes each layer's `output + pos` on as the next layer's query; the result is
    the plain layer-by-layer evaluation's."""
    from mvdetr_amd.world_feat import DeformTransWorldFeat
    torch.manual_seed(1)
    N, H, W, C = 3, 16, 24, 128
    h, w = H // 2, W // 2
    ys, xs = torch.meshgrid(torch.arange(h) + 0.5, torch.arange(w) + 0.5, indexing="ij")
    ref = torch.stack([xs / w, ys / h], -1).reshape(1, h * w, 1, 1, 2).repeat(N, 1, N, 4, 1).view(-1, N, 4, 2)
    wf = DeformTransWorldFeat(N, (H, W), C, hidden_dim=C, reference_points=ref).cuda().eval()
    tokens = torch.randn(2, N * h * w, C, device="cuda")
    with torch.no_grad():
        pos = wf.level_pos(h, w)
        got = wf.encoder(tokens, wf.spatial_shapes, wf.level_start_index, None, pos)
        out = tokens
        refs = wf.encoder.reference_points.unsqueeze(0).expand(2, -1, -1, -1, -1)
        for layer in wf.encoder.layers:
            out = layer(out, pos, refs, wf.spatial_shapes, wf.level_start_index)
    assert (got - out).abs().max().item() < 1e-5
