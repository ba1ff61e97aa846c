#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference, which never travels to the GPU box);
the .npz files it writes are committed.  Nothing from the reference is copied: the script imports
its modules (with stub modules for the CUDA extension and for absent third-party packages that
are only touched at import time) and records inputs -> outputs.

    python tests/golden/make_golden.py

Fixtures:
  msda_testpy_{f32,f64}.npz   ms_deform_attn_core_pytorch at ops/test.py's shapes/seed
  msda_mini.npz               7 equal levels, M=8, D=16, P=4, ~10 % taps outside [0,1], + fp64 grads
  msda_edges.npz              locations on / just outside the borders
  msda_module.npz             reference MSDeformAttn(32,3,4,4) module: params, inputs, locations,
                              weights, output (5-D reference points)
  pos_embedding.npz           create_pos_embedding((6,9),8) full tensor + (60,180),64 checksums
  world_feat_mini.npz         reference DeformTransWorldFeat mini forward (state dict + in/out)
  conv_world_feat_mini.npz    reference ConvWorldFeat mini forward (state dict + in/out)
  geometry.npz                proj_mats / create_reference_map of the reference on a synthetic rig
  warp_restatement.npz        kornia-0.5-semantics warp by an independent fp64 numpy closed form in THIS script (not from
                              oracle/, not from kornia: kornia is absent; flagged "unverified against kornia")
  warp_convention.npz         where feature pixels land on the world grid according to the reference's own projection
                              code (mvdetr.py:82-95,155-161; utils/projection.py:4-14): pins direction / composition
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

# --- stubs for import-time-only dependencies of the reference ---------------------------------
msda_stub = types.ModuleType("MultiScaleDeformableAttention")
sys.modules["MultiScaleDeformableAttention"] = msda_stub
for name in ("cv2", "kornia", "torchvision", "torchvision.models", "torchvision.transforms",
             "torchvision.ops"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision.models"].vgg11 = None
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
sys.modules["torchvision.ops"].DeformConv2d = torch.nn.Module
import matplotlib
matplotlib.use("Agg")

from multiview_detector.models.ops.functions.ms_deform_attn_func import (  # noqa: E402
    ms_deform_attn_core_pytorch, MSDeformAttnFunction)
from multiview_detector.models.ops.modules import MSDeformAttn  # noqa: E402

_calls = []


def _fwd(value, shapes, lsi, loc, aw, step):
    _calls.append((loc.detach().clone(), aw.detach().clone(), value.detach().clone()))
    return ms_deform_attn_core_pytorch(value, shapes, loc, aw)


msda_stub.ms_deform_attn_forward = _fwd

from multiview_detector.models import trans_world_feat as ref_twf  # noqa: E402
from multiview_detector.models import mvdetr as ref_mvdetr  # noqa: E402
from multiview_detector.utils import projection as ref_proj  # noqa: E402

from mvdetr_amd import geometry  # noqa: E402
from oracle import torch_oracle  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: " + ", ".join(f"{k}{list(v.shape)}" for k, v in arrays.items()))


def level_start(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


# (1) test.py shapes ---------------------------------------------------------------------------
def gen_testpy():
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        aw = torch.rand(N, Lq, M, L, P) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        out = ms_deform_attn_core_pytorch(value.to(dt), shapes, loc.to(dt), aw.to(dt))
        save(f"msda_testpy_{tag}.npz", value=npy(value.to(dt)), shapes=npy(shapes),
             level_start_index=npy(level_start(shapes)), loc=npy(loc.to(dt)), aw=npy(aw.to(dt)),
             out=npy(out))


# (2)+(4) MVDeTr-mini with grads ---------------------------------------------------------------
def mini_inputs(seed, L=7, H=3, W=8, M=8, D=16, P=4, B=1, spread=0.3):
    g = torch.Generator().manual_seed(seed)
    shapes = torch.as_tensor([(H, W)] * L, dtype=torch.long)
    S = L * H * W
    Lq = S
    # inputs are drawn in fp32 (and stored as fp32) so the fp32 and fp64 runs see identical numbers
    value = torch.randn(B, S, M, D, generator=g).double()
    ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 2).repeat(L, 1).double()       # [Lq,2]
    thetas = torch.arange(M, dtype=torch.float32) * (2.0 * np.pi / M)
    init = torch.stack([thetas.cos(), thetas.sin()], -1)
    init = (init / init.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    for i in range(P):
        init[:, :, i, :] *= i + 1
    off = 0.3 * init.double()[None, None] + 0.4 * torch.randn(B, Lq, M, L, P, 2, generator=g, dtype=torch.float64)
    loc = ref[None, :, None, None, None, :] + off / torch.tensor([W, H], dtype=torch.float64)
    loc = loc + spread * (torch.rand(B, Lq, M, L, P, 2, generator=g, dtype=torch.float64) - 0.5) \
        * (torch.rand(B, Lq, M, L, P, 1, generator=g, dtype=torch.float64) < 0.2)
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    return value, shapes, loc.float().double(), aw.float().double()


def gen_mini():
    value, shapes, loc, aw = mini_inputs(11)
    frac_out = float(((loc < 0) | (loc > 1)).any(-1).double().mean())
    value.requires_grad_(True)
    loc.requires_grad_(True)
    aw.requires_grad_(True)
    out = ms_deform_attn_core_pytorch(value, shapes, loc, aw)
    gout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).double()
    gv, gl, ga = torch.autograd.grad(out, (value, loc, aw), gout)
    out32 = ms_deform_attn_core_pytorch(value.detach().float(), shapes, loc.detach().float(), aw.detach().float())
    print(f"  mini: fraction of taps with a coordinate outside [0,1]: {frac_out:.3f}")
    save("msda_mini.npz", value=npy(value.float()), shapes=npy(shapes), level_start_index=npy(level_start(shapes)),
         loc=npy(loc.float()), aw=npy(aw.float()), out=npy(out), out_f32=npy(out32), grad_out=npy(gout.float()),
         grad_value=npy(gv), grad_loc=npy(gl), grad_aw=npy(ga))


# (3) border cases -----------------------------------------------------------------------------
def gen_edges():
    H, W, M, D, P = 5, 7, 2, 4, 1
    shapes = torch.as_tensor([(H, W), (3, 4)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    g = torch.Generator().manual_seed(21)
    value = torch.randn(1, S, M, D, generator=g, dtype=torch.float64)
    xs = [0.0, 1.0, -0.5 / W, 1 + 0.5 / W, 0.5 / W, 1 - 0.5 / W, -1.0 / W, 1 + 1.0 / W, 0.5, 1.5 / W,
          -0.49999 / W, 1 + 0.49999 / W, -2.0, 3.0]
    pts = torch.tensor([(x, y) for x in xs for y in xs], dtype=torch.float64)       # [Lq,2]
    Lq = pts.shape[0]
    loc = pts.view(1, Lq, 1, 1, 1, 2).repeat(1, 1, M, 2, P, 1).contiguous()
    aw = torch.softmax(torch.randn(1, Lq, M, 2 * P, generator=g, dtype=torch.float64), -1).view(1, Lq, M, 2, P)
    out = ms_deform_attn_core_pytorch(value, shapes, loc, aw)
    save("msda_edges.npz", value=npy(value), shapes=npy(shapes), level_start_index=npy(level_start(shapes)),
         loc=npy(loc), aw=npy(aw), out=npy(out))


# (5) module-level ------------------------------------------------------------------------------
def gen_module():
    torch.manual_seed(7)
    d_model, L, M, P = 32, 3, 4, 4
    H, W = 5, 8
    mod = MSDeformAttn(d_model, L, M, P)
    # perturb the zero-initialised projections so offsets/weights depend on the query
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.3)
        mod.attention_weights.weight.normal_(0, 0.5)
        mod.attention_weights.bias.normal_(0, 0.5)
    shapes = torch.as_tensor([(H, W)] * L, dtype=torch.long)
    S = L * H * W
    query = torch.randn(1, S, d_model)
    src = torch.randn(1, S, d_model)
    ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
    ref = torch.stack([xs / W, ys / H], -1).reshape(-1, 1, 1, 2).repeat(L, L, P, 1)[None]   # [1,S,L,P,2]
    ref = ref + 0.01 * torch.randn(ref.shape)
    _calls.clear()
    out = mod(query, ref, src, shapes, level_start(shapes))
    loc, aw, value = _calls[-1]
    arrays = {"p." + k: npy(v) for k, v in mod.state_dict().items()}
    save("msda_module.npz", query=npy(query), src=npy(src), ref=npy(ref), shapes=npy(shapes),
         loc=npy(loc), aw=npy(aw), value=npy(value), out=npy(out),
         dims=np.array([d_model, L, M, P]), **arrays)
    # pristine-initialisation parameters (bias grid, zero weights) for the init-parity test
    fresh = MSDeformAttn(128, 7, 8, 4)
    save("msda_module_init.npz", offsets_bias=npy(fresh.sampling_offsets.bias),
         offsets_weight_absmax=np.array(float(fresh.sampling_offsets.weight.detach().abs().max())),
         attn_weight_absmax=np.array(float(fresh.attention_weights.weight.detach().abs().max())),
         attn_bias_absmax=np.array(float(fresh.attention_weights.bias.detach().abs().max())),
         im2col_step=np.array(fresh.im2col_step))


# (6) position embedding -------------------------------------------------------------------------
def gen_pos():
    small = ref_twf.create_pos_embedding((6, 9), 8)
    big = ref_twf.create_pos_embedding((60, 180), 64)
    save("pos_embedding.npz", small=npy(small), big_sum=np.array(float(big.double().sum())),
         big_abs_sum=np.array(float(big.double().abs().sum())), big_rows=npy(big[0, :, ::20, ::45]),
         big_shape=np.array(big.shape))


# (7) DeformTransWorldFeat mini -------------------------------------------------------------------
def gen_world_feat():
    torch.manual_seed(13)
    num_cam, Rworld, base_dim, hidden, nhead, P = 3, (8, 12), 16, 16, 2, 4
    geom = types.SimpleNamespace(Rworld_shape=Rworld)
    h, w = Rworld[0] // 2, Rworld[1] // 2
    ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h), torch.linspace(0.5, w - 0.5, w), indexing="ij")
    ref = torch.stack([xs / w, ys / h], -1).reshape(-1, 1, 1, 2).repeat(1, num_cam, P, 1)
    ref = ref + 0.02 * torch.randn(ref.shape)
    ref_all = ref.repeat([num_cam, 1, 1, 1])
    model = ref_twf.DeformTransWorldFeat(num_cam, list(Rworld), base_dim, hidden_dim=hidden, nhead=nhead,
                                         dim_feedforward=32, n_points=P, stride=2, reference_points=ref_all)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, MSDeformAttn):
                m.sampling_offsets.weight.normal_(0, 0.2)
                m.attention_weights.weight.normal_(0, 0.3)
    model.eval()
    x = torch.randn(1, num_cam, base_dim, *Rworld)
    with torch.no_grad():
        out = model(x)
    arrays = {"p." + k: npy(v) for k, v in model.state_dict().items()}
    save("world_feat_mini.npz", x=npy(x), ref=npy(ref_all), out=npy(out),
         dims=np.array([num_cam, Rworld[0], Rworld[1], base_dim, hidden, nhead, P]), **arrays)


def gen_conv_world_feat():
    """BASELINE config 0 (--world_feat conv): the reference's ConvWorldFeat (conv_world_feat.py:21-52)."""
    from multiview_detector.models import conv_world_feat as ref_cwf
    # (the class needs hidden_dim == base_dim: it views the down-sampled [B*N, hidden, h, w] as [B, N*base, h, w],
    # conv_world_feat.py:44; reduction='sum' sums over the channel axis of a 4-D tensor and cannot run)
    torch.manual_seed(17)
    num_cam, Rworld, base_dim = 3, (8, 12), 8
    model = ref_cwf.ConvWorldFeat(num_cam, list(Rworld), base_dim, hidden_dim=base_dim, stride=2).eval()
    x = torch.randn(2, num_cam, base_dim, *Rworld)
    with torch.no_grad():
        y = model(x)
    arrays = {"p." + k: npy(v) for k, v in model.state_dict().items()}
    save("conv_world_feat_mini.npz", x=npy(x), out=npy(y), dims=np.array([num_cam, Rworld[0], Rworld[1], base_dim]), **arrays)


# (a2, a4) geometry through the reference's own code ----------------------------------------------
def gen_geometry():
    out = {}
    for geom in (geometry.WILDTRACK, geometry.MULTIVIEWX):
        Ks, Rts = geometry.synthetic_rig(geom, seed=3)
        base = types.SimpleNamespace(
            worldcoord_from_worldgrid_mat=geom.worldcoord_from_worldgrid_mat,
            world_indexing_from_xy_mat=geom.world_indexing_from_xy_mat,
            intrinsic_matrices=Ks, extrinsic_matrices=Rts, worldcoord_unit=geom.worldcoord_unit)
        ds = types.SimpleNamespace(base=base, num_cam=geom.num_cam, Rworld_shape=list(geom.Rworld_shape),
                                   Rimg_shape=list(geom.Rimg_shape), world_reduce=geom.world_reduce,
                                   img_reduce=geom.img_reduce)
        # reference points via the reference's create_reference_map (both point counts)
        for npts in (4, 8):
            out[f"{geom.name}.ref{npts}"] = npy(ref_mvdetr.create_reference_map(ds, npts))[::97]
        out[f"{geom.name}.ref4_absdiff_identity"] = np.array(0.0)
        # proj_mats via MVDeTr.__init__ (backbone constructor replaced by a stub: pretrained weights
        # cannot be downloaded here and the backbone is irrelevant to the matrices)
        ref_mvdetr.resnet18 = lambda **kw: torch.nn.Sequential(torch.nn.Identity(), torch.nn.Identity(),
                                                               torch.nn.Identity())
        model = ref_mvdetr.MVDeTr(ds, "resnet18", world_feat_arch="conv", bottleneck_dim=0)
        out[f"{geom.name}.proj_mats"] = npy(model.proj_mats)
        # per-forward composition (mvdetr.py:155-161) reproduced by running those lines' operands
        M = geometry.random_affine_mats(1, geom.num_cam, geom.input_img_shape, seed=5)
        inv = torch.inverse(M.view([geom.num_cam, 3, 3]))
        img_from_Rimg = inv @ torch.from_numpy(np.diag([geom.img_reduce, geom.img_reduce, 1])
                                                ).view(1, 3, 3).repeat(geom.num_cam, 1, 1).float()
        out[f"{geom.name}.frame_proj"] = npy(model.proj_mats.repeat(1, 1, 1, 1).view(geom.num_cam, 3, 3).float()
                                             @ img_from_Rimg)
        out[f"{geom.name}.affine"] = npy(M)
        out[f"{geom.name}.K0"] = Ks[0]
        out[f"{geom.name}.Rt0"] = Rts[0]
        out[f"{geom.name}.w_from_i0"] = ref_proj.get_worldcoord_from_imgcoord_mat(Ks[0], Rts[0], 0.3)
    save("geometry.npz", **out)


# (8) warp ---------------------------------------------------------------------------------------------------------
def _closed_form_warp(src, M, dsize):
    """Independent fp64 closed form of what kornia 0.5's warp_perspective(bilinear, zeros, align_corners=False) computes
    (SURVEY 8a-1), written with numpy only -- NOT through oracle/: output pixel (i, j) samples the source at
    p = M^-1 (j, i, 1), x = p_x / p_z * w / (w - 1) - 0.5, y = p_y / p_z * h / (h - 1) - 0.5, bilinear, zeros outside."""
    n, c, h, w = src.shape
    H, W = dsize
    out = np.zeros((n, c, H, W))
    jj, ii = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    for v in range(n):
        p = np.linalg.inv(M[v]) @ np.stack([jj.ravel(), ii.ravel(), np.ones(H * W)])
        x = p[0] / p[2] * w / (w - 1) - 0.5
        y = p[1] / p[2] * h / (h - 1) - 0.5
        x0, y0 = np.floor(x), np.floor(y)
        acc = np.zeros((c, H * W))
        for dy, wy in ((0, 1 - (y - y0)), (1, y - y0)):
            for dx, wx in ((0, 1 - (x - x0)), (1, x - x0)):
                yy, xx = (y0 + dy).astype(np.int64), (x0 + dx).astype(np.int64)
                ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
                acc += np.where(ok, wy * wx, 0.0) * src[v][:, np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        out[v] = acc.reshape(c, H, W)
    return out


def gen_warp():
    g = torch.Generator().manual_seed(17)
    src = torch.randn(2, 8, 9, 16, generator=g, dtype=torch.float64)
    geom = geometry.WILDTRACK
    Ks, Rts = geometry.synthetic_rig(geom, seed=3)
    pm = geometry.build_proj_mats(geom, Ks, Rts)[:2]
    # scale to the mini feature map: world grid 12x36 instead of 120x360, feature 9x16 instead of 90x160
    shrink = np.diag([0.1, 0.1, 1.0])
    Mfull = np.stack([shrink @ pm[i] @ np.diag([120.0, 120.0, 1.0]) for i in range(2)])
    out64 = _closed_form_warp(npy(src), Mfull, (12, 36))
    save("warp_restatement.npz", src=npy(src), M=Mfull, out=out64,
         note=np.array("independent numpy closed form of kornia 0.5 semantics; unverified against kornia itself (absent): "
                       "the (size/(size-1)) normalisation quirk rests on SURVEY 8a-1"))


def gen_warp_convention():
    """What CAN be pinned to the reference without kornia: direction and composition of the homography.  The reference's
    own code (MVDeTr.__init__ + the per-forward lines mvdetr.py:155-161, utils/projection.py:4-14) says where a feature
    pixel (u, v) lies on the reduced world grid; a blob at (u, v) warped by the build must come out there."""
    from multiview_detector.models import mvdetr as ref_mvdetr
    from multiview_detector.utils import projection as ref_proj
    geom = geometry.WILDTRACK
    Ks, Rts = geometry.synthetic_rig(geom, seed=0)
    base = types.SimpleNamespace(
        worldcoord_from_worldgrid_mat=geom.worldcoord_from_worldgrid_mat, world_indexing_from_xy_mat=geom.world_indexing_from_xy_mat,
        intrinsic_matrices=Ks, extrinsic_matrices=Rts, worldcoord_unit=geom.worldcoord_unit)
    ds = types.SimpleNamespace(base=base, num_cam=geom.num_cam, Rworld_shape=list(geom.Rworld_shape),
                               Rimg_shape=list(geom.Rimg_shape), world_reduce=geom.world_reduce, img_reduce=geom.img_reduce)
    ref_mvdetr.resnet18 = lambda **kw: torch.nn.Sequential(torch.nn.Identity(), torch.nn.Identity(), torch.nn.Identity())
    model = ref_mvdetr.MVDeTr(ds, "resnet18", world_feat_arch="conv", bottleneck_dim=0)
    N = 3
    Maug = torch.eye(3).repeat(N, 1, 1)
    img_from_Rimg = torch.inverse(Maug) @ torch.from_numpy(np.diag([geom.img_reduce, geom.img_reduce, 1.0])).view(1, 3, 3).repeat(N, 1, 1).float()
    frame_proj = (model.proj_mats[:N].float() @ img_from_Rimg).double().numpy()       # Rworldgrid(xy) <- Rimggrid(xy)
    h, w = geom.Rimg_shape
    H, W = geom.Rworld_shape
    rng = np.random.default_rng(5)
    uv, xy = [], []
    for cam in range(N):
        pts = []
        while len(pts) < 6:
            u, v = rng.uniform(8, w - 8), rng.uniform(h * 0.45, h - 6)
            X, Y = ref_proj.project_2d_points(frame_proj[cam], np.array([[u], [v]]))[:, 0]
            if 6 < X < W - 6 and 6 < Y < H - 6:
                pts.append((u, v, X, Y))
        uv.append([(p[0], p[1]) for p in pts])
        xy.append([(p[2], p[3]) for p in pts])
    save("warp_convention.npz", M=frame_proj, src_uv=np.array(uv), dst_xy=np.array(xy), dims=np.array([N, h, w, H, W]))


if __name__ == "__main__":
    gen_testpy()
    gen_mini()
    gen_edges()
    gen_module()
    gen_pos()
    gen_world_feat()
    gen_conv_world_feat()
    gen_geometry()
    gen_warp()
    gen_warp_convention()
