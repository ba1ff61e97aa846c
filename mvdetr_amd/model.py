"""Minimal caller of the fusion path: an MVDeTr-shaped detector whose warp and deformable attention
run on the HIP kernels.  Everything that is not the hot path (ResNet-18 trunk, heads) is ordinary
PyTorch-ROCm, as in the reference.

Mirrors multiview_detector/models/mvdetr.py:74-218 (constructor arithmetic, forward order, output
tuple) and the dilated ResNet-18 trunk of multiview_detector/models/resnet.py:33-70,120-186
(``replace_stride_with_dilation=[False, True, True]``, children()[:-2] -> stride 8, 512 channels;
parameter names follow ``base.<child index>...`` so reference checkpoints line up).  Weights are
seeded random: there is no network for the pretrained download (resnet.py:213-216).
"""
from __future__ import annotations

import threading

import torch
from torch import nn

from . import geometry
from .ops import warp_perspective
from .world_feat import ConvWorldFeat, DeformTransWorldFeat


class BasicBlock(nn.Module):
    """3x3-3x3 residual block; only the FIRST conv is dilated (resnet.py:46-51 of the fork)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    """1x1 - 3x3 (strided / dilated) - 1x1 residual block, expansion 4 (resnet.py:72-112): the block of the
    ResNet-50 that BASELINE.json's configs[3] names.  The reference defines resnet50 (resnet.py:244) but wires only
    vgg11 / resnet18 into the detector (mvdetr.py:97-107)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


def resnet_trunk(depth=18, replace_stride_with_dilation=(False, True, True), in_channels=3) -> nn.Sequential:
    """conv1, bn1, relu, maxpool, layer1..4 as one Sequential (the reference slices
    list(resnet18(...).children())[:-2], mvdetr.py:103-105).  depth 18: BasicBlock x [2,2,2,2], 512 channels out;
    depth 50: Bottleneck x [3,4,6,3], 2048 channels out (resnet.py:226-250)."""
    block, counts = {18: (BasicBlock, (2, 2, 2, 2)), 50: (Bottleneck, (3, 4, 6, 3))}[depth]
    exp = getattr(block, "expansion", 1)
    state = {"inplanes": 64, "dilation": 1}

    def make_layer(planes, blocks, stride=1, dilate=False):
        prev = state["dilation"]
        if dilate:
            state["dilation"] *= stride
            stride = 1
        down = None
        if stride != 1 or state["inplanes"] != planes * exp:
            down = nn.Sequential(nn.Conv2d(state["inplanes"], planes * exp, 1, stride, bias=False), nn.BatchNorm2d(planes * exp))
        layers = [block(state["inplanes"], planes, stride, down, prev)]
        state["inplanes"] = planes * exp
        layers += [block(planes * exp, planes, dilation=state["dilation"]) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    trunk = nn.Sequential(
        nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
        nn.MaxPool2d(3, 2, 1),
        make_layer(64, counts[0]), make_layer(128, counts[1], 2, replace_stride_with_dilation[0]),
        make_layer(256, counts[2], 2, replace_stride_with_dilation[1]), make_layer(512, counts[3], 2, replace_stride_with_dilation[2]))
    for m in trunk.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return trunk


def resnet18_trunk(replace_stride_with_dilation=(False, True, True), in_channels=3) -> nn.Sequential:
    return resnet_trunk(18, replace_stride_with_dilation, in_channels)


def output_head(in_dim, feat_dim, out_dim):
    # mvdetr.py:24-30
    if feat_dim:
        return nn.Sequential(nn.Conv2d(in_dim, feat_dim, 3, padding=1), nn.ReLU(), nn.Conv2d(feat_dim, out_dim, 1))
    return nn.Sequential(nn.Conv2d(in_dim, out_dim, 1))


class _ProjUpload:
    """Uploads of the per-frame projection matrices for one (device, thread): a ring of three pinned staging buffers (the
    host only waits for the upload issued three frames ago, not for the previous frame's -- which queues behind that
    frame's kernels), and the last upload is handed out again while the matrices do not change (no augmentation: the
    same device tensor every frame, so that consumers keyed on it -- the warp gradient's plan -- see it unchanged)."""
    RING = 3

    def __init__(self):
        self.bufs, self.events, self.next = [], [], 0
        self.last_host, self.last_dev = None, None

    def upload(self, proj, dev):
        if self.last_host is not None and self.last_dev is not None and self.last_dev.device == dev \
                and self.last_host.shape == proj.shape and torch.equal(self.last_host, proj):
            return self.last_dev
        if not self.bufs or self.bufs[0].shape != proj.shape:
            self.bufs = [torch.empty_like(proj).pin_memory() for _ in range(self.RING)]
            self.events = [None] * self.RING
            self.next = 0
        i = self.next
        self.next = (i + 1) % self.RING
        if self.events[i] is not None:
            self.events[i].synchronize()                                  # that buffer's previous upload has left it
        self.bufs[i].copy_(proj)
        out = self.bufs[i].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.events[i] = ev
        self.last_host, self.last_dev = proj.clone(), out
        return out


class MVDeTr(nn.Module):
    def __init__(self, geom: geometry.SceneGeometry, Ks, Rts, arch="resnet18", z=0, world_feat_arch="deform_trans",
                 bottleneck_dim=None, outfeat_dim=0, dropout=0.5, channels_last=True):
        super().__init__()
        self.geom = geom
        self.Rimg_shape, self.Rworld_shape = geom.Rimg_shape, geom.Rworld_shape
        self.img_reduce, self.num_cam = geom.img_reduce, geom.num_cam
        self.channels_last = channels_last
        bottleneck_dim = geom.feat_channels if bottleneck_dim is None else bottleneck_dim
        # image pixel -> reduced world grid, fp64 (mvdetr.py:82-95)
        self.register_buffer("proj_mats", torch.from_numpy(geometry.build_proj_mats(geom, Ks, Rts, z)), persistent=False)
        # host copy for the per-frame composition: the buffer above follows .to(device) (it is part of the reference's
        # attribute set, mvdetr.py:95), and reading it back every forward would be a full-stream sync per frame
        self._proj_mats_host, self._proj_host_src, self._proj_host_version = None, None, None
        self._transient = {}
        if arch not in ("resnet18", "resnet50"):
            raise ValueError("trunks: resnet18 (the reference's default, mvdetr.py:102-105) or resnet50 (BASELINE configs[3]; "
                             "resnet.py:244); vgg11 is not provided")
        self.base = resnet_trunk(int(arch[6:]))
        base_dim = 512 if arch == "resnet18" else 2048
        if bottleneck_dim:
            self.bottleneck = nn.Sequential(nn.Conv2d(base_dim, bottleneck_dim, 1), nn.Dropout2d(dropout))
            base_dim = bottleneck_dim
        else:
            self.bottleneck = nn.Identity()
        self.img_heatmap = output_head(base_dim, outfeat_dim, 1)
        self.img_offset = output_head(base_dim, outfeat_dim, 2)
        self.img_wh = output_head(base_dim, outfeat_dim, 2)
        self.world_feat_arch = world_feat_arch
        if world_feat_arch == "deform_trans":
            n_points = 4
            ref = geometry.create_reference_map(geom, Ks, Rts, n_points).repeat([geom.num_cam, 1, 1, 1])
            self.world_feat = DeformTransWorldFeat(geom.num_cam, geom.Rworld_shape, base_dim, hidden_dim=base_dim,
                                                   n_points=n_points, stride=2, reference_points=ref)
        elif world_feat_arch == "conv":
            self.world_feat = ConvWorldFeat(geom.num_cam, geom.Rworld_shape, base_dim, hidden_dim=base_dim)
        else:
            raise ValueError("world_feat_arch must be 'deform_trans' or 'conv'")
        self.world_heatmap = output_head(base_dim, outfeat_dim, 1)
        self.world_offset = output_head(base_dim, outfeat_dim, 2)
        # init (mvdetr.py:142-148)
        self.img_heatmap[-1].bias.data.fill_(-2.19)
        self.world_heatmap[-1].bias.data.fill_(-2.19)
        for head in (self.img_offset, self.img_wh, self.world_offset):
            for m in head.modules():
                if isinstance(m, nn.Conv2d) and m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def frame_proj_mats(self, M, device=None):
        """[B*N,3,3] fp32 reduced world grid <- feature pixel, composed on the host in fp32 exactly like
        mvdetr.py:155-161 (M is the dataloader's augmentation matrix and lives on the CPU there).  With ``device`` the
        result is uploaded from a pinned staging buffer with a non-blocking copy: nothing in a frame waits for the GPU,
        so the host can queue frame k+1 behind frame k (the reference composes on the CPU and does a pageable
        ``.to(device)`` every forward, mvdetr.py:194)."""
        pm = self.proj_mats
        # (the source tensor itself is held and compared with `is`: an id() alone can be reused by a later tensor)
        if pm is not self._proj_host_src or pm._version != self._proj_host_version:
            self._proj_mats_host = pm.detach().cpu().clone()              # one sync, after .to(device) / in-place edits
            self._proj_host_src, self._proj_host_version = pm, pm._version
        proj = geometry.compose_frame_proj_mats(self._proj_mats_host, M.cpu(), self.img_reduce)
        if device is None or torch.device(device).type == "cpu":
            return proj
        dev = torch.device(device)
        t = self._transient.setdefault((dev.type, dev.index, threading.get_ident()), _ProjUpload())
        return t.upload(proj, dev)

    # the pinned staging buffers / events / cached uploads are per (device, thread) run-time state, not part of the model:
    # they are not pickled, deep-copied or shared between DataParallel replicas' threads
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_transient"] = {}
        d["_proj_mats_host"], d["_proj_host_src"], d["_proj_host_version"] = None, None, None
        return d

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def features(self, imgs):
        B, N, C, H, W = imgs.shape
        x = imgs.reshape(B * N, C, H, W)
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        return self.bottleneck(self.base(x))

    def forward(self, imgs, M, visualize=False):
        B, N = imgs.shape[:2]
        proj = self.frame_proj_mats(M, imgs.device)
        feat = self.features(imgs)                                         # [B*N, C, h, w]
        imgs_heatmap, imgs_offset, imgs_wh = self.img_heatmap(feat), self.img_offset(feat), self.img_wh(feat)
        H, W = self.Rworld_shape
        C = feat.shape[1]
        nhwc = self.channels_last and self.world_feat_arch == "deform_trans"
        world = warp_perspective(feat, proj, (H, W), channels_last_out=nhwc)
        world = world.view(B, N, H, W, C) if nhwc else world.view(B, N, C, H, W)
        world = self.world_feat(world)
        return (self.world_heatmap(world), self.world_offset(world)), (imgs_heatmap, imgs_offset, imgs_wh)

    def hot_path(self, feat, proj):
        """warp + shadow transformer only (the path BASELINE.json's north_star names), for timing."""
        H, W = self.Rworld_shape
        N = self.num_cam
        B = feat.shape[0] // N
        nhwc = self.channels_last and self.world_feat_arch == "deform_trans"
        world = warp_perspective(feat, proj, (H, W), channels_last_out=nhwc)
        world = world.view(B, N, H, W, -1) if nhwc else world.view(B, N, -1, H, W)
        return self.world_feat(world)


def build_model(config="wildtrack", seed=0, world_feat_arch="deform_trans", **kw) -> MVDeTr:
    geom = geometry.GEOMETRIES[config]
    Ks, Rts = geometry.synthetic_rig(geom, seed=seed)
    torch.manual_seed(seed)
    return MVDeTr(geom, Ks, Rts, world_feat_arch=world_feat_arch, **kw)
