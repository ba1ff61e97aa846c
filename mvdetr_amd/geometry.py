"""Ground-plane geometry of the multiview fusion path (host side, numpy fp64).

Everything here is tiny 3x3 algebra that runs once per model (or once per
frame for the augmentation matrices); it produces the inputs of the two HIP
kernels: the per-view homographies of the warp and the reference points of
the shadow transformer.

Reference (read-only, cited for parity):
  * multiview_detector/utils/projection.py:4-14   project_2d_points
  * multiview_detector/utils/projection.py:27-43  get_{img,world}coord_from_*_mat
  * multiview_detector/models/mvdetr.py:33-71     create_reference_map
  * multiview_detector/models/mvdetr.py:82-95     MVDeTr.__init__ proj_mats
  * multiview_detector/models/mvdetr.py:155-161   per-forward composition with the
                                                  augmentation matrices
  * multiview_detector/datasets/Wildtrack.py:21-34, MultiviewX.py:21-34 (constants)
  * multiview_detector/datasets/frameDataset.py:66-71 (Rworld_shape / Rimg_shape)

There are no dataset files on either box, so cameras come from
``synthetic_rig`` (pinhole cameras on a ring around the ground plane, SURVEY
section 8d), not from calibration XML.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

_SWAP_XY = np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])


@dataclass(frozen=True)
class SceneGeometry:
    """The constants a dataset contributes to the path (no files, no images)."""

    name: str
    num_cam: int
    img_shape: Tuple[int, int]            # full-resolution camera image (H, W)
    worldgrid_shape: Tuple[int, int]      # ground grid (N_row, N_col)
    indexing: str                         # 'ij' (Wildtrack) or 'xy' (MultiviewX)
    worldcoord_unit: float                # metres per world-coordinate unit
    worldcoord_from_worldgrid_mat: np.ndarray = field(repr=False)
    world_reduce: int = 4
    img_reduce: int = 12
    feat_channels: int = 128

    @property
    def world_indexing_from_xy_mat(self) -> np.ndarray:
        # Wildtrack.py:28 / MultiviewX.py:26
        return _SWAP_XY.copy() if self.indexing == "ij" else np.eye(3)

    @property
    def Rworld_shape(self) -> Tuple[int, int]:
        # frameDataset.py:70
        return (self.worldgrid_shape[0] // self.world_reduce,
                self.worldgrid_shape[1] // self.world_reduce)

    @property
    def Rimg_shape(self) -> Tuple[int, int]:
        # frameDataset.py:71
        return (int(math.ceil(self.img_shape[0] / self.img_reduce)),
                int(math.ceil(self.img_shape[1] / self.img_reduce)))

    @property
    def input_img_shape(self) -> Tuple[int, int]:
        # frameDataset.py:66-67: images are resized to img_shape * 8 // img_reduce
        return (self.img_shape[0] * 8 // self.img_reduce,
                self.img_shape[1] * 8 // self.img_reduce)


WILDTRACK = SceneGeometry(
    name="wildtrack", num_cam=7, img_shape=(1080, 1920), worldgrid_shape=(480, 1440),
    indexing="ij", worldcoord_unit=0.01,
    worldcoord_from_worldgrid_mat=np.array([[2.5, 0, -300], [0, 2.5, -900], [0, 0, 1.0]]))

MULTIVIEWX = SceneGeometry(
    name="multiviewx", num_cam=6, img_shape=(1080, 1920), worldgrid_shape=(640, 1000),
    indexing="xy", worldcoord_unit=1.0,
    worldcoord_from_worldgrid_mat=np.array([[0.025, 0, 0], [0, 0.025, 0], [0, 0, 1.0]]))

# BASELINE.json config 5: 16 cameras at 1080p input (img_reduce 8 keeps the
# full-resolution frame), 256-channel world features on the Wildtrack grid.
STRESS16 = SceneGeometry(
    name="stress16", num_cam=16, img_shape=(1080, 1920), worldgrid_shape=(480, 1440),
    indexing="ij", worldcoord_unit=0.01,
    worldcoord_from_worldgrid_mat=np.array([[2.5, 0, -300], [0, 2.5, -900], [0, 0, 1.0]]),
    img_reduce=8, feat_channels=256)

# small scene for tests / smoke runs: 3 cameras, 144x256 inputs, 32-channel features, 24x72 world grid
MINI = SceneGeometry(
    name="mini", num_cam=3, img_shape=(216, 384), worldgrid_shape=(96, 288),
    indexing="ij", worldcoord_unit=0.01,
    worldcoord_from_worldgrid_mat=np.array([[12.5, 0, -300], [0, 12.5, -900], [0, 0, 1.0]]),
    feat_channels=32)

GEOMETRIES = {g.name: g for g in (WILDTRACK, MULTIVIEWX, STRESS16, MINI)}


def project_2d_points(project_mat: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """Apply a 3x3 homography to ``pts [K, 2]`` (projection.py:4-14)."""
    pts = np.asarray(pts, dtype=np.float64)
    hom = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1) @ np.asarray(project_mat).T
    return hom[:, :2] / hom[:, 2:3]


def get_imgcoord_from_worldcoord_mat(K: np.ndarray, Rt: np.ndarray, z: float = 0.0) -> np.ndarray:
    """3x3 map world (x, y, 1) on the plane at height z -> image pixel (projection.py:27-34)."""
    lift = np.array([[1, 0, 0], [0, 1, 0], [0, 0, z], [0, 0, 1.0]])
    return np.asarray(K, dtype=np.float64) @ np.asarray(Rt, dtype=np.float64) @ lift


def get_worldcoord_from_imgcoord_mat(K: np.ndarray, Rt: np.ndarray, z: float = 0.0) -> np.ndarray:
    """Inverse of the above (projection.py:37-43)."""
    return np.linalg.inv(get_imgcoord_from_worldcoord_mat(K, Rt, z))


def synthetic_rig(geom: SceneGeometry, seed: int = 0) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """Pinhole cameras on a ring around the ground plane, looking at its centre.

    Returns (intrinsics [3x3], extrinsics [3x4]) lists in the dataset's world
    units. K follows SURVEY 8d: f = 1743 px, principal point at the image centre.
    """
    rng = np.random.default_rng(seed)
    H, W = geom.img_shape
    f = 1743.0 * W / 1920.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    # extent of the plane in world coordinates (grid corner (0,0) and (N_row, N_col))
    g = geom.worldcoord_from_worldgrid_mat
    n0, n1 = geom.worldgrid_shape if geom.indexing == "ij" else geom.worldgrid_shape[::-1]
    lo = g @ np.array([0.0, 0.0, 1.0])
    hi = g @ np.array([float(n0), float(n1), 1.0])
    centre = np.array([(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 0.0])
    half = np.array([(hi[0] - lo[0]) / 2, (hi[1] - lo[1]) / 2])
    metre = 1.0 / geom.worldcoord_unit
    Ks, Rts = [], []
    for cam in range(geom.num_cam):
        ang = 2 * math.pi * (cam + 0.25 * rng.random()) / geom.num_cam
        radius = 1.15 + 0.2 * rng.random()
        height = (2.0 + 2.0 * rng.random()) * metre
        eye = np.array([centre[0] + radius * half[0] * math.cos(ang),
                        centre[1] + radius * half[1] * math.sin(ang), height])
        target = centre + np.array([0.15 * half[0] * (rng.random() - 0.5),
                                    0.15 * half[1] * (rng.random() - 0.5), 0.0])
        fwd = target - eye
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])           # world -> camera (x right, y down, z forward)
        t = -R @ eye
        Ks.append(K.copy())
        Rts.append(np.concatenate([R, t[:, None]], axis=1))
    return Ks, Rts


def _Rworldgrid_from_worldcoord(geom: SceneGeometry, zoom: float) -> np.ndarray:
    # mvdetr.py:45-47 and 82-84
    zoom_mat = np.diag([zoom, zoom, 1.0])
    return np.linalg.inv(geom.worldcoord_from_worldgrid_mat @ zoom_mat @ geom.world_indexing_from_xy_mat)


def build_proj_mats(geom: SceneGeometry, Ks: Sequence[np.ndarray], Rts: Sequence[np.ndarray],
                    z: float = 0.0) -> np.ndarray:
    """``[num_cam, 3, 3]`` fp64: reduced world grid (x, y) <- full-resolution image pixel.

    Restates MVDeTr.__init__ (mvdetr.py:82-95).
    """
    to_grid = _Rworldgrid_from_worldcoord(geom, geom.world_reduce)
    return np.stack([to_grid @ get_worldcoord_from_imgcoord_mat(Ks[c], Rts[c], z / geom.worldcoord_unit)
                     for c in range(geom.num_cam)])


def compose_frame_proj_mats(proj_mats, affine_mats, img_reduce: int):
    """Per-forward composition (mvdetr.py:155-161), done in fp32 torch like the reference.

    ``proj_mats [N,3,3]`` (fp64), ``affine_mats [B,N,3,3]`` (the augmentation matrices M,
    image <- augmented image) -> ``[B*N, 3, 3]`` fp32: reduced world grid <- feature-map pixel.
    """
    import torch

    B, N = affine_mats.shape[:2]
    inv_aff = torch.inverse(affine_mats.reshape(B * N, 3, 3).float())
    img_from_Rimg = inv_aff @ torch.diag(torch.tensor([float(img_reduce), float(img_reduce), 1.0])
                                         ).view(1, 3, 3).repeat(B * N, 1, 1)
    pm = torch.as_tensor(proj_mats).repeat(B, 1, 1, 1).view(B * N, 3, 3).float()
    return pm @ img_from_Rimg


def create_reference_map(geom: SceneGeometry, Ks: Sequence[np.ndarray], Rts: Sequence[np.ndarray],
                         n_points: int = 4, downsample: int = 2):
    """Reference points ``[H*W, num_cam, n_points, 2]`` (fp32 torch), normalised (x, y).

    Restates create_reference_map (mvdetr.py:33-71): every cell of the token map
    (Rworld_shape // downsample) is lifted to height z on each camera's ray and dropped back
    to the ground plane. With the wired-up ``n_points == 4`` all heights are 0, which makes the
    map the identity pixel-centre grid for every camera and point.
    """
    import torch

    H, W = geom.Rworld_shape
    H, W = H // downsample, W // downsample
    ys, xs = np.meshgrid(np.linspace(0.5, H - 0.5, H, dtype=np.float32),
                         np.linspace(0.5, W - 0.5, W, dtype=np.float32), indexing="ij")
    ref = np.stack([xs, ys], -1).reshape(-1, 2)
    if n_points == 4:
        zs = [0, 0, 0, 0]
    elif n_points == 8:
        zs = [-0.4, -0.2, 0, 0, 0.2, 0.4, 1, 1.8]
    else:
        raise ValueError("n_points must be 4 or 8")
    to_grid = _Rworldgrid_from_worldcoord(geom, geom.world_reduce * downsample)
    out = torch.zeros([H * W, geom.num_cam, n_points, 2])
    for cam in range(geom.num_cam):
        mat_0 = to_grid @ get_worldcoord_from_imgcoord_mat(Ks[cam], Rts[cam])
        for i, z in enumerate(zs):
            mat_z = to_grid @ get_worldcoord_from_imgcoord_mat(Ks[cam], Rts[cam], z / geom.worldcoord_unit)
            img_pts = project_2d_points(np.linalg.inv(mat_z), ref)
            out[:, cam, i, :] = torch.from_numpy(project_2d_points(mat_0, img_pts))
    out[..., 0] /= W
    out[..., 1] /= H
    return out


def random_affine_mats(batch: int, num_cam: int, img_hw: Tuple[int, int], seed: int = 0,
                       translate: float = 0.2, scale: Tuple[float, float] = (0.6, 1.4)):
    """Augmentation matrices ``M [B, N, 3, 3]`` of the shape random_affine produces
    (image_utils.py:9-35 with the defaults: no rotation, no shear): a scale about the
    image centre plus a translation of up to ``translate`` of the image size."""
    import torch

    rng = np.random.default_rng(seed)
    H, W = img_hw
    mats = np.zeros((batch, num_cam, 3, 3))
    for b in range(batch):
        for c in range(num_cam):
            s = rng.uniform(*scale)
            tx = rng.uniform(-translate, translate) * W
            ty = rng.uniform(-translate, translate) * H
            mats[b, c] = np.array([[s, 0, (1 - s) * W / 2 + tx], [0, s, (1 - s) * H / 2 + ty], [0, 0, 1.0]])
    return torch.from_numpy(mats).float()
