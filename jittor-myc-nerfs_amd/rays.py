"""Blender-format camera -> rays [N,6] (origin, direction), the input format of the render path.

Restates the ray-generation arithmetic of the reference loader so that synthetic 800x800 frames
have exactly the layout `OctreeRender_trilinear_fast` receives from `BlenderDataset`:
  * pixel directions      tensorf-myc/dataLoader/ray_utils.py:81-103  (get_ray_directions)
  * world rays            tensorf-myc/dataLoader/ray_utils.py:132-153 (get_rays)
  * focal / pose handling tensorf-myc/dataLoader/blender.py:33,69-76,91
"""
from __future__ import annotations

import json
import math
from typing import List, Sequence, Tuple

import numpy as np
import torch

BLENDER2OPENCV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64)  # blender.py:33


def focal_from_angle(camera_angle_x: float, width: int) -> float:
    """blender.py:69-70: focal = 0.5*800/tan(0.5*angle), rescaled by width/800."""
    return 0.5 * 800 / math.tan(0.5 * camera_angle_x) * (width / 800)


def get_ray_directions(H: int, W: int, focal: Sequence[float], center=None) -> np.ndarray:
    """(H,W,3) camera-space directions [-(i+.5-cx)/fx, (j+.5-cy)/fy, -1]  (ray_utils.py:91-101).

    fp32 element-wise numpy arithmetic only (no BLAS, no transcendental): IEEE add/mul/div/sqrt give the same
    bits on every host, so golden vectors generated in one container hold on the GPU box's CPU too."""
    i = (np.arange(W, dtype=np.float32) + np.float32(0.5))[None, :].repeat(H, 0)
    j = (np.arange(H, dtype=np.float32) + np.float32(0.5))[:, None].repeat(W, 1)
    cent = center if center is not None else [W / 2, H / 2]
    x = -(i - np.float32(cent[0])) / np.float32(focal[0])
    y = (j - np.float32(cent[1])) / np.float32(focal[1])
    return np.stack([x, y, -np.ones_like(x)], -1)


def get_rays(directions: np.ndarray, c2w: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """ray_utils.py:132-153: rays_d = directions @ c2w[:3,:3]^T (written as three fp32 multiply-adds, in k order),
    rays_o = c2w[:3,3] broadcast."""
    Rm = np.asarray(c2w, dtype=np.float32)[:3, :3]
    d = directions.astype(np.float32)
    rays_d = d[..., 0:1] * Rm[:, 0] + d[..., 1:2] * Rm[:, 1]
    rays_d = rays_d + d[..., 2:3] * Rm[:, 2]
    rays_o = np.broadcast_to(np.asarray(c2w, dtype=np.float32)[:3, 3], rays_d.shape)
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)


def frame_rays(transform_matrix, H: int, W: int, camera_angle_x: float) -> torch.Tensor:
    """One frame of a transforms_*.json -> rays [H*W,6] fp32 (blender.py:69-76,91,116-117)."""
    focal = focal_from_angle(camera_angle_x, W)
    dirs = get_ray_directions(H, W, [focal, focal])
    nrm = np.sqrt(dirs[..., 0] * dirs[..., 0] + dirs[..., 1] * dirs[..., 1] + dirs[..., 2] * dirs[..., 2])
    dirs = dirs / nrm[..., None]                                            # blender.py:75
    pose = np.asarray(transform_matrix, dtype=np.float64) @ BLENDER2OPENCV  # blender.py:91 (4x4, exact sign flips)
    o, d = get_rays(dirs, pose.astype(np.float32))
    return torch.from_numpy(np.ascontiguousarray(np.concatenate([o, d], 1), dtype=np.float32))


def load_transforms(path: str):
    """transforms_{split}.json: {"camera_angle_x", "frames":[{"file_path","transform_matrix"}]} (blender.py:63-66)."""
    with open(path) as f:
        meta = json.load(f)
    return float(meta["camera_angle_x"]), [np.asarray(fr["transform_matrix"], dtype=np.float64) for fr in meta["frames"]]


def sphere_poses(n: int, radius: float, elevation_deg: float = 30.0) -> List[np.ndarray]:
    """n synthetic `transform_matrix` entries on a sphere, looking at the origin.

    The reference pipeline (pose @ BLENDER2OPENCV, directions [-x, y, -1]) yields
    rays_d = -x*M[:,0] - y*M[:,1] + M[:,2] in terms of the json matrix M, so a matrix whose
    columns are (-right, up, forward, position) makes the centre pixel look along `forward`.
    """
    out = []
    el = math.radians(elevation_deg)
    for k in range(n):
        az = 2 * math.pi * k / n + 0.3
        pos = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
        fwd = -pos / np.linalg.norm(pos)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        M = np.eye(4)
        M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3] = -right, up, fwd, pos
        out.append(M)
    return out
