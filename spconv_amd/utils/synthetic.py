"""Seeded synthetic voxel scenes for tests and the benchmark (no dataset access).

* ``uniform_scene``: distinct voxels drawn uniformly from the grid -- the
  configuration BASELINE.json quotes (KITTI-shape 40x1280x1600, ~100k voxels,
  SURVEY.md section 8d cfg 2).  At that occupancy (0.12 %) a 3x3x3 SubM rulebook
  has ~1.03 pairs per voxel.
* ``lidar_like_scene``: voxels on a ground sheet plus box-shaped objects, which
  gives the neighbourhood density of real LiDAR sweeps (~6 pairs per voxel,
  cf. the reference fixture test/data/test_spconv.pkl: 6.28).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


def _with_batch(coords: np.ndarray, b: int) -> np.ndarray:
    return np.concatenate([np.full((coords.shape[0], 1), b, dtype=np.int32),
                           coords.astype(np.int32)], axis=1)


def uniform_scene(shape: Sequence[int], num_voxels: int, batch_size: int = 1,
                  seed: int = 0) -> np.ndarray:
    """int32 [batch_size * num_voxels, ndim + 1] rows (b, z, y, x), distinct per scene."""
    out = []
    vol = int(np.prod(shape))
    for b in range(batch_size):
        rng = np.random.default_rng(seed + b)
        lin = rng.choice(vol, size=min(num_voxels, vol), replace=False)
        coords = np.stack(np.unravel_index(lin, shape), axis=-1)
        out.append(_with_batch(coords, b))
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def lidar_like_scene(shape: Sequence[int] = (40, 1280, 1600), num_voxels: int = 100_000,
                     batch_size: int = 1, seed: int = 0) -> np.ndarray:
    """Surface-like occupancy: an undulating ground sheet (about 60 % of the voxels) and
    axis-aligned hollow boxes (cars / walls).  Voxels are distinct within a scene and
    ordered by a shuffled sweep, like a voxeliser's output."""
    Z, Y, X = [int(s) for s in shape]
    out = []
    for b in range(batch_size):
        rng = np.random.default_rng(seed + 1000 + b)
        lin = np.zeros((0,), dtype=np.int64)
        while lin.shape[0] < num_voxels:
            pts = []
            for _ in range(max(1, num_voxels // 4000)):      # ground patches
                cy, cx = rng.integers(0, Y), rng.integers(0, X)
                h, w = rng.integers(30, 80), rng.integers(30, 80)
                ys = np.arange(max(0, cy - h // 2), min(Y, cy + h // 2))
                xs = np.arange(max(0, cx - w // 2), min(X, cx + w // 2))
                yy, xx = np.meshgrid(ys, xs, indexing="ij")
                zz = (2 + np.sin(yy / 37.0) + np.cos(xx / 53.0)).astype(np.int64).clip(0, Z - 1)
                keep = rng.random(yy.shape) < 0.8
                pts.append(np.stack([zz[keep], yy[keep], xx[keep]], axis=-1))
            for _ in range(max(1, num_voxels // 1500)):      # hollow boxes
                cz = rng.integers(2, max(3, Z - 8))
                cy, cx = rng.integers(0, Y - 20), rng.integers(0, X - 20)
                dz, dy, dx = rng.integers(3, 8), rng.integers(6, 20), rng.integers(6, 20)
                zz, yy, xx = np.meshgrid(np.arange(cz, min(Z, cz + dz)), np.arange(cy, cy + dy),
                                         np.arange(cx, cx + dx), indexing="ij")
                shell = ((zz == zz.min()) | (zz == zz.max()) | (yy == yy.min())
                         | (yy == yy.max()) | (xx == xx.min()) | (xx == xx.max()))
                keep = shell & (rng.random(zz.shape) < 0.7)
                pts.append(np.stack([zz[keep], yy[keep], xx[keep]], axis=-1))
            c = np.concatenate(pts, axis=0)
            new = np.ravel_multi_index((c[:, 0], c[:, 1], c[:, 2]), (Z, Y, X))
            lin = np.unique(np.concatenate([lin, new]))
        # drop whole trailing structures rather than random voxels: keeps neighbourhoods dense
        lin = lin[:num_voxels] if lin.shape[0] < num_voxels * 1.02 else rng.permutation(lin)[:num_voxels]
        rng.shuffle(lin)
        coords = np.stack(np.unravel_index(lin, (Z, Y, X)), axis=-1)
        out.append(_with_batch(coords, b))
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def random_features(n: int, channels: int, seed: int = 0, low: float = -1.0,
                    high: float = 1.0) -> np.ndarray:
    return np.random.default_rng(seed + 77).uniform(low, high, size=(n, channels)).astype(np.float32)


def random_weight(out_channels: int, ksize: Sequence[int], in_channels: int,
                  seed: int = 0) -> np.ndarray:
    """KRSC weight, U(-1, 1) like the reference's tests (test/test_all_algo.py:200)."""
    shape = (out_channels, *ksize, in_channels)
    return np.random.default_rng(seed + 99).uniform(-1, 1, size=shape).astype(np.float32)
