"""Loaders for the committed golden fixtures (see make_golden.py)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_case(name):
    return dict(np.load(os.path.join(HERE, f"case_{name}.npz")))


def case_names():
    return sorted(f[5:-4] for f in os.listdir(HERE) if f.startswith("case_") and f.endswith(".npz"))


def lidar_scene(shuffle_seed=0):
    """int32 [125562, 4] (b=0, z, y, x) coordinates of the reference's real-LiDAR fixture
    (test/data/test_spconv.pkl), in a seeded shuffled order, and the spatial shape."""
    d = np.load(os.path.join(HERE, "lidar_scene.npz"))
    shape = [int(v) for v in d["shape"]]
    lin = np.cumsum(d["delta"].astype(np.int64))
    if shuffle_seed is not None:
        np.random.default_rng(shuffle_seed).shuffle(lin)
    coords = np.stack(np.unravel_index(lin, shape), axis=-1).astype(np.int32)
    idx = np.concatenate([np.zeros((coords.shape[0], 1), dtype=np.int32), coords], axis=1)
    return np.ascontiguousarray(idx), shape
