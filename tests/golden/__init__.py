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


# ---- vectors produced by executing the reference's own CPU code (make_ref_golden.py) ----
def ref_case_names():
    return sorted(f[4:-4] for f in os.listdir(HERE) if f.startswith("ref_") and f.endswith(".npz"))


def load_ref_case(name):
    d = dict(np.load(os.path.join(HERE, f"ref_{name}.npz")))
    for k in ("shape", "ksize", "stride", "pad", "dil", "out_shape"):
        d[k] = [int(v) for v in d[k]]
    d["bs"], d["subm"], d["transposed"] = int(d["bs"]), bool(d["subm"]), bool(d["transposed"])
    return d


def ref_digests():
    import json
    with open(os.path.join(HERE, "ref_digests.json")) as f:
        return json.load(f)


def digest(*arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def ref_big_inputs(name):
    """Inputs of a ref_digests.json entry, regenerated from seeds / the committed fixture; chain
    levels need the previous level's output coordinates, which the caller supplies."""
    from spconv_amd.utils import synthetic
    if name == "cfg1_subm":
        return synthetic.uniform_scene([64, 64, 64], 5000, 1, seed=0), [64, 64, 64]
    if name == "cfg2_subm":
        return synthetic.uniform_scene([40, 1280, 1600], 100_000, 1, seed=0), [40, 1280, 1600]
    if name in ("fixture_subm", "fixture_chain_l0"):
        return lidar_scene()
    raise KeyError(name)
