#!/usr/bin/env python
"""Golden vectors of the reference's OWN CPU voxeliser and max-pool loops, executed
here through oracle/_ref (rendered from spconv/csrc/sparse/pointops.py:493-766 and maxpool.py:590-703 where they lie;
needs /root/reference):

    python tests/golden/make_ref_8f_golden.py     ->  tests/golden/p2v_ref.npz, tests/golden/pool_ref.npz

p2v_ref.npz   a seeded clustered point cloud (several points per voxel, points outside the range, both caps hit)
              and what Point2VoxelCPU::point_to_voxel_static / point_to_voxel_empty_mean_static returned for it.
pool_ref.npz  a strided rulebook (k3 s2 p1) of a seeded scene, dyadic fp32 features and output gradients (so that
              sums are exact in any order), and what the driver loops of pytorch/ops.py:1899-1975 over
              IndiceMaxPoolCPU::forward / backward returned; plus global_pool_rearrange of a batched index list
              with deleted rows.
tests/test_oracle.py checks the restatements in oracle/ against them bit for bit on any box."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from oracle import ref  # noqa: E402


def cloud(n, seed, nfeat=4):
    rng = np.random.default_rng(seed)
    lo, hi = (-1.0, -5.0, -3.0) + (0.0,) * (nfeat - 3), (9.0, 5.0, 3.0) + (1.0,) * (nfeat - 3)
    pts = rng.uniform(lo, hi, (n, nfeat)).astype(np.float32)
    pts[: n // 2, :3] = pts[: n // 2, :3] * 0.05 + np.array([4.0, 0.0, 0.0], dtype=np.float32)   # clusters
    return pts


def p2v_meta(vsize_xyz, range_xyz):
    """Point2VoxelCPU's constructor arithmetic (pointops.py:541-567): zyx order, float32, std::round."""
    vs = np.asarray(vsize_xyz, dtype=np.float32)[::-1].copy()
    lo = np.asarray(range_xyz[:3], dtype=np.float32)[::-1].copy()
    hi = np.asarray(range_xyz[3:], dtype=np.float32)[::-1].copy()
    grid = np.round((hi - lo) / vs).astype(np.int64)
    return vs, np.concatenate([lo, hi]), grid


def main():
    assert ref.build() is not None, "oracle/_ref needs /root/reference"
    out = {}
    vs, cr, grid = p2v_meta([0.1, 0.1, 0.2], [0, -4, -2, 8, 4, 2])
    pts = cloud(6000, 1)
    out.update(points=pts, vsize=vs, coors_range=cr, grid_size=grid.astype(np.int32))
    for tag, mv, mp, mean in (("plain", 8000, 5, False), ("mean", 8000, 5, True), ("capped_mean", 300, 3, True)):
        v, i, c, pid = ref.point2voxel(pts, vs, cr, grid, mv, mp, mean)
        out[f"{tag}_args"] = np.array([mv, mp, int(mean)], dtype=np.int32)
        out[f"{tag}_voxels"], out[f"{tag}_indices"], out[f"{tag}_num"], out[f"{tag}_pid"] = v, i, c, pid
    np.savez_compressed(os.path.join(HERE, "p2v_ref.npz"), **out)

    rng = np.random.default_rng(703)
    shape, n, C = [20, 24, 28], 2500, 6
    lin = rng.choice(2 * int(np.prod(shape)), n, replace=False)
    b, rest = np.divmod(lin, int(np.prod(shape)))
    idx = np.concatenate([b[:, None], np.stack(np.unravel_index(rest, shape), axis=-1)], axis=1).astype(np.int32)
    out_inds, pair, num, _ = oracle.get_indice_pairs(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, None, False, False)
    rout_inds, rpair, rnum, _ = ref.get_indice_pairs(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, None, False, False)
    assert np.array_equal(pair, rpair) and np.array_equal(num, rnum) and np.array_equal(out_inds, rout_inds)
    n_out = out_inds.shape[0]
    f = (rng.integers(-64, 65, (n, C)) / 64.0).astype(np.float32)
    f[7::7] = f[6::7][: f[7::7].shape[0]]                    # ties: several inputs reach the maximum
    dout = (rng.integers(-32, 33, (n_out, C)) / 64.0).astype(np.float32)
    o = ref.indice_maxpool(f, pair, num, n_out)
    din = ref.indice_maxpool_backward(f, o, dout, pair, num)
    coords = idx.copy()
    coords[::11, 0] = -1                                     # deleted rows belong to no scene
    gp_out, gp_cnt = ref.global_pool_rearrange(coords, 2)
    np.savez_compressed(os.path.join(HERE, "pool_ref.npz"), indices=idx, shape=np.array(shape, dtype=np.int32),
                        pair=pair, num=num, n_out=np.int32(n_out), features=f, dout=dout, out=o, din=din,
                        gp_coords=coords, gp_out=gp_out, gp_counts=gp_cnt)
    for name in ("p2v_ref.npz", "pool_ref.npz"):
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)))


if __name__ == "__main__":
    main()
