"""GPU voxeliser against the sequential CPU restatement of the
reference's Point2VoxelCPU (oracle.point2voxel): bit-exact voxels, indices (zyx), counts and
per-point voxel ids, including the max_voxels / max_points caps and points outside the range."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _cloud(n, seed, lo=(-1.0, -5.0, -3.0), hi=(9.0, 5.0, 3.0), nfeat=4):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(lo + (0.0,) * (nfeat - 3), hi + (1.0,) * (nfeat - 3), (n, nfeat)).astype(np.float32)
    # clusters -> several points per voxel
    pts[: n // 2, :3] = pts[: n // 2, :3] * 0.05 + np.array([4.0, 0.0, 0.0], dtype=np.float32)
    return pts


@pytest.mark.parametrize("n,max_voxels,max_points,empty_mean",
                         [(20000, 40000, 5, False), (20000, 40000, 5, True), (20000, 300, 3, True),
                          (50, 100, 1, False), (0, 10, 2, False)])
def test_point2voxel_bit_exact(cuda, n, max_voxels, max_points, empty_mean):
    from spconv_amd.pytorch.utils import PointToVoxel
    vsize, rng_xyz = [0.1, 0.1, 0.2], [0, -4, -2, 8, 4, 2]
    gen = PointToVoxel(vsize, rng_xyz, 4, max_voxels, max_points, device=cuda)
    pts = _cloud(n, 1)
    v, i, c, pid = gen.generate_voxel_with_id(torch.from_numpy(pts).to(cuda), True, empty_mean)
    rv, ri, rc, rpid = oracle.point2voxel(pts, gen.vsize, gen.coors_range, gen.grid_size, max_voxels,
                                          max_points, empty_mean)
    assert v.shape[0] == rv.shape[0]
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(c.cpu().numpy(), rc)
    np.testing.assert_array_equal(pid.cpu().numpy(), rpid)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)
    if n:
        assert (rpid == -1).any() and rc.max() == max_points      # caps and range checks are exercised


def test_point2voxel_grid_beyond_32_bits(cuda):
    """A grid with more than 2**32 cells takes the wide-key form of the hash table."""
    from spconv_amd.pytorch.utils import PointToVoxel
    vsize, rng_xyz = [0.003, 0.004, 0.004], [0, -4, -2, 8, 4, 2]      # 2667 x 2000 x 1000 cells
    gen = PointToVoxel(vsize, rng_xyz, 4, 30000, 4, device=cuda)
    assert int(np.prod(np.asarray(gen.grid_size, dtype=np.int64))) > 2 ** 32
    pts = _cloud(20000, 5)
    pts[:10000, :3] = pts[10000:, :3] + 1e-4                          # several points per voxel
    v, i, c, pid = gen.generate_voxel_with_id(torch.from_numpy(pts).to(cuda), True, True)
    rv, ri, rc, rpid = oracle.point2voxel(pts, gen.vsize, gen.coors_range, gen.grid_size, 30000, 4, True)
    np.testing.assert_array_equal(i.cpu().numpy(), ri)
    np.testing.assert_array_equal(c.cpu().numpy(), rc)
    np.testing.assert_array_equal(pid.cpu().numpy(), rpid)
    np.testing.assert_array_equal(v.cpu().numpy(), rv)


def test_point2voxel_feeds_a_sparse_conv(cuda):
    """points -> voxels -> mean features -> SparseConvTensor -> SubMConv3d: the pipeline of
    docs/USAGE.md (indices ZYX, batch column prepended)."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.utils import PointToVoxel, gather_features_by_pc_voxel_id
    gen = PointToVoxel([0.2, 0.2, 0.4], [0, -4, -2, 8, 4, 2], 4, 20000, 4, device=cuda)
    pts = torch.from_numpy(_cloud(8000, 2)).to(cuda)
    voxels, coords, num, pid = gen.generate_voxel_with_id(pts, empty_mean=True)
    feats = voxels.sum(dim=1) / num.clamp_min(1).unsqueeze(1).float() * 0 + voxels[:, 0]   # first point
    idx = torch.cat([torch.zeros_like(coords[:, :1]), coords], dim=1).contiguous()
    x = spconv.SparseConvTensor(feats.half(), idx, gen.grid_size, 1)
    y = spconv.SubMConv3d(4, 16, 3).to(cuda).half()(x)
    assert y.features.shape == (voxels.shape[0], 16)
    back = gather_features_by_pc_voxel_id(y.features, pid, invalid_value=-1)
    assert back.shape == (8000, 16) and bool((back[pid < 0] == -1).all())
    assert torch.equal(back[pid >= 0], y.features[pid[pid >= 0]])


@pytest.mark.parametrize("tag", ["plain", "mean", "capped_mean"])
def test_point2voxel_equals_the_reference_code_executed(cuda, tag, monkeypatch):
    """tests/golden/p2v_ref.npz: what the reference's own Point2VoxelCPU (pointops.py:493-766) returned.  Voxel set,
    numbering, stored points, counts and per-point ids are identical by default; with SPCONV_AMD_REFERENCE_QUIRKS=1
    the mean fill of the empty slots is too (the reference's accumulator is carried from voxel to voxel)."""
    import os
    from spconv_amd import constants
    from spconv_amd.pytorch.utils import PointToVoxel
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "p2v_ref.npz"))
    mv, mp, mean = (int(v) for v in d[f"{tag}_args"])
    gen = PointToVoxel([0.1, 0.1, 0.2], [0, -4, -2, 8, 4, 2], 4, mv, mp, device=cuda)
    np.testing.assert_array_equal(np.asarray(gen.grid_size), d["grid_size"])
    np.testing.assert_array_equal(np.asarray(gen.vsize, dtype=np.float32), d["vsize"])
    pts = torch.from_numpy(d["points"]).to(cuda)
    monkeypatch.setattr(constants, "REFERENCE_QUIRKS", True)
    v, i, c, pid = gen.generate_voxel_with_id(pts, True, bool(mean))
    np.testing.assert_array_equal(i.cpu().numpy(), d[f"{tag}_indices"])
    np.testing.assert_array_equal(c.cpu().numpy(), d[f"{tag}_num"])
    np.testing.assert_array_equal(pid.cpu().numpy(), d[f"{tag}_pid"])
    np.testing.assert_array_equal(v.cpu().numpy(), d[f"{tag}_voxels"])
    monkeypatch.setattr(constants, "REFERENCE_QUIRKS", False)
    v2, i2, c2, pid2 = gen.generate_voxel_with_id(pts, True, bool(mean))
    np.testing.assert_array_equal(i2.cpu().numpy(), d[f"{tag}_indices"])
    np.testing.assert_array_equal(pid2.cpu().numpy(), d[f"{tag}_pid"])
    stored = (np.arange(mp)[None, :] < d[f"{tag}_num"][:, None])
    np.testing.assert_array_equal(v2.cpu().numpy()[stored], d[f"{tag}_voxels"][stored])


def test_point2voxel_key_order_is_the_first_seen_result_renumbered(cuda):
    """PointToVoxel(..., key_order=True) (not in the reference, whose voxels come out in point / hash-slot order:
    pytorch/utils.py:23-160): the same voxels, numbered by ascending (z, y, x) key; pc_voxel_id follows; two scenes
    concatenated in batch order are in coordinate-key order, and a first-level SubM rulebook built from their rank map
    (ops.attach_rank_map: no hash table) equals the hash build bit for bit."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import ops
    from spconv_amd.pytorch.utils import PointToVoxel
    args = ([0.2, 0.2, 0.4], [0, -4, -2, 8, 4, 2], 4, 20000, 4)
    ref_gen, gen = PointToVoxel(*args, device=cuda), PointToVoxel(*args, device=cuda, key_order=True)
    scenes = []
    for seed in (2, 3):
        pts = torch.from_numpy(_cloud(8000, seed)).to(cuda)
        v0, c0, n0, p0 = ref_gen.generate_voxel_with_id(pts, empty_mean=True)
        v1, c1, n1, p1 = gen.generate_voxel_with_id(pts, empty_mean=True)
        Z, Y, X = gen.grid_size
        k0 = (c0[:, 0].long() * Y + c0[:, 1].long()) * X + c0[:, 2].long()
        k1 = (c1[:, 0].long() * Y + c1[:, 1].long()) * X + c1[:, 2].long()
        assert bool((k1[1:] > k1[:-1]).all())                      # ascending, unique
        order = torch.argsort(k0)
        assert torch.equal(c1, c0[order]) and torch.equal(v1, v0[order]) and torch.equal(n1, n0[order])
        assert torch.equal(p1 < 0, p0 < 0)
        inside = p0 >= 0
        assert torch.equal(c1[p1[inside]], c0[p0[inside]])         # every point still lands in its voxel
        scenes.append(c1)
    idx = torch.cat([torch.cat([torch.full_like(c[:, :1], b), c], dim=1) for b, c in enumerate(scenes)]).contiguous()
    rb_hash, _ = ops.build_rulebook(idx.clone(), 2, gen.grid_size, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
    assert ops.attach_rank_map(idx, 2, gen.grid_size, check=True)
    rb_rank, _ = ops.build_rulebook(idx, 2, gen.grid_size, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
    assert torch.equal(rb_rank.pair_fwd, rb_hash.pair_fwd) and torch.equal(rb_rank.mask_fwd, rb_hash.mask_fwd)
