"""GPU: sorted-order levels (include/spconv_amd.h "sorted-order levels"; csrc/rulebook.hip conv4_* /
subm_rank_probe_kernel).

The reference's GPU path fixes no order for the outputs of a strided convolution (sort + unique of the coordinate keys,
spconv/csrc/sparse/all.py:1533-1552, or hash-slot order, csrc/sparse/indices.py:1380-1425); the CPU path numbers them
first-seen (indices.py:1742-1771), and that build is pinned bit for bit against the oracle and the reference-executed
vectors in tests/test_gpu_rulebook.py.  The sorted build is pinned HERE as exactly that build with the rows renumbered
by ascending coordinate key: every artefact (coordinates, both pair tables, both masks, Native lists and counts) is
compared under the permutation, bit for bit.  The SubM build over a level's rank map is compared with the hash build of
the same rows, bit for bit."""
import numpy as np
import pytest
import torch

from util import gpu_rulebook, scene, to_np

pytestmark = pytest.mark.gpu

SORTED_CASES = [
    # (shape, n, bs, ksize, stride, padding, dilation)
    ([19, 18, 17], 1500, 2, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]),
    ([19, 18, 17], 1500, 2, [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1]),
    ([19, 18, 17], 1500, 1, [3, 3, 3], [3, 3, 3], [2, 2, 2], [1, 1, 1]),
    ([41, 64, 64], 6000, 2, [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1]),
    ([60, 50], 900, 2, [3, 3], [2, 2], [1, 1], [1, 1]),                         # 2-d
    ([41, 400, 352], 60000, 4, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]),    # several prefix blocks (24 M cells)
    ([19, 18, 17], 1500, 2, [3, 3, 1], [2, 2, 1], [1, 1, 0], [1, 1, 1]),       # a kernel that is flat along x
    ([19, 18, 17], 1500, 2, [3, 3, 3], [2, 2, 2], [0, 0, 0], [1, 1, 1]),       # no padding
    ([7, 6, 9, 8], 700, 2, [2, 2, 2, 2], [2, 2, 2, 2], [0] * 4, [1] * 4),      # 4-d
    ([9000], 1200, 3, [2], [2], [0], [1]),                                     # 1-d (k3 s2 has 2 of 3 offsets: hash builder)
]


def _keys(ind, shape):
    key = ind[:, 0].astype(np.int64)
    for d, s in enumerate(shape):
        key = key * int(s) + ind[:, 1 + d]
    return key


def _check_renumbered(rs, rf, out_shape, n_keep=None):
    """rs (sorted build) == rf (first-seen build) with the output rows renumbered by ascending key."""
    of, os_ = to_np(rf.out_indices), to_np(rs.out_indices)
    kf = _keys(of, out_shape)
    order = np.argsort(kf, kind="stable")               # sorted position -> first-seen row
    if n_keep is None:
        n_keep = of.shape[0]
    assert rs.n_out == n_keep
    np.testing.assert_array_equal(os_[:n_keep], of[order][:n_keep])
    assert np.all(np.diff(_keys(os_[:n_keep], out_shape)) > 0)
    newrow = np.full(of.shape[0], -1, np.int64)          # first-seen row -> sorted row (or dropped)
    newrow[order[:n_keep]] = np.arange(n_keep)
    ren = lambda t: np.where(t >= 0, newrow[np.maximum(t, 0)], -1)
    pf_f, pf_s = to_np(rf.pair_fwd), to_np(rs.pair_fwd)
    np.testing.assert_array_equal(pf_s, pf_f[:, order[:n_keep]])
    pb_f, pb_s = to_np(rf.pair_bwd), to_np(rs.pair_bwd)
    np.testing.assert_array_equal(pb_s, ren(pb_f))
    np.testing.assert_array_equal(to_np(rs.mask_fwd), to_np(rf.mask_fwd)[order[:n_keep]])
    kv = pb_f.shape[0]
    want_mb = np.zeros_like(to_np(rf.mask_bwd)).view(np.uint32)
    for k in range(kv):
        want_mb[:, k // 32] |= ((pb_s[k] >= 0).astype(np.uint32) << np.uint32(k % 32))
    np.testing.assert_array_equal(to_np(rs.mask_bwd).view(np.uint32), want_mb)
    # Native lists: list k = the pairs of offset k in input order (indices.py:1767-1768)
    num_s, nat_s = to_np(rs.num_per_loc), to_np(rs.pair_native)
    for k in range(kv):
        ins = np.nonzero(pb_s[k] >= 0)[0]
        assert num_s[k] == ins.shape[0]
        np.testing.assert_array_equal(nat_s[0, k, :num_s[k]], ins)
        np.testing.assert_array_equal(nat_s[1, k, :num_s[k]], pb_s[k][ins])
    if n_keep == of.shape[0]:
        np.testing.assert_array_equal(num_s, to_np(rf.num_per_loc))


@pytest.mark.parametrize("shape,n,bs,ksize,stride,pad,dil", SORTED_CASES)
def test_sorted_build_is_the_first_seen_build_renumbered(cuda, shape, n, bs, ksize, stride, pad, dil):
    idx = scene(shape, n, bs, seed=11)
    rf, out_shape = gpu_rulebook(idx, bs, shape, ksize, stride, pad, dil, False)
    rs, out_shape_s = gpu_rulebook(idx, bs, shape, ksize, stride, pad, dil, False, out_order="sorted")
    assert list(out_shape) == list(out_shape_s) and rs.rankmap is not None
    _check_renumbered(rs, rf, out_shape)


def test_sorted_build_skips_deleted_rows(cuda):
    """Rows with a batch index outside [0, batch) ("deleted" points, docs/USAGE.md:150) create no output and no pair, as
    in the first-seen build."""
    shape, bs = [19, 18, 17], 2
    idx = scene(shape, 1500, bs, seed=13).copy()
    idx[::7, 0] = -1
    idx[3::11, 0] = bs
    args = ([3] * 3, [2] * 3, [1] * 3, [1] * 3)
    rf, out_shape = gpu_rulebook(idx, bs, shape, *args, False)
    rs, _ = gpu_rulebook(idx, bs, shape, *args, False, out_order="sorted")
    _check_renumbered(rs, rf, out_shape)
    dead = np.nonzero((idx[:, 0] < 0) | (idx[:, 0] >= bs))[0]
    assert np.all(to_np(rs.pair_bwd)[:, dead] == -1)


def test_sorted_build_with_an_output_bound_keeps_the_smallest_keys(cuda):
    shape, n, bs = [19, 18, 17], 1500, 2
    idx = scene(shape, n, bs, seed=3)
    args = ([3] * 3, [2] * 3, [1] * 3, [1] * 3)
    rf, out_shape = gpu_rulebook(idx, bs, shape, *args, False)
    bound = rf.n_out * 2 // 3
    rs, _ = gpu_rulebook(idx, bs, shape, *args, False, out_order="sorted", num_out_act_bound=bound)
    _check_renumbered(rs, rf, out_shape, n_keep=bound)


def test_geometries_without_compact_candidates_keep_the_first_seen_builder(cuda):
    """Stride 1, transposed convolutions and the (3,1,1) / (2,1,1) tail of VoxelBackBone8x (two of three offsets are
    candidates): `sorted` is a request, the hash builder answers (documented)."""
    shape, n = [19, 18, 17], 800
    idx = scene(shape, n, 1, seed=5)
    for ks, st, pd, tr in (([3] * 3, [1] * 3, [1] * 3, False), ([3] * 3, [2] * 3, [1] * 3, True),
                           ([3, 1, 1], [2, 1, 1], [0] * 3, False)):
        rf, _ = gpu_rulebook(idx, 1, shape, ks, st, pd, [1] * 3, False, tr)
        rs, _ = gpu_rulebook(idx, 1, shape, ks, st, pd, [1] * 3, False, tr, out_order="sorted")
        assert rs.rankmap is None
        np.testing.assert_array_equal(to_np(rs.out_indices), to_np(rf.out_indices))
        np.testing.assert_array_equal(to_np(rs.pair_fwd), to_np(rf.pair_fwd))


@pytest.mark.parametrize("n_in,need_bwd", [(60000, True), (300000, False)])       # (the second: the hash build takes its mask-pass form)
def test_subm_over_the_rank_map_equals_the_hash_build(cuda, n_in, need_bwd):
    from spconv_amd.pytorch import ops
    shape, bs = [41, 400, 352], 4
    idx = scene(shape, n_in, bs, seed=21)
    rs, out_shape = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, out_order="sorted")
    ind = rs.out_indices
    assert getattr(ind, "_spx_rankmap", None) is not None
    plain = ind.clone()                                   # the same rows without the map: the hash build
    kw = dict(need_bwd_table=need_bwd)
    b = ops.build_rulebook(plain, bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, **kw)[0]
    from spconv_amd import _lib
    L = _lib.load()
    try:
        for form in (1, 0, -1):       # row-owned kernel, probe kernel (mirror entries scattered), the size rule
            L.spx_set_option(b"SPX_SUBM_RANK_ROWS", form)
            a = ops.build_rulebook(ind, bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True, **kw)[0]
            torch.cuda.synchronize()
            for name in ("pair_fwd", "pair_bwd", "mask_fwd", "num_per_loc", "pair_native"):
                x, y = getattr(a, name), getattr(b, name)
                assert (x is None) == (y is None), name
                if x is not None:
                    assert torch.equal(x, y), (name, form)
    finally:
        L.spx_set_option(b"SPX_SUBM_RANK_ROWS", -1)
    # dilated SubM over the same map
    a = ops.build_rulebook(ind, bs, out_shape, [3] * 3, [1] * 3, [2] * 3, [2] * 3, [0] * 3, True)[0]
    b = ops.build_rulebook(plain, bs, out_shape, [3] * 3, [1] * 3, [2] * 3, [2] * 3, [0] * 3, True)[0]
    assert torch.equal(a.pair_fwd, b.pair_fwd) and torch.equal(a.pair_native, b.pair_native)
    # a map that does not describe the rows (other tensor, other shape) is not used
    assert ops._rankmap_of(plain, bs, out_shape, plain.shape[0], 27) is None
    assert ops._rankmap_of(ind, bs, [s + 1 for s in out_shape], ind.shape[0], 27) is None


def test_static_sorted_build_equals_the_two_call_build(cuda):
    """Static-shape form: padded inputs (dead rows), room for more outputs than exist, a bound below the count."""
    from spconv_amd.pytorch import ops
    shape, bs = [41, 200, 176], 2
    idx = scene(shape, 20000, bs, seed=9)
    n = idx.shape[0]
    pad = np.full((n + 777, 4), -1, np.int32)
    pad[:n] = idx
    args = ([3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3)
    dev = torch.device("cuda:0")
    ref, out_shape = ops.build_rulebook(torch.from_numpy(idx).to(dev), bs, shape, *args, out_order="sorted")
    for cap in (ref.n_out + 1000, ref.n_out - 1234):
        rb, _ = ops.build_rulebook(torch.from_numpy(pad).to(dev), bs, shape, *args, out_order="sorted",
                                   static_num_out=cap)
        torch.cuda.synchronize()
        found, overflow = (int(v) for v in rb.n_out_dev.tolist())
        assert found == ref.n_out and overflow == 0
        live = min(cap, ref.n_out)
        np.testing.assert_array_equal(to_np(rb.out_indices)[:live], to_np(ref.out_indices)[:live])
        assert np.all(to_np(rb.out_indices)[live:] == -1)
        pf, pf_ref = to_np(rb.pair_fwd), to_np(ref.pair_fwd)
        np.testing.assert_array_equal(pf[:, :live], pf_ref[:, :live])
        assert np.all(pf[:, live:] == -1)
        pb, pb_ref = to_np(rb.pair_bwd), to_np(ref.pair_bwd)
        np.testing.assert_array_equal(pb[:, :n], np.where(pb_ref < live, pb_ref, -1))
        assert np.all(pb[:, n:] == -1)
        np.testing.assert_array_equal(to_np(rb.mask_fwd)[:live], to_np(ref.mask_fwd)[:live])
        # the SubM layer behind it, over the padded level: live rows as the unpadded build, dead rows without pairs
        sub, _ = ops.build_rulebook(rb.out_indices, bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)
        want, _ = ops.build_rulebook(ref.out_indices[:live].clone(), bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3,
                                     [0] * 3, True)
        sp, wp = to_np(sub.pair_fwd), to_np(want.pair_fwd)
        np.testing.assert_array_equal(sp[:, :live], wp)
        centre = 13
        assert np.all(np.delete(sp[:, live:], centre, axis=0) == -1)


def test_backbone_in_sorted_order_equals_first_seen_order(cuda, monkeypatch):
    """The config-4 backbone: the two output orders give the same network function -- dense outputs and every weight
    gradient agree (fp32; only the summation order inside the weight gradients differs)."""
    import spconv_amd.pytorch as spconv
    from spconv_amd import constants
    from spconv_amd.utils import nets, synthetic
    shape, bs = [41, 160, 144], 2
    idx = torch.from_numpy(synthetic.lidar_like_scene(shape, 9000, bs, seed=4)).cuda()
    feat = torch.randn(idx.shape[0], 4, device="cuda")
    torch.manual_seed(1)
    net = nets.second_backbone(4, norm=False).cuda()
    res = {}
    for order in ("first_seen", "sorted"):
        monkeypatch.setattr(constants, "CONV_OUTPUT_ORDER", order)
        net.zero_grad(set_to_none=True)
        y = net(spconv.SparseConvTensor(feat, idx, shape, bs))
        d = y.dense()
        (d * torch.linspace(-1, 1, d.numel(), device="cuda").view_as(d)).sum().backward()
        res[order] = (d.detach().clone(), [p.grad.clone() for p in net.parameters()], y.indices.clone())
    a, b = res["first_seen"], res["sorted"]
    assert not torch.equal(a[2], b[2])                      # (the orders do differ)
    assert float((a[0] - b[0]).abs().max()) <= 1e-5 * float(a[0].abs().max())
    for ga, gb in zip(a[1], b[1]):
        assert float((ga - gb).abs().max()) <= 2e-4 * float(ga.abs().max())


def test_rank_map_is_dropped_when_the_index_tensor_was_written(cuda):
    """Round-5 ADVICE: the map is a Python attribute of the index tensor; an in-place edit of `out.indices` between the
    strided layer and the SubM layer behind it must not leave a stale map in use.  The tensor's version counter and
    storage are recorded with the map; after a write the SubM build takes the hash table (and is still right)."""
    from spconv_amd.pytorch import ops
    shape, bs = [41, 200, 176], 2
    idx = scene(shape, 20000, bs, seed=4)
    rs, out_shape = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, out_order="sorted")
    ind = rs.out_indices
    n = ind.shape[0]
    assert ops._rankmap_of(ind, bs, out_shape, n, 27) is not None
    want = ops.build_rulebook(ind.clone(), bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    # a user moves the first rows three cells along z, in place: other neighbourhoods, rows no longer in key order
    ind[:300, 1] = (ind[:300, 1] + 3) % out_shape[0]
    assert ops._rankmap_of(ind, bs, out_shape, n, 27) is None and getattr(ind, "_spx_rankmap", None) is None
    got = ops.build_rulebook(ind, bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    ref = ops.build_rulebook(ind.clone(), bs, out_shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
    assert torch.equal(got.pair_fwd, ref.pair_fwd) and not torch.equal(got.pair_fwd, want.pair_fwd)


def test_sorted_build_is_gated_by_grid_cells_per_input_row(cuda, monkeypatch):
    """Round-5 ADVICE (medium): the rank-map build costs grid cells, not rows.  A large, very sparse grid -- more than
    SPCONV_AMD_SORTED_MAX_CELLS_PER_ROW output cells per input row -- takes the first-seen hash build even when the
    caller asks for `sorted`; no rank map is attached and the rulebook is the first-seen one bit for bit."""
    from spconv_amd.pytorch import ops
    shape, bs = [200, 200, 200], 1                        # 1 M output cells
    idx = scene(shape, 1000, bs, seed=8)                  # ~1000 cells per input row > 512
    assert not ops._sorted_pays(idx.shape[0], bs, [100, 100, 100])
    rs, out_shape = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, out_order="sorted")
    rf, _ = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, out_order="first_seen")
    assert rs.rankmap is None and getattr(rs.out_indices, "_spx_rankmap", None) is None
    assert torch.equal(rs.out_indices, rf.out_indices) and torch.equal(rs.pair_fwd, rf.pair_fwd)
    monkeypatch.setattr(ops, "_SORTED_MAX_CELLS_PER_ROW", 0.0)       # the override: no gate
    rs2, _ = gpu_rulebook(idx, bs, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False, out_order="sorted")
    assert rs2.rankmap is not None
    _check_renumbered(rs2, rf, out_shape)


def test_explicit_mask_sort_also_applies_to_a_sorted_order_build(cuda):
    """SPCONV_DO_SORT=1 (`do_sort=True`): "explicit mask sort of every rulebook" -- also behind the rank-map builder
    (round-5 ADVICE: the sorted path returned before the sort)."""
    from spconv_amd.pytorch import ops
    shape, bs = [41, 200, 176], 2
    idx = torch.from_numpy(scene(shape, 20000, bs, seed=6)).to("cuda:0")
    args = ([3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3)
    rb, _ = ops.build_rulebook(idx, bs, shape, *args, out_order="sorted", do_sort=True)
    assert rb.rankmap is not None and rb.argsort_fwd is not None and rb.argsort_bwd is not None
    m = rb.mask_fwd.view(-1).to(torch.int64) & 0xffffffff
    assert bool((m[rb.argsort_fwd.long()].diff() >= 0).all())
    assert "fwd" in rb.sorted_tables and "bwd" in rb.sorted_tables


def _sorted_scene(shape, n, bs, seed, dev="cuda:0"):
    from spconv_amd.pytorch.utils import sort_voxels_by_coordinate
    idx = torch.from_numpy(scene(shape, n, bs, seed)).to(dev)
    return sort_voxels_by_coordinate(idx, shape, batch_size=bs)[0]


@pytest.mark.parametrize("n_per,form", [(3000, -1), (30000, 0), (30000, 1), (80000, -1)])
def test_level_one_in_key_order_builds_without_a_hash_table(cuda, n_per, form):
    """rows the CALLER hands over in ascending unique key order carry a rank map built from the
    rows themselves (spx_rankmap_from_sorted: row = rank); the SubM rulebook over it (spx_subm_rulebook_ranked) equals
    the hash build of the same rows bit for bit -- tables, masks, Native lists, counts; also with trailing dead rows
    (static shapes) and a dilated kernel."""
    from spconv_amd import _lib
    from spconv_amd.pytorch import ops
    shape, bs = [41, 400, 352], 4
    ind = _sorted_scene(shape, n_per, bs, seed=5)
    n = ind.shape[0]
    assert ops._rankmap_of(ind, bs, shape, n, 27) is not None            # (sort_voxels_by_coordinate attached it)
    plain = ind.clone()
    L = _lib.load()
    L.spx_set_option(b"SPX_SUBM_RANK_ROWS", form)
    try:
        for dil in (1, 2):
            args = ([3] * 3, [1] * 3, [dil] * 3, [dil] * 3, [0] * 3, True)
            a = ops.build_rulebook(ind, bs, shape, *args, need_bwd_table=True)[0]
            b = ops.build_rulebook(plain, bs, shape, *args, need_bwd_table=True)[0]
            torch.cuda.synchronize()
            for name in ("pair_fwd", "pair_bwd", "mask_fwd", "num_per_loc", "pair_native"):
                assert torch.equal(getattr(a, name), getattr(b, name)), (name, dil)
        # static shapes: the same rows padded with dead rows (batch -1) behind them
        pad = torch.full((n + 555, 4), -1, dtype=torch.int32, device=ind.device)
        pad[:n] = ind
        assert ops.attach_rank_map(pad, bs, shape, check=True)
        a = ops.build_rulebook(pad, bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
        b = ops.build_rulebook(pad.clone(), bs, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, [0] * 3, True)[0]
        for name in ("pair_fwd", "mask_fwd", "num_per_loc", "pair_native"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
    finally:
        L.spx_set_option(b"SPX_SUBM_RANK_ROWS", -1)


def test_rank_map_from_rows_refuses_rows_that_break_the_contract(cuda):
    from spconv_amd.pytorch import ops
    shape, bs = [41, 200, 176], 2
    ind = _sorted_scene(shape, 8000, bs, seed=2)
    n = ind.shape[0]
    # unsorted
    shuffled = ind[torch.randperm(n, device=ind.device)].contiguous()
    assert not ops.attach_rank_map(shuffled, bs, shape, check=True)
    assert getattr(shuffled, "_spx_rankmap", None) is None
    # a coordinate twice (still non-decreasing: only uniqueness is broken)
    dup = torch.cat([ind[:100], ind[99:100], ind[100:]]).contiguous()
    assert not ops.attach_rank_map(dup, bs, shape, check=True)
    # a live row behind a dead one
    holes = ind.clone()
    holes[50, 0] = -1
    assert not ops.attach_rank_map(holes, bs, shape, check=True)
    # a grid far too large for the rows: the map would cost more than the hash table it replaces
    assert not ops.attach_rank_map(ind[:10].contiguous(), 1, [2000, 2000, 400], check=True)
    # the untouched sorted rows pass, and the verdict is not cached anywhere: a second call passes too
    assert ops.attach_rank_map(ind.clone(), bs, shape, check=True)


def test_captured_pass_over_key_ordered_input_equals_eager(cuda):
    """StaticInference(key_ordered_input=True): the level-1 rank map is rebuilt from the static index buffer inside the
    graph for every scene; live rows equal the eager, unbounded pass (hash table at level 1) bit for bit."""
    import copy
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticInference, strided_layers
    from spconv_amd.pytorch.utils import sort_voxels_by_coordinate
    from torch import nn
    shape, bs = [32, 40, 40], 2
    torch.manual_seed(3)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="s0"), nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="s0"), nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False, indice_key="d1"), nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1")).to(cuda).half().eval()
    eager = copy.deepcopy(net)
    name = list(strided_layers(net))[0]
    runner = StaticInference(net, max_voxels=12_000, in_channels=8, spatial_shape=shape, batch_size=bs,
                             dtype=torch.float16, bounds={name: 13_000}, key_ordered_input=True)
    for n, seed in ((4500, 1), (2001, 2), (5999, 3)):
        idx = torch.from_numpy(scene(shape, n, bs, seed)).to(cuda)
        f = (torch.rand((idx.shape[0], 8), device=cuda) - 0.5).half()
        idx_s, f_s, _ = sort_voxels_by_coordinate(idx, shape, f, batch_size=bs, rank_map=False)
        with torch.no_grad():
            want = eager(spconv.SparseConvTensor(f_s, idx_s.clone(), shape, bs))     # (untagged: hash table at level 1)
        got = runner(f_s, idx_s)
        assert runner.overflowed() == {}
        k = want.indices.shape[0]
        assert torch.equal(got.indices[:k], want.indices) and torch.equal(got.features[:k], want.features)
    runner.release_bounds()


@pytest.mark.parametrize("shape,n,bs,dead", [([19, 18, 17], 1500, 2, 0), ([41, 400, 352], 60000, 4, 777),
                                             ([60, 50], 900, 2, 5), ([1600, 1280, 40], 20000, 1, 0),
                                             ([2000, 2000, 400], 30000, 1, 100),      # 31 key bits: two radix passes
                                             ([41, 1600, 1408], 110000, 4, 20000),    # BASELINE config 4's grid
                                             ([64, 64, 64], 100000, 1, 0),            # crowded buckets (38 % occupied)
                                             ([7, 6, 9, 8], 700, 2, 3), ([9000], 1200, 3, 7),     # 4-d, 1-d
                                             ([19, 18, 17], 3, 1, 0), ([128, 1600, 1408], 30000, 8, 0)])  # three rows; 2.3 G keys
def test_key_argsort_is_the_stable_sort_by_coordinate_key(cuda, shape, n, bs, dead):
    """spx_key_argsort (the entry sort of the static runners): order = stable argsort of the linear coordinate keys,
    dead rows (batch -1) behind every live row in their own order; the gathered index rows ride along."""
    from spconv_amd.pytorch import ops
    ind = scene(shape, n, bs, seed=4)
    rng = np.random.default_rng(1)
    ind = ind[rng.permutation(ind.shape[0])]
    if dead:                                    # dead rows anywhere (the runners' trail, the sort does not need that)
        pos = rng.choice(ind.shape[0], dead, replace=False)
        ind[pos, 0] = -1
    key = _keys(ind, shape)
    key[ind[:, 0] < 0] = np.iinfo(np.int64).max
    want = np.argsort(key, kind="stable")
    order, rows = ops.key_argsort(torch.from_numpy(ind).to(cuda), bs, shape)
    assert np.array_equal(to_np(order), want.astype(np.int32))
    want_rows = ind[want]
    want_rows[want_rows[:, 0] < 0] = -1         # (dead rows come out as -1 in every column)
    assert np.array_equal(to_np(rows), want_rows)
    assert ops.key_argsort(torch.from_numpy(ind).to(cuda), 1 << 12, [1 << 10, 1 << 10]) is None     # > 32 key bits
    # rows that travel with the sort (a level's features: 8-byte, 16-byte-multiple and odd row sizes)
    for width, dt in ((4, torch.float16), (16, torch.float32), (6, torch.float16)):
        feats = torch.randn(ind.shape[0], width, device=cuda).to(dt)
        o3, r3, f3 = ops.key_argsort(torch.from_numpy(ind).to(cuda), bs, shape, rows=feats)
        assert torch.equal(o3, order) and torch.equal(r3, rows) and torch.equal(f3, feats[order.long()])
    # rank_map=True: the bucket pass leaves the map spx_rankmap_from_sorted would build from the sorted rows, word for word
    flag = torch.ones((1,), dtype=torch.int32, device=cuda)
    order_m, rows_m = ops.key_argsort(torch.from_numpy(ind).to(cuda), bs, shape, rank_map=True, violation=flag)
    assert torch.equal(order_m, order) and torch.equal(rows_m, rows) and int(flag.item()) == 0
    ref = rows.clone()
    if ops.attach_rank_map(ref, bs, shape, check=True):
        got_map, want_map = rows_m._spx_rankmap[0], ref._spx_rankmap[0]
        W = (bs * int(np.prod(shape)) + 31) // 32
        off = ((W * 8 + 255) // 256) * 256 // 4
        assert torch.equal(got_map[:2 * W], want_map[:2 * W])
        assert torch.equal(got_map[off:off + (W + 2047) // 2048], want_map[off:off + (W + 2047) // 2048])
        assert rows_m._spx_rankmap[1:4] == ref._spx_rankmap[1:4]
    else:
        assert getattr(rows_m, "_spx_rankmap", None) is None           # (the same size gates)
    # a coordinate twice breaks the contract (the rank-map pass behind the sort raises its flag): still a permutation,
    # and every row outside the buckets of the doubled keys is where the sort puts it
    live = np.flatnonzero(ind[:, 0] >= 0)
    if live.size < 20:
        return
    ind2 = ind.copy()
    ind2[live[3]] = ind2[live[-7]]
    ind2[live[11]] = ind2[live[-7]]
    t2 = torch.from_numpy(ind2).to(cuda)
    order2, rows2 = ops.key_argsort(t2, bs, shape)
    o2 = to_np(order2)
    assert np.array_equal(np.sort(o2), np.arange(ind2.shape[0]))
    want_rows = ind2[o2]
    want_rows[want_rows[:, 0] < 0] = -1
    assert np.array_equal(to_np(rows2), want_rows)
    assert not ops.attach_rank_map(rows2, bs, shape, check=True)
    ops.key_argsort(t2, bs, shape, rank_map=True, violation=flag)
    assert int(flag.item()) == 1 or getattr(rows_m, "_spx_rankmap", None) is None


def test_entry_sort_of_the_static_runners(cuda):
    """entry_sort (default): a captured pass sorts its scene by coordinate key at the entry and runs level 1 over a rank
    map.  (a) a backbone's output (behind a strided layer: key order either way) equals the pass without the sort bit
    for bit; (b) a SubM-only stack's output comes back in the CALLER's row order; (c) the gradient of the input features
    of a captured training step arrives in the caller's order; (d) a scene with a coordinate twice raises the flag."""
    import copy
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch.static import StaticInference, StaticTrainingStep, strided_layers
    from torch import nn
    shape, bs = [32, 40, 40], 2
    torch.manual_seed(5)
    net = spconv.SparseSequential(
        spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="s0"), nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, bias=False, indice_key="s0"), nn.ReLU(),
        spconv.SparseConv3d(16, 32, 3, 2, 1, bias=False, indice_key="d1"), nn.ReLU(),
        spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="s1")).to(cuda).half().eval()
    name = list(strided_layers(net))[0]
    mk = lambda es: StaticInference(copy.deepcopy(net), max_voxels=12_000, in_channels=8, spatial_shape=shape,
                                    batch_size=bs, dtype=torch.float16, bounds={name: 13_000}, entry_sort=es)
    on, off = mk(None), mk(False)
    assert on.entry_sort and not off.entry_sort
    flat = spconv.SparseSequential(
        spconv.SubMConv3d(8, 16, 3, bias=False, indice_key="s0"), nn.ReLU(),
        spconv.SubMConv3d(16, 16, 3, bias=True, indice_key="s0")).to(cuda).half().eval()
    fon = StaticInference(copy.deepcopy(flat), 12_000, 8, shape, bs, torch.float16, bounds={})
    for n, seed in ((4500, 1), (2001, 2), (5999, 3)):
        idx = torch.from_numpy(scene(shape, n, bs, seed)).to(cuda)
        idx = idx[torch.randperm(idx.shape[0], device=cuda)].contiguous()
        f = (torch.rand((idx.shape[0], 8), device=cuda) - 0.5).half()
        a, b = on(f, idx), off(f, idx)
        assert not on.input_order_violation()
        assert torch.equal(a.indices, b.indices) and torch.equal(a.features, b.features)             # (a)
        k = idx.shape[0]
        key = _keys(to_np(idx), shape)
        assert np.array_equal(to_np(on.order)[:k], np.argsort(key, kind="stable"))
        with torch.no_grad():
            want = flat(spconv.SparseConvTensor(f, idx, shape, bs))
        got = fon(f, idx)                                                                               # (b)
        assert torch.equal(got.indices[:k], idx) and torch.equal(got.features[:k], want.features)
        assert bool((got.indices[k:, 0] < 0).all())
    # (c)
    tnet = copy.deepcopy(flat).train()
    eager = copy.deepcopy(tnet)
    idx = torch.from_numpy(scene(shape, 3000, bs, 7)).to(cuda)
    idx = idx[torch.randperm(idx.shape[0], device=cuda)].contiguous()
    k = idx.shape[0]
    f = (torch.rand((k, 8), device=cuda) - 0.5).half()
    g = ((torch.rand((k + 100, 16), device=cuda) - 0.5) * 0.2).half()
    g[k:] = 0
    step = StaticTrainingStep(tnet, k + 100, 8, shape, bs, torch.float16, bounds={}, out_grad=g, input_grad=True,
                              example=(f, idx))
    assert step.entry_sort
    out = step(f, idx)
    fe = f.clone().requires_grad_(True)
    ye = eager(spconv.SparseConvTensor(fe, idx, shape, bs))
    ye.features.backward(g[:k])
    assert torch.equal(out.indices[:k], idx) and torch.equal(out.features[:k], ye.features)
    assert torch.equal(step.features.grad[:k], fe.grad)                  # (dgrad: a row's sum does not see the row order)
    for pa, pb in zip(tnet.parameters(), eager.parameters()):
        rel = float((pa.grad.float() - pb.grad.float()).norm() / pb.grad.float().norm().clamp_min(1e-12))
        assert rel < 2e-3, rel
    # (d)
    dup = idx.clone()
    dup[5] = dup[900]
    step(f, dup)
    assert step.input_order_violation()
    step(f, idx)
    assert not step.input_order_violation()


def test_sort_voxels_by_coordinate_native_sort_equals_the_torch_path(cuda):
    """utils.sort_voxels_by_coordinate on CUDA int32 rows with a batch size takes spx_key_argsort (rank map written by the
    same pass); without one, or with a coordinate twice, torch.argsort -- same rows, same carried tensors, same order."""
    from spconv_amd.pytorch.utils import sort_voxels_by_coordinate
    shape, bs = [41, 400, 352], 4
    idx = torch.from_numpy(scene(shape, 30000, bs, seed=9)).to(cuda)
    idx = idx[torch.randperm(idx.shape[0], device=cuda)].contiguous()
    f = torch.randn(idx.shape[0], 5, device=cuda)
    a_idx, a_f, a_order = sort_voxels_by_coordinate(idx, shape, f, batch_size=bs)
    b_idx, b_f, b_order = sort_voxels_by_coordinate(idx, shape, f)                 # (no batch size: the general path)
    assert torch.equal(a_idx, b_idx) and torch.equal(a_f, b_f) and torch.equal(a_order, b_order)
    assert getattr(a_idx, "_spx_rankmap", None) is not None and getattr(b_idx, "_spx_rankmap", None) is None
    dup = idx.clone()
    dup[7] = dup[20000]
    c_idx, c_f, c_order = sort_voxels_by_coordinate(dup, shape, f, batch_size=bs)
    d_idx, d_f, d_order = sort_voxels_by_coordinate(dup, shape, f)
    assert torch.equal(c_idx, d_idx) and getattr(c_idx, "_spx_rankmap", None) is None
    assert torch.equal(c_f, f[c_order])
