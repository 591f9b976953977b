"""GPU parity: rulebook artefacts must be BIT-EXACT against the CPU oracle
(which restates spconv/csrc/sparse/indices.py:1639-1778).  All calls go through
the C ABI (spconv_amd/_lib.py -> libspconv_amd.so)."""
import numpy as np
import pytest
import torch

from util import (assert_rulebook_equal, dense_scene, gpu_rulebook, oracle_rulebook, scene, to_np)

pytestmark = pytest.mark.gpu

SUBM_CASES = [
    # (shape, n, bs, ksize, dilation)
    ([64, 64, 64], 5000, 1, [3, 3, 3], [1, 1, 1]),          # BASELINE cfg 1
    ([19, 18, 17], 1500, 2, [3, 3, 3], [1, 1, 1]),          # reference test_conv.py:248-274 shape
    ([19, 18, 17], 1500, 2, [3, 3, 3], [2, 2, 2]),
    ([21, 20, 19], 2000, 1, [3, 1, 3], [1, 1, 1]),
    ([21, 20, 19], 2000, 1, [5, 3, 3], [1, 1, 1]),          # kv = 45 > 32: two mask words
    ([40, 50], 700, 3, [3, 3], [1, 1]),                     # 2-d
    ([300], 120, 2, [5], [1]),                              # 1-d
    ([9, 8, 7, 6], 900, 1, [3, 3, 3, 3], [1, 1, 1, 1]),     # 4-d, kv = 81
    ([8, 8, 8], 3, 1, [3, 3, 3], [1, 1, 1]),                # tiny
]


@pytest.mark.parametrize("shape,n,bs,ksize,dil", SUBM_CASES)
def test_subm_rulebook_bit_exact(cuda, shape, n, bs, ksize, dil):
    idx = scene(shape, n, bs, seed=3)
    nd = len(shape)
    pad = [(k // 2) * d for k, d in zip(ksize, dil)]
    ref = oracle_rulebook(idx, bs, shape, ksize, [1] * nd, pad, dil, True)
    rb, _ = gpu_rulebook(idx, bs, shape, ksize, [1] * nd, pad, dil, True, need_bwd_table=True)
    assert_rulebook_equal(rb, ref, True)


def test_subm_dense_neighbourhoods(cuda):
    shape = [24, 24, 24]
    idx = dense_scene(shape, 3000, 2, seed=5)
    ref = oracle_rulebook(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    assert ref["num"].sum() > 4 * idx.shape[0]  # really dense
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                         need_bwd_table=True)
    assert_rulebook_equal(rb, ref, True)


def test_subm_duplicates_and_deleted_rows(cuda):
    """Duplicate coordinates: first index wins (unordered_map::insert, indices.py:1672);
    rows with batch index outside [0, batch) only keep the centre pair (indices.py:1678-1688)."""
    shape = [12, 12, 12]
    idx = dense_scene(shape, 400, 1, seed=9)
    idx = np.concatenate([idx, idx[:50], idx[10:30]], axis=0)         # duplicates
    idx[5, 0] = -1                                                     # "deleted" point
    idx[77, 0] = 3                                                     # batch >= batch_size
    idx = np.ascontiguousarray(idx)
    ref = oracle_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                         need_bwd_table=True)
    assert_rulebook_equal(rb, ref, True)


def _far_corner_scene(shape, n, bs, seed):
    """Dense cluster next to the far corner of a grid whose volume does not fit 32 bits: linear
    keys exceed 2**32, which takes the wide (int64 key) form of the hash table."""
    idx = dense_scene([30, 30, 30], n, bs, seed)
    idx[:, 1:] += np.asarray(shape, dtype=np.int32) - 12
    return np.ascontiguousarray(idx)


def test_subm_key_space_beyond_32_bits(cuda):
    shape = [3000, 2500, 2000]                       # 1.5e10 cells per scene
    idx = _far_corner_scene(shape, 400, 2, seed=2)
    assert int(idx[:, 1].max()) * shape[1] * shape[2] > 2 ** 32
    ref = oracle_rulebook(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    assert ref["num"].sum() > idx.shape[0]
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, need_bwd_table=True)
    assert_rulebook_equal(rb, ref, True)


def test_conv_key_space_beyond_32_bits(cuda):
    shape = [4000, 4000, 4000]                       # stride 2 -> 8e9 output cells
    idx = _far_corner_scene(shape, 400, 1, seed=4)
    ref = oracle_rulebook(idx, 1, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    rb, out_shape = gpu_rulebook(idx, 1, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    assert list(out_shape) == [2000] * 3
    assert_rulebook_equal(rb, ref, False)


def test_subm_empty(cuda):
    idx = np.zeros((0, 4), dtype=np.int32)
    rb, _ = gpu_rulebook(idx, 1, [8, 8, 8], [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    assert rb.pair_fwd.shape == (27, 0) and int(rb.num_per_loc.sum()) == 0


CONV_CASES = [
    # (shape, n, bs, ksize, stride, padding, dilation, transposed)
    ([19, 18, 17], 1500, 2, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False),
    ([19, 18, 17], 1500, 2, [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], False),
    ([19, 18, 17], 1500, 1, [3, 3, 3], [1, 1, 1], [0, 0, 0], [2, 2, 2], False),
    ([19, 18, 17], 1500, 1, [3, 3, 3], [3, 3, 3], [2, 2, 2], [1, 1, 1], False),
    ([19, 18, 17], 1500, 1, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], False),
    ([41, 64, 64], 6000, 2, [3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1], False),  # VoxelBackBone8x tail
    ([41, 64, 64], 6000, 2, [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1], False),
    ([10, 9, 9], 500, 2, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], True),       # transposed
    ([10, 9, 9], 500, 1, [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], True),
    ([60, 50], 900, 2, [3, 3], [2, 2], [1, 1], [1, 1], False),                     # 2-d
]


@pytest.mark.parametrize("shape,n,bs,ksize,stride,pad,dil,transposed", CONV_CASES)
def test_conv_rulebook_bit_exact(cuda, shape, n, bs, ksize, stride, pad, dil, transposed):
    idx = scene(shape, n, bs, seed=11)
    ref = oracle_rulebook(idx, bs, shape, ksize, stride, pad, dil, False, transposed)
    rb, out_shape = gpu_rulebook(idx, bs, shape, ksize, stride, pad, dil, False, transposed)
    assert list(out_shape) == list(ref["out_shape"])
    assert_rulebook_equal(rb, ref, False)


def test_conv_dilation_sharing_a_factor_with_stride(cuda):
    """k = 3, s = 2, d = 2, p = 2: an input whose coordinates are all even reaches all 27 outputs,
    not the prod(ceil(k/s)) = 8 the reference's hand-crafted bound assumes; with stride-aligned,
    well separated inputs the number of distinct outputs is 27 N.  The output hash table must be
    sized for that (no silently dropped pairs)."""
    shape = [64, 64, 64]
    rng = np.random.default_rng(5)
    g = np.stack(np.unravel_index(rng.choice(10 * 10 * 10, 600, replace=False), (10, 10, 10)), -1)
    idx = np.concatenate([np.zeros((600, 1), np.int64), 6 * g + 2], 1).astype(np.int32)   # even, 6 apart
    ks, st, pd, dl = [3] * 3, [2] * 3, [2] * 3, [2] * 3
    ref = oracle_rulebook(idx, 1, shape, ks, st, pd, dl, False)
    assert ref["n_out"] == 27 * 600
    rb, _ = gpu_rulebook(idx, 1, shape, ks, st, pd, dl, False)
    assert_rulebook_equal(rb, ref, False)


def test_conv_points_vanish_raises(cuda):
    """ops.py:260-262: zero active outputs is an error, with the reference's message."""
    idx = np.array([[0, 7, 3, 3]], dtype=np.int32)
    with pytest.raises(ValueError, match="points vanished"):
        gpu_rulebook(idx, 1, [8, 8, 8], [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1], False)


def test_subm_even_kernel_raises(cuda):
    idx = scene([8, 8, 8], 20, 1)
    with pytest.raises(RuntimeError, match="odd ksize"):
        gpu_rulebook(idx, 1, [8, 8, 8], [2, 2, 2], [1] * 3, [0] * 3, [1] * 3, True)


def test_mask_argsort_is_stable_sort(cuda):
    """spx_mask_argsort / spx_mask_argsort_kv (the reference: thrust's stable sort_by_key, all.py:935-991) against
    numpy's stable argsort: sizes around the 2048-key tiles of the one-launch-per-pass sort (single tile, many tiles,
    enough tiles that the look-back walks over several predecessors), full 32-bit keys (four 8-bit passes), 27-bit mask
    words (three 9-bit passes), narrow kernel volumes, heavy duplication."""
    from spconv_amd.pytorch import ops
    rng = np.random.default_rng(0)
    for n in (1, 63, 2048, 2049, 50_000, 1_000_003):
        for bits, kv in ((27, 0), (27, 27), (32, 0), (9, 9), (3, 3), (16, 16), (32, 32)):
            m = rng.integers(0, 1 << bits, size=(n, 1), dtype=np.int64).astype(np.uint32)
            m[rng.random(n) < 0.6] = 1 << min(13, bits - 1)                # many equal keys
            t = torch.from_numpy(m.view(np.int32)).to(cuda)
            perm = to_np(ops.mask_argsort(t, kv)).astype(np.int64)
            expect = np.argsort(m[:, 0], kind="stable")
            np.testing.assert_array_equal(perm, expect, err_msg=f"n={n} bits={bits} kv={kv}")
    # twice on the same stream without a synchronisation in between (the control block is zeroed by the call itself)
    m = rng.integers(0, 1 << 27, size=(300_000, 1), dtype=np.int64).astype(np.uint32)
    t = torch.from_numpy(m.view(np.int32)).to(cuda)
    a, b = ops.mask_argsort(t, 27), ops.mask_argsort(t, 27)
    assert torch.equal(a, b)
    np.testing.assert_array_equal(to_np(a).astype(np.int64), np.argsort(m[:, 0], kind="stable"))


def test_layout_conversions_round_trip(cuda):
    """spx_native_to_table / spx_table_to_native reproduce the builder's own artefacts."""
    from spconv_amd.pytorch import ops
    shape = [16, 16, 16]
    idx = dense_scene(shape, 900, 2, seed=2)
    for subm in (True, False):
        stride = [1] * 3 if subm else [2] * 3
        rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, stride, [1] * 3, [1] * 3, subm,
                             need_bwd_table=True)
        table, mask = ops._table_from_native(rb.pair_native, rb.num_per_loc, rb.n_out, subm, False)
        np.testing.assert_array_equal(to_np(table), to_np(rb.pair_fwd))
        np.testing.assert_array_equal(to_np(mask), to_np(rb.mask_fwd))
        tb, _ = ops._table_from_native(rb.pair_native, rb.num_per_loc, rb.n_in, subm, True)
        np.testing.assert_array_equal(to_np(tb), to_np(rb.pair_bwd))
        native, num = ops._native_from_table(rb.pair_fwd if subm else rb.pair_bwd, subm)
        np.testing.assert_array_equal(to_np(native), to_np(rb.pair_native))
        np.testing.assert_array_equal(to_np(num), to_np(rb.num_per_loc))


def test_full_size_properties_cfg2(cuda):
    """BASELINE cfg 2 (100k voxels, 40x1280x1600): size-independent rulebook properties plus
    a full comparison (the oracle still finishes in well under a second here)."""
    shape = [40, 1280, 1600]
    idx = scene(shape, 100_000, 1, seed=0)
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True,
                         need_bwd_table=True)
    fwd, bwd = to_np(rb.pair_fwd), to_np(rb.pair_bwd)
    kv, n = fwd.shape
    # mirror symmetry, identity centre, involution o -> i -> o
    np.testing.assert_array_equal(bwd, fwd[::-1])
    np.testing.assert_array_equal(fwd[kv // 2], np.arange(n))
    k = 5
    valid = fwd[k] >= 0
    np.testing.assert_array_equal(fwd[kv - 1 - k][fwd[k][valid]], np.nonzero(valid)[0])
    # masks are the column occupancy; lists are sorted by input index
    mask = to_np(rb.mask_fwd).view(np.uint32)[:, 0]
    np.testing.assert_array_equal(mask, ((fwd >= 0) << np.arange(kv)[:, None]).sum(0).astype(np.uint32))
    num, native = to_np(rb.num_per_loc), to_np(rb.pair_native)
    for kk in range(kv // 2):
        assert num[kk] == (fwd[kk] >= 0).sum()
        assert np.all(np.diff(native[0, kk, :num[kk]]) > 0)
    ref = oracle_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    assert_rulebook_equal(rb, ref, True)


@pytest.mark.parametrize("subm", [True, False])
def test_native_lists_built_on_demand_equal_the_builder(cuda, subm):
    """Inference rulebooks skip the ConvAlgo.Native lists (need_native=False); when something asks
    for them later they are derived from the dense tables and must equal what the builder writes."""
    shape, bs = [20, 22, 24], 2
    idx = dense_scene(shape, 2500, bs, seed=13)
    args = ([3] * 3, [1] * 3, [1] * 3, [1] * 3, True) if subm else ([3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    full, _ = gpu_rulebook(idx, bs, shape, *args)
    lazy, _ = gpu_rulebook(idx, bs, shape, *args, need_native=False)
    assert lazy._pair_native is None and lazy._num_per_loc is None
    np.testing.assert_array_equal(to_np(lazy.pair_fwd), to_np(full.pair_fwd))
    np.testing.assert_array_equal(to_np(lazy.mask_fwd), to_np(full.mask_fwd))
    np.testing.assert_array_equal(to_np(lazy.num_per_loc), to_np(full.num_per_loc))       # built here
    np.testing.assert_array_equal(to_np(lazy.pair_native), to_np(full.pair_native))
    ref = oracle_rulebook(idx, bs, shape, *args)
    assert_rulebook_equal(lazy, ref, subm)


# ---- against vectors produced by executing the reference's own CPU code (oracle/_ref) --------
from golden import digest, load_ref_case, ref_big_inputs, ref_case_names, ref_digests  # noqa: E402


@pytest.mark.parametrize("name", ref_case_names())
def test_gpu_rulebook_equals_reference_executed_vectors(cuda, name):
    import oracle
    c = load_ref_case(name)
    rb, out_shape = gpu_rulebook(c["indices"], c["bs"], c["shape"], c["ksize"], c["stride"], c["pad"], c["dil"],
                                 c["subm"], c["transposed"])
    assert list(out_shape) == c["out_shape"]
    n_in, n_out = c["indices"].shape[0], c["out_inds"].shape[0]
    fwd, bwd, mfwd, mbwd = oracle.dense_tables(c["pair"], c["num"], n_in, n_out, c["subm"])
    ref = dict(out_inds=c["out_inds"], pair=c["pair"], num=c["num"], fwd=fwd, bwd=bwd, mfwd=mfwd, mbwd=mbwd,
               n_in=n_in, n_out=n_out)
    assert_rulebook_equal(rb, ref, c["subm"])


def test_gpu_rulebook_equals_reference_digests_at_baseline_sizes(cuda):
    """config 1 / 2 scenes, the real-LiDAR fixture as SubM and as config 3's stride-2 chain: the
    SHA-256 of what the HIP builder returns equals the digest of what spconv's CPU code produced."""
    want = ref_digests()
    for name in ("cfg1_subm", "cfg2_subm", "fixture_subm"):
        idx, shape = ref_big_inputs(name)
        rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
        got = (digest(to_np(rb.out_indices)), digest(to_np(rb.pair_native)), digest(to_np(rb.num_per_loc)))
        assert got == (want[name]["out_inds"], want[name]["pair"], want[name]["num"]), name
    cur, cur_shape = ref_big_inputs("fixture_chain_l0")
    for level in range(3):
        w = want[f"fixture_chain_l{level}"]
        rb, out_shape = gpu_rulebook(cur, 1, cur_shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
        assert rb.n_out == w["n_out"] and list(out_shape) == w["out_shape"]
        got = (digest(to_np(rb.out_indices)), digest(to_np(rb.pair_native)), digest(to_np(rb.num_per_loc)))
        assert got == (w["out_inds"], w["pair"], w["num"]), f"chain level {level}"
        cur, cur_shape = to_np(rb.out_indices), list(out_shape)


def test_output_table_overflow_is_reported(cuda):
    """The regular-conv builder sizes its output hash table for the worst case; an insert that finds
    the table full must surface as an error of the count call, not as a silently truncated rulebook.
    The test shrinks the table through the test-only option SPX_TEST_CONV_TABLE_CAP (512 slots) and
    builds a rulebook with thousands of distinct outputs."""
    from spconv_amd import _lib
    L = _lib.load()
    shape = [30, 30, 30]
    idx = scene(shape, 4000, 1, 21)
    _lib.check(L.spx_set_option(b"SPX_TEST_CONV_TABLE_CAP", 512))
    try:
        with pytest.raises(RuntimeError, match="overflow"):
            gpu_rulebook(idx, 1, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    finally:
        _lib.check(L.spx_set_option(b"SPX_TEST_CONV_TABLE_CAP", 0))
    rb, _ = gpu_rulebook(idx, 1, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)   # and the full-size table works
    assert rb.n_out > 512


def test_num_out_act_bound_caps_the_outputs(cuda):
    """ops.get_indice_pairs(..., num_out_act_bound=B) (reference ops.py:263-266): at most B outputs; the
    first B of the unbounded rulebook (canonical first-seen order) survive with exactly their pairs."""
    from spconv_amd.pytorch import ops
    shape = [24, 24, 24]
    idx = scene(shape, 3000, 2, 5)
    t = torch.from_numpy(idx).to(cuda)
    args = (t, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, [0] * 3)
    full, _ = ops.build_rulebook(*args, False)
    B = full.n_out // 2
    rb, _ = ops.build_rulebook(*args, False, num_out_act_bound=B)
    assert rb.n_out == B and rb.out_indices.shape[0] == B
    np.testing.assert_array_equal(to_np(rb.out_indices), to_np(full.out_indices)[:B])
    np.testing.assert_array_equal(to_np(rb.pair_fwd), to_np(full.pair_fwd)[:, :B])
    fb = to_np(full.pair_bwd)
    np.testing.assert_array_equal(to_np(rb.pair_bwd), np.where(fb < B, fb, -1))
    # the Native lists hold exactly the surviving pairs, in the unbounded order
    nat, num = to_np(rb.pair_native), to_np(rb.num_per_loc)
    fnat, fnum = to_np(full.pair_native), to_np(full.num_per_loc)
    for k in range(27):
        keep = fnat[1, k, :fnum[k]] < B
        np.testing.assert_array_equal(nat[:, k, :num[k]], fnat[:, k, :fnum[k]][:, keep])
    # a bound above the real count changes nothing
    same, _ = ops.build_rulebook(*args, False, num_out_act_bound=full.n_out + 10)
    assert same.n_out == full.n_out


def test_subm_masks_from_a_table_pass_equal_the_atomic_form(cuda):
    """From 250 k voxels the SubM builder derives the mask words from the finished table instead of one atomicOr
    per entry (SPX_SUBM_MASK_PASS: -1 = by size): both forms, same tables and masks, at a size where the default
    switches."""
    from spconv_amd import _lib
    from spconv_amd.utils import synthetic
    shape = [41, 1600, 1408]
    idx = synthetic.lidar_like_scene(shape, 150_000, 2, seed=3)
    assert idx.shape[0] >= 250_000
    L = _lib.load()
    got = {}
    try:
        for v in (0, 1, -1):
            _lib.check(L.spx_set_option(b"SPX_SUBM_MASK_PASS", v))
            rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True, need_bwd_table=True)
            got[v] = [t.clone() for t in (rb.pair_fwd, rb.pair_bwd, rb.mask_fwd, rb.pair_native, rb.num_per_loc)]
    finally:
        _lib.check(L.spx_set_option(b"SPX_SUBM_MASK_PASS", -1))
    for v in (1, -1):
        assert all(torch.equal(a, b) for a, b in zip(got[0], got[v])), v
    centre = 1 << 13
    assert bool(((got[0][2].view(-1) & centre) != 0).all())       # every row has its centre bit
