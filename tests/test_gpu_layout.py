"""GPU: the density-aware rows layout of a SubM rulebook (spx_subm_layout, ops.rows_layout) -- the DEFAULT row
order of the layer modules, the job of the reference's default mask sort (SPCONV_DO_SORT = "1",
spconv/constants.py:121; pytorch/ops.py:346,550,763-785 -> all.py:935-991).

* the blob is a pure function of the mask words: restated in numpy here (class rule; main mask words with the
  appendix' rows zeroed; the appendix = rows with a neighbour in a stable order by lowest neighbour offset, with
  their mask words and pair columns) and compared bit for bit;
* a layout never changes a result: forward, dgrad, the fused backward and the int8 forward are bit-identical
  to the same launches over the row-order tables, for a regrouped (sparse) and an identity (dense) rulebook;
* it is what `net(x)` builds with a default environment, also inside a captured graph (nothing is read back).
"""
import numpy as np
import pytest
import torch

from util import dense_scene, gpu_rulebook, scene

pytestmark = pytest.mark.gpu

K3, ONE = [3] * 3, [1] * 3


def layout_ref(mask: np.ndarray, kv: int):
    """numpy restatement: (class, M, main mask words, appendix row list) -- the appendix holds the rows with a neighbour
    in a stable order by their LOWEST neighbour offset."""
    m = mask.astype(np.uint32).reshape(-1)
    n = m.shape[0]
    centre = np.uint32(1 << (kv // 2))
    rest = m & ~centre
    low = np.zeros(n, dtype=np.int64)
    nz = rest != 0
    low[nz] = np.log2((rest[nz] & (~rest[nz] + np.uint32(1))).astype(np.float64)).astype(np.int64) + 1   # lowest set bit
    heavy = int(nz.sum())
    cls = 1 if (heavy > 0 and 4 * heavy < n) else 0
    rows = np.nonzero(nz)[0]
    order = rows[np.argsort(low[rows], kind="stable")].astype(np.int32) if cls else np.zeros(0, dtype=np.int32)
    main = np.where(nz, np.uint32(0), m) if cls else m
    return cls, heavy, main, order


def check_blob(rb):
    from spconv_amd.pytorch import ops
    head, main, order, mask_a, pair_a = (t.cpu().numpy() for t in ops.layout_views(rb))
    mask = rb.mask_fwd.cpu().numpy().view(np.uint32).reshape(-1)
    pair = rb.pair_fwd.cpu().numpy()
    n, kv = rb.n_out, rb.kv
    cls, heavy, main_ref, order_ref = layout_ref(mask, kv)
    assert head[0] == cls and head[1] == heavy and head[2] == n and head[3] == kv, (head, cls, heavy)
    assert head[4] == order.shape[0] >= n // 4 + 256
    np.testing.assert_array_equal(main.view(np.uint32), main_ref)
    if cls:
        np.testing.assert_array_equal(order[:heavy], order_ref)
        np.testing.assert_array_equal(mask_a[:heavy].view(np.uint32), mask[order_ref])
        np.testing.assert_array_equal(pair_a[:, :heavy], pair[:, order_ref])
    return cls, heavy


@pytest.mark.parametrize("n,shape,expect", [
    (100_000, [40, 1280, 1600], 1),      # BASELINE config 2: ~3 % of the rows have a neighbour
    (40_000, [40, 400, 400], 1),         # 15 %
    (60_000, [24, 160, 160], 0),         # ~90 %: dense
    (33_001, [40, 1280, 1600], 1),       # ragged tail of the last block
])
def test_layout_blob_is_the_stable_partition_of_the_masks(cuda, n, shape, expect):
    idx = scene(shape, n, 1, 7)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    assert rb.layout is not None
    cls, heavy = check_blob(rb)
    assert cls == expect, (cls, heavy, n)


def test_layout_of_the_lidar_fixture_is_the_identity(cuda):
    from golden import lidar_scene
    idx, shape = lidar_scene()
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    cls, heavy = check_blob(rb)
    assert cls == 0 and heavy > 0.9 * idx.shape[0]


def test_layout_with_duplicates_deleted_rows_and_batches(cuda):
    """Rows with a batch index outside [0, batch) and repeated coordinates only keep their centre pair (or their
    k > centre half): they are ordinary rows to the partition."""
    shape = [40, 600, 600]
    idx = scene(shape, 50_000, 3, 11)
    idx[::97, 0] = -1
    idx[5::211] = idx[4::211][: idx[5::211].shape[0]]
    rb, _ = gpu_rulebook(idx, 3, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    check_blob(rb)


def _tensors(rb, C, K, dtype, seed, cuda):
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = (torch.rand((rb.n_in, C), generator=g) * 2 - 1).to(cuda, dtype)
    w = (torch.rand((K, 3, 3, 3, C), generator=g) * 2 - 1).to(cuda, dtype)
    d = ((torch.rand((rb.n_out, K), generator=g) * 2 - 1) * 0.2).to(cuda, dtype)
    return f, w, d


@pytest.mark.parametrize("sparse", [True, False, "small"])
@pytest.mark.parametrize("C,K,dtype", [(64, 64, torch.float16), (32, 64, torch.bfloat16), (16, 16, torch.float16),
                                       (64, 128, torch.float16), (32, 32, torch.float32)])
def test_layout_launches_are_bit_identical_to_row_order(cuda, sparse, C, K, dtype):
    from spconv_amd.pytorch import ops
    if sparse == "small":          # 32 768 rows: the smallest rulebook that gets a layout, 64-row tiles
        shape, idx = [40, 1280, 1600], scene([40, 1280, 1600], 32_768, 1, 3)
    elif sparse:                   # 1172 main tiles + 293 appendix workgroups, ~70 of them with rows
        shape, idx = [40, 1280, 1600], scene([40, 1280, 1600], 150_000, 1, 3)
    else:
        shape, idx = [24, 300, 300], scene([24, 300, 300], 150_000, 1, 3)
    bs = 1
    rb, _ = gpu_rulebook(idx, bs, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    cls, _ = check_blob(rb)
    assert cls == (1 if sparse else 0)
    f, w, d = _tensors(rb, C, K, dtype, 5, cuda)
    pair, mask, blob, to = ops.tables_of(rb, "fwd", K)
    assert to == 2 and blob is rb.layout and pair is rb.pair_fwd
    plan = ops._plan_of(rb)
    out0 = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    out1 = ops.igemm_fwd(f, w, pair, mask, blob, rb.n_out, 13, tile_order=to)
    assert torch.equal(out0, out1)
    din0 = ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True)
    din1 = ops.igemm_dgrad(d, w, pair, mask, blob, rb.n_in, True, tile_order=to)
    assert torch.equal(din0, din1)
    b0 = ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan)
    b1 = ops.igemm_bwd(f, d, w, pair, mask, blob, rb.pair_native, rb.num_per_loc, True, plan, tile_order=to)
    assert torch.equal(b0[0], b1[0]) and torch.equal(b0[1], b1[1])
    assert torch.equal(b1[0], din1)


@pytest.mark.parametrize("n,shape,lo,hi", [
    (60_000, [40, 380, 380], 0.20, 0.25),      # appendix nearly full: whole 128-row tiles, more tiles than the backward's budget
    (60_000, [40, 552, 552], 0.09, 0.15),      # 64- / 80-row appendix tiles
    (40_000, [40, 1280, 1600], 0.0, 0.02),     # a few hundred rows: 16-row tiles, most reserved workgroups leave at once
])
def test_layout_appendix_dealing_is_bit_identical_for_every_fill(cuda, n, shape, lo, hi):
    """The appendix' M rows are dealt to the reserved workgroups in 16-row blocks (csrc/igemm.hip): every fill of the
    appendix -- a handful of rows up to n / 4 -- gives the row-order results, forward, dgrad and fused backward (which
    deals to at most kAppBudget workgroups)."""
    from spconv_amd.pytorch import ops
    idx = scene(shape, n, 1, 21)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    cls, heavy = check_blob(rb)
    assert cls == 1 and lo * rb.n_out <= heavy <= hi * rb.n_out, (heavy, rb.n_out)
    f, w, d = _tensors(rb, 64, 64, torch.float16, 6, cuda)
    pair, mask, blob, to = ops.tables_of(rb, "fwd", 64)
    plan = ops._plan_of(rb)
    assert torch.equal(ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13),
                       ops.igemm_fwd(f, w, pair, mask, blob, rb.n_out, 13, tile_order=to))
    din0 = ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True)
    assert torch.equal(din0, ops.igemm_dgrad(d, w, pair, mask, blob, rb.n_in, True, tile_order=to))
    b0 = ops.igemm_bwd(f, d, w, rb.pair_fwd, rb.mask_fwd, None, rb.pair_native, rb.num_per_loc, True, plan)
    b1 = ops.igemm_bwd(f, d, w, pair, mask, blob, rb.pair_native, rb.num_per_loc, True, plan, tile_order=to)
    assert torch.equal(b0[0], b1[0]) and torch.equal(b0[1], b1[1]) and torch.equal(b1[0], din0)


@pytest.mark.parametrize("sparse", [True, False])
def test_layout_int8_forward_is_bit_identical(cuda, sparse):
    from spconv_amd.pytorch import ops
    shape = [40, 1280, 1600] if sparse else [24, 200, 200]
    idx = scene(shape, 70_000, 1, 9)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort="layout", need_native=False)
    rng = np.random.default_rng(1)
    C = K = 128
    f = torch.from_numpy(rng.integers(-127, 128, (rb.n_in, C), dtype=np.int8)).to(cuda)
    w = torch.from_numpy(rng.integers(-127, 128, (K, 3, 3, 3, C), dtype=np.int8)).to(cuda)
    sc = torch.from_numpy((rng.uniform(0.5, 1.5, K) * 1e-3).astype(np.float32)).to(cuda)
    bias = torch.from_numpy(rng.uniform(-1, 1, K).astype(np.float32)).to(cuda)
    pair, mask, blob, to = ops.tables_of(rb, "fwd", K)
    ref = ops.igemm_fwd_int8(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13, sc, bias, None, 0.0, torch.int8,
                             ops.Activation.ReLU, 0.0)
    for hint in ((False, True) if sparse else (False,)):   # the hint ("the host has SEEN class 1") only changes the launch shape
        got = ops.igemm_fwd_int8(f, w, pair, mask, blob, rb.n_out, 13, sc, bias, None, 0.0, torch.int8,
                                 ops.Activation.ReLU, 0.0, tile_order=to, sparse_hint=hint)
        assert torch.equal(ref, got)
    assert ops.sparse_neighbourhoods(rb) == sparse
    if sparse:                            # ... and with the row count the host read next to the class word
        got = ops.igemm_fwd_int8(f, w, pair, mask, blob, rb.n_out, 13, sc, bias, None, 0.0, torch.int8,
                                 ops.Activation.ReLU, 0.0, tile_order=to, sparse_hint=True, hint_rows=rb.heavy_rows)
        assert torch.equal(ref, got) and rb.heavy_rows > 0
        # a caller's number that is too small (a stale hint: round-4 ADVICE) cannot shrink the appendix grid below what
        # was read from this blob: every row is still computed
        got = ops.igemm_fwd_int8(f, w, pair, mask, blob, rb.n_out, 13, sc, bias, None, 0.0, torch.int8,
                                 ops.Activation.ReLU, 0.0, tile_order=to, sparse_hint=True, hint_rows=64)
        assert torch.equal(ref, got) and rb.heavy_rows > 64
        # ... on the streaming launch (appendix tiles + streaming main tiles, csrc/igemm_v4.h) and on one workgroup per tile
        from spconv_amd import _lib
        try:
            _lib.load().spx_set_option(b"SPX_I8_STREAM", 0)
            got = ops.igemm_fwd_int8(f, w, pair, mask, blob, rb.n_out, 13, sc, bias, None, 0.0, torch.int8,
                                     ops.Activation.ReLU, 0.0, tile_order=to, sparse_hint=True, hint_rows=rb.heavy_rows)
        finally:
            _lib.load().spx_set_option(b"SPX_I8_STREAM", 1)
        assert torch.equal(ref, got)
    else:
        # a WRONG hint (the host claims to have seen class 1 on a dense rulebook): the streaming launch reads the class
        # word on the device and takes the general body tile by tile -- same bits
        blob._spx_heavy = 4096
        got = ops.igemm_fwd_int8(f, w, pair, mask, blob, rb.n_out, 13, sc, bias, None, 0.0, torch.int8,
                                 ops.Activation.ReLU, 0.0, tile_order=to, sparse_hint=True, hint_rows=4096)
        assert torch.equal(ref, got)


def test_module_default_builds_the_layout_and_reuses_it(cuda):
    """net(x) with a default environment: the cached rulebook carries the layout, a second layer with the
    same indice_key reuses it, and the result equals the row-order launch bit for bit."""
    import spconv_amd.pytorch as spconv
    import spconv_amd.pytorch.conv as conv_mod
    from spconv_amd.pytorch import ops
    assert conv_mod.MODULE_DO_SORT == "layout", "tests run with SPCONV_DO_SORT unset"
    shape = [40, 1280, 1600]
    idx = scene(shape, 100_000, 1, 0)
    net = spconv.SparseSequential(spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="s"),
                                  spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="s")).to(cuda).half()
    g = torch.Generator(device="cpu").manual_seed(0)
    f = (torch.rand((idx.shape[0], 64), generator=g) * 2 - 1).to(cuda).half()
    x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), shape, 1)
    y = net(x)
    rb = y.indice_dict["s"].rulebook
    assert rb.layout is not None and int(rb.layout[0].item()) == 1
    assert rb.argsort_fwd is None and not rb.sorted_tables          # no radix sort, no table copies
    mid = ops.igemm_fwd(f, net[0].weight.detach(), rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    ref = ops.igemm_fwd(mid, net[1].weight.detach(), rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    assert torch.equal(y.features, ref)


def test_layout_build_sits_in_a_graph(cuda):
    """Nothing of the build is read back: rulebook + layout + forward captured once, replayed for a second scene
    of the other density class written into the same buffers."""
    import spconv_amd.pytorch as spconv
    shape = [40, 1000, 1000]
    n = 60_000
    sparse_idx = scene(shape, n, 1, 1)
    dense_idx = dense_scene(shape, n, 1, 2)          # the same rows in a 27th of the volume
    assert dense_idx.shape[0] == n
    net = spconv.SubMConv3d(32, 32, 3, bias=False, indice_key="g").to(cuda).half()
    g = torch.Generator(device="cpu").manual_seed(3)
    feats = (torch.rand((n, 32), generator=g) * 2 - 1).to(cuda).half()
    ind = torch.from_numpy(sparse_idx).to(cuda)

    def run():
        with torch.no_grad():
            y = net(spconv.SparseConvTensor(feats, ind, shape, 1))
        return y

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = run()
    classes = []
    for idx in (sparse_idx, dense_idx):
        ind.copy_(torch.from_numpy(idx).to(cuda))
        graph.replay()
        torch.cuda.synchronize()
        rb = y.indice_dict["g"].rulebook
        classes.append(int(rb.layout[0].item()))
        ref = run()
        assert torch.equal(y.features, ref.features)
    assert classes == [1, 0], classes


def test_capture_leaves_out_the_layout_of_class_0_rulebooks(cuda):
    """Inside a capture a SubM module whose rulebooks were of class 0 in the passes before (every level of a LiDAR
    backbone) builds no rows layout -- three launches per rulebook -- and its convolutions walk the row-order tables with the
    density hint the same passes left; a module that saw class 1 (sparse: config 2) keeps its layout.  Values are those
    of the pass with the layout, bit for bit (no normalisation layer in between: the per-row sums do not see the tiling)."""
    import copy
    import spconv_amd.pytorch as spconv
    from golden import lidar_scene
    from spconv_amd.pytorch import ops
    from spconv_amd.pytorch.static import StaticInference
    from util import scene
    from torch import nn
    for kind in ("dense", "sparse"):
        if kind == "dense":
            idx, shape = lidar_scene()
        else:
            shape = [40, 1600, 1280]
            idx = scene(shape, 100_000, 1, 3)
        ind = torch.from_numpy(idx).to(cuda)
        torch.manual_seed(2)
        net = spconv.SparseSequential(spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="a"), nn.ReLU(),
                                      spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="a")).to(cuda).half().eval()
        f = (torch.rand((ind.shape[0], 64), device=cuda) - 0.5).half()
        with torch.no_grad():
            for _ in range(3):                              # eager passes: the class word arrives, the module learns it
                want = net(spconv.SparseConvTensor(f, ind, shape, 1)).features
                torch.cuda.synchronize()
        assert bool(getattr(net[0], ops._CLS0_ATTR, False)) == (kind == "dense")
        built = []
        keep = ops.rows_layout
        ops.rows_layout = lambda rb: (built.append(rb.n_out), keep(rb))[1]
        try:
            runner = StaticInference(copy.deepcopy(net), ind.shape[0], 64, shape, 1, torch.float16, bounds={},
                                     entry_sort=False, warmup=1)
        finally:
            ops.rows_layout = keep
        # warm-up pass (eager) builds one; the captured pass builds one only for the sparse module
        assert len(built) == (1 if kind == "dense" else 2), (kind, built)
        got = runner(f, ind).features
        assert torch.equal(got, want)
