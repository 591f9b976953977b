"""The weight-stationary gather-GEMM of dense C = K = 64 layers (csrc/igemm_ws.hip: 512-row workgroups, weights nine
offsets at a time through LDS-DMA, free-running waves) against igemm_v4_kernel and the oracle.

Its contract is "the same arithmetic in the same order as v4": forward and dgrad must be BIT-IDENTICAL to the 128-row
kernel for every row order, kernel volume and epilogue -- which is what lets the host pick it from an asynchronous,
possibly late, possibly predicted density class (ops.poll_class) without the choice ever showing in a result.  The
reference has the same freedom (its tuner picks a tile shape per problem, convops.py:1150-1297) without the guarantee."""
import numpy as np
import pytest
import torch

import oracle
from util import gpu_rulebook, oracle_rulebook, rel_err, scene

pytestmark = pytest.mark.gpu

K3, ONE = [3] * 3, [1] * 3


def _force(v):
    from spconv_amd import _lib
    _lib.check(_lib.load().spx_set_option(b"SPX_WS", int(v)))


@pytest.fixture(autouse=True)
def _auto_mode_afterwards():
    yield
    _force(-1)


def _both(fn):
    """fn() under the 128-row kernel and under the weight-stationary one."""
    _force(0)
    a = fn()
    _force(1)
    b = fn()
    torch.cuda.synchronize()
    return a, b


def _operands(cuda, n_src, kshape, dtype, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = (torch.rand((n_src, 64), generator=g) * 2 - 1).to(dtype).to(cuda)
    w = ((torch.rand((64, *kshape, 64), generator=g) * 2 - 1) * 0.2).to(dtype).to(cuda)
    return f, w


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n", [700, 5000, 40_001])
def test_subm_forward_and_dgrad_are_bit_identical_to_v4(cuda, dtype, n):
    """Dense-ish SubM scenes of sizes that are no multiple of the 512-row tile (one partial workgroup, many, ...)."""
    from spconv_amd.pytorch import ops
    shape = [12, 40, 40] if n < 10_000 else [24, 64, 64]
    idx = scene(shape, n, 1, 3)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    f, w = _operands(cuda, n, (3, 3, 3), dtype, 1)
    bias = (torch.arange(64, device=cuda) * 0.01 - 0.3).to(dtype)
    a, b = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13))
    assert torch.equal(a, b)
    a, b = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13, bias, ops.Activation.ReLU, 0.0))
    assert torch.equal(a, b)
    a, b = _both(lambda: ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, n, True))
    assert torch.equal(a, b)
    # and against the oracle (the CPU restatement of the reference's Native loops)
    ref = oracle_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    out_ref = oracle.indice_conv(f.float().cpu(), w.float().cpu(), ref["pair"], ref["num"], n, subm=True)
    _force(1)
    out = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13)
    assert rel_err(out.float().cpu().numpy(), out_ref.numpy()) <= (2e-3 if dtype == torch.float16 else 1.2e-2)


@pytest.mark.parametrize("ksize,stride", [((3, 3, 3), 2), ((2, 2, 2), 2), ((3, 1, 1), 1), ((3, 3, 1), 1)])
def test_other_kernel_volumes_and_strided_tables(cuda, ksize, stride):
    """kv = 27 over a strided rulebook (no identity offset: plain ascending order), kv = 8 / 3 / 9 (a single weight
    phase), forward over pair_fwd and dgrad over pair_bwd."""
    from spconv_amd.pytorch import ops
    shape, n = [16, 48, 48], 9000
    idx = scene(shape, n, 2, 5)
    subm = stride == 1
    pad = [k // 2 for k in ksize] if subm else [1 if k == 3 else 0 for k in ksize]
    rb, _ = gpu_rulebook(idx, 2, shape, list(ksize), [stride] * 3, pad, ONE, subm)
    n_in, n_out = idx.shape[0], rb.n_out
    f, w = _operands(cuda, n_in, ksize, torch.float16, 2)
    kv = int(np.prod(ksize))
    ident = kv // 2 if subm else -1
    a, b = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n_out, ident))
    assert torch.equal(a, b)
    dout, _ = _operands(cuda, n_out, ksize, torch.float16, 3)
    table, mask = (rb.pair_fwd, rb.mask_fwd) if subm else (rb.pair_bwd, rb.mask_bwd)
    a, b = _both(lambda: ops.igemm_dgrad(dout, w, table, mask, None, n_in, subm))
    assert torch.equal(a, b)


def test_tables_in_tile_order(cuda):
    """The explicit mask sort (SPCONV_DO_SORT=1): tables read by tile position, rows named by the argsort."""
    from spconv_amd.pytorch import ops
    shape, n = [12, 40, 40], 6000
    idx = scene(shape, n, 1, 8)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True)
    ops.sort_rulebook(rb)
    pair_t, mask_t = rb.sorted_tables["fwd"]
    f, w = _operands(cuda, n, (3, 3, 3), torch.float16, 4)
    _force(0)
    base = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13)
    _force(1)
    got = ops.igemm_fwd(f, w, pair_t, mask_t, rb.argsort_fwd, n, 13, tile_order=1)
    assert torch.equal(base, got)
    got = ops.igemm_dgrad(f, w, pair_t, mask_t, rb.argsort_fwd, n, True, tile_order=1)
    _force(0)
    assert torch.equal(got, ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, n, True))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_lidar_fixture_full_size(cuda, dtype):
    """BASELINE config 2b (the reference's real-LiDAR fixture, 125 562 voxels, 788 888 pairs): both kernels, forward and
    dgrad, bit for bit; the layer module reaches the weight-stationary kernel by itself once the class word has arrived."""
    import spconv_amd.pytorch as spconv
    from golden import lidar_scene
    from spconv_amd.pytorch import ops
    idx, shape = lidar_scene()
    n = idx.shape[0]
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    f, w = _operands(cuda, n, (3, 3, 3), dtype, 6)
    fwd_ref, b = _both(lambda: ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, n, 13))
    assert torch.equal(fwd_ref, b)
    a, b = _both(lambda: ops.igemm_dgrad(f, w, rb.pair_fwd, rb.mask_fwd, None, n, True))
    assert torch.equal(a, b)
    # the class word: requested when the layout was built, taken without a wait once it is there
    torch.cuda.synchronize()
    ops.poll_class(rb)
    assert rb.sparse_class is False and rb.layout._spx_dense is True and 4 * rb.heavy_rows >= 3 * n
    _force(-1)
    net = spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="k").to(cuda, dtype)
    with torch.no_grad():
        net.weight.copy_(w)
    outs = []
    for _ in range(3):        # (first pass: class unknown -> 128-row tiles; later passes go by the prediction)
        x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(cuda), shape, 1)
        with torch.no_grad():
            outs.append(net(x).features)
        torch.cuda.synchronize()
    assert ops._pred_get(net) is True
    assert torch.equal(outs[0], fwd_ref) and torch.equal(outs[2], fwd_ref)


def test_sparse_scene_keeps_the_row_tiles(cuda):
    """BASELINE config 2 (uniform voxels, 1.03 pairs per voxel): class word 1 -- no dense hint, whatever the size."""
    from spconv_amd.pytorch import ops
    shape = [40, 1280, 1600]
    idx = scene(shape, 120_000, 1, 0)
    rb, _ = gpu_rulebook(idx, 1, shape, K3, ONE, ONE, ONE, True, do_sort="layout")
    torch.cuda.synchronize()
    ops.poll_class(rb)
    assert rb.sparse_class is True and rb.layout._spx_dense is False
    assert ops._with_dense_hint(2, rb.layout) == 2


def test_captured_fixture_step_runs_the_weight_stationary_kernel(cuda):
    """the CAPTURED training step of the reference's LiDAR fixture layer (what bench.py's
    `also.2b` times) must take igemm_ws_kernel for its forward -- inside a capture nothing can be polled, the launch goes
    by the prediction the eager warm-up passes left on the module.  `spx_launch_count` (the library's record of which
    kernel family a call dispatched) is read around the capture; gradients of the replay equal the eager step's."""
    import spconv_amd.pytorch as spconv
    from golden import lidar_scene
    from spconv_amd import _lib
    from spconv_amd.pytorch.static import StaticTrainingStep
    L = _lib.load()
    idx, shape = lidar_scene()
    n = idx.shape[0]
    torch.manual_seed(1)
    net = spconv.SparseSequential(spconv.SubMConv3d(64, 64, 3, bias=False, indice_key="k")).to(cuda).half()
    f = (torch.rand((n, 64), device=cuda) - 0.5).half()
    ind = torch.from_numpy(idx).to(cuda)
    g = ((torch.rand((n, 64), device=cuda) - 0.5) * 0.2).half()
    # eager passes first: the class word of the rows layout arrives, the module learns "dense"
    for _ in range(3):
        with torch.no_grad():
            net(spconv.SparseConvTensor(f, ind, shape, 1))
        torch.cuda.synchronize()
    before = {k: L.spx_launch_count(k) for k in (b"igemm_ws", b"igemm_v4", b"igemm_bwd")}
    # (entry_sort=False: the eager step below runs the caller's row order, and dW is compared bit for bit)
    step = StaticTrainingStep(net, n, 64, shape, 1, torch.float16, bounds={}, out_grad=g, input_grad=True,
                              example=(f, ind), warmup=1, entry_sort=False)
    after = {k: L.spx_launch_count(k) for k in before}
    # one warm-up pass + the captured pass: both forwards on the weight-stationary kernel, none on the 128-row tiles;
    # the backward is the fused launch (dgrad tiles + wgrad ranges: DESIGN.md 3.14 "the backward stays the fused launch")
    assert after[b"igemm_ws"] - before[b"igemm_ws"] == 2, (before, after)
    assert after[b"igemm_v4"] == before[b"igemm_v4"], (before, after)
    assert after[b"igemm_bwd"] - before[b"igemm_bwd"] == 2, (before, after)
    assert L.spx_launch_count(b"no_such_family") == -1
    step(f, ind)
    torch.cuda.synchronize()
    dw_graph, din_graph = net[0].weight.grad.clone(), step.features.grad.clone()
    net.zero_grad(set_to_none=True)
    fe = f.clone().requires_grad_(True)
    net(spconv.SparseConvTensor(fe, ind, shape, 1)).features.backward(g)
    torch.cuda.synchronize()
    assert torch.equal(net[0].weight.grad, dw_graph) and torch.equal(fe.grad, din_graph)
