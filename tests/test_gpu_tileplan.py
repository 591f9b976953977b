"""Tile plans (csrc/tileplan.hip) and the dense-neighbourhood gather-GEMM over them
(csrc/igemm.hip: igemm_halo_kernel): structural invariants of a plan, bit-identity of the halo kernel
with the plain kernel (same per-row arithmetic order), parity with the oracle on the reference's
LiDAR fixture at full size, and the spill path (tiles with more unique source rows than the halo holds)."""
import numpy as np
import pytest
import torch

import oracle
from util import dense_scene, gpu_rulebook, oracle_rulebook, rel_err, scene

pytestmark = pytest.mark.gpu

TILE, HALO, NOPAIR, SPILLED = 128, 384, 0xFFFF, 0xFFFE


def _force_plan(rb, direction="fwd"):
    from spconv_amd.pytorch import ops
    old = ops._TILE_MODE
    ops._TILE_MODE = "1"
    try:
        return ops.tile_plan(rb, direction)
    finally:
        ops._TILE_MODE = old


def _unpack(plan, n_dst, kv):
    p = plan.cpu().numpy()
    nt = (n_dst + TILE - 1) // TILE
    assert p[0] == 0x54504c31 and p[1] == n_dst and p[2] == nt and p[3] == kv and p[4] == HALO
    o = 16
    order = p[o:o + nt * TILE]; o += nt * TILE
    info = p[o:o + nt * 4].reshape(nt, 4); o += nt * 4
    halo = p[o:o + nt * HALO].reshape(nt, HALO); o += nt * HALO
    plocal = p[o:o + (nt * kv * TILE + 1) // 2].view(np.uint16)[:nt * kv * TILE].reshape(nt, kv, TILE)
    return order, info, halo, plocal


def _check_plan(plan, table, n_dst, kv):
    order, info, halo, plocal = _unpack(plan, n_dst, kv)
    nt = info.shape[0]
    assert sorted(order[:n_dst].tolist()) == list(range(n_dst)), "order is not a permutation"
    assert (order[n_dst:] == -1).all()
    table = table.cpu().numpy()
    spilled_tiles = 0
    for t in range(nt):
        rows = order[t * TILE:(t + 1) * TILE]
        H = info[t, 1]
        hl = halo[t, :H]
        assert len(np.unique(hl)) == H, "halo rows repeat"
        kmask = 0
        for k in range(kv):
            src = np.where(rows >= 0, table[k][np.maximum(rows, 0)], -1)
            sl = plocal[t, k]
            assert ((src < 0) == (sl == NOPAIR)).all()
            ok = (src >= 0) & (sl != SPILLED)
            assert (sl[ok] < H).all() and (hl[sl[ok]] == src[ok]).all(), "slot table points at the wrong row"
            if (src >= 0).any():
                kmask |= 1 << k
            sp = (src >= 0) & (sl == SPILLED)
            assert not sp.any() or info[t, 2] == 1
        assert int(np.uint32(info[t, 0])) == kmask
        assert info[t, 3] >= H and (info[t, 2] == 1) == (info[t, 3] > HALO)
        spilled_tiles += int(info[t, 2])
    return spilled_tiles


def test_plan_invariants_subm_and_conv(cuda):
    shape = [24, 40, 40]
    idx = dense_scene([30, 90, 90], 9000, 2, seed=5)
    rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    assert _check_plan(_force_plan(rb), rb.pair_fwd, rb.n_out, 27) == 0
    rb2, _ = gpu_rulebook(idx, 2, shape, [3] * 3, [2] * 3, [1] * 3, [1] * 3, False)
    _check_plan(_force_plan(rb2, "fwd"), rb2.pair_fwd, rb2.n_out, 27)
    _check_plan(_force_plan(rb2, "bwd"), rb2.pair_bwd, rb2.n_in, 27)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C,K", [(64, 64), (32, 64), (16, 32), (64, 16), (48, 32)])
def test_halo_kernel_is_bit_identical_to_plain_kernel(cuda, dtype, C, K):
    from spconv_amd.pytorch import ops
    shape = [20, 60, 60]
    idx = dense_scene([30, 120, 120], 20000, 2, seed=C + K)
    torch.manual_seed(C * 100 + K)
    for subm, stride in ((True, [1] * 3), (False, [2] * 3)):
        rb, _ = gpu_rulebook(idx, 2, shape, [3] * 3, stride, [1] * 3, [1] * 3, subm)
        f = torch.randn(rb.n_in, C, device=cuda).to(dtype)
        w = (torch.randn(K, 3, 3, 3, C, device=cuda) * 0.2).to(dtype)
        d = (torch.randn(rb.n_out, K, device=cuda) * 0.2).to(dtype)
        bias = torch.randn(K, device=cuda).to(dtype)
        ident = 13 if subm else -1
        pf = _force_plan(rb, "fwd")
        a = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, ident)
        b = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, ident, plan=pf)
        assert torch.equal(a, b), f"forward differs (subm={subm})"
        a = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, ident, bias, ops.Activation.ReLU)
        b = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, ident, bias, ops.Activation.ReLU, plan=pf)
        assert torch.equal(a, b), "fused bias + ReLU differs"
        if subm:
            a = ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True)
            b = ops.igemm_dgrad(d, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_in, True, plan=pf)
        else:
            pb = _force_plan(rb, "bwd")
            a = ops.igemm_dgrad(d, w, rb.pair_bwd, rb.mask_bwd, None, rb.n_in, False)
            b = ops.igemm_dgrad(d, w, rb.pair_bwd, rb.mask_bwd, None, rb.n_in, False, plan=pb)
        assert torch.equal(a, b), f"dgrad differs (subm={subm})"


def test_spilled_tiles_read_through_the_pair_table(cuda):
    """A fully occupied 32^3 block: a tile is a 2 x 2 column bundle (128 voxels) whose 3x3x3
    neighbourhood spans 4 x 4 columns = 512 unique rows > 384 halo slots."""
    from spconv_amd.pytorch import ops
    n = 32
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    idx = np.stack([np.zeros(n ** 3), z.ravel(), y.ravel(), x.ravel()], 1).astype(np.int32)
    idx = idx[np.random.default_rng(0).permutation(idx.shape[0])]
    rb, _ = gpu_rulebook(idx, 1, [n] * 3, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    plan = _force_plan(rb)
    assert _check_plan(plan, rb.pair_fwd, rb.n_out, 27) > 0
    f = torch.randn(rb.n_in, 32, device=cuda).half()
    w = (torch.randn(64, 3, 3, 3, 32, device=cuda) * 0.2).half()
    a = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13)
    b = ops.igemm_fwd(f, w, rb.pair_fwd, rb.mask_fwd, None, rb.n_out, 13, plan=plan)
    assert torch.equal(a, b)
    ref = oracle_rulebook(idx, 1, [n] * 3, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    want = oracle.indice_conv(f.float().cpu(), w.float().cpu(), ref["pair"], ref["num"], ref["n_out"], subm=True)
    assert rel_err(b.float().cpu().numpy(), want.numpy()) < 2e-3


def test_lidar_fixture_module_path_uses_the_plan_and_matches_the_oracle(cuda, monkeypatch):
    """Automatic mode on the reference's real-LiDAR fixture: the SubM rulebook is judged dense, the
    module's forward and the backward's dgrad run over the plan, results match the oracle."""
    import spconv_amd.pytorch as spconv
    from golden import lidar_scene
    from spconv_amd.pytorch import ops
    monkeypatch.setattr(ops, "_TILE_MODE", "auto")       # (off by default: the plain kernel is faster)
    idx, shape = lidar_scene()
    rng = np.random.default_rng(2)
    C = K = 64
    f = torch.from_numpy(rng.uniform(-1, 1, (idx.shape[0], C)).astype(np.float32)).half()
    dout = torch.from_numpy(rng.uniform(-0.2, 0.2, (idx.shape[0], K)).astype(np.float32)).half()
    net = spconv.SubMConv3d(C, K, 3, bias=False, indice_key="t").to(cuda).half().train()
    feats = f.to(cuda).requires_grad_(True)
    y = net(spconv.SparseConvTensor(feats, torch.from_numpy(idx).to(cuda), shape, 1))
    y.features.backward(dout.to(cuda))
    rb = y.indice_dict["t"].rulebook
    assert rb.tile_plans.get("fwd") is not None, "automatic mode did not plan a LiDAR-density rulebook"
    ref = oracle_rulebook(idx, 1, shape, [3] * 3, [1] * 3, [1] * 3, [1] * 3, True)
    w = net.weight.detach().float().cpu()
    out_ref = oracle.indice_conv(f.float(), w, ref["pair"], ref["num"], ref["n_out"], subm=True)
    din_ref, dw_ref = oracle.indice_conv_backward(f.float(), w, dout.float(), ref["pair"], ref["num"], subm=True)
    assert rel_err(y.features.detach().float().cpu().numpy(), out_ref.numpy()) < 2e-3
    assert rel_err(feats.grad.float().cpu().numpy(), din_ref.numpy()) < 2e-3
    assert rel_err(net.weight.grad.float().cpu().numpy(), dw_ref.numpy()) < 2e-3
    # a uniform-random scene of the same size is judged sparse: no plan, the fused backward stays
    idx2 = scene([40, 1280, 1600], 100_000, 1, 1)
    y2 = net(spconv.SparseConvTensor(torch.randn(idx2.shape[0], C, device=cuda).half(),
                                     torch.from_numpy(idx2).to(cuda), [40, 1280, 1600], 1))
    assert y2.indice_dict["t"].rulebook.tile_plans.get("fwd", "unset") is None
