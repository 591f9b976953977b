"""CPU: host-side logic of the drop-in API (no kernels run)."""
import os
import sys

import numpy as np
import pytest
import torch
from torch import nn

import spconv_amd.pytorch as spconv
from spconv_amd.pytorch import ops


def _x(n=6, C=4, shape=(4, 5, 6), bs=2):
    idx = torch.tensor([[i % bs, i % shape[0], (2 * i) % shape[1], (3 * i) % shape[2]] for i in range(n)],
                       dtype=torch.int32)
    return spconv.SparseConvTensor(torch.arange(n * C, dtype=torch.float32).view(n, C), idx, shape, bs)


def test_sparse_conv_tensor_container():
    x = _x()
    assert x.spatial_shape == [4, 5, 6] and x.batch_size == 2 and x.indice_dict == {}
    with pytest.raises(ValueError, match="replace_feature"):
        x.features = x.features * 2
    y = x.replace_feature(x.features + 1)
    assert y.indices is x.indices and y.indice_dict is x.indice_dict
    d = x.dense()
    assert d.shape == (2, 4, 4, 5, 6)
    i = x.indices.long()
    assert torch.equal(d[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]], x.features)
    assert torch.equal(x.dense(channels_first=False).permute(0, 4, 1, 2, 3), d)
    z = x + y
    assert torch.equal(z.features, 2 * x.features + 1)
    s = x.select_by_index(torch.tensor([0, 2]))
    assert s.features.shape[0] == 2 and s.indices.shape[0] == 2
    with pytest.raises(AssertionError, match="int32"):
        spconv.SparseConvTensor(torch.zeros(2, 3), torch.zeros(2, 4, dtype=torch.int64), [4, 4, 4], 1)
    r = spconv.SparseConvTensor.from_dense(d.permute(0, 2, 3, 4, 1).contiguous())
    assert r.batch_size == 2 and list(r.spatial_shape) == [4, 5, 6]
    assert r.find_indice_pair(None) is None and r.find_indice_pair("nope") is None


def test_module_signatures_and_weight_layout():
    m = spconv.SubMConv3d(16, 32, 3, indice_key="a")
    assert list(m.weight.shape) == [32, 3, 3, 3, 16] and list(m.bias.shape) == [32]   # KRSC
    assert m.subm and m.algo == spconv.ConvAlgo.MaskImplicitGemm and m.indice_key == "a"
    c = spconv.SparseConv3d(16, 32, (3, 1, 1), (2, 1, 1), padding=(0, 1, 1), bias=False)
    assert c.kernel_size == [3, 1, 1] and c.stride == [2, 1, 1] and c.padding == [0, 1, 1] and c.bias is None
    big = spconv.SubMConv3d(4, 4, 5)
    assert big.algo == spconv.ConvAlgo.Native                   # kv = 125 > 32 (conv.py:110-120)
    inv = spconv.SparseInverseConv3d(32, 16, 3, "a")
    assert inv.inverse and inv.indice_key == "a"
    t = spconv.SparseConvTranspose3d(8, 8, 3, 2)
    assert t.transposed
    with pytest.raises(AssertionError, match="groups"):
        spconv.SubMConv3d(4, 4, 3, groups=2)
    bound = 1 / np.sqrt(16 * 27)
    assert m.bias.abs().max() <= bound and m.weight.abs().max() <= np.sqrt(6 / ((1 + 5) * 16 * 27)) + 1e-6
    for cls, nd in ((spconv.SubMConv1d, 1), (spconv.SparseConv2d, 2), (spconv.SparseConv4d, 4)):
        assert cls(4, 4, 3).ndim == nd and cls.__name__.endswith(f"{nd}d")


def test_state_dict_round_trip_and_legacy_layout(monkeypatch):
    a, b = spconv.SparseConv3d(4, 8, 3), spconv.SparseConv3d(4, 8, 3)
    b.load_state_dict(a.state_dict())
    assert torch.equal(a.weight, b.weight) and torch.equal(a.bias, b.bias)
    import spconv_amd.pytorch.conv as conv_mod
    monkeypatch.setattr(conv_mod, "SAVED_WEIGHT_LAYOUT", "RSCK")       # spconv 1.x checkpoints
    sd = {"weight": a.weight.detach().permute(1, 2, 3, 4, 0).contiguous(), "bias": a.bias.detach()}
    c = spconv.SparseConv3d(4, 8, 3)
    c.load_state_dict(sd)
    assert torch.equal(c.weight, a.weight)


def test_sparse_sequential_routes_dense_layers_and_conv1x1():
    x = _x(n=10, C=4)
    net = spconv.SparseSequential(nn.Linear(4, 8), nn.ReLU(), spconv.SubMConv3d(8, 6, 1, bias=True),
                                  spconv.SparseBatchNorm(6), spconv.SparseReLU())
    assert len(net) == 5 and isinstance(net[2], spconv.SubMConv3d)
    y = net(x)                                  # kv == 1 -> torch.mm shortcut, works on CPU
    assert isinstance(y, spconv.SparseConvTensor) and y.features.shape == (10, 6)
    w = net[2].weight
    mid = torch.relu(net[0](x.features))
    expect = torch.mm(mid, w.view(8, 6)) + net[2].bias           # reference quirk conv.py:232-234
    bn = nn.BatchNorm1d(6)
    bn.load_state_dict(net[3].state_dict())
    assert torch.allclose(y.features, torch.relu(bn(expect)), atol=1e-5)
    dense = spconv.SparseSequential(spconv.SubMConv3d(4, 4, 1), spconv.ToDense())(x)
    assert dense.shape == (2, 4, 4, 5, 6)
    spconv.assign_name_for_sparse_modules(net)
    assert net[2]._sparse_unique_name == "2"


def test_cpu_tensors_are_rejected_loudly():
    x = _x()
    with pytest.raises(NotImplementedError, match="no CPU path"):
        spconv.SubMConv3d(4, 4, 3)(x)
    with pytest.raises(NotImplementedError, match="no CPU path"):
        ops.get_indice_pairs(x.indices, 2, x.spatial_shape, spconv.ConvAlgo.Native, [3] * 3, [1] * 3,
                             [1] * 3, [1] * 3, [0] * 3, True)


def test_output_size_helpers_match_reference_formula():
    assert ops.get_conv_output_size([41, 1600, 1408], [3] * 3, [2] * 3, [1] * 3, [1] * 3) == [21, 800, 704]
    assert ops.get_conv_output_size([8, 8], [-1, 3], [1, 1], [0, 0], [1, 1]) == [1, 6]
    assert ops.get_deconv_output_size([5, 5], [3, 3], [2, 2], [1, 1], [1, 1], [1, 1]) == [10, 10]
    with pytest.raises(ValueError):
        ops.get_deconv_output_size([5], [-1], [1], [0], [1], [0])


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under spconv_amd/ may reference it."""
    import os
    root = os.path.dirname(os.path.abspath(spconv.__file__))
    root = os.path.dirname(root)
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".sh")):
                text = open(os.path.join(d, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(d, f)
                assert "liboracle" not in text, os.path.join(d, f)


def test_table_modules_combine_aligned_tensors():
    """tables.py of the reference: JoinTable / AddTable / ConcatTable, Identity."""
    x = _x()
    y = x.replace_feature(x.features * 2)
    j = spconv.JoinTable()([x, y])
    assert j.features.shape == (6, 8) and torch.equal(j.features[:, 4:], y.features)
    assert j.indices is x.indices and j.indice_dict is x.indice_dict
    a = spconv.AddTable()([x, y, y])
    assert torch.equal(a.features, 5 * x.features)
    other = _x(n=5)
    with pytest.raises(AssertionError, match="AddTableMisaligned"):
        spconv.AddTable()([x, other])
    with pytest.raises(AssertionError, match="JoinTable"):
        spconv.JoinTable()([x, other])
    ct = spconv.ConcatTable().add(spconv.Identity()).add(spconv.Identity())
    outs = ct(x)
    assert len(outs) == 2 and outs[0] is x and outs[1] is x
    assert ct.input_spatial_size([3, 3, 3]) == [3, 3, 3]
    assert spconv.JoinTable().input_spatial_size(7) == 7


def test_install_as_spconv_aliases_every_submodule():
    import importlib
    import sys
    import spconv_amd
    saved = {k: v for k, v in sys.modules.items() if k == "spconv" or k.startswith("spconv.")}
    try:
        spconv_amd.install_as_spconv()
        for name in ("core", "conv", "functional", "ops", "modules", "pool", "hash", "utils", "tables",
                     "identity"):
            m = importlib.import_module(f"spconv.pytorch.{name}")
            assert m.__name__ == f"spconv_amd.pytorch.{name}"
        import spconv.pytorch as sp
        assert sp.SubMConv3d is spconv.SubMConv3d and sp.AddTable is spconv.AddTable
        from spconv.pytorch.quantization.intrinsic.qat import SparseConvBnReLU   # noqa: F401
        from spconv.pytorch.quantization.intrinsic.modules import SpconvBnAddReLUNd   # noqa: F401
        import spconv.pytorch.quantization as spq
        assert callable(spq.get_spconv_backend_config)
    finally:
        for k in [k for k in sys.modules if k == "spconv" or k.startswith("spconv.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_bench_gpus_flag_becomes_a_launcher(tmp_path):
    """`bench.py --gpus N` without a launcher's environment re-executes itself under
    torch.distributed.run with N ranks on 127.0.0.1 (BENCH_DRY_LAUNCH prints the command)."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_DRY_LAUNCH"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "7",
                          "--warmup", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(root, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    # under a launcher (WORLD_SIZE set) it must NOT spawn again
    import bench
    os.environ["WORLD_SIZE"] = "4"
    try:
        bench.maybe_spawn(bench.parse(["--gpus", "4"]), ["--gpus", "4"])
    finally:
        del os.environ["WORLD_SIZE"]


def test_kernel_timer_api_and_debug_dump(tmp_path, monkeypatch):
    """spconv/tools.py:23-78 surface; on a box without a GPU the timer is inert like the reference's
    CPU-only build.  SPCONV_DEBUG_SAVE_PATH pickles what a failing call was given."""
    import pickle
    from spconv_amd import tools
    t = tools.CUDAKernelTimer(True)
    with t.namespace("layer"), t.record("forward"):
        pass
    res = t.get_all_pair_time()
    assert isinstance(res, dict) and (not t.enable) == (res == {})
    assert tools.CUDAKernelTimer.collect_by_name("b", {"a.b.c": 1.0, "a.c": 2.0}) == {"a.b.c": 1.0}
    x = spconv.SparseConvTensor(torch.zeros(2, 4), torch.zeros(2, 4, dtype=torch.int32), [4, 4, 4], 1,
                                enable_timer=True)
    assert x._timer is not None and x.replace_feature(x.features)._timer is x._timer
    path = tmp_path / "dump.pkl"
    monkeypatch.setattr(tools, "SPCONV_DEBUG_SAVE_PATH", str(path))
    tools.save_debug_data(("indices", 3))
    assert pickle.load(open(path, "rb")) == ("indices", 3)


def test_peripheral_modules_resolve_under_the_spconv_name():
    """User code imports these by their reference names (spconv/core.py:22-36, debug_utils.py:21,
    pytorch/constants.py:37, pytorch/spatial.py:28)."""
    import spconv_amd
    spconv_amd.install_as_spconv()
    from spconv.core import AlgoHint, ConvAlgo
    from spconv.debug_utils import spconv_save_debug_data
    from spconv.pytorch.constants import PYTORCH_VERSION, is_amp_enabled
    from spconv.pytorch.spatial import RemoveDuplicate
    import spconv.pytorch as sp
    assert ConvAlgo is sp.ConvAlgo and AlgoHint.BackwardWeight.value == 4
    assert callable(spconv_save_debug_data) and is_amp_enabled() is False and len(PYTORCH_VERSION) == 3
    assert issubclass(RemoveDuplicate, sp.SparseModule)


def test_remove_duplicate_keeps_first_row_of_a_coordinate():
    import torch
    import spconv_amd.pytorch as sp
    from spconv_amd.pytorch.spatial import RemoveDuplicate
    idx = torch.tensor([[0, 1, 2, 3], [1, 1, 2, 3], [0, 1, 2, 3], [0, 0, 0, 0], [1, 1, 2, 3]], dtype=torch.int32)
    feat = torch.arange(5, dtype=torch.float32).view(5, 1)
    y = RemoveDuplicate()(sp.SparseConvTensor(feat, idx, [4, 4, 4], 2))
    assert y.indices.tolist() == [[0, 1, 2, 3], [1, 1, 2, 3], [0, 0, 0, 0]]
    assert y.features.view(-1).tolist() == [0.0, 1.0, 3.0]


def test_xcd_aware_mappings_are_bijective():
    """The block -> tile map of the fused backward (csrc/igemm.hip: xcd_tile_rot) and the block -> range
    hand-out of the wgrad plan (wgrad_plan2_kernel, rec[5]) restated in Python: every tile / range is
    taken exactly once, and a workgroup that runs on XCD x gets work of row eighth x."""
    def xcd_tile_rot(bid, ntiles, rot):
        q, r, cls, j = ntiles >> 3, ntiles & 7, bid & 7, bid >> 3
        phys = (cls + rot) & 7
        base = sum(q + (1 if ((y - rot) & 7) < r else 0) for y in range(phys))
        return base + j

    for ntiles in (1, 7, 8, 9, 63, 782, 981, 1000):
        for rot in range(8):
            tiles = [xcd_tile_rot(b, ntiles, rot) for b in range(ntiles)]
            assert sorted(tiles) == list(range(ntiles)), (ntiles, rot)
            if ntiles >= 64:          # physical XCD of dgrad block b is (b + rot) % 8: its tiles lie in eighth x
                for b in range(ntiles):
                    x = (b + rot) & 7
                    assert abs(tiles[b] / ntiles - (x + 0.5) / 8) <= 0.5 / 8 + 2 / ntiles

    def hand_out(xcd_of_range):
        """plan: ranges sorted by (xcd, rank) meet workgroups sorted by (b % 8, b // 8)."""
        G = len(xcd_of_range)
        order = sorted(range(G), key=lambda t: (xcd_of_range[t], t))
        blocks = sorted(range(G), key=lambda b: (b % 8, b // 8))
        take = [None] * G
        for rng, b in zip(order, blocks):
            take[b] = rng
        return take

    import random
    rnd = random.Random(0)
    for G in (1, 8, 178, 242, 384, 1023):
        xs = [min(7, (t * 8) // G) if rnd.random() < 0.9 else rnd.randrange(8) for t in range(G)]
        take = hand_out(xs)
        assert sorted(take) == list(range(G))
        if G >= 64:
            hits = sum(1 for b in range(G) if xs[take[b]] == b % 8)
            assert hits >= 0.8 * G


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_global_avg_pool_accumulates_in_fp32(dtype):
    """6 000 rows per scene: an fp16 row count stops at 2048, a bf16 one at 256 (round-2 ADVICE);
    the mean must match torch.mean over the scene's rows (which accumulates in fp32)."""
    import spconv_amd.pytorch as spconv
    g = torch.Generator().manual_seed(3)
    bs, n, C = 3, 6000, 8
    b = torch.arange(bs).repeat_interleave(n)
    idx = torch.stack([b, torch.arange(bs * n) % 50, torch.arange(bs * n) % 37, torch.arange(bs * n) % 41], 1).int()
    f = (torch.rand((bs * n, C), generator=g) + 0.5).to(dtype)          # all positive: the sum really grows
    x = spconv.SparseConvTensor(f, idx, [50, 37, 41], bs)
    got = spconv.SparseGlobalAvgPool()(x)
    assert got.dtype == dtype and got.shape == (bs, C)
    for i in range(bs):
        want = f[b == i].float().mean(dim=0)
        assert torch.allclose(got[i].float(), want, rtol=1e-2 if dtype == torch.bfloat16 else 2e-3, atol=0)
    # a scene without rows: NaN for the mean (torch.mean over no rows), lowest value for the max
    x2 = spconv.SparseConvTensor(f, idx, [50, 37, 41], bs + 1)
    assert torch.isnan(spconv.SparseGlobalAvgPool()(x2)[bs]).all()


def test_wgrad_plan_cost_space_cut_covers_every_pair_once():
    """wgrad_plan2_kernel (csrc/igemm.hip) restated: the pair lists are laid end to end in COST units
    (pairs + a fixed pad per non-empty list) and cut into G equal ranges; a segment is the part of a
    list's PAIRS inside a range.  Every pair belongs to exactly one segment, the closed form the second
    stage uses for the segments of a list equals their enumerated count, and the short lists of a sparse
    SubM rulebook no longer pile up in a few ranges."""
    def cut(lens, G, ov):
        pad = [ov if c > 0 else 0 for c in lens]
        start = [0]
        for c, p in zip(lens, pad):
            start.append(start[-1] + c + p)
        total = start[-1]
        per = max(1, -(-total // G))
        segs = []
        for t in range(G):
            lo = min(total, t * per)
            hi = min(total, lo + per)
            for k, c in enumerate(lens):
                p0 = start[k] + pad[k]
                pa = min(max(lo - p0, 0), c)
                pb = min(max(hi - p0, 0), c)
                if pb > pa:
                    segs.append((t, k, pa, pb))
        kc = [((start[k + 1] - 1) // per - (start[k] + pad[k]) // per + 1) if c > 0 else 0 for k, c in enumerate(lens)]
        return segs, kc

    cases = [([121] * 13 + [100000] + [121] * 13, 242),            # BASELINE config 2 (SubM, uniform scene)
             ([121] * 13 + [100000] + [121] * 13, 178),            # ... with room for the rows layout's appendix
             ([30000 + 997 * k for k in range(27)], 384),          # dense scene
             ([0, 5, 0, 1, 0, 0, 700, 0, 3], 7), ([1], 1), ([0, 0, 4096], 64)]
    for lens, G in cases:
        for ov in (0, 192):
            segs, kc = cut(lens, G, ov)
            for k, c in enumerate(lens):
                mine = sorted((a, b) for _, kk, a, b in segs if kk == k)
                assert len(mine) == kc[k], (lens, G, ov, k)
                pos = 0
                for a, b in mine:
                    assert a == pos and b > a
                    pos = b
                assert pos == c
            assert [s[:2] for s in segs] == sorted(s[:2] for s in segs)       # (range, list) order = (list, range) order
    # the point of the pad: segments per range at config 2
    worst = lambda ov: max(sum(1 for s in cut(cases[0][0], 242, ov)[0] if s[0] == t) for t in range(242))
    assert worst(0) >= 4 and worst(192) <= 2


def test_device_guard_switches_to_the_tensor_device(monkeypatch):
    """ops._on_device (torch's DeviceGuard convention) with a faked second device: a driver called with a
    tensor on cuda:1 while cuda:0 is current must run inside torch.cuda.device(cuda:1); with the tensor's
    device already current, or a CPU tensor, it must not touch the guard.  (A 1-GPU box cannot exercise
    this for real: round-2 verdict.)"""
    from spconv_amd.pytorch import ops

    class FakeDev:
        def __init__(self, index):
            self.index, self.type = index, "cuda"

    class FakeTensor(torch.Tensor):
        pass

    entered = []

    class Guard:
        def __init__(self, dev):
            self.dev = dev

        def __enter__(self):
            entered.append(self.dev.index)

        def __exit__(self, *a):
            entered.append(-self.dev.index)
            return False

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", Guard)
    seen = []

    @ops._on_device
    def driver(t, extra=None):
        seen.append(list(entered))
        return "ok"

    def fake(index):
        t = FakeTensor(torch.zeros(1))
        t.__class__ = type("T", (FakeTensor,), {"is_cuda": True, "device": FakeDev(index)})
        return t

    assert driver(fake(1)) == "ok" and seen[-1] == [1] and entered == [1, -1]     # guarded: entered, then left
    entered.clear()
    assert driver(fake(0)) == "ok" and entered == []                               # already current: no guard
    assert driver(torch.zeros(2)) == "ok" and entered == []                        # CPU tensor: no guard
    assert driver(None, extra=fake(1)) == "ok" and entered == []                   # only positional tensors decide


def test_compact_candidates_of_a_strided_convolution():
    """Python restatement of csrc/rulebook.hip CandIter (third-generation strided-conv rulebook): the valid
    offsets of an axis as a bit set computed WITHOUT integer division (float32 multiply by 1/s + round, exact
    below 2^21), their product walked in ascending offset index.  Against the brute force over all offsets with
    the reference's integer formula (query_npq, indices.py:174-203): same candidates, same order, and never more
    than prod ceil(k gcd(d, s) / s) of them (conv_max_out, the bound the [MJ, N] arrays are sized by)."""
    import math
    rng = np.random.default_rng(0)

    def bits(c, k, s, p, d, out):
        inv = np.float32(1.0) / np.float32(s)
        m = 0
        for r in range(k):
            h = c + p - r * d
            q = int(np.rint(np.float32(h) * inv))
            if q * s == h and 0 <= q < out:
                m |= 1 << r
        return m

    for trial in range(400):
        nd = int(rng.integers(1, 4))
        k = [int(rng.integers(1, 6)) for _ in range(nd)]
        s = [int(rng.integers(1, 5)) for _ in range(nd)]
        d = [int(rng.integers(1, 4)) for _ in range(nd)]
        p = [int(rng.integers(0, 3)) for _ in range(nd)]
        dims = [int(rng.integers(1, 2_000_000)) if trial % 7 == 0 else int(rng.integers(1, 40)) for _ in range(nd)]
        out = [(dims[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(nd)]
        if min(out) <= 0:
            continue
        bound = math.prod(min(k[i], -(-k[i] * math.gcd(d[i], s[i]) // s[i])) for i in range(nd))
        for _ in range(20):
            c = [int(rng.integers(0, dims[i])) for i in range(nd)]
            # brute force, ascending offset index (last axis fastest)
            want = []
            for kk in range(math.prod(k)):
                r, rest = [], kk
                for i in reversed(range(nd)):
                    r.append(rest % k[i])
                    rest //= k[i]
                r = r[::-1]
                h = [c[i] + p[i] - r[i] * d[i] for i in range(nd)]
                if all(h[i] % s[i] == 0 and 0 <= h[i] // s[i] < out[i] for i in range(nd)):
                    want.append((kk, tuple(h[i] // s[i] for i in range(nd))))
            # the bit-set odometer
            vm = [bits(c[i], k[i], s[i], p[i], d[i], out[i]) for i in range(nd)]
            got = []
            if all(vm):
                v = list(vm)
                while True:
                    r = [(x & -x).bit_length() - 1 for x in v]
                    kk = 0
                    for i in range(nd):
                        kk = kk * k[i] + r[i]
                    got.append((kk, tuple((c[i] + p[i] - r[i] * d[i]) // s[i] for i in range(nd))))
                    i = nd - 1
                    while i >= 0:
                        v[i] &= v[i] - 1
                        if v[i]:
                            break
                        v[i] = vm[i]
                        i -= 1
                    if i < 0:
                        break
            assert got == want, (k, s, p, d, dims, c)
            assert len(got) <= bound, (len(got), bound, k, s, d)


def test_first_seen_ranks_from_a_bit_map():
    """The numbering scheme of the same builder: first-seen flags of one table row as a bit map; the output number
    of an entry = entries of earlier blocks (scan of the block popcounts) + set bits of earlier words in its
    2048-bit block + set bits below it in its word.  Must equal the running count in input order."""
    rng = np.random.default_rng(1)
    n = 10_000
    flags = rng.random(n) < 0.3
    words = np.zeros((n + 31) // 32, np.uint32)
    for i in np.nonzero(flags)[0]:
        words[i >> 5] |= np.uint32(1 << (i & 31))
    pop = np.array([bin(int(w)).count("1") for w in words])
    nblk = (n + 2047) // 2048
    blockcount = np.array([pop[b * 64:(b + 1) * 64].sum() for b in range(nblk)])
    blockoff = np.concatenate([[0], np.cumsum(blockcount)[:-1]])
    wordpre = np.concatenate([np.concatenate([[0], np.cumsum(pop[b * 64:(b + 1) * 64])[:-1]]) for b in range(nblk)])
    want = np.cumsum(flags) - 1
    for i in np.nonzero(flags)[0]:
        below = bin(int(words[i >> 5]) & ((1 << (i & 31)) - 1)).count("1")
        assert blockoff[i // 2048] + wordpre[i >> 5] + below == want[i]


def test_static_shape_bookkeeping_on_the_host():
    """strided_layers / freeze_bounds (spconv_amd/pytorch/static.py) and the occupancy rule that picks the
    one-gather backward (ops._dense_rows): pure host logic."""
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import ops
    from spconv_amd.pytorch.core import Rulebook
    from spconv_amd.pytorch.static import freeze_bounds, strided_layers
    from spconv_amd.utils import nets
    net = nets.second_backbone(4)
    layers = strided_layers(net)
    assert len(layers) == 4 and all(not m.subm for m in layers.values())          # the four down-sampling convs
    names = list(layers)
    with pytest.raises(ValueError, match="no recorded voxel count"):
        freeze_bounds(net)
    used = freeze_bounds(net, {n: 1000 * (i + 1) for i, n in enumerate(names)})
    assert [layers[n].static_num_out for n in names] == [1000, 2000, 3000, 4000] == list(used.values())
    assert freeze_bounds(net, {}, margin=0) == {n: 0 for n in names}
    pool = spconv.SparseSequential(spconv.SparseMaxPool3d(2, 2), spconv.SubMConv3d(4, 4, 3), spconv.SparseConv3d(4, 4, 1))
    assert list(strided_layers(pool)) == ["0"]                                      # a 1x1x1 conv makes no new voxels

    rb = Rulebook(None, None, None, None, None, None, None, 400_000, 400_000, 27, True)
    rb.in_shape, rb.out_shape, rb.batch_size = [41, 1600, 1408], [41, 1600, 1408], 4
    assert not ops._dense_rows(rb, 400_000, "fwd")                                  # 0.1 % of the cells: pair lists
    rb.in_shape = rb.out_shape = [21, 800, 704]
    assert ops._dense_rows(rb, 313_127, "fwd") and ops._dense_rows(rb, 313_127, "bwd")   # 0.66 %: the tile walk
    assert not ops._dense_rows(None, 10, "fwd")


def test_committed_pmc_traffic_answers_every_key_bench_asks_for():
    """profiles/traffic.json is KEYED by workload (a per-key fragment copied over it made every
    lookup return None and the driver's line carry `traffic: null`)."""
    import bench
    for key, group in bench.TRAFFIC_KEYS:
        v = bench.pmc_traffic(key, group)
        assert isinstance(v, int) and v > 1_000_000, (key, group, v)
    r = bench.roofline_obj("fwd", 36.6e6, 0.0095, "k", bench.pmc_traffic("uniform-f16-c64-n100000", "fwd"))
    assert r["traffic"] and 0.3 < r["traffic_over_algorithmic"] < 3.0
    assert bench.roofline_obj("fwd", 1e6, 1.0, "k")["traffic_over_algorithmic"] is None


def test_sort_voxels_by_coordinate():
    """Helper for data loaders (not in the reference): rows in ascending (batch, z, y, x) key order."""
    import torch
    from spconv_amd.pytorch.utils import sort_voxels_by_coordinate
    ind = torch.tensor([[1, 0, 2, 3], [0, 5, 1, 1], [0, 0, 0, 7], [1, 0, 2, 2]], dtype=torch.int32)
    f = torch.arange(4).float().view(4, 1)
    i2, f2, order = sort_voxels_by_coordinate(ind, [8, 8, 8], f)
    assert i2.tolist() == [[0, 0, 0, 7], [0, 5, 1, 1], [1, 0, 2, 2], [1, 0, 2, 3]]
    assert f2.view(-1).tolist() == [2.0, 1.0, 3.0, 0.0] and order.tolist() == [2, 1, 3, 0]
    assert torch.equal(ind[order], i2)


def test_autograd_engine_rule_of_the_package_import(monkeypatch):
    """after `import spconv_amd.pytorch` a process that drives one GPU (or runs under a launcher) has
    its autograd backward on the calling thread -- the fast eager state is the default, not a recipe --, a process that
    sees several GPUs keeps torch's engine and is told once, and SPCONV_AMD_AUTOGRAD_THREADS=keep never touches it."""
    import warnings
    import torch
    import spconv_amd.pytorch as spconv
    calls = []
    monkeypatch.setattr(torch.autograd, "set_multithreading_enabled", lambda v: calls.append(v))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.delenv("SPCONV_AMD_AUTOGRAD_THREADS", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert spconv._autograd_on_the_calling_thread() == "calling thread" and calls == [False]
    calls.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert spconv._autograd_on_the_calling_thread() == "kept" and calls == []
    assert len(w) == 1 and "set_multithreading_enabled" in str(w[0].message)
    monkeypatch.setenv("LOCAL_RANK", "3")                     # torchrun: one process per GPU
    assert spconv._autograd_on_the_calling_thread() == "calling thread" and calls == [False]
    calls.clear()
    monkeypatch.setenv("SPCONV_AMD_AUTOGRAD_THREADS", "keep")
    assert spconv._autograd_on_the_calling_thread() == "kept" and calls == []
