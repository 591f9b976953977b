"""User hash table and misaligned sparse add.  The reference fixes no
entry order for the table (its GPU numbering follows atomic arrival order), so the tests check
the contract: exact membership, value round trips, arange values forming a permutation, and the
sparse sums against a dense accumulation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kdt,vdt", [(torch.int32, torch.int32), (torch.int64, torch.float32),
                                     (torch.int32, torch.int64), (torch.int64, torch.float64)])
def test_hash_table_contract(cuda, kdt, vdt):
    from spconv_amd.pytorch.hash import HashTable
    rng = np.random.default_rng(0)
    n = 5000
    keys_np = rng.choice(10_000_000, n, replace=False).astype(np.int64)
    vals_np = rng.integers(-1000, 1000, n)
    keys = torch.from_numpy(keys_np).to(cuda, kdt)
    vals = torch.from_numpy(vals_np).to(cuda).to(vdt)
    table = HashTable(cuda, kdt, vdt, max_size=2 * n)
    table.insert(keys, vals)
    table.insert(keys[:100], vals[:100])                      # re-inserting is idempotent
    got, missing = table.query(keys)
    assert not bool(missing.any()) and torch.equal(got, vals)
    other = torch.from_numpy(keys_np + 10_000_000).to(cuda, kdt)
    _, missing = table.query(other)
    assert bool(missing.all())
    # insert_exist_keys only touches existing keys
    mixed = torch.cat([keys[:10], other[:10]])
    flags = table.insert_exist_keys(mixed, torch.full((20,), 7, device=cuda).to(vdt))
    assert flags.cpu().tolist() == [0] * 10 + [1] * 10
    got, _ = table.query(keys[:10])
    assert bool((got == 7).all())
    k, v, count = table.items()
    c = int(count.item())
    assert c == n and set(k[:c].cpu().tolist()) == set(keys_np.tolist())
    if vdt in (torch.int32, torch.int64):
        count = table.assign_arange_()
        assert int(count.item()) == n
        ids, missing = table.query(keys)
        assert not bool(missing.any())
        assert sorted(ids.cpu().tolist()) == list(range(n))     # a permutation of arange(n)
        ids2, _ = table.query(keys)
        assert torch.equal(ids, ids2)


@pytest.mark.parametrize("fn_name", ["sparse_add_hash_based", "sparse_add"])
def test_sparse_add_misaligned(cuda, fn_name):
    import spconv_amd.pytorch as spconv
    from spconv_amd.pytorch import functional as Fsp
    from util import scene
    shape, bs, C = [12, 14, 16], 2, 8
    tens, dense = [], torch.zeros((bs, *shape, C))
    for seed in range(3):
        idx = scene(shape, 300 + 50 * seed, bs, seed)
        f = torch.randn(idx.shape[0], C)
        tens.append(spconv.SparseConvTensor(f.to(cuda), torch.from_numpy(idx).to(cuda), shape, bs))
        dense[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] += f
    out = getattr(Fsp, fn_name)(*tens)
    oi = out.indices.cpu().long()
    assert len({tuple(r) for r in oi.tolist()}) == oi.shape[0]                 # distinct coordinates
    assert oi.shape[0] == int((dense.abs().sum(-1) > 0).sum())                  # exactly the union
    got = out.features.cpu()
    want = dense[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3]]
    assert torch.allclose(got, want, atol=1e-5)
    # an operand that already covers the union keeps its rulebooks
    a = tens[0]
    a.indice_dict["k"] = object()
    same = getattr(Fsp, fn_name)(a, a.replace_feature(a.features * 2))
    assert same.features.shape[0] == a.features.shape[0] and "k" in same.indice_dict
    # the module form (tables.py:69-80 of the reference) is the same operation
    via_module = spconv.AddTableMisaligned()(tens)
    mi = via_module.indices.cpu().long()                   # row order of the union is unspecified
    assert {tuple(r) for r in mi.tolist()} == {tuple(r) for r in oi.tolist()}
    assert torch.allclose(via_module.features.cpu(), dense[mi[:, 0], mi[:, 1], mi[:, 2], mi[:, 3]], atol=1e-5)
