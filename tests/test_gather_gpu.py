"""K1 parity: HIP gather+concat vs the numpy oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref

pytestmark = pytest.mark.gpu


def _mk(dev, vocabs, dims, B, seed, ids_float=False, oob_frac=0.0):
    rng = np.random.default_rng(seed)
    tables = [rng.uniform(-0.05, 0.05, size=(v, d)).astype(np.float32) for v, d in zip(vocabs, dims)]
    ids = np.stack([rng.integers(0, v, size=B) for v in vocabs], axis=1).astype(np.int32)
    if oob_frac:
        m = rng.random(ids.shape) < oob_frac
        ids = np.where(m, np.where(rng.random(ids.shape) < 0.5, -1 - ids, ids + np.array(vocabs)[None, :]), ids).astype(np.int32)
    if ids_float:
        idsf = (ids + rng.uniform(0.0, 0.9, size=ids.shape)).astype(np.float32)
        # float32 rounding may bump to the next integer for large ids: keep ids small in such tests
        return tables, idsf
    return tables, ids


@pytest.mark.parametrize("D", [4, 8, 16, 32, 64, 128, 256])
@pytest.mark.parametrize("B", [1, 37, 1000])
def test_gather_uniform_bit_exact(dev, D, B):
    from recamd import ops
    F = 5
    vocabs = [11, 1000, 7, 333, 50]
    tables, ids = _mk(dev, vocabs, [D] * F, B, seed=D * 1000 + B)
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    out = ops.gather_concat(g, torch.from_numpy(ids).to(dev))
    torch.cuda.synchronize()
    exp = ref.gather_concat(tables, ids)
    assert out.shape == exp.shape
    assert np.array_equal(out.cpu().numpy().view(np.uint32), exp.view(np.uint32))


def test_gather_mixed_dims_generic_path(dev):
    from recamd import ops
    vocabs = [10, 20, 30, 40]
    dims = [3, 16, 5, 130]
    tables, ids = _mk(dev, vocabs, dims, 257, seed=7)
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    out = ops.gather_concat(g, torch.from_numpy(ids).to(dev))
    exp = ref.gather_concat(tables, ids)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), exp.view(np.uint32))


def test_gather_float_ids_truncate(dev):
    from recamd import ops
    vocabs = [100, 50, 70]
    tables, idsf = _mk(dev, vocabs, [16] * 3, 300, seed=3, ids_float=True)
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    out = ops.gather_concat(g, torch.from_numpy(idsf).to(dev))
    exp = ref.gather_concat(tables, idsf)  # oracle truncates toward zero (3.7 -> 3)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("dims", [[32] * 4, [5, 32, 7, 9]])
def test_gather_oob_zero_rows_and_flag(dev, dims):
    from recamd import ops
    vocabs = [10, 20, 30, 40]
    tables, ids = _mk(dev, vocabs, dims, 513, seed=11, oob_frac=0.2)
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    flag = ops.new_oob_flag(dev)
    out = ops.gather_concat(g, torch.from_numpy(ids).to(dev), oob_flag=flag)
    exp = ref.gather_concat(tables, ids, oob="zero")
    assert np.array_equal(out.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    assert int(flag.item()) == 1
    with pytest.raises(IndexError):
        ref.gather_concat(tables, ids, oob="raise")
    # in-range ids leave the flag untouched
    flag.zero_()
    ops.gather_concat(g, torch.from_numpy(np.zeros_like(ids)).to(dev), oob_flag=flag)
    assert int(flag.item()) == 0


def test_gather_into_wider_buffer_and_strided_ids(dev):
    from recamd import ops
    vocabs = [9, 8, 7]
    tables, ids = _mk(dev, vocabs, [8] * 3, 100, seed=5)
    # ids live inside a wider (B, 6) matrix (columns 2..4), output inside a (B, 40) buffer at col 16
    wide_ids = np.full((100, 6), 3, np.int32)
    wide_ids[:, 2:5] = ids
    t_ids = torch.from_numpy(wide_ids).to(dev)[:, 2:5]
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables], out_cols=[16, 24, 32])
    buf = torch.full((100, 40), -1.0, device=dev)
    ops.gather_concat(g, t_ids, out=buf)
    got = buf.cpu().numpy()
    assert np.array_equal(got[:, 16:], ref.gather_concat(tables, ids))
    assert (got[:, :16] == -1).all()


def test_gather_nan_inf_payload_bit_exact(dev):
    from recamd import ops
    t = np.zeros((4, 8), np.float32)
    t.view(np.uint32)[1, :] = 0x7FC12345  # NaN payload
    t[2, :] = np.inf
    t[3, :] = -0.0
    ids = np.array([[1], [2], [3], [0], [9]], np.int32)  # last is OOB
    g = ops.TableGroup([torch.from_numpy(t).to(dev)])
    out = ops.gather_concat(g, torch.from_numpy(ids).to(dev)).cpu().numpy()
    exp = ref.gather_concat([t], ids)
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))


def test_cpu_tensor_fails_loudly():
    from recamd import ops
    with pytest.raises(RuntimeError):
        ops.TableGroup([torch.zeros(4, 4)])


# ---- the LDS-DMA ring kernel (D = 128, int32 ids, plain concat output, >= 2048 rows) -----------------------------
@pytest.mark.parametrize("B,F", [(79, 26), (80, 26), (1000, 26), (4099, 3), (2048, 1), (700, 5), (65, 33), (5000, 64)])
def test_gather_ring_kernel_bit_exact(dev, B, F):
    """chunks of 32 rows over persistent waves: batch sizes that leave a ragged last chunk, waves with unequal chunk
    counts, out-of-range ids (zero rows + flag), fields > 32 (a chunk inside one sample)"""
    from recamd import ops
    rng = np.random.default_rng(B * 100 + F)
    D = 128
    vocabs = [int(v) for v in rng.integers(3, 200, size=F)]
    tables = [rng.normal(size=(v, D)).astype(np.float32) for v in vocabs]
    ids = np.stack([rng.integers(0, v, size=B) for v in vocabs], axis=1).astype(np.int32)
    ids[B // 3, F // 2] = -1
    ids[B - 1, F - 1] = vocabs[F - 1]
    g = ops.TableGroup([torch.from_numpy(t).to(dev) for t in tables])
    flag = ops.new_oob_flag(dev)
    out = ops.gather_concat(g, torch.from_numpy(ids).to(dev), oob_flag=flag).cpu().numpy()
    exp = ref.gather_concat(tables, ids, oob="zero")
    assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))
    assert int(flag.item()) == 1
    flag2 = ops.new_oob_flag(dev)
    ids2 = np.clip(ids, 0, np.asarray(vocabs)[None, :] - 1).astype(np.int32)
    out2 = ops.gather_concat(g, torch.from_numpy(ids2).to(dev), oob_flag=flag2).cpu().numpy()
    assert np.array_equal(out2.view(np.uint32), ref.gather_concat(tables, ids2).view(np.uint32))
    assert int(flag2.item()) == 0


def test_place_table_arena_reports_every_candidate(dev):
    """ops.place_table_arena: N candidate arenas probed with the caller's kernel, the fastest kept, every probe time
    reported; candidates = 1 is a plain allocation without a probe"""
    from recamd import ops
    F, V, D, B = 3, 5000, 32, 512
    ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
    out = torch.empty((B, F * D), device=dev)
    calls = []

    def probe(g, i):
        calls.append(i)
        ops.gather_concat(g, ids, out=out)
    arena, info = ops.place_table_arena(F, V, D, dev, candidates=3, probe=probe, probe_launches=4, probe_name="test")
    assert tuple(arena.shape) == (F, V, D) and arena.is_contiguous() and arena.dtype == torch.float32
    assert info["candidates"] == 3 and len(info["probe_us"]) == 3 and 0 <= info["chosen"] < 3 and info["probe"] == "test"
    assert all(t > 0 for t in info["probe_us"]) and calls
    arena.uniform_(-1, 1)
    got = ops.gather_concat(ops.TableGroup([arena[f] for f in range(F)]), ids)
    exp = torch.cat([arena[f][ids[:, f].long()] for f in range(F)], dim=1)
    assert torch.equal(got, exp)
    plain, info1 = ops.place_table_arena(F, V, D, dev, candidates=1)
    assert tuple(plain.shape) == (F, V, D) and info1 == {"candidates": 1}
    dflt, info2 = ops.place_table_arena(2, 1000, 16, dev, candidates=2, probe_launches=2)   # default probe: the gather
    assert tuple(dflt.shape) == (2, 1000, 16) and len(info2["probe_us"]) == 2
    both, info3 = ops.place_table_arena(F, V, D, dev, candidates=2, probe=[probe, probe], probe_launches=2)   # several kernels
    assert tuple(both.shape) == (F, V, D) and len(info3["probe_us"]) == 2 and all(len(t) == 2 for t in info3["probe_us"])

