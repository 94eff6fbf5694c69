"""Edge cases the reference's ops define: empty batches, single rows, maximum table count per
launch (and beyond: chunked), ragged tails, huge strides."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_empty_batch_everywhere(dev):
    from recamd import ops
    tabs = [torch.rand((10, 8), device=dev) for _ in range(3)]
    g = ops.TableGroup(tabs)
    ids = torch.zeros((0, 3), dtype=torch.int32, device=dev)
    assert ops.gather_concat(g, ids).shape == (0, 24)
    assert ops.pairwise_dot(torch.zeros((0, 4, 16), device=dev)).shape == (0, 6)
    assert ops.fm_layer(torch.zeros((0, 5), device=dev), torch.zeros((0, 3), device=dev), torch.zeros((5, 1), device=dev)).shape == (0, 1)
    assert ops.cross_network(torch.zeros((0, 8), device=dev), torch.zeros((2, 8), device=dev), torch.zeros((2, 8), device=dev)).shape == (0, 8)
    assert ops.dense(torch.zeros((0, 8), device=dev), torch.zeros((8, 4), device=dev)).shape == (0, 4)
    torch.cuda.synchronize()


@pytest.mark.parametrize("F", [64, 65, 130])
def test_many_tables_chunked_launches(dev, F):
    """REC_MAX_TABLES = 64 per launch; wider models are chunked by the host (same result)."""
    from recamd import ops
    rng = np.random.default_rng(F)
    tables = [rng.normal(size=(17, 8)).astype(np.float32) for _ in range(F)]
    ids = rng.integers(0, 17, size=(50, F)).astype(np.int32)
    g = ops.TableGroup([T(t, dev) for t in tables])
    out = ops.gather_concat(g, T(ids, dev)).cpu().numpy()
    assert np.array_equal(out, ref.gather_concat(tables, ids))


def test_ragged_tail_rows_not_multiple_of_wave(dev):
    """B*F not a multiple of 64 / of the rows-per-instruction: the tail chunk takes the masked path."""
    from recamd import ops
    rng = np.random.default_rng(5)
    for B, F, D in [(3, 7, 128), (65, 1, 64), (1, 1, 4), (129, 5, 16)]:
        tables = [rng.normal(size=(9, D)).astype(np.float32) for _ in range(F)]
        ids = rng.integers(0, 9, size=(B, F)).astype(np.int32)
        g = ops.TableGroup([T(t, dev) for t in tables])
        assert np.array_equal(ops.gather_concat(g, T(ids, dev)).cpu().numpy(), ref.gather_concat(tables, ids))


def test_id_extremes(dev):
    """INT32_MIN / INT32_MAX / huge floats / NaN ids are out of range -> zero rows + flag, never a fault."""
    from recamd import ops
    t = np.arange(40, dtype=np.float32).reshape(10, 4)
    g = ops.TableGroup([T(t, dev)])
    ids = np.array([[np.iinfo(np.int32).min], [np.iinfo(np.int32).max], [9], [0]], np.int32)
    flag = ops.new_oob_flag(dev)
    out = ops.gather_concat(g, T(ids, dev), oob_flag=flag).cpu().numpy()
    assert np.array_equal(out, np.stack([np.zeros(4), np.zeros(4), t[9], t[0]]).astype(np.float32))
    assert int(flag.item()) == 1
    idsf = np.array([[1e30], [-1e30], [np.nan], [np.inf], [9.999], [-0.9]], np.float32)
    out = ops.gather_concat(g, T(idsf, dev)).cpu().numpy()
    exp = np.stack([np.zeros(4)] * 4 + [t[9], t[0]]).astype(np.float32)   # 9.999 -> 9, -0.9 -> 0 (trunc toward zero)
    assert np.array_equal(out, exp)


def test_single_row_tables_and_vocab_one(dev):
    from recamd import ops
    t = np.array([[1.0, 2.0, 3.0, 4.0]], np.float32)
    g = ops.TableGroup([T(t, dev), T(t * 2, dev)])
    ids = np.zeros((5, 2), np.int32)
    assert np.array_equal(ops.gather_concat(g, T(ids, dev)).cpu().numpy(), np.tile(np.concatenate([t[0], 2 * t[0]]), (5, 1)))


def test_large_batch_gather_int64_row_count(dev):
    """R = B*F beyond 2^24 rows (32-bit divide path still exact) with tiny tables."""
    from recamd import ops
    B, F, D = 1_200_000, 16, 4
    tables = [torch.arange(8 * D, dtype=torch.float32, device=dev).view(8, D) + 100 * f for f in range(F)]
    g = ops.TableGroup(tables)
    ids = torch.randint(0, 8, (B, F), device=dev, dtype=torch.int32)
    out = ops.gather_concat(g, ids)
    exp = torch.cat([tables[f][ids[:, f].long()] for f in range(F)], dim=1)
    assert torch.equal(out, exp)
