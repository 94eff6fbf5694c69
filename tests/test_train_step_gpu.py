"""§8f-1: embedding backward (scatter-add with duplicate ids) + dense Keras-Adam step vs the oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("B,vocabs,dims", [(1, [5], [4]), (500, [7, 100, 3], [8, 8, 8]), (300, [10, 20], [5, 130]),
                                           (2000, [50] * 26, [16] * 26)])
def test_embedding_grad_scatter_add(dev, B, vocabs, dims):
    from recamd import ops
    rng = np.random.default_rng(B)
    ids = np.stack([rng.integers(-1, v + 1, size=B) for v in vocabs], axis=1).astype(np.int32)  # duplicates + OOB
    dy = rng.normal(size=(B, sum(dims))).astype(np.float32)
    gt = [torch.zeros((v, d), device=dev) for v, d in zip(vocabs, dims)]
    ops.embedding_grad(ops.TableGroup(gt), T(ids, dev), T(dy, dev))
    exp = ref.embedding_grad(ids, dy, vocabs, dims)
    for g, e in zip(gt, exp):
        assert close(g.cpu().numpy(), e, 1e-5)


@pytest.mark.parametrize("n", [1, 7, 1024, 100_003])
@pytest.mark.parametrize("l2", [0.0, 1e-4])
def test_adam_step_matches_keras_formula(dev, n, l2):
    from recamd import ops
    rng = np.random.default_rng(n)
    var = rng.normal(size=n).astype(np.float32) * 0.05
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    tv, tm, tvv = T(var, dev), T(m, dev), T(v, dev)
    ev, em, evv = var.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in (1, 2, 3):
        g = (rng.normal(size=n) * (rng.random(n) < 0.3)).astype(np.float32)   # sparse-ish gradient
        ops.adam_step(tv, tm, tvv, T(g, dev), step, lr=1e-3, l2=l2)
        ev, em, evv = ref.adam_step(ev, em, evv, g, step, lr=1e-3, l2=l2)
    assert close(tv.cpu().numpy(), ev, 1e-5)
    assert close(tm.cpu().numpy(), em, 1e-5)
    assert close(tvv.cpu().numpy(), evv, 1e-5)


def test_embedding_training_step_end_to_end(dev):
    """forward gather -> upstream gradient -> scatter-add -> Adam on the table, twice, vs the oracle."""
    from recamd import ops
    rng = np.random.default_rng(0)
    V, D, F, B = 40, 16, 3, 128
    tables = [rng.uniform(-0.05, 0.05, size=(V, D)).astype(np.float32) for _ in range(F)]
    tt = [T(t, dev) for t in tables]
    mm = [torch.zeros_like(t) for t in tt]
    vv = [torch.zeros_like(t) for t in tt]
    et = [t.astype(np.float64) for t in tables]
    em = [np.zeros_like(t) for t in et]
    ev = [np.zeros_like(t) for t in et]
    for step in (1, 2):
        ids = rng.integers(0, V, size=(B, F)).astype(np.int32)
        out = ops.gather_concat(ops.TableGroup(tt), T(ids, dev))
        dy = (2.0 * out).contiguous()                                  # d/dx of sum(x^2)
        gt = [torch.zeros_like(t) for t in tt]
        ops.embedding_grad(ops.TableGroup(gt), T(ids, dev), dy)
        for f in range(F):
            ops.adam_step(tt[f], mm[f], vv[f], gt[f], step, lr=1e-2, l2=1e-4)
        eout = ref.gather_concat(et, ids)
        eg = ref.embedding_grad(ids, 2.0 * eout, [V] * F, [D] * F)
        for f in range(F):
            et[f], em[f], ev[f] = ref.adam_step(et[f], em[f], ev[f], eg[f], step, lr=1e-2, l2=1e-4)
    for f in range(F):
        assert close(tt[f].cpu().numpy(), et[f], 2e-5)
