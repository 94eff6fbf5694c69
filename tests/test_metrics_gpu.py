"""§8f-2 first slice: Keras binary_crossentropy and AUC() (what every ctr train script compiles/evaluates) vs oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref

pytestmark = pytest.mark.gpu


def data(n, seed, sharp=False):
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < 0.3).astype(np.float32)
    logit = rng.normal(size=n) + (2.0 if sharp else 0.7) * (2 * y - 1)
    p = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    return y, p


@pytest.mark.parametrize("n", [1, 2, 1000, 300_001])
def test_bce_and_auc_match_keras_formulas(dev, n):
    from recamd import ops
    y, p = data(n, n)
    ty, tp = torch.from_numpy(y).to(dev), torch.from_numpy(p).to(dev)
    bce = float(ops.binary_crossentropy(ty, tp).cpu())
    auc = float(ops.auc(ty, tp).cpu())
    assert abs(bce - ref.binary_crossentropy(y, p)) <= 1e-5 * max(1.0, abs(ref.binary_crossentropy(y, p)))
    assert abs(auc - ref.keras_auc(y, p)) <= 1e-5


def test_extreme_predictions_and_threshold_edges(dev):
    """p exactly 0, 1 and exactly on thresholds: clipping in BCE, strict `>` in the AUC binning."""
    from recamd import ops
    thr = np.array([(i + 1) / 199.0 for i in range(198)], np.float32)
    p = np.concatenate([np.array([0.0, 1.0, 0.0, 1.0], np.float32), thr, np.nextafter(thr, np.float32(2.0))])
    rng = np.random.default_rng(0)
    y = (rng.random(p.size) < 0.5).astype(np.float32)
    ty, tp = torch.from_numpy(y).to(dev), torch.from_numpy(p).to(dev)
    assert abs(float(ops.auc(ty, tp).cpu()) - ref.keras_auc(y, p)) <= 1e-6
    e = ref.binary_crossentropy(y, p)
    assert abs(float(ops.binary_crossentropy(ty, tp).cpu()) - e) <= 1e-5 * max(1.0, e)


def test_auc_degenerate_labels(dev):
    from recamd import ops
    y, p = data(500, 3)
    for lab in (np.zeros_like(y), np.ones_like(y)):
        got = float(ops.auc(torch.from_numpy(lab).to(dev), torch.from_numpy(p).to(dev)).cpu())
        assert abs(got - ref.keras_auc(lab, p)) <= 1e-6


def test_perfect_and_random_rankings(dev):
    from recamd import ops
    y, p = data(20_000, 9, sharp=True)
    perfect = np.where(y > 0, 0.9, 0.1).astype(np.float32)
    assert abs(float(ops.auc(torch.from_numpy(y).to(dev), torch.from_numpy(perfect).to(dev)).cpu()) - 1.0) <= 1e-6
    const = np.full_like(y, 0.5)
    assert abs(float(ops.auc(torch.from_numpy(y).to(dev), torch.from_numpy(const).to(dev)).cpu()) - 0.5) <= 1e-6


@pytest.mark.parametrize("B,n,stride_pad", [(1, 1, 0), (37, 100, 0), (513, 7, 3), (4096, 100, 0)])
def test_pairwise_rank_loss(dev, B, n, stride_pad):
    """rec_pairwise_rank_loss_f32 vs the restated add_loss of src/match/sasrec/model.py:93-95 (also NCF :75-77)."""
    from recamd import ops
    rng = np.random.default_rng(B + n)
    lg = (rng.normal(size=(B, 1 + n)) * 2).astype(np.float32)
    wide = torch.zeros((B, 1 + n + stride_pad), dtype=torch.float32, device=dev)
    wide[:, :1 + n] = torch.from_numpy(lg).to(dev)
    got = float(ops.pairwise_rank_loss(wide[:, :1 + n])[0])
    exp = float(ref.pairwise_rank_loss(lg))
    assert abs(got - exp) <= 1e-5 * max(1.0, abs(exp))
    # the sasrec oracle computes the same number from its own logits
    assert abs(exp - float(np.mean(-np.log(ref.sigmoid(lg[:, :1].astype(np.float64))) -
                                   np.log(1 - ref.sigmoid(lg[:, 1:].astype(np.float64)))) / 2)) < 1e-12
