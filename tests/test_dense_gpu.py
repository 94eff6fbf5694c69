"""K11 parity: Dense on fp32 MFMA vs numpy fp64 (1e-5 relative)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(1, 1, 1), (5, 13, 64), (257, 479, 1024), (1000, 3341, 128), (130, 64, 1), (300, 256, 8),
                                   (129, 16, 130), (4096, 128, 64), (64, 100, 9)])
@pytest.mark.parametrize("act", [None, "relu", "sigmoid", "prelu"])
def test_dense(dev, M, K, N, act):
    from recamd import ops
    rng = np.random.default_rng(M + K + N)
    x = (rng.normal(size=(M, K)) * 0.5).astype(np.float32)
    W = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    alpha = rng.random(N).astype(np.float32) if act == "prelu" else None
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    out = ops.dense(t(x), t(W), t(b), act, None if alpha is None else t(alpha)).cpu().numpy()
    exp = ref.dense(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), act, alpha)
    assert close(out, exp)


def test_dense_asymmetric_layout(dev):
    """A = I with an ASYMMETRIC W catches a transposed C write (cdna guide §3)."""
    from recamd import ops
    K = N = 96
    W = np.arange(K * N, dtype=np.float32).reshape(K, N)
    x = np.eye(K, dtype=np.float32)
    out = ops.dense(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev)).cpu().numpy()
    assert np.array_equal(out, W)


def test_dense_rank3_and_strided_input(dev):
    from recamd import ops
    rng = np.random.default_rng(3)
    x = rng.normal(size=(6, 10, 16)).astype(np.float32)
    W = rng.normal(size=(16, 32)).astype(np.float32)
    out = ops.dense(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev))
    assert out.shape == (6, 10, 32)
    assert close(out.cpu().numpy(), ref.dense(x.astype(np.float64), W.astype(np.float64)))
    buf = rng.normal(size=(50, 40)).astype(np.float32)
    tb = torch.from_numpy(buf).to(dev)
    W2 = rng.normal(size=(37, 20)).astype(np.float32)
    out2 = ops.dense(tb[:, 3:], torch.from_numpy(W2).to(dev)).cpu().numpy()
    assert close(out2, buf[:, 3:].astype(np.float64) @ W2.astype(np.float64))
