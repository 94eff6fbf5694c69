"""K11 parity: Dense vs numpy fp64 at the kernel-level tolerance of tests/util.py:
|a-b| <= 1e-5 * max(|b|, 1e-3) + 2.5e-7 * (|x| @ |W| + |b|)  (the magnitude each output was accumulated from)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close, close_scaled

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
    assert close_scaled(out, exp, np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(b))


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
    assert close_scaled(out.cpu().numpy(), ref.dense(x.astype(np.float64), W.astype(np.float64)),
                        np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64))
    buf = rng.normal(size=(50, 40)).astype(np.float32)
    tb = torch.from_numpy(buf).to(dev)
    W2 = rng.normal(size=(37, 20)).astype(np.float32)
    out2 = ops.dense(tb[:, 3:], torch.from_numpy(W2).to(dev)).cpu().numpy()
    assert close_scaled(out2, buf[:, 3:].astype(np.float64) @ W2.astype(np.float64),
                        np.abs(buf[:, 3:]).astype(np.float64) @ np.abs(W2).astype(np.float64))


@pytest.mark.parametrize("M,K,N,aligned", [(2048, 512, 256, True), (1500, 479, 130, False), (1024, 42, 200, False),
                                           (4096, 64, 128, True)])
def test_dense_bf16x3_has_fp32_accuracy(dev, force, M, K, N, aligned):
    """The large-layer kernel rebuilds fp32 products from three bf16 terms (csrc/dense_bf16x3.hip): its error against
    the fp64 oracle must be at the level of the fp32-MFMA kernel's (forced: "dense" "f"), and the three x staging paths
    (aligned rows / unaligned rows via the transpose tile / short-K scalar loads) must agree with the oracle."""
    from recamd import ops
    rng = np.random.default_rng(K)
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    tx = t(x)
    assert (tx.stride(0) % 4 == 0) == aligned
    exp = ref.dense(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), "relu")
    force("dense", "b")
    got_b = ops.dense(tx, t(W), t(b), "relu").cpu().numpy()
    force("dense", "f")
    got_f = ops.dense(tx, t(W), t(b), "relu").cpu().numpy()
    err_b, err_f = np.abs(got_b - exp).max(), np.abs(got_f - exp).max()
    assert close_scaled(got_b, exp, np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(b))
    assert err_b <= 2.0 * err_f + 1e-7, (err_b, err_f)


def test_dense_bf16x3_tiny_and_mixed_magnitudes(dev, force):
    """values spanning 2^-20 .. 2^10 in one row: the hi/mid/lo split is exact per element, so small terms
    next to large ones keep their fp32 contribution"""
    from recamd import ops
    force("dense", "b")
    rng = np.random.default_rng(1)
    M, K, N = 1024, 128, 96
    x = (rng.normal(size=(M, K)) * np.exp2(rng.integers(-20, 11, size=(M, K)))).astype(np.float32)
    W = (rng.normal(size=(K, N)) * np.exp2(rng.integers(-10, 3, size=(K, N)))).astype(np.float32)
    got = ops.dense(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev)).cpu().numpy()
    exp = x.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(x.astype(np.float64)) @ np.abs(W.astype(np.float64))
    assert np.all(np.abs(got - exp) <= 2e-6 * scale + 1e-30)   # a-priori fp32 bound is K * 2^-24 = 7.6e-6


@pytest.mark.parametrize("M,K,N", [(479, 8192, 1024), (13, 8192, 512), (128, 2048, 9), (64, 5000, 64), (300, 2049, 130), (1, 600, 1)])
def test_dense_splitk(dev, M, K, N):
    """rec_dense_splitk_f32: x W with the reduction axis spread over gridDim.y on the bf16x3 kernel, partials summed in a
    fixed order — the weight-gradient shape (few output tiles, K = the batch).  fp32-accurate like the forward kernel."""
    import torch
    from recamd._lib import C
    rng = np.random.default_rng(M + N)
    x, W = rng.normal(size=(M, K)).astype(np.float32), rng.normal(size=(K, N)).astype(np.float32)
    tx, tw = torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev)
    out = torch.empty((M, N), device=dev)
    ws = torch.empty(int(C.dense_splitk_workspace_bytes(M, K, N)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    C.dense_splitk_f32(tx.data_ptr(), tx.stride(0), tw.data_ptr(), M, K, N, out.data_ptr(), ws.data_ptr(), st)
    exp = x.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64)
    from tests.util import close_scaled
    assert close_scaled(out.cpu().numpy(), exp, scale)
    again = torch.empty_like(out)
    C.dense_splitk_f32(tx.data_ptr(), tx.stride(0), tw.data_ptr(), M, K, N, again.data_ptr(), ws.data_ptr(), st)
    assert torch.equal(out, again)                              # deterministic


@pytest.mark.parametrize("M,K,N", [(1, 32, 1), (129, 32, 130), (300, 64, 128), (257, 96, 257), (1000, 480, 1024), (515, 1024, 200),
                                   (128, 2048, 128), (200, 3360, 136), (70, 4128, 64)])
def test_dense_hand_counted_pipelines(dev, force, M, K, N):
    """Aligned x rows, K % 32 == 0: the two hand-counted kernels (W planes by LDS-DMA, x by whole cache lines; "s" standard,
    "d" with the fragments one step ahead in registers — the dispatch takes it from K = 2048) sum each output's terms in
    the order of the compiler-scheduled kernel ("0"): bit-identical, and all agree with the fp64 oracle.  One to many
    rounds, ragged M and N tiles, a row stride wider than K."""
    from recamd import ops
    rng = np.random.default_rng(M * 7 + K + N)
    wide = rng.normal(size=(M, K + 8)).astype(np.float32)
    W = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(size=N).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    tx = t(wide)[:, :K]                                   # stride K + 8: still 16-B aligned rows
    assert tx.stride(0) % 4 == 0
    tw, tb = t(W), t(b)
    force("dense", "b")
    got = {}
    for arm in ("0", "s", "d"):
        force("dense_pipe", arm)
        got[arm] = ops.dense(tx, tw, tb, "relu").cpu().numpy()
    force("dense_pipe", None)
    default = ops.dense(tx, tw, tb, "relu").cpu().numpy()
    assert np.array_equal(got["s"], got["0"]) and np.array_equal(got["d"], got["0"]) and np.array_equal(default, got["0"])
    x = wide[:, :K]
    exp = ref.dense(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), "relu")
    assert close_scaled(got["s"], exp, np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(b))
