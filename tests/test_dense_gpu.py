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


def _h2_case(rng, M, K, N, xscale=None, wscale=None):
    x = rng.normal(size=(M, K)).astype(np.float32)
    W = (rng.normal(size=(K, N)) / np.sqrt(K)).astype(np.float32)
    if xscale is not None:
        x = (x * xscale).astype(np.float32)
    if wscale is not None:
        W = (W * wscale).astype(np.float32)
    return x, W, rng.normal(size=N).astype(np.float32)


@pytest.mark.parametrize("M,K,N", [(1024, 32, 128), (1500, 64, 130), (2048, 512, 256), (1100, 1024, 200), (1024, 2048, 136), (4099, 96, 12)])
@pytest.mark.parametrize("act", [None, "relu", "prelu"])
def test_dense_f16x2_matches_the_oracle_and_the_other_kernels(dev, force, M, K, N, act):
    """csrc/dense_f16x2.hip (the default for large layers with aligned rows and K % 32 == 0): rows of x and columns of W scaled
    by exact powers of two, two f16 terms each, three MFMAs per product.  Against fp64 it must be at the level of the fp32-MFMA
    kernel (an fmaf chain) and of the bf16x3 kernel, on ragged tiles, with bias / activations, and deterministic."""
    from recamd import ops
    rng = np.random.default_rng(M + K + N)
    x, W, b = _h2_case(rng, M, K, N)
    alpha = rng.random(N).astype(np.float32) if act == "prelu" else None
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)  # noqa: E731
    tx, tw, tb, ta = t(x), t(W), t(b), t(alpha)
    exp = ref.dense(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64), act, alpha)
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64) + np.abs(b)
    force("dense_pipe", "h")                                                 # f16x2 whatever N (by default: N >= 384, or row maxima handed over)
    got = ops.dense(tx, tw, tb, act, ta).cpu().numpy()
    again = ops.dense(tx, tw, tb, act, ta).cpu().numpy()
    assert np.array_equal(got, again)
    assert close_scaled(got, exp, scale)
    force("dense_pipe", "s")
    got_b = ops.dense(tx, tw, tb, act, ta).cpu().numpy()                     # bf16x3, hand-counted
    force("dense_pipe", None)
    force("dense", "f")
    got_f = ops.dense(tx, tw, tb, act, ta).cpu().numpy()                     # fp32 MFMA
    if N > 12:
        assert not np.array_equal(got, got_b)                                # it really is another kernel
    err, err_b, err_f = (np.abs(g - exp).max() for g in (got, got_b, got_f))
    assert err <= 2.0 * max(err_f, err_b) + 1e-7, (err, err_b, err_f)


def test_dense_f16x2_ranges(dev, force):
    """Per-row and per-column scaling: rows of x 2^-60 .. 2^60 apart and columns of W 2^-30 .. 2^30 apart keep fp32-level
    accuracy relative to their own magnitudes; values spanning 2^-20 .. 2^10 INSIDE a row keep the bound of the bf16x3 test;
    a zero row and a zero column give exact zeros (+ bias)."""
    from recamd import ops
    force("dense_pipe", "h")
    rng = np.random.default_rng(77)
    M, K, N = 1024, 128, 96
    x, W, b = _h2_case(rng, M, K, N, xscale=np.exp2(rng.integers(-60, 61, size=(M, 1))), wscale=np.exp2(rng.integers(-30, 31, size=(1, N))))
    x[5] = 0.0
    W[:, 7] = 0.0
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    got = ops.dense(t(x), t(W)).cpu().numpy().astype(np.float64)
    exp = x.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64)
    assert np.all(got[5] == 0.0) and np.all(got[:, 7] == 0.0)
    assert np.all(np.abs(got - exp) <= 2e-6 * scale + 1e-300)
    x2 = (rng.normal(size=(M, K)) * np.exp2(rng.integers(-20, 11, size=(M, K)))).astype(np.float32)
    W2 = (rng.normal(size=(K, N)) * np.exp2(rng.integers(-10, 3, size=(K, N)))).astype(np.float32)
    got2 = ops.dense(t(x2), t(W2)).cpu().numpy()
    exp2 = x2.astype(np.float64) @ W2.astype(np.float64)
    scale2 = np.abs(x2.astype(np.float64)) @ np.abs(W2.astype(np.float64))
    assert np.all(np.abs(got2 - exp2) <= 2e-6 * scale2 + 1e-30)


def test_dense_f16x2_extreme_columns_and_rows(dev, force):
    """one ldexp by the SUM of the row and column exponents scales back: a column of W around 2^100 against rows of x around
    2^-100 gives its O(1) results (a two-step scale would overflow on the way), and a product beyond fp32's range overflows to
    inf exactly where fp32 arithmetic does"""
    from recamd import ops
    force("dense_pipe", "h")
    rng = np.random.default_rng(78)
    M, K, N = 1024, 64, 128
    x, W, _ = _h2_case(rng, M, K, N)
    W[:, 3] *= np.float32(2.0 ** 100)
    x[10] *= np.float32(2.0 ** -100)
    x[11] *= np.float32(2.0 ** 100)
    got = ops.dense(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev)).cpu().numpy().astype(np.float64)
    exp = x.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64)
    fin = np.abs(exp) < 3.0e38
    assert np.all(np.abs(got[fin] - exp[fin]) <= 2e-6 * scale[fin] + 1e-300)
    assert 0.01 < abs(exp[10, 3]) < 100 and np.isinf(got[11, 3]) and not fin[11, 3]


@pytest.mark.parametrize("bad", [np.inf, -np.inf, np.nan])
def test_dense_f16x2_nonfinite_rows(dev, force, bad):
    """the contract of tests/test_nonfinite_gpu.py on the default path: a poisoned row is non-finite everywhere, every other
    row is bit-identical to the clean run; fp32 denormals are lost to within their own magnitude"""
    from recamd import ops
    force("dense_pipe", "h")
    rng = np.random.default_rng(3)
    M, K, N = 2048, 256, 128
    x, W, _ = _h2_case(rng, M, K, N)
    W[7, :] = 0.5
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    clean = ops.dense(t(x), t(W)).cpu().numpy()
    xb = x.copy()
    xb[100, 7] = bad
    xb[1500, 200] = bad
    got = ops.dense(t(xb), t(W)).cpu().numpy()
    assert not np.isfinite(got[100]).any() and not np.isfinite(got[1500]).any()
    rest = np.ones(M, bool)
    rest[[100, 1500]] = False
    assert np.array_equal(got[rest].view(np.uint32), clean[rest].view(np.uint32))
    xd = x.copy()
    xd[:, :16] = (rng.random((M, 16)) * 1e-39).astype(np.float32)
    gd = ops.dense(t(xd), t(W)).cpu().numpy()
    ed = xd.astype(np.float64) @ W.astype(np.float64)
    assert np.isfinite(gd).all()
    assert close_scaled(gd, ed, np.abs(xd).astype(np.float64) @ np.abs(W).astype(np.float64))


def test_dense_chain_hands_the_row_maxima_along(dev, force):
    """A tower (nn.dense_chain): every large layer's epilogue delivers max |out[r, :]|, the next layer scales by it instead of
    re-reading its input.  The chained result equals the layers applied one by one with the scaled kernel forced (the maxima
    are the same numbers either way: bit-identical), the maxima equal numpy's, and everything agrees with the fp64 oracle."""
    from recamd import nn, ops
    rng = np.random.default_rng(79)
    M, widths = 1500, [96, 128, 64, 160, 8]
    x = rng.normal(size=(M, widths[0])).astype(np.float32)
    layers = []
    for i in range(1, len(widths)):
        d = nn.Dense(widths[i], activation='relu' if i < len(widths) - 1 else None)
        d.build(widths[i - 1])
        d.set_weights({"kernel": (rng.normal(size=(widths[i - 1], widths[i])) / np.sqrt(widths[i - 1])).astype(np.float32),
                       "bias": rng.normal(size=widths[i]).astype(np.float32) * 0.1})
        layers.append(d)
    tx = torch.from_numpy(x).to(dev)
    am = torch.zeros(M, device=dev)
    y1 = ops.dense(tx, layers[0]._w["kernel"], layers[0]._w["bias"], "relu", out_absmax=am)
    assert np.array_equal(am.cpu().numpy(), np.abs(y1.cpu().numpy()).max(axis=1))
    force("dense_pipe", "h")                      # the scaled kernel for every layer, with or without maxima handed over
    chained_h = nn.dense_chain(layers, tx).cpu().numpy()
    amh = torch.zeros(M, device=dev)
    y1h = ops.dense(tx, layers[0]._w["kernel"], layers[0]._w["bias"], "relu", out_absmax=amh)
    assert np.array_equal(amh.cpu().numpy(), np.abs(y1h.cpu().numpy()).max(axis=1))      # from the epilogue's atomics
    h = tx
    for d in layers:
        h = d(h)
    assert np.array_equal(chained_h, h.cpu().numpy())
    force("dense_pipe", None)
    chained = nn.dense_chain(layers, tx).cpu().numpy()     # default dispatch: the first layer has no maxima and N < 384
    e = x.astype(np.float64)
    for i, d in enumerate(layers):
        w = d.get_weights()
        e = e @ w["kernel"].astype(np.float64) + w["bias"]
        if i < len(layers) - 1:
            e = np.maximum(e, 0)
    assert close(chained, e, 1e-5) and close(chained_h, e, 1e-5)


@pytest.mark.parametrize("case", ["all_positive", "cancelling", "heavy_tailed", "relu_sparse"])
def test_dense_f16x2_error_against_the_magnitude_it_summed(dev, force, case):
    """The f16x2 kernel's a-priori bound: per product it drops l_x l_w and the residual of the two-term split, each
    <= 2^-22 |x||w| (round to nearest), and accumulates in fp32 like every kernel here — so |err| <= c * sum_k |x||w| with c a
    small multiple of 2^-22 ~ 2.4e-7 from the split, plus the fp32 accumulation every kernel shares.  Worst-case style inputs: all-positive operands at
    K = 4096 (nothing cancels in the error), rows that cancel to ~0 (the error stays relative to what was summed, not to the
    result), log-normal magnitudes, and relu-sparse activations.  The fp32-MFMA kernel (an fmaf chain) is held to the same bound
    and the f16x2 error may not exceed twice its error."""
    from recamd import ops
    rng = np.random.default_rng({"all_positive": 1, "cancelling": 2, "heavy_tailed": 3, "relu_sparse": 4}[case])
    M, K, N = 1024, 4096, 128
    if case == "all_positive":
        x, W = rng.random((M, K)).astype(np.float32), rng.random((K, N)).astype(np.float32)
    elif case == "cancelling":
        h = rng.normal(size=(M, K // 2)).astype(np.float32)
        x = np.concatenate([h, -h], axis=1)
        w = rng.normal(size=(K // 2, N)).astype(np.float32)
        W = np.concatenate([w, w * np.float32(1 + 2.0 ** -12)], axis=0)       # x W = -2^-12 h w: ~4000 times smaller than its terms
    elif case == "heavy_tailed":
        x = (rng.normal(size=(M, K)) * np.exp(rng.normal(size=(M, K)) * 3)).astype(np.float32)
        W = (rng.normal(size=(K, N)) * np.exp(rng.normal(size=(K, N)) * 2)).astype(np.float32)
    else:
        x = np.maximum(rng.normal(size=(M, K)) - 1.0, 0).astype(np.float32)     # ~84 % zeros
        W = (rng.normal(size=(K, N)) / 64).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    exp = x.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(x).astype(np.float64) @ np.abs(W).astype(np.float64)
    force("dense_pipe", "h")
    got = ops.dense(t(x), t(W)).cpu().numpy().astype(np.float64)
    force("dense_pipe", None)
    force("dense", "f")
    got_f = ops.dense(t(x), t(W)).cpu().numpy().astype(np.float64)
    ratio = lambda g: float((np.abs(g - exp) / np.maximum(scale, 1e-300)).max())  # noqa: E731
    r, rf = ratio(got), ratio(got_f)
    # fp32 accumulation of same-signed terms is the larger part (a-priori K 2^-24 relative to the magnitude summed; measured
    # 2.4e-6 for f16x2 against 3.7 - 4.1e-6 for the fmaf chain on the all-positive and relu cases): f16x2 stays within the fp32
    # kernel's error plus its own 2^-22, and both far inside the a-priori bound
    assert rf <= K * 2.0 ** -24 and r <= K * 2.0 ** -24, (r, rf)
    assert r <= 1.5 * rf + 2.4e-7, (r, rf)
