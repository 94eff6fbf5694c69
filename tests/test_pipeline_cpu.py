"""The input-pipeline oracle (oracle/ref_pipeline.py) PINNED against the real scikit-learn the reference calls
(src/ctr/utils/data_process.py:66-68, :76-78): LabelEncoder on the string columns, MinMaxScaler on astype(int) values."""
import numpy as np
import pytest

from oracle import ref_pipeline as rp

sk = pytest.importorskip("sklearn.preprocessing")


def _column(rng, n, card, missing=0.1):
    toks = rng.integers(0, 2 ** 32 - 2, size=card, dtype=np.uint64).astype(np.uint32)
    col = toks[rng.integers(0, card, size=n)]
    strs = np.array([f"{int(t):08x}" for t in col], dtype=object)
    strs[rng.random(n) < missing] = np.nan
    return strs


@pytest.mark.parametrize("n,card", [(1, 1), (50, 7), (5000, 300), (2000, 2000)])
def test_label_encoder_matches_sklearn(n, card):
    import pandas as pd
    rng = np.random.default_rng(n + card)
    strs = _column(rng, n, card)
    s = pd.Series(strs).fillna('-1').astype(str)                # data_process.py:63 + :68
    exp = sk.LabelEncoder().fit_transform(s)
    tok = rp.hex_tokens(strs)
    vocab = rp.label_encode_fit(tok)
    got = rp.label_encode_transform(vocab, tok)
    assert np.array_equal(got, exp.astype(np.int32))
    assert len(vocab) == len(np.unique(s))                       # feat_num of sparseFeature (:81)


def test_minmax_matches_sklearn():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.integers(-3, 5000, size=(400, 5)).astype(np.float64) + rng.random((400, 5)) * 0.9 * np.sign(rng.normal(size=(400, 5))),
                        np.full((400, 1), 7.3)], axis=1).astype(np.float32)       # last column constant
    exp = sk.MinMaxScaler().fit_transform(x.astype(int)).astype(np.float32)       # per column, as intended
    mn, mx = rp.minmax_fit(x)
    got = rp.minmax_transform(x, mn, mx)
    assert np.array_equal(got, exp)


def test_pad_sequences_semantics():
    seqs = [[1, 2, 3], [], [4, 5, 6, 7, 8, 9], [10]]
    out = rp.pad_sequences(seqs, 4)
    assert out.tolist() == [[0, 1, 2, 3], [0, 0, 0, 0], [6, 7, 8, 9], [0, 0, 0, 10]]     # pre-pad, keep the LAST items
    assert rp.pad_sequences(seqs, 4, padding="post", truncating="post").tolist() == \
        [[1, 2, 3, 0], [0, 0, 0, 0], [4, 5, 6, 7], [10, 0, 0, 0]]
