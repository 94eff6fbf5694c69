"""§8f-3 on the device: label encoding (bit-exact ids), hashing, min-max scaling, pad_sequences, the pinned
double-buffered feeder — against oracle/ref_pipeline.py, which tests/test_pipeline_cpu.py pins to scikit-learn."""
import numpy as np
import pytest
import torch

from oracle import ref_pipeline as rp

pytestmark = pytest.mark.gpu


def _tokens(rng, n, F, card):
    cols = []
    for f in range(F):
        vocab = rng.integers(0, 2 ** 32 - 2, size=card + f, dtype=np.uint64).astype(np.uint32)
        col = vocab[rng.integers(0, len(vocab), size=n)]
        col[rng.random(n) < 0.07] = rp.MISSING
        cols.append(col)
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("n,F,card", [(1, 1, 1), (300, 3, 10), (5000, 26, 400), (4000, 70, 50)])
def test_label_encode_bit_exact(dev, n, F, card):
    from recamd.pipeline import LabelEncoder
    rng = np.random.default_rng(n + F)
    tok = _tokens(rng, n, F, card)
    enc = LabelEncoder(dev).fit(tok)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    ids = enc.transform(torch.from_numpy(tok.view(np.int32)).to(dev), unseen_flag=flag).cpu().numpy()
    for f in range(F):
        vocab = rp.label_encode_fit(tok[:, f])
        assert np.array_equal(enc.vocab_host[f], vocab)
        assert np.array_equal(ids[:, f], rp.label_encode_transform(vocab, tok[:, f]))
    assert int(flag.item()) == 0
    # an unseen token: id -1 and the flag (sklearn raises)
    other = tok.copy()
    other[0, 0] = np.uint32(0xFFFFFFFE)
    if not np.any(enc.vocab_host[0] == np.uint32(0xFFFFFFFE)):
        ids2 = enc.transform(torch.from_numpy(other.view(np.int32)).to(dev), unseen_flag=flag).cpu().numpy()
        assert ids2[0, 0] == -1 and int(flag.item()) == 1


def test_hash_ids_in_range_and_deterministic(dev):
    from recamd.pipeline import hash_ids
    rng = np.random.default_rng(4)
    tok = _tokens(rng, 3000, 5, 100)
    sizes = [7, 1000, 1, 65536, 12345]
    t = torch.from_numpy(tok.view(np.int32)).to(dev)
    a, b = hash_ids(t, sizes, seed=9).cpu().numpy(), hash_ids(t, sizes, seed=9).cpu().numpy()
    assert np.array_equal(a, b)
    for f, v in enumerate(sizes):
        assert a[:, f].min() >= 0 and a[:, f].max() < v
    # equal tokens hash equally, and the hash spreads: a 1000-bucket column uses most buckets
    same = tok[:, 1] == tok[0, 1]
    assert np.all(a[same, 1] == a[0, 1])
    assert len(np.unique(a[:, 3])) > 90
    assert not np.array_equal(a, hash_ids(t, sizes, seed=10).cpu().numpy())


@pytest.mark.parametrize("M,N", [(1, 1), (700, 13), (70000, 3)])
def test_minmax_matches_oracle(dev, M, N):
    from recamd.pipeline import MinMaxScaler
    rng = np.random.default_rng(M)
    x = (rng.integers(-50, 100000, size=(M, N)) + rng.random((M, N)) * 0.9).astype(np.float32)
    if N > 1:
        x[:, -1] = 3.7                                      # constant column -> 0
    sc = MinMaxScaler()
    got = sc.fit_transform(torch.from_numpy(x).to(dev)).cpu().numpy()
    mn, mx = rp.minmax_fit(x)
    assert np.array_equal(sc.data_min_.cpu().numpy(), mn.astype(np.float32))
    assert np.array_equal(sc.data_max_.cpu().numpy(), mx.astype(np.float32))
    assert np.array_equal(got, rp.minmax_transform(x, mn, mx))      # fp64 arithmetic, rounded once: bit-exact


@pytest.mark.parametrize("maxlen", [1, 4, 10, 200])
@pytest.mark.parametrize("mode", [("pre", "pre"), ("post", "post"), ("pre", "post")])
def test_pad_sequences(dev, maxlen, mode):
    from recamd.pipeline import pad_sequences, ragged
    rng = np.random.default_rng(maxlen)
    seqs = [list(rng.integers(1, 1000, size=int(rng.integers(0, 30)))) for _ in range(257)]
    seqs[3] = []
    v, o = ragged(seqs)
    got = pad_sequences(torch.from_numpy(v).to(dev), torch.from_numpy(o).to(dev), maxlen, mode[0], mode[1]).cpu().numpy()
    assert np.array_equal(got, rp.pad_sequences(seqs, maxlen, mode[0], mode[1]))


def test_batch_feeder_pipeline(dev):
    """raw host columns -> pinned staging -> H2D + encode + scale on the copy stream -> the fused kernel: every batch
    equals the one-shot transform of the same rows, also for the ragged last batch."""
    from recamd import ops
    from recamd.pipeline import BatchFeeder, LabelEncoder, MinMaxScaler
    rng = np.random.default_rng(8)
    n, nd, F, D, bs = 1000, 13, 26, 128, 192
    dense = (rng.integers(0, 500, size=(n, nd)) + rng.random((n, nd))).astype(np.float32)
    tok = _tokens(rng, n, F, 60)
    enc = LabelEncoder(dev).fit(tok)
    sc = MinMaxScaler().fit(torch.from_numpy(dense).to(dev))
    tables = [torch.from_numpy(rng.normal(size=(len(v), D)).astype(np.float32)).to(dev) for v in enc.vocab_host]
    g = ops.TableGroup(tables)
    mn, mx = rp.minmax_fit(dense)
    e_dense = rp.minmax_transform(dense, mn, mx)
    e_ids = np.stack([rp.label_encode_transform(enc.vocab_host[f], tok[:, f]) for f in range(F)], axis=1)
    seen = 0
    for d_dense, d_ids in BatchFeeder(dense, tok, bs, encoder=enc, scaler=sc, device=dev):
        b = d_ids.shape[0]
        assert np.array_equal(d_ids.cpu().numpy(), e_ids[seen:seen + b])
        assert np.array_equal(d_dense.cpu().numpy(), e_dense[seen:seen + b])
        emb = ops.gather_concat(g, d_ids)                              # the path consumes the feeder's output
        assert emb.shape == (b, F * D)
        seen += b
    assert seen == n


def test_create_criteo_dataset_mirror(dev):
    """the reference's loader surface on a synthetic Criteo-shaped frame: ids == sklearn LabelEncoder per column,
    dense == MinMaxScaler on astype(int) per column, feat_num == number of distinct strings."""
    import pandas as pd
    from sklearn.preprocessing import LabelEncoder, MinMaxScaler
    from ctr.utils.data_process import create_criteo_dataset
    rng = np.random.default_rng(12)
    n = 600
    df = pd.DataFrame({"label": rng.integers(0, 2, size=n)})
    for i in range(1, 14):
        col = rng.integers(0, 3000, size=n).astype(np.float64)
        col[rng.random(n) < 0.1] = np.nan
        df[f"I{i}"] = col
    for i in range(1, 27):
        vocab = [f"{int(v):08x}" for v in rng.integers(0, 2 ** 32 - 2, size=20 + i, dtype=np.uint64)]
        col = np.array(vocab, dtype=object)[rng.integers(0, len(vocab), size=n)]
        col[rng.random(n) < 0.1] = np.nan
        df[f"C{i}"] = col
    fc, (trX, trY), (teX, teY) = create_criteo_dataset(df, embed_dim=8, read_part=False, test_size=0.2, seed=1, device=dev)
    perm = np.random.default_rng(1).permutation(n)
    order = np.concatenate([perm[int(np.ceil(n * 0.2)):], perm[:int(np.ceil(n * 0.2))]])
    ids = np.concatenate([trX[1], teX[1]])
    dense = np.concatenate([trX[0], teX[0]])
    sparse_features = ['C' + str(i) for i in range(1, 27)]
    dense_features = ['I' + str(i) for i in range(1, 14)]
    ref_df = df.copy()
    ref_df[sparse_features] = ref_df[sparse_features].fillna('-1')
    ref_df[dense_features] = ref_df[dense_features].fillna(0)
    for j, feat in enumerate(sparse_features):
        exp = LabelEncoder().fit_transform(ref_df[feat].astype(str))
        assert np.array_equal(ids[:, j], exp[order].astype(np.int32))
        assert fc[1][j] == {'feat': feat, 'feat_num': len(ref_df[feat].unique()), 'embed_dim': 8}
    exp_dense = MinMaxScaler().fit_transform(ref_df[dense_features].astype(int)).astype(np.float32)
    assert np.array_equal(dense, exp_dense[order])
    assert np.array_equal(np.concatenate([trY, teY]), df['label'].to_numpy()[order].astype(np.int32))
    assert trX[1].dtype == np.int32 and trX[0].dtype == np.float32 and len(teY) == 120


def test_match_pad_sequences_mirror(dev):
    from match.utils.data_process import pad_sequences
    seqs = [[5, 6, 7], [1], [], list(range(1, 30))]
    assert np.array_equal(pad_sequences(seqs, 10, device=dev), rp.pad_sequences(seqs, 10))
