"""Model-level parity: the mirrored ctr/match classes (HIP path, through the C ABI) vs the fp64
numpy oracle fed with the SAME explicit weights.  Tolerance 1e-5 * max(1, |b|) on fp32 logits."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def sparse_cols(vocabs, D):
    return [{'feat': f'C{i}', 'feat_num': v, 'embed_dim': D} for i, v in enumerate(vocabs)]


def dense_cols(n):
    return [{'feat': f'I{i}'} for i in range(n)]


def randomize(model, rng, scale=0.3):
    """Replace every weight by a seeded random value (BN variances stay positive) so that parity
    does not depend on initialisers."""
    w = model.get_weights()
    new = {}
    for k, v in w.items():
        if k.endswith('moving_variance'):
            new[k] = rng.uniform(0.5, 1.5, size=v.shape).astype(np.float32)
        elif k.endswith('embeddings'):
            new[k] = rng.uniform(-0.5, 0.5, size=v.shape).astype(np.float32)
        else:
            new[k] = (rng.normal(size=v.shape) * scale).astype(np.float32)
    model.set_weights(new)
    return model.get_weights()


def dnn_params(w, prefix, n):
    layers = [(w[f'{prefix}/dense_{i}/kernel'], w[f'{prefix}/dense_{i}/bias']) for i in range(n)]
    bn = dict(gamma=w[f'{prefix}/bn/gamma'], beta=w[f'{prefix}/bn/beta'], mean=w[f'{prefix}/bn/moving_mean'],
              var=w[f'{prefix}/bn/moving_variance'])
    return dict(layers=layers, bn=bn)


def inputs(rng, B, vocabs, nd):
    dense = rng.random((B, nd)).astype(np.float32)
    ids = np.stack([rng.integers(0, v, size=B) for v in vocabs], axis=1).astype(np.int32)
    return dense, ids


def test_fm_model_config1(dev):
    """BASELINE configs[0]: FM on Criteo-sample-shaped inputs, batch 256, k = 10."""
    from ctr.fm.model import FM
    rng = np.random.default_rng(2020)
    vocabs = [int(v) for v in rng.integers(3, 2000, size=26)]
    m = FM([dense_cols(13), sparse_cols(vocabs, 8)], k=10)
    w = randomize(m, rng, 0.05)
    dense, ids = inputs(rng, 256, vocabs, 13)
    out = m([dense, ids]).cpu().numpy()
    assert close(out, ref.fm_model_onehot(dense, ids, vocabs, w['w0'], w['w'], w['V']))


@pytest.mark.parametrize("D,B", [(8, 300), (5, 64), (16, 2048)])      # 2048 rows: the DNN takes the row maxima from the gather
def test_deepfm(dev, D, B):
    from ctr.deep_fm.model import DeepFM
    rng = np.random.default_rng(D)
    vocabs = [int(v) for v in rng.integers(3, 200, size=26)]
    m = DeepFM([dense_cols(13), sparse_cols(vocabs, D)], hidden_units=(32, 16, 8))
    dense, ids = inputs(rng, B, vocabs, 13)
    m([dense, ids])  # lazy build
    w = randomize(m, rng, 0.1)
    out = m([dense, ids]).cpu().numpy()
    tables = [w[f'embed_{i}/embeddings'] for i in range(26)]
    exp = ref.deepfm_forward(dense, ids, tables, w['fm/w'], dnn_params(w, 'dnn', 3), (w['dense/kernel'], w['dense/bias']))
    assert close(out, exp)


@pytest.mark.parametrize("B,D", [(200, 8), (2048, 16)])      # 2048 rows: the DNN takes the row maxima from the gather
def test_dcn(dev, B, D):
    from ctr.dcn.model import DCN
    rng = np.random.default_rng(4)
    vocabs = [int(v) for v in rng.integers(3, 200, size=26)]
    m = DCN(sparse_cols(vocabs, D), hidden_units=[32, 16, 8])
    _, ids = inputs(rng, B, vocabs, 0)
    m(ids)
    w = randomize(m, rng, 0.05)
    out = m(ids).cpu().numpy()
    tables = [w[f'embed_{i}/embeddings'] for i in range(26)]
    exp = ref.dcn_forward(ids, tables, w['cross_network/cross_weights'], w['cross_network/cross_bias'],
                          dnn_params(w, 'dnn_network', 3), (w['dense_final/kernel'], w['dense_final/bias']))
    assert close(out, exp)


@pytest.mark.parametrize("interaction,D,bot", [('cat', 8, [16, 8, 4]), ('dot', 128, [32, 128]), ('dot', 16, [8, 16])])
def test_dlrm(dev, interaction, D, bot):
    from ctr.dlrm.model import DLRM
    rng = np.random.default_rng(D)
    F = 26 if D != 16 else 4
    vocabs = [int(v) for v in rng.integers(3, 300, size=F)]
    m = DLRM([dense_cols(13), sparse_cols(vocabs, D)], bot_dnn_hidden_units=bot, top_dnn_hidden_units=[32, 16],
             interaction=interaction)
    dense, ids = inputs(rng, 130, vocabs, 13)
    m([dense, ids])
    w = randomize(m, rng, 0.1)
    out = m([dense, ids]).cpu().numpy()
    tables = [w[f'embed_{i}/embeddings'] for i in range(F)]
    exp = ref.dlrm_forward(dense, ids, tables, dnn_params(w, 'bot_dnn', len(bot)), dnn_params(w, 'top_dnn', 2),
                           (w['final_dense/kernel'], w['final_dense/bias']), interaction)
    assert close(out, exp)


def test_autoint_intended_config3_shape(dev):
    """BASELINE configs[2] shape: 39 fields (26 sparse + 13 dense) x dim 16, 3 layers, 2 heads, S=16."""
    from ctr.autoint.model import AutoInt
    rng = np.random.default_rng(3)
    vocabs = [int(v) for v in rng.integers(3, 500, size=26)]
    m = AutoInt([dense_cols(13), sparse_cols(vocabs, 16)], att_hidden_units=16, head_num=2, att_layer_num=3,
                use_res=True)
    dense, ids = inputs(rng, 96, vocabs, 13)
    m([dense, ids])
    w = randomize(m, rng, 0.2)
    out = m([dense, ids]).cpu().numpy()
    tables = [w[f'embed_{i}/embeddings'] for i in range(26)]
    emb = ref.gather_concat([t.astype(np.float64) for t in tables], ids).reshape(96, 26, 16)
    x3 = np.concatenate([emb, dense[:, :, None].astype(np.float64) * w['dense_embed'][None].astype(np.float64)], axis=1)
    layers = [dict(Wq=w[f'attention_{i}/Wq'], Wk=w[f'attention_{i}/Wk'], Wv=w[f'attention_{i}/Wv'],
                   W0=w[f'attention_{i}/W0']) for i in range(3)]
    exp = ref.autoint_forward_intended(x3, layers, (w['final_dense/kernel'], w['final_dense/bias']), 2, 16, 'relu', True)
    assert close(out, exp)


@pytest.mark.parametrize("B,nd,F,heads,L,res", [(96, 13, 26, 2, 3, True), (5, 1, 7, 1, 1, False), (130, 3, 40, 2, 2, True),
                                                (33, 64 - 17, 17, 1, 4, False)])
def test_autoint_one_launch_forward(dev, monkeypatch, B, nd, F, heads, L, res):
    """rec_autoint_forward_f32 (lookup + dense-field embedding + interacting layers + Dense(1) + sigmoid in one launch)
    against the oracle, and the mirror's two paths against each other; out-of-range ids read as zero rows + flag."""
    from ctr.autoint.model import AutoInt
    from recamd import ops
    rng = np.random.default_rng(B + nd)
    vocabs = [int(v) for v in rng.integers(3, 300, size=F)]
    cols = [dense_cols(nd), sparse_cols(vocabs, 16)]
    m = AutoInt(cols, att_hidden_units=16, head_num=heads, att_layer_num=L, use_res=res)
    dense, ids = inputs(rng, B, vocabs, nd)
    m([dense, ids])
    w = randomize(m, rng, 0.2)
    out = m([dense, ids]).cpu().numpy()
    m.fused = False
    out_layers = m([dense, ids]).cpu().numpy()
    m.fused = True
    tables = [w[f'embed_{i}/embeddings'] for i in range(F)]
    emb = ref.gather_concat([t.astype(np.float64) for t in tables], ids).reshape(B, F, 16)
    x3 = emb if nd == 0 else np.concatenate(
        [emb, dense[:, :, None].astype(np.float64) * w['dense_embed'][None].astype(np.float64)], axis=1)
    layers = [dict(Wq=w[f'attention_{i}/Wq'], Wk=w[f'attention_{i}/Wk'], Wv=w[f'attention_{i}/Wv'],
                   W0=w.get(f'attention_{i}/W0')) for i in range(L)]
    exp = ref.autoint_forward_intended(x3, layers, (w['final_dense/kernel'], w['final_dense/bias']), heads, 16, 'relu', res)
    assert close(out, exp) and close(out_layers, exp)
    # the op itself, with an out-of-range id
    ids2 = ids.copy()
    ids2[B // 2, F - 1] = vocabs[F - 1] + 3
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    t_ids, t_dense = torch.from_numpy(ids2.astype(np.int32)).to(dev), torch.from_numpy(dense.astype(np.float32)).to(dev)
    got = ops.autoint_forward(m._group, t_ids, t_dense, m._w['dense_embed'],
                              [(A._w['Wq'], A._w['Wk'], A._w['Wv'], A._w.get('W0')) for A in m.attention_layers], heads, 16,
                              'relu', m.final_dense._w['kernel'], m.final_dense._w.get('bias'), oob_flag=flag)
    assert got is not None and int(flag.item()) == 1
    emb2 = ref.gather_concat([t.astype(np.float64) for t in tables], ids2).reshape(B, F, 16)
    x3b = emb2 if nd == 0 else np.concatenate([emb2, x3[:, F:]], axis=1)
    exp2 = ref.autoint_forward_intended(x3b, layers, (w['final_dense/kernel'], w['final_dense/bias']), heads, 16, 'relu', res)
    assert close(got.cpu().numpy(), exp2)


def test_autoint_as_written_sample_mixing(dev):
    """The reference's 2-D call (autoint/model.py:48-51): output batch is B / att_hidden_units."""
    from ctr.autoint.model import AutoInt
    rng = np.random.default_rng(5)
    vocabs = [int(v) for v in rng.integers(3, 100, size=6)]
    S = 8
    m = AutoInt([dense_cols(3), sparse_cols(vocabs, 4)], att_hidden_units=S, mode='as_written')
    dense, ids = inputs(rng, 64, vocabs, 3)
    m([dense, ids])
    w = randomize(m, rng, 0.3)
    out = m([dense, ids]).cpu().numpy()
    assert out.shape == (64 // S, 1)
    tables = [w[f'embed_{i}/embeddings'] for i in range(6)]
    L = dict(Wq=w['attention_0/Wq'], Wk=w['attention_0/Wk'], Wv=w['attention_0/Wv'])
    exp = ref.autoint_forward_as_written(dense, ids, tables, L, (w['final_dense/kernel'], w['final_dense/bias']), S)
    assert close(out, exp)
    with pytest.raises(ValueError):
        m([dense[:60], ids[:60]])


def _din_setup(maxlen, rng, d=8, n_user=3, n_item=3, vocab=20):
    sparse_feature_dict, ui, ii, bi = {}, {}, {}, {}
    for i in range(n_user):
        sparse_feature_dict[f'user_sparse_{i}'] = (vocab, d)
        ui[f'user_sparse_{i}'] = i
    for i in range(n_item):
        sparse_feature_dict[f'item_sparse_{i}'] = (vocab, d)
        ii[f'item_sparse_{i}'] = i
    idx = 0
    for ml in range(maxlen):
        for i in range(n_item):
            bi[f'item_sparse_{ml}_{i}'] = idx
            idx += 1
    return sparse_feature_dict, [ui, ii, bi]


@pytest.mark.parametrize("ffn_act", ['prelu', 'dice'])
def test_din_intended(dev, ffn_act):
    """Canonical DIN (config 4 semantics at small size): AttentionLayer pooling, sigmoid scores,
    pre-padded histories with id 0."""
    from ctr.din.model import DIN
    rng = np.random.default_rng(11)
    maxlen, d, B, vocab = 12, 8, 50, 20
    sfd, sfi = _din_setup(maxlen, rng, d, vocab=vocab)
    m = DIN(sfd, sfi, att_hidden_units=64, ffn_hidden_units=(16, 8), att_activation='sigmoid', ffn_activation=ffn_act,
            maxlen=maxlen)
    ud = rng.random((B, 5)).astype(np.float32)
    us = rng.integers(0, vocab, size=(B, 3)).astype(np.float32)
    idn = rng.random((B, 5)).astype(np.float32)
    its = rng.integers(0, vocab, size=(B, 3)).astype(np.float32)
    lens = rng.integers(0, maxlen + 1, size=B)
    beh = rng.integers(1, vocab, size=(B, maxlen, 3))
    beh[np.arange(maxlen)[None, :] < (maxlen - lens)[:, None]] = 0  # pre-padding
    beh = beh.reshape(B, maxlen * 3).astype(np.float32) + 0.25       # float ids, truncated by the cast
    beh[beh < 1] = 0
    m([ud, us, idn, its, beh])
    w = randomize(m, rng, 0.2)
    out = m([ud, us, idn, its, beh]).cpu().numpy()
    # oracle composition
    f64 = lambda a: np.asarray(a, np.float64)  # noqa: E731
    T = {k: f64(w[f'embed_{k}/embeddings']) for k in sfd}
    user_emb = np.concatenate([ref.embedding_lookup(T[f'user_sparse_{i}'], us[:, i]) for i in range(3)], axis=-1)
    user_embed = np.concatenate([f64(ud), user_emb], axis=-1)
    item_emb = np.concatenate([ref.embedding_lookup(T[f'item_sparse_{i}'], its[:, i]) for i in range(3)], axis=-1)
    item_embed = np.concatenate([f64(its), item_emb], axis=-1)
    behf = beh.reshape(B, maxlen, 3)
    beh_emb = np.concatenate([ref.embedding_lookup(T[f'item_sparse_{i}'], behf[:, :, i]) for i in range(3)], axis=-1)
    mask = (ref.cast_ids(behf[:, :, 0]) != 0).astype(np.float64)
    pooled = ref.din_attention_layer(item_emb, beh_emb, beh_emb, mask, w['attention_layer/kernel'],
                                     w['attention_layer/bias'], 'sigmoid')
    x = np.concatenate([user_embed, item_embed, pooled], axis=-1)
    x = ref.batch_norm_inference(x, f64(w['bn/gamma']), f64(w['bn/beta']), f64(w['bn/moving_mean']), f64(w['bn/moving_variance']))
    for i in range(2):
        x = ref.dense(x, f64(w[f'ffn_{i}/kernel']), f64(w[f'ffn_{i}/bias']))
        if ffn_act == 'prelu':
            x = ref.activation(x, 'prelu', f64(w[f'ffn_{i}/prelu/alpha']))
        else:
            x = ref.dice(x, float(w[f'ffn_{i}/dice/alpha']), f64(w[f'ffn_{i}/dice/bn/moving_mean']),
                         f64(w[f'ffn_{i}/dice/bn/moving_variance']))
    exp = ref.sigmoid(ref.dense(x, f64(w['final_output/kernel']), f64(w['final_output/bias'])))
    assert close(out, exp)
    # the unfused path (materialised history + AttentionLayer) gives the same result
    m.fuse_history = False
    assert close(m([ud, us, idn, its, beh]).cpu().numpy(), exp)


def test_din_as_written(dev):
    """maxlen > 1: the reference's concat (din/model.py:79-81) has mismatched batch sizes -> raises.
    maxlen == 1 runs: self-attention over one position."""
    from ctr.din.model import DIN
    rng = np.random.default_rng(12)
    sfd, sfi = _din_setup(3, rng)
    m = DIN(sfd, sfi, att_hidden_units=16, ffn_hidden_units=(8,), att_activation='relu', maxlen=3, mode='as_written')
    B = 10
    args = [rng.random((B, 5)).astype(np.float32), rng.integers(0, 20, size=(B, 3)).astype(np.float32),
            rng.random((B, 5)).astype(np.float32), rng.integers(0, 20, size=(B, 3)).astype(np.float32),
            rng.integers(0, 20, size=(B, 9)).astype(np.float32)]
    with pytest.raises(ValueError):
        m(args)
    sfd, sfi = _din_setup(1, rng)
    m1 = DIN(sfd, sfi, att_hidden_units=16, ffn_hidden_units=(8,), att_activation='relu', maxlen=1, mode='as_written')
    args[4] = rng.integers(0, 20, size=(B, 3)).astype(np.float32)
    out = m1(args)
    assert out.shape == (B, 1) and torch.isfinite(out).all()


def _sasrec_params(w, i):
    p = f'encoder_{i}'
    return dict(Wq=w[f'{p}/mha/wq/kernel'], bq=w[f'{p}/mha/wq/bias'], Wk=w[f'{p}/mha/wk/kernel'], bk=w[f'{p}/mha/wk/bias'],
                Wv=w[f'{p}/mha/wv/kernel'], bv=w[f'{p}/mha/wv/bias'],
                W1=w[f'{p}/ffn/conv1/kernel'], b1=w[f'{p}/ffn/conv1/bias'], W2=w[f'{p}/ffn/conv2/kernel'], b2=w[f'{p}/ffn/conv2/bias'],
                ln1_g=w[f'{p}/layernorm1/gamma'], ln1_b=w[f'{p}/layernorm1/beta'],
                ln2_g=w[f'{p}/layernorm2/gamma'], ln2_b=w[f'{p}/layernorm2/beta'])


@pytest.mark.parametrize("blocks,heads,last_row_only", [(1, 1, True), (2, 2, True), (2, 1, False)])
def test_sasrec(dev, blocks, heads, last_row_only):
    """The commented model_test spec of the reference (sasrec/model.py:121-127): seq_item len S dim 64,
    pos_item len 1, neg_item len 100, att_hidden_unit=64; pre-padded sequences."""
    from match.sasrec.model import SASRec
    rng = np.random.default_rng(blocks * 10 + heads)
    V, S, n, B = 100, 20, 100, 40
    user_features = [{'feat': 'user_id', 'feat_num': 100, 'feat_len': 1, 'embed_dim': 8},
                     {'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': 64},
                     {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': 64},
                     {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': 64}]
    item_features = [{'feat': 'item_id', 'feat_num': 100, 'feat_len': 1, 'embed_dim': 32}]
    m = SASRec(user_features, item_features, blocks=blocks, num_heads=heads, att_hidden_unit=64, seq_len=S, neg_len=n,
               last_row_only=last_row_only)
    lens = rng.integers(0, S + 1, size=B)
    lens[0] = 0  # an all-padding sequence
    seq = rng.integers(1, V, size=(B, S))
    seq[np.arange(S)[None, :] < (S - lens)[:, None]] = 0
    seq = seq.astype(np.int32)
    pos = rng.integers(1, V, size=(B, 1)).astype(np.int32)
    neg = rng.integers(1, V, size=(B, n)).astype(np.int32)
    m([seq, pos, neg])
    w = randomize(m, rng, 0.15)
    for i in range(blocks):  # keep LN gains near 1
        for ln in ('layernorm1', 'layernorm2'):
            m.set_weights({f'encoder_{i}/{ln}/gamma': 1 + 0.1 * rng.normal(size=64).astype(np.float32)})
    w = m.get_weights()
    logits = m([seq, pos, neg]).cpu().numpy()
    exp, loss = ref.sasrec_forward(seq, pos, neg, w['user_embed_seq_item/embeddings'], w['user_embed_pos_item/embeddings'],
                                   w['user_embed_neg_item/embeddings'], [_sasrec_params(w, i) for i in range(blocks)], heads)
    assert close(logits, exp, 1e-5)
    assert abs(float(m.losses[0]) - loss) <= 1e-5 * max(1.0, abs(loss))
    # KAT: an all-zero sequence gives logits exactly 0.0
    assert np.all(logits[0] == 0.0)


@pytest.mark.parametrize("G", [1, 8])
def test_sasrec_row_sharded_simulated_ranks(dev, G):
    """BASELINE configs[4] as stated: SASRec S=200, d=64 with the seq/pos/neg tables row-sharded (row % G) behind the
    sharded lookup, G ranks simulated on one GPU (tests/shard_oracle.py::PeersShardedTables: a rank's requests are served from the owner's
    shard, which is what the RCCL all-to-all pair delivers; the exchange itself is covered by tests/test_shard_gpu.py
    and tests/test_dist_cpu.py).  Every rank's logits equal the oracle's sasrec_forward on the UNSHARDED tables."""
    from match.sasrec.model import SASRec
    from recamd.dist import shard_table
    from tests.shard_oracle import PeersShardedTables
    rng = np.random.default_rng(50 + G)
    V, S, n, B, d = 5000, 200, 100, 24, 64
    uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d},
          {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
          {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
    full = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, seq_len=S, neg_len=n)

    def batch():
        lens = rng.integers(0, S + 1, size=B)
        seq = rng.integers(1, V, size=(B, S))
        seq[np.arange(S)[None, :] < (S - lens)[:, None]] = 0
        return [seq.astype(np.int32), rng.integers(0, V, size=(B, 1)).astype(np.int32),
                rng.integers(0, V, size=(B, n)).astype(np.int32)]
    full(batch())
    randomize(full, rng, 0.15)
    w = full.get_weights()
    tabs = {k: w[f'user_embed_{k}/embeddings'] for k in ('seq_item', 'pos_item', 'neg_item')}
    ranks = [SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, seq_len=S, neg_len=n, sharded=(r, G),
                    shard_factory=PeersShardedTables) for r in range(G)]
    if G > 1:
        for m in ranks:
            m._sharded.link_peers([x._sharded for x in ranks])
    for r, m in enumerate(ranks):
        m(batch())                                                    # builds the lazily created layers
        ws = {k: v for k, v in w.items() if not k.startswith('user_embed_')}
        for k, t in tabs.items():
            ws[f'user_embed_{k}/embeddings'] = shard_table(torch.from_numpy(t), r, G).numpy()
        m.set_weights(ws)
    for r in range(G):
        seq, pos, neg = batch()
        logits = ranks[r]([seq, pos, neg]).cpu().numpy()
        exp, loss = ref.sasrec_forward(seq, pos, neg, tabs['seq_item'], tabs['pos_item'], tabs['neg_item'],
                                       [_sasrec_params(w, 0)], 1)
        assert close(logits, exp, 1e-5)
        assert abs(float(ranks[r].losses[0]) - loss) <= 1e-5 * max(1.0, abs(loss))
        if G > 1:
            st = ranks[r]._sharded.describe()
            assert st["unique_sent"] < st["ids"]                      # pad ids dropped + duplicates merged


def test_youtube_dnn_towers(dev):
    from match.youtube_dnn.model import YoutubeDNN
    rng = np.random.default_rng(6)
    B = 77
    user_features = [{'feat': 'user_id', 'feat_num': 50, 'feat_len': 1, 'embed_dim': 16},
                     {'feat': 'age', 'feat_num': 7, 'feat_len': 1, 'embed_dim': 4},
                     {'feat': 'gender', 'feat_num': 2, 'feat_len': 1, 'embed_dim': 4}]
    item_features = [{'feat': 'movie_id', 'feat_num': 80, 'feat_len': 1, 'embed_dim': 16}] + \
                    [{'feat': f'g{i}', 'feat_num': 2, 'feat_len': 1, 'embed_dim': 4} for i in range(19)]
    m = YoutubeDNN(user_features, item_features)
    u = {f['feat']: rng.integers(0, f['feat_num'], size=(B, 1)).astype(np.float32) for f in user_features}
    it = {f['feat']: rng.integers(0, f['feat_num'], size=(B, 1)).astype(np.float32) for f in item_features}
    labels = rng.integers(0, 2, size=(B, 1)).astype(np.int32)
    m([u, it, labels])
    w = randomize(m, rng, 0.2)
    item_out, user_out = m([u, it, labels])
    ul = [(w[f'user_dnn/dense_{i}/kernel'], w[f'user_dnn/dense_{i}/bias']) for i in range(2)]
    il = [(w[f'item_dnn/dense_{i}/kernel'], w[f'item_dnn/dense_{i}/bias']) for i in range(2)]
    eu, ei = ref.youtube_dnn_towers([u[f['feat']] for f in user_features], [w[f"user_embed_{f['feat']}/embeddings"] for f in user_features],
                                    [it[f['feat']] for f in item_features], [w[f"item_embed_{f['feat']}/embeddings"] for f in item_features],
                                    ul, il)
    assert user_out.shape == (B, 1, 32) and item_out.shape == (B, 1, 32)
    assert close(user_out.cpu().numpy(), eu)
    assert close(item_out.cpu().numpy(), ei)
