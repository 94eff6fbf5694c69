"""§8f-4 widening: zoo models that reuse the embedding lookup (Deep&Crossing, Wide&Deep, ESMM) — mirrored classes on
the HIP path vs the fp64 numpy oracle fed with the same explicit weights.  Tolerance 1e-5 * max(1, |ref|)."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.test_models_gpu import dense_cols, dnn_params, inputs, randomize, sparse_cols
from tests.util import close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,act,alpha,beta", [(1, None, 1.0, 1.0), (1000, 'relu', 1.0, 1.0), (4096, 'sigmoid', 0.5, 0.5),
                                              (777, 'tanh', -2.0, 0.25)])
def test_axpby_act(dev, n, act, alpha, beta):
    from recamd import ops
    rng = np.random.default_rng(n)
    a, b = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    got = ops.axpby_act(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), alpha, beta, act).cpu().numpy()
    exp = ref.activation(alpha * a.astype(np.float64) + beta * b.astype(np.float64), act)
    assert close(got, exp)


@pytest.mark.parametrize("D,B,hidden", [(8, 300, (32, 16)), (5, 65, (7,))])
def test_deep_crossing(dev, D, B, hidden):
    from ctr.deep_crossing.model import Deep_Crossing
    rng = np.random.default_rng(D)
    vocabs = [int(v) for v in rng.integers(3, 200, size=26)]
    m = Deep_Crossing(sparse_cols(vocabs, D), hidden_units=hidden)
    _, ids = inputs(rng, B, vocabs, 1)
    m(ids)  # lazy build
    w = randomize(m, rng, 0.1)
    out = m(ids).cpu().numpy()
    tables = [w[f'embed_{i}/embeddings'] for i in range(26)]
    res = [(w[f'res_{i}/layer1/kernel'], w[f'res_{i}/layer1/bias'], w[f'res_{i}/layer2/kernel'], w[f'res_{i}/layer2/bias'])
           for i in range(len(hidden))]
    exp = ref.deep_crossing_forward(ids, tables, res, (w['dense/kernel'], w['dense/bias']))
    assert out.shape == (B, 1)
    assert close(out, exp)


@pytest.mark.parametrize("D,B", [(8, 257), (6, 64)])
def test_wide_deep(dev, D, B):
    from ctr.wide_deep.model import WideDeep
    rng = np.random.default_rng(10 + D)
    vocabs = [int(v) for v in rng.integers(3, 300, size=26)]
    m = WideDeep([dense_cols(13), sparse_cols(vocabs, D)], hidden_units=(64, 32, 8))
    dense, ids = inputs(rng, B, vocabs, 13)
    m([dense, ids])
    w = randomize(m, rng, 0.1)
    out = m([dense, ids]).cpu().numpy()
    tables = [w[f'embed_{i}/embeddings'] for i in range(26)]
    layers = [(w[f'dnn_network/dense_{i}/kernel'], w[f'dnn_network/dense_{i}/bias']) for i in range(3)]
    exp = ref.wide_deep_forward(dense, ids, tables, (w['linear/dense/kernel'], w['linear/dense/bias']), layers,
                                (w['final_dense/kernel'], w['final_dense/bias']))
    assert close(out, exp)


def test_esmm(dev):
    """Shapes of build_graph (src/ctr/esmm/model.py:93-101): 5 numerical + 5 categorical user inputs, 5 + 3 item."""
    from ctr.esmm.model import ESMM
    rng = np.random.default_rng(7)
    user_feats = {f'u{i}': (int(rng.integers(5, 60)), 4 + i) for i in range(5)}
    item_feats = {f'i{i}': (int(rng.integers(5, 60)), 6) for i in range(3)}
    cols = {**user_feats, **item_feats}
    user_dict = {k: (i,) for i, k in enumerate(user_feats)}
    item_dict = {k: (i,) for i, k in enumerate(item_feats)}
    m = ESMM(cols, [user_dict, item_dict], hidden_units=[32, 16])
    B = 200

    def tower_inputs():
        un = rng.random((B, 5)).astype(np.float32)
        uc = np.stack([rng.integers(0, user_feats[k][0], size=B) for k in user_feats], axis=1).astype(np.float32)
        inum = rng.random((B, 5)).astype(np.float32)
        ic = np.stack([rng.integers(0, item_feats[k][0], size=B) for k in item_feats], axis=1).astype(np.float32)
        return [un, uc, inum, ic]

    x = tower_inputs() + tower_inputs()
    m(x)
    w = randomize(m, rng, 0.1)
    ctr, ctcvr = (t.cpu().numpy() for t in m(x))
    ut = [w[f'embed_{k}/embeddings'] for k in user_feats]
    it = [w[f'embed_{k}/embeddings'] for k in item_feats]

    def head(p):
        return dict(bn=dict(gamma=w[f'{p}/bn/gamma'], beta=w[f'{p}/bn/beta'], mean=w[f'{p}/bn/moving_mean'],
                            var=w[f'{p}/bn/moving_variance']),
                    dense=(w[f'{p}/dense/kernel'], w[f'{p}/dense/bias']), out=(w[f'{p}/out/kernel'], w[f'{p}/out/bias']))

    args = (ut, list(range(5)), it, list(range(3)), dnn_params(w, 'user_dnn', 2), dnn_params(w, 'item_dnn', 2))
    e_ctr = ref.esmm_tower(*x[:4], *args, head('ctr_head'))
    e_cvr = ref.esmm_tower(*x[4:], *args, head('cvr_head'))
    assert ctr.shape == (B, 1) and ctcvr.shape == (B, 1)
    assert close(ctr, e_ctr)
    assert close(ctcvr, e_ctr * e_cvr)


@pytest.mark.parametrize("n", [1, 100, 70_000])
def test_mul_act_and_cosine_flat(dev, n):
    from recamd import ops
    rng = np.random.default_rng(n)
    a, b = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    assert close(ops.mul_act(ta, tb, 'sigmoid').cpu().numpy(), ref.sigmoid(a.astype(np.float64) * b))
    assert close(ops.mul_act(ta, tb).cpu().numpy(), a.astype(np.float64) * b)
    assert close(ops.cosine_flat(ta, tb).cpu().numpy(), np.array([ref.cosine_flat(a, b)]))
    assert close(ops.cosine_flat(ta, tb, sigmoid=True).cpu().numpy(), ref.sigmoid(np.array([ref.cosine_flat(a, b)])))


def match_cols(names, vocabs, D):
    return [{'feat': n, 'feat_num': v, 'feat_len': 1, 'embed_dim': D} for n, v in zip(names, vocabs)]


def test_dssm_towers_and_output(dev):
    from match.dssm.model import Dssm
    from recamd.retrieval import IndexFlatIP
    rng = np.random.default_rng(21)
    ucols = match_cols(['user_id', 'gender', 'age'], [300, 3, 8], 8)
    icols = match_cols(['movie_id', 'genre'], [500, 20], 8)
    m = Dssm(ucols, icols)
    B = 150
    uin = {c['feat']: rng.integers(0, c['feat_num'], size=(B, 1)).astype(np.float32) for c in ucols}
    iin = {c['feat']: rng.integers(0, c['feat_num'], size=(B, 1)).astype(np.float32) for c in icols}
    m([uin, iin])
    w = randomize(m, rng, 0.2)
    out = m([uin, iin]).cpu().numpy()
    uids = np.concatenate([uin[c['feat']] for c in ucols], axis=1)
    iids = np.concatenate([iin[c['feat']] for c in icols], axis=1)
    ut = [w[f"user_embed_{c['feat']}/embeddings"] for c in ucols]
    it = [w[f"item_embed_{c['feat']}/embeddings"] for c in icols]
    ud = [(w[f'user_dnn/dense_{i}/kernel'], w[f'user_dnn/dense_{i}/bias']) for i in range(2)]
    idn = [(w[f'item_dnn/dense_{i}/kernel'], w[f'item_dnn/dense_{i}/bias']) for i in range(2)]
    e_out, e_u, e_i = ref.dssm_forward(uids, iids, ut, it, ud, idn)
    assert out.shape == (1, 1)                                       # the whole-batch cosine of the reference
    assert close(out, e_out)
    assert close(m.user_dnn_out.cpu().numpy(), e_u) and close(m.item_dnn_out.cpu().numpy(), e_i)
    # the retrieval step of dssm_train.py:63-78 on the towers
    index = IndexFlatIP(e_i.shape[-1])
    index.add(m.item_dnn_out[:, 0, :])
    D, I = index.search(m.user_dnn_out[:, 0, :], 10)
    eD, _ = ref.topk_inner_product(e_u[:, 0, :], e_i[:, 0, :], 10)
    assert np.all(np.abs(D - eD) <= 1e-5 * np.maximum(1.0, np.abs(eD)))


@pytest.mark.parametrize("neg", [1, 10])
def test_ncf(dev, neg):
    from match.ncf.model import NCF
    rng = np.random.default_rng(30 + neg)
    m = NCF({'feat': 'user_id', 'feat_num': 100, 'embed_dim': 8}, {'feat': 'item_id', 'feat_num': 120, 'embed_dim': 8},
            neg_num=neg)
    B = 77
    user = rng.integers(0, 100, size=(B, 1)).astype(np.int32)
    pos = rng.integers(0, 120, size=(B, 1)).astype(np.int32)
    negs = rng.integers(0, 120, size=(B, neg)).astype(np.int32)
    m([user, pos, negs])
    w = randomize(m, rng, 0.2)
    out = m([user, pos, negs]).cpu().numpy()
    layers = [(w[f'dnn/dense_{i}/kernel'], w[f'dnn/dense_{i}/bias']) for i in range(3)]
    exp = ref.ncf_forward(user, pos, negs, w['user_embedding/embeddings'], w['item_embedding/embeddings'],
                          w['neg_item_embedding/embeddings'], layers, (w['dense/kernel'], w['dense/bias']))
    assert out.shape == (B, 1 + neg)
    assert close(out, exp)
    loss = ref.pairwise_rank_loss(exp)                                  # the add_loss of src/match/ncf/model.py:75-77
    assert abs(float(m.losses[0]) - loss) <= 1e-5 * max(1.0, abs(loss))


@pytest.mark.parametrize("D", [8, 6])
def test_match_fm(dev, D):
    from match.fm.model import FM
    rng = np.random.default_rng(40 + D)
    ucols = match_cols(['user_id', 'gender', 'age'], [300, 3, 8], D)
    icols = match_cols(['movie_id', 'genre'], [500, 20], D)
    m = FM(ucols, icols, k=16)
    B = 203
    uin = {c['feat']: rng.integers(0, c['feat_num'], size=(B, 1)).astype(np.float32) for c in ucols}
    iin = {c['feat']: rng.integers(0, c['feat_num'], size=(B, 1)).astype(np.float32) for c in icols}
    w = randomize(m, rng, 0.2)
    out = m([uin, iin]).cpu().numpy()
    uids = np.concatenate([uin[c['feat']] for c in ucols], axis=1)
    iids = np.concatenate([iin[c['feat']] for c in icols], axis=1)
    ut = [w[f"user_embed_{c['feat']}/embeddings"] for c in ucols]
    it = [w[f"item_embed_{c['feat']}/embeddings"] for c in icols]
    e_out, e_u, e_i = ref.match_fm_forward(uids, iids, ut, it, w['w0'], w['w'], w['V'])
    assert out.shape == (B, 1)
    assert close(out, e_out)
    assert np.array_equal(m.user_embeds.cpu().numpy(), e_u.astype(np.float32))
    assert np.array_equal(m.item_embeds.cpu().numpy(), e_i.astype(np.float32))
