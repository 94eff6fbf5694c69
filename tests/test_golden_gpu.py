"""HIP kernels vs the committed golden vectors (tests/golden/*.npz, oracle fp64 outputs)."""
import os

import numpy as np
import pytest
import torch

from tests.util import close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_golden_gather(dev):
    from recamd import ops
    z = load("gather")
    g = ops.TableGroup([T(z[f"table_{i}"], dev) for i in range(5)])
    out = ops.gather_concat(g, T(z["ids"], dev)).cpu().numpy()
    assert np.array_equal(out, z["expected"].astype(np.float32))  # bit-exact copy


def test_golden_fm_model(dev):
    from recamd import ops
    z = load("fm_model")
    out = ops.fm_onehot(T(z["dense"], dev), T(z["ids"], dev), [int(v) for v in z["vocab"]], T(z["w0"], dev), T(z["w"], dev),
                        T(z["V"], dev)).cpu().numpy()
    assert close(out, z["expected"])


def test_golden_fm_layer(dev):
    from recamd import ops
    z = load("fm_layer")
    assert close(ops.fm_layer(T(z["first"], dev), T(z["second"], dev), T(z["w"], dev)).cpu().numpy(), z["expected"])


def test_golden_cross(dev):
    from recamd import ops
    z = load("cross")
    assert close(ops.cross_network(T(z["x"], dev), T(z["W"], dev), T(z["Bv"], dev)).cpu().numpy(), z["expected"])


def test_golden_pairwise_dot(dev):
    from recamd import ops
    z = load("pairwise_dot")
    assert close(ops.pairwise_dot(T(z["x"], dev)).cpu().numpy(), z["expected"])


def test_golden_dlrm_fused_both_kernels(dev):
    """int32 ids take the LDS-ring / fp32-MFMA kernel (pairwise_dot_ring.hip), float ids (the Keras
    Embedding cast) the register-tiled kernel (pairwise_dot.hip): both against the same golden vectors."""
    from recamd import ops
    z = load("dlrm_dot")
    g = ops.TableGroup([T(z[f"table_{i}"], dev) for i in range(26)])
    for ids in (T(z["ids"], dev), T(z["ids"].astype(np.float32), dev)):
        out = ops.gather_pairwise_dot(g, ids, T(z["dense"], dev)).cpu().numpy()
        assert close(out[:, :351], z["expected"])
        assert np.array_equal(out[:, 351:], z["dense"])


def test_golden_mha_ctr(dev):
    from recamd import ops
    z = load("mha_ctr")
    x = T(z["x"], dev)
    out = ops.mha_ctr(x, x, x, T(z["Wq"], dev), T(z["Wk"], dev), T(z["Wv"], dev), T(z["W0"], dev), int(z["H"]), int(z["S"]),
                      "relu").cpu().numpy()
    assert close(out, z["expected"])


def test_golden_din_attention(dev):
    from recamd import ops
    z = load("din_attention")
    k = T(z["k"], dev)
    out = ops.din_attention_pool(T(z["q"], dev), k, k, T(z["mask"], dev), T(z["W"], dev), T(z["b"], dev), "sigmoid").cpu().numpy()
    assert close(out, z["expected"])


def test_golden_sasrec(dev):
    from match.sasrec.model import SASRec
    z = load("sasrec")
    V, dm = z["T_seq"].shape
    S, n = z["seq"].shape[1], z["neg"].shape[1]
    uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': dm},
          {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': dm},
          {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': dm}]
    m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=dm, ffn_hidden_unit=128, seq_len=S, neg_len=n)
    m([z["seq"], z["pos"], z["neg"]])
    p = 'encoder_0'
    m.set_weights({
        'user_embed_seq_item/embeddings': z["T_seq"], 'user_embed_pos_item/embeddings': z["T_pos"],
        'user_embed_neg_item/embeddings': z["T_neg"],
        f'{p}/mha/wq/kernel': z["b0_Wq"], f'{p}/mha/wq/bias': z["b0_bq"], f'{p}/mha/wk/kernel': z["b0_Wk"],
        f'{p}/mha/wk/bias': z["b0_bk"], f'{p}/mha/wv/kernel': z["b0_Wv"], f'{p}/mha/wv/bias': z["b0_bv"],
        f'{p}/ffn/conv1/kernel': z["b0_W1"], f'{p}/ffn/conv1/bias': z["b0_b1"], f'{p}/ffn/conv2/kernel': z["b0_W2"],
        f'{p}/ffn/conv2/bias': z["b0_b2"], f'{p}/layernorm1/gamma': z["b0_ln1_g"], f'{p}/layernorm1/beta': z["b0_ln1_b"],
        f'{p}/layernorm2/gamma': z["b0_ln2_g"], f'{p}/layernorm2/beta': z["b0_ln2_b"]})
    out = m([z["seq"], z["pos"], z["neg"]]).cpu().numpy()
    assert close(out, z["expected"], 1e-5)


def test_golden_dcn(dev):
    """whole DCN forward (fused gather + dots, closed-form cross / Dense(1) logit) against the fixture produced by the
    oracle's literal CrossNetwork recurrence + concat + Dense(1); layer_num = len(hidden_units) = 2 (dcn/model.py:32)"""
    from ctr.dcn.model import DCN
    z = load("dcn")
    F = int(z["F"])
    V, D = z["table_0"].shape
    m = DCN([{'feat': f'C{i}', 'feat_num': V, 'embed_dim': D} for i in range(F)], hidden_units=[16, 8])
    m(z["ids"])
    w = {f'embed_{i}/embeddings': z[f"table_{i}"] for i in range(F)}
    w.update({'cross_network/cross_weights': z["cW"], 'cross_network/cross_bias': z["cB"],
              'dnn_network/dense_0/kernel': z["W0"], 'dnn_network/dense_0/bias': z["b0"],
              'dnn_network/dense_1/kernel': z["W1"], 'dnn_network/dense_1/bias': z["b1"],
              'dnn_network/bn/gamma': z["bn_g"], 'dnn_network/bn/beta': z["bn_b"], 'dnn_network/bn/moving_mean': z["bn_m"],
              'dnn_network/bn/moving_variance': z["bn_v"], 'dense_final/kernel': z["Wf"], 'dense_final/bias': z["bf"]})
    m.set_weights(w)
    assert close(m(z["ids"]).cpu().numpy(), z["expected"])


def test_golden_topk(dev):
    from recamd import ops
    z = load("topk")
    D, I = ops.topk_inner_product(T(z["q"], dev), T(z["items"], dev), int(z["k"]))
    assert close(D.cpu().numpy(), z["expected"])
    s = np.einsum('qkd,qd->qk', z["items"][I.cpu().numpy()].astype(np.float64), z["q"].astype(np.float64))
    assert close(s, z["expected"])


def test_golden_adam(dev):
    from recamd import ops
    z = load("adam")
    var, m, v = T(z["var"], dev), T(z["m"], dev), T(z["v"], dev)
    ops.adam_step(var, m, v, T(z["grad"], dev), int(z["step"]), lr=float(z["lr"]), l2=float(z["l2"]))
    got = np.stack([var.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy()])
    assert close(got, z["expected"])


def test_golden_bce_auc(dev):
    from recamd import ops
    z = load("bce_auc")
    y, p = T(z["y"], dev), T(z["p"], dev)
    got = np.array([float(ops.binary_crossentropy(y, p).cpu()), float(ops.auc(y, p).cpu())])
    assert close(got, z["expected"])
