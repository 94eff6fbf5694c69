"""The training oracle (oracle/ref_train.py, fp64 torch autograd over oracle/ref_torch.py) against the independent numpy
restatement (oracle/ref_numpy.py) in inference mode, finite differences of its gradients, and the documented dropout
mask — no GPU.  Both oracles are test infrastructure; parity with the reference itself is unpinned (TensorFlow is not
importable, the reference ships no fixtures)."""
import numpy as np

from oracle import ref_numpy as ref
from oracle import ref_train as rt


def _sasrec_weights(rng, V, d, fh, blocks):
    W = {f"user_embed_{k}/embeddings": rng.normal(size=(V, d)) * 0.3 for k in ("seq_item", "pos_item", "neg_item")}
    for b in range(blocks):
        e = f"encoder_{b}/"
        for n in ("wq", "wk", "wv"):
            W[e + f"mha/{n}/kernel"], W[e + f"mha/{n}/bias"] = rng.normal(size=(d, d)) * 0.2, rng.normal(size=d) * 0.1
        W[e + "ffn/conv1/kernel"], W[e + "ffn/conv1/bias"] = rng.normal(size=(d, fh)) * 0.2, rng.normal(size=fh) * 0.1
        W[e + "ffn/conv2/kernel"], W[e + "ffn/conv2/bias"] = rng.normal(size=(fh, d)) * 0.2, rng.normal(size=d) * 0.1
        for ln in ("layernorm1", "layernorm2"):
            W[e + ln + "/gamma"], W[e + ln + "/beta"] = 1 + 0.1 * rng.normal(size=d), 0.1 * rng.normal(size=d)
    return W


def test_sasrec_forward_and_loss_agree_with_the_numpy_oracle():
    rng = np.random.default_rng(0)
    B, S, n, V, d, fh, H, blocks = 6, 9, 5, 30, 8, 12, 2, 2
    W = _sasrec_weights(rng, V, d, fh, blocks)
    seq = rng.integers(0, V, size=(B, S))
    seq[0] = 0
    pos, neg = rng.integers(1, V, size=(B, 1)), rng.integers(1, V, size=(B, n))
    P = {k: rt.T(v) for k, v in W.items()}
    logits, loss = rt.sasrec_forward(P, seq, pos, neg, blocks, H)
    bl = []
    for b in range(blocks):
        e = f"encoder_{b}/"
        bl.append(dict(Wq=W[e + "mha/wq/kernel"], bq=W[e + "mha/wq/bias"], Wk=W[e + "mha/wk/kernel"], bk=W[e + "mha/wk/bias"],
                       Wv=W[e + "mha/wv/kernel"], bv=W[e + "mha/wv/bias"], ln1_g=W[e + "layernorm1/gamma"],
                       ln1_b=W[e + "layernorm1/beta"], W1=W[e + "ffn/conv1/kernel"], b1=W[e + "ffn/conv1/bias"],
                       W2=W[e + "ffn/conv2/kernel"], b2=W[e + "ffn/conv2/bias"], ln2_g=W[e + "layernorm2/gamma"],
                       ln2_b=W[e + "layernorm2/beta"]))
    exp, eloss = ref.sasrec_forward(seq, pos, neg, W["user_embed_seq_item/embeddings"], W["user_embed_pos_item/embeddings"],
                                    W["user_embed_neg_item/embeddings"], bl, H)
    assert np.allclose(logits.numpy(), exp, rtol=1e-11, atol=1e-12)
    assert abs(float(loss) - float(eloss)) < 1e-12
    assert np.all(exp[0] == 0.0)                       # all-padding sequence: logits exactly 0


def test_fm_forward_and_gradient_by_finite_differences():
    rng = np.random.default_rng(1)
    B, nd, vocab, k = 7, 3, [4, 6, 5], 4
    L = nd + sum(vocab)
    W = {"w0": rng.normal(size=1) * 0.1, "w": rng.normal(size=(L, 1)) * 0.3, "V": rng.normal(size=(k, L)) * 0.3}
    dense = rng.random((B, nd))
    ids = np.stack([rng.integers(-1, v + 1, size=B) for v in vocab], axis=1)       # incl. out-of-range ids
    y = (rng.random(B) < 0.5).astype(np.float64)
    p = rt.predict("fm", W, [dense, ids], vocab=vocab)
    exp = ref.fm_model_gather(dense, ids, vocab, W["w0"], W["w"], W["V"]).reshape(-1)
    assert np.allclose(p, exp, rtol=1e-11, atol=1e-12)
    grads, loss = rt.gradients("fm", W, [dense, ids], y, vocab=vocab)

    def loss_at(name, idx, h):
        W2 = {kk: v.copy() for kk, v in W.items()}
        W2[name][idx] += h
        P = {kk: rt.T(v) for kk, v in W2.items()}
        return float(rt.keras_bce(rt.fm_forward(P, dense, ids, vocab), rt.T(y)))
    for name, idx in (("w0", (0,)), ("w", (nd + 2, 0)), ("V", (1, nd + vocab[0] + 3)), ("V", (2, 1))):
        fd = (loss_at(name, idx, 1e-6) - loss_at(name, idx, -1e-6)) / 2e-6
        assert abs(fd - grads[name][idx]) <= 1e-7 + 1e-5 * abs(fd), (name, idx, fd, grads[name][idx])


def test_din_and_autoint_training_steps_move_the_loss_down():
    """sanity of the two remaining oracle forwards under autograd: a few Adam steps on a fixed batch reduce the loss"""
    rng = np.random.default_rng(2)
    B, F, V, D, nd, S, H = 16, 4, 11, 4, 2, 4, 2
    W = {f"embed_{i}/embeddings": rng.normal(size=(V, D)) * 0.3 for i in range(F)}
    W["dense_embed"] = rng.normal(size=(nd, D)) * 0.3
    for n in ("Wq", "Wk", "Wv", "W0"):
        W["attention_0/" + n] = rng.normal(size=(D, H * S)) * 0.3
    W["final_dense/kernel"], W["final_dense/bias"] = rng.normal(size=((F + nd) * H * S, 1)) * 0.3, np.zeros(1)
    dense, ids = rng.random((B, nd)), rng.integers(0, V, size=(B, F))
    y = (rng.random(B) < 0.5).astype(np.float64)
    opt = rt.AdamOracle(lr=1e-2)
    losses = [rt.train_step("autoint", W, opt, [dense, ids], y, {}, H=H, S=S, use_res=True)[1] for _ in range(15)]
    assert losses[-1] < losses[0]
    # DIN
    ukeys, ikeys, maxlen, d = ["u0"], ["i0", "i1"], 3, 4
    Wd = {f"embed_{k}/embeddings": rng.normal(size=(9, d)) * 0.3 for k in ukeys + ikeys}
    width = 2 + d + 2 + 2 * d + 2 * d
    Wd.update({"attention_layer/kernel": rng.normal(size=(8 * d, 1)) * 0.3, "attention_layer/bias": np.zeros(1),
               "bn/gamma": np.ones(width), "bn/beta": np.zeros(width), "bn/moving_mean": np.zeros(width),
               "bn/moving_variance": np.ones(width), "ffn_0/kernel": rng.normal(size=(width, 6)) * 0.3,
               "ffn_0/bias": np.zeros(6), "ffn_0/prelu/alpha": np.full(6, 0.1),
               "final_output/kernel": rng.normal(size=(6, 1)) * 0.3, "final_output/bias": np.zeros(1)})
    inputs = [rng.random((B, 2)), rng.integers(0, 9, size=(B, 1)).astype(float), rng.random((B, 2)),
              rng.integers(0, 9, size=(B, 2)).astype(float), rng.integers(0, 9, size=(B, maxlen * 2)).astype(float)]
    kw = dict(user_keys=ukeys, item_keys=ikeys, maxlen=maxlen, att_activation="sigmoid", ffn_activation="prelu", n_ffn=1)
    opt = rt.AdamOracle(lr=1e-2)
    losses = [rt.train_step("din", Wd, opt, inputs, y, {}, **kw)[1] for _ in range(15)]
    assert losses[-1] < losses[0]
    assert not np.allclose(Wd["bn/moving_mean"], 0.0)           # training-mode BatchNormalization moved its statistics


def test_dropout_mask_contract():
    """rec_dropout_f32's mask as the oracle states it: a pure function of (seed, element index), keep rate 1 - rate,
    independent across seeds; rate 0 keeps everything"""
    n = 200000
    assert rt.dropout_mask(n, 0.0, 123).all()
    for rate in (0.1, 0.5, 0.9):
        a, b = rt.dropout_mask(n, rate, 7), rt.dropout_mask(n, rate, 8)
        assert np.array_equal(a, rt.dropout_mask(n, rate, 7))
        assert abs(a.mean() - (1 - rate)) < 0.01
        assert abs((a & b).mean() - (1 - rate) ** 2) < 0.01      # two seeds: independent masks
    assert np.array_equal(rt.dropout_mask(100, 0.5, 7), rt.dropout_mask(n, 0.5, 7)[:100])    # prefix property
