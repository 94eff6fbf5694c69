"""§8f-1/-2: backward kernels, one-step parity of the training-mode models, the fit/evaluate harness — against the
fp64 torch-autograd oracle (oracle/ref_train.py).  Tolerance: |a-b| <= 1e-5 * max(|b|, floor) with the floor stated per
check (gradients of a mean loss over B samples are O(1/B))."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from oracle import ref_train as rt

pytestmark = pytest.mark.gpu


def close(a, b, tol=1e-5, floor=1e-3):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    ok = np.abs(a - b) <= tol * np.maximum(np.abs(b), floor)
    if not ok.all():
        w = np.unravel_index(np.argmax(np.abs(a - b) / np.maximum(np.abs(b), floor)), a.shape)
        print(f"close: worst at {w}: got {a[w]!r} exp {b[w]!r}; {int((~ok).sum())} of {ok.size} off")
    return bool(ok.all())


def G(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


# ---- kernels -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N", [(1, 1), (5, 3), (64, 64), (65, 130), (1000, 17), (300, 513)])
def test_transpose_and_colsum(dev, M, N):
    from recamd import train as tr
    rng = np.random.default_rng(M + N)
    a, b, w = rng.normal(size=(M, N)), rng.normal(size=(M, N)), rng.normal(size=M)
    ta = G(a, dev)
    assert np.array_equal(tr.transpose(ta).cpu().numpy(), ta.cpu().numpy().T)
    af, bf, wf = ta.cpu().numpy().astype(np.float64), G(b, dev).cpu().numpy().astype(np.float64), G(w, dev).cpu().numpy().astype(np.float64)
    assert close(tr.colsum(ta).cpu().numpy(), af.sum(0), floor=1.0)
    assert close(tr.colsum(ta, G(b, dev), G(w, dev)).cpu().numpy(), (af * bf * wf[:, None]).sum(0), floor=1.0)
    assert close(tr.sum_squares(ta).cpu().numpy(), [(af ** 2).sum()], floor=1.0)


@pytest.mark.parametrize("M,N", [(7, 5), (256, 40), (1000, 130)])
def test_batchnorm_training_forward_backward(dev, M, N):
    from recamd import nn, train as tr
    rng = np.random.default_rng(M)
    x = rng.normal(size=(M, N)) * 2 + 0.5
    dy = rng.normal(size=(M, N))
    bn = nn.BatchNormalization()
    bn.build(N)
    w0 = {"gamma": 1 + 0.2 * rng.normal(size=N), "beta": rng.normal(size=N) * 0.1,
          "moving_mean": rng.normal(size=N) * 0.1, "moving_variance": rng.uniform(0.5, 1.5, size=N)}
    bn.set_weights({k: v.astype(np.float32) for k, v in w0.items()})
    tape = tr.Tape()
    xv = tr.Var(G(x, dev))
    out = tr.bn_fwd(tape, bn, "bn", xv)
    out.g = G(dy, dev)
    tape.backward()
    P = {"bn/" + k: rt.T(np.float32(v), grad=k in ("gamma", "beta")) for k, v in w0.items()}
    xt = rt.T(np.float32(x), grad=True)
    nm = {}
    yt = rt.bn_train(xt, P, "bn", nm)
    yt.backward(rt.T(np.float32(dy)))
    # floors = the scale of the operands each value is a difference / sum of (normalised values are O(1), the
    # column sums O(sqrt(M))): errors are fp32 rounding of THOSE, not of the (possibly tiny) result
    assert close(out.v.cpu().numpy(), yt.detach().numpy(), floor=1.0)
    assert close(xv.g.cpu().numpy(), xt.grad.numpy(), floor=1.0)
    assert close(tape.grads["bn/gamma"].cpu().numpy(), P["bn/gamma"].grad.numpy(), floor=float(np.sqrt(M)))
    assert close(tape.grads["bn/beta"].cpu().numpy(), P["bn/beta"].grad.numpy(), floor=float(np.sqrt(M)))
    got = bn.get_weights()
    assert close(got["moving_mean"], nm["bn/moving_mean"].numpy())
    assert close(got["moving_variance"], nm["bn/moving_variance"].numpy())


@pytest.mark.parametrize("act", [None, "relu", "sigmoid", "tanh"])
def test_dense_backward(dev, act):
    from recamd import nn, train as tr
    rng = np.random.default_rng(7)
    M, K, N = 300, 40, 24
    x, W, b, dy = rng.normal(size=(M, K)), rng.normal(size=(K, N)) * 0.3, rng.normal(size=N) * 0.1, rng.normal(size=(M, N))
    layer = nn.Dense(N, activation=act)
    layer.build(K)
    layer.set_weights({"kernel": W.astype(np.float32), "bias": b.astype(np.float32)})
    tape = tr.Tape()
    xv = tr.Var(G(x, dev))
    y = tr.dense_fwd(tape, layer, "d", xv)
    y.g = G(dy, dev)
    tape.backward()
    xt, Wt, bt = rt.T(np.float32(x), True), rt.T(np.float32(W), True), rt.T(np.float32(b), True)
    yt = rt.act(xt @ Wt + bt, act)
    yt.backward(rt.T(np.float32(dy)))
    assert close(y.v.cpu().numpy(), yt.detach().numpy(), floor=1.0)      # sums of K = 40 products of O(1) operands
    assert close(xv.g.cpu().numpy(), xt.grad.numpy(), floor=1.0)
    assert close(tape.grads["d/kernel"].cpu().numpy(), Wt.grad.numpy(), floor=1.0)
    assert close(tape.grads["d/bias"].cpu().numpy(), bt.grad.numpy(), floor=1.0)


def test_bce_sigmoid_grad(dev):
    from recamd._lib import C
    rng = np.random.default_rng(2)
    n = 5000
    z = rng.normal(size=n) * 4
    z[:4] = [40.0, -40.0, 17.0, -17.0]            # saturated: outside the clip the gradient is exactly 0
    y = (rng.random(n) < 0.4).astype(np.float32)
    zt = rt.T(np.float32(z), True)
    loss = rt.keras_bce(torch.sigmoid(zt), rt.T(y))
    loss.backward()
    p = torch.sigmoid(G(z, dev))
    dz = torch.empty(n, device=dev)
    C.bce_sigmoid_grad_f32(G(y, dev).data_ptr(), p.data_ptr(), n, 1.0 / n, dz.data_ptr(), torch.cuda.current_stream().cuda_stream)
    pin = p.cpu().numpy()
    got = dz.cpu().numpy()
    # (1) the kernel's arithmetic, exactly: the same formula in fp64 on the SAME fp32 probabilities
    p64, y64, e = pin.astype(np.float64), y.astype(np.float64), 1e-7
    pc = np.clip(p64, np.float32(e), np.float32(1) - np.float32(e))
    inside = (pin > np.float32(e)) & (pin < np.float32(1) - np.float32(e))
    exp = np.where(inside, (-(y64 / (pc + e)) + (1 - y64) / (1 - pc + e)) * p64 * (1 - p64), 0.0) / n
    assert close(got, exp, tol=1e-5, floor=1.0 / n)
    assert np.all(got[~inside] == 0.0)                # outside the clip the gradient is exactly 0
    # (2) against autograd through the fp64 sigmoid, where the probability form is well conditioned (fp32 p near 1
    # quantises 1 - p: at p = 1 - 1.4e-6 the loss gradient moves by 3e-3 relative — inherent to BCE on probabilities)
    well = (pin > 1e-3) & (pin < 1 - 1e-3)
    assert close(got[well], zt.grad.numpy()[well], tol=2e-4, floor=1.0 / n)


@pytest.mark.parametrize("F,D", [(26, 128), (8, 128), (3, 128), (8, 64)])
def test_gather_pairwise_dot_backward(dev, F, D):
    from recamd import ops
    from recamd._lib import C
    rng = np.random.default_rng(F + D)
    B, V = 37, 11
    tables = [rng.normal(size=(V, D)).astype(np.float32) for _ in range(F)]
    ids = rng.integers(-1, V + 1, size=(B, F)).astype(np.int32)        # duplicates and out-of-range ids
    dense = rng.normal(size=(B, D)).astype(np.float32)
    n = F + 1
    P_ = n * (n - 1) // 2
    dz = rng.normal(size=(B, P_ + D)).astype(np.float32)
    tt = [torch.from_numpy(t).to(dev) for t in tables]
    gt = [torch.zeros_like(t) for t in tt]
    g, gg = ops.TableGroup(tt), ops.TableGroup(gt)
    t_ids, t_dense, t_dz = torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev), torch.from_numpy(dz).to(dev)
    dd = torch.empty((B, D), device=dev)
    C.gather_pairwise_dot_grad_f32(g.descs, gg.descs, t_ids.data_ptr(), t_ids.stride(0), t_dense.data_ptr(), t_dense.stride(0),
                                   B, t_dz.data_ptr(), t_dz.stride(0), 1, dd.data_ptr(), dd.stride(0),
                                   torch.cuda.current_stream().cuda_stream)
    P = {f"embed_{i}/embeddings": rt.T(tables[i], True) for i in range(F)}
    dt = rt.T(dense, True)
    emb = rt.gather_concat(P, ids)
    X = torch.cat([emb.view(B, F, D), dt[:, None, :]], dim=1)
    Z = X @ X.transpose(1, 2)
    li, lj = zip(*[(i, j) for i in range(n) for j in range(i)])
    out = torch.cat([Z[:, list(li), list(lj)], dt], dim=-1)
    out.backward(rt.T(dz))
    assert close(dd.cpu().numpy(), dt.grad.numpy(), floor=1.0)
    for i in range(F):
        assert close(gt[i].cpu().numpy(), P[f"embed_{i}/embeddings"].grad.numpy(), floor=1.0)


def test_fm_layer_and_cross_backward(dev):
    from ctr.layers.modules import FM, CrossNetwork
    from recamd import train as tr
    rng = np.random.default_rng(5)
    B, L1, M = 50, 29, 24
    first, second, w, dout = rng.normal(size=(B, L1)), rng.normal(size=(B, M)), rng.normal(size=(L1, 1)), rng.normal(size=(B, 1))
    fm = FM(L1)
    fm.set_weights({"w": w.astype(np.float32)})
    tape = tr.Tape()
    fv, sv = tr.Var(G(first, dev)), tr.Var(G(second, dev))
    out = tr.fm_fwd(tape, fm, "fm", fv, sv)
    out.g = G(dout, dev)
    tape.backward()
    ft, st_, wt = rt.T(np.float32(first), True), rt.T(np.float32(second), True), rt.T(np.float32(w), True)
    o = (torch.sum(ft @ wt) + 0.5 * (st_.sum(1) ** 2 - (st_ ** 2).sum(1))).reshape(-1, 1)
    o.backward(rt.T(np.float32(dout)))
    assert close(out.v.cpu().numpy(), o.detach().numpy(), floor=1.0)
    assert close(fv.g.cpu().numpy(), ft.grad.numpy(), floor=1.0)
    assert close(sv.g.cpu().numpy(), st_.grad.numpy(), floor=1.0)
    assert close(tape.grads["fm/w"].cpu().numpy(), wt.grad.numpy(), floor=10.0)
    # cross network, 3 layers
    dim, L = 40, 3
    x, W, Bv, g = rng.normal(size=(B, dim)) * 0.5, rng.normal(size=(L, dim)) * 0.2, rng.normal(size=(L, dim)) * 0.1, rng.normal(size=(B, dim))
    cn = CrossNetwork(L)
    cn.build(dim)
    cn.set_weights({"cross_weights": W.astype(np.float32), "cross_bias": Bv.astype(np.float32)})
    tape = tr.Tape()
    xv = tr.Var(G(x, dev))
    y = tr.cross_fwd(tape, cn, "c", xv)
    y.g = G(g, dev)
    tape.backward()
    xt, Wt, Bt = rt.T(np.float32(x), True), rt.T(np.float32(W), True), rt.T(np.float32(Bv), True)
    xl = xt
    for l in range(L):
        xl = xt * (xl @ Wt[l])[:, None] + Bt[l] + xl
    xl.backward(rt.T(np.float32(g)))
    assert close(y.v.cpu().numpy(), ref.cross_network(np.float32(x), np.float32(W), np.float32(Bv)), floor=1e-1)
    assert close(xv.g.cpu().numpy(), xt.grad.numpy(), floor=1e-1)
    assert close(tape.grads["c/cross_weights"].cpu().numpy(), Wt.grad.numpy(), floor=1.0)
    assert close(tape.grads["c/cross_bias"].cpu().numpy(), Bt.grad.numpy(), floor=1.0)


def test_dense_cache_invalidated_by_optimizer_write(dev):
    """round-1 ADVICE: ops.dense caches pre-split weights by torch's version counter; rec_adam_f32 writes through a raw
    pointer.  dense -> adam_step on W -> dense must use the UPDATED W."""
    from recamd import ops
    rng = np.random.default_rng(11)
    M, K, N = 2048, 256, 128                     # large enough for the prepared-weights (bf16x3) path
    x, W = G(rng.normal(size=(M, K)), dev), G(rng.normal(size=(K, N)) * 0.1, dev)
    y0 = ops.dense(x, W).cpu().numpy()
    m, v, g = torch.zeros_like(W), torch.zeros_like(W), G(rng.normal(size=(K, N)), dev)
    ops.adam_step(W, m, v, g, 1, lr=0.05)
    y1 = ops.dense(x, W).cpu().numpy()
    exp = x.cpu().numpy().astype(np.float64) @ W.cpu().numpy().astype(np.float64)
    assert close(y1, exp, floor=1.0)                 # 256-term sums of O(0.1) products
    assert not np.allclose(y0, exp, atol=1e-3)


# ---- one optimiser step of the models ----------------------------------------------------------------------------
def _setup(kind, dev, rng, B=96, scale=0.3):
    from ctr.dcn.model import DCN
    from ctr.deep_fm.model import DeepFM
    from ctr.dlrm.model import DLRM
    F, V, D, nd = 5, 23, 128 if kind.startswith("dlrm_dot") else 8, 6
    sparse = [{'feat': f'C{i}', 'feat_num': V + i, 'embed_dim': D} for i in range(F)]
    densec = [{'feat': f'I{i}'} for i in range(nd)]
    dense = rng.random((B, nd)).astype(np.float32)
    ids = np.stack([rng.integers(0, V + i, size=B) for i in range(F)], axis=1).astype(np.int32)
    ids[B // 2:] = ids[:B - B // 2]
    y = (rng.random(B) < 0.4).astype(np.float32)
    if kind == "dlrm_dot":
        m, okind, kw, inputs = DLRM([densec, sparse], [32, D], [48, 16], interaction='dot', embed_reg=1e-4), "dlrm", {"interaction": "dot"}, [dense, ids]
    elif kind == "dlrm_cat":
        m, okind, kw, inputs = DLRM([densec, sparse], [16, 8], [24, 8], interaction='cat', embed_reg=1e-4), "dlrm", {"interaction": "cat"}, [dense, ids]
    elif kind == "deepfm":
        m, okind, kw, inputs = DeepFM([densec, sparse], (32, 16), embed_reg=1e-4, fm_w_reg=1e-3), "deepfm", {}, [dense, ids]
    else:
        m, okind, kw, inputs = DCN(sparse, [24, 12], embed_reg=1e-4), "dcn", {}, ids
    m(inputs)                                   # builds the lazily created layers
    from tests.test_models_gpu import randomize
    randomize(m, rng, scale)
    for k in [k for k in m.get_weights() if k.endswith("gamma")]:
        m.set_weights({k: (1 + 0.1 * rng.normal(size=m.get_weights()[k].shape)).astype(np.float32)})
    return m, okind, kw, inputs, y


@pytest.mark.parametrize("kind", ["dlrm_dot", "dlrm_cat", "deepfm", "dcn"])
@pytest.mark.parametrize("sparse", [False, True])
def test_one_training_step_matches_oracle(dev, kind, sparse):
    """BCE loss + Keras-Adam on synthetic data: every parameter after TWO steps equals the oracle's (1e-5 of max(|w|,
    1e-2)); with the lazy row-wise embedding update the touched rows still match an oracle that only updates touched
    rows (checked through the dense parameters and the looked-up rows)."""
    from recamd import train as tr
    rng = np.random.default_rng({"dlrm_dot": 1, "dlrm_cat": 2, "deepfm": 3, "dcn": 4}[kind])
    m, okind, kw, inputs, y = _setup(kind, dev, rng)
    w0 = tr_weights(m)
    W = {k: v.astype(np.float64) for k, v in w0.items()}
    l2 = tr.default_l2(m)
    opt = tr.Adam(m, 1e-2, l2=l2, sparse_embeddings=sparse)
    state = tr.TrainState(m)
    oo = rt.AdamOracle(lr=1e-2)
    for step in range(1 if sparse else 2):       # the oracle is the exact (dense) form: the lazy form agrees with it
        p, loss = tr.train_step(m, opt, state, inputs, y)     # on the touched rows of the FIRST step only
        ep, ebce, ereg = rt.train_step(okind, W, oo, inputs, y, l2, **kw)
        assert close(p.cpu().numpy().reshape(-1), ep, floor=1.0)          # probabilities: 1e-5 absolute
        assert abs(float(loss.item()) - ebce) <= 1e-5 * max(1.0, abs(ebce))
    got = tr_weights(m)
    ids = inputs[1] if isinstance(inputs, list) else inputs
    for k, e in W.items():
        if sparse and k.endswith("embeddings"):
            f = int(k.split("/")[0].split("_")[1])
            rows = np.unique(ids[:, f])
            assert np.all(np.abs(got[k][rows] - e[rows]) <= 1e-5 * np.maximum(np.abs(e[rows]), 1e-2) + 1e-3 * 1e-2), k
            untouched = np.setdiff1d(np.arange(e.shape[0]), rows)
            assert np.array_equal(got[k][untouched], w0[k][untouched]), k          # the documented deviation
        else:   # Adam divides by sqrt(v) + eps: where |g| is of the order of eps the update amplifies the rounding of g:
            # allow 1e-3 of one full-size update (lr) on top of the 1e-5 band
            # (an element whose gradient is ~0 gets an update ~ lr * g / eps_hat: a handful may move by up to 1 % of lr)
            d = np.abs(got[k] - e)
            band = 1e-5 * np.maximum(np.abs(e), 1e-2) + 1e-3 * 1e-2
            assert (d > band).mean() <= 1e-3 and d.max() <= 1e-2 * 1e-2, (k, float(d.max()), int((d > band).sum()), d.size)


def tr_weights(m):
    from recamd import train as tr
    return {k: v.detach().cpu().numpy() for k, v in tr.named_weights(m).items()}


def test_data_parallel_replicas_stay_identical(dev):
    """Two replicas, different batches, gradients merged as MirroredStrategy's all-reduce does (sum of the replicas'
    1/world-scaled gradients): both replicas end with identical weights, equal to the oracle's step on the mean of the
    two replica losses (BatchNormalization statistics stay per replica)."""
    from recamd import train as tr
    rng = np.random.default_rng(9)
    ma, okind, kw, inp_a, ya = _setup("deepfm", dev, rng)
    mb, _, _, inp_b, yb = _setup("deepfm", dev, np.random.default_rng(10))
    mb.set_weights(ma.get_weights())
    W = {k: v.astype(np.float64) for k, v in tr_weights(ma).items()}
    l2 = tr.default_l2(ma)
    reps = [(ma, tr.Adam(ma, 1e-2, l2=l2), tr.TrainState(ma), inp_a, ya), (mb, tr.Adam(mb, 1e-2, l2=l2), tr.TrainState(mb), inp_b, yb)]
    outs = [tr.compute_gradients(m, st, inp, y, 0.5) for m, _, st, inp, y in reps]
    names = sorted(outs[0][2])
    for k in names:                                              # the all-reduce (sum), done by hand
        s = outs[0][2][k] + outs[1][2][k]
        outs[0][2][k], outs[1][2][k] = s, s.clone()
    for k in sorted(reps[0][2]._grads):
        s = reps[0][2]._grads[k] + reps[1][2]._grads[k]
        reps[0][2]._grads[k].copy_(s)
        reps[1][2]._grads[k].copy_(s)
    for (m, opt, st, _, _), (_, _, grads) in zip(reps, outs):
        opt.apply(grads, st)
    wa, wb = tr_weights(ma), tr_weights(mb)
    for k in wa:
        if not rt.is_moving(k):
            assert np.array_equal(wa[k], wb[k]), k
    # oracle: loss = (bce_a + bce_b) / 2 + reg
    P = {k: rt.T(v, grad=not rt.is_moving(k)) for k, v in W.items()}
    pa = rt.deepfm_forward(P, inp_a[0], inp_a[1], training=True)
    pb = rt.deepfm_forward(P, inp_b[0], inp_b[1], training=True)
    (0.5 * (rt.keras_bce(pa, rt.T(ya)) + rt.keras_bce(pb, rt.T(yb))) + rt.reg_loss(P, l2)).backward()
    oo = rt.AdamOracle(lr=1e-2)
    oo.apply(W, {k: v.grad.numpy() for k, v in P.items() if v.requires_grad and v.grad is not None})
    for k, e in W.items():
        if not rt.is_moving(k):
            assert close(wa[k], e, floor=1e-2), k


def test_sparse_adam_refuses_data_parallel(dev):
    """the lazy row-wise Adam updates (and clears) only this replica's rows; with merged gradients the other
    replicas' rows would be neither applied nor cleared and the replicas would diverge — refused, not silently wrong."""
    from recamd import train as tr
    rng = np.random.default_rng(9)
    m, okind, kw, inp, y = _setup("deepfm", dev, rng)
    merge = lambda t: t  # noqa: E731
    with pytest.raises(NotImplementedError):
        tr.Trainer(m).compile(sparse_embeddings=True, allreduce=merge, world=2)
    opt = tr.Adam(m, 1e-2, l2=tr.default_l2(m), sparse_embeddings=True)
    with pytest.raises(NotImplementedError):
        tr.train_step(m, opt, tr.TrainState(m), inp, y, allreduce=merge, world=2)
    tr.train_step(m, opt, tr.TrainState(m), inp, y)            # a single replica: fine


def test_fit_evaluate_early_stopping_checkpoint(dev, tmp_path):
    """compile / fit / evaluate as src/ctr/deep_fm/train.py:44-68 runs them, on seeded synthetic data: per-epoch loss
    and AUC equal the oracle loop's; EarlyStopping(patience=1, restore_best_weights=True) restores the best epoch's
    weights; a weights-only checkpoint round-trips."""
    from recamd import train as tr
    rng = np.random.default_rng(21)
    m, okind, kw, inputs, y = _setup("deepfm", dev, rng, B=403, scale=0.05)   # 403 * 0.2 is not an integer
    w_init = m.get_weights()                   # inference-mode BN must not saturate the sigmoid: sane moving statistics
    m.set_weights({k: (np.zeros_like(v) if k.endswith("moving_mean") else np.ones_like(v))
                   for k, v in w_init.items() if "moving_" in k})
    m.set_weights({k: (v * 0.1).astype(np.float32) for k, v in w_init.items() if k.endswith("embeddings")})
    W = {k: v.astype(np.float64) for k, v in tr_weights(m).items()}
    l2 = tr.default_l2(m)
    trainer = tr.Trainer(m).compile(learning_rate=5e-3)
    es = tr.EarlyStopping(monitor="val_loss", patience=1, restore_best_weights=True)
    hist = trainer.fit(inputs, y, batch_size=64, epochs=3, validation_split=0.2, callbacks=[es], shuffle=True, seed=3)
    # the oracle loop
    n = len(y)
    n_tr = int(np.floor(n * (1.0 - 0.2)))      # Keras: split_at = floor(n * (1 - validation_split)) -> 322 train / 81 val
    assert n_tr == 322 and n - n_tr == 81 and int(n * 0.2) == 80      # (int(n * 0.2) would give 323 / 80)
    oo = rt.AdamOracle(lr=5e-3)
    sl = lambda idx: [inputs[0][idx], inputs[1][idx]]  # noqa: E731
    e_hist = {"loss": [], "auc": [], "val_loss": [], "val_auc": []}
    best, best_w = None, None
    for epoch in range(3):
        order = np.random.default_rng(3 + epoch).permutation(n_tr)
        tot, preds, labs = 0.0, [], []
        for lo in range(0, n_tr, 64):
            idx = order[lo:lo + 64]
            p, bce, reg = rt.train_step(okind, W, oo, sl(idx), y[idx], l2)
            tot += (bce + reg) * len(idx)
            preds.append(p)
            labs.append(y[idx])
        e_hist["loss"].append(tot / n_tr)
        e_hist["auc"].append(ref.keras_auc(np.concatenate(labs), np.concatenate(preds)))
        # evaluate() runs in batches too, and the FM layer's first-order term is ONE scalar per BATCH
        # (src/ctr/layers/modules.py:65): predictions depend on the batch a sample is evaluated in — same chunks here
        pv = np.concatenate([rt.predict(okind, W, sl(slice(lo, min(n, lo + 64)))) for lo in range(n_tr, n, 64)])
        P = {k: rt.T(v) for k, v in W.items()}
        vl = ref.binary_crossentropy(y[n_tr:], pv.astype(np.float32), dtype=np.float32) + float(rt.reg_loss(P, l2))
        e_hist["val_loss"].append(vl)
        e_hist["val_auc"].append(ref.keras_auc(y[n_tr:], pv))
        if best is None or vl < best:
            best, best_w = vl, {k: v.copy() for k, v in W.items()}
        elif True:
            W = best_w
            break
    k = len(hist["loss"])
    print("hist", hist, "oracle", e_hist)
    assert k == len(e_hist["loss"])
    for key in ("loss", "val_loss"):
        assert close(hist[key], e_hist[key][:k], tol=2e-4, floor=1e-1), key
    for key in ("auc", "val_auc"):
        assert np.all(np.abs(np.asarray(hist[key]) - np.asarray(e_hist[key][:k])) <= 5e-3), key
    got = tr_weights(m)
    for name, e in W.items():
        assert close(got[name], e, tol=2e-3, floor=1e-1), name
    path = str(tmp_path / "w.npz")
    trainer.save_weights(path)
    before = trainer.evaluate(inputs, y)
    trainer.load_state({kk: torch.zeros_like(v) for kk, v in tr.named_weights(m).items() if kk.startswith("dense/")})
    trainer.load_weights(path)
    after = trainer.evaluate(inputs, y)
    assert before == after


def test_compile_accepts_the_keras_optimizer_string(dev):
    """INTEGRATION.md §A writes tr.compile(optimizer='adam', learning_rate=...) as the reference's scripts do"""
    from recamd import train as tr
    m, _, _, inputs, y = _setup("dcn", dev, np.random.default_rng(5), B=32)
    t = tr.Trainer(m).compile(optimizer='adam', learning_rate=1e-3)
    assert isinstance(t.opt, tr.Adam) and t.opt.lr0 == 1e-3
    with pytest.raises(NotImplementedError):
        tr.Trainer(m).compile(optimizer='sgd')
    hist = t.fit(inputs, y, batch_size=16, epochs=1)
    assert len(hist["loss"]) == 1 and np.isfinite(hist["loss"][0])
