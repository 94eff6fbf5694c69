"""Generates the committed golden vectors (tests/golden/*.npz): small seeded inputs + the fp64
output of oracle/ref_numpy.py.  The reference cannot be executed (TensorFlow absent), so these are
oracle-generated fixtures ("parity unpinned"), cross-checked by the torch restatement and the KATs in
tests/test_oracle_kat.py.  The GPU tests compare the HIP kernels against these files too.

    python tests/golden/make_golden.py      # rewrites the .npz files deterministically
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import ref_numpy as ref  # noqa: E402


def _keys(z):
    return z.files if hasattr(z, "files") else list(z.keys())


def _tables(z, n):
    return [z[f"table_{i}"] for i in range(n)]


CASES = {
    "gather": lambda z: ref.gather_concat(_tables(z, int(z["F"])), z["ids"]),
    "fm_model": lambda z: ref.fm_model_onehot(z["dense"], z["ids"], [int(v) for v in z["vocab"]], z["w0"], z["w"], z["V"]),
    "fm_layer": lambda z: ref.fm_layer(z["first"], z["second"], z["w"]),
    "cross": lambda z: ref.cross_network(z["x"], z["W"], z["Bv"]),
    "pairwise_dot": lambda z: ref.pairwise_dot(z["x"]),
    "mha_ctr": lambda z: ref.mha_ctr(z["x"], z["x"], z["x"], z["Wq"], z["Wk"], z["Wv"], z["W0"], int(z["H"]), int(z["S"]), "relu"),
    "din_attention": lambda z: ref.din_attention_layer(z["q"], z["k"], z["k"], z["mask"], z["W"], z["b"], "sigmoid"),
    "sasrec": lambda z: ref.sasrec_forward(z["seq"], z["pos"], z["neg"], z["T_seq"], z["T_pos"], z["T_neg"],
                                           [{k[3:]: z[k] for k in _keys(z) if k.startswith("b0_")}], 1)[0],
    "dlrm_dot": lambda z: ref.pairwise_dot(np.concatenate(
        [ref.gather_concat(_tables(z, int(z["F"])), z["ids"]).reshape(z["ids"].shape[0], int(z["F"]), -1),
         z["dense"][:, None, :]], axis=1)),
    # --- widened scope: DCN end to end, retrieval top-k, Adam step, BCE + AUC
    "dcn": lambda z: ref.dcn_forward(z["ids"], _tables(z, int(z["F"])), z["cW"], z["cB"],   # 2 cross layers = len(hidden_units)
                                     dict(layers=[(z["W0"], z["b0"]), (z["W1"], z["b1"])],
                                          bn=dict(gamma=z["bn_g"], beta=z["bn_b"], mean=z["bn_m"], var=z["bn_v"])),
                                     (z["Wf"], z["bf"])),
    "topk": lambda z: ref.topk_inner_product(z["q"], z["items"], int(z["k"]))[0],
    "adam": lambda z: np.stack(ref.adam_step(z["var"], z["m"], z["v"], z["grad"], int(z["step"]), lr=float(z["lr"]),
                                             l2=float(z["l2"]))),
    "bce_auc": lambda z: np.array([ref.binary_crossentropy(z["y"], z["p"]), ref.keras_auc(z["y"], z["p"])]),
}


def build_inputs():
    rng = np.random.default_rng(20260101)
    f32 = lambda a: np.asarray(a, np.float32)  # noqa: E731
    cases = {}
    # gather: 5 fields, mixed-in OOB ids
    vocabs = [11, 50, 7, 33, 20]
    d = {f"table_{i}": f32(rng.uniform(-0.05, 0.05, size=(v, 16))) for i, v in enumerate(vocabs)}
    d["ids"] = np.stack([rng.integers(-1, v + 1, size=40) for v in vocabs], axis=1).astype(np.int32)
    d["F"] = np.int32(5)
    cases["gather"] = d
    # ctr FM (config 1 shape, small vocab)
    vocab = [int(v) for v in rng.integers(2, 30, size=26)]
    L = 13 + sum(vocab)
    cases["fm_model"] = dict(dense=f32(rng.random((32, 13))), ids=np.stack([rng.integers(0, v, size=32) for v in vocab], 1).astype(np.int32),
                             vocab=np.array(vocab, np.int64), w0=f32([0.05]), w=f32(rng.normal(size=(L, 1)) * 0.05),
                             V=f32(rng.normal(size=(10, L)) * 0.05))
    cases["fm_layer"] = dict(first=f32(rng.normal(size=(48, 13 + 26 * 8)) * 0.1), second=f32(rng.normal(size=(48, 26 * 8)) * 0.1),
                             w=f32(rng.normal(size=(13 + 26 * 8, 1)) * 0.05))
    cases["cross"] = dict(x=f32(rng.normal(size=(24, 208)) * 0.1), W=f32(rng.normal(size=(3, 208)) * 0.05),
                          Bv=f32(rng.normal(size=(3, 208)) * 0.05))
    cases["pairwise_dot"] = dict(x=f32(rng.normal(size=(16, 27, 128)) * 0.1))
    cases["mha_ctr"] = dict(x=f32(rng.normal(size=(8, 39, 16)) * 0.5), Wq=f32(rng.normal(size=(16, 32)) * 0.25),
                            Wk=f32(rng.normal(size=(16, 32)) * 0.25), Wv=f32(rng.normal(size=(16, 32)) * 0.25),
                            W0=f32(rng.normal(size=(16, 32)) * 0.25), H=np.int32(2), S=np.int32(16))
    lens = rng.integers(0, 21, size=12)
    mask = (np.arange(20)[None, :] >= (20 - lens)[:, None]).astype(np.float32)
    cases["din_attention"] = dict(q=f32(rng.normal(size=(12, 64))), k=f32(rng.normal(size=(12, 20, 64))), mask=mask,
                                  W=f32(rng.normal(size=(256, 1)) * 0.1), b=f32([0.1]))
    V, S, dm, n, B = 60, 12, 64, 10, 6
    g = lambda *s: f32(rng.normal(size=s) * 0.15)  # noqa: E731
    blk = dict(Wq=g(dm, dm), bq=g(dm), Wk=g(dm, dm), bk=g(dm), Wv=g(dm, dm), bv=g(dm), W1=g(dm, 128), b1=g(128), W2=g(128, dm),
               b2=g(dm), ln1_g=f32(1 + 0.1 * rng.normal(size=dm)), ln1_b=g(dm), ln2_g=f32(1 + 0.1 * rng.normal(size=dm)), ln2_b=g(dm))
    seq = rng.integers(1, V, size=(B, S))
    sl = rng.integers(0, S + 1, size=B)
    seq[np.arange(S)[None, :] < (S - sl)[:, None]] = 0
    d = dict(seq=seq.astype(np.int32), pos=rng.integers(1, V, size=(B, 1)).astype(np.int32),
             neg=rng.integers(1, V, size=(B, n)).astype(np.int32), T_seq=f32(rng.uniform(-0.5, 0.5, size=(V, dm))),
             T_pos=f32(rng.uniform(-0.5, 0.5, size=(V, dm))), T_neg=f32(rng.uniform(-0.5, 0.5, size=(V, dm))))
    d.update({"b0_" + k: v for k, v in blk.items()})
    cases["sasrec"] = d
    d = {f"table_{i}": f32(rng.uniform(-0.05, 0.05, size=(40, 128))) for i in range(26)}
    d["ids"] = rng.integers(0, 40, size=(10, 26)).astype(np.int32)
    d["dense"] = f32(rng.random((10, 128)))
    d["F"] = np.int32(26)
    cases["dlrm_dot"] = d
    # ---- cases added later draw from their own generator so that the files above stay byte-identical
    rng2 = np.random.default_rng(20260102)
    f2 = lambda a: np.asarray(a, np.float32)  # noqa: E731
    F, D, V, B = 6, 8, 25, 20
    dim = F * D
    d = {f"table_{i}": f2(rng2.uniform(-0.5, 0.5, size=(V, D))) for i in range(F)}
    d.update(ids=rng2.integers(0, V, size=(B, F)).astype(np.int32), F=np.int32(F),
             cW=f2(rng2.normal(size=(2, dim)) * 0.2), cB=f2(rng2.normal(size=(2, dim)) * 0.1),
             W0=f2(rng2.normal(size=(dim, 16)) * 0.2), b0=f2(rng2.normal(size=16) * 0.1),
             W1=f2(rng2.normal(size=(16, 8)) * 0.2), b1=f2(rng2.normal(size=8) * 0.1),
             bn_g=f2(1 + 0.1 * rng2.normal(size=dim)), bn_b=f2(rng2.normal(size=dim) * 0.1),
             bn_m=f2(rng2.normal(size=dim) * 0.1), bn_v=f2(rng2.uniform(0.5, 1.5, size=dim)),
             Wf=f2(rng2.normal(size=(dim + 8, 1)) * 0.2), bf=f2([0.05]))
    cases["dcn"] = d
    cases["topk"] = dict(q=f2(rng2.normal(size=(37, 32))), items=f2(rng2.normal(size=(300, 32))), k=np.int32(10))
    n = 1000
    cases["adam"] = dict(var=f2(rng2.normal(size=n) * 0.05), m=f2(rng2.normal(size=n) * 0.01),
                         v=f2(rng2.random(n) * 1e-3), grad=f2(rng2.normal(size=n) * (rng2.random(n) < 0.3)),
                         step=np.int32(3), lr=np.float32(1e-3), l2=np.float32(1e-4))
    y = (rng2.random(4000) < 0.3).astype(np.float32)
    p = 1.0 / (1.0 + np.exp(-(rng2.normal(size=4000) + 1.2 * (2 * y - 1))))
    cases["bce_auc"] = dict(y=y, p=f2(p))
    return cases


def main():
    for name, inp in build_inputs().items():
        exp = np.asarray(CASES[name](inp), np.float64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), expected=exp, **inp)
        print(f"{name}: expected {exp.shape}")


if __name__ == "__main__":
    main()
