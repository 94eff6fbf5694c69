"""§8f-1/-2 timing: one optimiser step (training-mode forward, backward, Keras-Adam) of every mirror that has a training
forward, at the BASELINE shapes scaled to training batch sizes.  Prints one JSON object; not part of bench.py's contract.
The step includes the exact dense Adam over every table row (28 B per parameter), which dominates the models with large
tables — `sparse_embeddings=True` (lazy row-wise Adam) is timed beside it for the (B, F) id models."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]
from recamd import train as tr  # noqa: E402


def step_time(model, inputs, y, steps=8, sparse=False):
    opt = tr.Adam(model, 1e-3, l2=tr.default_l2(model), sparse_embeddings=sparse)
    state = tr.TrainState(model)
    for _ in range(2):
        tr.train_step(model, opt, state, inputs, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(model, opt, state, inputs, y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    rng = np.random.default_rng(0)
    res = {}

    def rec(name, model, inputs, y, B, sparse_too=False, note=""):
        params = sum(v.numel() for v in tr.named_weights(model).values())
        ms = step_time(model, inputs, y)
        res[name] = {"batch": B, "parameters": params, "ms_per_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1), "note": note}
        if sparse_too:
            ms2 = step_time(model, inputs, y, sparse=True)
            res[name]["lazy_rowwise_adam"] = {"ms_per_step": round(ms2, 3), "samples_per_s": round(B / ms2 * 1e3, 1)}
        print(name, res[name], flush=True)

    F, V, nd = 26, 100_000, 13
    sparse128 = [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': 128} for i in range(F)]
    sparse16 = [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': 16} for i in range(F)]
    densec = [{'feat': f'I{i}'} for i in range(nd)]
    B = 8192
    dense = rng.random((B, nd)).astype(np.float32)
    ids = rng.integers(0, V, size=(B, F)).astype(np.int32)
    y = (rng.random(B) < 0.3).astype(np.float32)
    from ctr.dlrm.model import DLRM
    from ctr.deep_fm.model import DeepFM
    from ctr.dcn.model import DCN
    from ctr.autoint.model import AutoInt
    from ctr.fm.model import FM
    from ctr.din.model import DIN
    from ctr.wide_deep.model import WideDeep
    from match.sasrec.model import SASRec
    from match.ncf.model import NCF
    if os.environ.get("FULL") == "1":
        # BASELINE configs[1] at its real size: 26 x 1M x 128 tables (13.3 GB of weights, as much again for each Adam moment and
        # for the dense gradient arena), batch 65 536 — the exact Keras-Adam step touches every row, the lazy one ~1.7 M of them
        Vf, Bf2 = 1_000_000, 65536
        sp = [{'feat': f'C{i}', 'feat_num': Vf, 'embed_dim': 128} for i in range(F)]
        dn = torch.rand((Bf2, nd), device="cuda:0")
        idf2 = torch.randint(0, Vf, (Bf2, F), device="cuda:0", dtype=torch.int32)
        yf = (torch.rand(Bf2, device="cuda:0") < 0.3).float()
        m = DLRM([densec, sp], [512, 256, 128], [1024, 1024, 512, 256], interaction='dot')
        m([dn, idf2])
        params = sum(v.numel() for v in tr.named_weights(m).values())
        out = {"batch": Bf2, "parameters": params}
        ms = step_time(m, [dn, idf2], yf, steps=4, sparse=True)
        out["lazy_rowwise_adam"] = {"ms_per_step": round(ms, 3), "samples_per_s": round(Bf2 / ms * 1e3, 1)}
        print("lazy", out, flush=True)
        torch.cuda.empty_cache()
        ms = step_time(m, [dn, idf2], yf, steps=3)
        out["exact_dense_adam"] = {"ms_per_step": round(ms, 3), "samples_per_s": round(Bf2 / ms * 1e3, 1),
                                   "GB_touched_per_step": round(params * 28 / 1e9, 1)}
        out["peak_device_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
        print(json.dumps({"DLRM dot 26x1Mx128, batch 65536 (configs[1] size)": out}))
        return
    m = DLRM([densec, sparse128], [512, 256, 128], [1024, 512, 256], interaction='dot')
    m([dense, ids]); rec("DLRM dot 26x100kx128", m, [dense, ids], y, B, True)
    m = DeepFM([densec, sparse128], (256, 128, 64))
    m([dense, ids]); rec("DeepFM 26x100kx128", m, [dense, ids], y, B, True)
    m = DCN(sparse128, [256, 128, 64])
    m(ids); rec("DCN 26x100kx128", m, ids, y, B, True)
    m = WideDeep([densec, sparse128], [256, 128, 64])
    m([dense, ids]); rec("Wide&Deep 26x100kx128", m, [dense, ids], y, B)
    m = AutoInt([densec, sparse16], att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
    Ba = 4096
    m([dense[:Ba], ids[:Ba]]); rec("AutoInt 39x16, 3 layers, 2 heads (configs[2] shape)", m, [dense[:Ba], ids[:Ba]], y[:Ba], Ba)
    vocab = [4000] * F
    m = FM([densec, [{'feat': f'C{i}', 'feat_num': v, 'embed_dim': 8} for i, v in enumerate(vocab)]], k=10)
    Bf = 512
    idf = rng.integers(0, 4000, size=(Bf, F)).astype(np.int32)
    rec("classic FM k=10, 104k features (src/ctr/fm/train.py: batch 512)", m, [dense[:Bf], idf], y[:Bf], Bf)
    T, Dd, Bd = 100, 64, 1024
    ukeys, ikeys = ['user_sparse_0'], ['item_sparse_0', 'item_sparse_1', 'item_sparse_2']
    sfd = {k: (V, Dd) for k in ukeys + ikeys}
    idx = [{k: i for i, k in enumerate(ukeys)}, {k: i for i, k in enumerate(ikeys)},
           {f'item_sparse_{ml}_{i}': ml * 3 + i for ml in range(T) for i in range(3)}]
    m = DIN(sfd, idx, ffn_hidden_units=(256, 128, 64), att_activation='sigmoid', ffn_activation='prelu', maxlen=T, dnn_dropout=0.5)
    beh = rng.integers(1, V, size=(Bd, T, 3)).astype(np.float32)
    for b in range(Bd):
        beh[b, :rng.integers(0, T)] = 0
    inp = [rng.random((Bd, 5)).astype(np.float32), rng.integers(0, V, size=(Bd, 1)).astype(np.float32),
           rng.random((Bd, 5)).astype(np.float32), rng.integers(0, V, size=(Bd, 3)).astype(np.float32), beh.reshape(Bd, -1)]
    m(inp); rec("DIN T=100, d=192, PReLU, dropout 0.5 (configs[3] shape)", m, inp, y[:Bd], Bd)
    S, n_neg, Bs, d = 200, 100, 512, 64
    cols = [{'feat': k, 'feat_num': V, 'feat_len': n, 'embed_dim': d} for k, n in (('seq_item', S), ('pos_item', 1), ('neg_item', n_neg))]
    m = SASRec(cols, [], att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n_neg)
    seq = rng.integers(1, V, size=(Bs, S)).astype(np.int32)
    for b in range(Bs):
        seq[b, :rng.integers(0, S)] = 0
    sin = [seq, rng.integers(1, V, size=(Bs, 1)).astype(np.int32), rng.integers(1, V, size=(Bs, n_neg)).astype(np.int32)]
    m(sin); rec("SASRec S=200, d=64, 100 negatives (configs[4] shape, 100k-row tables)", m, sin, None, Bs)
    m = NCF({'feat': 'u', 'feat_num': V, 'embed_dim': 32}, {'feat': 'i', 'feat_num': V, 'embed_dim': 32}, neg_num=10)
    Bn = 4096
    nin = [rng.integers(0, V, size=(Bn, 1)).astype(np.int32), rng.integers(0, V, size=(Bn, 1)).astype(np.int32),
           rng.integers(0, V, size=(Bn, 10)).astype(np.int32)]
    m(nin); rec("NCF dim 32, 10 negatives, dropout 0.2", m, nin, None, Bn)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
