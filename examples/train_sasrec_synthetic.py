"""SASRec (src/match/sasrec/model.py) trained on its own add_loss, then scored with the one-launch forward kernel and an
exact inner-product top-k over the item table (the faiss step of the match train scripts) — synthetic sessions in which
the next item follows the last one, so that there is something to learn.

    python examples/train_sasrec_synthetic.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]

from match.sasrec.model import SASRec                     # noqa: E402
from match.utils.data_process import pad_sequences        # noqa: E402
from recamd import train as tr                            # noqa: E402
from recamd.retrieval import IndexFlatIP                  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    V, S, n_neg, d, n = 2000, 50, 20, 64, 8192
    starts = rng.integers(1, V, size=n)
    lens = rng.integers(2, S + 1, size=n)
    ragged = [((starts[i] + np.arange(lens[i])) % (V - 1) + 1).astype(np.int32) for i in range(n)]   # item t+1 follows item t
    seq = pad_sequences(ragged, maxlen=S)                                        # pre-padding with id 0 (Keras default)
    pos = ((seq[:, -1] % (V - 1)) + 1).astype(np.int32)[:, None]                 # the successor of the last item
    neg = rng.integers(1, V, size=(n, n_neg)).astype(np.int32)
    cols = [{'feat': k, 'feat_num': V, 'feat_len': ln, 'embed_dim': d} for k, ln in
            (('seq_item', S), ('pos_item', 1), ('neg_item', n_neg))]
    model = SASRec(cols, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n_neg)
    x = [np.asarray(seq, np.int32), pos, neg]
    trainer = tr.Trainer(model).compile(learning_rate=2e-3)
    hist = trainer.fit(x, None, batch_size=1024, epochs=6, validation_split=0.1, verbose=1)
    assert hist["loss"][-1] < hist["loss"][0]
    # retrieval: score every item of the POSITIVE table against the sequence state of the first 256 sessions
    model([a[:256] for a in x])
    index = IndexFlatIP(d)
    index.add(model.user_embed_layers['embed_pos_item'].table)
    _, top = index.search(model.embed[:, 0, :].contiguous(), 10)
    hit = float(np.mean([pos[i, 0] in top[i] for i in range(256)]))
    print(f"loss {hist['loss'][0]:.4f} -> {hist['loss'][-1]:.4f}; hit@10 of the true successor over {V} items: {hit:.3f}")
    return hit


if __name__ == "__main__":
    main()
