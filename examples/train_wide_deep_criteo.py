"""src/ctr/wide_deep/train.py, end to end on this stack, with a synthetic Criteo-format frame standing in for the CSV
(there is no dataset here): create_criteo_dataset (label-encode + min-max on the device) -> WideDeep -> compile(BCE, Adam,
AUC) -> fit(EarlyStopping(patience=1, restore_best_weights=True), validation_split=0.1) -> evaluate.

(The same script runs DeepFM by swapping the class — but DeepFM's FM layer, as the reference wrote it, adds ONE first-order
scalar summed over the whole batch to every sample (src/ctr/layers/modules.py:65): at batch 4096 that saturates the
sigmoid from the first step, here exactly as it would there.)

    python examples/train_wide_deep_criteo.py         # needs an MI355X and the in-tree build (__graft_entry__.build())
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")]

from ctr.wide_deep.model import WideDeep                  # noqa: E402  (same import path as the reference)
from ctr.utils.data_process import create_criteo_dataset  # noqa: E402
from recamd import train as tr                            # noqa: E402


def synthetic_criteo(n=20000, seed=0):
    """label, I1..I13 (counts with missing values), C1..C26 (8-hex-digit tokens with missing values); the label depends on
    a few columns so that there is something to learn"""
    rng = np.random.default_rng(seed)
    df = {}
    dense = rng.gamma(1.0, 30.0, size=(n, 13)).round()
    dense[rng.random((n, 13)) < 0.1] = np.nan
    cats = rng.zipf(1.3, size=(n, 26)) % np.array([50 + 37 * i for i in range(26)])
    score = 0.02 * np.nan_to_num(dense[:, 0]) - 0.8 * (cats[:, 0] % 3 == 0) + 0.9 * (cats[:, 1] % 2) - 1.0
    df["label"] = (rng.random(n) < 1 / (1 + np.exp(-score))).astype(int)
    for i in range(13):
        df[f"I{i + 1}"] = dense[:, i]
    for i in range(26):
        col = np.array([f"{(int(v) * 2654435761) & 0xffffffff:08x}" for v in cats[:, i]], dtype=object)
        col[rng.random(n) < 0.05] = np.nan
        df[f"C{i + 1}"] = col
    return pd.DataFrame(df)


def main():
    embed_dim, dnn_dropout, hidden_units = 8, 0.5, [256, 128, 64]          # src/ctr/wide_deep/train.py:29-32
    learning_rate, batch_size, epochs = 0.001, 4096, 5
    feature_columns, (train_X, train_y), (test_X, test_y) = create_criteo_dataset(
        synthetic_criteo(), embed_dim=embed_dim, read_part=False, test_size=0.2)
    model = WideDeep(feature_columns, hidden_units=hidden_units, dnn_dropout=dnn_dropout)
    model([train_X[0][:8], train_X[1][:8]])               # builds the lazily created layers
    model.summary()
    trainer = tr.Trainer(model).compile(learning_rate=learning_rate)
    hist = trainer.fit(train_X, train_y, batch_size=batch_size, epochs=epochs, validation_split=0.1, verbose=1,
                       callbacks=[tr.EarlyStopping(monitor="val_loss", patience=1, restore_best_weights=True)])
    loss, auc = trainer.evaluate(test_X, test_y, batch_size=batch_size)
    print(f"test loss {loss:.4f}  test AUC {auc:.4f}")
    assert auc > 0.55 and hist["loss"][-1] < hist["loss"][0]
    return auc


if __name__ == "__main__":
    main()
