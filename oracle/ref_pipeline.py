"""CPU restatement of the reference's input pipeline for the path (TEST INFRASTRUCTURE ONLY).

  label_encode_fit / label_encode_transform   sklearn.preprocessing.LabelEncoder as used at
        src/ctr/utils/data_process.py:66-68 (`le.fit_transform(data_df[feat].astype(str))` after `fillna('-1')`, :63):
        classes = sorted unique STRINGS, id = index of the string in that sorted list.
  minmax_fit / minmax_transform                sklearn MinMaxScaler on `astype(int)` values (:76-78), in the intended
        per-column form (as written the (n, 13) result is assigned to one column, which pandas rejects — SURVEY §2.1).
  pad_sequences                                tf.keras.preprocessing.sequence.pad_sequences(seqs, maxlen) defaults
        (padding='pre', truncating='pre', value=0, dtype int32) as used at src/match/utils/data_process.py:138.

PINNED: tests/test_pipeline_cpu.py checks the first two against the real scikit-learn (importable here).  pad_sequences
is restated from the Keras documentation (Keras is not importable: parity unpinned for that one function)."""
import numpy as np

MISSING = np.uint32(0xFFFFFFFF)


def hex_tokens(col):
    """Criteo categorical column (8-digit lowercase hex strings, NaN / '' / None for missing) -> uint32 tokens;
    missing -> MISSING (the '-1' of fillna('-1'))."""
    out = np.empty(len(col), np.uint32)
    for i, s in enumerate(col):
        if s is None or (isinstance(s, float) and np.isnan(s)) or s == "" or s == "-1":
            out[i] = MISSING
        else:
            out[i] = np.uint32(int(s, 16))
    return out


def token_strings(tok):
    """the strings the reference would see for these tokens"""
    return np.array(["-1" if t == MISSING else f"{int(t):08x}" for t in tok])


def label_encode_fit(tok):
    """sorted unique in STRING order: '-1' first, then the hex strings (fixed width -> numeric order)"""
    strs = np.unique(token_strings(tok))                       # np.unique sorts lexicographically, like sklearn
    return np.array([MISSING if s == "-1" else np.uint32(int(s, 16)) for s in strs], np.uint32)


def label_encode_transform(vocab, tok):
    lut = {int(v): i for i, v in enumerate(vocab)}
    return np.array([lut.get(int(t), -1) for t in tok], np.int32)


def minmax_fit(x):
    xi = np.trunc(np.asarray(x, np.float64))                   # astype(int) truncates toward zero
    return xi.min(axis=0), xi.max(axis=0)


def minmax_transform(x, mn, mx):
    xi = np.trunc(np.asarray(x, np.float64))
    rng = np.asarray(mx, np.float64) - np.asarray(mn, np.float64)
    scale = np.where(rng == 0, 1.0, 1.0 / np.where(rng == 0, 1.0, rng))
    return (xi * scale + (0.0 - np.asarray(mn, np.float64) * scale)).astype(np.float32)


def pad_sequences(seqs, maxlen, padding="pre", truncating="pre", value=0):
    out = np.full((len(seqs), maxlen), value, np.int32)
    for i, s in enumerate(seqs):
        s = list(s)
        if len(s) == 0:
            continue
        t = s[-maxlen:] if truncating == "pre" else s[:maxlen]
        if padding == "pre":
            out[i, maxlen - len(t):] = t
        else:
            out[i, :len(t)] = t
    return out
