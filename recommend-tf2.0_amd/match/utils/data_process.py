"""Mirror of the path-facing part of src/match/utils/data_process.py: the padding step of `create_sasrec_dataset`
(:136-141) — `pad_sequences(hist, maxlen=maxlen)` — on the device.  The MovieLens loading, negative sampling
(unseeded `random.randint`, :100-105) and the train/val/test bookkeeping are host-side ETL outside the path."""
import numpy as np


def pad_sequences(sequences, maxlen, padding='pre', truncating='pre', value=0, device=None):
    """tf.keras.preprocessing.sequence.pad_sequences(sequences, maxlen) -> (n, maxlen) int32 numpy array; the rows are
    padded / truncated by the rec_pad_sequences_i32 kernel (pre-padding with 0 and keeping the LAST maxlen items by
    default: the most recent item ends up in the last slot, which SASRec reads at src/match/sasrec/model.py:88)."""
    import torch

    from recamd import pipeline
    dev = device or torch.device("cuda", torch.cuda.current_device())
    v, o = pipeline.ragged(sequences)
    out = pipeline.pad_sequences(torch.from_numpy(v).to(dev), torch.from_numpy(o).to(dev), maxlen, padding, truncating, value)
    return out.cpu().numpy()


def sasrec_inputs(hist, pos_id, neg_id, maxlen):
    """[pad_sequences(hist, maxlen), pos, neg] as create_sasrec_dataset assembles its train / val / test lists"""
    return [pad_sequences(hist, maxlen), np.asarray(pos_id), np.asarray(neg_id)]
