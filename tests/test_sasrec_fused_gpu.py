"""rec_sasrec_last_row_f32 (SASRec, one block / one head / last position, in ONE launch) against the oracle's full
sasrec_forward (src/match/sasrec/model.py:60-97 restated), and against the layer-by-layer path of the mirror."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def make_case(rng, B, S, n_neg, V, fh, scale=0.15, lens=None, n_pos=1):
    d = 64
    P = dict(Wq=rng.normal(size=(d, d)) * scale, bq=rng.normal(size=d) * scale, Wk=rng.normal(size=(d, d)) * scale,
             bk=rng.normal(size=d) * scale, Wv=rng.normal(size=(d, d)) * scale, bv=rng.normal(size=d) * scale,
             W1=rng.normal(size=(d, fh)) * scale, b1=rng.normal(size=fh) * scale, W2=rng.normal(size=(fh, d)) * scale,
             b2=rng.normal(size=d) * scale, ln1_g=1 + 0.1 * rng.normal(size=d), ln1_b=0.1 * rng.normal(size=d),
             ln2_g=1 + 0.1 * rng.normal(size=d), ln2_b=0.1 * rng.normal(size=d))
    P = {k: v.astype(np.float32) for k, v in P.items()}
    T = [rng.normal(size=(V, d)).astype(np.float32) * 0.5 for _ in range(3)]
    if lens is None:
        lens = rng.integers(0, S + 1, size=B)
    seq = rng.integers(1, V, size=(B, S))
    seq[np.arange(S)[None, :] < (S - lens)[:, None]] = 0          # pre-padding
    pos = rng.integers(0, V, size=(B, n_pos)).astype(np.int32)     # id 0 is a valid candidate row
    neg = rng.integers(0, V, size=(B, n_neg)).astype(np.int32)
    return P, T, seq.astype(np.int32), pos, neg


def weights_of(P, dev):
    order = ["Wq", "bq", "Wk", "Wv", "bv", "ln1_g", "ln1_b", "W1", "b1", "W2", "b2", "ln2_g", "ln2_b"]
    return [torch.from_numpy(np.ascontiguousarray(P[k])).to(dev) for k in order]


def run(dev, P, T, seq, pos, neg, pad_id=0, eps=1e-6, flag=None):
    from recamd import ops
    tt = [torch.from_numpy(t).to(dev) for t in T]
    ts, tp, tn = (torch.from_numpy(a).to(dev) for a in (seq, pos, neg))
    logits, si = ops.sasrec_last_row(weights_of(P, dev), eps, eps, tt[0], ts, pad_id, ts[:, -1], ts.stride(0), tt[1], tp,
                                     tt[2], tn, oob_flag=flag)
    torch.cuda.synchronize()
    return logits.cpu().numpy(), si.cpu().numpy()


@pytest.mark.parametrize("B,S,n_neg,fh", [(40, 20, 100, 128), (7, 1, 1, 64), (33, 5, 31, 128), (19, 64, 32, 64),
                                          (21, 65, 33, 128), (16, 200, 100, 128), (5, 256, 128, 128), (9, 320, 130, 64), (130, 37, 63, 64)])
def test_matches_the_oracle(dev, B, S, n_neg, fh):
    rng = np.random.default_rng(B * 1000 + S)
    P, T, seq, pos, neg = make_case(rng, B, S, n_neg, 300, fh)
    seq[0] = 0                                    # an all-padding sequence: logits exactly 0
    if B > 2:
        seq[1] = rng.integers(1, 300, size=S)     # a full sequence
    logits, si = run(dev, P, T, seq, pos, neg)
    exp, _ = ref.sasrec_forward(seq, pos, neg, T[0], T[1], T[2], [P], 1)
    assert close(logits, exp, 1e-5)
    assert np.all(logits[0] == 0.0) and np.all(si[0] == 0.0)


def test_pads_in_the_middle_and_masked_last_position(dev):
    """Not pre-padded: pad ids anywhere (zero rows that still take part in the softmax), and a padded LAST position
    (query masked -> uniform attention; output row multiplied by 0)."""
    rng = np.random.default_rng(5)
    B, S, n = 24, 50, 20
    P, T, seq, pos, neg = make_case(rng, B, S, n, 100, 128, lens=np.full(24, 50))
    seq[rng.random(size=seq.shape) < 0.3] = 0
    seq[:6, -1] = 0
    seq[6:, -1] = 7
    logits, si = run(dev, P, T, seq, pos, neg)
    exp, _ = ref.sasrec_forward(seq, pos, neg, T[0], T[1], T[2], [P], 1)
    assert close(logits, exp, 1e-5)
    assert np.all(logits[:6] == 0.0)


def test_out_of_range_ids_are_zero_rows_and_flagged(dev):
    rng = np.random.default_rng(6)
    B, S, n, V = 12, 30, 10, 80
    P, T, seq, pos, neg = make_case(rng, B, S, n, V, 128, lens=np.full(12, 30))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    run(dev, P, T, seq, pos, neg, flag=flag)
    assert int(flag.item()) == 0
    seq2, neg2 = seq.copy(), neg.copy()
    seq2[3, 4], seq2[5, 9] = V + 5, -3
    neg2[2, 1] = V
    logits, _ = run(dev, P, T, seq2, pos, neg2, flag=flag)
    assert int(flag.item()) == 1
    # the oracle's embedding_lookup answers out-of-range ids with zero rows too (TF-GPU GatherV2 semantics)
    exp, _ = ref.sasrec_forward(seq2, pos, neg2, T[0], T[1], T[2], [P], 1)
    assert close(logits, exp, 1e-5)
    assert logits[2, 2] == 0.0


def test_strided_ids_and_pad_id_minus_one(dev):
    """The row-sharded caller: ids are views into a longer index vector, pads arrive as -1 (not 0), the mask comes
    from the ORIGINAL sequence ids."""
    from recamd import ops
    rng = np.random.default_rng(7)
    B, S, n, V = 17, 23, 9, 60
    P, T, seq, pos, neg = make_case(rng, B, S, n, V, 64)
    exp, _ = ref.sasrec_forward(seq, pos, neg, T[0], T[1], T[2], [P], 1)
    tt = [torch.from_numpy(t).to(dev) for t in T]
    wide = torch.full((B, S + 11), 12345, dtype=torch.int32, device=dev)
    seq_m = np.where(seq == 0, -1, seq).astype(np.int32)
    wide[:, 3:3 + S] = torch.from_numpy(seq_m).to(dev)
    orig = torch.from_numpy(seq).to(dev)
    tp, tn = torch.from_numpy(pos).to(dev), torch.from_numpy(neg).to(dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    logits, _ = ops.sasrec_last_row(weights_of(P, dev), 1e-6, 1e-6, tt[0], wide[:, 3:3 + S], -1, orig[:, -1], orig.stride(0),
                                    tt[1], tp, tt[2], tn, oob_flag=flag)
    assert close(logits.cpu().numpy(), exp, 1e-5)
    assert int(flag.item()) == 0                   # pad_id rows are not out-of-range rows


def test_model_paths_agree(dev, monkeypatch):
    """The mirror's one-launch path equals its layer-by-layer path (`fused = False`) on the same weights."""
    from match.sasrec.model import SASRec
    rng = np.random.default_rng(8)
    V, S, n, B = 500, 200, 100, 300
    uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': 64},
          {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': 64},
          {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': 64}]
    m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=64, ffn_hidden_unit=128, seq_len=S, neg_len=n)
    lens = rng.integers(1, S + 1, size=B)
    seq = rng.integers(1, V, size=(B, S))
    seq[np.arange(S)[None, :] < (S - lens)[:, None]] = 0
    seq = seq.astype(np.int32)
    pos = rng.integers(1, V, size=(B, 1)).astype(np.int32)
    neg = rng.integers(1, V, size=(B, n)).astype(np.int32)
    a = m([seq, pos, neg]).cpu().numpy()
    emb_a = m.embed.cpu().numpy()
    m.fused = False
    b = m([seq, pos, neg]).cpu().numpy()
    assert close(a, b, 1e-5) and close(emb_a, m.embed.cpu().numpy(), 1e-5)


def test_rejects_unsupported_shapes(dev):
    from recamd import ops
    rng = np.random.default_rng(9)
    P, T, seq, pos, neg = make_case(rng, 4, 6, 3, 20, 128)
    P["W1"] = np.zeros((64, 192), np.float32)
    P["b1"] = np.zeros(192, np.float32)
    P["W2"] = np.zeros((192, 64), np.float32)
    with pytest.raises(RuntimeError, match="ffn_hidden"):
        run(dev, P, T, seq, pos, neg)
    assert not ops.sasrec_last_row_supported(32, 128, 10, 5) and not ops.sasrec_last_row_supported(64, 128, 2000, 5)
    assert ops.sasrec_last_row_supported(64, 128, 200, 101)
