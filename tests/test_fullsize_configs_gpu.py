"""BASELINE configs[2..4] at their FULL sizes (AutoInt 4096 x 39 x 16, DIN 8192 x T 100 x d 192 over 1M-row tables, SASRec
8192 x S 200 x d 64 over 10M-row tables) — the sizes bench.py times.  The fp64 oracle cannot afford a whole batch in
seconds, so each test compares a random SUBSET of the samples with the oracle (the rows those samples touch are pulled
into compact tables and the ids remapped: the oracle then sees a small vocabulary holding exactly the same values) and
checks size-independent properties on the whole batch: idempotence (bit-identical relaunch), agreement of the one-launch
kernels with the layer-by-layer paths, invariance to the order of a DIN history's real slots, pad-only SASRec sequences
giving logits exactly 0."""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import close

pytestmark = pytest.mark.gpu


def need(dev, gib):
    free, _ = torch.cuda.mem_get_info()
    if free < gib * 2 ** 30:
        pytest.skip(f"needs ~{gib} GiB of HBM")


def compact(table: torch.Tensor, ids: np.ndarray):
    """(rows of `table` that `ids` touch, ids remapped into them); id 0 stays 0 (pad id) by keeping row 0 first"""
    uniq = np.unique(np.concatenate([[0], ids.reshape(-1)]))
    small = table[torch.from_numpy(uniq).long().to(table.device)].cpu().numpy()
    return small, np.searchsorted(uniq, ids).astype(ids.dtype)


def test_din_config4_full_size(dev):
    need(dev, 4)
    from recamd import ops
    B, T, V, Dt, n_tab = 8192, 100, 1_000_000, 64, 3
    d = n_tab * Dt
    gen = torch.Generator(device=dev).manual_seed(4)
    tabs = [torch.empty((V, Dt), device=dev).uniform_(-0.05, 0.05, generator=gen) for _ in range(n_tab)]
    g = ops.TableGroup(tabs)
    lens = torch.randint(1, T + 1, (B,), device=dev, generator=gen)
    ids = torch.randint(1, V, (B, T, n_tab), device=dev, dtype=torch.int32, generator=gen)
    ids[torch.arange(T, device=dev)[None, :] < (T - lens)[:, None]] = 0            # pre-padding (pad_sequences)
    q = torch.rand((B, d), device=dev, generator=gen)
    W = torch.randn((4 * d, 1), device=dev, generator=gen) * 0.05
    b = torch.full((1,), 0.1, device=dev)
    out = ops.gather_din_attention_pool(q, g, ids, None, W, b, 'sigmoid', mask_from_ids=True)
    assert torch.equal(out, ops.gather_din_attention_pool(q, g, ids, None, W, b, 'sigmoid', mask_from_ids=True))
    # the fused kernel against the two-step path (materialised history rows, then pooling) on a slice
    nb = 512
    beh = ops.gather_concat(g, ids[:nb].reshape(nb * T, n_tab)).view(nb, T, d)
    mask = (ids[:nb, :, 0] != 0).to(torch.float32).contiguous()
    two = ops.din_attention_pool(q[:nb].contiguous(), beh, beh, mask, W, b, 'sigmoid')
    assert close(out[:nb].cpu().numpy(), two.cpu().numpy())
    # subset vs the numpy oracle
    rows = np.random.default_rng(0).choice(B, size=96, replace=False)
    ids_h = ids[torch.from_numpy(rows).to(dev)].cpu().numpy()
    parts = []
    for t in range(n_tab):
        small, rid = compact(tabs[t], ids_h[:, :, t])
        parts.append(ref.embedding_lookup(small.astype(np.float64), rid))
    beh_h = np.concatenate(parts, axis=-1)
    mask_h = (ids_h[:, :, 0] != 0).astype(np.float64)
    exp = ref.din_attention_layer(q[torch.from_numpy(rows).to(dev)].cpu().numpy(), beh_h, beh_h, mask_h, W.cpu().numpy(),
                                  b.cpu().numpy(), 'sigmoid')
    assert close(out[torch.from_numpy(rows).to(dev)].cpu().numpy(), exp)
    # the pooling is a softmax-weighted sum over the REAL slots: reversing their order changes nothing but rounding
    ids_rev = ids.clone()
    for s in (0, 1, 2, 3):
        L = int(lens[s])
        ids_rev[s, T - L:] = torch.flip(ids[s, T - L:], dims=[0])
    out_rev = ops.gather_din_attention_pool(q[:4].contiguous(), g, ids_rev[:4].contiguous(), None, W, b, 'sigmoid', mask_from_ids=True)
    assert close(out_rev.cpu().numpy(), out[:4].cpu().numpy())
    # a convex combination of the history rows: inside their per-column range
    lo, hi = beh.amin(dim=1), beh.amax(dim=1)                      # pad rows are zero rows: only widen the range
    assert bool(((out[:nb] >= lo - 1e-6) & (out[:nb] <= hi + 1e-6)).all())


def test_sasrec_config5_full_size(dev, monkeypatch):
    need(dev, 12)
    from match.sasrec.model import SASRec
    B, S, n, V, d = 8192, 200, 100, 10_000_000, 64
    uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d},
          {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
          {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
    m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n)
    gen = torch.Generator(device=dev).manual_seed(5)
    lens = torch.randint(0, S + 1, (B,), device=dev, generator=gen)
    lens[0] = 0                                                         # an all-padding sequence
    seq = torch.randint(1, V, (B, S), device=dev, dtype=torch.int32, generator=gen)
    seq[torch.arange(S, device=dev)[None, :] < (S - lens)[:, None]] = 0
    pos = torch.randint(1, V, (B, 1), device=dev, dtype=torch.int32, generator=gen)
    neg = torch.randint(1, V, (B, n), device=dev, dtype=torch.int32, generator=gen)
    m([seq[:8], pos[:8], neg[:8]])                                       # builds the block
    rng = np.random.default_rng(5)
    w = {}
    for k, v in m.get_weights().items():
        if k.endswith('embeddings'):
            continue                                                   # the 7.7 GB of tables keep their initial values
        w[k] = (1 + 0.1 * rng.normal(size=v.shape) if k.endswith('gamma') else rng.normal(size=v.shape) * 0.15).astype(np.float32)
    m.set_weights(w)
    logits = m([seq, pos, neg])
    assert torch.equal(logits, m([seq, pos, neg]))                       # the one-launch kernel is deterministic
    assert bool((logits[0] == 0).all())                                  # all-padding sequence: logits exactly 0
    # the one-launch kernel against the layer-by-layer path on a slice
    m.fused = False
    layers = m([seq[:256], pos[:256], neg[:256]])
    m.fused = True
    assert close(logits[:256].cpu().numpy(), layers.cpu().numpy(), 1e-5)
    # subset vs the numpy oracle on compacted tables
    rows = np.concatenate([[0], np.random.default_rng(1).choice(np.arange(1, B), size=47, replace=False)])
    ridx = torch.from_numpy(rows).to(dev)
    tb = m.user_embed_layers
    t_seq, r_seq = compact(tb['embed_seq_item'].table, seq[ridx].cpu().numpy())
    t_pos, r_pos = compact(tb['embed_pos_item'].table, pos[ridx].cpu().numpy())
    t_neg, r_neg = compact(tb['embed_neg_item'].table, neg[ridx].cpu().numpy())
    ww = w
    e = 'encoder_0/'
    P = dict(Wq=ww[e + 'mha/wq/kernel'], bq=ww[e + 'mha/wq/bias'], Wk=ww[e + 'mha/wk/kernel'], bk=ww[e + 'mha/wk/bias'],
             Wv=ww[e + 'mha/wv/kernel'], bv=ww[e + 'mha/wv/bias'], ln1_g=ww[e + 'layernorm1/gamma'], ln1_b=ww[e + 'layernorm1/beta'],
             W1=ww[e + 'ffn/conv1/kernel'], b1=ww[e + 'ffn/conv1/bias'], W2=ww[e + 'ffn/conv2/kernel'], b2=ww[e + 'ffn/conv2/bias'],
             ln2_g=ww[e + 'layernorm2/gamma'], ln2_b=ww[e + 'layernorm2/beta'])
    exp, _ = ref.sasrec_forward(r_seq, r_pos, r_neg, t_seq, t_pos, t_neg, [P], 1)
    assert close(logits[ridx].cpu().numpy(), exp, 1e-5)


def test_autoint_config3_full_size(dev, monkeypatch):
    need(dev, 2)
    from ctr.autoint.model import AutoInt
    B, F, nd, D, V = 4096, 26, 13, 16, 100_000
    fc = [[{'feat': f'I{i}'} for i in range(nd)], [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': D} for i in range(F)]]
    m = AutoInt(fc, att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
    gen = torch.Generator(device=dev).manual_seed(3)
    dense = torch.rand((B, nd), device=dev, generator=gen)
    ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen)
    m([dense[:8], ids[:8]])
    rng = np.random.default_rng(3)
    w = {k: (rng.normal(size=v.shape) * 0.2).astype(np.float32) for k, v in m.get_weights().items() if not k.endswith('embeddings')}
    m.set_weights(w)
    out = m([dense, ids])
    assert torch.equal(out, m([dense, ids]))
    m.fused = False
    out_layers = m([dense[:512], ids[:512]])
    m.fused = True
    assert close(out[:512].cpu().numpy(), out_layers.cpu().numpy())
    rows = np.random.default_rng(2).choice(B, size=128, replace=False)
    ridx = torch.from_numpy(rows).to(dev)
    ids_h, dense_h = ids[ridx].cpu().numpy(), dense[ridx].cpu().numpy().astype(np.float64)
    emb = []
    for f in range(F):
        small, rid = compact(m.embed_layers[f'embed_{f}'].table, ids_h[:, f])
        emb.append(ref.embedding_lookup(small.astype(np.float64), rid))
    x3 = np.concatenate([np.stack(emb, axis=1), dense_h[:, :, None] * w['dense_embed'][None].astype(np.float64)], axis=1)
    layers = [dict(Wq=w[f'attention_{i}/Wq'], Wk=w[f'attention_{i}/Wk'], Wv=w[f'attention_{i}/Wv'], W0=w[f'attention_{i}/W0'])
              for i in range(3)]
    exp = ref.autoint_forward_intended(x3, layers, (w['final_dense/kernel'], w['final_dense/bias']), 2, 16, 'relu', True)
    assert close(out[ridx].cpu().numpy(), exp)
