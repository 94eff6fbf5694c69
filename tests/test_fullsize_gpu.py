"""BASELINE full-size checks (configs[1]: 65 536 x 26 x 1M x 128) through size-independent properties,
plus oracle checks on sampled rows (the fp64 oracle cannot afford the whole batch in seconds).

  gather      : bit-exact vs an independent device gather (torch indexing) on the WHOLE output;
                checksum-of-checksums; sampled rows vs the numpy oracle.
  fused dot   : sampled samples vs the fp64 oracle; bilinearity (scaling the dense vector scales exactly
                its 26 dots by the same power of two); symmetry of the source (permuting two fields
                permutes dots); dense pass-through bit-exact; idempotence (two launches agree bit-for-bit).
"""
import numpy as np
import pytest
import torch

from oracle import ref_numpy as ref
from tests.util import RING_ORDER, close, close_dot, fmaf_chain_dot

pytestmark = pytest.mark.gpu

B, F, V, D = 65536, 26, 1_000_000, 128


@pytest.fixture(scope="module")
def world(dev):
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * 2 ** 30:
        pytest.skip("needs ~16 GB of HBM")
    from recamd import ops
    gen = torch.Generator(device=dev).manual_seed(0)
    arena = torch.empty((F, V, D), dtype=torch.float32, device=dev).uniform_(-0.05, 0.05, generator=gen)
    ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen)
    dense = torch.rand((B, D), device=dev, generator=gen)
    g = ops.TableGroup([arena[f] for f in range(F)])
    yield arena, ids, dense, g
    del arena


def test_full_gather_bit_exact_and_checksums(world, dev):
    from recamd import ops
    arena, ids, dense, g = world
    out = ops.gather_concat(g, ids)
    fidx = torch.arange(F, device=dev)[None, :].expand(B, F)
    exp = arena[fidx.reshape(-1), ids.long().reshape(-1)].view(B, F * D)   # independent device gather
    assert torch.equal(out.view(torch.int32), exp.view(torch.int32))
    # checksum of checksums over int32 views (order-independent, exact)
    a = out.view(torch.int32).to(torch.int64).sum(dim=1)
    b = exp.view(torch.int32).to(torch.int64).sum(dim=1)
    assert int(a.sum()) == int(b.sum()) and torch.equal(a, b)
    # sampled rows vs the numpy oracle
    rows = np.random.default_rng(1).integers(0, B, size=64)
    ids_h = ids[rows].cpu().numpy()
    tabs_h = [arena[f][torch.from_numpy(ids_h[:, f]).long().to(dev)].cpu().numpy() for f in range(F)]
    for r, row in enumerate(rows):
        want = np.concatenate([tabs_h[f][r] for f in range(F)])
        assert np.array_equal(out[row].cpu().numpy(), want)


def test_full_fused_dot_properties(world, dev):
    from recamd import ops
    arena, ids, dense, g = world
    P = 27 * 26 // 2
    out = ops.gather_pairwise_dot(g, ids, dense)
    out2 = ops.gather_pairwise_dot(g, ids, dense)
    assert torch.equal(out, out2)                                   # idempotent / deterministic
    assert torch.equal(out[:, P:], dense)                           # pass-through bit-exact
    # sampled samples vs the fp64 oracle
    rows = np.random.default_rng(2).integers(0, B, size=48)
    ids_h = ids[rows].cpu().numpy()
    emb = np.stack([arena[f][torch.from_numpy(ids_h[:, f]).long().to(dev)].cpu().numpy() for f in range(F)], axis=1)
    X = np.concatenate([emb, dense[rows].cpu().numpy()[:, None, :]], axis=1)
    got = out[rows][:, :P].cpu().numpy()
    assert close_dot(got, X)                                        # kernel-level tolerance on config-2 data
    assert np.array_equal(got.view(np.uint32), fmaf_chain_dot(X, RING_ORDER).view(np.uint32)) or \
        (got.view(np.uint32) == fmaf_chain_dot(X, RING_ORDER).view(np.uint32)).mean() > 0.9999   # the arithmetic, pinned
    # bilinearity: dense * 2 (exact in fp32) doubles exactly the dots (26, j) and leaves the rest
    out_s = ops.gather_pairwise_dot(g, ids, dense * 2.0)
    last = 26 * 25 // 2
    assert torch.equal(out_s[:, :last], out[:, :last])
    assert torch.equal(out_s[:, last:P], out[:, last:P] * 2.0)
    # swapping fields 0 and 1 (tables and id columns) leaves dot (1,0) unchanged and swaps (i,0) <-> (i,1)
    g_sw = ops.TableGroup([arena[1], arena[0]] + [arena[f] for f in range(2, F)])
    ids_sw = ids.clone()
    ids_sw[:, 0], ids_sw[:, 1] = ids[:, 1], ids[:, 0]
    out_sw = ops.gather_pairwise_dot(g_sw, ids_sw, dense)
    # (products commute exactly and the chain order is the same: bit-identical)
    assert torch.equal(out_sw[:, 0], out[:, 0])
    i = 5
    assert torch.equal(out_sw[:, i * (i - 1) // 2 + 0], out[:, i * (i - 1) // 2 + 1])
    # fused (LDS ring + fp32 MFMA, k-ordered chain) vs materialised gather + the register-tiled kernel (tree sum):
    # different summation orders, both within the kernel-level tolerance of the fp64 result
    Xd = torch.cat([ops.gather_concat(g, ids[:1024]).view(1024, F, D), dense[:1024, None, :]], dim=1).contiguous()
    assert close_dot(ops.pairwise_dot(Xd).cpu().numpy(), Xd.cpu().numpy())
    assert close_dot(out[:1024, :P].cpu().numpy(), Xd.cpu().numpy())


def test_full_dlrm_model_forward_sampled_vs_oracle(world, dev, force):
    """The whole DLRM model (src/ctr/dlrm/model.py:42-54, dot interaction) on the configs[1] tables at batch 65 536 — what
    bench.py's `model_forward` times: bottom MLP 13-512-256-128, fused gather + pairwise dot, top MLP 479-1024-1024-512-256-1.
    Its Dense layers run on the f16x2 kernel with row maxima handed along the towers.  Sampled samples vs the fp64 oracle
    (the oracle gathers only their rows), the forward is deterministic, and the bf16x3 kernels give the same logits to 1e-5."""
    from ctr.dlrm.model import DLRM
    from recamd import ops
    from tests.test_models_gpu import dense_cols, dnn_params, randomize
    arena, ids, dense128, g = world
    nd = 13
    rng = np.random.default_rng(7)
    m = DLRM([dense_cols(nd), [{'feat': f'C{i}', 'feat_num': 1, 'embed_dim': D} for i in range(F)]], [512, 256, D],
             [1024, 1024, 512, 256], interaction='dot')
    dense = torch.rand((B, nd), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    m([dense[:8], torch.zeros((8, F), dtype=torch.int32, device=dev)])        # lazy build on the one-row tables
    w = randomize(m, rng, 0.08)
    for f in range(F):                                                       # then the real tables, as bench.py does
        layer = m.embed_layers['embed_%d' % f]
        layer._w["embeddings"], layer.input_dim = arena[f], V
    m._group = ops.TableGroup([m.embed_layers['embed_%d' % f].table for f in range(F)])
    out = m([dense, ids])
    assert torch.equal(out, m([dense, ids]))
    rows = np.random.default_rng(8).integers(0, B, size=64)
    ids_h = ids[rows].cpu().numpy()
    # compact tables holding just the sampled rows (row j of table f = the row sample j looks up)
    tabs = [arena[f][torch.from_numpy(ids_h[:, f]).long().to(dev)].cpu().numpy() for f in range(F)]
    cid = np.tile(np.arange(len(rows), dtype=np.int32)[:, None], (1, F))
    exp = ref.dlrm_forward(dense[rows].cpu().numpy(), cid, tabs, dnn_params(w, 'bot_dnn', 3), dnn_params(w, 'top_dnn', 4),
                           (w['final_dense/kernel'], w['final_dense/bias']), 'dot')
    got = out[rows].cpu().numpy()
    assert close(got, exp)
    force("dense_pipe", "s")                                                 # the six-MFMA scheme on the same model
    alt = m([dense, ids])[rows].cpu().numpy()
    assert close(alt, exp) and np.abs(alt - got).max() <= 1e-5
