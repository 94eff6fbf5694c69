"""The C ABI only enqueues on the given stream (no allocation, no sync), so whole forwards can be
captured into a HIP graph and replayed (launch-bound models such as AutoInt at batch 4096)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):  # warm-up on the side stream (lazy builds, workspace allocation)
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def test_autoint_forward_graph_replay(dev):
    from ctr.autoint.model import AutoInt
    rng = np.random.default_rng(0)
    vocabs = [int(v) for v in rng.integers(10, 1000, size=26)]
    fc = [[{'feat': f'I{i}'} for i in range(13)], [{'feat': f'C{i}', 'feat_num': v, 'embed_dim': 16} for i, v in enumerate(vocabs)]]
    m = AutoInt(fc, att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
    B = 512
    dense = torch.rand((B, 13), device=dev)
    ids = torch.stack([torch.randint(0, v, (B,), device=dev, dtype=torch.int32) for v in vocabs], dim=1)
    eager = m([dense, ids]).clone()
    g, out = _capture(lambda: m([dense, ids]))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    # new inputs through the same static buffers
    dense.copy_(torch.rand((B, 13), device=dev))
    ids.copy_(torch.stack([torch.randint(0, v, (B,), device=dev, dtype=torch.int32) for v in vocabs], dim=1))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, m([dense, ids]))


def test_dlrm_dot_forward_graph_replay(dev):
    from ctr.dlrm.model import DLRM
    rng = np.random.default_rng(1)
    vocabs = [int(v) for v in rng.integers(10, 500, size=26)]
    fc = [[{'feat': f'I{i}'} for i in range(13)], [{'feat': f'C{i}', 'feat_num': v, 'embed_dim': 128} for i, v in enumerate(vocabs)]]
    m = DLRM(fc, bot_dnn_hidden_units=[64, 128], top_dnn_hidden_units=[64, 32], interaction='dot')
    B = 300
    dense = torch.rand((B, 13), device=dev)
    ids = torch.stack([torch.randint(0, v, (B,), device=dev, dtype=torch.int32) for v in vocabs], dim=1)
    eager = m([dense, ids]).clone()
    g, out = _capture(lambda: m([dense, ids]))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
