"""Secondary measurements (not the headline bench): BASELINE configs[2..4] forward at full size on
one MI355X, per-kernel time from HIP events, with the roofline each one is bounded by.

  config 3  AutoInt 39 fields x dim 16, 3 layers, 2 heads (S=16), use_res, batch 4096   (fp32 MFMA bound)
  config 4  DIN history pooling, T = 100, d = 192 (3 tables x 64), batch 8192           (HBM bound)
  config 5  SASRec S = 200, d = 64, 1 block, neg 100, batch 8192 on ONE GPU (tables 10M x 64 x 3) (MFMA bound)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
from recamd import ops  # noqa: E402

dev = torch.device("cuda:0")
F32_MFMA_PEAK = 157.3  # TFLOP/s
HBM_PEAK = 8000.0      # GB/s


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def config3():
    from ctr.autoint.model import AutoInt
    B, F, nd, D = 4096, 26, 13, 16
    fc = [[{'feat': f'I{i}'} for i in range(nd)], [{'feat': f'C{i}', 'feat_num': 100_000, 'embed_dim': D} for i in range(F)]]
    m = AutoInt(fc, att_hidden_units=16, head_num=2, att_layer_num=3, use_res=True)
    dense = torch.rand((B, nd), device=dev)
    ids = torch.randint(0, 100_000, (B, F), device=dev, dtype=torch.int32)
    ms = timeit(lambda: m([dense, ids]))
    # the forward is ~12 short launches: capture it once into a HIP graph and replay
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m([dense, ids])
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        m([dense, ids])
    g_ms = timeit(graph.replay)
    x = torch.rand((B, 39, 16), device=dev)
    L = m.attention_layers[0]._w
    k_ms = timeit(lambda: ops.mha_ctr(x, x, x, L['Wq'], L['Wk'], L['Wv'], L['W0'], 2, 16, 'relu'))
    flop_l1 = 354_432  # layer 1 per sample
    return {"config": "AutoInt 39x16, 3 layers H=2 S=16, B=4096", "forward_ms": round(ms, 4),
            "samples_per_s": round(B / ms * 1e3, 1), "forward_hipgraph_ms": round(g_ms, 4),
            "hipgraph_samples_per_s": round(B / g_ms * 1e3, 1), "flop_per_sample": 1_382_784,
            "forward_tflops": round(B * 1_382_784 / ms / 1e9, 3),
            "mha_ctr_layer1_ms": round(k_ms, 4), "mha_ctr_layer1_tflops": round(B * flop_l1 / k_ms / 1e9, 3),
            "bound": "mfma_f32", "peak_tflops": F32_MFMA_PEAK,
            "frac_of_mfma_peak": round(B * 1_382_784 / ms / 1e9 / F32_MFMA_PEAK, 5)}


def config4():
    B, T, d = 8192, 100, 192
    tabs = [torch.empty((1_000_000, 64), device=dev).uniform_(-0.05, 0.05) for _ in range(3)]
    g = ops.TableGroup(tabs)
    lens = torch.randint(1, T + 1, (B,), device=dev)
    ids = torch.randint(1, 1_000_000, (B, T, 3), device=dev, dtype=torch.int32)
    pad = torch.arange(T, device=dev)[None, :] < (T - lens)[:, None]
    ids[pad] = 0
    mask = (ids[:, :, 0] != 0).float()
    q = torch.rand((B, d), device=dev)
    W = torch.randn((4 * d, 1), device=dev) * 0.05
    b = torch.zeros(1, device=dev)
    hist = torch.empty((B * T, d), device=dev)
    g_ms = timeit(lambda: ops.gather_concat(g, ids.view(B * T, 3), out=hist))
    hv = hist.view(B, T, d)
    p_ms = timeit(lambda: ops.din_attention_pool(q, hv, hv, mask, W, b, 'sigmoid'))
    f_ms = timeit(lambda: ops.gather_din_attention_pool(q, g, ids, None, W, b, 'sigmoid', mask_from_ids=True))
    fused_bytes = B * (T * d * 4 + T * 3 * 4 + 2 * d * 4)
    gather_bytes = B * T * 3 * (2 * 64 * 4 + 4)
    pool_bytes = B * (T * d * 4 + T * 4 + 2 * d * 4)
    return {"config": "DIN pooling T=100 d=192 B=8192", "history_gather_ms": round(g_ms, 4),
            "history_gather_GBs": round(gather_bytes / g_ms / 1e6, 1), "pool_ms": round(p_ms, 4),
            "pool_GBs": round(pool_bytes / p_ms / 1e6, 1), "bound": "hbm", "peak_GBs": HBM_PEAK,
            "fused_gather_pool_ms": round(f_ms, 4), "fused_GBs": round(fused_bytes / f_ms / 1e6, 1),
            "fused_frac": round(fused_bytes / f_ms / 1e6 / HBM_PEAK, 4),
            "fused_samples_per_s": round(B / f_ms * 1e3, 1),
            "pool_frac": round(pool_bytes / p_ms / 1e6 / HBM_PEAK, 4),
            "gather_frac": round(gather_bytes / g_ms / 1e6 / HBM_PEAK, 4)}


def config5():
    from match.sasrec.model import SASRec
    B, S, n, V, d = 8192, 200, 100, 10_000_000, 64
    uf = [{'feat': 'seq_item', 'feat_num': V, 'feat_len': S, 'embed_dim': d},
          {'feat': 'pos_item', 'feat_num': V, 'feat_len': 1, 'embed_dim': d},
          {'feat': 'neg_item', 'feat_num': V, 'feat_len': n, 'embed_dim': d}]
    res = {}
    for last in (True, False):
        m = SASRec(uf, [], blocks=1, num_heads=1, att_hidden_unit=d, ffn_hidden_unit=128, seq_len=S, neg_len=n,
                   last_row_only=last)
        lens = torch.randint(1, S + 1, (B,), device=dev)
        seq = torch.randint(1, V, (B, S), device=dev, dtype=torch.int32)
        seq[torch.arange(S, device=dev)[None, :] < (S - lens)[:, None]] = 0
        pos = torch.randint(1, V, (B, 1), device=dev, dtype=torch.int32)
        neg = torch.randint(1, V, (B, n), device=dev, dtype=torch.int32)
        ms = timeit(lambda: m([seq, pos, neg]), iters=10, warm=2)
        entry = {"forward_ms": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1)}
        if last:  # a few hundred microseconds of ~15 short launches: replay it from a HIP graph as well
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    m([seq, pos, neg])
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                m([seq, pos, neg])
            g_ms = timeit(graph.replay, iters=20, warm=3)
            entry.update({"forward_hipgraph_ms": round(g_ms, 4), "hipgraph_samples_per_s": round(B / g_ms * 1e3, 1)})
            del graph
        res["last_row_only" if last else "full_block"] = entry
        del m
        torch.cuda.empty_cache()
    q = torch.rand((B, S, d), device=dev)
    mask = torch.ones((B, S), device=dev)
    a_ms = timeit(lambda: ops.mha_rowmask(q, q, q, mask, 1), iters=5, warm=1)
    att_flop = B * 2 * (S * S * d * 2)
    res.update({"config": "SASRec S=200 d=64 1 block neg=100 B=8192 (one GPU)", "mha_rowmask_full_ms": round(a_ms, 3),
                "mha_rowmask_tflops": round(att_flop / a_ms / 1e9, 2), "bound": "mfma_f32", "peak_tflops": F32_MFMA_PEAK,
                "mha_frac_of_mfma_peak": round(att_flop / a_ms / 1e9 / F32_MFMA_PEAK, 4)})
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["3", "4", "5"]
    out = {}
    if "3" in which:
        out["config3"] = config3()
    if "4" in which:
        out["config4"] = config4()
    if "5" in which:
        out["config5"] = config5()
    print(json.dumps(out, indent=1))
