"""Model-level forward throughput on one MI355X at the BASELINE headline shape (65 536 x 26 sparse x dim 128), with
the paper-sized DLRM MLPs, for the default build and with rec_debug_force("dense", "f") (fp32-MFMA Dense) — i.e. what the
bf16x3 Dense buys end to end.  Tables use V = 200 000 rows per field to leave HBM for several models."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recommend-tf2.0_amd"))
from recamd._lib import C  # noqa: E402

dev = torch.device("cuda:0")
B, F, D, V, ND = 65536, 26, 128, 200_000, 13


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    from ctr.dcn.model import DCN
    from ctr.deep_crossing.model import Deep_Crossing
    from ctr.deep_fm.model import DeepFM
    from ctr.dlrm.model import DLRM
    from ctr.wide_deep.model import WideDeep
    sparse = [{'feat': f'C{i}', 'feat_num': V, 'embed_dim': D} for i in range(F)]
    dense_c = [{'feat': f'I{i}'} for i in range(ND)]
    gen = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen)
    dense = torch.rand((B, ND), device=dev, generator=gen)
    models = {
        "DLRM dot (bot 512-256-128, top 1024-1024-512-256)": (lambda: DLRM([dense_c, sparse], [512, 256, 128], [1024, 1024, 512, 256], interaction='dot'), True),
        "DLRM cat (reference form, same MLPs)": (lambda: DLRM([dense_c, sparse], [512, 256, 128], [1024, 1024, 512, 256], interaction='cat'), True),
        "DeepFM (256-128-64)": (lambda: DeepFM([dense_c, sparse], hidden_units=(256, 128, 64)), True),
        "DCN (256-128-64, 3 cross layers)": (lambda: DCN(sparse, hidden_units=(256, 128, 64)), False),
        "Wide&Deep (256-128-64)": (lambda: WideDeep([dense_c, sparse], hidden_units=(256, 128, 64)), True),
        "Deep&Crossing (256, 256)": (lambda: Deep_Crossing(sparse, hidden_units=(256, 256)), False),
    }
    res = {}
    for name, (mk, has_dense) in models.items():
        m = mk()
        x = [dense, ids] if has_dense else ids
        row = {}
        for impl in ("default", "f"):
            if impl == "f":
                C.debug_force("dense", "f")
            else:
                C.debug_force("dense", None)
            ms = timeit(lambda: m(x))
            row[impl] = {"forward_ms": round(ms, 3), "samples_per_s": round(B / ms * 1e3)}
        C.debug_force("dense", None)
        row["speedup_from_bf16x3_dense"] = round(row["f"]["forward_ms"] / row["default"]["forward_ms"], 2)
        res[name] = row
        print(json.dumps({name: row}), flush=True)
        del m
        torch.cuda.empty_cache()
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "models.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
