#!/usr/bin/env python3
"""A/B timing of the fused gather + pairwise-dot kernels at BASELINE configs[1] (65 536 x 26 x 1M x 128):
the LDS-ring / fp32-MFMA kernel in its ring configurations (REC_RING_CFG) and the register-tiled kernel
(REC_PAIRDOT_IMPL=valu, in a child process because the choice is read once).  Rotates 8 id batches so no
launch re-reads the previous launch's rows; prints per-launch p10/p50/p90 from HIP events."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")):
    sys.path.insert(0, p)


def run(cfgs, steps=60):
    import torch
    from recamd import ops
    dev = torch.device("cuda:0")
    B, F, V, D = 65536, 26, 1_000_000, 128
    gen = torch.Generator(device=dev).manual_seed(0)
    arena = torch.empty((F, V, D), dtype=torch.float32, device=dev)
    arena.uniform_(-0.05, 0.05, generator=gen)
    group = ops.TableGroup([arena[f] for f in range(F)])
    ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
    dense = torch.rand((B, D), device=dev, generator=gen)
    out = torch.empty((B, 480), dtype=torch.float32, device=dev)[:, :479]
    res = {}
    # device spin-up: the first ~100 ms after idle run slower (clocks, TLB); keep it out of every arm
    for i in range(300):
        ops.gather_pairwise_dot(group, ids[i % 8], dense, out=out)
    torch.cuda.synchronize()
    reps = int(os.environ.get("RING_AB_REPS", "3"))
    for cfg in list(cfgs) * reps:
        c, _, fl = str(cfg).partition(":")
        os.environ["REC_RING_CFG"] = c
        os.environ["REC_RING_FLAGS"] = fl or "0"
        for i in range(10):
            ops.gather_pairwise_dot(group, ids[i % 8], dense, out=out)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            ops.gather_pairwise_dot(group, ids[i % 8], dense, out=out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps))
        ops.gather_pairwise_dot(group, ids[0], dense, out=out)
        chk = out.double().sum().item()
        byt = B * 15844
        p50 = ts[len(ts) // 2]
        res[str(cfg)] = {"p10_us": round(ts[len(ts) // 10], 1), "p50_us": round(p50, 1),
                         "p90_us": round(ts[9 * len(ts) // 10], 1), "frac_p50": round(byt / (p50 * 1e-6) / 8e12, 4),
                         "checksum": chk}
        print(cfg, res[str(cfg)], flush=True)
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run(sys.argv[2:] or ["0"])
    else:
        cfgs = sys.argv[1:] or ["0", "1", "2", "3"]
        env = dict(os.environ)
        subprocess.run([sys.executable, __file__, "child"] + cfgs, env=env, check=False)
        env["REC_PAIRDOT_IMPL"] = "valu"
        print("register-tiled kernel (REC_PAIRDOT_IMPL=valu):", flush=True)
        subprocess.run([sys.executable, __file__, "child", "0"], env=env, check=False)
