#!/usr/bin/env python3
"""A/B of the materialised gather at BASELINE configs[1]: LDS-DMA ring kernel vs the register-staged kernel
(REC_GATHER_IMPL=regs, child process: the choice is read once).  8 rotating id batches, per-launch HIP events."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "recommend-tf2.0_amd")):
    sys.path.insert(0, p)


def run(steps=60, reps=3):
    import torch
    from recamd import ops
    dev = torch.device("cuda:0")
    B, F, V, D = 65536, 26, 1_000_000, 128
    gen = torch.Generator(device=dev).manual_seed(0)
    arena = torch.empty((F, V, D), dtype=torch.float32, device=dev)
    arena.uniform_(-0.05, 0.05, generator=gen)
    group = ops.TableGroup([arena[f] for f in range(F)])
    ids = [torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen) for _ in range(8)]
    out = torch.empty((B, F * D), dtype=torch.float32, device=dev)
    for i in range(300):
        ops.gather_concat(group, ids[i % 8], out=out)
    torch.cuda.synchronize()
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            ops.gather_concat(group, ids[i % 8], out=out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps))
        p50 = ts[len(ts) // 2]
        print({"impl": os.environ.get("REC_GATHER_IMPL", "ring"), "p10_us": round(ts[len(ts) // 10], 1), "p50_us": round(p50, 1),
               "p90_us": round(ts[9 * len(ts) // 10], 1), "frac_p50": round(B * 26728 / (p50 * 1e-6) / 8e12, 4)}, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        for impl in ("ring", "regs", "ring", "regs"):
            env = dict(os.environ)
            env["REC_GATHER_IMPL"] = impl
            subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
