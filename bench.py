#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): forward samples/sec of the DLRM sparse stage on synthetic
Criteo-shaped batches — 65 536 samples x 26 sparse fields x dim 128, 1M rows per table — on N
MI355X GPUs of one node.

A step = ONE pass of the hot path over one batch already resident in HBM: the fused
gather + pairwise-dot launch (rec_gather_pairwise_dot_f32: ids -> 26 embedding rows + the bottom-MLP
vector -> 351 dots ++ dense, (B, 479) out).  With --workload gather the step is the materialised
gather+concat (rec_gather_concat_f32, (B, 3328) out) — the kernel the north-star roofline target is
quoted on; its roofline is also reported beside the headline as "gather_roofline".

Multi-GPU (driver: torch.distributed.run, one rank per GPU): weak scaling, B samples per GPU.
--placement replicated (default): every GPU holds all 26 tables (13.3 GB of 288 GB) exactly like the
  reference's MirroredStrategy mirrors its variables; the forward has no data-path collective.
--placement rowshard: tables row-sharded cyclically (owner = id % G) with an RCCL all-to-all of ids
  out and rows back (recamd.dist.ShardedTables) — the xGMI-bound placement.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "recommend-tf2.0_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def pmc_traffic(kernel_substr, a):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc runs of this same bench).  PMC
    counters cannot be read from inside the timed run, so `traffic` is the latest committed
    measurement for the SAME workload shape, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        c = d.get("config", {})
        if (c.get("batch"), c.get("fields"), c.get("vocab"), c.get("dim"), c.get("ids")) != \
                (a.batch, a.fields, a.vocab, a.dim, a.ids):
            continue
        for k, v in d.get("kernels", {}).items():
            if kernel_substr in k:
                return int(v["traffic_bytes_per_launch"]), os.path.basename(f)
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["dlrm_fused", "gather"], default="dlrm_fused")
    ap.add_argument("--placement", choices=["replicated", "rowshard"], default="replicated")
    ap.add_argument("--ids", choices=["uniform", "zipf"], default="uniform")
    ap.add_argument("--batch", type=int, default=65536, help="samples per GPU")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--cpu-samples", type=int, default=16384)
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal knobs (one-GPU boxes only): REC_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # REC_BENCH_BACKEND=gloo replaces RCCL, to exercise the multi-rank control flow without N GPUs
    if os.environ.get("REC_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("REC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from recamd import ops

    B, F, V, D = a.batch, a.fields, a.vocab, a.dim
    n = F + 1
    P = n * (n - 1) // 2

    # ---- synthetic data, generated on device (there is no dataset; BASELINE config 2) ----------
    gen = torch.Generator(device=dev).manual_seed(0)
    if a.placement == "replicated":
        arena = torch.empty((F, V, D), dtype=torch.float32, device=dev)
        arena.uniform_(-0.05, 0.05, generator=gen)  # keras 'random_uniform' (dlrm/model.py:34)
        group = ops.TableGroup([arena[f] for f in range(F)])
        sharded = None
    else:
        from recamd.dist import ShardedTables
        rows_local = (V + world - 1 - rank) // world  # rows r with r % world == rank
        arena = torch.empty((F, rows_local, D), dtype=torch.float32, device=dev)
        arena.uniform_(-0.05, 0.05, generator=gen)
        sharded = ShardedTables([arena[f] for f in range(F)], [V] * F, rank, world)
        group = None
    gen_i = torch.Generator(device=dev).manual_seed(1 + rank)
    if a.ids == "uniform":
        ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32, generator=gen_i)
    else:  # Zipf(1.05) clipped to V — Criteo-like skew
        import numpy as np
        z = np.random.default_rng(1 + rank).zipf(1.05, size=(B, F))
        ids = torch.from_numpy(((z - 1) % V).astype("int32")).to(dev)
    dense = torch.rand((B, D), device=dev, generator=gen_i)
    # (B, 479) result with a 480-float (16-B aligned) row stride, as recamd.ops allocates it
    out_fused = torch.empty((B, (P + D + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :P + D]
    out_gather = torch.empty((B, F * D), dtype=torch.float32, device=dev)

    def step_fused():
        if sharded is None:
            ops.gather_pairwise_dot(group, ids, dense, out=out_fused)
        else:
            emb = sharded.lookup(ids)  # (B, F*D) through the all-to-all pair
            x = torch.cat([emb.view(B, F, D), dense[:, None, :]], dim=1)
            ops.pairwise_dot(x, out=out_fused[:, :P])
            out_fused[:, P:] = dense

    def step_gather():
        if sharded is None:
            ops.gather_concat(group, ids, out=out_gather)
        else:
            sharded.lookup(ids, out=out_gather)

    step = step_fused if a.workload == "dlrm_fused" else step_gather
    bytes_fused = B * (F * D * 4 + F * 4 + D * 4 + (P + D) * 4)     # 15 844 B/sample at 26x128
    bytes_gather = B * F * (2 * D * 4 + 4)                            # 26 728 B/sample at 26x128
    bytes_step = bytes_fused if a.workload == "dlrm_fused" else bytes_gather

    def barrier():
        if world > 1:
            dist.barrier()

    def timed(fn, steps):
        """barrier + sync, `steps` launches bracketed by HIP events on the launch stream, sync +
        barrier; returns (wall seconds, mean device ms per launch from the events)."""
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        return t1 - t0, e0.elapsed_time(e1) / steps

    for _ in range(a.warmup):
        step()
    wall, dev_ms = timed(step, a.steps)
    if world > 1:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    # the north-star roofline kernel (materialised gather) measured in the same run, rank-local
    gather_roof = None
    if a.workload == "dlrm_fused" and sharded is None:
        for _ in range(5):
            step_gather()
        _, g_ms = timed(step_gather, max(20, a.steps // 4))
        ach = bytes_gather / (g_ms * 1e-3) / 1e9
        g_traffic, g_src = pmc_traffic("gather_uniform_kernel", a)
        gather_roof = {"kernel": "gather_uniform_kernel", "bound": "hbm", "achieved": round(ach, 1),
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                       "traffic": g_traffic, "traffic_source": g_src,
                       "algorithmic_bytes_per_launch": bytes_gather, "ms_per_launch": round(g_ms, 4),
                       "samples_per_s": round(B / (g_ms * 1e-3), 1)}

    cpu_base = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        cpu_base = cpu_baseline(a, arena, ids, dense, F, V, D)

    if rank == 0:
        achieved = bytes_step / (dev_ms * 1e-3) / 1e9
        kern = ("rec::pairdot_kernel<32, 27, true, true, 0, true> (fused gather + pairwise dot, staged stores)"
                if a.workload == "dlrm_fused" else "rec::gather_uniform_kernel<32, 0>")
        traffic, traffic_src = (pmc_traffic("pairdot_kernel" if a.workload == "dlrm_fused" else
                                            "gather_uniform_kernel", a) if sharded is None else (None, None))
        res = {
            "metric": "forward samples/sec, Criteo-shape 65536x26 sparse x dim128",
            "value": round(world * B * a.steps / wall, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(wall / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("DLRM 26 sparse x 1M vocab x dim 128, batch 65536/GPU: fused embedding gather + "
                                    "pairwise-dot (BASELINE configs[1])" if a.workload == "dlrm_fused" else
                                    "DLRM-shape materialised embedding gather+concat, batch 65536/GPU"),
                       "step": a.workload, "batch_per_gpu": B, "global_batch": B * world, "fields": F,
                       "vocab_per_table": V, "dim": D, "ids": a.ids, "placement": a.placement,
                       "parallelism": f"dp{world}"},
            "roofline": {"kernel": kern, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": bytes_step, "ms_per_launch": round(dev_ms, 4)},
        }
        if gather_roof is not None:
            res["gather_roofline"] = gather_roof
        if cpu_base is not None:
            res["cpu_baseline"] = cpu_base
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(a, arena, ids, dense, F, V, D):
    """The CPU restatement (oracle/oracle_c.c, OpenMP, kind "port") of the same workload on the
    box's host cores: the reference's own CPU path cannot run (TensorFlow is not installed, the
    reference never travels to the GPU box).  Bounded sample: the first --cpu-samples samples of
    the same batch against the same 13.3 GB tables copied to host memory."""
    import numpy as np
    try:
        from oracle import c_oracle
        c_oracle.load()
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    Bc = min(a.cpu_samples, ids.shape[0])
    host = arena.cpu().numpy()  # (F, V, D) fp32
    tables = [host[f] for f in range(F)]
    ids_h = ids[:Bc].cpu().numpy()
    dense_h = dense[:Bc].cpu().numpy()
    n = F + 1
    scratch = np.empty((Bc, n * D), np.float32)
    out = np.empty((Bc, n * (n - 1) // 2 + D), np.float32)
    if a.workload == "dlrm_fused":
        fn = lambda: c_oracle.dlrm_gather_dot(tables, ids_h, dense_h, scratch, out)  # noqa: E731
    else:
        fn = lambda: c_oracle.gather_concat(tables, ids_h)  # noqa: E731
    fn()  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el >= a.cpu_seconds or reps >= 1000:
            break
    return {"value": round(Bc * reps / el, 1), "unit": "samples/s", "cores": c_oracle.num_threads(),
            "kind": "port",
            "sample": f"first {Bc} samples of the same batch x {reps} passes ({el:.1f} s), same 26x1Mx128 tables "
                      f"in host memory, C/OpenMP restatement of gather+concat+pairwise-dot (TensorFlow unavailable)"}


if __name__ == "__main__":
    main()
