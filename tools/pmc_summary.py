#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_bench.sh into profiles/<tag>_*:
  <tag>_kernel_stats_<workload>.csv      the --stats summary (top kernels) of each workload's profiled run
  <tag>_profiled_run_<workload>.json     the bench line printed by that run (its event-timed ms_per_launch must agree)
  <tag>_pmc_traffic[_<workload>].json    HBM bytes per launch of the workload's dominant kernel: FETCH_SIZE x 2 (gfx950: 128-B
                                         requests of 16-B-per-lane streaming reads are tallied at 64 B; MI355X_MICROARCH.md
                                         § HBM) + WRITE_SIZE, both KiB — against the algorithmic bytes the bench line states
usage: python tools/pmc_summary.py r03"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
dst = os.path.join(ROOT, "profiles")
KERNEL = {"dlrm_fused": "pairdot_ring", "gather": "gather_uniform_kernel", "din": "din_gather_pool_grp_kernel",
          "sasrec": "sasrec_last_row_kernel", "autoint": "mha_ctr_stack_kernel"}


def find(src, sub, pat):
    hits = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def counter_avg(path, counter):
    agg = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            a = agg.setdefault(row["Kernel_Name"], [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (s / n, n) for k, (s, n) in agg.items()}


def bench_line(path):
    try:
        lines = [ln for ln in open(path) if ln.startswith("{")]
        return json.loads(lines[-1])
    except Exception:  # noqa: BLE001
        return None


main_out = {"note": "rocprofv3 --pmc passes (separate runs) of `python bench.py --workload W --steps 50 --warmup 10 --cpu-seconds 0 "
                    "--no-side` on MI355X; counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                    "requests of 16 B/lane streaming reads at 64 B); WRITE_SIZE exact for 16 B/lane stores",
            "config": {"batch": 65536, "fields": 26, "vocab": 1000000, "dim": 128, "ids": "uniform"}, "kernels": {}}
for wl, ksub in KERNEL.items():
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{wl}")
    if not os.path.isdir(src):
        continue
    stats = find(src, "stats", "*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats_{wl}.csv"))
    line = bench_line(os.path.join(src, "bench_stats.json"))
    if line:
        json.dump(line, open(os.path.join(dst, f"{tag}_profiled_run_{wl}.json"), "w"))
    fetch, write = find(src, "fetch", "*counter_collection.csv"), find(src, "write", "*counter_collection.csv")
    if not (fetch and write and line):
        continue
    alg = line["roofline"].get("algorithmic_bytes_per_launch")
    fa, wa = counter_avg(fetch, "FETCH_SIZE"), counter_avg(write, "WRITE_SIZE")
    kernels = {}
    for k in fa:
        if ksub not in k or k not in wa:
            continue
        rd, wr = fa[k][0] * 1024 * 2, wa[k][0] * 1024
        kernels[k] = {"FETCH_SIZE_KiB_avg": fa[k][0], "WRITE_SIZE_KiB_avg": wa[k][0], "dispatches": fa[k][1],
                      "hbm_read_bytes_corrected": int(rd), "hbm_write_bytes": int(wr),
                      "traffic_bytes_per_launch": int(rd + wr), "algorithmic_bytes_per_launch": alg,
                      "traffic_over_algorithmic": round((rd + wr) / alg, 4)}
    if wl in ("dlrm_fused", "gather"):
        main_out["kernels"].update(kernels)
    else:
        cfg = {"workload": wl, "batch": line["config"]["batch_per_gpu"], "vocab": line["config"]["vocab_per_table"],
               "ids": "uniform"}
        if wl == "din":
            cfg["width"] = line["config"]["width"]
        json.dump({"note": main_out["note"] + "; the kernel reads the rows of real slots only, ids of every id batch differ: "
                           "algorithmic bytes = the mean over the rotated batches (bench line)",
                   "config": cfg, "kernels": kernels}, open(os.path.join(dst, f"{tag}_pmc_traffic_{wl}.json"), "w"), indent=1)
    print(wl, json.dumps({k[:60]: v["traffic_over_algorithmic"] for k, v in kernels.items()}))
if main_out["kernels"]:
    json.dump(main_out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)
