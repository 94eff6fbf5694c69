"""bench.py end to end on the GPU box, as the driver runs it: the default one-GPU command, and the plain `--gpus 2`
command (no launcher around it) under the one-GPU rehearsal knobs (both ranks share cuda:0, gloo instead of RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_carries_every_config(dev):
    """`python bench.py` (reduced steps / CPU budget): the headline on a plain allocation + gather_roofline + zipf +
    placed + configs{autoint, din, sasrec} + cpu_baseline + pcie_inclusive, each with a roofline measured in this run"""
    res = _run(["--steps", "10", "--warmup", "3", "--cpu-seconds", "1", "--cpu-samples", "2048", "--placed-candidates", "2"])
    assert res["n_gpus"] == 1 and res["metric"].startswith("forward samples/sec") and res["vs_baseline"] is None
    assert res["config"]["table_placement"] == {"candidates": 1}            # headline: no best-of-N
    roof = res["roofline"]
    assert roof["bound"] == "hbm" and 0.3 < roof["frac"] < 1.0 and roof["algorithmic_bytes_per_launch"] == 65536 * 15844
    assert abs(res["value"] - 65536 / (res["ms_per_step"] * 1e-3)) <= 1e-3 * res["value"]
    assert 0.3 < res["gather_roofline"]["frac"] < 1.0
    assert res["zipf"]["roofline"]["bound"] == "hbm" and res["zipf"]["unique_rows_per_launch"] < 65536 * 26
    assert res["placed"]["table_placement"]["candidates"] == 2 and len(res["placed"]["table_placement"]["probe_us"][0]) == 2
    for name, bound in (("autoint", "mfma"), ("din", "hbm"), ("sasrec", "hbm")):
        c = res["configs"][name]
        assert "error" not in c, c
        assert c["roofline"]["bound"] == bound and 0.05 < c["roofline"]["frac"] < 1.0
        assert c["launch_us"]["p10"] <= c["launch_us"]["p50"] <= c["launch_us"]["p90"]
    assert res["cpu_baseline"]["kind"] == "port" and res["cpu_baseline"]["cores"] >= 1
    assert res["cpu_baseline"]["torch_ops"]["value"] > 0
    assert "error" not in res["pcie_inclusive"]
    mf = res["model_forward"]                                              # the whole DLRM model on the headline's tables
    assert "error" not in mf, mf
    assert mf["batch"] == 65536 and res["ms_per_step"] < mf["ms_per_forward"] < 20 * res["ms_per_step"]
    assert 0.02 < mf["sparse_stage_share"] < 1.0
    assert mf["compared_samples"] == 8192 and mf["max_abs_diff_vs_fp32_mfma_dense"] <= 1e-5      # probabilities: |p| <= 1


def test_plain_gpus2_command_self_launches_and_reports_both_placements(dev):
    """`python bench.py --gpus 2` with NO launcher: the process starts its two ranks itself; the line carries the
    replicated headline and the row-sharded placement (pipelined exchange, local-shard bypass), and says which
    transport ran and how many ranks it saw."""
    res = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "4096", "--vocab", "50000", "--cpu-seconds", "0",
                "--spinup", "0.05", "--side-timeout", "240"],
               {"REC_BENCH_SHARE_GPU": "1", "REC_BENCH_BACKEND": "gloo"})
    assert res["n_gpus"] == 2 and res["config"]["placement"] == "replicated"
    assert res["config"]["launched_by"] == "bench.py" and res["config"]["control_backend"] == "gloo"
    assert abs(res["value"] - 2 * 4096 / (res["ms_per_step"] * 1e-3)) <= 1e-3 * res["value"]
    rs = res["rowshard"]
    assert "error" not in rs, rs
    ex = rs["exchange"]
    assert ex["world"] == 2 and ex["transport"] == "torch" and ex["comm"].startswith("torch.distributed/gloo")
    assert ex["rccl_ranks"] == 0                                   # gloo rehearsal: no RCCL communicator, and the line says so
    assert ex["bypass_local"] and ex["pipelined"] and ex["rows_prefetched"] > 0 and ex["prefetch_hits"] > 0
    assert ex["local_lookups"] > 0 and ex["unique_sent"] < ex["ids"]
    assert rs["value"] > 0


def test_rowshard_world1_reads_the_shard_in_place(dev):
    """--placement rowshard at one rank: no exchange, the fused kernel reads the shard in place (same launch)"""
    rep = _run(["--steps", "20", "--warmup", "5", "--batch", "16384", "--vocab", "200000", "--cpu-seconds", "0", "--no-side"])
    rs = _run(["--steps", "20", "--warmup", "5", "--batch", "16384", "--vocab", "200000", "--cpu-seconds", "0", "--no-side",
               "--placement", "rowshard"])
    assert rs["config"]["exchange"]["world"] == 1 and rs["config"]["exchange"]["lookups"] == 0
    assert rs["ms_per_step"] <= 1.15 * rep["ms_per_step"]
