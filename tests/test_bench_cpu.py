"""bench.py host logic that needs no GPU: the committed PMC traffic lookup, the argument surface the driver uses, and the
command line of the parked side-leg child of an N > 1 run."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pmc_traffic_reads_the_newest_committed_profile():
    b = load_bench()
    cfg = {"batch": 65536, "fields": 26, "vocab": 1000000, "dim": 128, "ids": "uniform"}
    traffic, src = b.pmc_traffic("pairdot_ring_kernel", cfg)
    assert src is not None and src.endswith("_pmc_traffic.json")
    alg = 65536 * 15844
    assert 0.99 * alg < traffic < 1.02 * alg                      # FETCH_SIZE x 2 + WRITE_SIZE ~ the algorithmic bytes
    g, _ = b.pmc_traffic("gather_uniform_kernel", cfg)
    assert 0.99 * 65536 * 26728 < g < 1.02 * 65536 * 26728
    assert b.pmc_traffic("pairdot_ring_kernel", dict(cfg, ids="zipf")) == (None, None)      # no profile for that shape
    d = json.load(open(os.path.join(ROOT, "profiles", src)))
    assert all(abs(v["traffic_over_algorithmic"] - 1.0) < 0.02 for v in d["kernels"].values())


def test_driver_flags_and_defaults(monkeypatch):
    b = load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "3"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup, a.workload, a.placement) == (4, 7, 3, "dlrm_fused", "replicated")
    assert a.arena_candidates == 6 and a.side_timeout <= 240
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps >= 20 and a.warmup >= 1          # defaults finish within minutes


def test_side_leg_command_line(monkeypatch):
    """the child of an N > 1 run measures the OTHER placement, inherits the workload flags, rendezvous on its own port"""
    b = load_bench()
    captured = {}

    class FakePopen:
        def __init__(self, cmd, **kw):
            captured["cmd"], captured["env"] = cmd, kw["env"]
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakePopen)
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "5", "--placement", "replicated", "--workload", "sasrec"])
    a = b.parse()
    alt, proc = b.spawn_side_leg(a)
    cmd = captured["cmd"]
    assert alt == "rowshard" and cmd[-6:] == ["--placement", "rowshard", "--side-leg", "--no-side", "--cpu-seconds", "0"]
    assert cmd.count("--placement") == 1 and "--workload" in cmd and "sasrec" in cmd and "--steps" in cmd
    assert captured["env"]["MASTER_PORT"] == str(29500 + 137)
