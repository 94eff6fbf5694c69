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
    assert a.arena_candidates == 1 and a.placed_candidates == 6 and a.side_timeout <= 240   # headline: a plain allocation
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


def test_self_launch_starts_one_process_per_gpu_before_any_gpu_call(monkeypatch, capsys):
    """`python bench.py --gpus N` with no launcher around it: N rank processes, the driver's contract in their
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rank 0's JSON line relayed, worst return code returned."""
    b = load_bench()
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            self.cmd, self.env = cmd, env
            self.rank = int(env["RANK"])
            started.append(self)
            import io
            self.stdout = io.StringIO('noise\n{"value": 1.0, "n_gpus": %s}\n' % env["WORLD_SIZE"]) if self.rank == 0 else None

        def poll(self):
            return 0

        def wait(self):
            return 0
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    a = b.parse()
    assert "torch" not in {m.split(".")[0] for m in sys.modules if m.startswith("bench_under_test")}    # no GPU library yet
    rc = b.launch_ranks(a)
    assert rc == 0 and len(started) == 4
    ports = {p.env["MASTER_PORT"] for p in started}
    assert len(ports) == 1 and all(p.env["MASTER_ADDR"] == "127.0.0.1" for p in started)
    assert [p.env["RANK"] for p in started] == ["0", "1", "2", "3"] == [p.env["LOCAL_RANK"] for p in started]
    assert all(p.env["WORLD_SIZE"] == "4" and p.cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] for p in started)
    out = capsys.readouterr().out.strip().splitlines()
    assert out == ['{"value": 1.0, "n_gpus": 4}']                       # exactly one line: rank 0's JSON


def test_self_launch_kills_the_other_ranks_when_one_dies(monkeypatch):
    b = load_bench()
    procs = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            self.rank, self.killed = int(env["RANK"]), False
            procs.append(self)
            import io
            self.stdout = io.StringIO("") if self.rank == 0 else None

        def poll(self):
            return 3 if self.rank == 1 else (-9 if self.killed else None)

        def kill(self):
            self.killed = True

        def wait(self):
            return 0
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    rc = b.launch_ranks(b.parse())
    assert rc != 0 and procs[0].killed and procs[2].killed and not procs[1].killed


def test_bench_multi_rank_end_to_end_without_a_gpu_fails_loudly(tmp_path):
    """the real thing, as the driver would run it, on a box without a GPU: the launcher starts the ranks, each rank
    refuses to run (no CPU fallback), the launcher reports it and returns non-zero — no hang, no partial line"""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-side"], env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return                                                       # covered by the GPU rehearsal test instead
    assert r.returncode != 0 and "needs a GPU" in r.stderr and "multi-GPU launch failed" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-2000:]
