"""Build-time invariants of the hand-counted Dense kernels (csrc/dense_bf16x3.hip): their global loads are inline asm that
lands asynchronously, so a destination register the compiler SPILLED between a load's issue and its wait would be clobbered
when the load arrives (a stamps build of round 3 faulted exactly that way).  hipcc cross-compiles here without a GPU; the
remarks of -Rpass-analysis=kernel-resource-usage give registers, spills and scratch per kernel."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recommend-tf2.0_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _kernel_resources(tmp_path, src, wanted):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CSRC}",
           "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(CSRC, src),
           "-o", str(tmp_path / (src + ".o"))]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = {}
    for blk in re.split(r"remark: Function Name: ", r.stderr)[1:]:
        name = blk.split()[0]
        for key in wanted:
            if f"{key}E" in name:
                num = lambda what: int(re.search(what + r": (\d+)", blk).group(1))  # noqa: E731
                seen[key] = dict(vgprs=num(r"    VGPRs"), spill=num(r"VGPRs Spill"), scratch=num(r"ScratchSize \[bytes/lane\]"),
                                 waves=num(r"Occupancy \[waves/SIMD\]"))
    assert set(seen) == set(wanted), seen
    for key, v in seen.items():
        assert v["spill"] == 0 and v["scratch"] == 0, (key, v)
    return seen


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hand_counted_dense_kernels_have_no_spills(tmp_path):
    b3 = _kernel_resources(tmp_path, "dense_bf16x3.hip", ["dense_bf16x3_pipe_kernel", "dense_bf16x3_pipe_deep_kernel"])
    assert b3["dense_bf16x3_pipe_kernel"]["waves"] >= 3          # three workgroups of four waves per CU
    assert b3["dense_bf16x3_pipe_deep_kernel"]["waves"] >= 2
    h2 = _kernel_resources(tmp_path, "dense_f16x2.hip", ["dense_f16x2_pipe_kernel"])
    assert h2["dense_f16x2_pipe_kernel"]["waves"] >= 3
