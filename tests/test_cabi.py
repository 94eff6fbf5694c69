"""C-ABI checks that need no GPU: the library loads, exports every entry point that
include/recamd.h declares, and validates arguments before touching the device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "recommend-tf2.0_amd", "recamd", "librecamd.so")
HDR = os.path.join(ROOT, "include", "recamd.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return ctypes.CDLL(LIB)


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rec_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["rec_gather_concat_f32", "rec_pairwise_dot_f32", "rec_gather_pairwise_dot_f32", "rec_fm_layer_f32",
                 "rec_cross_f32", "rec_fm_onehot_f32", "rec_dense_f32", "rec_mha_ctr_f32", "rec_din_attn_pool_f32",
                 "rec_mha_rowmask_f32", "rec_layernorm_residual_f32", "rec_gather_dot_scores_f32", "rec_shard_bucket_i32",
                 "rec_unpermute_rows_f32", "rec_version", "rec_last_error"]:
        assert must in names


def test_every_declared_symbol_is_exported(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in recamd.h but not exported: {missing}"


def test_version_and_error_string(lib):
    assert lib.rec_version() == 100
    # NULL tables -> REC_EINVAL (-1), before any HIP call
    rc = lib.rec_gather_concat_f32(None, 3, None, 0, ctypes.c_int64(3), ctypes.c_int64(1), None, ctypes.c_int64(4), None, None)
    assert rc == -1
    buf = ctypes.create_string_buffer(256)
    n = lib.rec_last_error(buf, 256)
    assert n > 0 and b"tables is NULL" in buf.value


def test_shape_validation_without_gpu(lib):
    class Desc(ctypes.Structure):
        _fields_ = [("base", ctypes.c_void_p), ("vocab", ctypes.c_int64), ("dim", ctypes.c_int32), ("out_col", ctypes.c_int32)]

    d = (Desc * 1)(Desc(0x1000, 10, 8, 0))
    # F out of range -> REC_ESHAPE (-2)
    assert lib.rec_gather_concat_f32(d, 65, ctypes.c_void_p(0x1000), 0, ctypes.c_int64(65), ctypes.c_int64(1),
                                     ctypes.c_void_p(0x1000), ctypes.c_int64(8), None, None) == -2
    # out_stride smaller than the concat width -> REC_ESHAPE
    assert lib.rec_gather_concat_f32(d, 1, ctypes.c_void_p(0x1000), 0, ctypes.c_int64(1), ctypes.c_int64(4),
                                     ctypes.c_void_p(0x1000), ctypes.c_int64(4), None, None) == -2
    # bad ids dtype -> REC_EINVAL
    assert lib.rec_gather_concat_f32(d, 1, ctypes.c_void_p(0x1000), 7, ctypes.c_int64(1), ctypes.c_int64(4),
                                     ctypes.c_void_p(0x1000), ctypes.c_int64(8), None, None) == -1
    # B == 0 is a no-op success
    assert lib.rec_gather_concat_f32(d, 1, ctypes.c_void_p(0x1000), 0, ctypes.c_int64(1), ctypes.c_int64(0),
                                     ctypes.c_void_p(0x1000), ctypes.c_int64(8), None, None) == 0
    lib.rec_fm_layer_workspace_floats.restype = ctypes.c_int64
    assert lib.rec_fm_layer_workspace_floats(ctypes.c_int64(65536)) >= 4096


def test_header_is_c99_and_a_c_consumer_links(tmp_path):
    """include/recamd.h is the boundary a C / cgo / JNI consumer binds: it must compile as plain C99 and link against
    librecamd.so without torch or python in the process (argument validation runs without a GPU)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "consumer.c"
    src.write_text("""
#include <stdio.h>
#include <string.h>
#include "recamd.h"
int main(void) {
  char buf[256];
  rec_table_desc d = {0};
  rec_sasrec_block blk;
  memset(&blk, 0, sizeof blk);
  if (rec_version() != 100) return 1;
  if (rec_gather_concat_f32(NULL, 3, NULL, REC_IDS_I32, 3, 1, NULL, 4, NULL, NULL) != REC_EINVAL) return 2;
  if (rec_last_error(buf, sizeof buf) <= 0 || !strstr(buf, "NULL")) return 3;
  if (rec_sasrec_last_row_supported(64, 128, 200, 101) != 1 || rec_sasrec_last_row_supported(48, 128, 200, 101) != 0) return 4;
  if (rec_sasrec_last_row_f32(&blk, NULL, 1, NULL, 0, 1, 0, NULL, 0, NULL, 0, NULL, 0, 0, NULL, 0, NULL, 0, 0, 1, 64, NULL,
                              NULL, 0, NULL, NULL) != REC_EINVAL) return 5;
  (void)d;
  puts("ok");
  return 0;
}
""")
    exe = tmp_path / "consumer"
    libdir = os.path.dirname(LIB)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                           "-lrecamd", f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_pybind_shim_imports_and_wraps_everything():
    import recamd
    c_names = {n[4:] for n in declared_functions() if n not in ("rec_version", "rec_last_error")}
    shim = set(dir(recamd.C))
    assert c_names <= shim, f"missing in pybind shim: {sorted(c_names - shim)}"


def test_ops_refuse_cpu_tensors():
    """The product path fails loudly without a GPU (no CPU fallback, no oracle routing)."""
    import torch
    from recamd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.TableGroup([torch.zeros(4, 4)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pairwise_dot(torch.zeros(2, 3, 4))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "recommend-tf2.0_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f"{f} references the oracle"


@pytest.mark.gpu
def test_integration_md_ctypes_stub_runs_on_the_gpu(dev):
    """INTEGRATION.md §B is executable documentation: its ctypes stub (no torch types, no pybind11) is extracted from the
    file, pointed at the built library and used to launch both entry points; results are checked against the oracle."""
    import numpy as np
    import torch
    from oracle import ref_numpy as ref
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# recamd_ffi\.py.*?)```", md, flags=re.S)
    assert m, "INTEGRATION.md lost its ctypes stub"
    code = m.group(1).replace("/path/to/repo/recommend-tf2.0_amd/recamd/librecamd.so", LIB)
    ns = {}
    exec(compile(code, "INTEGRATION.md#recamd_ffi", "exec"), ns)
    rng = np.random.default_rng(0)
    F, V, D, B = 26, 50, 128, 300
    tables = [rng.normal(size=(V, D)).astype(np.float32) for _ in range(F)]
    ids = rng.integers(0, V, size=(B, F)).astype(np.int32)
    dense = rng.normal(size=(B, D)).astype(np.float32)
    tt = [torch.from_numpy(t).to(dev) for t in tables]
    t_ids, t_dense = torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev)
    out = torch.empty((B, F * D), dtype=torch.float32, device=dev)
    ns["gather_concat"]([t.data_ptr() for t in tt], [V] * F, D, t_ids.data_ptr(), B, out.data_ptr())
    width = (F + 1) * F // 2 + D
    z = torch.empty((B, width), dtype=torch.float32, device=dev)
    ns["dlrm_interaction"]([t.data_ptr() for t in tt], [V] * F, D, t_ids.data_ptr(), t_dense.data_ptr(), B, z.data_ptr())
    torch.cuda.synchronize()
    emb = ref.gather_concat(tables, ids)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), emb.view(np.uint32))
    from tests.util import close_dot
    X = np.concatenate([emb.reshape(B, F, D), dense[:, None, :]], axis=1)
    assert close_dot(z.cpu().numpy()[:, :351], X)
    assert np.array_equal(z.cpu().numpy()[:, 351:], dense)


def test_int64_ids_saturate_instead_of_wrapping():
    """ids beyond int32 must not alias a valid row after the narrowing (2**32 + 5 wrapped to row 5)"""
    import numpy as np
    from recamd import nn
    x = np.array([[0, 5, 2 ** 31 - 1], [2 ** 32 + 5, -(2 ** 40), -3]], dtype=np.int64)
    got = nn.to_device_ids(x, "cpu")
    assert got.dtype.is_floating_point is False and got.element_size() == 4
    assert got.tolist() == [[0, 5, 2 ** 31 - 1], [2 ** 31 - 1, -1, -1]]
