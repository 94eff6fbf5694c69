"""python tools/exp/run_rowsize_ceiling.py — builds rowsize_ceiling.hip on the box and prints the read-only rate of
uniformly random rows by row size (see the .hip header)."""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "rowsize_ceiling.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                       os.path.join(here, "rowsize_ceiling.hip")])
lib = C.CDLL(so)
lib.run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
TABLE_BYTES = 8 << 30          # >> 256 MiB Infinity Cache
table = torch.empty(TABLE_BYTES // 4, device=dev, dtype=torch.float32).uniform_(-1, 1)
sink = torch.zeros(4, device=dev, dtype=torch.int32)
st = torch.cuda.current_stream().cuda_stream
TOTAL = 872 << 20              # bytes fetched per launch (the DLRM launch's row bytes)
for rowb in (256, 512):
    R = TOTAL // rowb
    nrows = TABLE_BYTES // rowb
    ids = [torch.randint(0, nrows, (R,), device=dev, dtype=torch.int32) for _ in range(4)]
    for U in (16,):
        for nt in (0, 1):
            for blocks in (256 * 8,):
                args = lambda i: (rowb, U, nt, table.data_ptr(), ids[i % 4].data_ptr(), R, sink.data_ptr(), blocks, st)
                for i in range(8):
                    rc = lib.run(*args(i))
                assert rc == 0, rc
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                n = 40
                for i in range(n):
                    lib.run(*args(i))
                b.record(); torch.cuda.synchronize()
                ms = a.elapsed_time(b) / n
                print(f"row {rowb:5d} B  U={U:2d} nt={nt} blocks={blocks:6d}: {ms*1e3:7.1f} us  {(R*rowb + R*4)/ms/1e9:6.2f} TB/s", flush=True)

# ---- the materialised gather's pattern with a minimal kernel: 512-B rows in (random or sequential), one nt stream out
lib.run_store.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
R = TOTAL // 512
out = torch.empty(R * 512 // 4, device=dev, dtype=torch.float32)
nrows = TABLE_BYTES // 512
for name, ids in (("random rows", [torch.randint(0, nrows, (R,), device=dev, dtype=torch.int32) for _ in range(4)]),
                  ("sequential rows (a copy)", [torch.arange(i * R, (i + 1) * R, device=dev, dtype=torch.int32) for i in range(4)])):
    for nt in (0, 1):
        for blocks in (256 * 8, 256 * 26):
            args = lambda i: (nt, table.data_ptr(), ids[i % 4].data_ptr(), R, sink.data_ptr(), out.data_ptr(), blocks, st)
            for i in range(8):
                rc = lib.run_store(*args(i))
            assert rc == 0, rc
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(40):
                lib.run_store(*args(i))
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 40
            print(f"read + nt store, {name}, nt loads={nt} blocks={blocks}: {ms*1e3:7.1f} us  {(2*R*512 + R*4)/ms/1e9:6.2f} TB/s", flush=True)
