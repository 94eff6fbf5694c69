import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "aos_load.so"))
lib.run.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
arena = torch.empty(F, V, D, device=dev).uniform_(-0.05, 0.05)
ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
dense = torch.randn(B, D, device=dev)
zero = torch.zeros(D, device=dev)
sink = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
for which, name in ((2, "coalesced"), (0, "aos_256B"), (1, "aos_alt32B")):
    for _ in range(3):
        lib.run(which, arena.data_ptr(), V, ids.data_ptr(), dense.data_ptr(), zero.data_ptr(), B, sink.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.run(which, arena.data_ptr(), V, ids.data_ptr(), dense.data_ptr(), zero.data_ptr(), B, sink.data_ptr(), st)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{name}: {ms*1e3:.1f} us  {B*27*512/ms/1e6:.0f} GB/s", flush=True)
