import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "inflight.so"))
lib.run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
B, F, V, D = 65536, 26, 1_000_000, 128
arena = torch.empty(F, V, D, device=dev).uniform_(-0.05, 0.05)
ids = torch.randint(0, V, (B, F), device=dev, dtype=torch.int32)
dense = torch.randn(B, D, device=dev)
sink = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
outp = torch.empty(B, 512, device=dev)
names = {0: "no stores", 1: "2 nt stores (dup lanes) stride 480", 2: "2 nt stores predicated", 3: "stride 512", 4: "plain stores", 5: "first 1 KB only", 6: "2 KB rows stride 512", 7: "wave-contiguous rows"}
cfgs = [(1, 3, 0, (1 << 16)), (1, 3, 0, (1 << 16) + (2 << 17)), (0, 2, 0, 0), (0, 3, 0, 0), (0, 4, 0, 0)]
for T, W, lds, chunk in cfgs:
    args = (T, W, arena.data_ptr(), V, ids.data_ptr(), dense.data_ptr(), B, sink.data_ptr(), st, lds, chunk, outp.data_ptr())
    for _ in range(3):
        rc = lib.run(*args)
    assert rc == 0, rc
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        lib.run(*args)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"T={T} tiles/wave  W={W} waves/SIMD {names[chunk >> 17] if T else 'role split: 3 loader waves + 1 writer wave per block'} ({T*W*4*13.8:.0f} KB/CU): {ms*1e3:.1f} us  {B*27*512/ms/1e6:.0f} GB/s", flush=True)
