// Experiment (round 3): WHY does the physical placement of a 13.3 GB table arena move random-row
// reads by up to 7 %, and which allocation call makes it deterministic?
//
// Stand-alone (no torch): allocates the arena by several mechanisms, times the same two minimal kernels on each
// (read-only random 512-B rows; the materialised gather's pattern = random rows in, one nontemporal stream out).
//   malloc          plain hipMalloc (what torch's caching allocator does for a block this size)
//   contig          hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   vmm<MiB>        hipMemCreate chunks of that size mapped back to back into one reserved VA range aligned to the chunk
//   frag+<mode>     first fragment VRAM (many 2-MiB allocations, every other one freed), then <mode>
// Usage: placement_probe <mode>[,<mode>...] [arenas-per-mode]      e.g.  placement_probe malloc,contig,vmm1024,vmm2 3
// Under rocprofv3 --pmc the dispatch order in the CSV is the print order here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e__ = (x);                                                              \
    if (e__ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1)))* gsrc_t;
typedef u32x4 __attribute__((address_space(1)))* gdst_t;

template <int U, int NT, int STORE>
__global__ __launch_bounds__(256) void k_rows(const char* __restrict__ table, const int* __restrict__ ids, int64_t R,
                                              uint32_t* sink, char* __restrict__ out) {
  constexpr int ROWB = 512, LPR = ROWB / 16, RPI = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int sub = lane / LPR, col = lane % LPR;
  u32x4 acc = {0, 0, 0, 0};
  for (int64_t r0 = wave * (RPI * U); r0 < R; r0 += nw * (RPI * U)) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t r = r0 + u * RPI + sub;
      if (r >= R) r = R - 1;
      const int id = ids[r];
      gsrc_t p = (gsrc_t)(uintptr_t)(table + (int64_t)id * ROWB + col * 16);
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    if (STORE) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t r = r0 + u * RPI + sub;
        if (r < R) __builtin_nontemporal_store(v[u], (gdst_t)(uintptr_t)(out + r * ROWB + col * 16));
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

struct Arena {
  char* p = nullptr;
  size_t bytes = 0;
  std::string how;
  std::vector<hipMemGenericAllocationHandle_t> handles;
  bool vmm = false, ext = false;
};

static const size_t kRows = 26ull * 1000000ull;
static const size_t kBytes = (kRows * 512 + (2u << 20) - 1) / (2u << 20) * (2u << 20);

static Arena alloc_arena(const std::string& mode) {
  Arena a;
  a.bytes = kBytes;
  a.how = mode;
  if (mode == "malloc") {
    CK(hipMalloc((void**)&a.p, a.bytes));
  } else if (mode == "contig") {
    hipError_t e = hipExtMallocWithFlags((void**)&a.p, a.bytes, hipDeviceMallocContiguous);
    if (e != hipSuccess) {
      printf("  contig: hipExtMallocWithFlags(Contiguous) -> %s\n", hipGetErrorString(e));
      (void)hipGetLastError();
      a.p = nullptr;
    }
    a.ext = true;
  } else if (mode.rfind("vmm", 0) == 0) {
    const size_t chunk = (size_t)atoll(mode.c_str() + 3) << 20;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t ck = (chunk + gran - 1) / gran * gran;
    const size_t total = (a.bytes + ck - 1) / ck * ck;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, ck, nullptr, 0));
    for (size_t off = 0; off < total; off += ck) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, ck, &prop, 0));
      CK(hipMemMap((char*)va + off, ck, 0, h, 0));
      a.handles.push_back(h);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    a.p = (char*)va;
    a.bytes = total;
    a.vmm = true;
    printf("  %s: granularity %zu KiB, chunk %zu MiB x %zu, va %p\n", mode.c_str(), gran >> 10, ck >> 20, a.handles.size(), va);
  } else {
    fprintf(stderr, "unknown mode %s\n", mode.c_str());
    exit(2);
  }
  return a;
}

static void free_arena(Arena& a) {
  if (!a.p) return;
  if (a.vmm) {
    CK(hipMemUnmap(a.p, a.bytes));
    for (auto h : a.handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(a.p, a.bytes));
  } else {
    CK(hipFree(a.p));
  }
  a.p = nullptr;
}

int main(int argc, char** argv) {
  const char* modes_arg = argc > 1 ? argv[1] : "malloc,contig,vmm1024,vmm2";
  const int per = argc > 2 ? atoi(argv[2]) : 3;
  const int launches = argc > 3 ? atoi(argv[3]) : 20;
  CK(hipSetDevice(0));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("device memory free %.1f GiB of %.1f GiB; arena %.2f GiB\n", fr / 1073741824.0, tot / 1073741824.0,
         kBytes / 1073741824.0);

  // ids + gather output + sink first (small, before anything fragments)
  const int64_t R = 65536 * 26;
  const int NBATCH = 4;
  std::vector<int> h(R);
  int* ids[NBATCH];
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int b = 0; b < NBATCH; ++b) {
    for (int64_t i = 0; i < R; ++i) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      h[i] = (int)((s >> 33) % kRows);
    }
    CK(hipMalloc((void**)&ids[b], R * 4));
    CK(hipMemcpy(ids[b], h.data(), R * 4, hipMemcpyHostToDevice));
  }
  char* out = nullptr;
  uint32_t* sink = nullptr;
  CK(hipMalloc((void**)&out, R * 512));
  CK(hipMalloc((void**)&sink, 64));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = 256 * 8;

  auto run = [&](int kind, const Arena& a, int n) {  // kind 0 read-only plain, 1 read-only nt, 2 gather pattern (plain loads)
    for (int i = 0; i < n; ++i) {
      if (kind == 0) hipLaunchKernelGGL((k_rows<16, 0, 0>), dim3(blocks), dim3(256), 0, st, a.p, ids[i % NBATCH], R, sink, out);
      if (kind == 1) hipLaunchKernelGGL((k_rows<16, 1, 0>), dim3(blocks), dim3(256), 0, st, a.p, ids[i % NBATCH], R, sink, out);
      if (kind == 2) hipLaunchKernelGGL((k_rows<16, 0, 1>), dim3(blocks), dim3(256), 0, st, a.p, ids[i % NBATCH], R, sink, out);
    }
  };
  auto timeit = [&](int kind, const Arena& a) {
    run(kind, a, 3);
    CK(hipEventRecord(e0, st));
    run(kind, a, launches);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / launches * 1e3f;
  };

  std::vector<std::string> modes;
  {
    std::string m(modes_arg);
    size_t p = 0;
    while (p != std::string::npos) {
      size_t q = m.find(',', p);
      modes.push_back(m.substr(p, q == std::string::npos ? q : q - p));
      p = q == std::string::npos ? q : q + 1;
    }
  }
  bool warm = false;
  for (const std::string& full : modes) {
    std::string mode = full;
    std::vector<void*> frag;
    if (mode.rfind("frag+", 0) == 0) {
      mode = mode.substr(5);
      // fragment: fill most of the free memory with 2-MiB blocks, free every other one
      size_t f2 = 0, t2 = 0;
      CK(hipMemGetInfo(&f2, &t2));
      const size_t n = (f2 - (8ull << 30)) / (2u << 20);
      frag.reserve(n);
      for (size_t i = 0; i < n; ++i) {
        void* p = nullptr;
        if (hipMalloc(&p, 2u << 20) != hipSuccess) {
          (void)hipGetLastError();
          break;
        }
        frag.push_back(p);
      }
      size_t kept = 0;
      for (size_t i = 0; i < frag.size(); ++i)
        if (i & 1) {
          CK(hipFree(frag[i]));
          frag[i] = nullptr;
        } else {
          ++kept;
        }
      printf("fragmented: %zu x 2 MiB allocated, %zu kept (every other one freed)\n", frag.size(), kept);
    }
    std::vector<Arena> as;
    for (int k = 0; k < per; ++k) {
      Arena a = alloc_arena(mode);
      if (!a.p) break;
      CK(hipMemsetAsync(a.p, 0, a.bytes, st));
      as.push_back(a);
    }
    CK(hipStreamSynchronize(st));
    if (!warm && !as.empty()) {  // clocks up: ~0.5 s of launches
      run(0, as[0], 1500);
      CK(hipStreamSynchronize(st));
      warm = true;
    }
    for (int pass = 0; pass < 2; ++pass)
      for (size_t k = 0; k < as.size(); ++k) {
        const float t0 = timeit(0, as[k]), t1 = timeit(1, as[k]), t2 = timeit(2, as[k]);
        const double rb = (double)R * 516;
        printf("%-12s pass %d arena %zu @ %p: read %7.1f us (%.2f TB/s)  read-nt %7.1f us (%.2f TB/s)  gather %7.1f us (%.2f TB/s)\n",
               full.c_str(), pass, k, (void*)as[k].p, t0, rb / t0 / 1e6, t1, rb / t1 / 1e6, t2, (double)R * 1028 / t2 / 1e6);
        fflush(stdout);
      }
    for (auto& a : as) free_arena(a);
    for (void* p : frag)
      if (p) CK(hipFree(p));
  }
  return 0;
}
