// Experiment: read-only ceiling of uniformly random table rows as a function of the row size (256 B = the d = 64
// rows of DIN / SASRec, 512 B = the DLRM rows, 1024 B).  Every lane loads 16 B, 64 / (ROWB / 16) rows per
// wave-instruction, U instructions in flight per lane; nothing is stored.  Answers: what is the practical HBM
// ceiling of the configs[3] / configs[4] kernels, whose rows are 256 B?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1)))* gsrc_t;

typedef u32x4 __attribute__((address_space(1)))* gdst_t;
// STORE: 0 = read only; 1 = every row is also written to out + r * ROWB with nontemporal stores (the materialised gather's
// traffic pattern: random rows in, one sequential stream out; with ids[r] = r it is a plain copy)
template <int ROWB, int U, int NT, int STORE = 0>
__global__ __launch_bounds__(256) void k_rows(const char* __restrict__ table, const int* __restrict__ ids, int64_t R,
                                              uint32_t* sink, char* __restrict__ out = nullptr) {
  constexpr int LPR = ROWB / 16, RPI = 64 / LPR;
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

template <int ROWB, int U, int NT>
static int go(const void* table, const int* ids, int64_t R, uint32_t* sink, int blocks, hipStream_t st) {
  hipLaunchKernelGGL((k_rows<ROWB, U, NT>), dim3(blocks), dim3(256), 0, st, (const char*)table, ids, R, sink);
  return (int)hipGetLastError();
}

extern "C" int run_store(int nt, const void* table, const int* ids, int64_t R, uint32_t* sink, void* out, int blocks,
                         hipStream_t st) {
  if (nt) hipLaunchKernelGGL((k_rows<512, 16, 1, 1>), dim3(blocks), dim3(256), 0, st, (const char*)table, ids, R, sink, (char*)out);
  else hipLaunchKernelGGL((k_rows<512, 16, 0, 1>), dim3(blocks), dim3(256), 0, st, (const char*)table, ids, R, sink, (char*)out);
  return (int)hipGetLastError();
}

extern "C" int run(int rowb, int U, int nt, const void* table, const int* ids, int64_t R, uint32_t* sink, int blocks,
                   hipStream_t st) {
#define CASE(RB, UU)                                                                              \
  if (rowb == RB && U == UU) return nt ? go<RB, UU, 1>(table, ids, R, sink, blocks, st) : go<RB, UU, 0>(table, ids, R, sink, blocks, st);
  CASE(256, 8) CASE(256, 16) CASE(512, 8) CASE(512, 16) CASE(1024, 8) CASE(1024, 16) CASE(128, 16) CASE(128, 8)
  return -1;
}
