// K1 for the BASELINE shape (D = 128, plain concat output): the materialised gather through a per-wave LDS ring.
//
// Same transport as the fused kernel (pairwise_dot_ring.hip): table rows travel HBM -> LDS by global_load_lds_dwordx4
// with the streaming (nt) policy — measured 6.1 TB/s of random 512-B rows against 5.5 TB/s for register loads — and
// leave LDS -> VGPR -> HBM as nontemporal 16-B stores.  A ring slot holds 32 consecutive (b, f) rows = 16 KiB; with the
// plain concat layout (out_stride = F * D, out_col[f] = f * D) row r of the launch lands at out + r * 512 B, so the
// store addresses need no per-row resolution.  Per step a wave issues [1 id DMA] [16 row DMAs] [16 stores], all by
// inline asm with hand-counted s_waitcnt vmcnt(N) (issue-order retirement), exactly as the fused kernel does.
// Other shapes (mixed dims, strided / offset outputs, float ids) stay on gather.hip.
#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(512))) float g_gring_zero_row[128];

#define REC_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")
#define REC_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory")

__device__ __forceinline__ uint32_t gr_lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void gr_glds4(const void* g, uint32_t lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds)
               : "memory");
}
// 16 pieces of 1 KiB, back to back, streaming policy
__device__ __forceinline__ void gr_burst16(const uint64_t (&g)[16], uint32_t lds) {
  unsigned keep;
#define P_(i) "s_nop 0\n\tglobal_load_lds_dwordx4 %" #i ", off nt\n\ts_add_u32 m0, m0, 0x400\n\t"
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\t" P_(2) P_(3) P_(4) P_(5) P_(6) P_(7) P_(8) P_(9) P_(10) P_(11) P_(12)
                   P_(13) P_(14) P_(15) P_(16) "s_nop 0\n\tglobal_load_lds_dwordx4 %17, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8]),
                 "v"(g[9]), "v"(g[10]), "v"(g[11]), "v"(g[12]), "v"(g[13]), "v"(g[14]), "v"(g[15])
               : "memory");
#undef P_
}
// nontemporal 16-B store under an explicit lane mask (a zero mask still issues: the count of memory operations per
// step stays static)
__device__ __forceinline__ void gr_store16_nt_masked(void* p, f32x4 v, uint64_t mask) {
  uint64_t keep;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\tglobal_store_dwordx4 %1, %2, off nt\n\ts_nop 1\n\ts_mov_b64 exec, %0"
               : "=&s"(keep)
               : "v"(p), "v"(v), "s"(mask)
               : "memory");
}

template <int S, int WPB>
__global__ __launch_bounds__(WPB * 64, 1) void gather_ring_kernel(TableSet ts, const int32_t* __restrict__ ids,
                                                                  int64_t ids_stride, int F, int64_t R,
                                                                  float* __restrict__ out, int* __restrict__ oob_flag) {
  constexpr int ROWS = 32, NDMA = 16, NST = 16;
  constexpr int SLOT = NDMA * 1024;
  constexpr int IDB = 256;
  constexpr int WAVE_LDS = S * SLOT + 2 * IDB;
  static_assert(S >= 1 && S <= 2, "vmcnt is a 6-bit counter: (S - 1) steps + one step's DMAs and stores must stay <= 63");
  extern __shared__ __attribute__((aligned(1024))) char lds_all[];
  // table descriptors by field in LDS: a lane-varying kernarg index would compile to vector-memory loads, which the
  // hand-counted waits below do not know about
  __shared__ const float* s_base[REC_MAX_TABLES];
  __shared__ int32_t s_vocab[REC_MAX_TABLES];
  if (threadIdx.x < REC_MAX_TABLES) {
    s_base[threadIdx.x] = ts.base[threadIdx.x];
    s_vocab[threadIdx.x] = ts.vocab[threadIdx.x];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* lds_wave = lds_all + w * WAVE_LDS;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(gr_lds_addr(lds_wave));
  const uint32_t idb_base = lds_base + S * SLOT;
  const int* idb = reinterpret_cast<const int*>(lds_wave + S * SLOT);

  const int64_t nchunks = (R + ROWS - 1) / ROWS;
  const int64_t nwaves = (int64_t)gridDim.x * WPB;
  const int64_t gw = (int64_t)blockIdx.x * WPB + w;
  const int64_t nk = gw < nchunks ? (nchunks - gw + nwaves - 1) / nwaves : 0;  // chunks gw + k * nwaves
  if (nk == 0) return;

  const int h = lane >> 5, c32 = lane & 31;
  const char* zrow = reinterpret_cast<const char*>(g_gring_zero_row);
  uint32_t bad = 0;

  // (b, f) of row `lane & 31` of chunk index kk (clamped for the tail: a valid element always)
  auto row_of = [&](int64_t kk, int64_t& b, int& f, bool& live) {
    const int64_t kc = kk < nk ? kk : nk - 1;
    const int64_t r = (gw + kc * nwaves) * ROWS + c32;
    live = kk < nk && r < R;
    const int64_t rc = r < R ? r : R - 1;
    if (R < (int64_t)0x7fffffff) {
      const uint32_t b32 = (uint32_t)rc / (uint32_t)F;
      b = b32;
      f = (int)((uint32_t)rc - b32 * (uint32_t)F);
    } else {
      b = rc / F;
      f = (int)(rc - b * F);
    }
  };
  auto issue_ids = [&](int64_t kk) {
    int64_t b;
    int f;
    bool live;
    row_of(kk, b, f, live);
    gr_glds4(ids + b * ids_stride + f, idb_base + (uint32_t)((kk & 1) * IDB));
  };
  auto row_addrs = [&](int64_t kk, uint64_t (&g)[16]) {
    int64_t b;
    int f;
    bool live;
    row_of(kk, b, f, live);
    const uint32_t id = (uint32_t)idb[(kk & 1) * 64 + lane];
    const bool ok = id < (uint32_t)s_vocab[f];
    bad |= (live && !ok) ? 1u : 0u;
    const char* src = (live && ok) ? reinterpret_cast<const char*>(s_base[f]) + ((uint64_t)id << 9) : zrow;
    const uint64_t a = reinterpret_cast<uint64_t>(src);
    const int alo = (int)(uint32_t)a, ahi = (int)(uint32_t)(a >> 32);
#pragma unroll
    for (int t = 0; t < NDMA; ++t) {
      const int row = 2 * t + h;
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(row * 4, alo);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(row * 4, ahi);
      g[t] = (((uint64_t)hi << 32) | lo) + (uint32_t)(c32 * 16);
    }
  };

  issue_ids(0);
#pragma unroll
  for (int p = 0; p < S; ++p) {
    if (p == 0) REC_VMCNT(0); else REC_VMCNT(NDMA);
    uint64_t g[16];
    row_addrs(p, g);
    REC_LGKMCNT0();
    issue_ids(p + 1);
    gr_burst16(g, lds_base + (uint32_t)(p * SLOT));
  }

  for (int64_t k0 = 0; k0 < nk; k0 += S) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t k = k0 + s;
      if (k >= nk) break;
      // issue order per step j: [ids(j+S+1)] [rows(j+S) x 16] [stores(j) x 16]; see pairwise_dot_ring.hip
      if (S == 1) {
        if (k == 0) REC_VMCNT(0); else REC_VMCNT(NST);
      } else {
        if (k == 0) REC_VMCNT(NDMA); else REC_VMCNT(NDMA + NST);
      }
      uint64_t g[16];
      row_addrs(k + S, g);
      const char* slot = lds_wave + s * SLOT;
      f32x4 v[NDMA];
#pragma unroll
      for (int t = 0; t < NDMA; ++t) v[t] = *reinterpret_cast<const f32x4*>(slot + t * 1024 + lane * 16);
      REC_LGKMCNT0();
      issue_ids(k + S + 1);
      gr_burst16(g, lds_base + (uint32_t)(s * SLOT));
      // rows 2t + h of this chunk land at out + (r0 + 2t + h) * 512 B
      const int64_t r0 = (gw + k * nwaves) * ROWS;
      char* o = reinterpret_cast<char*>(out) + (r0 + h) * 512 + c32 * 16;
#pragma unroll
      for (int t = 0; t < NDMA; ++t) {
        const uint64_t mask = __ballot(r0 + 2 * t + h < R);
        gr_store16_nt_masked(o + (int64_t)t * 1024, v[t], mask);
      }
    }
  }
  REC_VMCNT(0);
  if (bad && oob_flag) *oob_flag = 1;
}

// returns false when the shape is not covered
bool gather128_ring_dispatch(const TableSet& ts, int F, const void* ids, int64_t ids_stride, int64_t R, float* out,
                             int* oob, hipStream_t st) {
  if (R < 64 * 32) return false;  // tiny launches: nothing to pipeline
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  constexpr int S = 2, WPB = 4;
  constexpr int LDS = WPB * (S * 16 * 1024 + 512);
  auto kern = gather_ring_kernel<S, WPB>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return false;
    attr_set = true;
  }
  const int64_t nchunks = (R + 31) / 32;
  int64_t grid = cus;
  const int64_t need = (nchunks + WPB - 1) / WPB;
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WPB * 64), LDS, st, ts, reinterpret_cast<const int32_t*>(ids),
                     ids_stride, F, R, out, oob);
  return true;
}

}  // namespace rec
