// K1 — fused multi-table embedding gather + concat for gfx950.
//
// Replaces the reference's  tf.concat([Embedding_f(sparse_inputs[:, f]) for f], axis=-1)
// (src/ctr/deep_fm/model.py:53, dcn/model.py:47, dlrm/model.py:45, autoint/model.py:46, ...):
// F StridedSlice + F GatherV2 + one ConcatV2 that re-copies everything become ONE launch whose
// only HBM traffic is the algorithmic minimum: ids once, each table row once, each output once.
//
// Roofline: HBM.  Algorithmic bytes per looked-up row = 2*D*4 + 4.
//
// Design (uniform-D fast path, D*4 bytes = LPR lanes x 16 B):
//   * a wave owns 64 consecutive (b,f) rows.  Lane l resolves row l once: id load (one coalesced
//     256-B read per wave), range check, source/destination address.
//   * the 64 rows are then moved by 64/RPI wave-instructions, RPI = 64/LPR rows each: every lane
//     moves 16 B (global_load_dwordx4 -> nontemporal global_store_dwordx4); the row addresses
//     travel lane->lanes through ds_bpermute (LDS crossbar, no LDS memory).
//   * loads are issued in batches of 16 per lane before the first store, so a wave keeps 16 KiB
//     in flight — several times the ~64 KiB/CU that Little's law needs to cover HBM latency at
//     8 TB/s.  Measured on MI355X (65 536 x 26 x 128, uniform ids): batches of 4 / 8 / 16 / 32 ->
//     65.9 / 70.0 / 71.0 / 68.1 % of 8 TB/s; plain instead of nontemporal stores -> 63.4 %;
//     nontemporal LOADS +0.2 % on uniform ids but they would bypass the caches that serve
//     Zipf-distributed ids (83.8 % of peak), so loads stay plain.  Round 2: the same gather through a per-wave
//     LDS-DMA ring with streaming loads (the transport that lifted the fused kernel from 0.59 to 0.71) is bit-exact
//     and NOT faster here (322-325 us both, tools/exp/gather_variants/gather_ring.hip, tools/exp/gather_ab.py): with
//     half of the traffic being the 872 MB output stream, the read path is not what bounds this kernel.
//   * output stores are nontemporal: the 872 MB concat output is write-once and must not evict
//     hot embedding rows from L2 / Infinity Cache (matters for Zipf-distributed ids).
#include <stdlib.h>

#include "common.h"

namespace rec {

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src_lane) {
  uint32_t lo = __shfl((uint32_t)v, src_lane, 64);
  uint32_t hi = __shfl((uint32_t)(v >> 32), src_lane, 64);
  return ((uint64_t)hi << 32) | lo;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// addresses rebuilt from shuffled integers must be tagged global (address space 1), or the
// compiler falls back to flat_load/flat_store, which also tick lgkmcnt and serialise against
// the ds_bpermute address shuffles.
typedef const u32x4 __attribute__((address_space(1)))* gsrc_t;
typedef u32x4 __attribute__((address_space(1)))* gdst_t;

// output store of a 16-B row piece.  REC_GATHER_STORE selects the cache policy (A/B builds): 0 = nontemporal,
// 1 = sc0 sc1 (system-scope write-through), 2 = sc0 sc1 nt, 3 = sc1
#ifndef REC_GATHER_STORE
#define REC_GATHER_STORE 0
#endif
#ifndef REC_GATHER_LOAD_NT
#define REC_GATHER_LOAD_NT 0     // 1 = nontemporal row loads (A/B builds)
#endif
__device__ __forceinline__ void row_store(u32x4 v, uint64_t addr) {
#if REC_GATHER_STORE == 0
  __builtin_nontemporal_store(v, reinterpret_cast<gdst_t>(addr));
#elif REC_GATHER_STORE == 1
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(addr), "v"(v) : "memory");
#elif REC_GATHER_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(addr), "v"(v) : "memory");
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(addr), "v"(v) : "memory");
#endif
}

// 4 blocks per CU = at most 128 VGPRs (D = 128 compiled to 129 without the bound: three waves per SIMD instead of four).
// Tried in round 2 (same box, tools/exp/gather_load_ab.sh, gather_plain_ab.sh): nontemporal row LOADS 0.680 -> 0.697 on
// uniform ids but 0.844 -> 0.738 on Zipf ids (REC_GATHER_LOAD_NT, off); destination addresses computed instead of
// shuffled for the plain concat layout (92 VGPRs, five waves per SIMD): no gain, removed.  What does move this kernel by
// 7 % is WHERE the 13.3 GB of tables were allocated: recamd.ops.place_table_arena.
template <int LPR, int IDS_F32>
__global__ __launch_bounds__(256, 4) void gather_uniform_kernel(
    TableSet ts, const void* __restrict__ ids, int64_t ids_stride, int F, int64_t R,
    float* __restrict__ out, int64_t out_stride, int* __restrict__ oob) {
  constexpr int D = LPR * 4;
  constexpr int RPI = 64 / LPR;          // rows moved per wave-instruction
  constexpr int NIT = LPR;               // 64 rows / RPI
  constexpr int U = NIT < 16 ? NIT : 16;  // loads in flight per lane (16 measured 1.4 % over 8)

  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chunk = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const int64_t r = chunk * 64 + lane;

  // src carries "row reads as zeros" in bit 0 (table rows are 16-B aligned): the load itself is
  // always issued, from a valid address, so the compiler keeps it branch-free and batched.
  uint64_t src = reinterpret_cast<uint64_t>(ts.base[0]) | 1u;
  float* dst = nullptr;
  if (r < R) {
    int64_t b;
    int f;
    if (R < (int64_t)0x7fffffff) {  // wave-uniform: 32-bit divide is much cheaper
      uint32_t b32 = (uint32_t)r / (uint32_t)F;
      b = b32;
      f = (int)((uint32_t)r - b32 * (uint32_t)F);
    } else {
      b = r / F;
      f = (int)(r - b * F);
    }
    const int32_t id = load_id<IDS_F32>(ids, b * ids_stride + f);
    dst = out + b * out_stride + ts.out_col[f];
    if ((uint32_t)id < (uint32_t)ts.vocab[f]) {
      src = reinterpret_cast<uint64_t>(ts.base[f] + (int64_t)id * D);
    } else if (oob) {
      *oob = 1;
    }
  }
  const int sub = lane / LPR;
  const int col = (lane % LPR) * 4;
  const bool full = (chunk + 1) * 64 <= R;  // wave-uniform: no tail rows in this chunk

#pragma unroll 1
  for (int it0 = 0; it0 < NIT; it0 += U) {
    u32x4 v[U];
    uint64_t d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = (it0 + u) * RPI + sub;
      const uint64_t s = shfl_u64(src, j);
      d[u] = shfl_u64(reinterpret_cast<uint64_t>(dst), j);
#if REC_GATHER_LOAD_NT
      u32x4 t = __builtin_nontemporal_load(reinterpret_cast<gsrc_t>((s & ~(uint64_t)1) + col * 4));
#else
      u32x4 t = *reinterpret_cast<gsrc_t>((s & ~(uint64_t)1) + col * 4);
#endif
      // bitwise mask (not a select, not a multiply): keeps the load unconditional and copies
      // inf/nan payloads bit-exactly; bit 0 set -> all-zero row
      const uint32_t keep = (uint32_t)(s & 1) - 1u;
      v[u] = t & keep;
    }
    if (full) {
#pragma unroll
      for (int u = 0; u < U; ++u) row_store(v[u], d[u] + col * 4);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (d[u]) row_store(v[u], d[u] + col * 4);
    }
  }
}

// Round 3 A/B (profiles/r03_gather_persist_ab.txt): a variant with PERSISTENT waves and the ids of a wave's next 64-row
// chunk prefetched before the current chunk's rows move measured 307-328 us against 307-317 us for this kernel on the same
// box (uniform ids; Zipf: 256-261 vs 248-263): the id -> row dependency is not what bounds it.  Not kept.

// Generic path: any per-field dim / alignment.  One wave per (b,f) row group; dword copies.
template <int IDS_F32>
__global__ __launch_bounds__(256) void gather_generic_kernel(
    TableSet ts, const void* __restrict__ ids, int64_t ids_stride, int F, int64_t R,
    float* __restrict__ out, int64_t out_stride, int* __restrict__ oob) {
  constexpr int ROWS_PER_WAVE = 8;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  for (int k = 0; k < ROWS_PER_WAVE; ++k) {
    const int64_t r = wave * ROWS_PER_WAVE + k;
    if (r >= R) return;
    const int64_t b = r / F;
    const int f = (int)(r - b * F);
    const int32_t id = load_id<IDS_F32>(ids, b * ids_stride + f);
    const int dim = ts.dim[f];
    float* d = out + b * out_stride + ts.out_col[f];
    const bool ok = (uint32_t)id < (uint32_t)ts.vocab[f];
    if (!ok && oob && lane == 0) *oob = 1;
    const float* s = ts.base[f] + (int64_t)(ok ? id : 0) * dim;
    for (int c = lane; c < dim; c += 64) d[c] = ok ? s[c] : 0.f;
  }
}

int fill_table_set(const rec_table_desc* tables, int32_t F, TableSet* ts, const char* who) {
  REC_CHECK_ARG(tables != nullptr, REC_EINVAL, "%s: tables is NULL", who);
  REC_CHECK_ARG(F >= 1 && F <= REC_MAX_TABLES, REC_ESHAPE, "%s: F=%d outside [1,%d]", who, F,
                REC_MAX_TABLES);
  for (int f = 0; f < F; ++f) {
    REC_CHECK_ARG(tables[f].base != nullptr, REC_EINVAL, "%s: tables[%d].base is NULL", who, f);
    REC_CHECK_ARG(tables[f].vocab >= 1 && tables[f].vocab <= 0x7fffffffLL, REC_ESHAPE,
                  "%s: tables[%d].vocab=%lld unsupported", who, f, (long long)tables[f].vocab);
    REC_CHECK_ARG(tables[f].dim >= 1, REC_ESHAPE, "%s: tables[%d].dim=%d", who, f, tables[f].dim);
    REC_CHECK_ARG(tables[f].out_col >= 0, REC_ESHAPE, "%s: tables[%d].out_col=%d", who, f,
                  tables[f].out_col);
    ts->base[f] = tables[f].base;
    ts->vocab[f] = (int32_t)tables[f].vocab;
    ts->dim[f] = tables[f].dim;
    ts->out_col[f] = tables[f].out_col;
  }
  for (int f = F; f < REC_MAX_TABLES; ++f) {
    ts->base[f] = tables[0].base;
    ts->vocab[f] = 0;
    ts->dim[f] = 0;
    ts->out_col[f] = 0;
  }
  return REC_OK;
}

template <int IDS_F32>
static int launch_gather(const TableSet& ts, int lpr, bool fast, const void* ids,
                         int64_t ids_stride, int F, int64_t R, float* out, int64_t out_stride,
                         int* oob, hipStream_t st) {
  const int64_t chunks = (R + 63) / 64;
  const int64_t blocks = (chunks + 3) / 4;
  REC_CHECK_ARG(blocks <= 0x7fffffffLL, REC_ESHAPE, "rec_gather_concat_f32: batch too large");
#define REC_LAUNCH_LPR(L)                                                                    \
  case L:                                                                                    \
    hipLaunchKernelGGL((gather_uniform_kernel<L, IDS_F32>), dim3((unsigned)blocks), dim3(256), \
                       0, st, ts, ids, ids_stride, F, R, out, out_stride, oob);              \
    break;
  if (fast) {
    switch (lpr) {
      REC_LAUNCH_LPR(1)
      REC_LAUNCH_LPR(2)
      REC_LAUNCH_LPR(4)
      REC_LAUNCH_LPR(8)
      REC_LAUNCH_LPR(16)
      REC_LAUNCH_LPR(32)
      REC_LAUNCH_LPR(64)
      default:
        fast = false;
    }
  }
#undef REC_LAUNCH_LPR
  if (!fast) {
    const int64_t waves = (R + 7) / 8;
    const int64_t gblocks = (waves + 3) / 4;
    REC_CHECK_ARG(gblocks <= 0x7fffffffLL, REC_ESHAPE, "rec_gather_concat_f32: batch too large");
    hipLaunchKernelGGL((gather_generic_kernel<IDS_F32>), dim3((unsigned)gblocks), dim3(256), 0,
                       st, ts, ids, ids_stride, F, R, out, out_stride, oob);
  }
  REC_CHECK_LAUNCH("rec_gather_concat_f32");
  return REC_OK;
}

}  // namespace rec

extern "C" int rec_gather_concat_f32(const rec_table_desc* tables, int32_t F, const void* ids,
                                     int32_t ids_dtype, int64_t ids_stride, int64_t B,
                                     float* out, int64_t out_stride, int32_t* oob_flag,
                                     void* stream) {
  using namespace rec;
  const char* who = "rec_gather_concat_f32";
  TableSet ts;
  int rc = fill_table_set(tables, F, &ts, who);
  if (rc != REC_OK) return rc;
  REC_CHECK_ARG(B == 0 || (ids != nullptr && out != nullptr), REC_EINVAL, "%s: NULL ids/out", who);
  REC_CHECK_ARG(ids_dtype == REC_IDS_I32 || ids_dtype == REC_IDS_F32, REC_EINVAL,
                "%s: bad ids_dtype %d", who, ids_dtype);
  REC_CHECK_ARG(B >= 0 && ids_stride >= F, REC_ESHAPE, "%s: B=%lld ids_stride=%lld F=%d", who,
                (long long)B, (long long)ids_stride, F);
  int64_t width = 0;
  for (int f = 0; f < F; ++f) {
    int64_t end = (int64_t)tables[f].out_col + tables[f].dim;
    if (end > width) width = end;
  }
  REC_CHECK_ARG(out_stride >= width, REC_ESHAPE, "%s: out_stride=%lld < concat width %lld", who,
                (long long)out_stride, (long long)width);
  if (B == 0) return REC_OK;

  // fast path: uniform dim, D/4 a power of two <= 64, everything 16-B aligned
  const int D = tables[0].dim;
  bool fast = (D % 4 == 0) && aligned16(out) && (out_stride % 4 == 0);
  for (int f = 0; f < F && fast; ++f)
    fast = tables[f].dim == D && aligned16(tables[f].base) && (tables[f].out_col % 4 == 0);
  const int lpr = D / 4;
  if (fast) fast = lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0;

  const int64_t R = B * F;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ids_dtype == REC_IDS_F32)
    return launch_gather<1>(ts, lpr, fast, ids, ids_stride, F, R, out, out_stride, oob_flag, st);
  return launch_gather<0>(ts, lpr, fast, ids, ids_stride, F, R, out, out_stride, oob_flag, st);
}
