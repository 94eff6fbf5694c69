// K1+K5 fused for D = 128 — the headline kernel (BASELINE configs[1]): embedding gather + DLRM pairwise dot with the
// sample tiles streamed through a per-wave LDS ring by LDS-DMA and the Gram matrix taken on the fp32 matrix cores.
//
// Reference op: the interaction of the paper cited at src/ctr/dlrm/model.py:7 over the rows gathered at
// src/ctr/dlrm/model.py:45 (Z = X X^T, strictly-lower triangle, order (i,j), i>j, row-major; see pairwise_dot.hip).
//
// Why this shape (round-1 findings): the register-tiled VALU kernel (pairwise_dot.hip) needs ~1100
// VALU wave-instructions per sample and holds every tile in VGPRs, so a wave that computes has nothing in flight and
// the CU's bytes in flight sag whenever waves compute together (201-213 us against 159 us for the loads alone).  Here
//   * a sample's 27 rows (13.5 KiB) travel HBM -> LDS by global_load_lds_dwordx4 (no VGPR destination): every ring slot
//     is in flight again as soon as its rows have been copied to registers, independent of what the wave computes;
//     a 512-B row is fetched by ONE half-wave instruction (one DRAM page visit per row);
//   * the arithmetic is 96 v_mfma_f32_16x16x4_f32 per sample (tiles (0,0), (1,0), (1,1) of the 32x32 Gram matrix,
//     32 k-steps each) on operands read from LDS in MFMA layout — zero VALU for the 351 x 128 multiply-adds and no
//     cross-lane reduction.  The f32 MFMA is bit-for-bit a k-ordered fmaf chain (exact fp32 products, one rounding per
//     step): +-inf / NaN / denormals behave as in fp32 arithmetic (no bf16 split);
//   * MFMA pipe demand: 96 x 32 cycles = 3072 cycles per sample per SIMD = 82 us of 4 busy SIMDs per CU at 2.4 GHz
//     for the whole 65 536-sample batch, against >= 159 us of HBM time: the matrix pipe runs beside the DMA stream.
//
// LDS image of a slot: row R at byte R*512, its 16-B chunk p stored at chunk position p ^ 2(R & 7) — the XOR goes on
// the per-lane SOURCE address of the DMA (an LDS-DMA destination is lane-linear), and makes the operand reads
// (lane (r = l & 15, q = l >> 4) reads chunk 4j + q of rows r and r + 16) conflict-free ds_read_b128.
//
// Every vector-memory operation of the loop is issued by inline asm and counted by hand (s_waitcnt vmcnt(N) retires
// them in issue order): per iteration  [1 id DMA for the sample S+1 ahead] [NDMA row DMAs for the sample S ahead]
// [NST output stores].  Samples past a wave's share are replaced by a zero row (L2-resident), never skipped, so
// the counts are static.  The ids themselves arrive by DMA too (64 x 4 B), which keeps the id -> address -> row
// dependency chain one iteration ahead of its use and free of register-destination hazards.
#include <stdlib.h>

#include "common.h"
#include "ring_dma.h"

namespace rec {

__device__ __attribute__((aligned(512))) float g_ring_zero_row[128];

// POL: bit 0 = streaming row loads, bit 1 = plain (not nt) result stores; experiment builds: bits 2-4 = store scope
// arm, bits 5-7 = load policy arm
constexpr int load_mode(int pol) { return (pol >> 5) ? (pol >> 5) + 1 : (pol & 1); }

// N rows per sample (F table rows + the dense row if HAS_DENSE), S ring slots per wave, WPB waves per block.
template <int N, bool HAS_DENSE, bool APPEND, int S, int WPB, int ABL, int POL>
__global__ __launch_bounds__(WPB * 64, 1) void pairdot_ring_kernel(
    TableSet ts, const int32_t* __restrict__ ids, int64_t ids_stride, const float* __restrict__ dense,
    int64_t dense_stride, int B, float* __restrict__ out, int64_t out_stride, int* __restrict__ oob_flag) {
  constexpr int F = HAS_DENSE ? N - 1 : N;
  constexpr int P = N * (N - 1) / 2;
  constexpr int W = P + (APPEND ? 128 : 0);
  constexpr int W4 = (W + 3) / 4;            // 16-B groups per output row
  constexpr int NST = (ABL & 2) ? 0 : (W4 + 63) / 64;  // store instructions per sample (ABL: experiments)
  constexpr int NDMA = (N + 1) / 2;          // row DMAs per sample (two rows each)
  constexpr int PER_IT = 1 + NDMA + NST;     // vector-memory ops per loop iteration
  constexpr int SLOT = NDMA * 1024;          // bytes per ring slot
  constexpr int STAGE = (W4 * 4 + 4) * 4;    // staged output row + a dump word for masked accumulator entries
  constexpr int IDB = 256;                   // one id DMA
  constexpr int WAVE_LDS = S * SLOT + STAGE + 2 * IDB;
  static_assert(N >= 17 && N <= 32, "two 16-row tiles");
  static_assert(S >= 1 && S <= 3 && (S - 1) * PER_IT + NST <= 63, "vmcnt is a 6-bit counter");

  extern __shared__ __attribute__((aligned(1024))) char lds_all[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* lds_wave = lds_all + w * WAVE_LDS;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lds_wave));
  float* stage = reinterpret_cast<float*>(lds_wave + S * SLOT);
  const uint32_t idb_base = lds_base + S * SLOT + STAGE;
  const int* idb = reinterpret_cast<const int*>(lds_wave + S * SLOT + STAGE);

  const int nwaves = gridDim.x * WPB;
  const int gw = blockIdx.x * WPB + w;            // wave-uniform
  const int nk = gw < B ? (B - gw + nwaves - 1) / nwaves : 0;  // samples of this wave: b = gw + k * nwaves
  if (nk == 0) return;

  // ---- lane constants -------------------------------------------------------------------------
  const int h = lane >> 5, c32 = lane & 31;
  // row-address resolution: lane f < F owns field f
  const int fcl = lane < F ? lane : F - 1;
  const char* my_base = reinterpret_cast<const char*>(ts.base[fcl]);
  const uint32_t my_vocab = lane < F ? (uint32_t)ts.vocab[fcl] : 0u;
  const char* zrow = reinterpret_cast<const char*>(g_ring_zero_row);
  // DMA t moves rows 2t (lanes < 32) and 2t + 1: this lane's source chunk within its row, by t & 3
  uint32_t dma_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) dma_off[t] = (uint32_t)((c32 ^ (2 * ((2 * t + h) & 7))) * 16);
  // operand reads: lane (r, q) reads chunk 4j + q of rows r and r + 16
  const int r = lane & 15, q = lane >> 4;
  const int R1 = r + 16 < N ? r + 16 : N - 1;  // rows >= N: any valid row (their products are never stored)
  const uint32_t rd0 = (uint32_t)(r * 512), rd1 = (uint32_t)(R1 * 512);
  const uint32_t sw0 = (uint32_t)(2 * (r & 7)), sw1 = (uint32_t)(2 * (R1 & 7));
  // epilogue: accumulator register v of lane (c = lane & 15, g = lane >> 4) is Z[I][J], I = 4g + v (+16), J = c (+16)
  constexpr int DUMP = W4 * 4;
  int slot00[4], slot10[4], slot11[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int I0 = 4 * q + v, I1 = 16 + 4 * q + v, J0 = r, J1 = 16 + r;
    slot00[v] = (J0 < I0) ? I0 * (I0 - 1) / 2 + J0 : DUMP;
    slot10[v] = (I1 < N) ? I1 * (I1 - 1) / 2 + J0 : DUMP;
    slot11[v] = (J1 < I1 && I1 < N) ? I1 * (I1 - 1) / 2 + J1 : DUMP;
  }
  if (lane < 4) stage[W + lane < DUMP + 4 ? W + lane : DUMP] = 0.f;  // pad columns of the staged row
  uint32_t bad = 0;

  // id DMA for sample index kk (clamped: a valid address always) into id buffer kk & 1
  auto issue_ids = [&](int kk) {
    const int kc = kk < nk ? kk : nk - 1;
    const int64_t b = (int64_t)gw + (int64_t)kc * nwaves;
    glds4(ids + b * ids_stride + fcl, idb_base + (uint32_t)((kk & 1) * IDB));
  };
  // source addresses of the row pieces of sample kk (its ids must have landed in id buffer kk & 1): lane f resolves
  // field f (range check, zero row for a bad id or a sample past the end), then piece t's lanes fetch the addresses
  // of rows 2t / 2t + 1 from lanes 2t / 2t + 1 (ds_bpermute, all issued together)
  auto row_addrs = [&](int kk, uint64_t (&g)[16]) {
    const bool live = kk < nk;
    const int64_t b = (int64_t)gw + (int64_t)(live ? kk : 0) * nwaves;
    const uint32_t id = (uint32_t)idb[(kk & 1) * 64 + lane];
    const bool ok = id < my_vocab;
    bad |= (live && lane < F && !ok) ? 1u : 0u;
    const char* src = zrow;
    if (HAS_DENSE) src = (live && lane == F) ? reinterpret_cast<const char*>(dense + b * dense_stride) : src;
    src = (live && ok) ? my_base + ((uint64_t)id << 9) : src;
    const uint64_t a = reinterpret_cast<uint64_t>(src);
    const int alo = (int)(uint32_t)a, ahi = (int)(uint32_t)(a >> 32);
#pragma unroll
    for (int t = 0; t < NDMA; ++t) {
      int row = 2 * t + h;
      row = row < N ? row : N - 1;  // odd N: the last piece's upper half repeats row N-1 (lands in the slot's slack)
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(row * 4, alo);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(row * 4, ahi);
      g[t] = (((uint64_t)hi << 32) | lo) + dma_off[t & 3];
    }
  };

  // ---- prologue: ids of sample 0, then S refills ----------------------------------------------
  issue_ids(0);
#pragma unroll
  for (int p = 0; p < S; ++p) {
    if (p == 0) REC_VMCNT(0); else REC_VMCNT(NDMA);  // ids of sample p have landed
    uint64_t g[16];
    row_addrs(p, g);
    REC_LGKMCNT0();
    issue_ids(p + 1);
    glds16_burst<NDMA, load_mode(POL)>(g, lds_base + (uint32_t)(p * SLOT));
  }

  for (int k0 = 0; k0 < nk; k0 += S) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int k = k0 + s;
      if (k >= nk) break;  // wave-uniform
      // One wait per step: the rows of sample k (slot s) AND the ids of sample k + S have landed.  Issue order is
      // [ids(j+S+1)] [rows(j+S) x NDMA] [stores(j) x NST] per step j (prologue steps: no stores), so the ids of
      // k + S are older than all but the NDMA + NST youngest operations (NDMA before the first stores exist), and
      // for S >= 2 the rows of sample k are older still; for S = 1 they are the youngest but NST.
      if (S == 1) {
        if (k == 0) REC_VMCNT(0); else REC_VMCNT(NST);
      } else {
        if (k == 0) REC_VMCNT(NDMA); else REC_VMCNT(NDMA + NST);
      }
      uint64_t g[16];
      row_addrs(k + S, g);
      const char* slot = lds_wave + s * SLOT;
      f32x4 x0[8], x1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        x0[j] = *reinterpret_cast<const f32x4*>(slot + rd0 + (((uint32_t)(4 * j + q) ^ sw0) * 16));
        x1[j] = *reinterpret_cast<const f32x4*>(slot + rd1 + (((uint32_t)(4 * j + q) ^ sw1) * 16));
      }
      f32x4 dv = {0.f, 0.f, 0.f, 0.f};
      if constexpr (APPEND)
        dv = *reinterpret_cast<const f32x4*>(slot + (N - 1) * 512 + ((c32 ^ (2 * ((N - 1) & 7))) * 16));

      // refill this slot with sample k + S as soon as its operands are in registers; request the ids after that
      REC_LGKMCNT0();
      issue_ids(k + S + 1);
      glds16_burst<NDMA, load_mode(POL)>(g, lds_base + (uint32_t)(s * SLOT));
      // keep the matrix work below the burst (an MFMA is a register-only instruction: nothing else orders it)
      asm volatile("" : "+v"(x0[0]), "+v"(x1[0]));

      f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a10 = a00, a11 = a00;
      if constexpr (ABL & 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a00 += x0[j] + x1[j];
      } else
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j][i], x0[j][i], a00, 0, 0, 0);
          a10 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j][i], x0[j][i], a10, 0, 0, 0);
          a11 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j][i], x1[j][i], a11, 0, 0, 0);
        }
      }

#pragma unroll
      for (int v = 0; v < 4; ++v) {
        stage[slot00[v]] = a00[v];
        stage[slot10[v]] = a10[v];
        stage[slot11[v]] = a11[v];
      }
      if constexpr (APPEND) {
        if (lane < 32) {
          stage[P + 4 * c32 + 0] = dv.x;
          stage[P + 4 * c32 + 1] = dv.y;
          stage[P + 4 * c32 + 2] = dv.z;
          stage[P + 4 * c32 + 3] = dv.w;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int64_t b = (int64_t)gw + (int64_t)k * nwaves;
      f32x4* orow = reinterpret_cast<f32x4*>(out + b * out_stride);
#pragma unroll
      for (int u = 0; u < NST; ++u) {
        int g = lane + 64 * u;
        g = g < W4 ? g : W4 - 1;  // surplus lanes repeat the last group
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * g);
        if constexpr (((POL >> 2) & 7) != 0) gstore16_scope<((POL >> 2) & 7)>(orow + g, v);
        else if constexpr (POL & 2) gstore16(orow + g, v);
        else gstore16_nt(orow + g, v);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  REC_VMCNT(0);  // the zero-row DMAs of the tail still target this wave's LDS
  if (bad && oob_flag) *oob_flag = 1;
}

template <int N, bool HAS_DENSE, bool APPEND, int S, int WPB, int ABL = 0, int POL = 15>
static bool launch_ring(const TableSet& ts, const int32_t* ids, int64_t ids_stride, const float* dense,
                        int64_t dense_stride, int B, float* out, int64_t out_stride, int* oob, int blocks_per_cu,
                        int cus, hipStream_t st) {
  constexpr int P = N * (N - 1) / 2;
  constexpr int W4 = (P + (APPEND ? 128 : 0) + 3) / 4;
  constexpr int NDMA = (N + 1) / 2;
  constexpr int WAVE_LDS = S * NDMA * 1024 + (W4 * 4 + 4) * 4 + 512;
  constexpr int LDS = WAVE_LDS * WPB;
  auto kern = pairdot_ring_kernel<N, HAS_DENSE, APPEND, S, WPB, ABL, POL>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess)
      return false;
    attr_set = true;
  }
  int64_t grid = (int64_t)cus * blocks_per_cu;
  const int64_t need = ((int64_t)B + WPB - 1) / WPB;
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WPB * 64), LDS, st, ts, ids, ids_stride, dense, dense_stride, B,
                     out, out_stride, oob);
  return true;
}

// returns false when the shape is not covered (the caller falls through to the register-tiled kernel)
bool pairdot128_ring_dispatch(const TableSet& ts, int F, bool has_dense, int ids_f32, const void* ids,
                              int64_t ids_stride, const float* dense, int64_t dense_stride, int64_t B, float* out,
                              int64_t out_stride, int append_dense, int* oob, hipStream_t st) {
  if (ids_f32 || B > 0x7fffffffLL || B < 1) return false;
  if (!aligned16(out) || (out_stride & 3)) return false;
  const int n = F + (has_dense ? 1 : 0);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int32_t* ids32 = reinterpret_cast<const int32_t*>(ids);
  const int P = n * (n - 1) / 2;
  if (out_stride < (P + (append_dense ? 128 : 0) + 3) / 4 * 4) return false;
  // Ring geometry (measured on MI355X, 65 536 x 27 x 128, rotating id batches, after spin-up; tools/exp/ring_ab.py,
  // profiles/r02_ring_ab.txt): S = 2 slots x 4 waves (one per SIMD), one 122-KiB block per CU, streaming (nt) row
  // loads + default-policy stores = 181-183 us (0.71 of the 8 TB/s roofline; with sc0 sc1 write-through stores
  // 168-176 us = 0.74-0.77, shipped); nt stores 192; default loads 198;
  // 3 slots x 3 waves 193-195; 1 slot x 8 waves 215; the register-tiled kernel 197-200.
#define REC_RING_GO(N_, HD_, AP_, S_, W_, A_, P_, BPC_)                                                             \
  return launch_ring<N_, HD_, AP_, S_, W_, A_, P_>(ts, ids32, ids_stride, dense, dense_stride, (int)B, out, out_stride, \
                                                   oob, BPC_, cus, st)
#ifdef REC_RING_EXPERIMENTS  // A/B builds only (EXTRA_HIPFLAGS=-DREC_RING_EXPERIMENTS): geometry / policy / ablations
  if (n == 27 && has_dense && append_dense) {
    int cfg = 0;
    if (const char* e = getenv("REC_RING_CFG")) cfg = atoi(e);
    switch (cfg) {
      case 1: REC_RING_GO(27, true, true, 1, 4, 0, 3, 2);
      case 2: REC_RING_GO(27, true, true, 3, 3, 0, 3, 1);
      case 6: REC_RING_GO(27, true, true, 1, 2, 0, 3, 4);
      case 7: REC_RING_GO(27, true, true, 2, 4, 0, 1, 1);
      case 8: REC_RING_GO(27, true, true, 2, 4, 0, 0, 1);
      case 9: REC_RING_GO(27, true, true, 2, 4, 0, 2, 1);
      case 40: REC_RING_GO(27, true, true, 2, 4, 0, 3, 1);            // stores plain (round-2 first version)
      case 41: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 1, 1);    // stores sc0
      case 42: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 2, 1);    // stores sc1
      case 43: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 3, 1);    // stores sc0 sc1
      case 44: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 4, 1);    // stores sc1 nt
      case 45: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 5, 1);    // stores sc0 sc1 nt
      case 51: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 3 + 32 * 1, 1);    // stores sc0 sc1, loads sc1 nt
      case 52: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 3 + 32 * 2, 1);    // stores sc0 sc1, loads sc0 sc1 nt
      case 53: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 3 + 32 * 3, 1);    // stores sc0 sc1, loads sc1
      case 54: REC_RING_GO(27, true, true, 2, 4, 0, 3 + 4 * 3 + 32 * 4, 1);    // stores sc0 sc1, loads sc0 sc1
      case 10: REC_RING_GO(27, true, true, 2, 4, 1, 15, 1);
      case 20: REC_RING_GO(27, true, true, 2, 4, 2, 15, 1);
      case 30: REC_RING_GO(27, true, true, 2, 4, 3, 15, 1);
      default: break;
    }
  }
#endif
  if (n == 27 && has_dense && append_dense) REC_RING_GO(27, true, true, 2, 4, 0, 15, 1);
  if (n == 27 && has_dense && !append_dense) REC_RING_GO(27, true, false, 2, 4, 0, 15, 1);
  if (n == 26 && !has_dense) REC_RING_GO(26, false, false, 2, 4, 0, 15, 1);
#undef REC_RING_GO
  return false;
}

}  // namespace rec
