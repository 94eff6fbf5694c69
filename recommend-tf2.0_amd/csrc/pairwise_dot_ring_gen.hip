// K1+K5 fused on the LDS-DMA ring for the shapes around the headline: D in {64, 128, 256}, any 17 <= n <= 32 rows per
// sample (n = F table rows + the dense row).  Same transport and arithmetic as pairwise_dot_ring.hip (read its header
// first: LDS-DMA rows with the XOR on the source address, Gram tiles on v_mfma_f32_16x16x4_f32 straight from LDS,
// hand-counted s_waitcnt vmcnt); that file stays the tuned, fully specialised instantiation of the BASELINE configs[1]
// shape (n = 27, D = 128) and is not touched by the generalisation.
//
// Reference op: the interaction of the paper cited at src/ctr/dlrm/model.py:7 over the rows gathered at
// src/ctr/dlrm/model.py:45 — order (i,j), i>j, row-major.
//
// What is general here
//   * a ring UNIT is RB bytes of each of the sample's rows (RB = 256 for D = 64, 512 for D = 128 / 256):
//       D = 64   one unit per sample, 4 rows per DMA instruction (16 lanes x 16 B per row), S = 4 slots per wave
//       D = 128  one unit per sample, 2 rows per DMA instruction, S = 2
//       D = 256  TWO units per sample (k = 0..127, 128..255): the Gram accumulators carry over the two halves, the
//                epilogue runs after the second — the LDS per wave, and therefore the bytes in flight per CU, stay those
//                of the D = 128 kernel (a whole 27 x 1 KiB sample per slot would not fit twice)
//     the XOR swizzle p ^ 2(R & 7) of the 16-B chunk index is conflict-free for ds_read_b128's four 16-lane groups
//     whenever the row pitch is a multiple of 256 B (MI355X_MICROARCH.md §LDS), i.e. for both RB;
//   * n is a RUN-TIME value; only the instruction COUNTS per unit are compile-time (NDMA row DMAs, NST result stores:
//     s_waitcnt needs static counts), so one instantiation serves every n with the same ceil(n / rows-per-DMA) and
//     ceil(width / 256): rows past n repeat row n-1 (never stored), store lanes past the width repeat the last group.
// Static issue order per step u (unit u of this wave; slot u % S):  [ids of unit u+S+1] [NDMA row DMAs of unit u+S]
// [NST stores of unit u, only when it is a sample's last unit].
#include <utility>

#include "common.h"
#include "ring_dma.h"

namespace rec {

__device__ __attribute__((aligned(1024))) float g_ring_gen_zero_row[256];

// compile-time unrolled `for (s = 0; s < S; ++s) if (!f(s)) break;` — the slot index selects instruction counts
template <class Fn, int... I>
__device__ __forceinline__ void static_for_slots(Fn&& f, std::integer_sequence<int, I...>) {
  (void)(f(std::integral_constant<int, I>{}) && ...);
}

template <int RB, int KH, int NDMA, int NST, int S, int WPB>
__global__ __launch_bounds__(WPB * 64, 1) void pairdot_ring_gen_kernel(
    TableSet ts, int N, int has_dense, int append, const int32_t* __restrict__ ids, int64_t ids_stride,
    const float* __restrict__ dense, int64_t dense_stride, int B, float* __restrict__ out, int64_t out_stride,
    int* __restrict__ oob_flag) {
  constexpr int CH = RB / 16;                // 16-B chunks per row piece
  constexpr int RPI = 64 / CH;               // rows per DMA instruction
  constexpr int J = RB / 64;                 // operand reads (4 k-steps each) per row piece
  constexpr int NV = (8 / RPI) > 0 ? 8 / RPI : 1;  // distinct (row & 7) patterns of a DMA lane over t
  constexpr int PER_IT = 1 + NDMA + NST;
  constexpr int SLOT = NDMA * 1024;
  constexpr int STAGE = (NST * 256 + 4) * 4;  // staged output row (<= NST * 64 groups) + dump word
  constexpr int IDB = 256;
  constexpr int WAVE_LDS = S * SLOT + STAGE + 2 * IDB;
  constexpr int DG = RB * KH;                // bytes of a whole table row
  static_assert(RB == 256 || RB == 512, "");
  static_assert(S >= 2 && S % KH == 0 && (S - 1) * PER_IT + NST <= 63, "vmcnt is a 6-bit counter");
  static_assert(NDMA <= 16, "");

  extern __shared__ __attribute__((aligned(1024))) char lds_all[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* lds_wave = lds_all + w * WAVE_LDS;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lds_wave));
  float* stage = reinterpret_cast<float*>(lds_wave + S * SLOT);
  const uint32_t idb_base = lds_base + S * SLOT + STAGE;
  const int* idb = reinterpret_cast<const int*>(lds_wave + S * SLOT + STAGE);

  const int nwaves = gridDim.x * WPB;
  const int gw = blockIdx.x * WPB + w;
  const int nk = gw < B ? (B - gw + nwaves - 1) / nwaves : 0;  // samples of this wave: b = gw + k * nwaves
  if (nk == 0) return;
  const int nu = nk * KH;                                      // units of this wave

  const int F = has_dense ? N - 1 : N;
  const int P = N * (N - 1) / 2;
  const int W = P + (append ? DG / 4 : 0);
  const int W4 = (W + 3) / 4;
  const int DUMP = NST * 256;                // one word past the largest staged row

  // ---- lane constants -------------------------------------------------------------------------
  const int sub = lane / CH, cl = lane % CH;  // DMA: this lane's row within the piece, its chunk within the row
  const int fcl = lane < F ? lane : F - 1;
  const char* my_base = reinterpret_cast<const char*>(ts.base[fcl]);
  const uint32_t my_vocab = lane < F ? (uint32_t)ts.vocab[fcl] : 0u;
  const char* zrow = reinterpret_cast<const char*>(g_ring_gen_zero_row);
  uint32_t dma_off[NV];  // source chunk of this lane for piece t, by t % NV: the XOR goes on the SOURCE address
#pragma unroll
  for (int t = 0; t < NV; ++t) dma_off[t] = (uint32_t)((cl ^ (2 * ((RPI * t + sub) & 7))) * 16);
  const int r = lane & 15, q = lane >> 4;
  const int R1 = r + 16 < N ? r + 16 : N - 1;  // rows >= N: any valid row (their products are never stored)
  const uint32_t rd0 = (uint32_t)(r * RB), rd1 = (uint32_t)(R1 * RB);
  const uint32_t sw0 = (uint32_t)(2 * (r & 7)), sw1 = (uint32_t)(2 * (R1 & 7));
  int slot00[4], slot10[4], slot11[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int I0 = 4 * q + v, I1 = 16 + 4 * q + v, J0 = r, J1 = 16 + r;
    slot00[v] = (J0 < I0) ? I0 * (I0 - 1) / 2 + J0 : DUMP;
    slot10[v] = (I1 < N) ? I1 * (I1 - 1) / 2 + J0 : DUMP;
    slot11[v] = (J1 < I1 && I1 < N) ? I1 * (I1 - 1) / 2 + J1 : DUMP;
  }
  if (lane < 4 && W + lane < W4 * 4) stage[W + lane] = 0.f;  // pad columns of the staged row
  uint32_t bad = 0;

  // ids of the sample of unit uu (clamped: a valid address always) into id buffer (sample & 1); issued once per unit,
  // i.e. KH times per sample with the same values — the counts stay static
  auto issue_ids = [&](int uu) {
    const int kk = uu / KH;
    const int kc = kk < nk ? kk : nk - 1;
    const int64_t b = (int64_t)gw + (int64_t)kc * nwaves;
    glds4(ids + b * ids_stride + fcl, idb_base + (uint32_t)((kk & 1) * IDB));
  };
  auto row_addrs = [&](int uu, uint64_t (&g)[16]) {
    const int kk = uu / KH, hh = uu % KH;
    const bool live = kk < nk;
    const int64_t b = (int64_t)gw + (int64_t)(live ? kk : 0) * nwaves;
    const uint32_t id = (uint32_t)idb[(kk & 1) * 64 + lane];
    const bool ok = id < my_vocab;
    bad |= (live && lane < F && !ok) ? 1u : 0u;
    const char* src = zrow;
    if (has_dense && live && lane == F) src = reinterpret_cast<const char*>(dense + b * dense_stride) + hh * RB;
    if (live && ok) src = my_base + (uint64_t)id * DG + hh * RB;
    const uint64_t a = reinterpret_cast<uint64_t>(src);
    const int alo = (int)(uint32_t)a, ahi = (int)(uint32_t)(a >> 32);
#pragma unroll
    for (int t = 0; t < NDMA; ++t) {
      int row = RPI * t + sub;
      row = row < N ? row : N - 1;  // the last piece's surplus lanes repeat row N-1 (lands in the slot's slack)
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(row * 4, alo);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(row * 4, ahi);
      g[t] = (((uint64_t)hi << 32) | lo) + dma_off[t % NV];
    }
  };

  // ---- prologue: ids of unit 0, then S refills ------------------------------------------------
  issue_ids(0);
#pragma unroll
  for (int p = 0; p < S; ++p) {
    if (p == 0) REC_VMCNT(0); else REC_VMCNT(NDMA);  // ids of unit p have landed
    uint64_t g[16];
    row_addrs(p, g);
    REC_LGKMCNT0();
    issue_ids(p + 1);
    glds16_burst<NDMA, 1>(g, lds_base + (uint32_t)(p * SLOT));
  }

  f32x4 a00 = {0.f, 0.f, 0.f, 0.f}, a10 = a00, a11 = a00;
  for (int u0 = 0; u0 < nu; u0 += S) {
    static_for_slots([&](auto sc) -> bool {
      constexpr int s = decltype(sc)::value;
      const int u = u0 + s;
      if (u >= nu) return false;  // wave-uniform; nu and S are multiples of KH: only at a sample boundary
      constexpr bool kFirst = (KH == 1) || (s % KH == 0);
      constexpr bool kLast = (KH == 1) || (s % KH == KH - 1);
      constexpr bool kPrevLast = (KH == 1) || ((s + KH - 1) % KH == KH - 1);  // did step u-1 issue stores?
      // one wait per step: the rows of unit u (slot s) AND the ids of unit u + S have landed.  ids(u+S) were the first
      // operation of step u-1; younger than them: rows(u+S-1) (NDMA) and the stores of step u-1 (NST, if it had any).
      if (u == 0) REC_VMCNT(NDMA);
      else if constexpr (kPrevLast) REC_VMCNT(NDMA + NST);
      else REC_VMCNT(NDMA);
      uint64_t g[16];
      row_addrs(u + S, g);
      const char* slot = lds_wave + s * SLOT;
      f32x4 x0[J], x1[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        x0[j] = *reinterpret_cast<const f32x4*>(slot + rd0 + (((uint32_t)(4 * j + q) ^ sw0) * 16));
        x1[j] = *reinterpret_cast<const f32x4*>(slot + rd1 + (((uint32_t)(4 * j + q) ^ sw1) * 16));
      }
      f32x4 dv = {0.f, 0.f, 0.f, 0.f};
      if (append)  // this unit's piece of the dense row (row N-1), chunk cl of lanes < CH
        dv = *reinterpret_cast<const f32x4*>(slot + (uint32_t)((N - 1) * RB) +
                                             (((uint32_t)cl ^ (uint32_t)(2 * ((N - 1) & 7))) * 16));

      // refill this slot with unit u + S as soon as its operands are in registers; request the ids after that
      REC_LGKMCNT0();
      issue_ids(u + S + 1);
      glds16_burst<NDMA, 1>(g, lds_base + (uint32_t)(s * SLOT));
      asm volatile("" : "+v"(x0[0]), "+v"(x1[0]));  // keep the matrix work below the burst

      if constexpr (kFirst) a00 = a10 = a11 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a00 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j][i], x0[j][i], a00, 0, 0, 0);
          a10 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j][i], x0[j][i], a10, 0, 0, 0);
          a11 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j][i], x1[j][i], a11, 0, 0, 0);
        }
      }
      if (append && lane < CH) {
        float* d = stage + P + (KH == 1 ? 0 : (s % KH) * (RB / 4)) + 4 * cl;
        d[0] = dv.x;
        d[1] = dv.y;
        d[2] = dv.z;
        d[3] = dv.w;
      }
      if constexpr (kLast) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          stage[slot00[v]] = a00[v];
          stage[slot10[v]] = a10[v];
          stage[slot11[v]] = a11[v];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t b = (int64_t)gw + (int64_t)(u / KH) * nwaves;
        f32x4* orow = reinterpret_cast<f32x4*>(out + b * out_stride);
#pragma unroll
        for (int t = 0; t < NST; ++t) {
          int gi = lane + 64 * t;
          gi = gi < W4 ? gi : W4 - 1;  // surplus lanes repeat the last group
          const f32x4 v = *reinterpret_cast<const f32x4*>(stage + 4 * gi);
          gstore16_scope<3>(orow + gi, v);  // sc0 sc1 write-through (pairwise_dot_ring.hip)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      return true;
    }, std::make_integer_sequence<int, S>{});
  }
  REC_VMCNT(0);  // the zero-row DMAs of the tail still target this wave's LDS
  if (bad && oob_flag) *oob_flag = 1;
}

template <int RB, int KH, int NDMA, int NST, int S, int WPB>
static bool launch_ring_gen(const TableSet& ts, int n, bool has_dense, bool append, const int32_t* ids,
                            int64_t ids_stride, const float* dense, int64_t dense_stride, int B, float* out,
                            int64_t out_stride, int* oob, int cus, hipStream_t st) {
  constexpr int WAVE_LDS = S * NDMA * 1024 + (NST * 256 + 4) * 4 + 512;
  constexpr int LDS = WAVE_LDS * WPB;
  static_assert(LDS <= 160 * 1024, "ring does not fit the CU's LDS");
  auto kern = pairdot_ring_gen_kernel<RB, KH, NDMA, NST, S, WPB>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) !=
        hipSuccess)
      return false;
    attr_set = true;
  }
  int64_t grid = cus;  // one block per CU, persistent waves
  const int64_t need = ((int64_t)B + WPB - 1) / WPB;
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WPB * 64), LDS, st, ts, n, has_dense ? 1 : 0, append ? 1 : 0, ids,
                     ids_stride, dense, dense_stride, B, out, out_stride, oob);
  return true;
}

// returns false when the shape is not covered (the caller falls through to the register-tiled / generic kernels)
bool pairdot_ring_gen_dispatch(const TableSet& ts, int F, int D, bool has_dense, int ids_f32, const void* ids,
                               int64_t ids_stride, const float* dense, int64_t dense_stride, int64_t B, float* out,
                               int64_t out_stride, int append_dense, int* oob, hipStream_t st) {
  if (ids_f32 || B > 0x7fffffffLL || B < 1) return false;
  if (!aligned16(out) || (out_stride & 3)) return false;
  const int n = F + (has_dense ? 1 : 0);
  if (n < 17 || n > 32 || (D != 64 && D != 128 && D != 256)) return false;
  for (int f = 0; f < F; ++f)
    if (!aligned16(ts.base[f])) return false;
  if (has_dense && (!aligned16(dense) || (dense_stride & 3))) return false;
  const bool append = has_dense && append_dense;
  const int W = n * (n - 1) / 2 + (append ? D : 0);
  const int W4 = (W + 3) / 4;
  if (out_stride < W4 * 4) return false;
  const int nst = (W4 + 63) / 64;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int32_t* ids32 = reinterpret_cast<const int32_t*>(ids);
#define REC_GEN_GO(RB_, KH_, NDMA_, NST_, S_)                                                                          \
  return launch_ring_gen<RB_, KH_, NDMA_, NST_, S_, 4>(ts, n, has_dense, append, ids32, ids_stride, dense, dense_stride, \
                                                       (int)B, out, out_stride, oob, cus, st)
  // instruction counts per unit: NDMA = ceil(n / rows per DMA), NST = ceil(width / 256 floats)
  if (D == 64) {  // 4 rows per DMA: n 17..20 -> 5, 21..24 -> 6, 25..28 -> 7, 29..32 -> 8;  width <= 560: NST 1..3
    const int nd = (n + 3) / 4;
#define REC_D64(ND_)                                       \
  if (nd == ND_) {                                         \
    if (nst == 1) REC_GEN_GO(256, 1, ND_, 1, 4);           \
    if (nst == 2) REC_GEN_GO(256, 1, ND_, 2, 4);           \
    if (nst == 3) REC_GEN_GO(256, 1, ND_, 3, 4);           \
  }
    REC_D64(5) REC_D64(6) REC_D64(7) REC_D64(8)
#undef REC_D64
    return false;
  }
  const int nd = (n + 1) / 2;  // 2 rows per DMA: 9..16
#define REC_D128(KH_, ND_)                                 \
  if (nd == ND_) {                                         \
    if (nst == 1) REC_GEN_GO(512, KH_, ND_, 1, 2);         \
    if (nst == 2) REC_GEN_GO(512, KH_, ND_, 2, 2);         \
    if (nst == 3) REC_GEN_GO(512, KH_, ND_, 3, 2);         \
  }
  if (D == 128) {
    REC_D128(1, 9) REC_D128(1, 10) REC_D128(1, 11) REC_D128(1, 12) REC_D128(1, 13) REC_D128(1, 14) REC_D128(1, 15)
    REC_D128(1, 16)
  } else {
    REC_D128(2, 9) REC_D128(2, 10) REC_D128(2, 11) REC_D128(2, 12) REC_D128(2, 13) REC_D128(2, 14) REC_D128(2, 15)
    REC_D128(2, 16)
  }
#undef REC_D128
#undef REC_GEN_GO
  return false;
}

}  // namespace rec
