// K5 (D = 128 specialisation) — DLRM pairwise dot on the fp32 matrix cores, fused with the gather.
//
// Why a second implementation: the register-tiled VALU kernel (pairwise_dot.hip) needs ~3000 VALU
// instructions per two samples; at 3 waves/SIMD mostly ONE wave per SIMD is in its compute phase,
// and one wave alone issues a VALU op only every 4 cycles, so that kernel is issue-bound
// (measured ~200 us for 65 536 x 27 x 128 whether or not the rows are gathered) instead of HBM
// bound (~190 us).  Z = X X^T is a genuinely dense 27x128x27 contraction per sample, so here it
// runs on v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain): 3 lower-triangle 16x16 tiles x
// 32 k-steps = 96 MFMAs = 3072 SIMD-cycles per sample, ~82 us for the whole batch on 1024 SIMDs,
// hidden under the HBM time, and the VALU is left with address arithmetic only.
//
// Data flow per sample (one wave = one sample, one wave per workgroup, no barriers):
//   ids -> row addresses (lane f resolves field f) -> 14 global_load_dwordx4 per lane (two 512-B
//   rows per wave-instruction, coalesced) -> ds_write_b128 into a [row][512 B] LDS image with the
//   16-B chunk index XOR-swizzled by (row & 15) -> ds_read_b128 in MFMA operand layout
//   (lane l: row l&15 of a 16-row tile, chunks 4c + (l>>4)): conflict-free on both sides ->
//   96 MFMAs -> strictly-lower-triangle extraction straight from the accumulators.
//   The same VGPR is the A operand (row i of tile I) and the B operand (column j of tile J^T), so
//   each tile is read from LDS once.  The next sample's rows are requested before the MFMA block
//   (software prefetch through the staging VGPRs), so HBM latency overlaps the matrix work.
//
// Roofline: HBM; algorithmic bytes per sample as in pairwise_dot.hip (15 844 B at F=26, dense on).
#include <stdlib.h>

#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1)))* grow_t;

__device__ __forceinline__ uint64_t shfl64m(uint64_t v, int src_lane) {
  uint32_t lo = __shfl((uint32_t)v, src_lane, 64);
  uint32_t hi = __shfl((uint32_t)(v >> 32), src_lane, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <int N, bool GATHER, bool HAS_DENSE, int IDS_F32, bool PREFETCH>
__global__ __launch_bounds__(256, PREFETCH ? 2 : 3) void pairdot128_mfma_kernel(
    TableSet ts, const void* __restrict__ ids, int64_t ids_stride, const float* __restrict__ xin,
    int64_t xin_stride, int64_t B, float* __restrict__ out, int64_t out_stride, int append_dense,
    int* __restrict__ oob) {
  constexpr int D = 128;
  constexpr int NP = (N + 1) / 2;        // row pairs = load instructions per lane
  constexpr int NR = 2 * NP;             // LDS rows (last one may be a dummy)
  constexpr int NT = N > 16 ? 2 : 1;     // 16-row tiles
  constexpr int P = N * (N - 1) / 2;
  constexpr int F = GATHER ? (HAS_DENSE ? N - 1 : N) : 0;
  static_assert(N >= 2 && N <= 32, "N in [2,32]");

  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // each wave of the workgroup owns a private [NR][32] x 16-B image: no barriers anywhere
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4* lds = reinterpret_cast<u32x4*>(lds_raw) + wave_in_block * (NR * 32);

  const int lane = threadIdx.x & 63;
  const int half = lane >> 5;
  const int sl = lane & 31;

  u32x4 stg[NP];

  // issue the (coalesced) row loads of sample b into the staging registers
  auto request = [&](int64_t b) {
    uint64_t src;
    if constexpr (GATHER) {
      src = reinterpret_cast<uint64_t>(ts.base[0]) | 1u;  // bit 0: row reads as zeros
      if (lane < F) {
        const int32_t id = load_id<IDS_F32>(ids, b * ids_stride + lane);
        if ((uint32_t)id < (uint32_t)ts.vocab[lane]) {
          src = reinterpret_cast<uint64_t>(ts.base[lane] + (int64_t)id * D);
        } else if (oob) {
          *oob = 1;
        }
      }
      if constexpr (HAS_DENSE) {
        if (lane == F) src = reinterpret_cast<uint64_t>(xin + b * xin_stride);
      }
    } else {
      src = lane < N ? reinterpret_cast<uint64_t>(xin + b * xin_stride + (int64_t)lane * D)
                     : (reinterpret_cast<uint64_t>(xin) | 1u);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const uint64_t s = shfl64m(src, 2 * p + half);  // lanes >= N hold "zero row"
      u32x4 t = *reinterpret_cast<grow_t>((s & ~(uint64_t)1) + sl * 16);
      const uint32_t keep = (uint32_t)(s & 1) - 1u;
      stg[p] = t & keep;
    }
  };

  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  if (b >= B) return;
  if constexpr (PREFETCH) request(b);

  // MFMA operand addressing (constant per lane)
  const int trow = lane & 15;
  const int g = lane >> 4;

  for (; b < B; b += nwaves) {
    if constexpr (!PREFETCH) request(b);
    // ---- staging registers -> swizzled LDS image
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int row = 2 * p + half;
      lds[row * 32 + (sl ^ (row & 15))] = stg[p];
    }
    // dense pass-through needs the last row's registers: keep a copy before they are refilled
    u32x4 dense_regs;
    if constexpr (HAS_DENSE) dense_regs = stg[(N - 1) / 2];

    // ---- LDS -> MFMA operands: lane holds row (16 I + trow), chunks 4c + g, c = 0..7
    f32x4 xa[NT][8];
#pragma unroll
    for (int I = 0; I < NT; ++I) {
      int row = 16 * I + trow;
      row = row < NR ? row : NR - 1;  // rows >= N only feed discarded outputs
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int chunk = 4 * c + g;
        xa[I][c] = __builtin_bit_cast(f32x4, lds[row * 32 + (chunk ^ (row & 15))]);
      }
    }

    // ---- prefetch the next sample while the matrix cores work
    if constexpr (PREFETCH) {
      const int64_t bn = b + nwaves;
      if (bn < B) request(bn);
    }

    // ---- Z tiles: z00 = X0 X0^T, z10 = X1 X0^T, z11 = X1 X1^T
    f32x4 z00 = {0.f, 0.f, 0.f, 0.f}, z10 = z00, z11 = z00;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = xa[0][c][e];
        z00 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, z00, 0, 0, 0);
        if constexpr (NT == 2) {
          const float a1 = xa[1][c][e];
          z10 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a0, z10, 0, 0, 0);
          z11 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, z11, 0, 0, 0);
        }
      }
    }

    // ---- epilogue: D[i][j] sits at lane (j = lane&15, i = 4*(lane>>4) + e)
    float* orow = out + b * out_stride;
    const int j0 = lane & 15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i0 = 4 * g + e;
      if (i0 < N && j0 < i0) orow[i0 * (i0 - 1) / 2 + j0] = z00[e];
      if constexpr (NT == 2) {
        const int i1 = 16 + i0;
        if (i1 < N) {
          orow[i1 * (i1 - 1) / 2 + j0] = z10[e];
          const int j1 = 16 + j0;
          if (j1 < i1) orow[i1 * (i1 - 1) / 2 + j1] = z11[e];
        }
      }
    }
    if constexpr (HAS_DENSE) {
      if (append_dense && half == ((N - 1) & 1)) {
        float* od = orow + P + sl * 4;  // only 4-B aligned in general (P odd)
        od[0] = __uint_as_float(dense_regs.x);
        od[1] = __uint_as_float(dense_regs.y);
        od[2] = __uint_as_float(dense_regs.z);
        od[3] = __uint_as_float(dense_regs.w);
      }
    }
  }
}

template <int N, bool GATHER, bool HAS_DENSE, int IDS_F32>
static int launch_mfma(const TableSet& ts, const void* ids, int64_t ids_stride, const float* xin,
                       int64_t xin_stride, int64_t B, float* out, int64_t out_stride,
                       int append_dense, int* oob, hipStream_t st) {
  constexpr int NR = 2 * ((N + 1) / 2);
  const size_t lds = (size_t)NR * 512;
  // persistent waves: as many as the LDS lets reside (160 KiB / CU), capped by the batch
  int wpb = 2;  // waves per workgroup
  if (const char* e = getenv("REC_PAIRDOT_WPB")) wpb = atoi(e) > 0 ? atoi(e) : wpb;
  int per_cu = (int)((160 * 1024) / (lds * wpb));  // workgroups per CU by LDS
  if (const char* e = getenv("REC_PAIRDOT_WAVES_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) / wpb : per_cu;
  int64_t grid = (int64_t)256 * per_cu;
  if (grid * wpb > B) grid = (B + wpb - 1) / wpb;
  const size_t lds_block = lds * wpb;
  if (lds_block > 64 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(pairdot128_mfma_kernel<N, GATHER, HAS_DENSE, IDS_F32, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
    hipFuncSetAttribute(reinterpret_cast<const void*>(pairdot128_mfma_kernel<N, GATHER, HAS_DENSE, IDS_F32, false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_block);
  }
  const char* pf = getenv("REC_PAIRDOT_PREFETCH");
  if (pf && pf[0] == '1')
    hipLaunchKernelGGL((pairdot128_mfma_kernel<N, GATHER, HAS_DENSE, IDS_F32, true>), dim3((unsigned)grid),
                       dim3(64 * wpb), lds_block, st, ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,
                       append_dense, oob);
  else
    hipLaunchKernelGGL((pairdot128_mfma_kernel<N, GATHER, HAS_DENSE, IDS_F32, false>), dim3((unsigned)grid),
                       dim3(64 * wpb), lds_block, st, ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,
                       append_dense, oob);
  return 0;
}

// returns true when a D=128 MFMA instantiation exists for n and the launch was issued
bool pairdot128_mfma_dispatch(const TableSet& ts, bool gather, bool has_dense, int ids_f32, int n,
                              const void* ids, int64_t ids_stride, const float* xin,
                              int64_t xin_stride, int64_t B, float* out, int64_t out_stride,
                              int append_dense, int* oob, hipStream_t st) {
#define REC_MFMA_N(N_)                                                                            \
  if (n == (N_)) {                                                                                \
    if (!gather)                                                                                  \
      launch_mfma<N_, false, false, 0>(ts, ids, ids_stride, xin, xin_stride, B, out, out_stride, \
                                       0, oob, st);                                               \
    else if (has_dense && ids_f32)                                                                \
      launch_mfma<N_, true, true, 1>(ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,   \
                                     append_dense, oob, st);                                      \
    else if (has_dense)                                                                           \
      launch_mfma<N_, true, true, 0>(ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,   \
                                     append_dense, oob, st);                                      \
    else if (ids_f32)                                                                             \
      launch_mfma<N_, true, false, 1>(ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,  \
                                      0, oob, st);                                                \
    else                                                                                          \
      launch_mfma<N_, true, false, 0>(ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,  \
                                      0, oob, st);                                                \
    return true;                                                                                  \
  }
  REC_MFMA_N(27)
  REC_MFMA_N(26)
  REC_MFMA_N(9)
  REC_MFMA_N(4)
#undef REC_MFMA_N
  return false;
}

}  // namespace rec
