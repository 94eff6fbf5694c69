// K5, D = 128, ONE sample per wave (64 lanes x 2 columns): the occupancy-oriented layout of the
// register-tiled pairwise-dot kernel.
//
// pairwise_dot.hip keeps a sample in a half-wave with 16 B per lane per row: 4n tile VGPRs
// (108 at n = 27) cap it at 3 waves/SIMD, and with so few waves the ~3000-instruction compute
// phase of one wave is rarely covered by another wave's loads (measured on MI355X: loads alone
// 159 us, compute alone 131 us, together 203-220 us for 65 536 x 27 x 128).  Here every lane
// holds 8 B of every row (2n = 54 tile VGPRs), so 5-6 waves/SIMD are resident and load phases
// overlap compute phases; the price is one more reduce-scatter level (v_permlane32_swap) and
// 512-B instead of 1-KiB load instructions.
//
// Results are staged in a wave-private LDS row and written as aligned 16-B vectors (needs a 16-B
// aligned `out` with out_stride % 4 == 0; otherwise the dispatcher keeps the half-wave kernel).
#include <stdlib.h>

#include "common.h"

namespace rec {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const u32x2 __attribute__((address_space(1)))* grow2_t;

// one sample per wave => row k's address is wave-uniform: v_readlane puts it in an SGPR pair and
// the load uses the scalar-base form (no address VGPRs, no ds_bpermute)
__device__ __forceinline__ uint64_t shfl64w(uint64_t v, int src_lane) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src_lane);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src_lane);
  return ((uint64_t)hi << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ float dppm(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}

// reduce-scatter level over lane distance STEP (see pairwise_dot.hip): lanes with (lane & STEP)==0
// end with `a` summed over {l, partner}, the others with `b`.
template <int STEP>
__device__ __forceinline__ float rs64(float a, float b, int lane) {
  if constexpr (STEP == 32) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if constexpr (STEP == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    constexpr int CTRL = STEP == 8 ? 0x128 : STEP == 4 ? 0x141 : STEP == 2 ? 0x4E : 0xB1;
    const float ta = a + dppm<CTRL>(a);
    const float tb = b + dppm<CTRL>(b);
    return (lane & STEP) ? tb : ta;
  }
}

__device__ __forceinline__ float rs64_step(int step, float a, float b, int lane) {
  switch (step) {
    case 32: return rs64<32>(a, b, lane);
    case 16: return rs64<16>(a, b, lane);
    case 8: return rs64<8>(a, b, lane);
    case 4: return rs64<4>(a, b, lane);
    case 2: return rs64<2>(a, b, lane);
    default: return rs64<1>(a, b, lane);
  }
}

__device__ __forceinline__ int bitrev6(int v) {
  return ((v & 1) << 5) | ((v & 2) << 3) | ((v & 4) << 1) | ((v & 8) >> 1) | ((v & 16) >> 3) | ((v & 32) >> 5);
}

template <int N, bool GATHER, bool HAS_DENSE, int IDS_F32>
__global__ __launch_bounds__(256, 6) void pairdot128_w64_kernel(
    TableSet ts, const void* __restrict__ ids, int64_t ids_stride, const float* __restrict__ xin,
    int64_t xin_stride, int64_t B, float* __restrict__ out, int64_t out_stride, int append_dense,
    int* __restrict__ oob) {
  constexpr int D = 128;
  constexpr int P = N * (N - 1) / 2;
  constexpr int F = GATHER ? (HAS_DENSE ? N - 1 : N) : 0;
  constexpr int WP = (P + D + 3) / 4 * 4;
  static_assert(N <= 64, "one lane resolves one field");

  __shared__ __attribute__((aligned(16))) float otile_all[4 * WP];
  const int lane = threadIdx.x & 63;
  const unsigned ulane = (unsigned)lane;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b = (int64_t)blockIdx.x * 4 + wave_in_block;
  if (b >= B) return;  // wave-uniform
  float* otile = otile_all + wave_in_block * WP;

  f32x2 x[N];
  if constexpr (GATHER) {
    uint64_t src = reinterpret_cast<uint64_t>(ts.base[0]) | 1u;  // bit 0: row reads as zeros
    if (lane < F) {
      const int32_t id = load_id<IDS_F32>(ids, b * ids_stride + lane);
      if ((uint32_t)id < (uint32_t)ts.vocab[lane]) {
        src = reinterpret_cast<uint64_t>(ts.base[lane] + (int64_t)id * D);
      } else if (oob) {
        *oob = 1;
      }
    }
    const bool any_bad = __ballot((src & 1) && lane < F) != 0ull;  // wave-uniform
    if (!any_bad) {
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const uint64_t s = shfl64w(src, k);
        x[k] = __builtin_bit_cast(f32x2, reinterpret_cast<grow2_t>(s)[ulane]);  // SGPR base + lane offset
      }
    } else {
#pragma unroll
      for (int k = 0; k < F; ++k) {
        const uint64_t s = shfl64w(src, k);
        u32x2 t = reinterpret_cast<grow2_t>(s & ~(uint64_t)1)[ulane];
        const uint32_t keep = (uint32_t)(s & 1) - 1u;
        x[k] = __builtin_bit_cast(f32x2, t & keep);
      }
    }
    if constexpr (HAS_DENSE) x[N - 1] = *reinterpret_cast<const f32x2*>(xin + b * xin_stride + lane * 2);
  } else {
    const float* base = xin + b * xin_stride + lane * 2;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = *reinterpret_cast<const f32x2*>(base + (int64_t)i * D);
  }

  const int q_of_lane = bitrev6(lane);
  float lvl[6];
  int p = 0;  // compile-time after full unrolling
#pragma unroll
  for (int i = 1; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) {
      float c = x[i].x * x[j].x;
      c = fmaf(x[i].y, x[j].y, c);
      const int k = p % 64;
#pragma unroll
      for (int L = 0; L < 6; ++L) {
        if ((k >> L) & 1) {
          c = rs64_step(32 >> L, lvl[L], c, lane);
        } else {
          lvl[L] = c;
          break;
        }
      }
      if (k == 63) otile[(p - k) + q_of_lane] = c;
      // keep program order: without this the scheduler hoists hundreds of independent products
      // ahead of the reductions and the live range explosion costs two waves/SIMD of occupancy
      __builtin_amdgcn_sched_barrier(0);
      ++p;
    }
  }
  if constexpr (P % 64 != 0) {
    constexpr int BASE = P - P % 64;
#pragma unroll
    for (int k = P % 64; k < 64; ++k) {
      float c = 0.f;
#pragma unroll
      for (int L = 0; L < 6; ++L) {
        if ((k >> L) & 1) {
          c = rs64_step(32 >> L, lvl[L], c, lane);
        } else {
          lvl[L] = c;
          break;
        }
      }
      if (k == 63) {
        if (q_of_lane < P % 64) otile[BASE + q_of_lane] = c;
      }
    }
  }
  int W = P;
  if (append_dense) {
    otile[P + lane * 2] = x[N - 1].x;
    otile[P + lane * 2 + 1] = x[N - 1].y;
    W = P + D;
  }
  const int W4 = (W + 3) >> 2;
  if (lane < (W4 << 2) - W) otile[W + lane] = 0.f;  // pad columns
  const f32x4* t4 = reinterpret_cast<const f32x4*>(otile);
  f32x4* o4 = reinterpret_cast<f32x4*>(out + b * out_stride);
  for (int v = lane; v < W4; v += 64) o4[v] = t4[v];
}

// returns true when the launch was issued (D = 128, staged output possible, n instantiated)
bool pairdot128_w64_dispatch(const TableSet& ts, bool gather, bool has_dense, int ids_f32, int n,
                             const void* ids, int64_t ids_stride, const float* xin,
                             int64_t xin_stride, int64_t B, float* out, int64_t out_stride,
                             int append_dense, int* oob, hipStream_t st) {
  const int P = n * (n - 1) / 2;
  const int W = P + (append_dense ? 128 : 0);
  if (!(aligned16(out) && out_stride % 4 == 0 && out_stride >= (W + 3) / 4 * 4)) return false;
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define REC_W64_N(N_)                                                                              \
  if (n == (N_)) {                                                                                 \
    if (!gather)                                                                                   \
      hipLaunchKernelGGL((pairdot128_w64_kernel<N_, false, false, 0>), grid, block, 0, st, ts, ids, \
                         ids_stride, xin, xin_stride, B, out, out_stride, 0, oob);                 \
    else if (has_dense && ids_f32)                                                                 \
      hipLaunchKernelGGL((pairdot128_w64_kernel<N_, true, true, 1>), grid, block, 0, st, ts, ids,  \
                         ids_stride, xin, xin_stride, B, out, out_stride, append_dense, oob);      \
    else if (has_dense)                                                                            \
      hipLaunchKernelGGL((pairdot128_w64_kernel<N_, true, true, 0>), grid, block, 0, st, ts, ids,  \
                         ids_stride, xin, xin_stride, B, out, out_stride, append_dense, oob);      \
    else if (ids_f32)                                                                              \
      hipLaunchKernelGGL((pairdot128_w64_kernel<N_, true, false, 1>), grid, block, 0, st, ts, ids, \
                         ids_stride, xin, xin_stride, B, out, out_stride, 0, oob);                 \
    else                                                                                           \
      hipLaunchKernelGGL((pairdot128_w64_kernel<N_, true, false, 0>), grid, block, 0, st, ts, ids, \
                         ids_stride, xin, xin_stride, B, out, out_stride, 0, oob);                 \
    return true;                                                                                   \
  }
  REC_W64_N(27)
  REC_W64_N(26)
#undef REC_W64_N
  return false;
}

}  // namespace rec
