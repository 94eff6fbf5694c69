// K5 — DLRM pairwise-dot interaction, standalone and fused with the embedding gather (K1+K5).
//
// The reference's DLRM.call (src/ctr/dlrm/model.py:42-54) only concatenates; the interaction is
// the one of the paper its header cites (src/ctr/dlrm/model.py:7):  Z = X X^T over the n = F+1
// vectors of a sample, strictly-lower triangle, order (i,j), i>j, row-major.
//
// Roofline: HBM.  Fused form, per sample (F=26, D=128): 26*512 B rows + 104 B ids + 512 B dense
// read, (351+128)*4 B written = 15 844 B; 89 856 flop (5.7 flop/B, far below the 19.7 flop/B
// ridge) -> the gathered tile never goes back to HBM (the materialised concat costs 26 728 B).
//
// Design (gfx950, wave = 64):
//   * LPR = D/4 lanes own one sample (D=128: a half-wave per sample, two samples per wave).  Each
//     lane loads 16 B of every one of the n rows straight into VGPRs (global_load_dwordx4, n
//     loads in flight per lane = 13.8 KiB per sample in flight) — register-staged tile: the
//     dots below need no LDS at all, so LDS stays free and occupancy is VGPR-bound only.
//   * every pair (i,j) costs 4 v_fma per lane on the lane's 4 columns; the LPR partial sums of
//     LPR consecutive pairs are then reduced with a wavefront *reduce-scatter* (permlane16_swap /
//     DPP row_ror / row_half_mirror / quad_perm butterflies): log2(LPR) levels, each level halves
//     the number of live values, so the cost is ~2.3 VALU per pair instead of 5 shuffles+adds,
//     and lane l ends up owning pair (32*g + bitrev(l)) -> one coalesced 128-B store per group.
//   * the reduction is streamed with a compile-time "binary counter" (value k is merged at level
//     L when bit L of k is set), so only log2(LPR) partials are live next to the 4n tile VGPRs.
//   * fp32 accumulation order differs from a k-ordered dot (columns are split over lanes, then
//     tree-reduced); parity tolerance is 1e-5 relative, stated in tests/test_pairwise_dot_gpu.py.
#include <stdlib.h>

#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const f32x4 __attribute__((address_space(1)))* grow_t;
#ifndef REC_PAIRDOT_PACKED
#define REC_PAIRDOT_PACKED 0
#endif

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src_lane) {
  uint32_t lo = __shfl((uint32_t)v, src_lane, 64);
  uint32_t hi = __shfl((uint32_t)(v >> 32), src_lane, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}

// One reduce-scatter level over lane distance STEP.  `a` is the earlier value of the pair, `b`
// the later one; on return lanes with (lane & STEP) == 0 hold a summed over {l, partner(l)} and
// the other lanes hold b summed likewise.
template <int STEP>
__device__ __forceinline__ float rs_combine(float a, float b, int lane) {
  if constexpr (STEP == 32) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if constexpr (STEP == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    constexpr int CTRL = STEP == 8 ? 0x128      /* row_ror:8            partner l^8 */
                         : STEP == 4 ? 0x141    /* row_half_mirror      partner l^7 */
                         : STEP == 2 ? 0x4E     /* quad_perm:[2,3,0,1]  partner l^2 */
                                     : 0xB1;    /* quad_perm:[1,0,3,2]  partner l^1 */
    const float ta = a + dpp_mov<CTRL>(a);
    const float tb = b + dpp_mov<CTRL>(b);
    return (lane & STEP) ? tb : ta;
  }
}

// `step` is a compile-time constant after full unrolling; the switch folds to one case.
__device__ __forceinline__ float rs_combine_step(int step, float a, float b, int lane) {
  switch (step) {
    case 32: return rs_combine<32>(a, b, lane);
    case 16: return rs_combine<16>(a, b, lane);
    case 8: return rs_combine<8>(a, b, lane);
    case 4: return rs_combine<4>(a, b, lane);
    case 2: return rs_combine<2>(a, b, lane);
    default: return rs_combine<1>(a, b, lane);
  }
}

template <int N>
constexpr int ilog2() { return N <= 1 ? 0 : 1 + ilog2<N / 2>(); }

template <int BITS>
__device__ __forceinline__ int bitrev(int v) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < BITS; ++i) r |= ((v >> i) & 1) << (BITS - 1 - i);
  return r;
}

// Row sources -------------------------------------------------------------------------------
// plain: X is a materialised (B, n, D) tensor
struct PlainSrc {
  const float* x;
  int n;
};

// LPR lanes per sample, N vectors per sample (compile time), GATHER: rows 0..F-1 come from the
// tables through ids, row F (if HAS_DENSE) from `dense`.
#ifndef REC_PAIRDOT_MIN_WAVES
#define REC_PAIRDOT_MIN_WAVES 1
#endif
template <int LPR, int N, bool GATHER, bool HAS_DENSE, int IDS_F32, bool OUT_LDS>
__global__ __launch_bounds__(256, REC_PAIRDOT_MIN_WAVES) void pairdot_kernel(
    TableSet ts, const void* __restrict__ ids, int64_t ids_stride, const float* __restrict__ xin,
    int64_t xin_stride /* plain: sample stride; gather: dense stride */, int64_t B,
    float* __restrict__ out, int64_t out_stride, int append_dense, int* __restrict__ oob) {
  constexpr int D = LPR * 4;
  constexpr int SPW = 64 / LPR;  // samples per wave
  constexpr int LOG = ilog2<LPR>();
  constexpr int P = N * (N - 1) / 2;
  constexpr int F = GATHER ? (HAS_DENSE ? N - 1 : N) : 0;

  const int ids_f32 = IDS_F32 | (append_dense >> 16);  // runtime ids dtype (bit 16), keeps the
  append_dense &= 0xffff;                               // instantiation count (and build time) down
  const int lane = threadIdx.x & 63;
  const int sl = lane % LPR;          // lane within the sample
  const int sw = lane / LPR;          // sample within the wave
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const int64_t b_raw = wave * SPW + sw;
  const bool live = b_raw < B;
  const int64_t b = live ? b_raw : B - 1;  // dead lanes recompute the last sample, never store

  f32x4 x[N];
  if constexpr (GATHER) {
    // resolve this sample's F row addresses LPR fields at a time, then broadcast lane->sample
#pragma unroll
    for (int f0 = 0; f0 < F; f0 += LPR) {
      const int f = f0 + sl;
      uint64_t src = reinterpret_cast<uint64_t>(ts.base[0]) | 1u;  // bit 0: row reads as zeros
      if (f < F) {
        const int32_t id = ids_f32 ? load_id<1>(ids, b * ids_stride + f) : load_id<0>(ids, b * ids_stride + f);
        if ((uint32_t)id < (uint32_t)ts.vocab[f]) {
          src = reinterpret_cast<uint64_t>(ts.base[f] + (int64_t)id * D);
        } else if (oob) {
          *oob = 1;
        }
      }
#pragma unroll
      for (int k = 0; k < LPR; ++k) {
        if (f0 + k < F) {
          const uint64_t s = shfl64(src, sw * LPR + k);
          f32x4 t = *reinterpret_cast<grow_t>((s & ~(uint64_t)1) + sl * 16);
          const uint32_t keep = (uint32_t)(s & 1) - 1u;
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          u32x4 tb = __builtin_bit_cast(u32x4, t) & keep;
          x[f0 + k] = __builtin_bit_cast(f32x4, tb);
        }
      }
    }
    if constexpr (HAS_DENSE) {
      x[N - 1] = *reinterpret_cast<const f32x4*>(xin + b * xin_stride + sl * 4);
    }
  } else {
    const float* base = xin + b * xin_stride + sl * 4;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = *reinterpret_cast<const f32x4*>(base + (int64_t)i * D);
  }

  float* orow = out + b * out_stride;
  const int q_of_lane = bitrev<LOG>(sl);  // pair (within a group of LPR) this lane ends up with

  // OUT_LDS: the P (+D) results of a sample are staged in a wave-private LDS row and leave as
  // 16-B aligned, fully coalesced global_store_dwordx4 (needs a 16-B aligned `out` and
  // out_stride % 4 == 0, i.e. a padded row stride such as 480 for 479 columns).  The direct path
  // below scatters 4-B stores at 4-B alignment; on MI355X those cost ~45 us per 65 536 samples
  // (measured by ablation), the staged form ~free.
  constexpr int WP = (P + D + 3) / 4 * 4;  // padded row width in LDS (floats)
  __shared__ __attribute__((aligned(16))) float otile_all[OUT_LDS ? 4 * SPW * WP : 4];
  float* otile = otile_all + (OUT_LDS ? (wave_in_block * SPW + sw) * WP : 0);

  float lvl[LOG > 0 ? LOG : 1];
  int p = 0;  // compile-time after full unrolling
#pragma unroll
  for (int i = 1; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) {
#if REC_PAIRDOT_PACKED
      // two v_pk_*_f32 + one add instead of four scalar FMAs: a wave that computes alone on its
      // SIMD issues one VALU op per 4 cycles either way, the packed ops carry twice the work
      f32x2 c2 = x[i].xy * x[j].xy;
      c2 = __builtin_elementwise_fma(x[i].zw, x[j].zw, c2);
      float c = c2.x + c2.y;
#else
      float c = x[i].x * x[j].x;
      c = fmaf(x[i].y, x[j].y, c);
      c = fmaf(x[i].z, x[j].z, c);
      c = fmaf(x[i].w, x[j].w, c);
#endif
      const int k = p % LPR;
#pragma unroll
      for (int L = 0; L < LOG; ++L) {
        if ((k >> L) & 1) {
          c = rs_combine_step(LPR >> (L + 1), lvl[L], c, lane);
        } else {
          lvl[L] = c;
          break;
        }
      }
      if (k == LPR - 1) {  // group complete: lane sl owns pair (p - k) + bitrev(sl)
        if constexpr (OUT_LDS) otile[(p - k) + q_of_lane] = c;
        else if (live) orow[(p - k) + q_of_lane] = c;
      }
      ++p;
    }
  }
  // tail group: pad with zeros up to a full group so the counter flushes
  if constexpr (P % LPR != 0) {
    constexpr int BASE = P - P % LPR;
#pragma unroll
    for (int k = P % LPR; k < LPR; ++k) {
      float c = 0.f;
#pragma unroll
      for (int L = 0; L < LOG; ++L) {
        if ((k >> L) & 1) {
          c = rs_combine_step(LPR >> (L + 1), lvl[L], c, lane);
        } else {
          lvl[L] = c;
          break;
        }
      }
      if (k == LPR - 1) {
        if constexpr (OUT_LDS) {
          if (q_of_lane < P % LPR) otile[BASE + q_of_lane] = c;
        } else if (live && q_of_lane < P % LPR) {
          orow[BASE + q_of_lane] = c;
        }
      }
    }
  }
  if constexpr (OUT_LDS) {
    int W = P;  // columns of this launch
    if (append_dense) {
      float* od = otile + P + sl * 4;  // P is odd in general: 4-B aligned LDS writes
      od[0] = x[N - 1].x;
      od[1] = x[N - 1].y;
      od[2] = x[N - 1].z;
      od[3] = x[N - 1].w;
      W = P + D;
    }
    const int W4 = (W + 3) >> 2;                     // float4 per row, pad columns written as 0
    if (sl < (W4 << 2) - W) otile[W + sl] = 0.f;
    // wave-private tile: LDS ops of one wave execute in order, no barrier needed
    const f32x4* t4 = reinterpret_cast<const f32x4*>(otile);
    f32x4* o4 = reinterpret_cast<f32x4*>(orow);
    for (int v = sl; v < W4; v += LPR) {
      const f32x4 t = t4[v];
      if (live) o4[v] = t;
    }
  } else if (append_dense && live) {
    float* od = orow + P + sl * 4;  // row start is only 4-B aligned in general (P odd)
    od[0] = x[N - 1].x;
    od[1] = x[N - 1].y;
    od[2] = x[N - 1].z;
    od[3] = x[N - 1].w;
  }
}

// Generic fallback (any n, D): one wave per sample, rows staged in LDS, one pair per lane step.
__global__ __launch_bounds__(256) void pairdot_generic_kernel(const float* __restrict__ x, int64_t B,
                                                              int n, int D, float* __restrict__ out,
                                                              int64_t out_stride) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  float* tile = lds + (size_t)w * n * (D + 1);
  if (b < B) {
    for (int e = lane; e < n * D; e += 64) tile[(e / D) * (D + 1) + e % D] = x[b * (int64_t)n * D + e];
  }
  __syncthreads();
  if (b >= B) return;
  const int P = n * (n - 1) / 2;
  for (int p = lane; p < P; p += 64) {
    int i = (int)((1.f + sqrtf(1.f + 8.f * (float)p)) * 0.5f);
    while (i * (i - 1) / 2 > p) --i;
    while ((i + 1) * i / 2 <= p) ++i;
    const int j = p - i * (i - 1) / 2;
    const float* xi = tile + i * (D + 1);
    const float* xj = tile + j * (D + 1);
    float acc = 0.f;
    for (int k = 0; k < D; ++k) acc = fmaf(xi[k], xj[k], acc);
    out[b * out_stride + p] = acc;
  }
}

// Generic fused fallback (any n <= 64 rows, any D): one wave per sample, the gathered rows staged in LDS.
__global__ __launch_bounds__(256) void pairdot_generic_gather_kernel(TableSet ts, int F, const void* __restrict__ ids,
                                                                     int ids_f32, int64_t ids_stride,
                                                                     const float* __restrict__ dense,
                                                                     int64_t dense_stride, int64_t B, int D,
                                                                     float* __restrict__ out, int64_t out_stride,
                                                                     int append_dense, int* __restrict__ oob) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  const int n = F + (dense ? 1 : 0);
  float* tile = lds + (size_t)w * n * (D + 1);
  if (b < B) {
    for (int i = 0; i < n; ++i) {
      const float* src = nullptr;
      if (i < F) {
        const int32_t id = ids_f32 ? load_id<1>(ids, b * ids_stride + i) : load_id<0>(ids, b * ids_stride + i);
        if ((uint32_t)id < (uint32_t)ts.vocab[i]) src = ts.base[i] + (int64_t)id * D;
        else if (oob && lane == 0) *oob = 1;
      } else {
        src = dense + b * dense_stride;
      }
      for (int c = lane; c < D; c += 64) tile[i * (D + 1) + c] = src ? src[c] : 0.f;
    }
  }
  __syncthreads();
  if (b >= B) return;
  const int P = n * (n - 1) / 2;
  for (int p = lane; p < P; p += 64) {
    int i = (int)((1.f + sqrtf(1.f + 8.f * (float)p)) * 0.5f);
    while (i * (i - 1) / 2 > p) --i;
    while ((i + 1) * i / 2 <= p) ++i;
    const int j = p - i * (i - 1) / 2;
    const float* xi = tile + i * (D + 1);
    const float* xj = tile + j * (D + 1);
    float acc = 0.f;
    for (int k = 0; k < D; ++k) acc = fmaf(xi[k], xj[k], acc);
    out[b * out_stride + p] = acc;
  }
  if (append_dense)
    for (int c = lane; c < D; c += 64) out[b * out_stride + P + c] = tile[(n - 1) * (D + 1) + c];
}

int fill_table_set(const rec_table_desc* tables, int32_t F, TableSet* ts, const char* who);
// D = 128 fused form on the LDS-DMA ring + fp32 matrix cores (pairwise_dot_ring.hip): the default for the shapes it
// covers (int32 ids, 16-B aligned padded output rows).  rec_debug_force("pairdot", "v") keeps the register-tiled
// kernel of this file for A/B measurements.  Earlier variants that lost (LDS-transposed fp32 MFMA, software-pipelined
// VALU, bf16x3 Gram) live under tools/exp/pairdot_variants/ and are not part of the library.
bool pairdot128_ring_dispatch(const TableSet& ts, int F, bool has_dense, int ids_f32, const void* ids,
                              int64_t ids_stride, const float* dense, int64_t dense_stride, int64_t B, float* out,
                              int64_t out_stride, int append_dense, int* oob, hipStream_t st);
// the same transport for the shapes around it: D in {64, 128, 256}, 17 <= n <= 32 (pairwise_dot_ring_gen.hip)
bool pairdot_ring_gen_dispatch(const TableSet& ts, int F, int D, bool has_dense, int ids_f32, const void* ids,
                               int64_t ids_stride, const float* dense, int64_t dense_stride, int64_t B, float* out,
                               int64_t out_stride, int append_dense, int* oob, hipStream_t st);
static bool use_ring() {  // rec_debug_force("pairdot", "v") keeps the register-tiled kernel (tests / A/B only)
  const char* e = forced("pairdot");
  return !(e && e[0] == 'v');
}

template <int LPR, int N, bool GATHER, bool HAS_DENSE, int IDS_F32>
static void launch_pairdot(const TableSet& ts, const void* ids, int64_t ids_stride, const float* xin,
                           int64_t xin_stride, int64_t B, float* out, int64_t out_stride,
                           int append_dense, int* oob, hipStream_t st) {
  constexpr int SPW = 64 / LPR;
  const int64_t waves = (B + SPW - 1) / SPW;
  const int64_t blocks = (waves + 3) / 4;
  const int W = N * (N - 1) / 2 + ((append_dense & 0xffff) ? LPR * 4 : 0);
  const bool staged = aligned16(out) && out_stride % 4 == 0 && out_stride >= (W + 3) / 4 * 4;
  constexpr int tpb = 256;  // 64 / 128 / 256 threads per workgroup measured equal (212 / 206 / 210 us)
  const int64_t nblk = blocks;
  if (staged)
    hipLaunchKernelGGL((pairdot_kernel<LPR, N, GATHER, HAS_DENSE, IDS_F32, true>), dim3((unsigned)nblk),
                       dim3(tpb), 0, st, ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,
                       append_dense, oob);
  else
    hipLaunchKernelGGL((pairdot_kernel<LPR, N, GATHER, HAS_DENSE, IDS_F32, false>), dim3((unsigned)nblk),
                       dim3(tpb), 0, st, ts, ids, ids_stride, xin, xin_stride, B, out, out_stride,
                       append_dense, oob);
}

}  // namespace rec

// (LPR, N) instantiations of the register-tiled kernel.  N = F+1 = 27 is the DLRM/Criteo shape;
// the small ones serve tests and narrower models; everything else takes the generic kernel
// (plain) or is rejected (fused).
#define REC_PAIRDOT_SHAPES(X) \
  X(32, 27) X(32, 26) X(32, 9) X(32, 4) X(16, 27) X(16, 9) X(8, 9) X(4, 27) X(4, 5)

extern "C" int rec_pairwise_dot_f32(const float* x, int64_t B, int32_t n, int32_t D, float* out,
                                    int64_t out_stride, void* stream) {
  using namespace rec;
  const char* who = "rec_pairwise_dot_f32";
  REC_CHECK_ARG(B >= 0 && n >= 1 && D >= 1, REC_ESHAPE, "%s: B=%lld n=%d D=%d", who, (long long)B, n, D);
  const int64_t P = (int64_t)n * (n - 1) / 2;
  if (B == 0 || P == 0) return REC_OK;  // nothing to write (n == 1 has no pairs)
  REC_CHECK_ARG(x && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(out_stride >= P, REC_ESHAPE, "%s: out_stride=%lld < P=%lld", who,
                (long long)out_stride, (long long)P);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TableSet ts{};
  const bool vec_ok = D % 4 == 0 && aligned16(x);
#define REC_TRY(LPR_, N_)                                                                         \
  if (vec_ok && D == (LPR_)*4 && n == (N_)) {                                                     \
    launch_pairdot<LPR_, N_, false, false, 0>(ts, nullptr, 0, x, (int64_t)n * D, B, out, out_stride, \
                                              0, nullptr, st);                                    \
    REC_CHECK_LAUNCH(who);                                                                        \
    return REC_OK;                                                                                \
  }
  REC_PAIRDOT_SHAPES(REC_TRY)
#undef REC_TRY
  const size_t lds = (size_t)4 * n * (D + 1) * sizeof(float);
  REC_CHECK_ARG(lds <= 160 * 1024, REC_ESHAPE, "%s: n=%d D=%d tile does not fit LDS", who, n, D);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pairdot_generic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
  }
  const int64_t blocks = (B + 3) / 4;
  hipLaunchKernelGGL(pairdot_generic_kernel, dim3((unsigned)blocks), dim3(256), lds, st, x, B, n, D,
                     out, out_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_pairwise_dot_f32(const rec_table_desc* tables, int32_t F, const void* ids,
                                           int32_t ids_dtype, int64_t ids_stride,
                                           const float* dense, int64_t dense_stride, int64_t B,
                                           float* out, int64_t out_stride, int32_t append_dense,
                                           int32_t* oob_flag, void* stream) {
  using namespace rec;
  const char* who = "rec_gather_pairwise_dot_f32";
  TableSet ts;
  int rc = fill_table_set(tables, F, &ts, who);
  if (rc != REC_OK) return rc;
  REC_CHECK_ARG(B == 0 || (ids && out), REC_EINVAL, "%s: NULL ids/out", who);
  REC_CHECK_ARG(ids_dtype == REC_IDS_I32 || ids_dtype == REC_IDS_F32, REC_EINVAL,
                "%s: bad ids_dtype %d", who, ids_dtype);
  REC_CHECK_ARG(B >= 0 && ids_stride >= F, REC_ESHAPE, "%s: B=%lld ids_stride=%lld F=%d", who,
                (long long)B, (long long)ids_stride, F);
  const int D = tables[0].dim;
  for (int f = 0; f < F; ++f) {
    REC_CHECK_ARG(tables[f].dim == D, REC_ESHAPE, "%s: tables must share one dim (got %d vs %d)",
                  who, tables[f].dim, D);
    REC_CHECK_ARG(aligned16(tables[f].base), REC_EINVAL, "%s: tables[%d].base not 16-B aligned", who, f);
  }
  const int n = F + (dense ? 1 : 0);
  const int64_t P = (int64_t)n * (n - 1) / 2;
  REC_CHECK_ARG(!append_dense || dense, REC_EINVAL, "%s: append_dense without dense", who);
  REC_CHECK_ARG(out_stride >= P + (append_dense ? D : 0), REC_ESHAPE, "%s: out_stride too small", who);
  if (dense) {
    REC_CHECK_ARG(aligned16(dense) && dense_stride % 4 == 0 && dense_stride >= D, REC_EINVAL,
                  "%s: dense must be 16-B aligned with stride %% 4 == 0", who);
  }
  if (B == 0) return REC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (D == 128 && use_ring() &&
      pairdot128_ring_dispatch(ts, F, dense != nullptr, ids_dtype == REC_IDS_F32, ids, ids_stride, dense,
                               dense_stride, B, out, out_stride, append_dense, oob_flag, st)) {
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
  if (use_ring() && pairdot_ring_gen_dispatch(ts, F, D, dense != nullptr, ids_dtype == REC_IDS_F32, ids, ids_stride, dense,
                                              dense_stride, B, out, out_stride, append_dense, oob_flag, st)) {
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
#define REC_TRY(LPR_, N_)                                                                          \
  if (D == (LPR_)*4 && n == (N_)) {                                                                \
    const int flags = (ids_dtype == REC_IDS_F32) ? (1 << 16) : 0;                                  \
    if (dense)                                                                                     \
      launch_pairdot<LPR_, N_, true, true, 0>(ts, ids, ids_stride, dense, dense_stride, B, out,    \
                                              out_stride, append_dense | flags, oob_flag, st);     \
    else                                                                                           \
      launch_pairdot<LPR_, N_, true, false, 0>(ts, ids, ids_stride, nullptr, 0, B, out,            \
                                               out_stride, flags, oob_flag, st);                   \
    REC_CHECK_LAUNCH(who);                                                                         \
    return REC_OK;                                                                                 \
  }
  REC_PAIRDOT_SHAPES(REC_TRY)
#undef REC_TRY
  {  // any other (D, n): generic LDS-staged kernel (correct for every shape the tile fits, not tuned)
    const size_t lds = (size_t)4 * n * (D + 1) * sizeof(float);
    REC_CHECK_ARG(lds <= 160 * 1024, REC_ESHAPE, "%s: n=%d D=%d tile does not fit LDS", who, n, D);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pairdot_generic_gather_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(pairdot_generic_gather_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), lds, st, ts, F, ids,
                       ids_dtype == REC_IDS_F32 ? 1 : 0, ids_stride, dense, dense_stride, B, D, out, out_stride,
                       append_dense ? 1 : 0, oob_flag);
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
}
