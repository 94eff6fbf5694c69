// K1+K5 fused, software-pipelined across samples (D = 128, F = 26 gathered rows + the dense row).
//
// pairwise_dot.hip's register-tiled kernel alternates "27 KiB of row loads" and "~3000 VALU
// instructions" per wave; at 3 waves/SIMD the bytes in flight per CU are what bounds it (DESIGN.md
// §5).  This variant keeps the same arithmetic (half-wave per sample, 16 B of every row per lane,
// wavefront reduce-scatter) but orders the 351 pairs so that rows RETIRE one after another during
// the second part of the compute phase (csrc/gen/gen_pairdot_pipe.py); the registers of a retired
// row immediately receive the same row of the wave's NEXT sample, so most of the next tile is in
// flight while the current one is still being reduced.  Waves are persistent (grid = resident
// waves) and prefetch the next sample's ids one sample ahead.
//
// Nothing but the refill loads touches vector memory inside the compute phase (the slot -> output
// position table and the staged results live in LDS), so the compiler's in-order vmcnt waits stay
// exact.
#include <stdlib.h>

#include "common.h"
#include "pairdot_pipe_schedule.inc"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
typedef const u32x4p __attribute__((address_space(1)))* growp_t;

__device__ __forceinline__ uint64_t shfl64p(uint64_t v, int src_lane) {
  uint32_t lo = __shfl((uint32_t)v, src_lane, 64);
  uint32_t hi = __shfl((uint32_t)(v >> 32), src_lane, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <int CTRL>
__device__ __forceinline__ float dpp_movp(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}

template <int STEP>
__device__ __forceinline__ float rs_combine_p(float a, float b, int lane) {
  if constexpr (STEP == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    constexpr int CTRL = STEP == 8 ? 0x128 : STEP == 4 ? 0x141 : STEP == 2 ? 0x4E : 0xB1;
    const float ta = a + dpp_movp<CTRL>(a);
    const float tb = b + dpp_movp<CTRL>(b);
    return (lane & STEP) ? tb : ta;
  }
}

// pushes the partial of sequence slot SLOT into the binary counter; a full group of 32 slots leaves
// lane l with the sum of slot 32g + bitrev5(l), which goes to otile[outpos[that slot]]
template <int SLOT>
__device__ __forceinline__ void pd_push(float c, float (&lvl)[5], int lane, float* otile,
                                        const int* __restrict__ outpos_lds, int q_of_lane) {
  constexpr int k = SLOT % 32;
  if constexpr (k & 1) {
    c = rs_combine_p<16>(lvl[0], c, lane);
    if constexpr (k & 2) {
      c = rs_combine_p<8>(lvl[1], c, lane);
      if constexpr (k & 4) {
        c = rs_combine_p<4>(lvl[2], c, lane);
        if constexpr (k & 8) {
          c = rs_combine_p<2>(lvl[3], c, lane);
          if constexpr (k & 16) {
            c = rs_combine_p<1>(lvl[4], c, lane);
            // opaque index: without it the compiler hoists the 11 per-group LDS addresses out of the task
            // loop, spills them, and every scratch reload costs an s_waitcnt vmcnt(0) over the refills
            int idx = (SLOT - 31) + q_of_lane;
            asm volatile("" : "+v"(idx));
            const int op = outpos_lds[idx];
            if (op >= 0) otile[op] = c;
          } else {
            lvl[4] = c;
          }
        } else {
          lvl[3] = c;
        }
      } else {
        lvl[2] = c;
      }
    } else {
      lvl[1] = c;
    }
  } else {
    lvl[0] = c;
  }
}

__device__ __forceinline__ int bitrev5p(int v) {
  return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

__constant__ int kPdOutpos[PD_NSLOTS] = PD_OUTPOS_INIT;

// Out-of-range ids read this all-zero row: the refill load then needs no masking, i.e. has NO consumer
// next to it (a mask right after the load would force an s_waitcnt vmcnt(0) at every refill).
__device__ __attribute__((aligned(16))) float g_pd_zero_row[128];

// N = PD_N = 27 vectors: rows 0..25 gathered, row 26 = dense[b].
template <int IDS_F32>
__global__ __launch_bounds__(256, 3) void pairdot_pipe_kernel(TableSet ts, const void* __restrict__ ids,
                                                           int64_t ids_stride, const float* __restrict__ dense,
                                                           int64_t dense_stride, int64_t B,
                                                           float* __restrict__ out, int64_t out_stride,
                                                           int append_dense, int* __restrict__ oob) {
  constexpr int N = PD_N, F = N - 1, D = 128, P = PD_NPAIRS;
  constexpr int WP = (P + D + 3) / 4 * 4;
  __shared__ __attribute__((aligned(16))) float otile_all[4 * 2 * WP];
  __shared__ int outpos_lds[PD_NSLOTS];
  for (int e = threadIdx.x; e < PD_NSLOTS; e += blockDim.x) outpos_lds[e] = kPdOutpos[e];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int sl = lane & 31, sw = lane >> 5;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* otile = otile_all + (wave_in_block * 2 + sw) * WP;
  const int q_of_lane = bitrev5p(sl);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t ntasks = (B + 1) >> 1;  // two samples per wave-task
  int64_t task = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  if (task >= ntasks) return;  // wave-uniform

  // lane f < 26 resolves field f of its half-wave's sample; lane 26 the dense row (loop invariant parts)
  // kept in LDS, not in VGPRs: the kernel sits at the 168-VGPR edge of 3 waves/SIMD and a scratch
  // spill would be fatal here (scratch reloads count in vmcnt and would wait for every refill in flight)
  __shared__ uint64_t base_lds[256];
  __shared__ uint32_t vocab_lds[256];
  base_lds[threadIdx.x] = reinterpret_cast<uint64_t>(ts.base[sl < F ? sl : 0]);
  vocab_lds[threadIdx.x] = (uint32_t)ts.vocab[sl < F ? sl : 0];
  const uint64_t zero_row = reinterpret_cast<uint64_t>(g_pd_zero_row);

  auto sample_of = [&](int64_t t) {
    int64_t b = 2 * t + sw;
    return b < B ? b : B - 1;
  };
  auto load_my_id = [&](int64_t t) -> int32_t {
    return sl < F ? load_id<IDS_F32>(ids, sample_of(t) * ids_stride + sl) : 0;
  };
  auto resolve = [&](int64_t t, int32_t id) -> uint64_t {
    uint64_t src = zero_row;
    if (sl < F) {
      int ti = threadIdx.x;
      asm volatile("" : "+v"(ti));  // recompute the LDS address instead of keeping (and spilling) it
      if ((uint32_t)id < vocab_lds[ti]) src = base_lds[ti] + (uint64_t)(uint32_t)id * (D * 4);
      else if (oob) *oob = 1;
    } else if (sl == F) {
      src = reinterpret_cast<uint64_t>(dense + sample_of(t) * dense_stride);
    }
    return src;
  };
  auto fetch_row = [&](uint64_t src, int r) -> f32x4 {
    const uint64_t s = shfl64p(src, sw * 32 + r);
    return __builtin_bit_cast(f32x4, *reinterpret_cast<growp_t>(s + sl * 16));
  };

  f32x4 x[N];
  {
    const uint64_t src0 = resolve(task, load_my_id(task));
#pragma unroll
    for (int r = 0; r < N; ++r) x[r] = fetch_row(src0, r);
  }

  // ids are prefetched TWO tasks ahead (loop-carried), so resolving the next task's row addresses at
  // the top of a task never waits on memory
  int32_t id_nx = (task + nwaves < ntasks) ? load_my_id(task + nwaves) : 0;
  for (; task < ntasks; task += nwaves) {
    const int64_t b_raw = 2 * task + sw;
    const bool live = b_raw < B;
    const int64_t b = live ? b_raw : B - 1;
    const int64_t tnext = task + nwaves;
    const bool has_next = __builtin_amdgcn_readfirstlane((int)(tnext < ntasks)) != 0;  // wave-uniform
    uint64_t src_next = zero_row;
    if (has_next) src_next = resolve(tnext, id_nx);
    if (tnext + nwaves < ntasks) id_nx = load_my_id(tnext + nwaves);
    float lvl[5];

#define PD_PAIR(SLOT, PI, QI)                                           \
  {                                                                     \
    float c = x[PI].x * x[QI].x;                                        \
    c = fmaf(x[PI].y, x[QI].y, c);                                      \
    c = fmaf(x[PI].z, x[QI].z, c);                                      \
    c = fmaf(x[PI].w, x[QI].w, c);                                      \
    pd_push<SLOT>(c, lvl, lane, otile, outpos_lds, q_of_lane);          \
  }
#define PD_REFILL(R)                                                    \
  {                                                                     \
    if (R == N - 1 && append_dense) {                                   \
      float* od = otile + P + sl * 4;                                   \
      od[0] = x[R].x;                                                   \
      od[1] = x[R].y;                                                   \
      od[2] = x[R].z;                                                   \
      od[3] = x[R].w;                                                   \
    }                                                                   \
    x[R] = fetch_row(src_next, R); /* last task: every lane points at the zero row (L2 hit) */ \
  }
    PD_SCHEDULE
#undef PD_PAIR
#undef PD_REFILL
    // tail: slots PD_NPAIRS .. PD_NSLOTS-1 are padding (zero partials) so that the last group flushes
    pd_push<PD_NSLOTS - 1>(0.f, lvl, lane, otile, outpos_lds, q_of_lane);

    // ---- staged results -> aligned 16-B stores
    int W = P;
    if (append_dense) W = P + D;
    const int W4 = (W + 3) >> 2;
    if (sl < (W4 << 2) - W) otile[W + sl] = 0.f;
    const f32x4* t4 = reinterpret_cast<const f32x4*>(otile);
    f32x4* o4 = reinterpret_cast<f32x4*>(out + b * out_stride);
    for (int v = sl; v < W4; v += 32) {
      const f32x4 t = t4[v];
      if (live) o4[v] = t;
    }
  }
}

// returns true when the launch was issued (n == 27 with dense, D == 128, aligned padded output)
bool pairdot128_pipe_dispatch(const TableSet& ts, bool has_dense, int ids_f32, int n, const void* ids,
                              int64_t ids_stride, const float* dense, int64_t dense_stride, int64_t B, float* out,
                              int64_t out_stride, int append_dense, int* oob, hipStream_t st) {
  if (n != PD_N || !has_dense) return false;
  const int W = PD_NPAIRS + (append_dense ? 128 : 0);
  if (!(aligned16(out) && out_stride % 4 == 0 && out_stride >= (W + 3) / 4 * 4)) return false;
  const int64_t ntasks = (B + 1) / 2;
  int per_cu = 3;  // ~150 VGPRs -> 3 waves/SIMD -> 3 workgroups of 4 waves per CU
  if (const char* e = getenv("REC_PAIRDOT_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
  int64_t blocks = (ntasks + 3) / 4;
  if (blocks > (int64_t)256 * per_cu) blocks = (int64_t)256 * per_cu;
  if (ids_f32)
    hipLaunchKernelGGL((pairdot_pipe_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, ts, ids, ids_stride, dense,
                       dense_stride, B, out, out_stride, append_dense, oob);
  else
    hipLaunchKernelGGL((pairdot_pipe_kernel<0>), dim3((unsigned)blocks), dim3(256), 0, st, ts, ids, ids_stride, dense,
                       dense_stride, B, out, out_stride, append_dense, oob);
  return true;
}

}  // namespace rec
