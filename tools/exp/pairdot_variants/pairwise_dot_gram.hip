// K5 on the bf16 matrix cores with fp32 accuracy ("bf16x3"): fused gather + pairwise dot for D = 128, n <= 32.
//
// STATUS: opt-in (REC_PAIRDOT_IMPL=gram), parity-green, 215-226 us at 65 536 x 27 x 128 against 201-213 us for
// the shipped register-tiled VALU kernel.  Ablation of THIS kernel (T=2 tiles/wave, 2 waves/SIMD; build with
// -DREC_GRAM_ABL=n, tools/exp/run_abl.sh): loads only 158 us; + split/MFMA 204 us; + LDS staging 214 us; + global
// stores 226 us; loads + staging + stores without compute 193 us.  The compute phase is the expensive part: hipcc
// emits each k-step as a block of ~45 split VALU ops followed by 6 dependent MFMAs, so a wave's own VALU and MFMA
// work does not overlap (~3000 cycles per sample instead of ~1600), and with 2 waves per SIMD nothing else hides it.
// Next: interleave splits and MFMAs by hand (sched_group_barrier), 4 instead of 6 MFMAs per k-step
// (S += HH^T, MM^T;  T += HM^T, HL^T;  Z = S + T + T^T in the epilogue), loader/consumer wave roles.
//
// Why try it: the VALU kernel (pairwise_dot.hip) spends ~1200 VALU instructions per sample on the 351 x 128 FMAs
// + the wavefront reduce-scatter.  fp32 MFMA (pairwise_dot_mfma.hip) is too slow to help (256 flop/clk/CU).  The
// bf16 MFMA is 16x faster, and an fp32 value splits EXACTLY into three bf16 terms x = h + m + l (8 + 8 + 8
// mantissa bits, by truncation), each product of two bf16 is exact in the fp32 accumulator, so
//     X X^T = H H^T + (H M^T + M H^T) + (H L^T + L H^T) + M M^T  + O(2^-24 |x_i||x_j|)
// has the same error as fp32 arithmetic (tests: 1e-5 against the fp64 oracle, same as the VALU kernel).
//
// Layout: ONE wave per sample.  v_mfma_f32_32x32x16_bf16 wants lane (r = lane&31, h = lane>>5) to hold row r,
// k-slice 8h..8h+7 for both A and B, and for a Gram matrix A and B are the same registers.  So lane (r,h) loads
// its own row straight from HBM: k-step s reads floats [16s + 8h, +8) = two 16-B loads; the two lanes of a row
// cover 64 contiguous bytes per k-step (measured 5.7 TB/s, within 4 % of fully coalesced rows; tools/exp/).
// No LDS transposition, no cross-lane traffic on the way in; row r < F is table r's row ids[b, r], row F the
// dense vector, rows >= n a zero row.
//
// Pipelining: waves are persistent and hold T = 2 samples in registers; when the last k-step of a sample has been
// split, the wave issues that slot's NEXT sample as one burst of 16 loads (+ its dense row, + the id of the
// sample after), then stages its results and turns to the other slot, whose rows have been arriving meanwhile.
// Three measured facts shape the loop (tools/exp/inflight.hip, run_inflight.py):
//  * a row's pieces must be requested close together in time: refilling k-step by k-step as registers free up
//    reads at 4.9 TB/s, whole-sample bursts at 5.6-5.8 TB/s (one DRAM page visit per 512-B row instead of eight);
//  * 8 samples in flight per CU already saturate the loads (158 us); more tiles in flight buy nothing;
//  * every vector-memory op in the loop must be unconditional straight-line code, or hipcc's waitcnt insertion
//    falls back to vmcnt(0) and a slot's k-step wait also waits for the other slot's fresh burst; stores in
//    flight make every counted wait over-wait (they complete out of order with loads), so output rows are staged
//    8 per wave and flushed together.
//
// Epilogue: the 32x32 accumulator tile has column j on the lane and row i in the register, so each lane writes
// its strictly-lower-triangle entries to their output slot i(i-1)/2 + j in a wave-private LDS row, the dense
// passthrough is appended, and the row leaves as aligned nontemporal 16-B stores.
#include "bf16x3.h"
#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gp4;

__device__ __attribute__((aligned(512))) float g_gram_zero_row[128];

// T tiles (samples) per wave in registers, W waves per SIMD: 2 x 2 keeps 16 tiles per CU in flight with
// 256 registers per wave (no spill risk); a wave computes one tile while its other tile is still arriving.
#ifndef REC_GRAM_TILES
#define REC_GRAM_TILES 2
#endif
#ifndef REC_GRAM_MIN_WAVES
#define REC_GRAM_MIN_WAVES 2
#endif
#ifndef REC_GRAM_NT
#define REC_GRAM_NT 1
#endif
#ifndef REC_GRAM_ABL
#define REC_GRAM_ABL 0  // experiments: 1 no global stores, 2 no split/MFMA, 4 no staging
#endif
constexpr int GRAM_STAGE = 3904;  // floats of output staging per wave (4 waves: 62 464 B of LDS per block)
#ifndef REC_GRAM_BATCH
#define REC_GRAM_BATCH 8  // output rows staged per wave before one flush (capped by GRAM_STAGE / row size)
#endif

// 8 fp32 (two float4) -> three bf16x8 fragments with x = h + m + l exactly (bf16x3.h)
__device__ __forceinline__ void split8(const f32x4 a0, const f32x4 a1, bf16x8& H, bf16x8& M, bf16x8& L) {
  const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  u32x4 h, m, l;
  bf16x3::split8(x, h, m, l);
  H = __builtin_bit_cast(bf16x8, h);
  M = __builtin_bit_cast(bf16x8, m);
  L = __builtin_bit_cast(bf16x8, l);
}

// Every vector-memory operation in the loop is unconditional and in straight-line code (ids and rows of samples
// past the end are clamped / pointed at the zero row, surplus store lanes repeat the row's last 16 B), so hipcc's
// s_waitcnt insertion can count exactly: waiting for one slot's k-step must not wait for the other slot's burst.
// (With exec-masked loads/stores in branches it falls back to vmcnt(0) and the double buffer is lost.)
template <int IDS_F32, int T, int WAVES, int NR, bool HAS_DENSE, bool APPEND>
__global__ __launch_bounds__(256, WAVES) void pairdot_gram_kernel(
    TableSet ts, int F, const void* __restrict__ ids, int64_t ids_stride, const float* __restrict__ dense,
    int64_t dense_stride, int b_begin, int B, int iters, float* __restrict__ out, int64_t out_stride,
    int* __restrict__ oob_flag) {
  // samples of this launch: b_begin + wave + (t + T*k) * nwaves for k < iters, t < T; the host makes that range
  // rectangular (a second one-shot launch takes the remainder), so the loop body has no exits
  __shared__ __attribute__((aligned(16))) float stage_all[4][GRAM_STAGE];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  float* stage0 = stage_all[w];
  const int n = F + (HAS_DENSE ? 1 : 0);
  const int P = n * (n - 1) / 2;
  const int W = P + (APPEND ? 128 : 0);
  const int W4 = (W + 3) >> 2;  // 16-B groups per output row; pad columns are written as 0
  const int nwaves = gridDim.x * 4;
  const int b0 = b_begin + blockIdx.x * 4 + w;
  if (b0 >= B) return;
  // R output rows of SR floats are staged per wave and flushed together: every store in flight makes the
  // s_waitcnt vmcnt(N) of the load pipeline over-wait (stores complete out of order with loads, so they cannot be
  // counted as "younger"), which at the end of a slot's k-loop means waiting for the other slot's fresh burst.
  // One flush per R samples pays that once instead of R times.
  const int SR = W4 * 4 + 4;  // last float of a row = dummy slot for the masked-out accumulator entries
  const int R = GRAM_STAGE / SR < REC_GRAM_BATCH ? GRAM_STAGE / SR : REC_GRAM_BATCH;
  for (int k = 0; k < R; ++k)
    if (lane < 4) stage0[k * SR + W + lane] = 0.f;  // pad columns (never overwritten)

  // lane constants: my row's table
  const float* base = nullptr;
  uint32_t vocab = 0;
  for (int f = 0; f < F; ++f)
    if (r == f) {
      base = ts.base[f];
      vocab = (uint32_t)ts.vocab[f];
    }
  const float* zrow = g_gram_zero_row;
  const bool is_tab = r < F, is_dense = HAS_DENSE && r == F;
  const int rF = r < F ? r : F - 1;
  uint32_t bad = 0;

  auto ldid = [&](int bb) -> int32_t {  // unconditional: clamped to a valid element
    const int bc = bb < B ? bb : B - 1;
    return load_id<IDS_F32>(ids, (int64_t)bc * ids_stride + rF);
  };
  auto resolve = [&](int bb, int32_t id) -> gp4 {  // branch-free
    const bool live = bb < B;
    const bool ok = (uint32_t)id < vocab;
    bad |= (live && is_tab && !ok) ? 1u : 0u;
    const uintptr_t p_tab = (uintptr_t)base + ((uintptr_t)(uint32_t)id << 9);
    const uintptr_t p_dense = (uintptr_t)dense + (uintptr_t)((int64_t)bb * dense_stride * 4);
    uintptr_t p = (uintptr_t)zrow;
    p = (live && is_dense) ? p_dense : p;
    p = (live && is_tab && ok) ? p_tab : p;
    return (gp4)(p + h * 32);
  };

  // slot t holds the wave's samples j = t + T*k, k = 0, 1, ...; sample j is b0 + j * nwaves (giving a wave a
  // contiguous run of samples instead measured 5 % slower)
  auto sample = [&](int j) -> int { return b0 + j * nwaves; };
  const int jmax = iters * T;
  f32x4 x[T][16];
  f32x4 dvn[T];  // the dense row (for the passthrough columns) travels with the tile's burst
  int32_t idn[T];
  auto dense_ptr = [&](int bb) -> gp4 {
    const uintptr_t p = bb < B ? (uintptr_t)dense + (uintptr_t)((int64_t)bb * dense_stride * 4) : (uintptr_t)zrow;
    return (gp4)(p + r * 16);
  };
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int bt = sample(t);
    gp4 pc = resolve(bt, ldid(bt));
    idn[t] = ldid(t + T < jmax ? sample(t + T) : B);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      x[t][2 * s] = pc[4 * s];
      x[t][2 * s + 1] = pc[4 * s + 1];
    }
    if constexpr (APPEND) dvn[t] = *dense_ptr(bt);
  }

  auto flush_row = [&](const float* srow, int b) {
    f32x4* orow = reinterpret_cast<f32x4*>(out + (int64_t)b * out_stride);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      int q = lane + 64 * k;
      q = q < W4 ? q : W4 - 1;  // surplus lanes repeat the last group
      const f32x4 v = *reinterpret_cast<const f32x4*>(srow + 4 * q);
      if (REC_GRAM_ABL & 1) {
        if (v.x == 12345.678f) orow[q] = v;
      } else if (REC_GRAM_NT)
        __builtin_nontemporal_store(v, orow + q);
      else
        orow[q] = v;
    }
  };

  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int j = it * T + t;
      const int b = sample(j);  // wave-uniform, < B by construction
      const int rr = j % R;
      float* stage = stage0 + rr * SR;
      if constexpr (APPEND) {  // both half-waves write the same values
        stage[P + 4 * r + 0] = dvn[t].x;
        stage[P + 4 * r + 1] = dvn[t].y;
        stage[P + 4 * r + 2] = dvn[t].z;
        stage[P + 4 * r + 3] = dvn[t].w;
      }

      f32x16 acc;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.f;
      if (REC_GRAM_ABL & 2) {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[s] = x[t][s].x + x[t][s].y + x[t][s].z + x[t][s].w;
      } else
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x8 H, M, L;
        split8(x[t][2 * s], x[t][2 * s + 1], H, M, L);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, H, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, M, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(M, H, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, L, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(L, H, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(M, M, acc, 0, 0, 0);
      }
      // Burst refill of this slot: all 16 loads of its next sample back to back.  Issuing a row's 64-B pieces
      // close together in time is worth 15 % of HBM throughput over refilling k-step by k-step
      // (tools/exp/inflight.hip: 162 us vs 188 us loads-only): one DRAM page visit per 512-B row, not eight.
      // The id of the sample after that is loaded BEFORE the burst, so by the time this slot's rows have
      // arrived (in-order return) the id has too.
      auto refill = [&]() {
        const int bnext = j + T < jmax ? sample(j + T) : B;  // B = "none": zero row
        gp4 pn = resolve(bnext, idn[t]);
        __builtin_amdgcn_sched_barrier(0);
        idn[t] = ldid(j + 2 * T < jmax ? sample(j + 2 * T) : B);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          x[t][2 * s] = pn[4 * s];
          x[t][2 * s + 1] = pn[4 * s + 1];
        }
        if constexpr (APPEND) dvn[t] = *dense_ptr(bnext);
        __builtin_amdgcn_sched_barrier(0);
      };
      refill();

      // accumulator register k of lane (r, h) is Z[i][j], i = (k&3) + 8(k>>2) + 4h, j = r
      if (REC_GRAM_ABL & 4) {
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tsum += acc[k];
        if (tsum == 12345.678f) stage[lane] = tsum;
      } else
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = (k & 3) + 8 * (k >> 2) + 4 * h;
        const int slot = (r < i && i < n) ? i * (i - 1) / 2 + r : SR - 1;  // dummy slot: no branches
        stage[slot] = acc[k];
      }
      if (rr == R - 1) {  // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = 0; k < R; ++k) flush_row(stage0 + k * SR, sample(j - (R - 1) + k));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  {  // rows staged since the last full batch
    const int left = jmax % R;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int k = 0; k < left; ++k) flush_row(stage0 + k * SR, sample(jmax - left + k));
  }
  if (bad && oob_flag) *oob_flag = 1;
}

// One-shot form (REC_GRAM_CFG=os; 231 us, i.e. slower than the persistent form's 215-226 us: 16 one-sample waves per
// CU each pay the descriptor -> id -> row dependency chain before their burst starts): one wave per sample, no loop.  The wave resolves its row, issues the sample as one burst, runs
// the k-steps as the pieces arrive, stages and stores its output row and exits -- stores are the last thing a wave
// does, so they never sit in front of a load wait, and with no loop-carried state the kernel fits 4 waves per SIMD.
template <int IDS_F32, int NR, bool HAS_DENSE, bool APPEND>
__global__ __launch_bounds__(256, 4) void pairdot_gram_oneshot_kernel(
    TableSet ts, int F, const void* __restrict__ ids, int64_t ids_stride, const float* __restrict__ dense,
    int64_t dense_stride, int B, float* __restrict__ out, int64_t out_stride, int* __restrict__ oob_flag) {
  __shared__ __attribute__((aligned(16))) float stage_all[4][640];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
  float* stage = stage_all[w];
  const int n = F + (HAS_DENSE ? 1 : 0);
  const int P = n * (n - 1) / 2;
  const int W = P + (APPEND ? 128 : 0);
  const int W4 = (W + 3) >> 2;
  const int b = blockIdx.x * 4 + w;
  if (b >= B) return;
  if (lane < 4) stage[W + lane] = 0.f;

  const bool is_tab = r < F, is_dense = HAS_DENSE && r == F;
  const int rF = r < F ? r : F - 1;
  const float* base = ts.base[rF];
  const uint32_t vocab = is_tab ? (uint32_t)ts.vocab[rF] : 0u;
  const int32_t id = load_id<IDS_F32>(ids, (int64_t)b * ids_stride + rF);
  const bool ok = (uint32_t)id < vocab;
  uintptr_t p = (uintptr_t)g_gram_zero_row;
  p = is_dense ? (uintptr_t)dense + (uintptr_t)((int64_t)b * dense_stride * 4) : p;
  p = (is_tab && ok) ? (uintptr_t)base + ((uintptr_t)(uint32_t)id << 9) : p;
  gp4 pc = (gp4)(p + h * 32);
  f32x4 x[16];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    x[2 * s] = pc[4 * s];
    x[2 * s + 1] = pc[4 * s + 1];
  }
  f32x4 dv = {0.f, 0.f, 0.f, 0.f};
  if constexpr (APPEND) dv = *(gp4)(uintptr_t)(dense + (int64_t)b * dense_stride + r * 4);
  __builtin_amdgcn_sched_barrier(0);  // keep the burst whole: hipcc otherwise sinks loads between the MFMAs

  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bf16x8 H, M, L;
    split8(x[2 * s], x[2 * s + 1], H, M, L);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, H, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, M, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(M, H, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, L, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(L, H, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(M, M, acc, 0, 0, 0);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = (k & 3) + 8 * (k >> 2) + 4 * h;
    const int slot = (r < i && i < n) ? i * (i - 1) / 2 + r : 639;
    stage[slot] = acc[k];
  }
  if constexpr (APPEND) {
    stage[P + 4 * r + 0] = dv.x;
    stage[P + 4 * r + 1] = dv.y;
    stage[P + 4 * r + 2] = dv.z;
    stage[P + 4 * r + 3] = dv.w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  f32x4* orow = reinterpret_cast<f32x4*>(out + (int64_t)b * out_stride);
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    int q = lane + 64 * k;
    q = q < W4 ? q : W4 - 1;
    __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(stage + 4 * q), orow + q);
  }
  if (is_tab && !ok && oob_flag) *oob_flag = 1;
}

// returns false when the shape is not covered (caller falls through to the VALU kernels)
bool pairdot128_gram_dispatch(const TableSet& ts, int F, bool has_dense, int ids_f32, const void* ids,
                              int64_t ids_stride, const float* dense, int64_t dense_stride, int64_t B, float* out,
                              int64_t out_stride, int append_dense, int* oob, hipStream_t st) {
  const int n = F + (has_dense ? 1 : 0);
  if (n < 2 || n > 32 || B > 0x7fffffffLL) return false;
  if (!aligned16(out) || (out_stride & 3)) return false;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int tiles = REC_GRAM_TILES, per_cu = REC_GRAM_MIN_WAVES;  // blocks of 4 waves per CU = waves per SIMD
  if (const char* e = getenv("REC_GRAM_CFG")) {  // "<tiles><waves>", A/B measurements only: 13, 22
    if (e[0] == '1' && e[1] == '3') tiles = 1, per_cu = 3;
    if (e[0] == '2' && e[1] == '2') tiles = 2, per_cu = 2;
    if (e[0] == 'o') tiles = 0;  // one-shot waves
  }
  const int P = n * (n - 1) / 2;
  const bool append = append_dense != 0;
  const int W4 = (P + (append ? 128 : 0) + 3) / 4;
  const int nr = (W4 + 63) / 64;  // 1..3
  // persistent part: a rectangle of iters x tiles x (4 * grid) samples; remainder: one sample per wave
  const int64_t grid_main = (int64_t)cus * per_cu;
  const int64_t chunk = grid_main * 4 * (tiles > 0 ? tiles : 1);
  const int iters = (int)(B / chunk);
  const int B_main = (int)(iters * chunk);
  const int64_t grid_rem = (B - B_main + 3) / 4;
#define REC_GRAM_GO(I_, T_, W_, NR_, HD_, AP_)                                                                      \
  do {                                                                                                              \
    if (main_part)                                                                                                  \
      hipLaunchKernelGGL((pairdot_gram_kernel<I_, T_, W_, NR_, HD_, AP_>), dim3((unsigned)grid_main), dim3(256), 0, \
                         st, ts, F, ids, ids_stride, dense, dense_stride, 0, B_main, iters, out, out_stride, oob);  \
    else                                                                                                            \
      hipLaunchKernelGGL((pairdot_gram_kernel<I_, 1, 3, NR_, HD_, AP_>), dim3((unsigned)grid_rem), dim3(256), 0,    \
                         st, ts, F, ids, ids_stride, dense, dense_stride, B_main, (int)B, 1, out, out_stride, oob); \
  } while (0)
#define REC_GRAM_NR(I_, T_, W_, HD_, AP_)                   \
  do {                                                      \
    if (nr == 1) REC_GRAM_GO(I_, T_, W_, 1, HD_, AP_);      \
    else if (nr == 2) REC_GRAM_GO(I_, T_, W_, 2, HD_, AP_); \
    else REC_GRAM_GO(I_, T_, W_, 3, HD_, AP_);              \
  } while (0)
#define REC_GRAM_DENSE(I_, T_, W_)                          \
  do {                                                      \
    if (!has_dense) REC_GRAM_NR(I_, T_, W_, false, false);  \
    else if (!append) REC_GRAM_NR(I_, T_, W_, true, false); \
    else REC_GRAM_NR(I_, T_, W_, true, true);               \
  } while (0)
  if (tiles == 0) {
    const int64_t grid = (B + 3) / 4;
#define REC_GRAM_OS(I_, NR_, HD_, AP_)                                                                             \
  hipLaunchKernelGGL((pairdot_gram_oneshot_kernel<I_, NR_, HD_, AP_>), dim3((unsigned)grid), dim3(256), 0, st, ts, \
                     F, ids, ids_stride, dense, dense_stride, (int)B, out, out_stride, oob)
#define REC_GRAM_OS_NR(I_, HD_, AP_)               \
  do {                                             \
    if (nr == 1) REC_GRAM_OS(I_, 1, HD_, AP_);     \
    else if (nr == 2) REC_GRAM_OS(I_, 2, HD_, AP_); \
    else REC_GRAM_OS(I_, 3, HD_, AP_);             \
  } while (0)
#define REC_GRAM_OS_D(I_)                              \
  do {                                                 \
    if (!has_dense) REC_GRAM_OS_NR(I_, false, false);  \
    else if (!append) REC_GRAM_OS_NR(I_, true, false); \
    else REC_GRAM_OS_NR(I_, true, true);               \
  } while (0)
    if (ids_f32) REC_GRAM_OS_D(1); else REC_GRAM_OS_D(0);
#undef REC_GRAM_OS_D
#undef REC_GRAM_OS_NR
#undef REC_GRAM_OS
    return true;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const bool main_part = pass == 0;
    if (main_part ? iters == 0 : grid_rem == 0) continue;
    if (tiles == 1) {
      if (ids_f32) REC_GRAM_DENSE(1, 1, 3); else REC_GRAM_DENSE(0, 1, 3);
    } else {
      if (ids_f32) REC_GRAM_DENSE(1, 2, 2); else REC_GRAM_DENSE(0, 2, 2);
    }
  }
#undef REC_GRAM_DENSE
#undef REC_GRAM_NR
#undef REC_GRAM_GO
  return true;
}

}  // namespace rec
