// K11 on the bf16 matrix cores with fp32 accuracy: out = act(x W + b) for the large Dense layers of the towers.
//
// fp32 MFMA runs at 256 flop/clk/CU; bf16 MFMA at 4096.  An fp32 value splits EXACTLY into three bf16 terms
// (x = h + m + l, 8 + 8 + 8 mantissa bits, by truncation; csrc/pairwise_dot_gram.hip uses the same split) and
// every product of two bf16 values is exact in the fp32 accumulator, so
//     x W = Xh Wh + (Xh Wm + Xm Wh) + (Xh Wl + Xl Wh) + Xm Wm + O(2^-24 |x||w|)
// is as accurate as fp32 FMA arithmetic (tests: 1e-5 against the fp64 oracle, like the fp32 kernel) at 6 bf16
// MFMAs per fp32-equivalent one: 16 / 6 = 2.7x the fp32 matrix-core peak.
//
// Workgroup = 128 x 128 output tile, 4 waves (2 x 2, each 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 tiles), K in
// steps of 16.  Staging: thread (row = tid & 127, kh = tid >> 7) loads 8 consecutive k of one x row and of one W
// column (the W loads are coalesced across the lanes of a wave: consecutive columns), splits them in registers
// (~45 VALU ops per operand per k-step, hidden under the 24 MFMAs a wave issues per k-step) and writes one 16-B
// fragment per plane to LDS laid out [plane][kh][row]: exactly the MFMA operand of lane (row & 31, kh), so the
// compute phase is 12 conflict-free ds_read_b128 per 24 MFMAs.  LDS is double-buffered, one barrier per k-step.
// Epilogue: bias + activation, then each wave transposes its 64 x 64 tile through LDS so that it leaves as 16-B
// row-major stores (65 536 x 13 x 512, an output-bound layer: 0.046 -> 0.033 ms).
#include <stdlib.h>

#include "bf16x3.h"
#include "common.h"
#include "ring_dma.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace b3 {
constexpr int BM = 128, BN = 128, BK = 16;
#ifndef REC_DENSE_PD
#define REC_DENSE_PD 2
#endif
// global loads run PD k-steps ahead of the MFMAs that consume them, in PD register sets (a k-step is ~0.35 us of matrix
// work per wave, an L2 / HBM round trip 0.6 - 2 us: one step ahead leaves the wave waiting at the LDS write)
constexpr int PD = REC_DENSE_PD;

using bf16x3::split8;
}  // namespace b3

// XMODE: 1 = aligned x rows (two dwordx4 per thread), 2 = unaligned rows via the borrowed transpose tile,
// 0 = unaligned rows, scalar loads (short K).  A template parameter: one kernel with all three paths needs 182
// VGPRs (2 workgroups per CU), the specialised ones 158 (3 per CU), worth 10 %.
template <int XMODE, bool BPREP>
__global__ __launch_bounds__(256, 2) void dense_bf16x3_kernel(const float* __restrict__ x, int64_t x_stride,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ alpha, int act, int64_t M,
                                                              int K, int N, float* __restrict__ out,
                                                              int64_t out_stride, int out_vec, const u32x4* __restrict__ Wp, int Np,
                                                              int xcd_map, int Kc = 0, int64_t out_split = 0) {
  using namespace b3;
  constexpr int x_vec = XMODE;
  // split-K (rec_dense_splitk_f32; gridDim.y slices of Kc reduction steps each, Kc a multiple of 16): slice y multiplies
  // columns [y Kc, (y + 1) Kc) of x by the matching rows of W into its own partial output
  if (Kc > 0) {
    const int ks = blockIdx.y;
    x += (int64_t)ks * Kc;
    W += (int64_t)ks * Kc * N;
    K = K - ks * Kc < Kc ? K - ks * Kc : Kc;
    out += (int64_t)ks * out_split;
  }
  // [stage][operand A/B][plane h/m/l][kh][row] of 16-B fragments: 2*2*3*2*128*16 B = 48 KiB
  __shared__ u32x4 frag[2][2][3][2][128];
  // rows of x that are not 16-B aligned (x_stride % 4 != 0, e.g. a tight (M, 479) matrix) cannot be read as two
  // dwordx4 per thread; they are loaded coalesced along k (lane -> k, 4 rows per wave-instruction) and transposed to
  // the row-per-thread staging map through an fp32 tile that borrows the B-operand region of the stage being
  // filled (two extra barriers per k-step on that path only; 48 KiB total keeps 3 workgroups per CU)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int l32 = lane & 31, half = lane >> 5;
  // Workgroup -> tile map, XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2),
  // so XCD c gets the M tiles c, c + 8, ... and walks ALL column tiles of one M tile back to back: the x tile is
  // fetched from HBM once and the other column blocks hit it in that XCD's L2 (instead of every column block
  // re-reading x from HBM -- with x taken out of the loop the same kernel runs at 2.2x the rate).
  const int ntn = (N + BN - 1) / BN;
  const int64_t ntm = (M + BM - 1) / BM;
  const int64_t L = blockIdx.x;
  int64_t mt;
  int nt_;
  if (xcd_map) {
    const int xcd = (int)(L & 7);
    const int64_t slot = L >> 3;
    mt = (slot / ntn) * 8 + xcd;
    nt_ = (int)(slot % ntn);
  } else {  // few rows, wide W: M tiles fastest, so that the workgroups in flight share a W column block
    mt = L % ntm;
    nt_ = (int)(L / ntm);
  }
  if (mt >= ntm || nt_ >= ntn) return;  // padding workgroups of the last group of 8 M tiles
  const int64_t m0 = mt * BM;
  const int n0 = nt_ * BN;
  const int srow = tid & 127, skh = tid >> 7;
  const int64_t gm = m0 + srow;
  const int gn = n0 + srow;
  const bool m_ok = gm < M, n_ok = gn < N;
  const float* xrow = x + (m_ok ? gm : 0) * x_stride;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float avs[PD][8], bvs[BPREP ? 1 : PD][8];
  u32x4 bqs[BPREP ? PD : 1][3];
  auto gload = [&](int k0, float (&av)[8], float (&bv)[8], u32x4 (&bq)[3]) {
    const int kb = k0 + 8 * skh;
    if constexpr (x_vec == 2) {
      const int kk = tid & 15, r0 = tid >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t row = m0 + r0 + 16 * j;
        av[j] = (row < M && k0 + kk < K) ? x[row * x_stride + k0 + kk] : 0.f;
      }
    } else if (x_vec == 1 && kb + 8 <= K) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(xrow + kb);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(xrow + kb + 4);
      av[0] = a0.x, av[1] = a0.y, av[2] = a0.z, av[3] = a0.w;
      av[4] = a1.x, av[5] = a1.y, av[6] = a1.z, av[7] = a1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = (kb + j < K) ? xrow[kb + j] : 0.f;
    }
    // rows >= M read row 0 (xrow) and are NOT zeroed: an A row only feeds its own C row, which the epilogue never
    // stores — and a select on the loaded values here would make every gload wait for its own loads (it did: the
    // compiler placed s_waitcnt vmcnt(3) right behind the W loads, exposing one x round trip per k-step)
    if constexpr (BPREP) {
      // prepared weights: [k / 8][plane][Np] fragments, already split (rec_dense_prepare_f32); padded with zeros
      const int64_t k8 = (k0 >> 3) + skh;
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[p] = Wp[(k8 * 3 + p) * Np + n0 + srow];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = (n_ok && kb + j < K) ? W[(int64_t)(kb + j) * N + gn] : 0.f;
    }
  };
  auto lwrite = [&](int st, float (&av)[8], float (&bv)[8], u32x4 (&bq)[3]) {
    u32x4 h, m, l;
    if constexpr (x_vec == 2) {
      float(*xs)[17] = reinterpret_cast<float(*)[17]>(&frag[st][1][0][0][0]);  // 128 x 17 floats < 12 KiB
      const int kk = tid & 15, r0 = tid >> 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) xs[r0 + 16 * j][kk] = av[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = xs[srow][8 * skh + j];
    }
    split8(av, h, m, l);
    frag[st][0][0][skh][srow] = h;
    frag[st][0][1][skh][srow] = m;
    frag[st][0][2][skh][srow] = l;
    if constexpr (x_vec == 2) __syncthreads();  // the borrowed tile has been read by everyone
    if constexpr (BPREP) {
      frag[st][1][0][skh][srow] = bq[0];
      frag[st][1][1][skh][srow] = bq[1];
      frag[st][1][2][skh][srow] = bq[2];
    } else {
      split8(bv, h, m, l);
      frag[st][1][0][skh][srow] = h;
      frag[st][1][1][skh][srow] = m;
      frag[st][1][2][skh][srow] = l;
    }
  };

  const int nk = (K + BK - 1) / BK;
#define REC_SET_(s) avs[s], bvs[BPREP ? 0 : (s)], bqs[BPREP ? (s) : 0]
  // prologue: k-steps 0 .. PD-1 leave; step 0 goes to LDS
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < nk) gload(s * BK, REC_SET_(s));
  lwrite(0, REC_SET_(0));
  __syncthreads();
  auto compute = [&](int st) {
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[t][p] = __builtin_bit_cast(bf16x8, frag[st][0][p][half][wm * 64 + t * 32 + l32]);
        b[t][p] = __builtin_bit_cast(bf16x8, frag[st][1][p][half][wn * 64 + t * 32 + l32]);
      }
    // (round 3 A/B, profiles/r03_dense_prio_ab.txt: this cluster at wave priority 1 or 3 — the T5 recipe of the cdna
    // guide, which helps the AutoInt stack — costs 7-11 % here: 0.362 -> 0.401 ms at 65 536 x 1024 x 512; not kept)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);  // h h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);  // h m
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);  // m h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);  // h l
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);  // l h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);  // m m
        acc[i][j] = c;
      }
  };
  // iteration ks: set ks % PD (step ks, already in LDS) is refilled with step ks + PD; the MFMAs of step ks run; step
  // ks + 1 (loaded PD - 1 iterations ago into set (ks + 1) % PD) is split and written to the other LDS stage
  for (int ks0 = 0; ks0 < nk; ks0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int ks = ks0 + u;
      if (ks >= nk) break;
      const int st = ks & 1;
      if (ks + PD < nk) gload((ks + PD) * BK, REC_SET_(u));
      compute(st);
      if (ks + 1 < nk) lwrite(st ^ 1, REC_SET_((u + 1) % PD));
      __syncthreads();
    }
  }
#undef REC_SET_

  // epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
  if (out_vec) {
    // 16-B stores: each wave transposes its 64 x 64 tile through LDS in two 32-row halves (the fragment buffers
    // are free after the last barrier of the k loop): 8 KiB + pad per wave
    constexpr int LDO = 64 + 4;
    float* ot = reinterpret_cast<float*>(&frag[0][0][0][0][0]) + wv * 32 * LDO;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        const float bb = (bias && col < N) ? bias[col] : 0.f;
        const float al = (alpha && col < N) ? alpha[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ot[((r & 3) + 8 * (r >> 2) + 4 * half) * LDO + j * 32 + l32] = act_apply(acc[i][j][r] + bb, act, al);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = e * 64 + lane, rr = idx >> 4, c4 = idx & 15;
        const int64_t row = m0 + wm * 64 + i * 32 + rr;
        const int col = n0 + wn * 64 + 4 * c4;
        if (row < M && col < N)  // N % 4 == 0: a 16-B group is inside or outside as a whole
          *reinterpret_cast<f32x4*>(out + row * out_stride + col) = *reinterpret_cast<const f32x4*>(ot + rr * LDO + 4 * c4);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l32;
      if (col >= N) continue;
      const float bb = bias ? bias[col] : 0.f;
      const float al = alpha ? alpha[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M) out[row * out_stride + col] = act_apply(acc[i][j][r] + bb, act, al);
      }
    }
}

// ---- the same tile with hand-counted loads (aligned x rows, prepared W, K % 32 == 0) ---------------------------------
// In the kernel above the compiler owns the s_waitcnt placement, and across the loop's control flow it gives up on the
// issue order: the ISA has `s_waitcnt vmcnt(2) / (1) / (0)` in front of the LDS writes of the PREVIOUS step's W planes —
// it also waits for the loads it has just issued for the next step, so every k-step pays a full L2 / HBM round trip
// (0.6 - 2 us against 0.35 us of matrix work per step; measured: prefetch distances 1, 2, 3, 4 all within 7 %).  With a
// straight-line body and plain loads it derives exact counts but sinks the loads next to their uses.  So the loop's
// loads are inline asm in a fixed order and the counts are static.
// Round 2's version of this kernel moved both operands through registers ([x lo][x hi][W h][W m][W l] per thread and
// k-step, two register sets; kept as text in tools/exp/dense_variants/).  Round 3 took it apart (65 536 x 1024 x 512,
// profiles/r03_dense_ablate*.txt, r03_mfma_peak.txt, r03_dense_stamps*.txt):
//   * the matrix pipe alone sustains 2.1 - 2.27 PFLOP/s of this MFMA whatever the occupancy — the clock falls to
//     1.35 - 1.6 GHz as waves are added: a power limit, 350 - 378 TFLOP/s fp32-equivalent at six MFMAs per product, 0.19 ms
//     for this layer; with this kernel's LDS traffic and barrier beside it 304 - 320;
//   * without any global load 0.362 -> 0.294 ms, with loads that always hit the L1 0.328 ms, without the split arithmetic
//     0.355 ms: the loads cost a fifth, half of it in the memory system.  A thread read 32 B of its x row per k-step and the
//     other half of that 128-B line one k-step later, after the CU's three workgroups had pulled 60 KB through a 32-KB L1:
//     every x line came from the L2 twice.
// So here
//   * x is loaded a ROUND (two k-steps = one 128-B line per row and k-half pair) at a time: four 16-B pieces per thread,
//     all issued together, each line fetched once;
//   * the W planes never visit registers: global_load_lds_dwordx4 drops the 64 fragments a wave used to load, wait for
//     and ds_write straight into the LDS stage (24 VGPRs and three ds_write_b128 per thread and k-step fewer).  The DMA
//     for step ks + 1 is issued right after the barrier that retires that stage's readers and must land by the barrier
//     that ends step ks: one k-step (~2 us with three workgroups per CU) for an L2 hit;
//   * addresses are a scalar base plus one 32-bit lane offset per operand: no 64-bit vector arithmetic in the loop.
// 140 VGPRs, three workgroups per CU, bit-identical results: 0.362 -> 0.350 ms, 65 536 x 3456 x 128 0.347 -> 0.325,
// 8192 x 4096 x 4096 200 -> 211 TFLOP/s (profiles/r03_dense_pipe2_ab.txt).  Not kept after A/Bs on the same box: a
// different issue priority per resident wave slot (the workgroups of a CU do not run in lock-step; r03_dense_slotprio_ab.txt),
// MFMAs ordered term by term over the four accumulators and the split's VALU work placed between them with
// sched_group_barrier (r03_dense_pipe2_order_ab.txt): all within 1 %.
// One register set suffices: a round's four pieces are requested at the start of its predecessor's second k-step (both of
// the set's halves have been split by then), their first half is split at the end of that step, the second half one step
// later.  Counts (issue order in the odd step: [x x4][W x3]): vmcnt(3) before the split — the x pieces are older than the
// three W loads — and vmcnt(0) before every barrier (the DMA'd planes must be in LDS when the stage is released).
// The wait asm returns a zero the consumers fold in (xor into the x values): the data dependency that keeps their
// instructions below the wait.  (A tied 128-bit "+v" operand would be the natural way; this toolchain lowers it as if the
// four elements were equal.)  The loads land asynchronously, so a register the compiler spilled between a load's issue and
// its wait would be clobbered when the load lands: both kernels below compile without spills, keep it that way
// (-Rpass-analysis=kernel-resource-usage).
#ifndef REC_DENSE_PIPE_WG
#define REC_DENSE_PIPE_WG 3
#endif
// Experiment builds only (-DREC_DENSE_STAMPS, tools/exp/dense_stamps.py): s_memtime at the phase boundaries of every k-step
// (with scheduling barriers: the phases do not overlap as they do in the product build), summed per wave and written OVER
// the first floats of the workgroup's output tile (row m0 + wave, columns n0 .. n0 + 6)
#ifdef REC_DENSE_STAMPS
#define REC_DSTAMP(i) do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph[i] += t_ - t_last; t_last = t_; } while (0)
#endif
// loads with a scalar base and a 32-bit per-lane offset: no 64-bit address arithmetic in the loop, one VGPR per operand
__device__ __forceinline__ void gl16s(u32x4& dst, uint32_t voff, const void* sbase, int imm) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(imm) : "memory");
}
#define REC_DWAIT_X2(n, tok, a, b) \
  asm volatile("s_waitcnt vmcnt(" #n ")\n\tv_mov_b32 %0, 0" : "=v"(tok) : "v"(a), "v"(b) : "memory")
// the pieces' second pair is consumed one step after the wait that covered it: a token without a wait ties its readers
// below that wait (volatile asm statements keep their order)
#define REC_DTOKEN_X2(tok, a, b) asm volatile("v_mov_b32 %0, 0" : "=v"(tok) : "v"(a), "v"(b) : "memory")
__global__ __launch_bounds__(256, REC_DENSE_PIPE_WG) void dense_bf16x3_pipe_kernel(const float* __restrict__ x, int64_t x_stride,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ alpha, int act, int64_t M,
                                                                    int K, int N, float* __restrict__ out,
                                                                    int64_t out_stride, int out_vec,
                                                                    const u32x4* __restrict__ Wp, int Np, int xcd_map) {
  using namespace b3;
  __shared__ u32x4 frag[2][2][3][2][128];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int l32 = lane & 31, half = lane >> 5;
  const int ntn = (N + BN - 1) / BN;
  const int64_t ntm = (M + BM - 1) / BM;
  const int64_t L = blockIdx.x;
  int64_t mt;
  int nt_;
  if (xcd_map) {
    const int xcd = (int)(L & 7);
    const int64_t slot = L >> 3;
    mt = (slot / ntn) * 8 + xcd;
    nt_ = (int)(slot % ntn);
  } else {
    mt = L % ntm;
    nt_ = (int)(L / ntm);
  }
  if (mt >= ntm || nt_ >= ntn) return;
  const int64_t m0 = mt * BM;
  const int n0 = nt_ * BN;
  const int srow = tid & 127, skh = tid >> 7;
  const int64_t gm = m0 + srow;
  // scalar bases (uniform: block index arithmetic) + per-lane byte offsets; rows >= M read row m0 (never stored)
  const float* xbase = x + m0 * x_stride;
  const uint32_t xoff = (uint32_t)(((gm < M ? srow : 0) * x_stride + 8 * skh) * 4);
  const u32x4* wbase = Wp + n0;                                          // + (ks * 2) * 3 * Np per k-step
  const uint32_t woff = (uint32_t)((skh * 3 * Np + srow) * 16);
  // LDS byte address of this wave's 64 fragments of plane 0 in W stage 0 (stage: + 24 KiB, plane: + 4 KiB)
  const uint32_t wlds = __builtin_amdgcn_readfirstlane(lds_addr(&frag[0][1][0][skh][srow & 64]));

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 xa[4];        // one round of this thread's x pieces: [step of the round * 2 + piece]
#ifdef REC_DENSE_STAMPS
  uint32_t ph[6] = {0u, 0u, 0u, 0u, 0u, 0u};      // wave-uniform: scalar registers
  uint32_t t_last = (uint32_t)__builtin_amdgcn_s_memtime();
  const uint32_t t_entry = t_last;
#define REC_DSTAMP2(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); REC_DSTAMP(i); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define REC_DSTAMP2(i) do { } while (0)
#endif
  auto issue_x = [&](int r) {
    const float* px = xbase + r * 2 * BK;
    gl16s(xa[0], xoff, px, 0);
    gl16s(xa[1], xoff, px, 16);
    gl16s(xa[2], xoff, px, BK * 4);
    gl16s(xa[3], xoff, px, BK * 4 + 16);
  };
  auto issue_w = [&](int ks, int st) {
    const u32x4* pw = wbase + (int64_t)ks * 6 * Np;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(woff), "s"(pw), "s"(pw + Np), "s"(pw + 2 * Np),
          "s"(__builtin_amdgcn_readfirstlane(wlds + (uint32_t)st * (uint32_t)sizeof(frag[0])))
        : "memory");
  };
  auto lwrite_x = [&](int st, const u32x4& lo, const u32x4& hi, uint32_t tok) {
    float av[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = __builtin_bit_cast(float, lo[j] ^ tok), av[4 + j] = __builtin_bit_cast(float, hi[j] ^ tok);
    u32x4 h, m, l;
    split8(av, h, m, l);
    frag[st][0][0][skh][srow] = h;
    frag[st][0][1][skh][srow] = m;
    frag[st][0][2][skh][srow] = l;
  };
  auto compute = [&](int st) {
    bf16x8 a[2][3], b[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[t][p] = __builtin_bit_cast(bf16x8, frag[st][0][p][half][wm * 64 + t * 32 + l32]);
        b[t][p] = __builtin_bit_cast(bf16x8, frag[st][1][p][half][wn * 64 + t * 32 + l32]);
      }
    REC_DSTAMP2(0);   // barrier release -> operands in registers
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);  // h h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);  // h m
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);  // m h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);  // h l
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);  // l h
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);  // m m
        acc[i][j] = c;
      }
  };
  uint32_t tok;
  const int nr = K / (2 * BK);   // rounds, >= 1
  issue_x(0);
  issue_w(0, 0);
  REC_DWAIT_X2(3, tok, xa[0], xa[1]);
  lwrite_x(0, xa[0], xa[1], tok);
  REC_VMCNT(0);
  __syncthreads();
  for (int r = 0; r < nr; ++r) {
    const bool next = r + 1 < nr;
    // k-step 2r (stage 0): the W planes of step 2r + 1 fly into stage 1 while the second half of this round's x is split
    issue_w(2 * r + 1, 1);
    compute(0);
    REC_DSTAMP2(1);   // MFMA issue
    REC_DTOKEN_X2(tok, xa[2], xa[3]);
    lwrite_x(1, xa[2], xa[3], tok);
    REC_DSTAMP2(3);   // split + LDS writes
    REC_VMCNT(0);
    REC_DSTAMP2(4);   // W planes landed
    __syncthreads();
    REC_DSTAMP2(5);   // barrier
    // k-step 2r + 1 (stage 1): the next round's x lines and the W planes of its first step are requested up front
    if (next) {
      issue_x(r + 1);
      issue_w(2 * r + 2, 0);
    }
    compute(1);
    REC_DSTAMP2(1);
    if (next) {
      REC_DWAIT_X2(3, tok, xa[0], xa[1]);
      REC_DSTAMP2(2);  // x pieces landed
      lwrite_x(0, xa[0], xa[1], tok);
        REC_DSTAMP2(3);
      REC_VMCNT(0);
      REC_DSTAMP2(4);
    }
    __syncthreads();
    REC_DSTAMP2(5);
  }

  // epilogue (as above): C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
  if (out_vec) {
    constexpr int LDO = 64 + 4;
    float* ot = reinterpret_cast<float*>(&frag[0][0][0][0][0]) + wv * 32 * LDO;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        const float bb = (bias && col < N) ? bias[col] : 0.f;
        const float al = (alpha && col < N) ? alpha[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          ot[((q & 3) + 8 * (q >> 2) + 4 * half) * LDO + j * 32 + l32] = act_apply(acc[i][j][q] + bb, act, al);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = e * 64 + lane, rr = idx >> 4, c4 = idx & 15;
        const int64_t row = m0 + wm * 64 + i * 32 + rr;
        const int col = n0 + wn * 64 + 4 * c4;
        if (row < M && col < N)
          *reinterpret_cast<f32x4*>(out + row * out_stride + col) = *reinterpret_cast<const f32x4*>(ot + rr * LDO + 4 * c4);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
#ifdef REC_DENSE_STAMPS
    __syncthreads();
    if (lane == 0 && m0 + wv < M) {
      float* o = out + (m0 + wv) * out_stride + n0;
#pragma unroll
      for (int q = 0; q < 6; ++q) o[q] = (float)ph[q];
      o[6] = (float)((uint32_t)__builtin_amdgcn_s_memtime() - t_entry);
    }
#endif
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l32;
      if (col >= N) continue;
      const float bb = bias ? bias[col] : 0.f;
      const float al = alpha ? alpha[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        if (row < M) out[row * out_stride + col] = act_apply(acc[i][j][q] + bb, act, al);
      }
    }
}

// ---- the same again with the operand fragments one k-step ahead in registers (long K) ---------------------------------
// Ordered stamps of the kernel above (tools/exp/dense_stamps.py, profiles/r03_dense_stamps_pipe2.txt): per wave and k-step,
// 1080 cycles from the barrier's release until the twelve fragments are in registers, 1034 in the MFMAs, 511 splitting and
// writing, 222 at the barrier: after every barrier a workgroup's four waves ask the LDS for 48 KB at once and its MFMAs wait
// (the other two workgroups of the CU fill part of the gap).  Here a wave reads the fragments of step ks + 1 while its
// MFMAs of step ks run (second fragment set: 223 VGPRs, two workgroups per CU), so the stage written during step ks is the
// one for step ks + 2:
//   step ks:  [W DMA (ks + 2) -> stage ks & 1] [x round loads]   fragments(ks + 1) <- stage (ks + 1) & 1
//             24 MFMAs on fragments(ks), the split of x(ks + 2) between them -> stage ks & 1   vmcnt   barrier
// Still two LDS stages: a stage is read (into registers) during the step before the one that uses it and rewritten during
// that one.  x rounds go to two register sets, requested two steps before their first half is split (issue order in an
// even step [W x3][x x4], in an odd step [W x3]; vmcnt(4) / vmcnt(0) before the barrier: the odd step's wait also covers
// the round whose first half the next step splits).  The scheduler is told where things go: the fragment reads before
// the step's MFMAs (left alone it sinks them next to their use, behind the barrier), three of the split's VALU
// instructions behind each MFMA, the MFMAs term by term over the four accumulators (per accumulator the same order as
// everywhere: bit-identical results).
// Against the kernel above on one box (profiles/r03_dense_pipe3_ab.txt): 8192 x 4096 x 4096 211 -> 224 TFLOP/s,
// 65 536 x 3360 x 256 0.594 -> 0.580 ms, 65 536 x 3456 x 128 0.325 -> 0.328; K <= 1024 0.350 -> 0.354 ms (1024 x 512), 0.346 ->
// 0.363 (480 x 1024): the longer prologue and the third workgroup it gives up cost more than the hidden reads return, so
// the dispatch takes it from K = 2048.  With its x loads pinned to L1-resident lines it runs 9 % faster, with the W planes
// 2 % (r03_dense_pipe3_ablate.txt): what is left is the memory system's rate for 128 scattered x lines per round, not latency.
__global__ __launch_bounds__(256, 2) void dense_bf16x3_pipe_deep_kernel(const float* __restrict__ x, int64_t x_stride,
                                                                    const float* __restrict__ bias,
                                                                    const float* __restrict__ alpha, int act, int64_t M,
                                                                    int K, int N, float* __restrict__ out,
                                                                    int64_t out_stride, int out_vec,
                                                                    const u32x4* __restrict__ Wp, int Np, int xcd_map) {
  using namespace b3;
  __shared__ u32x4 frag[2][2][3][2][128];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int l32 = lane & 31, half = lane >> 5;
  const int ntn = (N + BN - 1) / BN;
  const int64_t ntm = (M + BM - 1) / BM;
  const int64_t L = blockIdx.x;
  int64_t mt;
  int nt_;
  if (xcd_map) {
    const int xcd = (int)(L & 7);
    const int64_t slot = L >> 3;
    mt = (slot / ntn) * 8 + xcd;
    nt_ = (int)(slot % ntn);
  } else {
    mt = L % ntm;
    nt_ = (int)(L / ntm);
  }
  if (mt >= ntm || nt_ >= ntn) return;
  const int64_t m0 = mt * BM;
  const int n0 = nt_ * BN;
  const int srow = tid & 127, skh = tid >> 7;
  const int64_t gm = m0 + srow;
  const float* xbase = x + m0 * x_stride;
  const uint32_t xoff = (uint32_t)(((gm < M ? srow : 0) * x_stride + 8 * skh) * 4);
  const u32x4* wbase = Wp + n0;
  const uint32_t woff = (uint32_t)((skh * 3 * Np + srow) * 16);
  const uint32_t wlds = __builtin_amdgcn_readfirstlane(lds_addr(&frag[0][1][0][skh][srow & 64]));

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 xa[2][4];               // [round & 1][step of the round * 2 + piece]
  bf16x8 fa[2][2][3], fb[2][2][3];   // [step & 1][tile][plane]
  auto issue_x = [&](int r, u32x4 (&xs)[4]) {
    const float* px = xbase + r * 2 * BK;
    gl16s(xs[0], xoff, px, 0);
    gl16s(xs[1], xoff, px, 16);
    gl16s(xs[2], xoff, px, BK * 4);
    gl16s(xs[3], xoff, px, BK * 4 + 16);
  };
  auto issue_w = [&](int ks, int st) {
    const u32x4* pw = wbase + (int64_t)ks * 6 * Np;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(woff), "s"(pw), "s"(pw + Np), "s"(pw + 2 * Np),
          "s"(__builtin_amdgcn_readfirstlane(wlds + (uint32_t)st * (uint32_t)sizeof(frag[0])))
        : "memory");
  };
  auto lwrite_x = [&](int st, const u32x4& lo, const u32x4& hi) {
    uint32_t tok;
    REC_DTOKEN_X2(tok, lo, hi);
    float av[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = __builtin_bit_cast(float, lo[j] ^ tok), av[4 + j] = __builtin_bit_cast(float, hi[j] ^ tok);
    u32x4 h, m, l;
    split8(av, h, m, l);
    frag[st][0][0][skh][srow] = h;
    frag[st][0][1][skh][srow] = m;
    frag[st][0][2][skh][srow] = l;
  };
  auto fread = [&](int st, bf16x8 (&a)[2][3], bf16x8 (&b)[2][3]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[t][p] = __builtin_bit_cast(bf16x8, frag[st][0][p][half][wm * 64 + t * 32 + l32]);
        b[t][p] = __builtin_bit_cast(bf16x8, frag[st][1][p][half][wn * 64 + t * 32 + l32]);
      }
  };
  auto mfmas = [&](const bf16x8 (&a)[2][3], const bf16x8 (&b)[2][3]) {
    // term by term over the four accumulators (per accumulator the same order as the kernels above: bit-identical), so that
    // consecutive MFMAs never depend on each other
    constexpr int pa[6] = {0, 0, 1, 0, 2, 1}, pb[6] = {0, 1, 0, 2, 0, 1};   // h h, h m, m h, h l, l h, m m
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa[t]], b[j][pb[t]], acc[i][j], 0, 0, 0);
  };
  const int nk = K / BK;          // even, >= 2
  const int nr = nk / 2;
  // prologue: both stages filled (steps 0 and 1), the fragments of step 0 in registers
  issue_w(0, 0);
  issue_w(1, 1);
  issue_x(0, xa[0]);
  if (nr > 1) {
    issue_x(1, xa[1]);
    REC_VMCNT(4);
  } else {
    REC_VMCNT(0);
  }
  lwrite_x(0, xa[0][0], xa[0][1]);
  lwrite_x(1, xa[0][2], xa[0][3]);
  REC_VMCNT(0);             // round 1 as well: step 0 splits its first half without a wait of its own
  __syncthreads();
  fread(0, fa[0], fb[0]);
  __syncthreads();          // every wave holds step 0's fragments: stage 0 may be refilled
  // one round = k-steps 2r (fragment set 0) and 2r + 1 (set 1); XS = the x set of round r + 1
  // the split's ~70 VALU instructions go between the MFMAs (three after each) instead of behind all 24: the matrix pipe
  // works through an MFMA for 32 cycles, the wave issues its vector work meanwhile
#define REC_INTERLEAVE()                                                   \
  do {                                                                     \
    _Pragma("unroll") for (int g_ = 0; g_ < 24; ++g_) {                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);        \
    }                                                                      \
  } while (0)
#define REC_ROUND3(r, XS, MORE, XNEXT)   /* MORE: a step 2r + 2 exists; XNEXT: a round r + 2 exists */ \
  do {                                                                               \
    if (MORE) issue_w(2 * (r) + 2, 0);                                               \
    if (XNEXT) issue_x((r) + 2, xa[(XS) ^ 1]);                                       \
    fread(1, fa[1], fb[1]);                                                          \
    __builtin_amdgcn_sched_barrier(0);   /* the reads go out before this step's MFMAs, not next to their use */ \
    mfmas(fa[0], fb[0]);                                                             \
    if (MORE) { lwrite_x(0, xa[XS][0], xa[XS][1]); REC_INTERLEAVE(); }               \
    if (XNEXT) REC_VMCNT(4); else REC_VMCNT(0);                                      \
    __syncthreads();                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                               \
    if (MORE) {                                                                      \
      issue_w(2 * (r) + 3, 1);                                                       \
      fread(0, fa[0], fb[0]);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                             \
    }                                                                                \
    mfmas(fa[1], fb[1]);                                                             \
    if (MORE) {                                                                      \
      lwrite_x(1, xa[XS][2], xa[XS][3]);                                             \
      REC_INTERLEAVE();                                                              \
      REC_VMCNT(0);                                                                  \
      __syncthreads();                                                               \
      __builtin_amdgcn_sched_barrier(0);                                             \
    }                                                                                \
  } while (0)
  int r = 0;
  for (; r + 3 < nr; r += 2) {       // steady state: no conditions inside
    REC_ROUND3(r, 1, true, true);
    REC_ROUND3(r + 1, 0, true, true);
  }
  // the last one to three rounds
  if (r < nr) {
    REC_ROUND3(r, 1, r + 1 < nr, r + 2 < nr);
    ++r;
  }
  if (r < nr) {
    REC_ROUND3(r, 0, r + 1 < nr, false);
    ++r;
  }
  if (r < nr) REC_ROUND3(r, 1, false, false);
#undef REC_ROUND3
#undef REC_INTERLEAVE
  __syncthreads();      // the epilogue's transpose tile reuses the stages

  if (out_vec) {
    constexpr int LDO = 64 + 4;
    float* ot = reinterpret_cast<float*>(&frag[0][0][0][0][0]) + wv * 32 * LDO;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l32;
        const float bb = (bias && col < N) ? bias[col] : 0.f;
        const float al = (alpha && col < N) ? alpha[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
          ot[((q & 3) + 8 * (q >> 2) + 4 * half) * LDO + j * 32 + l32] = act_apply(acc[i][j][q] + bb, act, al);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = e * 64 + lane, rr = idx >> 4, c4 = idx & 15;
        const int64_t row = m0 + wm * 64 + i * 32 + rr;
        const int col = n0 + wn * 64 + 4 * c4;
        if (row < M && col < N)
          *reinterpret_cast<f32x4*>(out + row * out_stride + col) = *reinterpret_cast<const f32x4*>(ot + rr * LDO + 4 * c4);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l32;
      if (col >= N) continue;
      const float bb = bias ? bias[col] : 0.f;
      const float al = alpha ? alpha[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        if (row < M) out[row * out_stride + col] = act_apply(acc[i][j][q] + bb, act, al);
      }
    }
}
#undef REC_DWAIT_X2
#undef REC_DSTAMP2
#undef REC_DTOKEN_X2


// W -> [ceil(K/16)*2][3 planes][Np = round_up(N, 128)] bf16x8 fragments (8 consecutive k of one column each)
__global__ __launch_bounds__(256) void dense_prepare_kernel(const float* __restrict__ W, int K, int N, int Np, int K8,
                                                            u32x4* __restrict__ Wp, unsigned int* __restrict__ clear) {
  using namespace b3;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < Np) clear[e] = 0u;        // the f16x2 form's column maxima accumulate by atomic max in the launch that follows
  if (e >= (int64_t)K8 * Np) return;
  const int k8 = (int)(e / Np), n = (int)(e - (int64_t)k8 * Np);
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = k8 * 8 + j;
    w[j] = (n < N && kk < K) ? W[(int64_t)kk * N + n] : 0.f;
  }
  u32x4 h, m, l;
  split8(w, h, m, l);
  Wp[((int64_t)k8 * 3 + 0) * Np + n] = h;
  Wp[((int64_t)k8 * 3 + 1) * Np + n] = m;
  Wp[((int64_t)k8 * 3 + 2) * Np + n] = l;
}

// out_parts[s] (M, N contiguous) = x[:, s Kc : (s + 1) Kc] W[s Kc : (s + 1) Kc, :] for s < splits, in ONE launch (no bias, no
// activation): the reduction axis of a few-tile product is spread over gridDim.y
bool dense_bf16x3_splitk(const float* x, int64_t x_stride, const float* W, int64_t M, int K, int N, int splits, int Kc,
                         float* out_parts, hipStream_t st) {
  const int64_t gx = (M + b3::BM - 1) / b3::BM;
  const int gy = (N + b3::BN - 1) / b3::BN;
  const int x_vec = (aligned16(x) && x_stride % 4 == 0) ? 1 : (Kc >= 64 ? 2 : 0);
  const int64_t total = ((gx + 7) / 8) * 8 * gy;
  if (total > 0x7fffffffLL || splits > 65535) return false;
  const dim3 grid((unsigned)total, (unsigned)splits);
  const int out_vec = (aligned16(out_parts) && N % 4 == 0) ? 1 : 0;
  const int Np = (N + 127) / 128 * 128;
#define REC_B3_SK(XM_)                                                                                                  \
  hipLaunchKernelGGL((dense_bf16x3_kernel<XM_, false>), grid, dim3(256), 0, st, x, x_stride, W, (const float*)nullptr,    \
                     (const float*)nullptr, (int)REC_ACT_NONE, M, K, N, out_parts, (int64_t)N, out_vec,                   \
                     (const u32x4*)nullptr, Np, 0, Kc, M * (int64_t)N)
  if (x_vec == 1) REC_B3_SK(1);
  else if (x_vec == 2) REC_B3_SK(2);
  else REC_B3_SK(0);
#undef REC_B3_SK
  return true;
}

// caller has validated shapes/pointers (rec_dense_f32 / rec_dense_prep_f32); Wp may be NULL
bool dense_bf16x3_dispatch(const float* x, int64_t x_stride, const float* W, const void* Wp, const float* bias,
                           const float* alpha, int act, int64_t M, int K, int N, float* out, int64_t out_stride,
                           hipStream_t st) {
  const int64_t gx = (M + b3::BM - 1) / b3::BM;
  const int gy = (N + b3::BN - 1) / b3::BN;
  if (gx > 0x7fffffffLL || gy > 65535) return false;
  // x staging: 1 = two dwordx4 per thread (aligned rows), 2 = coalesced along k + LDS transpose (unaligned rows,
  // K large enough to pay for the extra barrier), 0 = scalar loads per thread (unaligned, short K)
  const int x_vec = (aligned16(x) && x_stride % 4 == 0) ? 1 : (K >= 64 ? 2 : 0);
  const int xcd_map = M >= 4 * (int64_t)N ? 1 : 0;  // x is the big operand: keep its tile in one XCD's L2
  const int64_t total = ((gx + 7) / 8) * 8 * gy;  // M tiles padded to a multiple of 8 (one per XCD)
  if (total > 0x7fffffffLL) return false;
  const dim3 grid((unsigned)total);
  const int out_vec = (aligned16(out) && out_stride % 4 == 0 && N % 4 == 0) ? 1 : 0;
  const int Np = (N + 127) / 128 * 128;
  const u32x4* wp = static_cast<const u32x4*>(Wp);
#define REC_B3_GO(XM_, BP_)                                                                                      \
  hipLaunchKernelGGL((dense_bf16x3_kernel<XM_, BP_>), grid, dim3(256), 0, st, x, x_stride, W, bias, alpha, act, M, K, \
                     N, out, out_stride, out_vec, wp, Np, xcd_map)
  // aligned x rows + prepared W + K a multiple of 32: the hand-counted pipelines (their lane offsets are 32-bit);
  // rec_debug_force("dense_pipe", "0" | "s" | "d"): off / the standard one / the deep one whatever K
  const char* fp = forced("dense_pipe");
  const bool pipe_ok = !(fp && fp[0] == '0') && x_stride < (1 << 22) && Np < (1 << 24);
  if (pipe_ok && wp && x_vec == 1 && K % 32 == 0 && K >= 32) {
    const bool deep = fp && (fp[0] == 'd' || fp[0] == 's') ? fp[0] == 'd' : K >= 2048;
    if (deep)
      hipLaunchKernelGGL(dense_bf16x3_pipe_deep_kernel, grid, dim3(256), 0, st, x, x_stride, bias, alpha, act, M, K, N, out,
                         out_stride, out_vec, wp, Np, xcd_map);
    else
      hipLaunchKernelGGL(dense_bf16x3_pipe_kernel, grid, dim3(256), 0, st, x, x_stride, bias, alpha, act, M, K, N, out,
                         out_stride, out_vec, wp, Np, xcd_map);
    return true;
  }
  if (wp && x_vec != 2) {  // with the transpose-tile x path the prepared form measured slower (1.10 vs 0.88 ms at K = 3341)
    if (x_vec == 1) REC_B3_GO(1, true);
    else REC_B3_GO(0, true);
  } else {
    if (x_vec == 1) REC_B3_GO(1, false);
    else if (x_vec == 2) REC_B3_GO(2, false);
    else REC_B3_GO(0, false);
  }
#undef REC_B3_GO
  return true;
}

// the prepared buffer: [bf16x3 planes][f16x2 planes + inverse column scales + usable flag (csrc/dense_f16x2.hip)]
int64_t dense_f16x2_bytes(int K, int N);
void dense_f16x2_prepare_launch(const float* W, int K, int N, void* Wq, hipStream_t st);
int64_t dense_b3_prepared_bytes(int K, int N) {
  const int64_t K8 = (int64_t)((K + 15) / 16) * 2, Np = (N + 127) / 128 * 128;
  return K8 * 3 * Np * 16;
}
int64_t dense_prepared_bytes(int K, int N) { return dense_b3_prepared_bytes(K, N) + dense_f16x2_bytes(K, N); }

void dense_prepare_launch(const float* W, int K, int N, void* Wp, hipStream_t st) {
  const int K8 = (K + 15) / 16 * 2, Np = (N + 127) / 128 * 128;
  const int64_t total = (int64_t)K8 * Np;
  char* hq = static_cast<char*>(Wp) + dense_b3_prepared_bytes(K, N);
  hipLaunchKernelGGL(dense_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, K, N, Np, K8,
                     static_cast<u32x4*>(Wp), reinterpret_cast<unsigned int*>(hq + (int64_t)K8 * 2 * Np * 16));
  dense_f16x2_prepare_launch(W, K, N, hq, st);
}

}  // namespace rec

// ------------------------------------------------------------------------------------------------------------
// Row-streaming form for the zoo's skinny layers (K <= 128, N <= 128: attention projections, FFN, tower heads),
// which are HBM-bound (x in, out back, W tiny).  W is split once per workgroup into B-operand fragments resident in
// LDS; persistent waves stream 32-row tiles of x: a lane (row, k-half) loads its own row's 8-float pieces straight
// from global memory in MFMA operand layout (no LDS hop for x), one tile prefetched in registers; the 32 x N result
// is transposed through a wave-private LDS tile so that it leaves as 16-B row-major stores.
// ------------------------------------------------------------------------------------------------------------
namespace rec {

template <int KS, int NT>
__global__ __launch_bounds__(256, 3) void dense_b3_rows_kernel(const float* __restrict__ x, int64_t x_stride,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ alpha, int act, int64_t M,
                                                               int K, int N, float* __restrict__ out,
                                                               int64_t out_stride, int out_vec) {
  using namespace b3;
  constexpr int NC = 32 * NT;       // padded columns
  constexpr int LDO = NC + 4;       // output staging row stride (floats)
  extern __shared__ __attribute__((aligned(16))) unsigned char b3r_smem[];
  u32x4(*Wf)[3][2][NC] = reinterpret_cast<u32x4(*)[3][2][NC]>(b3r_smem);                      // [KS][3][2][NC]
  float* ost = reinterpret_cast<float*>(b3r_smem + (size_t)KS * 3 * 2 * NC * sizeof(u32x4));  // [4][32 * LDO]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;

  // W fragments: (ks, half, col) -> 8 consecutive k of one column
  for (int e = tid; e < KS * 2 * NC; e += 256) {
    const int col = e % NC, kh = (e / NC) & 1, ks = e / (2 * NC);
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = ks * 16 + kh * 8 + j;
      w[j] = (col < N && kk < K) ? W[(int64_t)kk * N + col] : 0.f;
    }
    u32x4 h, m, l;
    split8(w, h, m, l);
    Wf[ks][0][kh][col] = h;
    Wf[ks][1][kh][col] = m;
    Wf[ks][2][kh][col] = l;
  }
  float bcol[NT], acol[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 32 + l32;
    bcol[t] = (bias && col < N) ? bias[col] : 0.f;
    acol[t] = (alpha && col < N) ? alpha[col] : 0.f;
  }
  __syncthreads();

  const int64_t ntiles = (M + 31) / 32;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  int64_t rt = (int64_t)blockIdx.x * 4 + wv;
  f32x4 xa[KS][2];
  auto gload = [&](int64_t tile) {
    const int64_t row = tile * 32 + l32;
    const bool ok = tile < ntiles && row < M;
    const float* p = x + (ok ? row : 0) * x_stride + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ok && ks * 16 + 8 * half + 8 <= K) {
        xa[ks][0] = *reinterpret_cast<const f32x4*>(p + 16 * ks);
        xa[ks][1] = *reinterpret_cast<const f32x4*>(p + 16 * ks + 4);
      } else {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = (ok && ks * 16 + 8 * half + j < K) ? p[16 * ks + j] : 0.f;
        xa[ks][0] = f32x4{t[0], t[1], t[2], t[3]};
        xa[ks][1] = f32x4{t[4], t[5], t[6], t[7]};
      }
    }
  };
  if (rt < ntiles) gload(rt);
  float* st = ost + wv * 32 * LDO;
  for (; rt < ntiles; rt += nwaves) {
    u32x4 af[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float t[8] = {xa[ks][0].x, xa[ks][0].y, xa[ks][0].z, xa[ks][0].w,
                          xa[ks][1].x, xa[ks][1].y, xa[ks][1].z, xa[ks][1].w};
      split8(t, af[ks][0], af[ks][1], af[ks][2]);
    }
    gload(rt + nwaves);  // prefetch the wave's next tile (zeros past the end)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, af[ks][0]), am = __builtin_bit_cast(bf16x8, af[ks][1]),
                   al = __builtin_bit_cast(bf16x8, af[ks][2]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, Wf[ks][0][half][t * 32 + l32]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, Wf[ks][1][half][t * 32 + l32]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, Wf[ks][2][half][t * 32 + l32]);
        f32x16 c = acc[t];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        acc[t] = c;
      }
    }
    // bias + activation, then transpose through the wave's LDS tile: C[row = (r&3)+8(r>>2)+4 half][col = l32]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        st[row * LDO + t * 32 + l32] = act_apply(acc[t][r] + bcol[t], act, acol[t]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t row0 = rt * 32;
    if (out_vec) {  // N % 4 == 0, 16-B aligned rows
      const int n4 = N >> 2;
      for (int e = lane; e < 32 * n4; e += 64) {
        const int row = e / n4, c4 = e - row * n4;
        if (row0 + row < M)
          *reinterpret_cast<f32x4*>(out + (row0 + row) * out_stride + 4 * c4) =
              *reinterpret_cast<const f32x4*>(st + row * LDO + 4 * c4);
      }
    } else {
      for (int e = lane; e < 32 * N; e += 64) {
        const int row = e / N, c = e - row * N;
        if (row0 + row < M) out[(row0 + row) * out_stride + c] = st[row * LDO + c];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// returns false when the shape is not covered
bool dense_b3_rows_dispatch(const float* x, int64_t x_stride, const float* W, const float* bias, const float* alpha,
                            int act, int64_t M, int K, int N, float* out, int64_t out_stride, hipStream_t st) {
  if (K > 128 || N > 128 || !aligned16(x) || x_stride % 4 != 0) return false;
  const int ks = (K + 15) / 16, nt = (N + 31) / 32;
  if (ks * nt > 16) return false;  // W fragments: ks * nt * 3 KiB of LDS
  const int out_vec = (aligned16(out) && out_stride % 4 == 0 && N % 4 == 0) ? 1 : 0;
  const int64_t ntiles = (M + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  if (blocks > 256 * 3) blocks = 256 * 3;
  bool done = false;
#define REC_B3R(KS_, NT_)                                                                                         \
  if (!done && ks <= KS_ && nt == NT_) {                                                                          \
    const size_t lds = (size_t)KS_ * 3 * 2 * (32 * NT_) * 16 + (size_t)4 * 32 * (32 * NT_ + 4) * sizeof(float);   \
    if (lds > 64 * 1024 &&                                                                                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(dense_b3_rows_kernel<KS_, NT_>),                        \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                  \
      return false;                                                                                               \
    hipLaunchKernelGGL((dense_b3_rows_kernel<KS_, NT_>), dim3((unsigned)blocks), dim3(256), lds, st, x, x_stride, \
                       W, bias, alpha, act, M, K, N, out, out_stride, out_vec);                                   \
    done = true;                                                                                                  \
  }
  REC_B3R(1, 1) REC_B3R(2, 1) REC_B3R(4, 1)
  REC_B3R(1, 2) REC_B3R(2, 2) REC_B3R(4, 2) REC_B3R(8, 2)
  REC_B3R(1, 3) REC_B3R(2, 3) REC_B3R(4, 3)
  REC_B3R(1, 4) REC_B3R(2, 4) REC_B3R(4, 4)
#undef REC_B3R
  return done;
}

}  // namespace rec
