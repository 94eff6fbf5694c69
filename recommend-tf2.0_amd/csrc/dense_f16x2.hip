// K11, large layers, second scheme: fp32-accurate Dense on the f16 matrix cores with THREE MFMAs per product instead of the
// six of the bf16x3 scheme (csrc/dense_bf16x3.hip), for callers that hand over one float of workspace per row.
//
// The matrix pipe of this chip is power-limited (tools/exp/mfma_peak.hip: 2.1 - 2.27 PFLOP/s of 16-bit MFMAs whatever the
// occupancy), so what a fp32-accurate GEMM costs is the number of MFMAs per product.  bf16 has fp32's exponent and 8
// significant bits: three terms, six products.  f16 has 11 significant bits — two terms reach 22, three products
// (h h, h l, l h; l l is 2^-22 of the result) — but only five exponent bits.  The range is bought with EXACT power-of-two
// scales: every row of x is scaled so that its largest magnitude lies in [2^14, 2^15), every column of W likewise (once, at
// prepare time), and the result is scaled back in the epilogue:
//     out[r][c] = act( ldexp(sum_k xs[r][k] ws[k][c], -(ex[r] + ew[c])) + b[c] ),   xs = x 2^ex[r] = xh + xl (+ <= 2^-22 |xs|)
// Powers of two change no mantissa bit, so the scheme's only errors are the dropped l l term and the residual of the
// two-term split (both <= 2^-22 relative to |x||w| per product, round-to-nearest: unbiased) on top of the fp32
// accumulation every kernel here has.  Measured against fp64 it is as accurate as the fmaf chain and the bf16x3 kernels or
// better (their splits truncate): tests/test_dense_gpu.py.  Elements more than 2^-18 below their row's maximum lose their
// low term to f16's subnormal range: an absolute error below 2^-40 of that maximum.  Rows whose maximum is 0, subnormal
// or non-finite are not scaled (inf / NaN go through the MFMAs and poison exactly their own row / column).  The
// epilogue scales back with ONE v_ldexp_f32 by the sum of the two integer exponents: no intermediate that could overflow
// where the result does not.
//
// Kernel structure = the hand-counted pipeline of csrc/dense_bf16x3.hip: 128 x 128 tile, four waves, W planes global -> LDS
// by LDS-DMA one k-step ahead, x a whole 128-B line per row and round, scalar bases + 32-bit lane offsets, counted
// s_waitcnt vmcnt.  Two planes per operand: 32 KB of LDS stages, 12 MFMAs and 8 ds_read_b128 per wave and k-step.
#include "common.h"
#include "ring_dma.h"

namespace rec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace h2 {
constexpr int BM = 128, BN = 128, BK = 16;

// the exponent e of the power-of-two scale that brings a magnitude into [2^14, 2^15) (applied with v_ldexp_f32: any normal
// maximum is covered, 2^e itself need not be representable); 0 = no scaling for zero / subnormal / non-finite maxima
__device__ __forceinline__ int scale_exp(float absmax) {
  const uint32_t be = (__builtin_bit_cast(uint32_t, absmax) >> 23) & 0xffu;
  return (be == 0u || be == 255u) ? 0 : 141 - (int)be;      // 14 - (be - 127)
}

// eight scaled fp32 values -> the f16 fragments of their two terms (round to nearest)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hq, u32x4& lq) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 p = {v[2 * j], v[2 * j + 1]};
    const f16x2 h = __builtin_convertvector(p, f16x2);
    const f32x2 r = p - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hq[j] = __builtin_bit_cast(uint32_t, h);
    lq[j] = __builtin_bit_cast(uint32_t, l);
  }
}
}  // namespace h2

// absmax[r] = max_k |x[r][k]| (NaNs are skipped: they reach the outputs through the MFMAs).  One wave per row.
__global__ __launch_bounds__(256) void row_absmax_kernel(const float* __restrict__ x, int64_t x_stride, int64_t M, int K,
                                                         int vec, float* __restrict__ absmax) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const float* row = x + r * x_stride;
  float m = 0.f;
  if (vec) {
    for (int k = lane * 4; k < K; k += 256) {
      const rec_f32x4_t v = *reinterpret_cast<const rec_f32x4_t*>(row + k);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  } else {
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(row[k]));
  }
  m = wave_max(m);
  if (lane == 0) absmax[r] = m;
}

// prepared form: [ceil(K/16)*2][2 planes][Np] f16x8 fragments of W 2^ew[c] (8 consecutive k of one column each), then Np
// column maxima (the bit patterns of max_k |W[k][c]|; the kernels derive the exponent).  Two small kernels: column maxima (atomic
// max over 64-row chunks; the area is zeroed by the bf16x3 prepare kernel that runs first), then the split.
__global__ __launch_bounds__(256) void dense_f16x2_colmax_kernel(const float* __restrict__ W, int K, int N,
                                                                 unsigned int* __restrict__ cmax) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int k0 = blockIdx.y * 64, k1 = k0 + 64 < K ? k0 + 64 : K;
  float m = 0.f;
  for (int k = k0; k < k1; ++k) m = fmaxf(m, fabsf(W[(int64_t)k * N + n]));
  atomicMax(cmax + n, __builtin_bit_cast(unsigned int, m));
}
__global__ __launch_bounds__(256) void dense_f16x2_prepare_kernel(const float* __restrict__ W, int K, int N, int Np, int K8,
                                                                  u32x4* __restrict__ Wq, const unsigned int* __restrict__ cmax) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)K8 * Np) return;
  const int k8 = (int)(t / Np), n = (int)(t - (int64_t)k8 * Np);
  const int e = h2::scale_exp(__builtin_bit_cast(float, cmax[n]));
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = k8 * 8 + j;
    w[j] = (n < N && kk < K) ? ldexpf(W[(int64_t)kk * N + n], e) : 0.f;
  }
  u32x4 h, l;
  h2::split8(w, h, l);
  Wq[((int64_t)k8 * 2 + 0) * Np + n] = h;
  Wq[((int64_t)k8 * 2 + 1) * Np + n] = l;
}
namespace {
__device__ __forceinline__ void gl16s(u32x4& dst, uint32_t voff, const void* sbase, int imm) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(voff), "s"(sbase), "n"(imm) : "memory");
}
}  // namespace
#define REC_HWAIT_X2(n, tok, a, b) \
  asm volatile("s_waitcnt vmcnt(" #n ")\n\tv_mov_b32 %0, 0" : "=v"(tok) : "v"(a), "v"(b) : "memory")
#define REC_HTOKEN_X2(tok, a, b) asm volatile("v_mov_b32 %0, 0" : "=v"(tok) : "v"(a), "v"(b) : "memory")

// Counts (see csrc/dense_bf16x3.hip for the scheme): odd step issues [x x4][W x2] — vmcnt(2) before the split, vmcnt(0) before
// every barrier.  No register the loads land in may be spilled (tests/test_kernel_resources_cpu.py).
__global__ __launch_bounds__(256, 3) void dense_f16x2_pipe_kernel(const float* __restrict__ x, int64_t x_stride,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ alpha, int act, int64_t M, int K,
                                                                  int N, float* __restrict__ out, int64_t out_stride,
                                                                  int out_vec, const u32x4* __restrict__ Wq, int Np,
                                                                  int xcd_map, const float* __restrict__ absmax,
                                                                  float* __restrict__ out_absmax) {
  using namespace h2;
  // [stage][operand A/B][plane h/l][kh][row]: 32 KiB of stages; the epilogue's four 32 x 68 transpose tiles need 34 KiB
  constexpr int kStageBytes = 2 * 2 * 2 * 2 * 128 * 16, kTileBytes = 4 * 32 * (64 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kTileBytes > kStageBytes ? kTileBytes : kStageBytes];
  u32x4 (*frag)[2][2][2][128] = reinterpret_cast<u32x4 (*)[2][2][2][128]>(lds_raw);
  __shared__ int ex_s[128];                 // the tile's row exponents (epilogue)
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
  const float* xbase = x + m0 * x_stride;                                 // rows >= M read row m0 (never stored)
  const uint32_t xoff = (uint32_t)(((gm < M ? srow : 0) * x_stride + 8 * skh) * 4);
  const int K8 = K / 8;
  const u32x4* wbase = Wq + n0;                                           // + (ks * 2) * 2 * Np per k-step
  const uint32_t woff = (uint32_t)((skh * 2 * Np + srow) * 16);
  const float* cmaxw = reinterpret_cast<const float*>(Wq + (int64_t)K8 * 2 * Np);
  const uint32_t wlds = __builtin_amdgcn_readfirstlane(lds_addr(&frag[0][1][0][skh][srow & 64]));
  const int ex = scale_exp(absmax[gm < M ? gm : m0]);
  if (skh == 0) ex_s[srow] = ex;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 xa[4];        // one round of this thread's x pieces: [step of the round * 2 + piece]
  auto issue_x = [&](int r) {
    const float* px = xbase + r * 2 * BK;
    gl16s(xa[0], xoff, px, 0);
    gl16s(xa[1], xoff, px, 16);
    gl16s(xa[2], xoff, px, BK * 4);
    gl16s(xa[3], xoff, px, BK * 4 + 16);
  };
  auto issue_w = [&](int ks, int st) {
    const u32x4* pw = wbase + (int64_t)ks * 4 * Np;
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(woff), "s"(pw), "s"(pw + Np),
          "s"(__builtin_amdgcn_readfirstlane(wlds + (uint32_t)st * (uint32_t)(kStageBytes / 2)))
        : "memory");
  };
  auto lwrite_x = [&](int st, const u32x4& lo, const u32x4& hi, uint32_t tok) {
    float av[8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      av[j] = ldexpf(__builtin_bit_cast(float, lo[j] ^ tok), ex), av[4 + j] = ldexpf(__builtin_bit_cast(float, hi[j] ^ tok), ex);
    u32x4 h, l;
    split8(av, h, l);
    frag[st][0][0][skh][srow] = h;
    frag[st][0][1][skh][srow] = l;
  };
  auto compute = [&](int st) {
    f16x8 a[2][2], b[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        a[t][p] = __builtin_bit_cast(f16x8, frag[st][0][p][half][wm * 64 + t * 32 + l32]);
        b[t][p] = __builtin_bit_cast(f16x8, frag[st][1][p][half][wn * 64 + t * 32 + l32]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], c, 0, 0, 0);  // h h
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], c, 0, 0, 0);  // h l
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], c, 0, 0, 0);  // l h
        acc[i][j] = c;
      }
  };
  uint32_t tok;
  const int nr = K / (2 * BK);   // rounds, >= 1
  issue_x(0);
  issue_w(0, 0);
  REC_HWAIT_X2(2, tok, xa[0], xa[1]);
  lwrite_x(0, xa[0], xa[1], tok);
  REC_VMCNT(0);
  __syncthreads();
  for (int r = 0; r < nr; ++r) {
    const bool next = r + 1 < nr;
    issue_w(2 * r + 1, 1);
    compute(0);
    REC_HTOKEN_X2(tok, xa[2], xa[3]);
    lwrite_x(1, xa[2], xa[3], tok);
    REC_VMCNT(0);
    __syncthreads();
    if (next) {
      issue_x(r + 1);
      issue_w(2 * r + 2, 0);
    }
    compute(1);
    if (next) {
      REC_HWAIT_X2(2, tok, xa[0], xa[1]);
      lwrite_x(0, xa[0], xa[1], tok);
      REC_VMCNT(0);
    }
    __syncthreads();
  }

  // epilogue: C[row = (q&3) + 8*(q>>2) + 4*(lane>>5)][col = lane&31]; the accumulators go through the transpose as they are and
  // are scaled back by 2^-(ex[row] + ew[col]) in one ldexp
  const int col_ok_base = n0 + wn * 64;
  if (out_vec) {
    constexpr int LDO = 64 + 4;
    float* ot = reinterpret_cast<float*>(lds_raw) + wv * 32 * LDO;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 16; ++q) ot[((q & 3) + 8 * (q >> 2) + 4 * half) * LDO + j * 32 + l32] = acc[i][j][q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int idx = e * 64 + lane, rr = idx >> 4, c4 = idx & 15;
        const int64_t row = m0 + wm * 64 + i * 32 + rr;
        const int col = col_ok_base + 4 * c4;
        float m4 = 0.f;
        if (row < M && col < N) {
          const int er = ex_s[wm * 64 + i * 32 + rr];
          const rec_f32x4_t cm = *reinterpret_cast<const rec_f32x4_t*>(cmaxw + col);      // Np is padded: always in range
          rec_f32x4_t v = *reinterpret_cast<const rec_f32x4_t*>(ot + rr * LDO + 4 * c4);
          v.x = ldexpf(v.x, -(er + scale_exp(cm.x)));
          v.y = ldexpf(v.y, -(er + scale_exp(cm.y)));
          v.z = ldexpf(v.z, -(er + scale_exp(cm.z)));
          v.w = ldexpf(v.w, -(er + scale_exp(cm.w)));
          const rec_f32x4_t bb = bias ? *reinterpret_cast<const rec_f32x4_t*>(bias + col) : rec_f32x4_t{0.f, 0.f, 0.f, 0.f};
          const rec_f32x4_t al = alpha ? *reinterpret_cast<const rec_f32x4_t*>(alpha + col) : rec_f32x4_t{0.f, 0.f, 0.f, 0.f};
          v.x = act_apply(v.x + bb.x, act, al.x);
          v.y = act_apply(v.y + bb.y, act, al.y);
          v.z = act_apply(v.z + bb.z, act, al.z);
          v.w = act_apply(v.w + bb.w, act, al.w);
          *reinterpret_cast<rec_f32x4_t*>(out + row * out_stride + col) = v;
          m4 = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        if (out_absmax) {
          // the next layer's row scale: the 16 lanes of a row reduce, one atomic max per row and 64-column block (magnitudes
          // compare as unsigned integers; out_absmax was zeroed before the launch)
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) m4 = fmaxf(m4, __shfl_xor(m4, o, 64));
          if (c4 == 0 && row < M) atomicMax(reinterpret_cast<unsigned int*>(out_absmax) + row, __builtin_bit_cast(unsigned int, m4));
        }
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
      const int col = col_ok_base + j * 32 + l32;
      if (col >= N) continue;
      const int ec = scale_exp(cmaxw[col]);
      const float bb = bias ? bias[col] : 0.f;
      const float al = alpha ? alpha[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int lr = wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * half;
        const int64_t row = m0 + lr;
        if (row < M) out[row * out_stride + col] = act_apply(ldexpf(acc[i][j][q], -(ex_s[lr] + ec)) + bb, act, al);
      }
    }
}

// Round 3 also built this kernel with a 256 x 128 workgroup tile (each wave 128 x 64: 12 fragment reads and one barrier per 24
// MFMAs instead of 8 and one per 12; a thread staging a whole 128-B x line; 221 VGPRs, two workgroups per CU; text in
// tools/exp/dense_variants/dense_f16x2_wide.hip.txt).  Bit-identical, and slower where it matters: 65 536 x 1024 x 512
// 0.274 vs 0.258 ms, x 1024 x 1024 0.534 vs 0.487; 2 - 3 % ahead only for K >= 2048 (profiles/r03_dense_f16x2_wide_ab.txt) —
// four resident workgroups per CU hide this kernel's serial parts better than larger tiles shrink them.  The same holds for
// the register-prefetched ("deep") form of the bf16x3 file applied here (170 VGPRs, two workgroups per CU; text in
// tools/exp/dense_variants/dense_f16x2_deep.hip.txt): bit-identical, 0.312 vs 0.261 ms at 1024 x 512, behind at every shape but
// 65 536 x 3456 x 128 (profiles/r03_dense_f16x2_deep_ab.txt).
#undef REC_HWAIT_X2
#undef REC_HTOKEN_X2

int64_t dense_f16x2_bytes(int K, int N) {
  const int64_t K8 = (int64_t)((K + 15) / 16) * 2, Np = (N + 127) / 128 * 128;
  return K8 * 2 * Np * 16 + Np * 4;
}

// the column-maxima area must be zero when this runs (dense_prepare_launch: the bf16x3 prepare kernel clears it)
void dense_f16x2_prepare_launch(const float* W, int K, int N, void* Wq, hipStream_t st) {
  const int K8 = (K + 15) / 16 * 2, Np = (N + 127) / 128 * 128;
  unsigned int* cmax = reinterpret_cast<unsigned int*>(static_cast<char*>(Wq) + (int64_t)K8 * 2 * Np * 16);
  hipLaunchKernelGGL(dense_f16x2_colmax_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)((K + 63) / 64)), dim3(256), 0, st, W, K,
                     N, cmax);
  const int64_t total = (int64_t)K8 * Np;
  hipLaunchKernelGGL(dense_f16x2_prepare_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, K, N, Np, K8,
                     static_cast<u32x4*>(Wq), cmax);
}

// absmax: M floats of workspace, filled here (one pass over x) unless absmax_valid.  out_absmax (optional, M floats, ZEROED by
// the caller): receives max_c |out[r][c]| — the next layer's row maxima — from the epilogue's atomic maxima (rows of 16-B
// aligned outputs) or from a pass over out.
// Returns false when the shape is not covered.
void row_absmax_launch(const float* x, int64_t x_stride, int64_t M, int K, float* absmax, hipStream_t st) {
  const int vec = (aligned16(x) && x_stride % 4 == 0 && K % 4 == 0) ? 1 : 0;
  hipLaunchKernelGGL(row_absmax_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, x_stride, M, K, vec, absmax);
}

bool dense_f16x2_dispatch(const float* x, int64_t x_stride, const void* Wq, const float* bias, const float* alpha, int act,
                          int64_t M, int K, int N, float* out, int64_t out_stride, float* absmax, int absmax_valid,
                          float* out_absmax, hipStream_t st) {
  const int64_t gx = (M + h2::BM - 1) / h2::BM;
  const int gy = (N + h2::BN - 1) / h2::BN;
  const int Np = (N + 127) / 128 * 128;
  if (!(aligned16(x) && x_stride % 4 == 0 && K % 32 == 0 && K >= 32 && x_stride < (1 << 22) && Np < (1 << 24))) return false;
  const int64_t total = ((gx + 7) / 8) * 8 * gy;
  if (total > 0x7fffffffLL || (M + 3) / 4 > 0x7fffffffLL) return false;
  if (bias && !aligned16(bias)) return false;
  if (alpha && !aligned16(alpha)) return false;
  if (!absmax_valid) row_absmax_launch(x, x_stride, M, K, absmax, st);
  const int xcd_map = M >= 4 * (int64_t)N ? 1 : 0;
  const int out_vec = (aligned16(out) && out_stride % 4 == 0 && N % 4 == 0) ? 1 : 0;
  hipLaunchKernelGGL(dense_f16x2_pipe_kernel, dim3((unsigned)total), dim3(256), 0, st, x, x_stride, bias, alpha, act, M, K, N,
                     out, out_stride, out_vec, static_cast<const u32x4*>(Wq), Np, xcd_map, absmax, out_vec ? out_absmax : nullptr);
  if (out_absmax && !out_vec) row_absmax_launch(out, out_stride, M, N, out_absmax, st);
  return true;
}

}  // namespace rec
