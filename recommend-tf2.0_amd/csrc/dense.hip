// K11 — Dense: out = act(x @ W + bias) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32, k-ordered fma chain; there is no xf32/TF32 on gfx950, and 1e-5 parity needs fp32 anyway).
//
// Used for the MLP towers (src/ctr/layers/modules.py:129-135 with the BatchNormalization folded
// into W/bias by the host, src/match/layers/modules.py:21-26), Conv1D(k=1)
// (src/match/layers/modules.py:146-149) and the attention projections.  Not the headline kernel
// (SURVEY K11): a plain LDS-tiled 128x128x16 block, 4 waves as 2x2, each wave 2x2 tiles of 32x32.
// Roofline: fp32 MFMA (157.3 TFLOP/s dense).
#include <stdlib.h>

#include "common.h"

namespace rec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA = BM + 4;  // As[k][m]: +4 keeps rows 16-B aligned and staggers banks
constexpr int LDB = BN + 4;

__global__ __launch_bounds__(256) void dense_mfma_kernel(const float* __restrict__ x, int64_t x_stride,
                                                         const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ alpha, int act,
                                                         int64_t M, int K, int N,
                                                         float* __restrict__ out, int64_t out_stride) {
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int64_t m0 = (int64_t)blockIdx.x * BM;  // M tiles on grid.x (no 65535 limit)
  const int n0 = blockIdx.y * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging maps: A tile 128 rows x 16 k -> thread (row = tid/2, 8 consecutive k);
  //               B tile 16 k x 128 n   -> thread (k = tid/16, 8 consecutive n)
  const int a_row = tid >> 1, a_k = (tid & 1) * 8;
  const int b_k = tid >> 4, b_n = (tid & 15) * 8;

  for (int k0 = 0; k0 < K; k0 += BK) {
    float av[8], bv[8];
    {
      const int64_t gm = m0 + a_row;
      const float* pa = x + gm * x_stride + k0 + a_k;
#pragma unroll
      for (int e = 0; e < 8; ++e) av[e] = (gm < M && k0 + a_k + e < K) ? pa[e] : 0.f;
      const int gk = k0 + b_k;
      const float* pb = W + (int64_t)gk * N + n0 + b_n;
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (gk < K && n0 + b_n + e < N) ? pb[e] : 0.f;
    }
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int e = 0; e < 8; ++e) As[(a_k + e) * LDA + a_row] = av[e];
#pragma unroll
    for (int e = 0; e < 8; ++e) Bs[b_k * LDB + b_n + e] = bv[e];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kr = kk + (lane >> 5);
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = As[kr * LDA + wm * 64 + t * 32 + (lane & 31)];
        b[t] = Bs[kr * LDB + wn * 64 + t * 32 + (lane & 31)];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= N) continue;
      const float bb = bias ? bias[col] : 0.f;
      const float al = alpha ? alpha[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) out[row * out_stride + col] = act_apply(acc[i][j][r] + bb, act, al);
      }
    }
}

// narrow outputs (N <= 8, e.g. the final Dense(1)): one wave per row, lanes stride over K
template <int NMAX>
__global__ __launch_bounds__(256) void dense_narrow_kernel(const float* __restrict__ x, int64_t x_stride,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ alpha, int act,
                                                           int64_t M, int K, int N,
                                                           float* __restrict__ out, int64_t out_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float acc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float xv = x[row * x_stride + k];
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < N) acc[n] = fmaf(xv, W[(int64_t)k * N + n], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    const float s = wave_sum(acc[n]);
    if (lane == 0 && n < N)
      out[row * out_stride + n] = act_apply(s + (bias ? bias[n] : 0.f), act, alpha ? alpha[n] : 0.f);
  }
}


// N == 1 (the final Dense(1) of every ctr model), aligned rows: 16 lanes per row with 16-B loads, 4 rows per wave
__global__ __launch_bounds__(256) void dense_vec1_kernel(const float* __restrict__ x, int64_t x_stride,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ alpha, int act, int64_t M, int K,
                                                         float* __restrict__ out, int64_t out_stride) {
  const int lane = threadIdx.x & 63, sub = lane & 15;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
  const int64_t rc = row < M ? row : M - 1;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + rc * x_stride);
  const f32x4* w4 = reinterpret_cast<const f32x4*>(W);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k4 = sub; k4 < (K >> 2); k4 += 16) acc += xr[k4] * w4[k4];
  float s = acc.x + acc.y + acc.z + acc.w;
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0 && row < M) out[row * out_stride] = act_apply(s + (bias ? bias[0] : 0.f), act, alpha ? alpha[0] : 0.f);
}

// ------------------------------------------------------------------------------------------------
// Skinny Dense (K <= 128, N <= 128): the layers of this zoo are narrow (64x64 attention projections,
// 64<->128 FFN, 32..256-wide towers) while M is huge (batch x seq), so the op is HBM-bound:
// W (<= 64 KiB) is staged in LDS once per workgroup, every wave streams 32-row tiles of x straight
// from global memory into MFMA A operands (lane half h holds k in [h*K/2, (h+1)*K/2): 16-B loads,
// the k-permutation is matched on the B side) and keeps all N/32 accumulator tiles in registers:
// x is read once, out written once.
// ------------------------------------------------------------------------------------------------
template <int KH /* K/2 */, int NT /* ceil(N/32) */>
__global__ __launch_bounds__(256) void dense_skinny_kernel(const float* __restrict__ x, int64_t x_stride,
                                                           const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ alpha, int act,
                                                           int64_t M, int K, int N, int w_stride,
                                                           float* __restrict__ out, int64_t out_stride) {
  constexpr int LDW = NT * 32;
  constexpr int LDO = NT * 32 + 4;  // output staging row stride (16-B aligned rows, bank spread)
  constexpr int LDA = 2 * KH + 4;   // x tile row stride
  // the per-wave LDS region serves both the x tile (K <= 64 only) and, later, the output tile
  constexpr int LDT = (KH <= 32 && LDA > LDO) ? LDA : LDO;
  extern __shared__ __attribute__((aligned(16))) float Ws[];  // [K][LDW], zero padded columns
  float* Ot = Ws + (size_t)K * LDW;                           // [4 waves][32][LDT]
  const bool staged = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) && (out_stride % 4 == 0);
  const int tid = threadIdx.x;
  for (int e = tid; e < K * LDW; e += 256) {
    const int kk = e / LDW, n = e - kk * LDW;
    Ws[e] = n < N ? W[(int64_t)kk * w_stride + n] : 0.f;
  }
  __syncthreads();
  const int lane = tid & 63;
  const int rl = lane & 31, hf = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t ntiles = (M + 31) >> 5;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  constexpr int kh = KH;  // K == 2*KH exactly (dispatch): no runtime guards inside the MFMA loop
  for (int64_t t = (int64_t)blockIdx.x * 4 + wv; t < ntiles; t += nwaves) {
    float a[KH];
    if constexpr (KH > 32) {
      // K = 128: row-per-lane 16-B loads straight into the operand registers (measured faster than the
      // LDS transpose at this width: 0.434 vs 0.525 ms at 1.6M x 128 x 64)
      int64_t grow = t * 32 + rl;
      grow = grow < M ? grow : M - 1;
      const f32x4* xp = reinterpret_cast<const f32x4*>(x + grow * x_stride + hf * kh);
#pragma unroll
      for (int c = 0; c < KH / 4; ++c) {
        const f32x4 v = xp[c];
        a[4 * c] = v.x;
        a[4 * c + 1] = v.y;
        a[4 * c + 2] = v.z;
        a[4 * c + 3] = v.w;
      }
    } else {
      // x tile (32 rows x K) -> LDS with fully coalesced 16-B loads (K/4 lanes per row), then each lane
      // reads ITS row's half in MFMA operand order (row stride K+4 floats: conflict-free ds_read_b128)
      constexpr int K = 2 * KH;
      constexpr int LPRW = K / 4;          // lanes per row
      constexpr int RPI = 64 / LPRW;       // rows per wave-instruction
      float* at = Ot + wv * (32 * LDT);
      const int lr = lane / LPRW, lc = lane % LPRW;
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int rr = i * RPI + lr;
        int64_t grow = t * 32 + rr;
        grow = grow < M ? grow : M - 1;
        const f32x4 v = reinterpret_cast<const f32x4*>(x + grow * x_stride)[lc];
        *reinterpret_cast<f32x4*>(at + rr * LDA + lc * 4) = v;
      }
      const f32x4* ap = reinterpret_cast<const f32x4*>(at + rl * LDA + hf * kh);
#pragma unroll
      for (int c = 0; c < KH / 4; ++c) {
        const f32x4 v = ap[c];
        a[4 * c] = v.x;
        a[4 * c + 1] = v.y;
        a[4 * c + 2] = v.z;
        a[4 * c + 3] = v.w;
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KH; ++s2) {
      const float* wrow = Ws + (size_t)(hf * kh + s2) * LDW + rl;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2], wrow[j * 32], acc[j], 0, 0, 0);
    }
    const int64_t r0 = t * 32;
    if (staged) {
      // 4-B-per-lane stores are the slow path on this part (see pairwise_dot.hip): stage the 32 x N tile in
      // a wave-private LDS region and write 16 B per lane, whole rows at a time
      float* ot = Ot + wv * (32 * LDT);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 32 + rl;
        const float bb = (bias && col < N) ? bias[col] : 0.f;
        const float al = (alpha && col < N) ? alpha[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ot[((r & 3) + 8 * (r >> 2) + 4 * hf) * LDO + col] = act_apply(acc[j][r] + bb, act, al);
      }
      const int n4 = N >> 2;
      for (int v = lane; v < 32 * n4; v += 64) {
        const int rr = v / n4, c4 = v - rr * n4;
        if (r0 + rr < M)
          *reinterpret_cast<f32x4*>(out + (r0 + rr) * out_stride + c4 * 4) =
              *reinterpret_cast<const f32x4*>(ot + rr * LDO + c4 * 4);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 32 + rl;
        if (col < N) {
          const float bb = bias ? bias[col] : 0.f;
          const float al = alpha ? alpha[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t orow = r0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (orow < M) out[orow * out_stride + col] = act_apply(acc[j][r] + bb, act, al);
          }
        }
      }
    }
  }
}

template <int KH, int NT>
static void launch_skinny(const float* x, int64_t x_stride, const float* W, const float* bias,
                          const float* alpha, int act, int64_t M, int K, int N, int w_stride, float* out,
                          int64_t out_stride, hipStream_t st) {
  const int ldt = (KH <= 32 && (2 * KH + 4) > (NT * 32 + 4)) ? (2 * KH + 4) : (NT * 32 + 4);
  const size_t lds = ((size_t)K * NT * 32 + (size_t)4 * 32 * ldt) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_skinny_kernel<KH, NT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int64_t ntiles = (M + 31) / 32;
  int64_t blocks = (ntiles + 3) / 4;
  if (blocks > 256 * 4) blocks = 256 * 4;  // persistent: W is staged once per workgroup
  hipLaunchKernelGGL((dense_skinny_kernel<KH, NT>), dim3((unsigned)blocks), dim3(256), lds, st, x, x_stride, W,
                     bias, alpha, act, M, K, N, w_stride, out, out_stride);
}

bool dense_bf16x3_dispatch(const float* x, int64_t x_stride, const float* W, const void* Wp, const float* bias,
                           const float* alpha, int act, int64_t M, int K, int N, float* out, int64_t out_stride,
                           hipStream_t st);
int64_t dense_prepared_bytes(int K, int N);
int64_t dense_b3_prepared_bytes(int K, int N);
void dense_prepare_launch(const float* W, int K, int N, void* Wp, hipStream_t st);
bool dense_f16x2_dispatch(const float* x, int64_t x_stride, const void* Wq, const float* bias, const float* alpha, int act,
                          int64_t M, int K, int N, float* out, int64_t out_stride, float* absmax, int absmax_valid,
                          float* out_absmax, hipStream_t st);
void row_absmax_launch(const float* x, int64_t x_stride, int64_t M, int K, float* absmax, hipStream_t st);

bool dense_b3_rows_dispatch(const float* x, int64_t x_stride, const float* W, const float* bias, const float* alpha,
                            int act, int64_t M, int K, int N, float* out, int64_t out_stride, hipStream_t st);

// rec_debug_force("dense", ...): 's' = keep the fp32-MFMA skinny kernels (instead of the bf16x3 row-streaming form), 't' = fp32-MFMA tiled kernel for everything, 'f' = fp32 MFMA instead of the bf16x3 kernel,
// 'b' = bf16x3 kernel wherever it is applicable (A/B measurements)
static char dense_impl() {
  const char* e = forced("dense");
  return e ? e[0] : 0;
}

}  // namespace rec

using namespace rec;

static int dense_impl_f32(const char* who, const float* x, int64_t x_stride, const float* W, const void* prepared,
                          const float* bias, const float* alpha, int32_t act, int64_t M, int32_t K, int32_t N,
                          float* out, int64_t out_stride, void* stream) {
  REC_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && x_stride >= K && out_stride >= N, REC_ESHAPE,
                "%s: bad shape M=%lld K=%d N=%d", who, (long long)M, K, N);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_PRELU, REC_EINVAL, "%s: bad act %d", who, act);
  REC_CHECK_ARG(act != REC_ACT_PRELU || alpha, REC_EINVAL, "%s: PReLU needs alpha", who);
  if (M == 0) return REC_OK;
  REC_CHECK_ARG(x && W && out, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // N in (64, 128] with M >= 1024 goes to the bf16x3 kernel below (0.37 ms vs 0.56 ms at 1.6 M x 64 x 128)
  const bool skinny = N > 8 && N <= 128 && (N <= 64 || ((N - 64) % 4 == 0 && M < 1024)) && (K == 16 || K == 32 || K == 64 || K == 128) && M >= 256 && aligned16(x) &&
                      x_stride % 4 == 0 && dense_impl() != 't' && dense_impl() != 'b';
  // the smallest layers (K, N <= 64), many rows: bf16x3 row-streaming kernel (0.277 ms vs 0.300 ms at 1.6 M x 64 x 64;
  // it loses beyond that: 0.55 vs 0.37 ms at N = 128, 0.63 vs 0.43 ms at K = 128)
  if (N > 8 && K <= 64 && N <= 64 && M >= 1024 && dense_impl() != 't' && dense_impl() != 's' && dense_impl() != 'f' &&
      dense_b3_rows_dispatch(x, x_stride, W, bias, alpha, act, M, K, N, out, out_stride, st)) {
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
  if (skinny) {
    // N in (64, 128]: two column halves (x is read twice, still ahead of the tiled kernel at these widths)
    for (int n0 = 0; n0 < N; n0 += 64) {
      const int nn = N - n0 < 64 ? N - n0 : 64;
      const int nt = (nn + 31) / 32;
      const float* Wp = W + n0;
      const float* bp = bias ? bias + n0 : nullptr;
      const float* ap = alpha ? alpha + n0 : nullptr;
      float* op = out + n0;
      bool done = false;
#define REC_SK(KH_, NT_)                                                                              \
  if (!done && K / 2 == KH_ && nt == NT_) {                                                           \
    launch_skinny<KH_, NT_>(x, x_stride, Wp, bp, ap, act, M, K, nn, N, op, out_stride, st);           \
    done = true;                                                                                      \
  }
      REC_SK(8, 1) REC_SK(8, 2) REC_SK(16, 1) REC_SK(16, 2) REC_SK(32, 1) REC_SK(32, 2) REC_SK(64, 1) REC_SK(64, 2)
#undef REC_SK
      REC_CHECK_LAUNCH(who);
    }
    return REC_OK;
  }
  // large layers: bf16x3 on the bf16 matrix cores (fp32-accurate, 2.7x the fp32 MFMA peak)
  // ... and few-row products with a LONG reduction (the weight gradients dW = X^T dY of the training step: M = the
  // layer's input width, K = the batch): on the fp32-MFMA tile kernel 479 x 8192 x 1024 took 1.2 ms and six of them
  // were 62 % of a DLRM training step
  const bool big = N > 8 && (dense_impl() == 'b' || (M >= 1024 && (int64_t)K * N >= 64 * 64) ||
                             (M >= 128 && K >= 2048 && (int64_t)K * N >= 64 * 64));
  if (big && dense_impl() != 't' && dense_impl() != 'f' &&
      dense_bf16x3_dispatch(x, x_stride, W, prepared, bias, alpha, act, M, K, N, out, out_stride, st)) {
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
  if (N == 1 && (K & 3) == 0 && aligned16(x) && aligned16(W) && x_stride % 4 == 0) {
    hipLaunchKernelGGL(dense_vec1_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, x, x_stride, W, bias, alpha,
                       act, M, K, out, out_stride);
  } else if (N <= 8) {
    hipLaunchKernelGGL((dense_narrow_kernel<8>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x,
                       x_stride, W, bias, alpha, act, M, K, N, out, out_stride);
  } else {
    const int64_t gx = (M + BM - 1) / BM;
    const int gy = (N + BN - 1) / BN;
    REC_CHECK_ARG(gx <= 0x7fffffffLL && gy <= 65535, REC_ESHAPE, "%s: M or N too large", who);
    dim3 grid((unsigned)gx, (unsigned)gy);
    hipLaunchKernelGGL(dense_mfma_kernel, grid, dim3(256), 0, st, x, x_stride, W, bias, alpha, act, M,
                       K, N, out, out_stride);
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dense_f32(const float* x, int64_t x_stride, const float* W, const float* bias,
                             const float* alpha, int32_t act, int64_t M, int32_t K, int32_t N,
                             float* out, int64_t out_stride, void* stream) {
  return dense_impl_f32("rec_dense_f32", x, x_stride, W, nullptr, bias, alpha, act, M, K, N, out, out_stride, stream);
}

// ---- few output tiles, long reduction: split-K on the bf16x3 kernel ------------------------------------------------------
namespace rec {
bool dense_bf16x3_splitk(const float* x, int64_t x_stride, const float* W, int64_t M, int K, int N, int splits, int Kc,
                         float* out_parts, hipStream_t st);
}

static int splitk_plan(int64_t M, int32_t K, int32_t N, int* Kc) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int splits = (int)((512 + tiles - 1) / tiles);                    // ~2 workgroups per CU in flight
  const int max_by_k = K / 512 > 1 ? K / 512 : 1;                  // at least 512 reduction steps per slice
  if (splits > max_by_k) splits = max_by_k;
  if (splits > 64) splits = 64;
  int kc = (K + splits - 1) / splits;
  kc = (kc + 15) / 16 * 16;
  splits = (K + kc - 1) / kc;
  *Kc = kc;
  return splits;
}

extern "C" int64_t rec_dense_splitk_workspace_bytes(int64_t M, int32_t K, int32_t N) {
  if (M < 1 || K < 1 || N < 1) return 0;
  int kc;
  const int splits = splitk_plan(M, K, N, &kc);
  return (int64_t)splits * M * N * (int64_t)sizeof(float) + rec_colsum_workspace_bytes(splits, M * N);
}

extern "C" int rec_dense_splitk_f32(const float* x, int64_t x_stride, const float* W, int64_t M, int32_t K, int32_t N,
                                    float* out, void* workspace, void* stream) {
  const char* who = "rec_dense_splitk_f32";
  REC_CHECK_ARG(x && W && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(M >= 1 && K >= 1 && N >= 1 && x_stride >= K, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(M * (int64_t)N <= 0x7fffffffLL, REC_ESHAPE, "%s: output too large", who);
  int kc;
  const int splits = splitk_plan(M, K, N, &kc);
  float* parts = static_cast<float*>(workspace);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  REC_CHECK_ARG(dense_bf16x3_splitk(x, x_stride, W, M, K, N, splits, kc, parts, st), REC_ESHAPE, "%s: grid too large", who);
  REC_CHECK_LAUNCH(who);
  // out = sum of the slices, fixed order, fp64 (deterministic)
  return rec_colsum_f32(parts, M * (int64_t)N, nullptr, 0, nullptr, splits, M * (int64_t)N, out, parts + (int64_t)splits * M * N,
                        stream);
}

extern "C" int64_t rec_dense_prepared_bytes(int32_t K, int32_t N) {
  if (K < 1 || N < 1) return 0;
  return dense_prepared_bytes(K, N);
}

extern "C" int rec_dense_prepare_f32(const float* W, int32_t K, int32_t N, void* prepared, void* stream) {
  const char* who = "rec_dense_prepare_f32";
  REC_CHECK_ARG(K >= 1 && N >= 1, REC_ESHAPE, "%s: K=%d N=%d", who, K, N);
  REC_CHECK_ARG(W && prepared && aligned16(prepared), REC_EINVAL, "%s: NULL or unaligned pointer", who);
  dense_prepare_launch(W, K, N, prepared, reinterpret_cast<hipStream_t>(stream));
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dense_prep_f32(const float* x, int64_t x_stride, const float* W, const void* prepared,
                                  const float* bias, const float* alpha, int32_t act, int64_t M, int32_t K,
                                  int32_t N, float* out, int64_t out_stride, void* stream) {
  REC_CHECK_ARG(!prepared || aligned16(prepared), REC_EINVAL, "rec_dense_prep_f32: prepared not 16-B aligned");
  return dense_impl_f32("rec_dense_prep_f32", x, x_stride, W, prepared, bias, alpha, act, M, K, N, out, out_stride,
                        stream);
}

// rec_dense_prep_f32 with one float of workspace per row: large layers whose shape the f16x2 kernel covers (aligned x rows,
// K % 32 == 0, prepared weights) run on it — three f16 MFMAs per product instead of six bf16 ones, same
// accuracy (csrc/dense_f16x2.hip); everything else is rec_dense_prep_f32.  The pass over x that finds the row maxima costs
// 4 M K bytes of reading against ~a quarter of the GEMM's time saved: it pays from N = 384 (measured: 65 536 x 3456 x 128 is
// 28 % slower with it, x 1024 27 % faster), so narrower layers take the kernel only when their producer delivered the maxima
// (absmax_valid).  rec_debug_force("dense_pipe", "0" | "s" | "d") or any forced "dense" kernel keeps the other kernels;
// "h" takes this one whatever N.
extern "C" int rec_dense_prep_rs_f32(const float* x, int64_t x_stride, const float* W, const void* prepared,
                                     const float* bias, const float* alpha, int32_t act, int64_t M, int32_t K, int32_t N,
                                     float* out, int64_t out_stride, float* row_absmax, int32_t absmax_valid,
                                     float* out_absmax, void* stream) {
  const char* who = "rec_dense_prep_rs_f32";
  REC_CHECK_ARG(!prepared || aligned16(prepared), REC_EINVAL, "%s: prepared not 16-B aligned", who);
  const char* fp = forced("dense_pipe");
  const bool other = dense_impl() != 0 || (fp && (fp[0] == '0' || fp[0] == 's' || fp[0] == 'd'));
  const bool pays = absmax_valid || N >= 384 || (fp && fp[0] == 'h');
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (prepared && row_absmax && !other && pays && x && W && out && M >= 1024 && N > 8 && K >= 32 && x_stride >= K &&
      out_stride >= N && act >= REC_ACT_NONE && act <= REC_ACT_PRELU && (act != REC_ACT_PRELU || alpha)) {
    const void* Wq = static_cast<const char*>(prepared) + dense_b3_prepared_bytes(K, N);
    if (dense_f16x2_dispatch(x, x_stride, Wq, bias, alpha, act, M, K, N, out, out_stride, row_absmax, absmax_valid, out_absmax,
                             st)) {
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
  }
  const int rc = dense_impl_f32(who, x, x_stride, W, prepared, bias, alpha, act, M, K, N, out, out_stride, stream);
  if (rc == REC_OK && out_absmax && M > 0) {      // another kernel answered: the maxima come from a pass over its output
    row_absmax_launch(out, out_stride, M, N, out_absmax, st);
    REC_CHECK_LAUNCH(who);
  }
  return rc;
}
