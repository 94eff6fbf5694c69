// K6 on the bf16 matrix cores with fp32 accuracy ("bf16x3"): the AutoInt interacting layer
// (src/ctr/layers/modules.py:285-325) for head size S = 16, input width din in {16, 32}, one input tensor
// (xq == xk == xv, the AutoInt case), N <= 64 fields.  One wave per sample; after the weights are staged nothing
// touches LDS except the weight fragments, and the sample's rows are loaded from global memory straight in MFMA
// operand layout.
//
// v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, g = lane >> 4) supplies 8 k-values 8g..8g+7 of row/column i for A
// and B alike; the accumulator has its column on lane & 15 and rows 4g..4g+3 in its four registers.  That makes
// every intermediate the next product's operand WITHOUT moving data, by choosing orientations:
//   Q^T, K^T = W^T X^T   (A = weight fragment, B = row fragment): column = field on the lane, rows = head dims 4g+r
//   V        = X W       (A = row fragment, B = weight fragment): column = head dim on the lane, rows = fields 4g+r
//   S^T      = K Q^T     : lane (key, g) holds K[key][4g..4g+3] = its K^T accumulator, lane (query, g) its Q^T one
//   O^T      = V^T P^T   : lane (dim, g) holds V[keys 4g..4g+3][dim] = its V accumulator; lane (query, g) holds
//                          P[query][keys 4g..4g+3] = its score accumulator
//   R^T      = W0^T X^T  : residual in the O^T layout
// with the k-slot convention "element j < 4 of group g carries index 4g + j, elements 4..7 are zero" on both
// operands (only 16 of the 32 k-slots carry data in the score / PV products: S = 16).  Each fp32 value is split
// exactly into three bf16 terms and each product rebuilt from six MFMAs (hh, hm, mh, hl, lh, mm): fp32 accuracy.
#include <math.h>
#include <stdlib.h>

#include "attention_ctr.h"
#include "bf16x3.h"
#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace cb3 {
struct Frag {  // three bf16x8 planes
  u32x4 p[3];
};
template <int NV>
__device__ __forceinline__ Frag split(const float* x) {
  Frag f;
  bf16x3::split<NV>(x, f.p[0], f.p[1], f.p[2]);
  return f;
}
__device__ __forceinline__ Frag split4(const f32x4 a) {
  const float x[4] = {a.x, a.y, a.z, a.w};
  return split<4>(x);
}
__device__ __forceinline__ f32x4 mfma6(const Frag& a, const Frag& b, f32x4 c) {
  const bf16x8 ah = __builtin_bit_cast(bf16x8, a.p[0]), am = __builtin_bit_cast(bf16x8, a.p[1]),
               al = __builtin_bit_cast(bf16x8, a.p[2]);
  const bf16x8 bh = __builtin_bit_cast(bf16x8, b.p[0]), bm = __builtin_bit_cast(bf16x8, b.p[1]),
               bl = __builtin_bit_cast(bf16x8, b.p[2]);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
  return c;
}
}  // namespace cb3

template <int NT>  // 16-field tiles: N <= 16 * NT
__global__ __launch_bounds__(256) void mha_ctr_b3_kernel(const float* __restrict__ x, int64_t B, int N, int din,
                                                         const float* __restrict__ Wq, const float* __restrict__ Wk,
                                                         const float* __restrict__ Wv, const float* __restrict__ W0,
                                                         int H, int act, float* __restrict__ out) {
  using namespace cb3;
  // weight fragments [proj 4][head][plane 3][k-group 4][col 16]: element j of (g, col) = W[8g + j][16 head + col]
  extern __shared__ __attribute__((aligned(16))) u32x4 wf[];
  const int HS = H * 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < 4 * H * 64; e += 256) {
    const int col = e & 15, g = (e >> 4) & 3, hh = (e >> 6) % H, pr = e / (64 * H);
    const float* W = pr == 0 ? Wq : (pr == 1 ? Wk : (pr == 2 ? Wv : W0));
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = 8 * g + j;
      w[j] = (W && kk < din) ? W[(int64_t)kk * HS + hh * 16 + col] : 0.f;
    }
    const Frag f = split<8>(w);
#pragma unroll
    for (int p = 0; p < 3; ++p) wf[(((pr * H + hh) * 3 + p) * 4 + g) * 16 + col] = f.p[p];
  }
  __syncthreads();
  const int lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int64_t b = (int64_t)blockIdx.x * 4 + (tid >> 6);
  if (b >= B) return;  // wave-uniform; no barrier below
  auto wfrag = [&](int pr, int hh) {
    Frag f;
#pragma unroll
    for (int p = 0; p < 3; ++p) f.p[p] = wf[(((pr * H + hh) * 3 + p) * 4 + g) * 16 + lr];
    return f;
  };

  // the sample's rows in operand layout: lane (field, g) holds x[field][8g .. 8g+7]
  Frag xf[NT];
#pragma unroll
  for (int rt = 0; rt < NT; ++rt) {
    const int n = rt * 16 + lr;
    float v[8];
    if (n < N && 8 * g < din) {
      const float* p = x + (b * N + n) * (int64_t)din + 8 * g;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
      v[0] = a0.x, v[1] = a0.y, v[2] = a0.z, v[3] = a0.w, v[4] = a1.x, v[5] = a1.y, v[6] = a1.z, v[7] = a1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    xf[rt] = split<8>(v);
  }
  const float scale = 4.0f * 1.4426950408889634f;  // "/ (S ** -0.5)" = x sqrt(16), folded with log2(e)
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  for (int h = 0; h < H; ++h) {
    const Frag wq = wfrag(0, h), wk = wfrag(1, h), wv = wfrag(2, h);
    Frag qf[NT], kf[NT], vf[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      f32x4 a = mfma6(wq, xf[rt], zero);   // Q^T: column = field, rows = dims 4g + r
      f32x4 c = mfma6(wk, xf[rt], zero);   // K^T
      f32x4 d = mfma6(xf[rt], wv, zero);   // V: column = dim, rows = fields 4g + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = act_apply(a[r], act, 0.f);
        c[r] = act_apply(c[r], act, 0.f);
        d[r] = act_apply(d[r], act, 0.f);
      }
      qf[rt] = split4(a);
      kf[rt] = split4(c);
      vf[rt] = split4(d);
    }
    const Frag w0 = W0 ? wfrag(3, h) : Frag{};
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      // transposed scores of this query tile against every key tile: column = query, rows = keys 4g + r
      f32x4 sc[NT];
      float mloc = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        sc[kt] = mfma6(kf[kt], qf[qt], zero);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const float v = key < N ? sc[kt][r] * scale : -INFINITY;
          sc[kt][r] = v;
          mloc = fmaxf(mloc, v);
        }
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      float lsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = exp2f(sc[kt][r] - mloc);  // pad keys: exp2(-inf) = 0
          sc[kt][r] = p;
          lsum += p;
        }
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      const float inv = 1.f / lsum;
      // O^T = V^T P^T
      f32x4 o = zero;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        const Frag pf = split4(sc[kt] * inv);
        o = mfma6(vf[kt], pf, o);
      }
      if (W0) {  // residual branch in the same (query on lane, dims in registers) layout
        const f32x4 rr = mfma6(w0, xf[qt], zero);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_nan(o[r] + act_apply(rr[r], act, 0.f));
      }
      const int qi = qt * 16 + lr;
      if (qi < N) *reinterpret_cast<f32x4*>(out + (b * N + qi) * (int64_t)HS + h * 16 + 4 * g) = o;
    }
  }
}

// ---- the same layer on the fp32 matrix cores, register-resident (round 2) ------------------------------------------
// The orientation scheme above maps one-to-one onto v_mfma_f32_16x16x4_f32: lane (i = lane & 15, g = lane >> 4) supplies
// ONE k-value per instruction, and step r of four uses k = 4g + r — exactly accumulator register r of a previous
// product (column on the lane, rows 4g + r in the registers).  So every intermediate is still the next product's
// operand as it stands, a 16-deep contraction is four MFMAs on the four registers of each operand, and nothing is
// split: the ~1400 VALU instructions per sample of the three-term bf16 split disappear, the kernel drops from 178 to
// ~110 registers (4 instead of 2 waves per SIMD: all 4096 waves of config 3 resident at once) and computes in exact
// fp32 (an fmaf chain: +-inf / NaN behave as in the reference's fp32 ops).  Cost: 128 matrix-pipe cycles per 16x16x16
// product instead of 96, i.e. ~7-11k MFMA cycles per sample — still far below what the split version spent per sample.
namespace cf32 {
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}
}  // namespace cf32

// ---- a STACK of interacting layers in one launch (AutoInt, BASELINE configs[2]: 3 layers) ---------------------------
// Layer l's output for head h — O^T with the query on the lane and dims 4g + r in the registers — is exactly the row
// fragment x[field][16 h + 4g + r] that layer l+1 loads as its operand (din = 16 H, k-step h).  So the activations of
// a sample never leave the wave's registers between layers: one launch, weights of all layers staged in LDS once per
// workgroup, the (B, N, 16 H) intermediates are neither written nor re-read, and the per-launch fixed costs (launch,
// weight staging, first-touch latency of the rows: most of a 37-us layer at 4096 samples = one wave each) are paid once.


namespace cf32 {
// VALU diet (rocprofv3 counters, round 2: the first version issued 2 169 VALU instructions per sample and layer against
// 240 MFMAs — the softmax, not the matrix pipe, set its 32 us): v_exp_f32 directly (arguments are <= 0: no range
// fix-up), the sqrt(S) log2(e) factor folded into Q once per element instead of once per score, the 1/sum applied to
// the 4 output registers instead of the 12 probabilities, the key mask only on the last key tile, relu as a compile-time
// case.
// All-reduce over the wave's four 16-lane rows (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: gfx950's v_permlane32_swap
// (upper half of one register <-> lower half of the other) and v_permlane16_swap (odd rows <-> even rows) exchange the
// halves in registers.  __shfl_xor(v, 16 | 32) compiles to ds_bpermute_b32, an LDS-pipe round trip, and the softmax has
// four of them in its dependency chain per (head, query tile): max -> exp -> sum -> 1/sum.
__device__ __forceinline__ float rows_max(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int ACT>
__device__ __forceinline__ float act_ct(float v, int act) {
  if constexpr (ACT == REC_ACT_RELU) return relu_nan(v);
  else if constexpr (ACT == REC_ACT_NONE) return v;
  else return act_apply(v, act, 0.f);
}

// Wave priority by phase (round 3).  A layer is a chain of matrix bursts (12-36 MFMAs) and VALU phases (activations,
// softmax); three waves share a SIMD's matrix pipe and its vector issue port, arbitrated by priority, then age
// (MI355X_MICROARCH.md § Two waves per SIMD).  With every wave at priority 0 the pipe was 0.57 busy: a wave that reaches a
// matrix burst queues behind older waves' VALU.  2 (shipped) = matrix bursts at priority 3, VALU phases at 0: the wave that
// has MFMAs to issue wins the port (an MFMA holds it for a few cycles of its 32), VALU fills the gaps.  Same-box A/B
// (tools/exp/autoint_prio_ab.sh, profiles/r03_autoint_prio_ab*.txt): 0 = 77.3-77.8 us, 1 (VALU phases high) = 75.0-75.2,
// 2 = 72.9-73.3; the s_setprio instructions also pin the phase boundaries for the scheduler (146 VGPRs, no spill,
// instead of 168 + 6 spilled).
#ifndef REC_AUTOINT_PRIO
#define REC_AUTOINT_PRIO 2
#endif

__device__ __forceinline__ void prio_valu() {
#if REC_AUTOINT_PRIO == 1
  __builtin_amdgcn_s_setprio(3);
#elif REC_AUTOINT_PRIO == 2
  __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void prio_mfma() {
#if REC_AUTOINT_PRIO == 1
  __builtin_amdgcn_s_setprio(0);
#elif REC_AUTOINT_PRIO == 2
  __builtin_amdgcn_s_setprio(3);
#endif
}

template <int NT, int KS, int H, int ACT>
__device__ __forceinline__ void ctr_layer(const f32x4 (&xf)[NT][KS], const f32x4* __restrict__ wl, bool has_res, int act,
                                          int N, int lr, int g, f32x4 (&xo)[NT][H]) {
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float scale = 4.0f * 1.4426950408889634f;  // "/ (S ** -0.5)" = x sqrt(16), folded with log2(e)
#pragma unroll
  for (int h = 0; h < H; ++h) {
    f32x4 wq[KS], wk[KS], wv[KS], w0[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wq[ks] = wl[(((0 * H + h) * KS + ks) * 4 + g) * 16 + lr];
      wk[ks] = wl[(((1 * H + h) * KS + ks) * 4 + g) * 16 + lr];
      wv[ks] = wl[(((2 * H + h) * KS + ks) * 4 + g) * 16 + lr];
      w0[ks] = has_res ? wl[(((3 * H + h) * KS + ks) * 4 + g) * 16 + lr] : zero;
    }
    f32x4 qf[NT], kf[NT], vf[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      f32x4 a = zero, c = zero, d = zero;
      prio_mfma();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a = mfma4(wq[ks], xf[rt][ks], a);   // Q^T: column = field, rows = dims 4g + r
        c = mfma4(wk[ks], xf[rt][ks], c);   // K^T
        d = mfma4(xf[rt][ks], wv[ks], d);   // V: column = dim, rows = fields 4g + r
      }
      prio_valu();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = act_ct<ACT>(a[r], act) * scale;   // scores are linear in Q: scale once here
        c[r] = act_ct<ACT>(c[r], act);
        d[r] = act_ct<ACT>(d[r], act);
      }
      qf[rt] = a, kf[rt] = c, vf[rt] = d;
    }
    // (round 3 A/B: the three query tiles as ONE scores burst / one softmax phase / one PV burst — a third of the phase
    // switches for 11 more registers — 73.2-73.5 us either way, profiles/r03_autoint_prio_ab3.txt; not kept)
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 sc[NT];
      float mloc = -INFINITY;
      prio_mfma();
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) sc[kt] = mfma4(kf[kt], qf[qt], zero);
      prio_valu();
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        // sc: transposed scores (x sqrt(S) log2 e): column = query, rows = keys 4g + r
        if (kt == NT - 1) {                      // only the last key tile can hold padding keys
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[kt][r] = kt * 16 + 4 * g + r < N ? sc[kt][r] : -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sc[kt][r]);
      }
      mloc = rows_max(mloc);
      float lsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(sc[kt][r] - mloc);  // argument <= 0; pad keys: exp2(-inf) = 0
          sc[kt][r] = p;
          lsum += p;
        }
      lsum = rows_sum(lsum);
      const float inv = __builtin_amdgcn_rcpf(lsum);   // v_rcp_f32 (1 ulp; lsum in [1, N]) instead of the IEEE division sequence
      f32x4 o = zero;
      prio_mfma();
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) o = mfma4(vf[kt], sc[kt], o);   // O^T = V^T P^T (unnormalised)
      f32x4 rr = zero;
      if (has_res) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) rr = mfma4(w0[ks], xf[qt][ks], rr);
      }
      prio_valu();
      o *= inv;
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_nan(o[r] + act_ct<ACT>(rr[r], act));
      }
      xo[qt][h] = o;   // rows of padded fields (>= N) carry values nobody reads: keys >= N are masked, queries >= N unused
    }
  }
}

template <int KS, int H>
__device__ __forceinline__ void stage_weights(f32x4* __restrict__ wl, const float* Wq, const float* Wk, const float* Wv,
                                              const float* W0, int tid) {
  constexpr int HS = H * 16;
  for (int e = tid; e < 4 * H * KS * 64; e += 256) {
    const int col = e & 15, g = (e >> 4) & 3, ks = (e >> 6) % KS, hh = (e / (64 * KS)) % H, pr = e / (64 * KS * H);
    const float* W = pr == 0 ? Wq : (pr == 1 ? Wk : (pr == 2 ? Wv : W0));
    f32x4 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = W ? W[(int64_t)(16 * ks + 4 * g + r) * HS + hh * 16 + col] : 0.f;
    wl[(((pr * H + hh) * KS + ks) * 4 + g) * 16 + col] = w;
  }
}
}  // namespace cf32

// workgroups per CU the register budget is cut for (tools/exp/autoint_wg_ab.sh, configs[2], same box): unconstrained 196 VGPRs
// (two waves per SIMD) 83.6 us; 2 -> 176 VGPRs 80.8 us; 3 -> 168 VGPRs + 2 spilled, three waves per SIMD, 76.6 us (shipped):
// a third wave per SIMD overlaps one sample's softmax VALU with another's matrix work.  Round 3, with the phase priorities:
// 2 / 3 / 4 workgroups per CU = 76.2 / 73.3 / 79.5 us (4 spills 23 registers)
#ifndef REC_AUTOINT_MINWG
#define REC_AUTOINT_MINWG 3
#endif
template <int NT, int KS0, int H, int ACT, bool IO = false>
// (wider shapes — NT = 4 or din = 32 with two heads — would spill 26-62 registers under that budget: they keep two)
__global__ __launch_bounds__(256, (NT <= 3 && KS0 == 1) ? REC_AUTOINT_MINWG : 2) void mha_ctr_stack_kernel(const float* __restrict__ x, int64_t B, int N, CtrStackArgs wa,
                                                            int L, int act, float* __restrict__ out, CtrFusedIo io = {},
                                                            int cus = 0, int stagger = 0) {
  using namespace cf32;
  extern __shared__ __attribute__((aligned(16))) f32x4 wstack[];
  constexpr int din0 = 16 * KS0, HS = 16 * H;
  constexpr int SZ0 = 4 * H * KS0 * 64, SZ1 = 4 * H * H * 64;   // f32x4 per layer
  const int tid = threadIdx.x;
  stage_weights<KS0, H>(wstack, wa.Wq[0], wa.Wk[0], wa.Wv[0], wa.W0[0], tid);
  for (int l = 1; l < L; ++l) stage_weights<H, H>(wstack + SZ0 + (l - 1) * SZ1, wa.Wq[l], wa.Wk[l], wa.Wv[l], wa.W0[l], tid);
  __syncthreads();
  const int lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // A/B knob (rec_debug_force("autoint_stagger", n)): the workgroups that share a CU start n x 1024 cycles apart, so the
  // waves of a SIMD are not all in their matrix phase (or all in their softmax phase) at the same time
  if (stagger > 0 && cus > 0) {
    const int slot = (int)(blockIdx.x / (unsigned)cus);
    for (int i = 0; i < slot * stagger; ++i) __builtin_amdgcn_s_sleep(16);
  }
  // the sample's rows in operand layout: lane (field, g) holds x[field][16 ks + 4g .. + 3]
  auto load_x0 = [&](int64_t b, f32x4(&x0)[NT][KS0]) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      const int n = rt * 16 + lr;
#pragma unroll
      for (int ks = 0; ks < KS0; ++ks) {
        if constexpr (IO) {
          f32x4 v = zero;
          if (n < io.n_sparse) {          // an embedding row, fetched by id; out-of-range ids read as zero rows
            const int32_t id = io.ids[b * io.ids_stride + n];
            const bool ok = (uint32_t)id < (uint32_t)io.ts.vocab[n];
            if (!ok && io.oob) *io.oob = 1;
            const f32x4 row = *reinterpret_cast<const f32x4*>(io.ts.base[n] + (int64_t)(ok ? id : 0) * din0 + 16 * ks + 4 * g);
            v = ok ? row : zero;
          } else if (n < N) {             // a dense value times its embedding row
            const int j = n - io.n_sparse;
            v = *reinterpret_cast<const f32x4*>(io.dense_embed + j * din0 + 16 * ks + 4 * g) * io.dense[b * io.dense_stride + j];
          }
          x0[rt][ks] = v;
        } else {
          x0[rt][ks] = n < N ? *reinterpret_cast<const f32x4*>(x + (b * N + n) * (int64_t)din0 + 16 * ks + 4 * g) : zero;
        }
      }
    }
  };
  // persistent over samples: b = wave + k * waves (weights are staged once per workgroup).  (Round 3 A/B: requesting the
  // NEXT sample's rows before this sample's layers costs 12 more registers under the 168 cap — 14 spilled instead of 2 —
  // and runs 80.2 us against 76.8 (with the phase priorities below and no spill: 73.7 vs 73.3, still nothing); starting
  // the workgroups of a CU 2-16 k cycles apart: 80.6-83.5 us.  Neither the input latency nor phase lockstep is what the
  // kernel waits for: profiles/r03_autoint_ab.txt, r03_autoint_prio_ab2.txt.)
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + (tid >> 6); b < B; b += nwaves) {
    f32x4 x0[NT][KS0];
    load_x0(b, x0);
    f32x4 ya[NT][H], yb[NT][H];
    ctr_layer<NT, KS0, H, ACT>(x0, wstack, wa.W0[0] != nullptr, act, N, lr, g, ya);
    for (int l = 1; l < L; ++l) {   // ping-pong in registers
      ctr_layer<NT, H, H, ACT>(ya, wstack + SZ0 + (l - 1) * SZ1, wa.W0[l] != nullptr, act, N, lr, g, yb);
#pragma unroll
      for (int rt = 0; rt < NT; ++rt)
#pragma unroll
        for (int h = 0; h < H; ++h) ya[rt][h] = yb[rt][h];
    }
    if constexpr (IO) {
      // sigmoid(Dense(1)(flatten(out))): lane (field, g) holds out[field][16 h + 4g .. + 3]
      float acc = 0.f;
#pragma unroll
      for (int qt = 0; qt < NT; ++qt) {
        const int qi = qt * 16 + lr;
        if (qi < N) {
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(io.head_w + qi * HS + h * 16 + 4 * g);
            const f32x4 pr = ya[qt][h] * wv;
            acc += (pr.x + pr.y) + (pr.z + pr.w);
          }
        }
      }
      acc = wave_sum(acc) + (io.head_b ? io.head_b[0] : 0.f);
      if (lane == 0) io.head_out[b] = act_apply(acc, REC_ACT_SIGMOID, 0.f);
      if (!out) continue;
    }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      const int qi = qt * 16 + lr;
      if (qi < N) {
#pragma unroll
        for (int h = 0; h < H; ++h) *reinterpret_cast<f32x4*>(out + (b * N + qi) * (int64_t)HS + h * 16 + 4 * g) = ya[qt][h];
      }
    }
  }
}

// returns false when the stack is not covered (the caller runs the layers one by one)
bool mha_ctr_stack_dispatch(const float* x, int64_t B, int N, int din, const CtrStackArgs& wa, int L, int H, int S, int act,
                            float* out, hipStream_t st, const CtrFusedIo* io) {
  if (S != 16 || !(din == 16 || din == 32) || N > 64 || N < 1 || L < 1 || L > 4 || !(H == 1 || H == 2)) return false;
  if (!io && (!aligned16(x) || !aligned16(out))) return false;
  if (io && ((out && !aligned16(out)) || !aligned16(io->head_w) || !aligned16(io->dense_embed))) return false;
  for (int l = 0; l < L; ++l)
    if (!wa.Wq[l] || !wa.Wk[l] || !wa.Wv[l]) return false;
  const int NT = (N + 15) / 16, KS0 = din / 16;
  const size_t lds = sizeof(f32x4) * ((size_t)4 * H * KS0 * 64 + (size_t)(L - 1) * 4 * H * H * 64);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int64_t blocks = (B + 3) / 4;
  // workgroups per CU that are RESIDENT at once (the kernels use 196 VGPRs: two waves per SIMD = two workgroups of four
  // waves): with more, the surplus workgroups run as a second round and stage the weights again.  rec_debug_force("autoint_wg", n): A/B
  const int wg_per_cu = forced("autoint_wg") ? atoi(forced("autoint_wg")) : 0;
  const int wg_res = wg_per_cu > 0 ? wg_per_cu : ((NT <= 3 && KS0 == 1) ? REC_AUTOINT_MINWG : 2);   // as the kernel's launch bounds
  if (blocks > (int64_t)cus * wg_res) blocks = (int64_t)cus * wg_res;
  const dim3 grid((unsigned)blocks), block(256);
  const int stagger = forced("autoint_stagger") ? atoi(forced("autoint_stagger")) : 0;
#define REC_CST(NT_, KS_, H_)                                                                                       \
  do {                                                                                                              \
    if (io && act == REC_ACT_RELU)                                                                                  \
      hipLaunchKernelGGL((mha_ctr_stack_kernel<NT_, KS_, H_, REC_ACT_RELU, true>), grid, block, lds, st, x, B, N, wa, L, \
                         act, out, *io, cus, stagger);                                                              \
    else if (io)                                                                                                    \
      hipLaunchKernelGGL((mha_ctr_stack_kernel<NT_, KS_, H_, -1, true>), grid, block, lds, st, x, B, N, wa, L, act, out, \
                         *io, cus, stagger);                                                                        \
    else if (act == REC_ACT_RELU)                                                                                   \
      hipLaunchKernelGGL((mha_ctr_stack_kernel<NT_, KS_, H_, REC_ACT_RELU>), grid, block, lds, st, x, B, N, wa, L, act, \
                         out, CtrFusedIo{}, cus, stagger);                                                          \
    else                                                                                                            \
      hipLaunchKernelGGL((mha_ctr_stack_kernel<NT_, KS_, H_, -1>), grid, block, lds, st, x, B, N, wa, L, act, out,  \
                         CtrFusedIo{}, cus, stagger);                                                               \
  } while (0)
#define REC_CST_NT(KS_, H_)              \
  do {                                   \
    if (NT == 1) REC_CST(1, KS_, H_);    \
    else if (NT == 2) REC_CST(2, KS_, H_); \
    else if (NT == 3) REC_CST(3, KS_, H_); \
    else REC_CST(4, KS_, H_);            \
  } while (0)
  if (KS0 == 1 && H == 1) REC_CST_NT(1, 1);
  else if (KS0 == 1 && H == 2) REC_CST_NT(1, 2);
  else if (KS0 == 2 && H == 1) REC_CST_NT(2, 1);
  else REC_CST_NT(2, 2);
#undef REC_CST_NT
#undef REC_CST
  return true;
}

// AutoInt case only: one input tensor, S = 16, din in {16, 32}; returns false otherwise
bool mha_ctr_b3_dispatch(const float* xq, const float* xk, const float* xv, int64_t B, int N, int din, const float* Wq,
                         const float* Wk, const float* Wv, const float* W0, int H, int S, int act, float* out,
                         hipStream_t st) {
  if (S != 16 || !(din == 16 || din == 32) || N > 64 || xq != xk || xk != xv) return false;
  if (!aligned16(xq) || !aligned16(out) || H > 8) return false;
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  const int NT = (N + 15) / 16;
  // default for H <= 2: the register-resident fp32-MFMA kernel (a stack of one layer);
  // rec_debug_force("mha_ctr", "b") keeps the bf16x3 kernel of this file for tests / A/B, and it serves H > 2
  const bool use_b3 = forced("mha_ctr") && forced("mha_ctr")[0] == 'b';
  if (!use_b3) {
    CtrStackArgs wa{};
    wa.Wq[0] = Wq, wa.Wk[0] = Wk, wa.Wv[0] = Wv, wa.W0[0] = W0;
    if (mha_ctr_stack_dispatch(xq, B, N, din, wa, 1, H, S, act, out, st, nullptr)) return true;
  }
  const size_t lds = (size_t)4 * H * 3 * 4 * 16 * sizeof(u32x4);
#define REC_CB3(NT_)                                                                                              \
  hipLaunchKernelGGL((mha_ctr_b3_kernel<NT_>), grid, block, lds, st, xq, B, N, din, Wq, Wk, Wv, W0, H, act, out)
  if (NT == 1) REC_CB3(1);
  else if (NT == 2) REC_CB3(2);
  else if (NT == 3) REC_CB3(3);
  else REC_CB3(4);
#undef REC_CB3
  return true;
}

}  // namespace rec
