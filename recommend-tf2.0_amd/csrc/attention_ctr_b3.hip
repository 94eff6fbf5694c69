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

// AutoInt case only: one input tensor, S = 16, din in {16, 32}; returns false otherwise
bool mha_ctr_b3_dispatch(const float* xq, const float* xk, const float* xv, int64_t B, int N, int din, const float* Wq,
                         const float* Wk, const float* Wv, const float* W0, int H, int S, int act, float* out,
                         hipStream_t st) {
  if (S != 16 || !(din == 16 || din == 32) || N > 64 || xq != xk || xk != xv) return false;
  if (!aligned16(xq) || !aligned16(out) || H > 8) return false;
  const size_t lds = (size_t)4 * H * 3 * 4 * 16 * sizeof(u32x4);
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  const int NT = (N + 15) / 16;
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
