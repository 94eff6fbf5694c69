// Row-wise interaction kernels (one wave, or a sub-wave lane group, per sample row):
//   K3  rec_fm_layer_f32          FM layer of DeepFM          src/ctr/layers/modules.py:57-72
//   K4  rec_cross_f32             DCN CrossNetwork            src/ctr/layers/modules.py:105-112
//   K2  rec_fm_onehot_f32         ctr FM model, gather form   src/ctr/fm/model.py:34-53
//   K9  rec_layernorm_residual_f32  LN(x + r) [* row mask]    src/match/layers/modules.py:173-185
//   K10 rec_gather_dot_scores_f32 SASRec last-position scores src/match/sasrec/model.py:88-96
// All are HBM-bound streaming reductions: every input element is read exactly once, rows are read
// with 16-B vector loads where the address allows (scalar head/tail otherwise), reductions are
// wavefront shuffles, no LDS.
#include <stdlib.h>

#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- a row reader that vectorises the 16-B aligned body of an arbitrarily aligned fp32 row ----
// calls f(value, column) for every column owned by this lane (lane-strided, wave-wide)
template <typename Fn>
__device__ __forceinline__ void for_each_in_row(const float* __restrict__ p, int L, int lane, Fn f) {
  const int mis = (int)((reinterpret_cast<uintptr_t>(p) >> 2) & 3);
  int head = mis ? 4 - mis : 0;
  if (head > L) head = L;
  if (lane < head) f(p[lane], lane);
  const int nvec = (L - head) >> 2;
  const f32x4* pv = reinterpret_cast<const f32x4*>(p + head);
  for (int v = lane; v < nvec; v += 64) {
    const f32x4 t = pv[v];
    const int c = head + 4 * v;
    f(t.x, c);
    f(t.y, c + 1);
    f(t.z, c + 2);
    f(t.w, c + 3);
  }
  const int done = head + 4 * nvec;
  if (done + lane < L) f(p[done + lane], done + lane);
}

// ------------------------------------------------------------------------------------------------
// K3 — FM layer.  Pass 1: per row  lin_b = first[b].w,  sec_b = 0.5((sum x)^2 - sum x^2);
// out[b] = sec_b, and each block writes the sum of its rows' lin_b to partial[block].
// Pass 2: every block re-reduces the (<= 4096) partials in the same fixed order (bit-identical
// in all blocks, run-to-run reproducible) and adds the batch-global scalar to its rows.
// ------------------------------------------------------------------------------------------------
constexpr int kFmMaxPartials = 4096;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The second-order term 0.5((sum x)^2 - sum x^2) cancels catastrophically in fp32 (both terms
// ~ M*var(x), difference ~ 0): the kernel is HBM-bound, so the three running sums are kept in
// fp64 for free and rounded to fp32 once.
__global__ __launch_bounds__(256) void fm_layer_rows_kernel(
    const float* __restrict__ first, int64_t first_stride, int L1, const float* __restrict__ w,
    const float* __restrict__ second, int64_t second_stride, int M, int64_t B, int rows_per_wave,
    float* __restrict__ out, double* __restrict__ partial) {
  __shared__ double wsum[4];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wv) * rows_per_wave;
  double lin_acc = 0.0;  // lane-partial of this wave's rows
  for (int r = 0; r < rows_per_wave; ++r) {
    const int64_t b = row0 + r;
    if (b >= B) break;
    double lin = 0.0, s = 0.0, q = 0.0;
    for_each_in_row(first + b * first_stride, L1, lane,
                    [&](float v, int c) { lin = fma((double)v, (double)w[c], lin); });
    for_each_in_row(second + b * second_stride, M, lane, [&](float v, int) {
      s += (double)v;
      q = fma((double)v, (double)v, q);
    });
    s = wave_sum_f64(s);
    q = wave_sum_f64(q);
    lin_acc += lin;
    if (lane == 0) out[b] = (float)(0.5 * (s * s - q));
  }
  lin_acc = wave_sum_f64(lin_acc);
  if (lane == 0) wsum[wv] = lin_acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void fm_layer_finish_kernel(const double* __restrict__ partial,
                                                              int npartial, int64_t B,
                                                              float* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < npartial; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double first_order = red[0];
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b < B) out[b] = (float)((double)out[b] + first_order);
}

// ------------------------------------------------------------------------------------------------
// K1+K3 fused (DeepFM): LPR = D/4 lanes own a sample, stream its F rows (16 B per lane, 8 loads in
// flight), copy them into the concat buffer and keep s = sum x, q = sum x^2, lin = sum x w in fp64.
// The batch-global first-order scalar is finished by fm_layer_finish_kernel.
// ------------------------------------------------------------------------------------------------
template <int LPR, int IDS_F32>
__global__ __launch_bounds__(256) void gather_fm_kernel(TableSet ts, const void* __restrict__ ids,
                                                        int64_t ids_stride, int F,
                                                        const float* __restrict__ dense, int64_t dense_stride,
                                                        int nd, const float* __restrict__ w, int64_t B,
                                                        float* __restrict__ emb_out, int64_t emb_stride,
                                                        float* __restrict__ fm_out, double* __restrict__ partial,
                                                        int* __restrict__ oob, float* __restrict__ row_absmax) {
  constexpr int D = LPR * 4;
  constexpr int SPW = 64 / LPR;
  __shared__ double wsum[4];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int sl = lane % LPR, sw = lane / LPR;
  const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wv) * SPW + sw;
  const bool live = b_raw < B;
  const int64_t b = live ? b_raw : B - 1;
  double s = 0.0, q = 0.0, lin = 0.0;
  float amax = 0.f;      // max |element| of the sample's concat row: the scale the DNN's first Dense (f16x2 kernel) needs
  constexpr int U = 8;
  for (int f0 = 0; f0 < F; f0 += U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u < F ? f0 + u : F - 1;
      const int32_t id = load_id<IDS_F32>(ids, b * ids_stride + f);
      const bool ok = (uint32_t)id < (uint32_t)ts.vocab[f];
      if (!ok && oob && live) *oob = 1;
      const f32x4 t = *reinterpret_cast<const f32x4*>(ts.base[f] + (int64_t)(ok ? id : 0) * D + sl * 4);
      v[u] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u;
      if (f < F) {
        const int oc = ts.out_col[f] + sl * 4;
        if (live) *reinterpret_cast<f32x4*>(emb_out + b * emb_stride + oc) = v[u];
        const f32x4 wv4 = *reinterpret_cast<const f32x4*>(w + nd + f * D + sl * 4);  // host checks nd % 4 == 0
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
        s += (double)v[u].x + (double)v[u].y + (double)v[u].z + (double)v[u].w;
        q = fma((double)v[u].x, (double)v[u].x, q);
        q = fma((double)v[u].y, (double)v[u].y, q);
        q = fma((double)v[u].z, (double)v[u].z, q);
        q = fma((double)v[u].w, (double)v[u].w, q);
        lin = fma((double)v[u].x, (double)wv4.x, lin);
        lin = fma((double)v[u].y, (double)wv4.y, lin);
        lin = fma((double)v[u].z, (double)wv4.z, lin);
        lin = fma((double)v[u].w, (double)wv4.w, lin);
      }
    }
  }
  // dense part of the first-order term: lanes of the group stride over the nd dense columns
  for (int d0 = sl; d0 < nd; d0 += LPR) {
    const float dv = dense[b * dense_stride + d0];
    lin = fma((double)dv, (double)w[d0], lin);
    amax = fmaxf(amax, fabsf(dv));
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
    amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  }
  if (live && sl == 0) fm_out[b] = (float)(0.5 * (s * s - q));
  if (row_absmax && live && sl == 0) row_absmax[b] = amax;
  double lin_w = live ? lin : 0.0;
  lin_w = wave_sum_f64(lin_w);
  if (lane == 0) wsum[wv] = lin_w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// ------------------------------------------------------------------------------------------------
// K4 — CrossNetwork: x0 and x_l of one row live in VGPRs (VPL float4 per lane, dim <= 256*VPL);
// per layer one wave-wide dot (shuffle reduce) and one AXPY; w_l / b_l stream from L2.
// ------------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void cross_kernel(const float* __restrict__ x, int64_t x_stride,
                                                    int dim, const float* __restrict__ w,
                                                    const float* __restrict__ bv, int L, int64_t B,
                                                    float* __restrict__ out, int64_t out_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int nvec = dim >> 2;
  f32x4 x0[VPL], xl[VPL];
  const f32x4* px = reinterpret_cast<const f32x4*>(x + b * x_stride);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int idx = lane + 64 * v;
    x0[v] = idx < nvec ? px[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
    xl[v] = x0[v];
  }
  for (int l = 0; l < L; ++l) {
    const f32x4* pw = reinterpret_cast<const f32x4*>(w + (int64_t)l * dim);
    const f32x4* pb = reinterpret_cast<const f32x4*>(bv + (int64_t)l * dim);
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int idx = lane + 64 * v;
      if (idx < nvec) {
        const f32x4 wv = pw[idx];
        s = fmaf(xl[v].x, wv.x, s);
        s = fmaf(xl[v].y, wv.y, s);
        s = fmaf(xl[v].z, wv.z, s);
        s = fmaf(xl[v].w, wv.w, s);
      }
    }
    s = wave_sum(s);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int idx = lane + 64 * v;
      if (idx < nvec) {
        const f32x4 bb = pb[idx];
        xl[v].x = fmaf(x0[v].x, s, bb.x) + xl[v].x;
        xl[v].y = fmaf(x0[v].y, s, bb.y) + xl[v].y;
        xl[v].z = fmaf(x0[v].z, s, bb.z) + xl[v].z;
        xl[v].w = fmaf(x0[v].w, s, bb.w) + xl[v].w;
      }
    }
  }
  f32x4* po = reinterpret_cast<f32x4*>(out + b * out_stride);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int idx = lane + 64 * v;
    if (idx < nvec) po[idx] = xl[v];
  }
}

// ------------------------------------------------------------------------------------------------
// K4, closed form.  x_{l+1} = x0 (x_l . w_l) + b_l + x_l unrolls to x_l = alpha_l x0 + beta_l with
//   beta_l = b_0 + ... + b_{l-1}  (row independent),   s_l = alpha_l (x0 . w_l) + (beta_l . w_l),
//   alpha_{l+1} = alpha_l + s_l,  alpha_0 = 1.
// So a row needs L INDEPENDENT dots of x0 (no serial dependence through x_l), a scalar recurrence and
// one AXPY: x0 is the only per-row tile in VGPRs (half the registers of the literal form, twice the
// occupancy), w_l streams from L2 once per row instead of w_l and b_l, beta_L sits in LDS.
// Same value in exact arithmetic; fp32 rounding differs at the 1e-7 level (parity bar 1e-5).
// ------------------------------------------------------------------------------------------------
constexpr int kCrossMaxL = 8;

template <int VPL>
__global__ __launch_bounds__(256) void cross_closed_kernel(const float* __restrict__ x, int64_t x_stride,
                                                           int dim, const float* __restrict__ w,
                                                           const float* __restrict__ bv, int L, int64_t B,
                                                           float* __restrict__ out, int64_t out_stride) {
  extern __shared__ __attribute__((aligned(16))) float beta[];  // beta_L, dim floats
  __shared__ float cdot[kCrossMaxL];                             // beta_l . w_l
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int nvec = dim >> 2;
  // ---- block prologue: beta_L and the L constants c_l = beta_l . w_l (wave l computes c_l)
  for (int c = tid; c < dim; c += 256) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += bv[(int64_t)l * dim + c];
    beta[c] = acc;
  }
  for (int l = wv; l < L; l += 4) {
    float acc = 0.f;
    for (int c = lane; c < dim; c += 64) {
      float bl = 0.f;
      for (int k2 = 0; k2 < l; ++k2) bl += bv[(int64_t)k2 * dim + c];
      acc = fmaf(bl, w[(int64_t)l * dim + c], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) cdot[l] = acc;
  }
  __syncthreads();
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wv; b < B; b += nwaves) {
    f32x4 x0[VPL];
    const f32x4* px = reinterpret_cast<const f32x4*>(x + b * x_stride);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int idx = lane + 64 * v;
      x0[v] = idx < nvec ? px[idx] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float alpha = 1.f;
#pragma unroll 1
    for (int l = 0; l < L; ++l) {  // not unrolled: one layer's w tile in flight at a time (VGPR budget)
      const f32x4* pw = reinterpret_cast<const f32x4*>(w + (int64_t)l * dim);
      float acc = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int idx = lane + 64 * v;
        if (idx < nvec) {
          const f32x4 wv4 = pw[idx];
          acc = fmaf(x0[v].x, wv4.x, acc);
          acc = fmaf(x0[v].y, wv4.y, acc);
          acc = fmaf(x0[v].z, wv4.z, acc);
          acc = fmaf(x0[v].w, wv4.w, acc);
        }
      }
      acc = wave_sum(acc);
      alpha += fmaf(alpha, acc, cdot[l]);
    }
    f32x4* po = reinterpret_cast<f32x4*>(out + b * out_stride);
    const f32x4* pb = reinterpret_cast<const f32x4*>(beta);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int idx = lane + 64 * v;
      if (idx < nvec) {
        const f32x4 bb = pb[idx];
        f32x4 r;
        r.x = fmaf(alpha, x0[v].x, bb.x);
        r.y = fmaf(alpha, x0[v].y, bb.y);
        r.z = fmaf(alpha, x0[v].z, bb.z);
        r.w = fmaf(alpha, x0[v].w, bb.w);
        po[idx] = r;
      }
    }
  }
}

// generic fallback: any dim / alignment; x_l kept in the output row (global), one wave per row
__global__ __launch_bounds__(256) void cross_generic_kernel(const float* __restrict__ x,
                                                            int64_t x_stride, int dim,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bv, int L,
                                                            int64_t B, float* __restrict__ out,
                                                            int64_t out_stride) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* x0 = x + b * x_stride;
  float* xl = out + b * out_stride;
  for (int c = lane; c < dim; c += 64) xl[c] = x0[c];
  for (int l = 0; l < L; ++l) {
    float s = 0.f;
    for (int c = lane; c < dim; c += 64) s = fmaf(xl[c], w[(int64_t)l * dim + c], s);
    s = wave_sum(s);
    for (int c = lane; c < dim; c += 64) xl[c] = fmaf(x0[c], s, bv[(int64_t)l * dim + c]) + xl[c];
  }
}

// ------------------------------------------------------------------------------------------------
// K2 — ctr FM in gather form: one wave per sample, lanes stride over the nd dense + F sparse
// "active columns" of the never-materialised one-hot stack.
// ------------------------------------------------------------------------------------------------
struct FmOffsets {
  int64_t off[REC_MAX_TABLES];  // first stack column of field f (nd + sum_{g<f} vocab_g)
  int32_t vocab[REC_MAX_TABLES];
};

__global__ __launch_bounds__(256) void fm_onehot_kernel(
    const float* __restrict__ dense, int64_t dense_stride, int nd, const int32_t* __restrict__ ids,
    int64_t ids_stride, int F, FmOffsets fo, int64_t Ltot, const float* __restrict__ w0,
    const float* __restrict__ w, const float* __restrict__ V, int k, int64_t B,
    float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  // each lane owns "active columns" a = lane, lane+64, ... of [dense 0..nd) ++ [fields 0..F)
  float lin = 0.f;
  float second = 0.f;
  const int nact = nd + F;
  // linear term
  for (int a = lane; a < nact; a += 64) {
    if (a < nd) {
      lin = fmaf(dense[b * dense_stride + a], w[a], lin);
    } else {
      const int f = a - nd;
      const int32_t id = ids[b * ids_stride + f];
      if ((uint32_t)id < (uint32_t)fo.vocab[f]) lin += w[fo.off[f] + id];
    }
  }
  lin = wave_sum(lin);
  for (int kk = 0; kk < k; ++kk) {
    const float* Vk = V + (int64_t)kk * Ltot;
    float s = 0.f, q = 0.f;
    for (int a = lane; a < nact; a += 64) {
      float xv = 0.f, vv = 0.f;
      if (a < nd) {
        xv = dense[b * dense_stride + a];
        vv = Vk[a];
      } else {
        const int f = a - nd;
        const int32_t id = ids[b * ids_stride + f];
        if ((uint32_t)id < (uint32_t)fo.vocab[f]) {
          xv = 1.f;
          vv = Vk[fo.off[f] + id];
        }
      }
      const float t = xv * vv;
      s += t;
      q = fmaf(t, t, q);  // x^2 v^2 == (x v)^2
    }
    s = wave_sum(s);
    q = wave_sum(q);
    second += s * s - q;
  }
  if (lane == 0) {
    const float z = w0[0] + lin + 0.5f * second;
    out[b] = 1.f / (1.f + __expf(-z));
  }
}

// ------------------------------------------------------------------------------------------------
// K9 — y = LN(x + r) * gamma + beta [* row_mask]; one wave per row, d <= 1024
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ r,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ row_mask,
                                                        int64_t rows, int d,
                                                        float* __restrict__ out) {
  constexpr int EPL = 16;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* px = x + row * d;
  const float* pr = r ? r + row * d : nullptr;
  float v[EPL];
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = lane + 64 * e;
    v[e] = c < d ? (px[c] + (pr ? pr[c] : 0.f)) : 0.f;
    s += v[e];
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = lane + 64 * e;
    const float t = c < d ? v[e] - mu : 0.f;
    q = fmaf(t, t, q);
  }
  const float var = wave_sum(q) / (float)d;
  const float inv = 1.f / sqrtf(var + eps);
  const float m = row_mask ? row_mask[row] : 1.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = lane + 64 * e;
    if (c < d) out[row * d + c] = ((v[e] - mu) * inv * gamma[c] + beta[c]) * m;
  }
}

// d % 4 == 0, d/4 a power of two <= 64: LPR = d/4 lanes own a row (16 B each), 64/LPR rows per wave
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ r,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            const float* __restrict__ row_mask, int64_t rows,
                                                            float* __restrict__ out) {
  constexpr int D = LPR * 4;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sl = lane % LPR, sub = lane / LPR;
  const int64_t row_raw = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const bool live = row_raw < rows;
  const int64_t row = live ? row_raw : rows - 1;
  f32x4 v = reinterpret_cast<const f32x4*>(x + row * D)[sl];
  if (r) v += reinterpret_cast<const f32x4*>(r + row * D)[sl];
  float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s / (float)D;
  const f32x4 c = v - mu;
  float q = c.x * c.x;
  q = fmaf(c.y, c.y, q);
  q = fmaf(c.z, c.z, q);
  q = fmaf(c.w, c.w, q);
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float inv = 1.f / sqrtf(q / (float)D + eps);
  const float m = row_mask ? row_mask[row] : 1.f;
  const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[sl];
  const f32x4 b = reinterpret_cast<const f32x4*>(beta)[sl];
  const f32x4 y = (c * inv * g + b) * m;
  if (live) reinterpret_cast<f32x4*>(out + row * D)[sl] = y;
}

// ------------------------------------------------------------------------------------------------
// K10 — out[b, j] = seq_info[b] . table[ids[b, j]]; LPR = d/4 lanes per looked-up row
// ------------------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void gather_dot_kernel(
    const float* __restrict__ seq, int64_t seq_stride, const float* __restrict__ table,
    int32_t vocab, const int32_t* __restrict__ ids, int64_t ids_stride, int n, int64_t B,
    float* __restrict__ out, int64_t out_stride, int* __restrict__ oob) {
  constexpr int RPW = 64 / LPR;  // (b, j) rows per wave pass
  constexpr int D = LPR * 4;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;
  const int sl = lane % LPR;
  const int64_t total = B * (int64_t)n;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t idx = wave * RPW + sub;
  const bool live = idx < total;
  const int64_t ii = live ? idx : total - 1;
  const int64_t b = ii / n;
  const int j = (int)(ii - b * n);
  const int32_t id = ids[b * ids_stride + j];
  const bool ok = (uint32_t)id < (uint32_t)vocab;
  if (!ok && oob && live) *oob = 1;
  const f32x4 t = *reinterpret_cast<const f32x4*>(table + (int64_t)(ok ? id : 0) * D + sl * 4);
  const f32x4 q = *reinterpret_cast<const f32x4*>(seq + b * seq_stride + sl * 4);
  float acc = t.x * q.x;
  acc = fmaf(t.y, q.y, acc);
  acc = fmaf(t.z, q.z, acc);
  acc = fmaf(t.w, q.w, acc);
  if (!ok) acc = 0.f;
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (live && sl == 0) out[b * out_stride + j] = acc;
}

__global__ __launch_bounds__(256) void gather_dot_generic_kernel(
    const float* __restrict__ seq, int64_t seq_stride, const float* __restrict__ table,
    int32_t vocab, int d, const int32_t* __restrict__ ids, int64_t ids_stride, int n, int64_t B,
    float* __restrict__ out, int64_t out_stride, int* __restrict__ oob) {
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= B * (int64_t)n) return;
  const int64_t b = idx / n;
  const int j = (int)(idx - b * n);
  const int32_t id = ids[b * ids_stride + j];
  const bool ok = (uint32_t)id < (uint32_t)vocab;
  if (!ok && oob && lane == 0) *oob = 1;
  float acc = 0.f;
  if (ok)
    for (int c = lane; c < d; c += 64) acc = fmaf(table[(int64_t)id * d + c], seq[b * seq_stride + c], acc);
  acc = wave_sum(acc);
  if (lane == 0) out[b * out_stride + j] = acc;
}

// K1 + NV dot products per sample, in one pass over the gathered rows: out_dots[b, v] = <concat row b, Wd[v, :]>.
// DCN's whole cross tower reduces to this (closed form: x_l = alpha_l x0 + sum_{j<l} b_j needs only d_l = x0 . w_l),
// and because the cross output only meets Dense(1) afterwards, cross_x is never materialised either:
//   logit = alpha_L (x0 . w_c) + (sum_j b_j) . w_c + dnn_x . w_d + bias      (rec_dcn_logit_f32).
// LPR lanes own a sample (16 B of every row each); the NV weight vectors live in LDS (NV * width * 4 B per workgroup).
template <int LPR, int IDS_F32, int NV>
__global__ __launch_bounds__(256) void gather_dots_kernel(TableSet ts, const void* __restrict__ ids, int64_t ids_stride,
                                                          int F, const float* __restrict__ Wd, int width, int64_t B,
                                                          float* __restrict__ emb_out, int64_t emb_stride,
                                                          float* __restrict__ out_dots, int* __restrict__ oob,
                                                          float* __restrict__ row_absmax) {
  constexpr int D = LPR * 4;
  constexpr int SPW = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float wsh[];  // [NV][width]
  for (int e = threadIdx.x * 4; e < NV * width; e += 256 * 4)
    *reinterpret_cast<f32x4*>(wsh + e) = *reinterpret_cast<const f32x4*>(Wd + e);  // width % 4 == 0, Wd 16-B aligned
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int sl = lane % LPR, sw = lane / LPR;
  const int64_t b_raw = ((int64_t)blockIdx.x * 4 + wv) * SPW + sw;
  const bool live = b_raw < B;
  const int64_t b = live ? b_raw : B - 1;
  float acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;
  float amax = 0.f;      // max |element| of the sample's concat row (the DNN's first Dense scales by it: f16x2 kernel)
  constexpr int U = 8;
  for (int f0 = 0; f0 < F; f0 += U) {
    f32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u < F ? f0 + u : F - 1;
      const int32_t id = load_id<IDS_F32>(ids, b * ids_stride + f);
      const bool ok = (uint32_t)id < (uint32_t)ts.vocab[f];
      if (!ok && oob && live) *oob = 1;
      const f32x4 t = *reinterpret_cast<const f32x4*>(ts.base[f] + (int64_t)(ok ? id : 0) * D + sl * 4);
      r[u] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = f0 + u;
      if (f < F) {
        const int oc = ts.out_col[f] + sl * 4;
        if (live) __builtin_nontemporal_store(r[u], reinterpret_cast<f32x4*>(emb_out + b * emb_stride + oc));
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(r[u].x), fabsf(r[u].y)), fmaxf(fabsf(r[u].z), fabsf(r[u].w))));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsh + v * width + oc);
          acc[v] = fmaf(r[u].x, w4.x, fmaf(r[u].y, w4.y, fmaf(r[u].z, w4.z, fmaf(r[u].w, w4.w, acc[v]))));
        }
      }
    }
  }
  if (row_absmax) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (live && sl == 0) row_absmax[b] = amax;
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float a = acc[v];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (live && sl == 0) out_dots[b * NV + v] = a;
  }
}

// per sample: alpha = 1; for l < L: alpha += alpha * d[l] + G[l];  out = sigmoid(alpha * d[L] + c + extra[b])
__global__ __launch_bounds__(256) void dcn_logit_kernel(const float* __restrict__ dots, int L, const float* __restrict__ G,
                                                        float c, const float* __restrict__ extra, int64_t B,
                                                        float* __restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* d = dots + b * (L + 1);
  float alpha = 1.f;
  for (int l = 0; l < L; ++l) alpha += alpha * d[l] + G[l];
  const float z = alpha * d[L] + c + (extra ? extra[b] : 0.f);
  out[b] = 1.f / (1.f + expf(-z));
}

// ---- tiny elementwise epilogues ----------------------------------------------------------------
__global__ __launch_bounds__(256) void add_sigmoid_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, int64_t n,
                                                          float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = 1.f / (1.f + expf(-(a[i] + (b ? b[i] : 0.f))));
}

// out = act(alpha * a + beta * b): residual add + ReLU of Residual_Units, the 0.5/0.5 blend of Wide&Deep
__global__ __launch_bounds__(256) void axpby_act_kernel(const float* __restrict__ a, float alpha,
                                                        const float* __restrict__ b, float beta, int64_t n, int act,
                                                        float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = act_apply(alpha * a[i] + beta * b[i], act, 0.f);
}

__global__ __launch_bounds__(256) void axpby_act_vec_kernel(const f32x4* __restrict__ a, float alpha,
                                                            const f32x4* __restrict__ b, float beta, int64_t n4,
                                                            int act, f32x4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = a[i] * alpha + b[i] * beta;
  f32x4 o;
  o.x = act_apply(v.x, act, 0.f);
  o.y = act_apply(v.y, act, 0.f);
  o.z = act_apply(v.z, act, 0.f);
  o.w = act_apply(v.w, act, 0.f);
  out[i] = o;
}

// out = act(a * b) elementwise (NCF's sigmoid(user_embed * item_embed), ESMM's pCTR * pCVR)
__global__ __launch_bounds__(256) void mul_act_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      int64_t n, int act, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = act_apply(a[i] * b[i], act, 0.f);
}

// cosine over whole flattened tensors: per-block fp64 partials of <a,b>, <a,a>, <b,b>, then a fixed-order finish
__global__ __launch_bounds__(256) void cosine_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             int64_t n, double* __restrict__ part) {
  __shared__ double sh[3][4];
  double ab = 0.0, aa = 0.0, bb = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double x = a[i], y = b[i];
    ab += x * y;
    aa += x * x;
    bb += y * y;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    ab += __shfl_xor(ab, off);
    aa += __shfl_xor(aa, off);
    bb += __shfl_xor(bb, off);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[0][w] = ab, sh[1][w] = aa, sh[2][w] = bb;
  __syncthreads();
  if (threadIdx.x < 3)
    part[(int64_t)blockIdx.x * 3 + threadIdx.x] =
        sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

__global__ void cosine_finish_kernel(const double* __restrict__ part, int nblk, int apply_sigmoid,
                                     float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double ab = 0.0, aa = 0.0, bb = 0.0;
  for (int i = 0; i < nblk; ++i) ab += part[3 * i], aa += part[3 * i + 1], bb += part[3 * i + 2];
  const double c = ab / (sqrt(aa) * sqrt(bb));
  out[0] = (float)(apply_sigmoid ? 1.0 / (1.0 + exp(-c)) : c);
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ sc, int64_t total,
                                                         int d, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) out[i] = x[i] * sc[i / d];
}

// d % 4 == 0 and 16-B aligned pointers: one float4 per thread
__global__ __launch_bounds__(256) void scale_rows_vec_kernel(const f32x4* __restrict__ x,
                                                             const float* __restrict__ sc, int64_t total4,
                                                             int d4, f32x4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total4) out[i] = x[i] * sc[i / d4];
}

__global__ __launch_bounds__(256) void dice_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ alpha,
                                                   const float* __restrict__ mean,
                                                   const float* __restrict__ var, float eps,
                                                   int64_t total, int d, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % d);
  const float v = x[i];
  const float xn = (v - (mean ? mean[c] : 0.f)) / sqrtf((var ? var[c] : 1.f) + eps);
  const float p = 1.f / (1.f + expf(-xn));
  out[i] = alpha[0] * (1.f - p) * v + p * v;
}

}  // namespace rec

using namespace rec;

extern "C" int rec_add_sigmoid_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
  const char* who = "rec_add_sigmoid_f32";
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  if (n == 0) return REC_OK;
  REC_CHECK_ARG(a && out, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(add_sigmoid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a, b, n, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_axpby_act_f32(const float* a, float alpha, const float* b, float beta, int64_t n, int32_t act,
                                 float* out, void* stream) {
  const char* who = "rec_axpby_act_f32";
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_TANH, REC_EINVAL, "%s: act %d (none/relu/sigmoid/tanh)", who, act);
  if (n == 0) return REC_OK;
  REC_CHECK_ARG(a && b && out, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if ((n & 3) == 0 && aligned16(a) && aligned16(b) && aligned16(out)) {
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(axpby_act_vec_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const f32x4*>(a), alpha, reinterpret_cast<const f32x4*>(b), beta, n4, act,
                       reinterpret_cast<f32x4*>(out));
  } else {
    hipLaunchKernelGGL(axpby_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, alpha, b, beta, n,
                       act, out);
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_mul_act_f32(const float* a, const float* b, int64_t n, int32_t act, float* out, void* stream) {
  const char* who = "rec_mul_act_f32";
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_TANH, REC_EINVAL, "%s: act %d (none/relu/sigmoid/tanh)", who, act);
  if (n == 0) return REC_OK;
  REC_CHECK_ARG(a && b && out, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(mul_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a, b, n, act, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_cosine_flat_workspace_bytes(int64_t n) {
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  return blocks * 3 * (int64_t)sizeof(double);
}

extern "C" int rec_cosine_flat_f32(const float* a, const float* b, int64_t n, int32_t apply_sigmoid, float* out,
                                   void* workspace, void* stream) {
  const char* who = "rec_cosine_flat_f32";
  REC_CHECK_ARG(n >= 1, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  REC_CHECK_ARG(a && b && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  const int blocks = (int)(rec_cosine_flat_workspace_bytes(n) / (3 * sizeof(double)));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cosine_partial_kernel, dim3(blocks), dim3(256), 0, st, a, b, n, static_cast<double*>(workspace));
  hipLaunchKernelGGL(cosine_finish_kernel, dim3(1), dim3(64), 0, st, static_cast<const double*>(workspace), blocks,
                     apply_sigmoid, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_scale_rows_f32(const float* x, const float* row_scale, int64_t rows, int32_t d,
                                  float* out, void* stream) {
  const char* who = "rec_scale_rows_f32";
  REC_CHECK_ARG(rows >= 0 && d >= 1, REC_ESHAPE, "%s: rows=%lld d=%d", who, (long long)rows, d);
  if (rows == 0) return REC_OK;
  REC_CHECK_ARG(x && row_scale && out, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t total = rows * d;
  if (d % 4 == 0 && aligned16(x) && aligned16(out)) {
    const int64_t total4 = total / 4;
    hipLaunchKernelGGL(scale_rows_vec_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const f32x4*>(x), row_scale,
                       total4, d / 4, reinterpret_cast<f32x4*>(out));
  } else {
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, row_scale, total, d, out);
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

// ---- concat helpers: tf.concat becomes "write at a column offset" also for tensors no kernel of ours produced ------
namespace rec {
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int64_t ss, int src_f32,
                                                     int64_t M, int64_t N, float* __restrict__ dst, int64_t ds) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  dst[m * ds + n] = src_f32 ? src[m * ss + n] : (float)reinterpret_cast<const int32_t*>(src)[m * ss + n];
}
// out[b, j * D + c] = x[b, j] * E[j, c]   (AutoInt's dense features as fields: value-scaled embedding rows)
__global__ __launch_bounds__(256) void scale_embed_kernel(const float* __restrict__ x, int64_t xs,
                                                          const float* __restrict__ E, int64_t B, int nd, int D,
                                                          float* __restrict__ out, int64_t os) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)nd * D;
  if (i >= B * per) return;
  const int64_t b = i / per;
  const int e = (int)(i - b * per);
  out[b * os + e] = x[b * xs + e / D] * E[e];
}
}  // namespace rec

extern "C" int rec_copy2d_f32(const void* src, int64_t src_stride, int32_t src_is_f32, int64_t M, int64_t N, float* dst,
                              int64_t dst_stride, void* stream) {
  const char* who = "rec_copy2d_f32";
  REC_CHECK_ARG(M >= 0 && N >= 0 && src_stride >= N && dst_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  if (M * N == 0) return REC_OK;
  REC_CHECK_ARG(src && dst, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(rec::copy2d_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), static_cast<const float*>(src), src_stride, src_is_f32, M, N,
                     dst, dst_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_scale_embed_f32(const float* x, int64_t x_stride, const float* E, int64_t B, int32_t nd, int32_t D,
                                   float* out, int64_t out_stride, void* stream) {
  const char* who = "rec_scale_embed_f32";
  REC_CHECK_ARG(B >= 0 && nd >= 1 && D >= 1 && x_stride >= nd && out_stride >= (int64_t)nd * D, REC_ESHAPE,
                "%s: bad shape", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(x && E && out, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t n = B * nd * D;
  hipLaunchKernelGGL(rec::scale_embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, x_stride, E, B, nd, D, out, out_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dice_f32(const float* x, const float* alpha, const float* mean, const float* var,
                            float eps, int64_t rows, int32_t d, float* out, void* stream) {
  const char* who = "rec_dice_f32";
  REC_CHECK_ARG(rows >= 0 && d >= 1, REC_ESHAPE, "%s: rows=%lld d=%d", who, (long long)rows, d);
  if (rows == 0) return REC_OK;
  REC_CHECK_ARG(x && alpha && out, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t total = rows * d;
  hipLaunchKernelGGL(dice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, alpha, mean, var, eps, total, d, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_fm_layer_workspace_floats(int64_t B) {
  // fp64 block partials: rec_fm_layer_f32 uses <= kFmMaxPartials, the fused rec_gather_fm_f32 one per
  // workgroup (>= 4 samples each), + alignment slack
  int64_t n = (B + 3) / 4;
  if (n < kFmMaxPartials) n = kFmMaxPartials;
  return 2 * n + 2;
}

extern "C" int rec_fm_layer_f32(const float* first, int64_t first_stride, int32_t L1, const float* w,
                                const float* second, int64_t second_stride, int32_t M, int64_t B,
                                float* out, float* workspace, void* stream) {
  const char* who = "rec_fm_layer_f32";
  REC_CHECK_ARG(B >= 0 && L1 >= 1 && M >= 1 && first_stride >= L1 && second_stride >= M, REC_ESHAPE,
                "%s: bad shape B=%lld L1=%d M=%d", who, (long long)B, L1, M);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(first && w && second && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // rows per wave chosen so that the number of block partials stays <= kFmMaxPartials
  int64_t rpw = (B + (int64_t)4 * kFmMaxPartials - 1) / ((int64_t)4 * kFmMaxPartials);
  if (rpw < 1) rpw = 1;
  const int64_t blocks = (B + 4 * rpw - 1) / (4 * rpw);
  double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 7) & ~(uintptr_t)7);
  hipLaunchKernelGGL(fm_layer_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, first,
                     first_stride, L1, w, second, second_stride, M, B, (int)rpw, out, partial);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(fm_layer_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st,
                     partial, (int)blocks, B, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

namespace rec { int fill_table_set(const rec_table_desc* tables, int32_t F, TableSet* ts, const char* who); }

extern "C" int rec_gather_fm_absmax_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                                        int64_t ids_stride, const float* dense, int64_t dense_stride, int32_t nd,
                                        const float* w, int64_t B, float* emb_out, int64_t emb_stride, float* fm_out,
                                        float* workspace, int32_t* oob_flag, float* row_absmax, void* stream);
extern "C" int rec_gather_fm_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                                 int64_t ids_stride, const float* dense, int64_t dense_stride, int32_t nd,
                                 const float* w, int64_t B, float* emb_out, int64_t emb_stride,
                                 float* fm_out, float* workspace, int32_t* oob_flag, void* stream) {
  return rec_gather_fm_absmax_f32(tables, F, ids, ids_dtype, ids_stride, dense, dense_stride, nd, w, B, emb_out, emb_stride,
                                  fm_out, workspace, oob_flag, nullptr, stream);
}

extern "C" int rec_gather_fm_absmax_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                                        int64_t ids_stride, const float* dense, int64_t dense_stride, int32_t nd,
                                        const float* w, int64_t B, float* emb_out, int64_t emb_stride, float* fm_out,
                                        float* workspace, int32_t* oob_flag, float* row_absmax, void* stream) {
  const char* who = "rec_gather_fm_f32";
  TableSet ts;
  int rc = fill_table_set(tables, F, &ts, who);
  if (rc != REC_OK) return rc;
  const int D = tables[0].dim;
  const int lpr = D / 4;
  REC_CHECK_ARG(D % 4 == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0, REC_ESHAPE,
                "%s: D=%d (need D/4 a power of two <= 64)", who, D);
  for (int f = 0; f < F; ++f)
    REC_CHECK_ARG(tables[f].dim == D && aligned16(tables[f].base) && tables[f].out_col % 4 == 0, REC_ESHAPE,
                  "%s: tables must share dim, be 16-B aligned, out_col %% 4 == 0", who);
  REC_CHECK_ARG(ids_dtype == REC_IDS_I32 || ids_dtype == REC_IDS_F32, REC_EINVAL, "%s: bad ids_dtype", who);
  REC_CHECK_ARG(B >= 0 && nd >= 0 && ids_stride >= F, REC_ESHAPE, "%s: bad shape", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(ids && w && emb_out && fm_out && workspace && (nd == 0 || dense), REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(aligned16(emb_out) && emb_stride % 4 == 0 && nd % 4 == 0 && aligned16(w), REC_EINVAL,
                "%s: emb_out must be 16-B aligned with stride %% 4 == 0; w 16-B aligned and nd %% 4 == 0 "
                "(pad the dense block of w / the concat buffer)", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int spw = 64 / lpr;
  const int64_t waves = (B + spw - 1) / spw;
  const int64_t blocks = (waves + 3) / 4;
  REC_CHECK_ARG(blocks <= 0x7fffffffLL, REC_ESHAPE, "%s: batch too large", who);
  double* partial = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 7) & ~(uintptr_t)7);
#define REC_GFM(L_)                                                                                        \
  case L_:                                                                                                 \
    if (ids_dtype == REC_IDS_F32)                                                                          \
      hipLaunchKernelGGL((gather_fm_kernel<L_, 1>), dim3((unsigned)blocks), dim3(256), 0, st, ts, ids, ids_stride, \
                         F, dense, dense_stride, nd, w, B, emb_out, emb_stride, fm_out, partial, oob_flag, row_absmax); \
    else                                                                                                   \
      hipLaunchKernelGGL((gather_fm_kernel<L_, 0>), dim3((unsigned)blocks), dim3(256), 0, st, ts, ids, ids_stride, \
                         F, dense, dense_stride, nd, w, B, emb_out, emb_stride, fm_out, partial, oob_flag, row_absmax); \
    break;
  switch (lpr) { REC_GFM(1) REC_GFM(2) REC_GFM(4) REC_GFM(8) REC_GFM(16) REC_GFM(32) REC_GFM(64) }
#undef REC_GFM
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(fm_layer_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, partial,
                     (int)blocks, B, fm_out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_cross_f32(const float* x, int64_t x_stride, int32_t dim, const float* w,
                             const float* b, int32_t L, int64_t B, float* out, int64_t out_stride,
                             void* stream) {
  const char* who = "rec_cross_f32";
  REC_CHECK_ARG(B >= 0 && dim >= 1 && L >= 0 && x_stride >= dim && out_stride >= dim, REC_ESHAPE,
                "%s: bad shape B=%lld dim=%d L=%d", who, (long long)B, dim, L);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(x && out && (L == 0 || (w && b)), REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  const bool vec = dim % 4 == 0 && aligned16(x) && aligned16(out) && aligned16(w) && aligned16(b) &&
                   x_stride % 4 == 0 && out_stride % 4 == 0 && dim <= 4096;
  if (vec) {
    const int vpl = (dim / 4 + 63) / 64;
    const bool literal = forced("cross") && forced("cross")[0] == 'l';  // rec_debug_force: tests / A/B only
    if (!literal && L >= 1 && L <= kCrossMaxL) {
      int64_t blocks = (B + 3) / 4;
      int bpc = 4;  // 102 VGPRs -> 4 waves/SIMD = 4 workgroups per CU resident (measured 4/6/8: 0.466/0.523/0.513 ms)
      if (const char* e = forced("cross_bpc")) bpc = atoi(e) > 0 ? atoi(e) : bpc;
      if (blocks > 256 * bpc) blocks = 256 * bpc;  // persistent: the prologue runs once per workgroup
      const size_t lds = (size_t)dim * sizeof(float);
#define REC_CROSS_C(V)                                                                                \
  if (vpl <= V) {                                                                                     \
    hipLaunchKernelGGL((cross_closed_kernel<V>), dim3((unsigned)blocks), block, lds, st, x, x_stride, dim, w, b, \
                       L, B, out, out_stride);                                                        \
    REC_CHECK_LAUNCH(who);                                                                            \
    return REC_OK;                                                                                    \
  }
      REC_CROSS_C(1) REC_CROSS_C(2) REC_CROSS_C(4) REC_CROSS_C(8) REC_CROSS_C(13) REC_CROSS_C(16)
#undef REC_CROSS_C
    }
#define REC_CROSS(V)                                                                            \
  if (vpl <= V) {                                                                               \
    hipLaunchKernelGGL((cross_kernel<V>), grid, block, 0, st, x, x_stride, dim, w, b, L, B, out, \
                       out_stride);                                                             \
    REC_CHECK_LAUNCH(who);                                                                      \
    return REC_OK;                                                                              \
  }
    REC_CROSS(1) REC_CROSS(2) REC_CROSS(4) REC_CROSS(8) REC_CROSS(13) REC_CROSS(16)
#undef REC_CROSS
  }
  REC_CHECK_ARG(x != out, REC_EINVAL, "%s: generic path cannot run in place", who);
  hipLaunchKernelGGL(cross_generic_kernel, grid, block, 0, st, x, x_stride, dim, w, b, L, B, out,
                     out_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_fm_onehot_f32(const float* dense, int64_t dense_stride, int32_t nd,
                                 const int32_t* ids, int64_t ids_stride, int32_t F,
                                 const int64_t* vocab, const float* w0, const float* w,
                                 const float* V, int32_t k, int64_t B, float* out, void* stream) {
  const char* who = "rec_fm_onehot_f32";
  REC_CHECK_ARG((nd == 0 || dense) && (F == 0 || (ids && vocab)) && w0 && w && V && out, REC_EINVAL,
                "%s: NULL pointer", who);
  REC_CHECK_ARG(B >= 0 && nd >= 0 && F >= 0 && F <= REC_MAX_TABLES && k >= 0, REC_ESHAPE,
                "%s: bad shape B=%lld nd=%d F=%d k=%d", who, (long long)B, nd, F, k);
  if (B == 0) return REC_OK;
  FmOffsets fo;
  int64_t off = nd;
  for (int f = 0; f < REC_MAX_TABLES; ++f) {
    fo.off[f] = off;
    fo.vocab[f] = 0;
    if (f < F) {
      REC_CHECK_ARG(vocab[f] >= 1 && vocab[f] <= 0x7fffffffLL, REC_ESHAPE, "%s: vocab[%d]=%lld", who,
                    f, (long long)vocab[f]);
      fo.vocab[f] = (int32_t)vocab[f];
      off += vocab[f];
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(fm_onehot_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, dense,
                     dense_stride, nd, ids, ids_stride, F, fo, off, w0, w, V, k, B, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_layernorm_residual_f32(const float* x, const float* r, const float* gamma,
                                          const float* beta, float eps, const float* row_mask,
                                          int64_t rows, int32_t d, float* out, void* stream) {
  const char* who = "rec_layernorm_residual_f32";
  REC_CHECK_ARG(x && gamma && beta && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(rows >= 0 && d >= 1 && d <= 1024, REC_ESHAPE, "%s: rows=%lld d=%d (d <= 1024)", who,
                (long long)rows, d);
  if (rows == 0) return REC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int lpr = d / 4;
  if (d % 4 == 0 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && aligned16(x) && aligned16(out) && aligned16(gamma) &&
      aligned16(beta) && (!r || aligned16(r))) {
#define REC_LN(L_)                                                                                         \
  case L_: {                                                                                               \
    const int64_t waves = (rows + (64 / L_) - 1) / (64 / L_);                                              \
    hipLaunchKernelGGL((layernorm_vec_kernel<L_>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, x, r, \
                       gamma, beta, eps, row_mask, rows, out);                                             \
    break;                                                                                                 \
  }
    switch (lpr) { REC_LN(1) REC_LN(2) REC_LN(4) REC_LN(8) REC_LN(16) REC_LN(32) REC_LN(64) }
#undef REC_LN
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, r,
                     gamma, beta, eps, row_mask, rows, d, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_dot_scores_f32(const float* seq_info, int64_t seq_stride,
                                         const rec_table_desc* table, const int32_t* ids,
                                         int64_t ids_stride, int32_t n, int64_t B, float* out,
                                         int64_t out_stride, int32_t* oob_flag, void* stream) {
  const char* who = "rec_gather_dot_scores_f32";
  REC_CHECK_ARG(seq_info && table && table->base && ids && out, REC_EINVAL, "%s: NULL pointer", who);
  const int d = table->dim;
  REC_CHECK_ARG(B >= 0 && n >= 1 && d >= 1 && seq_stride >= d && ids_stride >= n && out_stride >= n &&
                    table->vocab >= 1 && table->vocab <= 0x7fffffffLL,
                REC_ESHAPE, "%s: bad shape", who);
  if (B == 0) return REC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t total = B * (int64_t)n;
  const int lpr = d / 4;
  const bool vec = d % 4 == 0 && (lpr & (lpr - 1)) == 0 && lpr <= 64 && aligned16(table->base) &&
                   aligned16(seq_info) && seq_stride % 4 == 0;
  if (vec) {
#define REC_GD(L_)                                                                               \
  case L_: {                                                                                     \
    const int64_t waves = (total + (64 / L_) - 1) / (64 / L_);                                   \
    hipLaunchKernelGGL((gather_dot_kernel<L_>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, \
                       st, seq_info, seq_stride, table->base, (int32_t)table->vocab, ids,        \
                       ids_stride, n, B, out, out_stride, oob_flag);                             \
    break;                                                                                       \
  }
    switch (lpr) {
      REC_GD(1) REC_GD(2) REC_GD(4) REC_GD(8) REC_GD(16) REC_GD(32) REC_GD(64)
    }
#undef REC_GD
  } else {
    hipLaunchKernelGGL(gather_dot_generic_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st,
                       seq_info, seq_stride, table->base, (int32_t)table->vocab, d, ids, ids_stride,
                       n, B, out, out_stride, oob_flag);
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_dots_absmax_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                                          int64_t ids_stride, const float* Wd, int32_t nv, int32_t width, int64_t B,
                                          float* emb_out, int64_t emb_stride, float* out_dots, int32_t* oob_flag,
                                          float* row_absmax, void* stream);
extern "C" int rec_gather_dots_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                                   int64_t ids_stride, const float* Wd, int32_t nv, int32_t width, int64_t B,
                                   float* emb_out, int64_t emb_stride, float* out_dots, int32_t* oob_flag,
                                   void* stream) {
  return rec_gather_dots_absmax_f32(tables, F, ids, ids_dtype, ids_stride, Wd, nv, width, B, emb_out, emb_stride, out_dots,
                                    oob_flag, nullptr, stream);
}

extern "C" int rec_gather_dots_absmax_f32(const rec_table_desc* tables, int32_t F, const void* ids, int32_t ids_dtype,
                                          int64_t ids_stride, const float* Wd, int32_t nv, int32_t width, int64_t B,
                                          float* emb_out, int64_t emb_stride, float* out_dots, int32_t* oob_flag,
                                          float* row_absmax, void* stream) {
  const char* who = "rec_gather_dots_f32";
  TableSet ts;
  int rc = fill_table_set(tables, F, &ts, who);
  if (rc != REC_OK) return rc;
  const int D = tables[0].dim;
  const int lpr = D / 4;
  REC_CHECK_ARG(D % 4 == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0, REC_ESHAPE,
                "%s: D=%d (need D/4 a power of two <= 64)", who, D);
  int maxcol = 0;
  for (int f = 0; f < F; ++f) {
    REC_CHECK_ARG(tables[f].dim == D && aligned16(tables[f].base) && tables[f].out_col % 4 == 0, REC_ESHAPE,
                  "%s: tables must share dim, be 16-B aligned, out_col %% 4 == 0", who);
    if (tables[f].out_col + D > maxcol) maxcol = tables[f].out_col + D;
  }
  REC_CHECK_ARG(ids_dtype == REC_IDS_I32 || ids_dtype == REC_IDS_F32, REC_EINVAL, "%s: bad ids_dtype", who);
  REC_CHECK_ARG(nv >= 1 && nv <= 8 && width >= maxcol && width % 4 == 0 && B >= 0 && ids_stride >= F, REC_ESHAPE,
                "%s: bad shape nv=%d (1..8) width=%d", who, nv, width);
  const size_t lds = (size_t)nv * width * sizeof(float);
  REC_CHECK_ARG(lds <= 64 * 1024, REC_ESHAPE, "%s: nv * width * 4 = %zu B of LDS (> 64 KiB)", who, lds);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(ids && Wd && emb_out && out_dots, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(aligned16(emb_out) && emb_stride % 4 == 0 && aligned16(Wd), REC_EINVAL,
                "%s: emb_out / Wd must be 16-B aligned, emb_stride %% 4 == 0", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int spw = 64 / lpr;
  const int64_t waves = (B + spw - 1) / spw;
  const int64_t blocks = (waves + 3) / 4;
  REC_CHECK_ARG(blocks <= 0x7fffffffLL, REC_ESHAPE, "%s: batch too large", who);
#define REC_GD3(L_, I_, V_)                                                                                       \
  hipLaunchKernelGGL((gather_dots_kernel<L_, I_, V_>), dim3((unsigned)blocks), dim3(256), lds, st, ts, ids, ids_stride, \
                     F, Wd, width, B, emb_out, emb_stride, out_dots, oob_flag, row_absmax)
#define REC_GD2(L_, I_)                                                            \
  switch (nv) {                                                                    \
    case 1: REC_GD3(L_, I_, 1); break;                                             \
    case 2: REC_GD3(L_, I_, 2); break;                                             \
    case 3: REC_GD3(L_, I_, 3); break;                                             \
    case 4: REC_GD3(L_, I_, 4); break;                                             \
    case 5: REC_GD3(L_, I_, 5); break;                                             \
    case 6: REC_GD3(L_, I_, 6); break;                                             \
    case 7: REC_GD3(L_, I_, 7); break;                                             \
    default: REC_GD3(L_, I_, 8); break;                                            \
  }
#define REC_GD1(L_)                                                   \
  case L_:                                                            \
    if (ids_dtype == REC_IDS_F32) { REC_GD2(L_, 1) } else { REC_GD2(L_, 0) } \
    break;
  switch (lpr) {
    REC_GD1(1) REC_GD1(2) REC_GD1(4) REC_GD1(8) REC_GD1(16) REC_GD1(32) REC_GD1(64)
    default: break;
  }
#undef REC_GD1
#undef REC_GD2
#undef REC_GD3
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dcn_logit_f32(const float* dots, int32_t L, const float* G, float c, const float* extra, int64_t B,
                                 float* out, void* stream) {
  const char* who = "rec_dcn_logit_f32";
  REC_CHECK_ARG(B >= 0 && L >= 0 && L <= 7, REC_ESHAPE, "%s: B=%lld L=%d", who, (long long)B, L);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(dots && out && (L == 0 || G), REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(dcn_logit_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), dots, L, G, c, extra, B, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}
