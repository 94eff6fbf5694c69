// T2 — backward kernels of the path's layers (the gradients Keras' fit() takes through the models of
// src/ctr/{dlrm,deep_fm,dcn}/model.py).  All HBM-bound row / column passes; the GEMM-shaped parts of a Dense backward
// (dX = dY W^T, dW = X^T dY) reuse rec_dense_f32 on operands transposed by rec_transpose_f32.
//
//   rec_transpose_f32            (M,N) -> (N,M) through a padded LDS tile
//   rec_act_grad_f32             dy *= act'(y)  for relu / sigmoid / tanh                (Dense(activation=...))
//   rec_colsum_f32               out[n] = sum_m w[m] a[m,n] b[m,n]   deterministic two-pass (bias / BN / cross grads)
//   rec_bn_train_f32 / _grad     BatchNormalization(training=True): batch statistics, moving-average update, backward
//                                (src/ctr/layers/modules.py:131 — a fresh BatchNormalization() in front of every DNN)
//   rec_bce_sigmoid_grad_f32     d mean(BCE(y, sigmoid(z))) / dz for the Keras probability form of the loss
//   rec_gather_pairwise_dot_grad_f32   backward of the fused gather + pairwise dot: table rows are re-gathered, dX = G X
//                                with G the symmetric matrix of the incoming pair gradients, embedding rows receive
//                                their gradient by 256-B fp32 atomics (IndexedSlices of tf.gather, densified)
//   rec_fm_layer_grad_f32        backward of the FM layer incl. its batch-global first-order scalar (modules.py:65)
//   rec_cross_layer_grad_f32     one layer of the DCN cross recurrence, backward (modules.py:105-112)
//   rec_adam_rows_f32            "lazy" Adam: only the rows a batch touched (documented deviation from the reference's
//                                dense update; see include/recamd.h)
#include <math.h>

#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- transpose --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, int64_t M, int64_t N, int64_t xs,
                                                        float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int64_t m = m0 + r, n = n0 + tx;
    tile[r][tx] = (m < M && n < N) ? x[m * xs + n] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int64_t n = n0 + r, m = m0 + tx;
    if (n < N && m < M) out[n * M + m] = tile[tx][r];
  }
}

// ---- activation derivative from the OUTPUT -----------------------------------------------------------------
__global__ __launch_bounds__(256) void act_grad_kernel(float* __restrict__ dy, int64_t dys, const float* __restrict__ y,
                                                       int64_t ys, int64_t M, int64_t N, int act) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float v = y[m * ys + n];
  float g = dy[m * dys + n];
  if (act == REC_ACT_RELU) g = v > 0.f ? g : 0.f;
  else if (act == REC_ACT_SIGMOID) g *= v * (1.f - v);
  else if (act == REC_ACT_TANH) g *= 1.f - v * v;
  dy[m * dys + n] = g;
}

// ---- deterministic column sums -------------------------------------------------------------------------------
// pass 1: block (chunk of 256 rows, 64 columns) -> partial[chunk][n] (fp32 sums of <= 256 terms, fixed order);
// pass 2: out[n] = sum over chunks in order, accumulated in fp64.
constexpr int kColRows = 256;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ a, int64_t as,
                                                             const float* __restrict__ b, int64_t bs,
                                                             const float* __restrict__ rw, int64_t M, int64_t N,
                                                             float* __restrict__ part) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  const int64_t m0 = (int64_t)blockIdx.y * kColRows;
  float acc = 0.f;
  if (n < N) {
    for (int r = ty; r < kColRows; r += 4) {  // each thread walks its rows in increasing order
      const int64_t m = m0 + r;
      if (m >= M) break;
      float v = a[m * as + n];
      if (b) v *= b[m * bs + n];
      if (rw) v *= rw[m];
      acc += v;
    }
  }
  sh[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && n < N) part[(int64_t)blockIdx.y * N + n] = (sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx]);
}
// 64 columns per workgroup; the four waves take the chunks c = 0, 1, 2, 3 (mod 4) — each sums its chunks in increasing
// order in fp64 — and the four sums are combined in a fixed order: deterministic, and four times shorter than one thread
// walking all chunks of its column (M = 1e5 rows = 400 chunks: 62-95 us per call in the attention models' steps).
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int64_t chunks, int64_t N,
                                                            float* __restrict__ out, float scale) {
  __shared__ double sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  double s = 0.0;
  if (n < N)
    for (int64_t c = ty; c < chunks; c += 4) s += (double)part[c * N + n];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) out[n] = (float)(((sh[0][tx] + sh[1][tx]) + (sh[2][tx] + sh[3][tx])) * (double)scale);
}

// ---- BatchNormalization, training mode ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_finish_kernel(const float* __restrict__ mean_in,
                                                              const float* __restrict__ sq_in, int64_t N, float eps,
                                                              float momentum, float* __restrict__ moving_mean,
                                                              float* __restrict__ moving_var,
                                                              float* __restrict__ save_mean,
                                                              float* __restrict__ save_inv) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float mu = mean_in[n];
  float var = sq_in[n] - mu * mu;  // biased batch variance (tf.nn.moments)
  var = var > 0.f ? var : 0.f;
  save_mean[n] = mu;
  save_inv[n] = 1.f / sqrtf(var + eps);
  if (moving_mean) moving_mean[n] = moving_mean[n] * momentum + mu * (1.f - momentum);
  if (moving_var) moving_var[n] = moving_var[n] * momentum + var * (1.f - momentum);
}
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t xs, int64_t M, int64_t N,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ inv,
                                                       float* __restrict__ y, int64_t ys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float xh = (x[m * xs + n] - mean[n]) * inv[n];
  y[m * ys + n] = xh * (gamma ? gamma[n] : 1.f) + (beta ? beta[n] : 0.f);
}
// dx = gamma inv (dy - mean_m(dy) - xhat mean_m(dy xhat))
__global__ __launch_bounds__(256) void bn_grad_kernel(const float* __restrict__ x, int64_t xs,
                                                      const float* __restrict__ dy, int64_t dys, int64_t M, int64_t N,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ inv, const float* __restrict__ s_dy,
                                                      const float* __restrict__ s_dyxh, float* __restrict__ dx,
                                                      int64_t dxs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float xh = (x[m * xs + n] - mean[n]) * inv[n];
  const float g = gamma ? gamma[n] : 1.f;
  dx[m * dxs + n] = g * inv[n] * (dy[m * dys + n] - s_dy[n] - xh * s_dyxh[n]);  // s_* are already means over the batch
}
// xhat written out (dgamma = colsum(dy * xhat))
__global__ __launch_bounds__(256) void bn_xhat_kernel(const float* __restrict__ x, int64_t xs, int64_t M, int64_t N,
                                                      const float* __restrict__ mean, const float* __restrict__ inv,
                                                      float* __restrict__ xh) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  xh[i] = (x[m * xs + n] - mean[n]) * inv[n];
}

// ---- loss gradient ----------------------------------------------------------------------------------------
// L = mean_i -(y log(pc + e) + (1-y) log(1 - pc + e)), pc = clip(p, e, 1-e), p = sigmoid(z)  (metrics.hip)
// dL/dz_i = scale * [ -(y/(pc+e)) + (1-y)/(1-pc+e) ] * [e < p < 1-e] * p (1-p)
__global__ __launch_bounds__(256) void bce_sigmoid_grad_kernel(const float* __restrict__ y, const float* __restrict__ p,
                                                               int64_t n, float scale, float* __restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float e = 1e-7f, pi = p[i], yi = y[i];
  const float pc = fminf(fmaxf(pi, e), 1.f - e);
  const float inside = (pi > e && pi < 1.f - e) ? 1.f : 0.f;
  dz[i] = scale * inside * (-(yi / (pc + e)) + (1.f - yi) / (1.f - pc + e)) * pi * (1.f - pi);
}

// ---- fused gather + pairwise dot, backward -----------------------------------------------------------------
// One wave per sample.  Lane l owns columns {l, l+64, ...} of every row (CPL = D/64 columns per lane); the N rows are
// re-gathered (coalesced 256-B segments), the pair gradients of the sample are wave-uniform (scalar loads), and
//   dX_i = sum_{j != i} g(i,j) X_j,   g(i,j) = dz[pair(max(i,j), min(i,j))]
// Table rows get their gradient by atomics (each wave-instruction covers 256 contiguous bytes of one row: the
// full-rate shape, MI355X_MICROARCH.md § Global float atomics); the dense row's gradient is stored.
template <int N, int CPL>
__global__ __launch_bounds__(256) void pairdot_grad_kernel(TableSet ts, TableSet gs, const int32_t* __restrict__ ids,
                                                           int64_t ids_stride, const float* __restrict__ dense,
                                                           int64_t dense_stride, int has_dense, int64_t B,
                                                           const float* __restrict__ dz, int64_t dz_stride,
                                                           int append_dense, float* __restrict__ d_dense,
                                                           int64_t dd_stride) {
  constexpr int D = 64 * CPL;
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (b >= B) return;
  const int F = has_dense ? N - 1 : N;
  float x[N][CPL];
  float* gdst[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float* src = nullptr;
    gdst[i] = nullptr;
    if (i < F) {
      const int32_t id = ids[b * ids_stride + i];
      if ((uint32_t)id < (uint32_t)ts.vocab[i]) {
        src = ts.base[i] + (int64_t)id * D;
        gdst[i] = const_cast<float*>(gs.base[i]) + (int64_t)id * D;
      }
    } else {
      src = dense + b * dense_stride;
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) x[i][c] = src ? src[lane + 64 * c] : 0.f;
  }
  const float* g = dz + b * dz_stride;  // wave-uniform address: scalar loads
  constexpr int P = N * (N - 1) / 2;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      const int hi = i > j ? i : j, lo = i > j ? j : i;
      const float gij = g[hi * (hi - 1) / 2 + lo];
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc[c] = fmaf(gij, x[j][c], acc[c]);
    }
    if (i < F) {
      if (gdst[i]) {  // wave-uniform
#pragma unroll
        for (int c = 0; c < CPL; ++c) atomicAdd(gdst[i] + lane + 64 * c, acc[c]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        float v = acc[c];
        if (append_dense) v += g[P + lane + 64 * c];  // the pass-through columns
        d_dense[b * dd_stride + lane + 64 * c] = v;
      }
    }
  }
}

// generic n (<= 64 rows) and D: rows re-read from global memory per pair (L1/L2 hits); correctness fallback
__global__ __launch_bounds__(256) void pairdot_grad_generic_kernel(TableSet ts, TableSet gs, int F,
                                                                   const int32_t* __restrict__ ids, int64_t ids_stride,
                                                                   const float* __restrict__ dense, int64_t dense_stride,
                                                                   int64_t B, int D, const float* __restrict__ dz,
                                                                   int64_t dz_stride, int append_dense,
                                                                   float* __restrict__ d_dense, int64_t dd_stride) {
  __shared__ const float* src_all[4][64];
  __shared__ float* dst_all[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;
  const int n = F + (dense ? 1 : 0);
  if (lane < n) {
    const float* s = nullptr;
    float* d = nullptr;
    if (lane < F) {
      const int32_t id = ids[b * ids_stride + lane];
      if ((uint32_t)id < (uint32_t)ts.vocab[lane]) {
        s = ts.base[lane] + (int64_t)id * D;
        d = const_cast<float*>(gs.base[lane]) + (int64_t)id * D;
      }
    } else {
      s = dense + b * dense_stride;
    }
    src_all[w][lane] = s;
    dst_all[w][lane] = d;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* g = dz + b * dz_stride;
  const int P = n * (n - 1) / 2;
  for (int i = 0; i < n; ++i) {
    if (i < F && !dst_all[w][i]) continue;  // wave-uniform: out-of-range id, nothing to update
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const float* xj = src_all[w][j];
        if (!xj) continue;
        const int hi = i > j ? i : j, lo = i > j ? j : i;
        acc = fmaf(g[hi * (hi - 1) / 2 + lo], xj[c], acc);
      }
      if (i < F) atomicAdd(dst_all[w][i] + c, acc);
      else d_dense[b * dd_stride + c] = acc + (append_dense ? g[P + c] : 0.f);
    }
  }
}

// ---- FM layer, backward --------------------------------------------------------------------------------------
// out[b] = S + 0.5 ((sum_j x_bj)^2 - sum_j x_bj^2),  S = sum_{b',l} first[b',l] w[l]   (one scalar for the batch)
// d second[b,j] = dout[b] (sum_j' x_bj' - x_bj);  d first[b',l] = w[l] T;  d w[l] = T sum_b' first[b',l];  T = sum_b dout[b]
__global__ __launch_bounds__(256) void fm_second_grad_kernel(const float* __restrict__ second, int64_t ss, int64_t M,
                                                             const float* __restrict__ dout, int64_t B,
                                                             float* __restrict__ d_second, int64_t dss) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  double s = 0.0;
  for (int64_t j = lane; j < M; j += 64) s += (double)second[b * ss + j];
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  const float S = (float)s, g = dout[b];
  for (int64_t j = lane; j < M; j += 64) d_second[b * dss + j] = g * (S - second[b * ss + j]);
}
__global__ __launch_bounds__(256) void fm_first_grad_kernel(const float* __restrict__ w, const float* __restrict__ colsum_first,
                                                            const float* __restrict__ T, int64_t L1, int64_t B,
                                                            float* __restrict__ d_first, int64_t dfs,
                                                            float* __restrict__ dw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float t = T[0];
  if (i < L1 && dw) dw[i] = t * colsum_first[i];
  if (d_first && i < B * L1) {
    const int64_t b = i / L1, l = i - b * L1;
    d_first[b * dfs + l] = t * w[l];
  }
}

// ---- one cross layer, backward ------------------------------------------------------------------------------
// forward: x_{l+1} = x0 s + b_l + x_l,  s = x_l . w_l  (per row).  Given g = dL/dx_{l+1} (dim floats per row):
//   ds = g . x0;   dx0 += g s;   g <- g + ds w_l  (= dL/dx_l);   ds[b] is kept for  dw_l = sum_b ds[b] x_l[b,:]
__global__ __launch_bounds__(256) void cross_grad_kernel(const float* __restrict__ x0, const float* __restrict__ xl,
                                                         const float* __restrict__ w, int64_t dim, int64_t B,
                                                         float* __restrict__ g, float* __restrict__ dx0,
                                                         float* __restrict__ ds_out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* x0r = x0 + b * dim;
  const float* xlr = xl + b * dim;
  float* gr = g + b * dim;
  float* dx0r = dx0 + b * dim;
  double s = 0.0, ds = 0.0;
  for (int64_t c = lane; c < dim; c += 64) {
    s += (double)xlr[c] * (double)w[c];
    ds += (double)gr[c] * (double)x0r[c];
  }
  for (int off = 32; off >= 1; off >>= 1) {
    s += __shfl_xor(s, off);
    ds += __shfl_xor(ds, off);
  }
  const float sf = (float)s, dsf = (float)ds;
  for (int64_t c = lane; c < dim; c += 64) {
    const float gc = gr[c];
    dx0r[c] += gc * sf;
    gr[c] = gc + dsf * w[c];
  }
  if (lane == 0) ds_out[b] = dsf;
}

// ---- lazy (row-wise) Adam ------------------------------------------------------------------------------------
// one wave per looked-up (b, f); the first wave to stamp a row with this step's number updates it (and clears its
// gradient), the others skip: every touched row is updated exactly once.  Rows no lookup touched keep var, m, v.
__global__ __launch_bounds__(256) void adam_rows_kernel(TableSet var, TableSet mt, TableSet vt, TableSet gt, TableSet stamp,
                                                        const int32_t* __restrict__ ids, int64_t ids_stride, int F,
                                                        int64_t R, int32_t step, float lr_t, float b1, float b2,
                                                        float eps, float l2x2) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int64_t b = r / F;
  const int f = (int)(r - b * F);
  const int32_t id = ids[b * ids_stride + f];
  if ((uint32_t)id >= (uint32_t)var.vocab[f]) return;
  int32_t* st = reinterpret_cast<int32_t*>(const_cast<float*>(stamp.base[f])) + id;
  int32_t old = 0;
  if (lane == 0) old = atomicExch(st, step);
  old = __shfl(old, 0);
  if (old == step) return;  // another lookup of this step already owns the row
  const int dim = var.dim[f];
  float* w = const_cast<float*>(var.base[f]) + (int64_t)id * dim;
  float* m = const_cast<float*>(mt.base[f]) + (int64_t)id * dim;
  float* v = const_cast<float*>(vt.base[f]) + (int64_t)id * dim;
  float* g = const_cast<float*>(gt.base[f]) + (int64_t)id * dim;
  for (int c = lane; c < dim; c += 64) {
    const float wc = w[c];
    const float gc = g[c] + wc * l2x2;
    const float mm = m[c] * b1 + gc * (1.f - b1);
    const float vv = v[c] * b2 + gc * gc * (1.f - b2);
    m[c] = mm;
    v[c] = vv;
    w[c] = wc - lr_t * mm / (sqrtf(vv) + eps);
    g[c] = 0.f;
  }
}

int fill_table_set(const rec_table_desc* tables, int32_t F, TableSet* ts, const char* who);

}  // namespace rec

using namespace rec;

extern "C" int rec_transpose_f32(const float* x, int64_t M, int64_t N, int64_t x_stride, float* out, void* stream) {
  const char* who = "rec_transpose_f32";
  REC_CHECK_ARG(M >= 0 && N >= 0 && x_stride >= N, REC_ESHAPE, "%s: M=%lld N=%lld", who, (long long)M, (long long)N);
  if (M == 0 || N == 0) return REC_OK;
  REC_CHECK_ARG(x && out, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t gx = (N + 63) / 64, gy = (M + 63) / 64;
  REC_CHECK_ARG(gy <= 65535 * 1024LL, REC_ESHAPE, "%s: too many rows", who);
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, x, M, N,
                     x_stride, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_act_grad_f32(float* dy, int64_t dy_stride, const float* y, int64_t y_stride, int64_t M, int64_t N,
                                int32_t act, void* stream) {
  const char* who = "rec_act_grad_f32";
  REC_CHECK_ARG(act == REC_ACT_NONE || act == REC_ACT_RELU || act == REC_ACT_SIGMOID || act == REC_ACT_TANH,
                REC_ENOTIMPL, "%s: activation %d has no backward here (PReLU's alpha gradient is not built)", who, act);
  if (act == REC_ACT_NONE || M * N == 0) return REC_OK;
  REC_CHECK_ARG(dy && y, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(act_grad_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                     dy_stride, y, y_stride, M, N, act);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M < 0 || N < 0) return 0;
  const int64_t chunks = (M + kColRows - 1) / kColRows;
  return (int64_t)sizeof(float) * (chunks > 0 ? chunks : 1) * (N > 0 ? N : 1);
}

static int colsum_launch(const float* a, int64_t as, const float* b, int64_t bs, const float* rw, int64_t M, int64_t N,
                         float* out, float scale, void* ws, hipStream_t st, const char* who) {
  const int64_t chunks = (M + kColRows - 1) / kColRows;
  REC_CHECK_ARG(chunks <= 65535, REC_ESHAPE, "%s: too many rows", who);
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, st, a, as,
                     b, bs, rw, M, N, part);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, part, chunks, N, out,
                     scale);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_colsum_f32(const float* a, int64_t a_stride, const float* b, int64_t b_stride, const float* row_w,
                              int64_t M, int64_t N, float* out, void* workspace, void* stream) {
  const char* who = "rec_colsum_f32";
  REC_CHECK_ARG(M >= 0 && N >= 1 && a_stride >= N && (!b || b_stride >= N), REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(out && workspace && (a || M == 0), REC_EINVAL, "%s: NULL pointer", who);
  if (M == 0) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * N, (hipStream_t)stream);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: %s", who, hipGetErrorString(e));
    return REC_OK;
  }
  return colsum_launch(a, a_stride, b, b_stride, row_w, M, N, out, 1.f, workspace, (hipStream_t)stream, who);
}

extern "C" int rec_bn_train_f32(const float* x, int64_t x_stride, int64_t M, int64_t N, const float* gamma,
                                const float* beta, float eps, float momentum, float* moving_mean, float* moving_var,
                                float* y, int64_t y_stride, float* save_mean, float* save_inv, void* workspace,
                                void* stream) {
  const char* who = "rec_bn_train_f32";
  REC_CHECK_ARG(M >= 1 && N >= 1 && x_stride >= N && y_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(x && y && save_mean && save_inv && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = (hipStream_t)stream;
  const float invM = 1.f / (float)M;
  // save_mean <- E[x], save_inv <- E[x^2] (scratch), then finished in place
  int rc = colsum_launch(x, x_stride, nullptr, 0, nullptr, M, N, save_mean, invM, workspace, st, who);
  if (rc != REC_OK) return rc;
  rc = colsum_launch(x, x_stride, x, x_stride, nullptr, M, N, save_inv, invM, workspace, st, who);
  if (rc != REC_OK) return rc;
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, save_mean, save_inv, N,
                     eps, momentum, moving_mean, moving_var, save_mean, save_inv);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, x, x_stride, M, N, gamma,
                     beta, save_mean, save_inv, y, y_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_bn_train_grad_f32(const float* x, int64_t x_stride, const float* dy, int64_t dy_stride, int64_t M,
                                     int64_t N, const float* gamma, const float* save_mean, const float* save_inv,
                                     float* dx, int64_t dx_stride, float* dgamma, float* dbeta, void* workspace,
                                     void* stream) {
  const char* who = "rec_bn_train_grad_f32";
  REC_CHECK_ARG(M >= 1 && N >= 1 && x_stride >= N && dy_stride >= N && dx_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(x && dy && dx && dgamma && dbeta && save_mean && save_inv && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = (hipStream_t)stream;
  // workspace: [xhat (M*N)] [mean(dy) (N)] [mean(dy xhat) (N)] [colsum partials]
  float* xh = static_cast<float*>(workspace);
  float* mdy = xh + M * N;
  float* mdyxh = mdy + N;
  float* part = mdyxh + N;
  hipLaunchKernelGGL(bn_xhat_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, x, x_stride, M, N,
                     save_mean, save_inv, xh);
  REC_CHECK_LAUNCH(who);
  int rc = colsum_launch(dy, dy_stride, nullptr, 0, nullptr, M, N, dbeta, 1.f, part, st, who);      // dbeta = sum dy
  if (rc != REC_OK) return rc;
  rc = colsum_launch(dy, dy_stride, xh, N, nullptr, M, N, dgamma, 1.f, part, st, who);               // dgamma = sum dy xhat
  if (rc != REC_OK) return rc;
  // the same sums as means over the batch (finish kernel over a single "chunk" = a scaled copy)
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, dbeta, (int64_t)1, N,
                     mdy, 1.f / (float)M);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, st, dgamma, (int64_t)1, N,
                     mdyxh, 1.f / (float)M);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(bn_grad_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, st, x, x_stride, dy, dy_stride,
                     M, N, gamma, save_mean, save_inv, mdy, mdyxh, dx, dx_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_bn_train_grad_workspace_bytes(int64_t M, int64_t N) {
  if (M < 0 || N < 0) return 0;
  return (int64_t)sizeof(float) * (M * N + 2 * N) + rec_colsum_workspace_bytes(M, N) + 1024;
}

extern "C" int rec_bce_sigmoid_grad_f32(const float* y_true, const float* p, int64_t n, float scale, float* dlogit,
                                        void* stream) {
  const char* who = "rec_bce_sigmoid_grad_f32";
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  if (n == 0) return REC_OK;
  REC_CHECK_ARG(y_true && p && dlogit, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(bce_sigmoid_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y_true,
                     p, n, scale, dlogit);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_pairwise_dot_grad_f32(const rec_table_desc* tables, const rec_table_desc* grad_tables, int32_t F,
                                                const int32_t* ids, int64_t ids_stride, const float* dense,
                                                int64_t dense_stride, int64_t B, const float* dz, int64_t dz_stride,
                                                int32_t append_dense, float* d_dense, int64_t d_dense_stride,
                                                void* stream) {
  const char* who = "rec_gather_pairwise_dot_grad_f32";
  TableSet ts, gs;
  int rc = fill_table_set(tables, F, &ts, who);
  if (rc != REC_OK) return rc;
  rc = fill_table_set(grad_tables, F, &gs, who);
  if (rc != REC_OK) return rc;
  const int D = tables[0].dim;
  for (int f = 0; f < F; ++f)
    REC_CHECK_ARG(tables[f].dim == D && grad_tables[f].dim == D && grad_tables[f].vocab == tables[f].vocab, REC_ESHAPE,
                  "%s: tables and gradient tables must share shapes", who);
  REC_CHECK_ARG(!append_dense || dense, REC_EINVAL, "%s: append_dense without dense", who);
  REC_CHECK_ARG(!dense || d_dense, REC_EINVAL, "%s: NULL d_dense", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(ids && dz, REC_EINVAL, "%s: NULL pointer", who);
  const int n = F + (dense ? 1 : 0);
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((B + 3) / 4);
#define REC_PDG(N_, CPL_)                                                                                         \
  if (n == (N_) && D == 64 * (CPL_)) {                                                                            \
    hipLaunchKernelGGL((pairdot_grad_kernel<N_, CPL_>), dim3(blocks), dim3(256), 0, st, ts, gs, ids, ids_stride, \
                       dense, dense_stride, dense ? 1 : 0, B, dz, dz_stride, append_dense, d_dense, d_dense_stride); \
    REC_CHECK_LAUNCH(who);                                                                                        \
    return REC_OK;                                                                                                \
  }
  REC_PDG(27, 2) REC_PDG(26, 2) REC_PDG(9, 2) REC_PDG(4, 2) REC_PDG(9, 1) REC_PDG(4, 1) REC_PDG(27, 1)
#undef REC_PDG
  REC_CHECK_ARG(n <= 64, REC_ESHAPE, "%s: n=%d rows per sample (max 64)", who, n);
  hipLaunchKernelGGL(pairdot_grad_generic_kernel, dim3(blocks), dim3(256), 0, st, ts, gs, F, ids, ids_stride, dense,
                     dense_stride, B, D, dz, dz_stride, append_dense, d_dense, d_dense_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_fm_layer_grad_f32(const float* first, int64_t first_stride, int64_t L1, const float* second,
                                     int64_t second_stride, int64_t M, const float* w, const float* dout, int64_t B,
                                     float* d_first, int64_t d_first_stride, float* d_second, int64_t d_second_stride,
                                     float* dw, void* workspace, void* stream) {
  const char* who = "rec_fm_layer_grad_f32";
  REC_CHECK_ARG(B >= 1 && L1 >= 1 && M >= 1, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(first && second && w && dout && d_second && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = (hipStream_t)stream;
  // workspace: [T (1 float, padded to 64)] [colsum(first) (L1)] [partials]
  float* T = static_cast<float*>(workspace);
  float* cs = T + 64;
  float* part = cs + L1;
  int rc = colsum_launch(dout, 1, nullptr, 0, nullptr, B, 1, T, 1.f, part, st, who);
  if (rc != REC_OK) return rc;
  rc = colsum_launch(first, first_stride, nullptr, 0, nullptr, B, L1, cs, 1.f, part, st, who);
  if (rc != REC_OK) return rc;
  hipLaunchKernelGGL(fm_second_grad_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, second, second_stride, M,
                     dout, B, d_second, d_second_stride);
  REC_CHECK_LAUNCH(who);
  const int64_t nthreads = d_first ? B * L1 : L1;
  hipLaunchKernelGGL(fm_first_grad_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, w, cs, T, L1, B,
                     d_first, d_first_stride, dw);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_fm_layer_grad_workspace_bytes(int64_t B, int64_t L1) {
  return (int64_t)sizeof(float) * (64 + L1) + rec_colsum_workspace_bytes(B, L1 > 1 ? L1 : 1) + 256;
}

extern "C" int rec_cross_layer_grad_f32(const float* x0, const float* xl, const float* w, int64_t dim, int64_t B, float* g,
                                        float* dx0, float* ds, void* stream) {
  const char* who = "rec_cross_layer_grad_f32";
  REC_CHECK_ARG(B >= 0 && dim >= 1, REC_ESHAPE, "%s: bad shape", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(x0 && xl && w && g && dx0 && ds, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(cross_grad_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x0, xl, w, dim,
                     B, g, dx0, ds);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_adam_rows_f32(const rec_table_desc* var, const rec_table_desc* m, const rec_table_desc* v,
                                 const rec_table_desc* grad, const rec_table_desc* stamp, int32_t F, const int32_t* ids,
                                 int64_t ids_stride, int64_t B, float lr, float beta1, float beta2, float eps,
                                 int64_t step, float l2, void* stream) {
  const char* who = "rec_adam_rows_f32";
  TableSet tv, tm, tvv, tg, tstamp;
  int rc = fill_table_set(var, F, &tv, who);
  if (rc == REC_OK) rc = fill_table_set(m, F, &tm, who);
  if (rc == REC_OK) rc = fill_table_set(v, F, &tvv, who);
  if (rc == REC_OK) rc = fill_table_set(grad, F, &tg, who);
  if (rc == REC_OK) rc = fill_table_set(stamp, F, &tstamp, who);
  if (rc != REC_OK) return rc;
  REC_CHECK_ARG(step >= 1 && step < 0x7fffffffLL && B >= 0 && ids_stride >= F, REC_ESHAPE, "%s: bad arguments", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(ids, REC_EINVAL, "%s: NULL ids", who);
  const double t = (double)step;
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
  const int64_t R = B * F;
  hipLaunchKernelGGL(adam_rows_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, tv, tm, tvv, tg,
                     tstamp, ids, ids_stride, F, R, (int32_t)step, lr_t, beta1, beta2, eps, 2.f * l2);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}
