// §8f-2 (first slice): the two numbers every ctr train script reports — Keras `binary_crossentropy` and
// `tf.keras.metrics.AUC()` (src/ctr/deep_fm/train.py:50-51,68: compile(loss=binary_crossentropy, metrics=[AUC()]),
// evaluate(...)[1]).  Both are one pass over (labels, predictions): HBM-bound, 8 B per sample.
//
//  BCE:  p clipped to [1e-7, 1 - 1e-7] (Keras epsilon), mean over samples of -(y log p + (1-y) log(1-p)).
//  AUC:  Keras defaults: 200 thresholds {0 - 1e-7, 1/199 .. 198/199, 1 + 1e-7}, ROC curve, 'interpolation'
//        (trapezoid) summation.  A prediction's bin = number of thresholds it exceeds (fp32 compare, like
//        `predictions > thresholds`), positives / negatives are histogrammed per bin (LDS-private, then integer
//        atomics -> deterministic), and TP/FP per threshold are suffix sums of the histograms.
#include <math.h>

#include "common.h"

namespace rec {

constexpr int AUC_T = 200;  // num_thresholds

__device__ __forceinline__ float auc_threshold(int j) {
  if (j == 0) return 0.0f - 1e-7f;
  if (j == AUC_T - 1) return 1.0f + 1e-7f;
  return (float)((double)j / (double)(AUC_T - 1));  // (i+1)/(num_thresholds-1), i = j-1, rounded to fp32
}

__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ y, const float* __restrict__ p,
                                                          int64_t n, double* __restrict__ part) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pc = fminf(fmaxf(p[i], 1e-7f), 1.0f - 1e-7f);
    const float yy = y[i];
    // Keras backend.binary_crossentropy on probabilities: clip to [eps, 1-eps], THEN log(p + eps) / log(1 - p + eps)
    // (tf.keras.backend: `bce = target * log(output + epsilon()); bce += (1 - target) * log(1 - output + epsilon())`).
    // In graph mode TF may instead recover the logits of a Sigmoid-op output and call
    // sigmoid_cross_entropy_with_logits (no clip): the two differ only on saturated predictions.  Parity unpinned
    // (no fixtures, TensorFlow not importable).
    acc += (double)(-(yy * logf(pc + 1e-7f) + (1.0f - yy) * logf(1.0f - pc + 1e-7f)));
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// add_loss of the match models (src/match/sasrec/model.py:93-95, src/match/ncf/model.py:75-77):
//   mean over (b, j) of [-log sigmoid(pos[b, 0]) - log(1 - sigmoid(neg[b, j]))] / 2   ((B,1) + (B,n) broadcasting)
// logits (B, 1 + n): column 0 = pos, columns 1.. = neg.  Written as the reference writes it (sigmoid, then log): a
// saturated score gives the same +inf TensorFlow gives.
__global__ __launch_bounds__(256) void pair_loss_partial_kernel(const float* __restrict__ logits, int64_t stride,
                                                                int64_t B, int n, double* __restrict__ part) {
  __shared__ double sh[4];
  double acc = 0.0;
  const int64_t total = B * (int64_t)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / n;
    const int j = (int)(i - b * n);
    const float sp = 1.f / (1.f + expf(-logits[b * stride]));
    const float sn = 1.f / (1.f + expf(-logits[b * stride + 1 + j]));
    acc += (double)(-logf(sp) - logf(1.f - sn));
  }
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void pair_loss_finish_kernel(const double* __restrict__ part, int nblk, int64_t n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[i];
  out[0] = (float)(s / (double)n * 0.5);
}

__global__ void bce_finish_kernel(const double* __restrict__ part, int nblk, int64_t n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < nblk; ++i) s += part[i];
  out[0] = (float)(s / (double)n);
}

// hist[0][b] = positives in bin b, hist[1][b] = negatives; bin b in [0, AUC_T]
__global__ __launch_bounds__(256) void auc_hist_kernel(const float* __restrict__ y, const float* __restrict__ p,
                                                       int64_t n, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int lh[2][AUC_T + 1];
  __shared__ float thr[AUC_T];
  for (int e = threadIdx.x; e < 2 * (AUC_T + 1); e += 256) (&lh[0][0])[e] = 0u;
  for (int e = threadIdx.x; e < AUC_T; e += 256) thr[e] = auc_threshold(e);
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = p[i];
    int lo = 0, hi = AUC_T;  // number of thresholds with v > thr[j] (thresholds ascending)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (v > thr[mid]) lo = mid + 1; else hi = mid;
    }
    atomicAdd(&lh[y[i] != 0.f ? 0 : 1][lo], 1u);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * (AUC_T + 1); e += 256) {
    const unsigned int c = (&lh[0][0])[e];
    if (c) atomicAdd(&hist[e], (unsigned long long)c);
  }
}

__global__ void auc_finish_kernel(const unsigned long long* __restrict__ hist, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // TP[j] = positives with p > thr[j] = sum of bins > j
  double tp[AUC_T], fp[AUC_T];
  double P = 0.0, Nn = 0.0, sp = 0.0, sn = 0.0;
  for (int b = 0; b <= AUC_T; ++b) P += (double)hist[b], Nn += (double)hist[AUC_T + 1 + b];
  for (int j = AUC_T - 1; j >= 0; --j) {
    sp += (double)hist[j + 1];
    sn += (double)hist[AUC_T + 1 + j + 1];
    tp[j] = sp;
    fp[j] = sn;
  }
  double auc = 0.0;
  for (int j = 0; j < AUC_T - 1; ++j) {
    // fp32 rates like Keras (div_no_nan)
    const float x0 = Nn > 0 ? (float)fp[j] / (float)Nn : 0.f, x1 = Nn > 0 ? (float)fp[j + 1] / (float)Nn : 0.f;
    const float y0 = P > 0 ? (float)tp[j] / (float)P : 0.f, y1 = P > 0 ? (float)tp[j + 1] / (float)P : 0.f;
    auc += (double)((x0 - x1) * ((y0 + y1) * 0.5f));
  }
  out[0] = (float)auc;
}

}  // namespace rec

using namespace rec;

static int metric_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int64_t rec_metrics_workspace_bytes(int64_t n) {
  (void)n;
  return 1024 * (int64_t)sizeof(double) + 2 * (AUC_T + 1) * (int64_t)sizeof(unsigned long long);
}

extern "C" int rec_binary_crossentropy_f32(const float* y_true, const float* y_pred, int64_t n, float* out,
                                           void* workspace, void* stream) {
  const char* who = "rec_binary_crossentropy_f32";
  REC_CHECK_ARG(n >= 1, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  REC_CHECK_ARG(y_true && y_pred && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int blocks = metric_blocks(n);
  hipLaunchKernelGGL(bce_partial_kernel, dim3(blocks), dim3(256), 0, st, y_true, y_pred, n,
                     static_cast<double*>(workspace));
  hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(64), 0, st, static_cast<const double*>(workspace), blocks, n,
                     out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_pairwise_rank_loss_f32(const float* logits, int64_t logits_stride, int64_t B, int32_t n_neg, float* out,
                                          void* workspace, void* stream) {
  const char* who = "rec_pairwise_rank_loss_f32";
  REC_CHECK_ARG(B >= 1 && n_neg >= 1 && logits_stride >= 1 + n_neg, REC_ESHAPE, "%s: B=%lld n_neg=%d stride=%lld", who,
                (long long)B, n_neg, (long long)logits_stride);
  REC_CHECK_ARG(logits && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int blocks = metric_blocks(B * (int64_t)n_neg);
  hipLaunchKernelGGL(pair_loss_partial_kernel, dim3(blocks), dim3(256), 0, st, logits, logits_stride, B, (int)n_neg,
                     static_cast<double*>(workspace));
  hipLaunchKernelGGL(pair_loss_finish_kernel, dim3(1), dim3(64), 0, st, static_cast<const double*>(workspace), blocks,
                     B * (int64_t)n_neg, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_auc_f32(const float* y_true, const float* y_pred, int64_t n, float* out, void* workspace,
                           void* stream) {
  const char* who = "rec_auc_f32";
  REC_CHECK_ARG(n >= 1, REC_ESHAPE, "%s: n=%lld", who, (long long)n);
  REC_CHECK_ARG(y_true && y_pred && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned long long* hist = reinterpret_cast<unsigned long long*>(static_cast<double*>(workspace) + 1024);
  REC_CHECK_ARG(hipMemsetAsync(hist, 0, 2 * (AUC_T + 1) * sizeof(unsigned long long), st) == hipSuccess, REC_EHIP,
                "%s: hipMemsetAsync failed", who);
  hipLaunchKernelGGL(auc_hist_kernel, dim3(metric_blocks(n)), dim3(256), 0, st, y_true, y_pred, n, hist);
  hipLaunchKernelGGL(auc_finish_kernel, dim3(1), dim3(64), 0, st, hist, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}
