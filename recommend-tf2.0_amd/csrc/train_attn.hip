// T3 — backward kernels of the attention-shaped layers and of the remaining model heads (what
// Keras' fit() differentiates in src/ctr/{fm,autoint,din}/model.py and src/match/sasrec/model.py).  The GEMM-shaped
// parts (projections, FFN) reuse the Dense backward of recamd/train.py; what is left are per-sample reductions over
// L2-resident operands, so every kernel here is a plain wave-per-row pass: deterministic (no atomics except the
// embedding-row scatter, which is the IndexedSlices sum of tf.gather), fp32 with fmaf chains, no tuning claims.
//
//   rec_attn_core_f32 / _grad_f32        softmax(scale q k^T [row-masked]) v per (sample, head) and its backward — the
//                                        core of the ctr MultiHeadAttention (src/ctr/layers/modules.py:221-283, scale
//                                        x sqrt(S), no mask) and of the match one (src/match/layers/modules.py:76-96,
//                                        scale 1/sqrt(depth), masked QUERY rows -> uniform attention, zero dq / dk)
//   rec_din_attn_pool_grad_f32           AttentionLayer backward (src/ctr/layers/modules.py:144-175)
//   rec_prelu_f32 / _grad_f32            tf.keras.layers.PReLU (per-feature alpha) as Dense activation (din/model.py:52)
//   rec_dice_train_f32 / _grad_f32            Dice (modules.py:327-337) around a training-mode BatchNormalization
//   rec_layernorm_residual_grad_f32      LN(x + r) gamma + beta [* row mask] backward (match/layers/modules.py:183-185)
//   rec_pairwise_rank_loss_grad_f32      add_loss of SASRec / NCF (match/sasrec/model.py:93-95)
//   rec_gather_dot_scores_grad_f32       logits[b, j] = table[ids[b, j]] . seq[b] backward (sasrec/model.py:88-91)
//   rec_fm_onehot_grad_f32               classic FM in gather form (src/ctr/fm/model.py:34-53) backward
//   rec_wgrad_small_f32                  dW = X^T dY for K, N <= 128 over a long batch axis (split over workgroups)
//   rec_dropout_f32                      Dropout(rate), training mode: counter-based mask (seed, element index), the
//                                        backward is the same call on dy
#include <math.h>

#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kNegMaskLogit = -4294967296.0f;   // float32(-2**32 + 1)

// ---- attention core -----------------------------------------------------------------------------------------------
// One wave per (b, h, i).  Lane l owns keys j = l, l + 64, ... (<= kMaxKeysPerLane of them): the logits and dP dots are
// lane-private loops over the head's S columns (k_j / v_j rows come from L2), the row softmax is two wave reductions,
// and the S-wide outputs (o_i or dq_i) are produced with lanes over columns reading the probability row from LDS.
constexpr int kMaxKeysPerLane = 8;   // Nk <= 512

// STAGED: one workgroup per (sample, head) keeps that head's K and V in LDS (row stride S + 1: the lane-private dots read
// one row per lane) and its waves walk the query rows — at SASRec's training shape (Nk = 200, S = 64) the unstaged form
// pulled 150 KB of K / V rows through L2 for EVERY query row (5.1 ms per call); unstaged remains for heads whose K and V do
// not fit (2 Nk (S + 1) floats + the per-wave rows <= 128 KiB).
constexpr int kAttnWaves = 8;
template <bool GRAD, bool STAGED>
__global__ __launch_bounds__(STAGED ? kAttnWaves * 64 : 256) void attn_row_kernel(
    const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk_g, const float* __restrict__ v,
    int64_t ldv_g, const float* __restrict__ row_mask, int64_t B, int Nq, int Nk, int H, int S, float scale,
    float* __restrict__ out, int64_t ldo, const float* __restrict__ dO, int64_t lddo, float* __restrict__ dq, int64_t lddq,
    float* __restrict__ Pws, float* __restrict__ dSws) {
  extern __shared__ float attn_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NW = STAGED ? kAttnWaves : 4;
  float* prow = attn_lds + (size_t)w * 2 * Nk;   // p_j
  float* grow = prow + Nk;                       // dS_j
  int64_t b;
  int h, i_first, i_step;
  if (STAGED) {
    b = blockIdx.x / H;
    h = blockIdx.x % H;
    i_first = w;
    i_step = NW;
  } else {
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    if (row >= B * H * Nq) return;                 // wave-uniform; no block barrier on this path
    i_first = (int)(row % Nq);
    h = (int)((row / Nq) % H);
    b = row / ((int64_t)Nq * H);
    i_step = Nq;                                   // exactly one query row
  }
  const float* kg = k + b * Nk * ldk_g + h * S;
  const float* vg = v + b * Nk * ldv_g + h * S;
  const float* kb = kg;
  const float* vb = vg;
  int64_t ldk = ldk_g, ldv = ldv_g;
  if (STAGED) {
    float* Ks = attn_lds + (size_t)NW * 2 * Nk;
    float* Vs = Ks + (size_t)Nk * (S + 1);
    for (int e = threadIdx.x; e < Nk * S; e += NW * 64) {
      const int j = e / S, c = e - j * S;
      Ks[j * (S + 1) + c] = kg[(int64_t)j * ldk_g + c];
      Vs[j * (S + 1) + c] = vg[(int64_t)j * ldv_g + c];
    }
    __syncthreads();
    kb = Ks;
    vb = Vs;
    ldk = ldv = S + 1;
  }
  for (int i = i_first; i < Nq; i += i_step) {
    const int64_t row = (b * H + h) * Nq + i;
    const float* qi = q + (b * Nq + i) * ldq + h * S;
    const bool masked = row_mask && row_mask[b * Nq + i] == 0.f;
    float s[kMaxKeysPerLane], p[kMaxKeysPerLane];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < kMaxKeysPerLane; ++t) {
      const int j = lane + 64 * t;
      s[t] = -INFINITY;
      if (j < Nk) {
        float a = 0.f;
        const float* kj = kb + (int64_t)j * ldk;
        for (int c = 0; c < S; ++c) a = fmaf(qi[c], kj[c], a);
        s[t] = masked ? kNegMaskLogit : a * scale;
      }
      mx = fmaxf(mx, s[t]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < kMaxKeysPerLane; ++t) {
      p[t] = (lane + 64 * t < Nk) ? expf(s[t] - mx) : 0.f;
      sum += p[t];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int t = 0; t < kMaxKeysPerLane; ++t) p[t] *= inv;
    if (!GRAD) {
#pragma unroll
      for (int t = 0; t < kMaxKeysPerLane; ++t)
        if (lane + 64 * t < Nk) prow[lane + 64 * t] = p[t];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float* oi = out + (b * Nq + i) * ldo + h * S;
      for (int c = lane; c < S; c += 64) {
        float a = 0.f;
        for (int j = 0; j < Nk; ++j) a = fmaf(prow[j], vb[(int64_t)j * ldv + c], a);
        oi[c] = a;
      }
      __builtin_amdgcn_wave_barrier();             // the next row overwrites prow
      continue;
    }
    const float* doi = dO + (b * Nq + i) * lddo + h * S;
    float dp[kMaxKeysPerLane];
    float delta = 0.f;
#pragma unroll
    for (int t = 0; t < kMaxKeysPerLane; ++t) {
      const int j = lane + 64 * t;
      dp[t] = 0.f;
      if (j < Nk) {
        float a = 0.f;
        const float* vj = vb + (int64_t)j * ldv;
        for (int c = 0; c < S; ++c) a = fmaf(doi[c], vj[c], a);
        dp[t] = a;
      }
      delta = fmaf(p[t], dp[t], delta);
    }
    delta = wave_sum(delta);
    float* Pw = Pws + row * Nk;
    float* Gw = dSws + row * Nk;
#pragma unroll
    for (int t = 0; t < kMaxKeysPerLane; ++t) {
      const int j = lane + 64 * t;
      if (j < Nk) {
        // a masked query row had every logit REPLACED by a constant (tf.where): no gradient reaches q_i or the keys
        const float g = masked ? 0.f : p[t] * (dp[t] - delta) * scale;
        Pw[j] = p[t];
        Gw[j] = g;
        grow[j] = g;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* dqi = dq + (b * Nq + i) * lddq + h * S;
    for (int c = lane; c < S; c += 64) {
      float a = 0.f;
      for (int j = 0; j < Nk; ++j) a = fmaf(grow[j], kb[(int64_t)j * ldk + c], a);
      dqi[c] = a;
    }
    __builtin_amdgcn_wave_barrier();               // the next row overwrites grow
  }
}

// dK_j = sum_i dS_ij q_i (scale already inside dS), dV_j = sum_i P_ij dO_i: lanes over columns.  STAGED: one workgroup per
// (sample, head) with that head's Q and dO in LDS, its waves walk the key rows; otherwise one wave per (b, h, j).
template <bool STAGED>
__global__ __launch_bounds__(STAGED ? kAttnWaves * 64 : 256) void attn_kv_grad_kernel(
    const float* __restrict__ q, int64_t ldq_g, const float* __restrict__ dO, int64_t lddo_g, int64_t B, int Nq, int Nk, int H,
    int S, const float* __restrict__ Pws, const float* __restrict__ dSws, float* __restrict__ dk, int64_t lddk,
    float* __restrict__ dv, int64_t lddv) {
  extern __shared__ float attn_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int NW = STAGED ? kAttnWaves : 4;
  int64_t b;
  int h, j_first, j_step;
  if (STAGED) {
    b = blockIdx.x / H;
    h = blockIdx.x % H;
    j_first = w;
    j_step = NW;
  } else {
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    if (row >= B * H * Nk) return;
    j_first = (int)(row % Nk);
    h = (int)((row / Nk) % H);
    b = row / ((int64_t)Nk * H);
    j_step = Nk;
  }
  const float* qg = q + b * Nq * ldq_g + h * S;
  const float* dog = dO + b * Nq * lddo_g + h * S;
  const float* qb = qg;
  const float* dob = dog;
  int64_t ldq = ldq_g, lddo = lddo_g;
  if (STAGED) {
    float* Qs = attn_lds;
    float* Ds = Qs + (size_t)Nq * S;
    for (int e = threadIdx.x; e < Nq * S; e += NW * 64) {
      const int i = e / S, c = e - i * S;
      Qs[e] = qg[(int64_t)i * ldq_g + c];
      Ds[e] = dog[(int64_t)i * lddo_g + c];
    }
    __syncthreads();
    qb = Qs;
    dob = Ds;
    ldq = lddo = S;
  }
  // STAGED: column j of P and dS (stride Nk in the workspace) is gathered into a wave-private LDS line first — read in the
  // accumulation loop straight from the workspace, every term waited for its own L2 round trip (1.8 ms per call at
  // SASRec's shape)
  float* pcol = attn_lds + (size_t)2 * Nq * S + (size_t)w * 2 * Nq;
  float* gcol = pcol + Nq;
  for (int j = j_first; j < Nk; j += j_step) {
    const float* Pb = Pws + ((b * H + h) * Nq) * (int64_t)Nk + j;
    const float* Gb = dSws + ((b * H + h) * Nq) * (int64_t)Nk + j;
    if (STAGED) {
      for (int i = lane; i < Nq; i += 64) {
        pcol[i] = Pb[(int64_t)i * Nk];
        gcol[i] = Gb[(int64_t)i * Nk];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    for (int c = lane; c < S; c += 64) {
      float ak = 0.f, av = 0.f;
      for (int i = 0; i < Nq; ++i) {
        const float g = STAGED ? gcol[i] : Gb[(int64_t)i * Nk];
        const float pp = STAGED ? pcol[i] : Pb[(int64_t)i * Nk];
        ak = fmaf(g, qb[(int64_t)i * ldq + c], ak);
        av = fmaf(pp, dob[(int64_t)i * lddo + c], av);
      }
      dk[(b * Nk + j) * lddk + h * S + c] = ak;
      dv[(b * Nk + j) * lddv + h * S + c] = av;
    }
    if (STAGED) __builtin_amdgcn_wave_barrier();   // the next key row overwrites the lines
  }
}

// ---- DIN AttentionLayer backward ------------------------------------------------------------------------------
// z_j = [q, k_j, q - k_j, q o k_j] . W + b ; a_j = act(z_j) ; logits = mask ? a_j : pad ; p = softmax ; out = sum p_j v_j
// One wave per sample.  Lane-private loops over d for the T scores / dP dots (lane = slot), the per-slot scalars go
// through LDS, then lanes over the d columns produce dq, dk, dv and this sample's row of the (B, 4d + 2) parameter-
// gradient partials [dW (4d) | db | dalpha]; rec_colsum_f32 over that matrix finishes them deterministically.
constexpr int kMaxSlotsPerLane = 4;  // T <= 256
__global__ __launch_bounds__(256) void din_pool_grad_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ mask,
                                                            int mask_mode, const float* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ alpha, int act,
                                                            const float* __restrict__ dout, int64_t B, int T, int d,
                                                            float* __restrict__ dq, float* __restrict__ dk,
                                                            float* __restrict__ dv, float* __restrict__ part) {
  extern __shared__ float din_g_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* pz = din_g_lds + (size_t)w * 2 * T;   // dz_j
  float* pp = pz + T;                          // p_j
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;
  const float* qb = q + b * d;
  const float* kb = k + b * (int64_t)T * d;
  const float* vb = v + b * (int64_t)T * d;
  const float* dob = dout + b * d;
  const float al = alpha ? alpha[0] : 0.f;
  float z[kMaxSlotsPerLane], a[kMaxSlotsPerLane], p[kMaxSlotsPerLane], dp[kMaxSlotsPerLane];
  bool pad[kMaxSlotsPerLane];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < kMaxSlotsPerLane; ++t) {
    const int j = lane + 64 * t;
    z[t] = 0.f; a[t] = -INFINITY; dp[t] = 0.f; pad[t] = true;
    if (j < T) {
      const float* kj = kb + (int64_t)j * d;
      const float* vj = vb + (int64_t)j * d;
      float acc = 0.f, accp = 0.f;
      for (int c = 0; c < d; ++c) {
        const float qc = qb[c], kc = kj[c];
        acc = fmaf(qc, W[c], acc);
        acc = fmaf(kc, W[d + c], acc);
        acc = fmaf(qc - kc, W[2 * d + c], acc);
        acc = fmaf(qc * kc, W[3 * d + c], acc);
        accp = fmaf(dob[c], vj[c], accp);
      }
      z[t] = acc + bias[0];
      dp[t] = accp;
      pad[t] = mask_mode == 0 ? true : (mask[b * T + j] == 0.f);
      a[t] = pad[t] ? kNegMaskLogit : act_apply(z[t], act, al);
    }
    mx = fmaxf(mx, a[t]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < kMaxSlotsPerLane; ++t) {
    p[t] = (lane + 64 * t < T) ? expf(a[t] - mx) : 0.f;
    sum += p[t];
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  float delta = 0.f;
#pragma unroll
  for (int t = 0; t < kMaxSlotsPerLane; ++t) {
    p[t] *= inv;
    delta = fmaf(p[t], dp[t], delta);
  }
  delta = wave_sum(delta);
  float dbs = 0.f, das = 0.f;
#pragma unroll
  for (int t = 0; t < kMaxSlotsPerLane; ++t) {
    const int j = lane + 64 * t;
    if (j < T) {
      float da = pad[t] ? 0.f : p[t] * (dp[t] - delta);   // padded slots: the logit is a constant
      float dz = da;
      if (act == REC_ACT_RELU) dz = z[t] > 0.f ? da : 0.f;
      else if (act == REC_ACT_SIGMOID) { const float sg = 1.f / (1.f + expf(-z[t])); dz = da * sg * (1.f - sg); }
      else if (act == REC_ACT_TANH) { const float th = tanhf(z[t]); dz = da * (1.f - th * th); }
      else if (act == REC_ACT_PRELU) { dz = z[t] >= 0.f ? da : da * al; das += z[t] >= 0.f ? 0.f : da * z[t]; }
      dbs += dz;
      pz[j] = dz;
      pp[j] = p[t];
    }
  }
  dbs = wave_sum(dbs);
  das = wave_sum(das);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float* pr = part + b * (int64_t)(4 * d + 2);
  for (int c = lane; c < d; c += 64) {
    const float qc = qb[c], w2 = W[d + c], w3 = W[2 * d + c], w4 = W[3 * d + c], doc = dob[c];
    const float uk = w2 - w3 + qc * w4;
    float skz = 0.f;      // sum_j dz_j k_jc
    for (int j = 0; j < T; ++j) {
      const float dz = pz[j], kc = kb[(int64_t)j * d + c];
      skz = fmaf(dz, kc, skz);
      dk[(b * T + j) * (int64_t)d + c] = dz * uk;
      dv[(b * T + j) * (int64_t)d + c] = pp[j] * doc;
    }
    dq[b * d + c] = dbs * (W[c] + w3) + skz * w4;
    pr[c] = dbs * qc;                  // d W1
    pr[d + c] = skz;                   // d W2
    pr[2 * d + c] = dbs * qc - skz;    // d W3
    pr[3 * d + c] = qc * skz;          // d W4
  }
  if (lane == 0) {
    pr[4 * d] = dbs;
    pr[4 * d + 1] = das;
  }
}

// ---- PReLU / Dice ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prelu_kernel(const float* __restrict__ z, int64_t zs, const float* __restrict__ alpha,
                                                    int64_t M, int64_t N, float* __restrict__ y, int64_t ys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float x = z[m * zs + n];
  y[m * ys + n] = x >= 0.f ? x : alpha[n] * x;
}
// dz = dy (z >= 0 ? 1 : alpha_n); nz = min(z, 0)  (d alpha_n = column sum of dy o nz)
__global__ __launch_bounds__(256) void prelu_grad_kernel(const float* __restrict__ z, int64_t zs,
                                                         const float* __restrict__ alpha, const float* __restrict__ dy,
                                                         int64_t dys, int64_t M, int64_t N, float* __restrict__ dz,
                                                         float* __restrict__ nz) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N, n = i - m * N;
  const float x = z[m * zs + n], g = dy[m * dys + n];
  dz[i] = x >= 0.f ? g : g * alpha[n];
  nz[i] = x >= 0.f ? 0.f : x;
}
// y = x (alpha + (1 - alpha) p), p = sigmoid(xn), xn = the un-affine BatchNormalization of x
__global__ __launch_bounds__(256) void dice_kernel(const float* __restrict__ x, const float* __restrict__ xn,
                                                   const float* __restrict__ alpha, int64_t n, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p = 1.f / (1.f + expf(-xn[i])), a = alpha[0];
  y[i] = x[i] * (a + (1.f - a) * p);
}
// dx_direct = dy (alpha + (1 - alpha) p); dxn = dy x (1 - alpha) p (1 - p); da_elem = dy x (1 - p)
__global__ __launch_bounds__(256) void dice_grad_kernel(const float* __restrict__ x, const float* __restrict__ xn,
                                                        const float* __restrict__ alpha, const float* __restrict__ dy,
                                                        int64_t n, float* __restrict__ dx, float* __restrict__ dxn,
                                                        float* __restrict__ da) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p = 1.f / (1.f + expf(-xn[i])), a = alpha[0], g = dy[i], xv = x[i];
  dx[i] = g * (a + (1.f - a) * p);
  dxn[i] = g * xv * (1.f - a) * p * (1.f - p);
  da[i] = g * xv * (1.f - p);
}

// ---- LayerNormalization(x + r) gamma + beta [* row mask], backward -------------------------------------------------
// wave per row; ds = d(x + r); xhat and the masked dy are written for the two column sums (d gamma = colsum(dym o xhat),
// d beta = colsum(dym))
__global__ __launch_bounds__(256) void ln_grad_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                      const float* __restrict__ gamma, const float* __restrict__ row_mask,
                                                      const float* __restrict__ dy, int64_t M, int d, float eps,
                                                      float* __restrict__ ds, float* __restrict__ xhat,
                                                      float* __restrict__ dym) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 4 + w;
  if (m >= M) return;
  const float rm = row_mask ? row_mask[m] : 1.f;
  float sum = 0.f;
  for (int c = lane; c < d; c += 64) sum += x[m * d + c] + (r ? r[m * d + c] : 0.f);
  const float mu = wave_sum(sum) / (float)d;
  float var = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float t = x[m * d + c] + (r ? r[m * d + c] : 0.f) - mu;
    var = fmaf(t, t, var);
  }
  const float inv = 1.f / sqrtf(wave_sum(var) / (float)d + eps);
  float a1 = 0.f, a2 = 0.f;   // mean(dxhat), mean(dxhat xhat)
  for (int c = lane; c < d; c += 64) {
    const float xh = (x[m * d + c] + (r ? r[m * d + c] : 0.f) - mu) * inv;
    const float g = dy[m * d + c] * rm;
    const float dxh = g * gamma[c];
    xhat[m * d + c] = xh;
    dym[m * d + c] = g;
    a1 += dxh;
    a2 = fmaf(dxh, xh, a2);
  }
  a1 = wave_sum(a1) / (float)d;
  a2 = wave_sum(a2) / (float)d;
  for (int c = lane; c < d; c += 64) {
    const float dxh = dym[m * d + c] * gamma[c];
    ds[m * d + c] = inv * (dxh - a1 - xhat[m * d + c] * a2);
  }
}

// ---- add_loss of SASRec / NCF, backward ----------------------------------------------------------------------------
// loss = mean_{b, j} [-log sigmoid(pos_b) - log(1 - sigmoid(neg_bj))] / 2 ; dlogits = scale * d loss / d logits
__global__ __launch_bounds__(256) void rank_loss_grad_kernel(const float* __restrict__ logits, int64_t ls, int64_t B, int n_neg,
                                                             float scale, float* __restrict__ dl, int64_t ds_) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = (int64_t)(n_neg + 1);
  if (i >= B * n) return;
  const int64_t b = i / n, j = i - b * n;
  const float sg = 1.f / (1.f + expf(-logits[b * ls + j]));
  const float c = scale / (2.f * (float)B * (float)n_neg);
  dl[b * ds_ + j] = j == 0 ? c * (float)n_neg * (sg - 1.f) : c * sg;
}

// ---- logits[b, j] = table[ids[b, j]] . seq[b], backward --------------------------------------------------------------
// wave per sample: dseq[b] (+)= sum_j dl[b, j] row_j ; grad_table[ids[b, j]] += dl[b, j] seq[b] (fp32 atomics, 256-B shape)
__global__ __launch_bounds__(256) void dot_scores_grad_kernel(const float* __restrict__ seq, const float* __restrict__ table,
                                                              float* __restrict__ gtable, int64_t vocab, int d,
                                                              const int32_t* __restrict__ ids, int64_t ids_stride, int n,
                                                              const float* __restrict__ dl, int64_t dls, int64_t B,
                                                              float* __restrict__ dseq, int accumulate) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;
  for (int c = lane; c < d; c += 64) {
    const float sc = seq[b * d + c];
    float a = accumulate ? dseq[b * d + c] : 0.f;
    for (int j = 0; j < n; ++j) {
      const int32_t id = ids[b * ids_stride + j];
      if ((uint32_t)id >= (uint32_t)vocab) continue;
      const float g = dl[b * dls + j];
      a = fmaf(g, table[(int64_t)id * d + c], a);
      atomicAdd(gtable + (int64_t)id * d + c, g * sc);
    }
    dseq[b * d + c] = a;
  }
}

// ---- classic FM (gather form), backward -----------------------------------------------------------------------------
// y = w0 + sum_c x_c w_c + 0.5 sum_k [(sum_c x_c V_kc)^2 - sum_c x_c^2 V_kc^2] over the non-zero columns c of the
// [dense | one-hot] stack: x_c = dense value for c < nd, 1 for c = nd + off_f + id_f (out-of-range id: no column).
// One wave per sample, lane = feature (nd + F <= 64), the k sums by wave reductions; the parameter gradients are
// scatter-added with fp32 atomics (w: (L), V: (k, L)); dw0 per-sample partial -> rec_colsum_f32.
struct FmOffsets { int32_t off[REC_MAX_TABLES]; int32_t vocab[REC_MAX_TABLES]; };
__global__ __launch_bounds__(256) void fm_onehot_grad_kernel(const float* __restrict__ dense, int64_t dense_stride, int nd,
                                                             const int32_t* __restrict__ ids, int64_t ids_stride, int F,
                                                             FmOffsets fo, const float* __restrict__ V, int kdim, int64_t L,
                                                             const float* __restrict__ dlogit, int64_t B,
                                                             float* __restrict__ dw, float* __restrict__ dV) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;
  const float g = dlogit[b];
  int64_t col = -1;
  float xv = 0.f;
  if (lane < nd) {
    col = lane;
    xv = dense[b * dense_stride + lane];
  } else if (lane < nd + F) {
    const int f = lane - nd;
    const int32_t id = ids[b * ids_stride + f];
    if ((uint32_t)id < (uint32_t)fo.vocab[f]) {
      col = (int64_t)nd + fo.off[f] + id;
      xv = 1.f;
    }
  }
  if (col >= 0) atomicAdd(dw + col, g * xv);
  for (int kk = 0; kk < kdim; ++kk) {
    const float vkc = col >= 0 ? V[(int64_t)kk * L + col] : 0.f;
    const float sk = wave_sum(xv * vkc);
    if (col >= 0) atomicAdd(dV + (int64_t)kk * L + col, g * (xv * sk - xv * xv * vkc));
  }
}

// ---- dropout -------------------------------------------------------------------------------------------------------
// keep element e iff hash(seed, e) >= rate * 2^32; kept values are scaled by 1 / (1 - rate) (tf.nn.dropout).  The mask
// is a pure function of (seed, e): the backward pass applies the same call to dy.
__device__ __forceinline__ uint32_t mix32(uint64_t x) {   // splitmix64 finaliser, upper half
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (uint32_t)(x >> 32);
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t n, uint32_t thresh, float keep_scale,
                                                      uint64_t seed, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = mix32(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
  y[i] = r >= thresh ? x[i] * keep_scale : 0.f;
}

// ---- dW = X^T dY for a SMALL output and a LONG reduction -----------------------------------------------------------------
// The Dense backward of the attention-shaped models multiplies (K x M) by (M x N) with K, N <= 128 and M = batch x
// positions (1e5 rows): one or two output tiles for the GEMM kernels, i.e. ONE workgroup walking the whole reduction
// (16 ms per call at AutoInt's configs[2] shape — 96 % of its training step).  Here the rows are split over workgroups:
// each stages 32-row slabs of X and dY in LDS and the per-workgroup partials are summed in a fixed order in fp64
// (deterministic).
constexpr int kWgSpan = 256, kWgSlab = 32;    // rows per workgroup (1024: too few workgroups to hide the slab loads), rows per LDS slab (at most)
constexpr int kWgLdsFloats = 12288;          // 48 KiB of slabs: wide layers take fewer rows per slab
// thread t = (n = t % N, kg = t / N) owns outputs (k, n) for k = kg, kg + G, ..., G = 256 / N k-groups: per staged row ONE
// dY read (conflict-free across the wave) feeds up to JMAX independent FMAs whose X operands are broadcast reads — the
// first version (outputs p = t, t + 256, ... with the row loop innermost) had one dependent LDS round trip per FMA and ran
// 0.9 ms per call at SASRec's shape.
// VEC (J and K multiples of 4): a thread's J consecutive k values are read four at a time (one broadcast ds_read_b128 per
// four FMAs; with one read per FMA the K = 64, N = 128 layer of SASRec's FFN ran 0.62 ms per call, LDS-issue bound).
template <int JMAX, bool VEC>
__global__ __launch_bounds__(256) void wgrad_small_partial_kernel(const float* __restrict__ x, int64_t ldx,
                                                                  const float* __restrict__ dy, int64_t ldy, int64_t M, int K,
                                                                  int N, int J, int slab, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];
  float* xs = wg_lds;                 // [slab][K]
  float* ds = wg_lds + slab * K;      // [slab][N]
  const int t = threadIdx.x;
  const int G = 256 / N;              // N <= 256 (host-checked); J = ceil(K / G) <= JMAX
  const int n = t % N, kg = t / N;
  const int k0 = kg * J;              // this thread's outputs: (k0 .. k0 + J - 1, n)
  const bool active = kg < G && k0 < K;
  const int64_t r_begin = (int64_t)blockIdx.x * kWgSpan;
  const int64_t r_end = r_begin + kWgSpan < M ? r_begin + kWgSpan : M;
  float acc[JMAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) acc[j] = 0.f;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += slab) {
    const int rows = (int)(r_end - r0 < slab ? r_end - r0 : slab);
    __syncthreads();
    for (int e = t; e < rows * K; e += 256) xs[e] = x[(r0 + e / K) * ldx + e % K];
    for (int e = t; e < rows * N; e += 256) ds[e] = dy[(r0 + e / N) * ldy + e % N];
    __syncthreads();
    if (active) {
      for (int r = 0; r < rows; ++r) {
        const float d = ds[r * N + n];
        const float* xr = xs + r * K + k0;
        if constexpr (VEC) {
#pragma unroll
          for (int j = 0; j < JMAX; j += 4) {
            if (j < J && k0 + j < K) {                 // K, J, k0 multiples of 4: the whole quad is inside
              const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + j);
              acc[j] = fmaf(xv.x, d, acc[j]);
              acc[j + 1] = fmaf(xv.y, d, acc[j + 1]);
              acc[j + 2] = fmaf(xv.z, d, acc[j + 2]);
              acc[j + 3] = fmaf(xv.w, d, acc[j + 3]);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < JMAX; ++j)
            if (j < J && k0 + j < K) acc[j] = fmaf(xr[j], d, acc[j]);
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int k = k0 + j;
      if (j < J && k < K) part[(int64_t)blockIdx.x * K * N + (int64_t)k * N + n] = acc[j];
    }
  }
}
// ---- Keras binary cross-entropy on a PROBABILITY that is not a sigmoid output (ESMM's pCTCVR = pCTR * pCVR) ----------------
// dL/dp_i = scale * [ -(y/(pc+e)) + (1-y)/(1-pc+e) ] * [e < p < 1-e], pc = clip(p, e, 1-e)
__global__ __launch_bounds__(256) void bce_prob_grad_kernel(const float* __restrict__ y, const float* __restrict__ p, int64_t n,
                                                            float scale, float* __restrict__ dp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float e = 1e-7f, pi = p[i], yi = y[i];
  const float pc = fminf(fmaxf(pi, e), 1.f - e);
  const float inside = (pi > e && pi < 1.f - e) ? 1.f : 0.f;
  dp[i] = scale * inside * (-(yi / (pc + e)) + (1.f - yi) / (1.f - pc + e));
}

}  // namespace rec

using namespace rec;

static inline unsigned blocks_of(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

extern "C" int64_t rec_attn_core_grad_workspace_bytes(int64_t B, int32_t Nq, int32_t Nk, int32_t H) {
  return 2 * B * H * (int64_t)Nq * Nk * (int64_t)sizeof(float);
}

static int attn_check(const char* who, int64_t B, int Nq, int Nk, int H, int S) {
  REC_CHECK_ARG(B >= 0 && Nq >= 1 && Nk >= 1 && H >= 1 && S >= 1, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(Nk <= 64 * kMaxKeysPerLane, REC_ENOTIMPL, "%s: Nk = %d > %d keys", who, Nk, 64 * kMaxKeysPerLane);
  REC_CHECK_ARG((size_t)4 * 2 * Nk * sizeof(float) <= 64 * 1024, REC_ENOTIMPL, "%s: Nk too large for LDS", who);
  REC_CHECK_ARG(B * H * (int64_t)Nq < ((int64_t)1 << 33) && B * H * (int64_t)Nk < ((int64_t)1 << 33), REC_ESHAPE,
                "%s: too many rows", who);
  return REC_OK;
}

// dynamic LDS above 64 KiB needs the attribute; staged forms are used when the head's operands fit 128 KiB
constexpr size_t kAttnStageMax = 128 * 1024;
template <typename K>
static bool raise_lds(K kern, size_t lds) {
  return lds <= 64 * 1024 ||
         hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
}
static size_t attn_row_staged_lds(int Nk, int S) { return ((size_t)kAttnWaves * 2 * Nk + (size_t)2 * Nk * (S + 1)) * sizeof(float); }

extern "C" int rec_attn_core_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                 const float* row_mask, int64_t B, int32_t Nq, int32_t Nk, int32_t H, int32_t S,
                                 float scale, float* out, int64_t ldo, void* stream) {
  const char* who = "rec_attn_core_f32";
  if (int rc = attn_check(who, B, Nq, Nk, H, S)) return rc;
  REC_CHECK_ARG(q && k && v && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(ldq >= (int64_t)H * S && ldk >= (int64_t)H * S && ldv >= (int64_t)H * S && ldo >= (int64_t)H * S, REC_ESHAPE,
                "%s: row stride < H * S", who);
  if (B == 0) return REC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t slds = attn_row_staged_lds(Nk, S);
  if (slds <= kAttnStageMax && B * H <= 0x7fffffffLL && raise_lds(attn_row_kernel<false, true>, slds)) {
    hipLaunchKernelGGL((attn_row_kernel<false, true>), dim3((unsigned)(B * H)), dim3(kAttnWaves * 64), slds, st, q, ldq, k, ldk,
                       v, ldv, row_mask, B, Nq, Nk, H, S, scale, out, ldo, (const float*)nullptr, (int64_t)0, (float*)nullptr,
                       (int64_t)0, (float*)nullptr, (float*)nullptr);
  } else {
    const size_t lds = (size_t)4 * 2 * Nk * sizeof(float);
    hipLaunchKernelGGL((attn_row_kernel<false, false>), dim3(blocks_of(B * H * Nq, 4)), dim3(256), lds, st, q, ldq, k, ldk, v,
                       ldv, row_mask, B, Nq, Nk, H, S, scale, out, ldo, (const float*)nullptr, (int64_t)0, (float*)nullptr,
                       (int64_t)0, (float*)nullptr, (float*)nullptr);
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_attn_core_grad_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                      const float* row_mask, const float* dout, int64_t lddo, int64_t B, int32_t Nq,
                                      int32_t Nk, int32_t H, int32_t S, float scale, float* dq, int64_t lddq, float* dk,
                                      int64_t lddk, float* dv, int64_t lddv, void* workspace, void* stream) {
  const char* who = "rec_attn_core_grad_f32";
  if (int rc = attn_check(who, B, Nq, Nk, H, S)) return rc;
  REC_CHECK_ARG(q && k && v && dout && dq && dk && dv && workspace, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t hs = (int64_t)H * S;
  REC_CHECK_ARG(ldq >= hs && ldk >= hs && ldv >= hs && lddo >= hs && lddq >= hs && lddk >= hs && lddv >= hs, REC_ESHAPE,
                "%s: row stride < H * S", who);
  if (B == 0) return REC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* Pws = reinterpret_cast<float*>(workspace);
  float* Gws = Pws + B * H * (int64_t)Nq * Nk;
  const size_t slds = attn_row_staged_lds(Nk, S);
  if (slds <= kAttnStageMax && B * H <= 0x7fffffffLL && raise_lds(attn_row_kernel<true, true>, slds)) {
    hipLaunchKernelGGL((attn_row_kernel<true, true>), dim3((unsigned)(B * H)), dim3(kAttnWaves * 64), slds, st, q, ldq, k, ldk,
                       v, ldv, row_mask, B, Nq, Nk, H, S, scale, (float*)nullptr, (int64_t)0, dout, lddo, dq, lddq, Pws, Gws);
  } else {
    const size_t lds = (size_t)4 * 2 * Nk * sizeof(float);
    hipLaunchKernelGGL((attn_row_kernel<true, false>), dim3(blocks_of(B * H * Nq, 4)), dim3(256), lds, st, q, ldq, k, ldk, v,
                       ldv, row_mask, B, Nq, Nk, H, S, scale, (float*)nullptr, (int64_t)0, dout, lddo, dq, lddq, Pws, Gws);
  }
  REC_CHECK_LAUNCH(who);
  const size_t klds = ((size_t)2 * Nq * S + (size_t)kAttnWaves * 2 * Nq) * sizeof(float);
  if (klds <= kAttnStageMax && B * H <= 0x7fffffffLL && raise_lds(attn_kv_grad_kernel<true>, klds)) {
    hipLaunchKernelGGL((attn_kv_grad_kernel<true>), dim3((unsigned)(B * H)), dim3(kAttnWaves * 64), klds, st, q, ldq, dout, lddo,
                       B, Nq, Nk, H, S, (const float*)Pws, (const float*)Gws, dk, lddk, dv, lddv);
  } else {
    hipLaunchKernelGGL((attn_kv_grad_kernel<false>), dim3(blocks_of(B * H * Nk, 4)), dim3(256), 0, st, q, ldq, dout, lddo, B, Nq,
                       Nk, H, S, (const float*)Pws, (const float*)Gws, dk, lddk, dv, lddv);
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_din_attn_pool_grad_f32(const float* q, const float* k, const float* v, const float* mask,
                                          int32_t mask_is_none, const float* W, const float* bias, int32_t act,
                                          const float* alpha, const float* dout, int64_t B, int32_t T, int32_t d, float* dq,
                                          float* dk, float* dv, float* partials, void* stream) {
  const char* who = "rec_din_attn_pool_grad_f32";
  REC_CHECK_ARG(q && k && v && W && bias && dout && dq && dk && dv && partials, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(B >= 0 && T >= 1 && d >= 1, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(T <= 64 * kMaxSlotsPerLane, REC_ENOTIMPL, "%s: T = %d > %d", who, T, 64 * kMaxSlotsPerLane);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_PRELU, REC_EINVAL, "%s: bad activation", who);
  if (B == 0) return REC_OK;
  const int mode = (mask && !mask_is_none) ? 1 : 0;
  hipLaunchKernelGGL(din_pool_grad_kernel, dim3(blocks_of(B, 4)), dim3(256), (size_t)4 * 2 * T * sizeof(float),
                     reinterpret_cast<hipStream_t>(stream), q, k, v, mask, mode, W, bias, alpha, act, dout, B, T, d, dq, dk,
                     dv, partials);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_prelu_f32(const float* z, int64_t z_stride, const float* alpha, int64_t M, int64_t N, float* y,
                             int64_t y_stride, void* stream) {
  const char* who = "rec_prelu_f32";
  REC_CHECK_ARG(z && alpha && y, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(M >= 0 && N >= 1 && z_stride >= N && y_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  if (M == 0) return REC_OK;
  hipLaunchKernelGGL(prelu_kernel, dim3(blocks_of(M * N, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z,
                     z_stride, alpha, M, N, y, y_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_prelu_grad_f32(const float* z, int64_t z_stride, const float* alpha, const float* dy, int64_t dy_stride,
                                  int64_t M, int64_t N, float* dz, float* neg_part, void* stream) {
  const char* who = "rec_prelu_grad_f32";
  REC_CHECK_ARG(z && alpha && dy && dz && neg_part, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(M >= 0 && N >= 1 && z_stride >= N && dy_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  if (M == 0) return REC_OK;
  hipLaunchKernelGGL(prelu_grad_kernel, dim3(blocks_of(M * N, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z,
                     z_stride, alpha, dy, dy_stride, M, N, dz, neg_part);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dice_train_f32(const float* x, const float* xn, const float* alpha, int64_t n, float* y, void* stream) {
  const char* who = "rec_dice_train_f32";
  REC_CHECK_ARG(x && xn && alpha && y, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: bad shape", who);
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(dice_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, xn, alpha,
                     n, y);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dice_train_grad_f32(const float* x, const float* xn, const float* alpha, const float* dy, int64_t n, float* dx,
                                 float* dxn, float* dalpha_elem, void* stream) {
  const char* who = "rec_dice_train_grad_f32";
  REC_CHECK_ARG(x && xn && alpha && dy && dx && dxn && dalpha_elem, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: bad shape", who);
  if (n == 0) return REC_OK;
  hipLaunchKernelGGL(dice_grad_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, xn,
                     alpha, dy, n, dx, dxn, dalpha_elem);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_layernorm_residual_grad_f32(const float* x, const float* residual, const float* gamma,
                                               const float* row_mask, const float* dy, int64_t M, int32_t d, float eps,
                                               float* ds, float* xhat, float* dy_masked, void* stream) {
  const char* who = "rec_layernorm_residual_grad_f32";
  REC_CHECK_ARG(x && gamma && dy && ds && xhat && dy_masked, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(M >= 0 && d >= 1, REC_ESHAPE, "%s: bad shape", who);
  if (M == 0) return REC_OK;
  hipLaunchKernelGGL(ln_grad_kernel, dim3(blocks_of(M, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, residual,
                     gamma, row_mask, dy, M, d, eps, ds, xhat, dy_masked);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_pairwise_rank_loss_grad_f32(const float* logits, int64_t logits_stride, int64_t B, int32_t n_neg,
                                               float scale, float* dlogits, int64_t dlogits_stride, void* stream) {
  const char* who = "rec_pairwise_rank_loss_grad_f32";
  REC_CHECK_ARG(logits && dlogits, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(B >= 1 && n_neg >= 1 && logits_stride >= n_neg + 1 && dlogits_stride >= n_neg + 1, REC_ESHAPE,
                "%s: bad shape", who);
  hipLaunchKernelGGL(rank_loss_grad_kernel, dim3(blocks_of(B * (n_neg + 1), 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), logits, logits_stride, B, n_neg, scale, dlogits, dlogits_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_dot_scores_grad_f32(const float* seq, const float* table, float* grad_table, int64_t vocab,
                                              int32_t d, const int32_t* ids, int64_t ids_stride, int32_t n,
                                              const float* dlogits, int64_t dlogits_stride, int64_t B, float* dseq,
                                              int32_t accumulate, void* stream) {
  const char* who = "rec_gather_dot_scores_grad_f32";
  REC_CHECK_ARG(seq && table && grad_table && ids && dlogits && dseq, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(B >= 0 && n >= 1 && d >= 1 && vocab >= 1 && ids_stride >= n && dlogits_stride >= n, REC_ESHAPE,
                "%s: bad shape", who);
  if (B == 0) return REC_OK;
  hipLaunchKernelGGL(dot_scores_grad_kernel, dim3(blocks_of(B, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), seq,
                     table, grad_table, vocab, d, ids, ids_stride, n, dlogits, dlogits_stride, B, dseq, accumulate);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_fm_onehot_grad_f32(const float* dense, int64_t dense_stride, int32_t n_dense, const int32_t* ids,
                                      int64_t ids_stride, int32_t F, const int32_t* vocab, const float* V, int32_t k,
                                      const float* dlogit, int64_t B, float* dw, float* dV, void* stream) {
  const char* who = "rec_fm_onehot_grad_f32";
  REC_CHECK_ARG(ids && vocab && V && dlogit && dw && dV && (dense || n_dense == 0), REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(B >= 0 && F >= 1 && F <= REC_MAX_TABLES && n_dense >= 0 && n_dense + F <= 64 && k >= 1, REC_ESHAPE,
                "%s: needs n_dense + F <= 64, F <= %d", who, REC_MAX_TABLES);
  FmOffsets fo;
  int64_t off = 0;
  for (int f = 0; f < F; ++f) {
    REC_CHECK_ARG(vocab[f] >= 1, REC_ESHAPE, "%s: vocab[%d] < 1", who, f);
    fo.off[f] = (int32_t)off;
    fo.vocab[f] = vocab[f];
    off += vocab[f];
    REC_CHECK_ARG(off + n_dense < 0x7fffffff, REC_ESHAPE, "%s: feature length overflows int32", who);
  }
  if (B == 0) return REC_OK;
  hipLaunchKernelGGL(fm_onehot_grad_kernel, dim3(blocks_of(B, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dense,
                     dense_stride, n_dense, ids, ids_stride, F, fo, V, k, (int64_t)n_dense + off, dlogit, B, dw, dV);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_dropout_f32(const float* x, int64_t n, float rate, uint64_t seed, float* y, void* stream) {
  const char* who = "rec_dropout_f32";
  REC_CHECK_ARG(x && y, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(n >= 0 && rate >= 0.f && rate < 1.f, REC_ESHAPE, "%s: rate must be in [0, 1)", who);
  if (n == 0) return REC_OK;
  const double t = (double)rate * 4294967296.0;
  const uint32_t thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n,
                     thresh, 1.f / (1.f - rate), seed, y);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_wgrad_small_workspace_bytes(int64_t M, int32_t K, int32_t N) {
  int64_t chunks = (M + kWgSpan - 1) / kWgSpan;
  chunks = chunks > 0 ? chunks : 1;
  // the partials, then the workspace of the deterministic column sum that finishes them
  return chunks * (int64_t)K * N * (int64_t)sizeof(float) + rec_colsum_workspace_bytes(chunks, (int64_t)K * N);
}

extern "C" int rec_wgrad_small_f32(const float* x, int64_t x_stride, const float* dy, int64_t dy_stride, int64_t M, int32_t K,
                                   int32_t N, float* out, void* workspace, void* stream) {
  const char* who = "rec_wgrad_small_f32";
  REC_CHECK_ARG(x && dy && out && workspace, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(M >= 1 && K >= 1 && N >= 1 && x_stride >= K && dy_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(N <= 256 && K + N <= kWgLdsFloats && (K + 256 / N - 1) / (256 / N) <= 64, REC_ENOTIMPL,
                "%s: needs N <= 256 and at most 64 outputs per thread (K = %d, N = %d): use rec_dense_f32 on the transposed "
                "operand", who, K, N);
  int slab = kWgLdsFloats / (K + N);
  slab = slab > kWgSlab ? kWgSlab : slab;
  const int J = (K + 256 / N - 1) / (256 / N);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t chunks = (M + kWgSpan - 1) / kWgSpan;
  REC_CHECK_ARG(chunks <= 0x7fffffffLL, REC_ESHAPE, "%s: too many rows", who);
  float* part = reinterpret_cast<float*>(workspace);
  const size_t lds = (size_t)slab * (K + N) * sizeof(float);
  const bool vec = J % 4 == 0 && K % 4 == 0;
#define REC_WG(JM_, V_)                                                                                                   \
  hipLaunchKernelGGL((wgrad_small_partial_kernel<JM_, V_>), dim3((unsigned)chunks), dim3(256), lds, st, x, x_stride, dy, \
                     dy_stride, M, K, N, J, slab, part)
  if (J <= 4) { if (vec) REC_WG(4, true); else REC_WG(4, false); }
  else if (J <= 16) { if (vec) REC_WG(16, true); else REC_WG(16, false); }
  else { if (vec) REC_WG(64, true); else REC_WG(64, false); }
#undef REC_WG
  REC_CHECK_LAUNCH(who);
  const int64_t P = (int64_t)K * N;
  // out[p] = sum over the chunks, fixed order, fp64 across 256-chunk groups (rec_colsum_f32 over the (chunks, P) partials)
  return rec_colsum_f32(part, P, nullptr, 0, nullptr, chunks, P, out, part + chunks * P, stream);
}

extern "C" int rec_bce_prob_grad_f32(const float* y_true, const float* p, int64_t n, float scale, float* dp, void* stream) {
  const char* who = "rec_bce_prob_grad_f32";
  REC_CHECK_ARG(n >= 0, REC_ESHAPE, "%s: bad shape", who);
  if (n == 0) return REC_OK;
  REC_CHECK_ARG(y_true && p && dp, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(bce_prob_grad_kernel, dim3(blocks_of(n, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), y_true, p,
                     n, scale, dp);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}
