// P1 — the step BEFORE the path, on the device: what the reference's host-side ETL hands the models,
// applied per batch to raw columns that arrive over PCIe.
//
//   rec_label_encode_u32     sklearn LabelEncoder.transform of src/ctr/utils/data_process.py:66-68: id = rank of the
//                            token in the column's sorted vocabulary (fit = sorted unique values, done once on the host).
//                            Criteo's categorical tokens are 8-digit hex strings; as uint32 they sort exactly like the
//                            strings, the missing-value token "-1" (fillna('-1'), :63) sorts before every digit -> the
//                            caller reserves REC_TOKEN_MISSING for it and it ranks first when present.
//   rec_hash_ids_u32         the production alternative when no vocabulary is kept: id = mix32(token ^ seed_f) mod V_f
//                            (not in the reference; "id remap/hash" of SURVEY §8f-3).
//   rec_minmax_fit_f32 / rec_minmax_scale_f32   MinMaxScaler of :76-78 in its intended per-column form on astype(int)
//                            values: (trunc(x) - min) / (max - min), constant columns -> 0 (sklearn's scale guard).
//                            (As written the reference assigns the (n, 13) result of fit_transform to ONE column,
//                            which pandas rejects)
//   rec_pad_sequences_i32    tf.keras pad_sequences(hist, maxlen) of src/match/utils/data_process.py:138 on a ragged
//                            (values, offsets) batch: pre-padding with `value`, pre-truncating (the LAST maxlen items).
// All HBM/PCIe-bound elementwise or row passes; ids are bit-exact integers.
#include <float.h>

#include "common.h"

namespace rec {

__device__ __forceinline__ uint32_t mix32(uint32_t h) {  // murmur3 finaliser
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

struct VocabSet {
  const uint32_t* vocab[REC_MAX_TABLES];
  int32_t size[REC_MAX_TABLES];
};

// sort key: the missing token ranks before everything else (string order of "-1" vs hex digits)
__device__ __forceinline__ uint64_t tok_key(uint32_t t) {
  return t == 0xffffffffu /* REC_TOKEN_MISSING */ ? 0ull : (uint64_t)t + 1ull;
}

__global__ __launch_bounds__(256) void label_encode_kernel(VocabSet vs, const uint32_t* __restrict__ tok,
                                                           int64_t tok_stride, int F, int64_t n, int32_t* __restrict__ ids,
                                                           int64_t ids_stride, int* __restrict__ unseen) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / F;
  const int f = (int)(i - b * F);
  const uint64_t key = tok_key(tok[b * tok_stride + f]);
  const uint32_t* v = vs.vocab[f];
  int lo = 0, hi = vs.size[f];
  while (lo < hi) {  // lower bound in the sorted vocabulary
    const int mid = (lo + hi) >> 1;
    if (tok_key(v[mid]) < key) lo = mid + 1; else hi = mid;
  }
  int32_t id = -1;
  if (lo < vs.size[f] && tok_key(v[lo]) == key) id = lo;
  else if (unseen) *unseen = 1;  // sklearn raises "y contains previously unseen labels"
  ids[b * ids_stride + f] = id;
}

__global__ __launch_bounds__(256) void hash_ids_kernel(const uint32_t* __restrict__ tok, int64_t tok_stride, int F, int64_t n,
                                                       VocabSet sizes, uint32_t seed, int32_t* __restrict__ ids,
                                                       int64_t ids_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / F;
  const int f = (int)(i - b * F);
  const uint32_t h = mix32(tok[b * tok_stride + f] ^ mix32(seed + 0x9e3779b9u * (uint32_t)(f + 1)));
  ids[b * ids_stride + f] = (int32_t)(h % (uint32_t)sizes.size[f]);
}

// column min / max of trunc(x): partial per block of 256 rows, then a fixed-order finish (deterministic)
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t xs, int64_t M, int N,
                                                             int trunc_int, float* __restrict__ pmin,
                                                             float* __restrict__ pmax) {
  __shared__ float smin[4][64], smax[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + tx;
  const int64_t m0 = (int64_t)blockIdx.y * 256;
  float lo = FLT_MAX, hi = -FLT_MAX;
  if (n < N)
    for (int r = ty; r < 256; r += 4) {
      const int64_t m = m0 + r;
      if (m >= M) break;
      float v = x[m * xs + n];
      if (trunc_int) v = truncf(v);
      lo = fminf(lo, v);
      hi = fmaxf(hi, v);
    }
  smin[ty][tx] = lo;
  smax[ty][tx] = hi;
  __syncthreads();
  if (ty == 0 && n < N) {
    pmin[(int64_t)blockIdx.y * N + n] = fminf(fminf(smin[0][tx], smin[1][tx]), fminf(smin[2][tx], smin[3][tx]));
    pmax[(int64_t)blockIdx.y * N + n] = fmaxf(fmaxf(smax[0][tx], smax[1][tx]), fmaxf(smax[2][tx], smax[3][tx]));
  }
}
__global__ __launch_bounds__(256) void minmax_finish_kernel(const float* __restrict__ pmin, const float* __restrict__ pmax,
                                                            int64_t chunks, int N, float* __restrict__ mn,
                                                            float* __restrict__ mx) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int64_t c = 0; c < chunks; ++c) {
    lo = fminf(lo, pmin[c * N + n]);
    hi = fmaxf(hi, pmax[c * N + n]);
  }
  mn[n] = lo;
  mx[n] = hi;
}

__global__ __launch_bounds__(256) void minmax_scale_kernel(const float* __restrict__ x, int64_t xs, int64_t M, int N,
                                                           const float* __restrict__ mn, const float* __restrict__ mx,
                                                           int trunc_int, float* __restrict__ out, int64_t os) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const int64_t m = i / N;
  const int n = (int)(i - m * N);
  float v = x[m * xs + n];
  if (trunc_int) v = truncf(v);
  // sklearn MinMaxScaler (feature_range (0,1)) works in float64 on the integer-valued column and the train scripts
  // cast the result to float32 (data_process.py:86): scale = 1 / (max - min) (constant columns keep scale 1),
  // X * scale + (0 - min * scale) — the same operations in fp64 here, rounded to fp32 once (HBM-bound: free)
  const double range = (double)mx[n] - (double)mn[n];
  const double scale = range == 0.0 ? 1.0 : 1.0 / range;
  // no fused multiply-add (hipcc contracts a*b+c by default): separate roundings, so the fp64 result is sklearn's bit
  // for bit (and the column minimum maps to exactly 0); see the pragma at the top of this function
  const double min_ = 0.0 - (double)mn[n] * scale;
  const double prod = (double)v * scale;
  out[m * os + n] = (float)(prod + min_);
}

// out[b, :] = pre-padded / pre-truncated row b of a ragged int32 batch
__global__ __launch_bounds__(256) void pad_sequences_kernel(const int32_t* __restrict__ values,
                                                            const int64_t* __restrict__ offsets, int64_t B, int maxlen,
                                                            int32_t pad, int pre_pad, int pre_trunc,
                                                            int32_t* __restrict__ out, int64_t os) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * maxlen) return;
  const int64_t b = i / maxlen;
  const int t = (int)(i - b * maxlen);
  const int64_t lo = offsets[b], len = offsets[b + 1] - lo;
  const int64_t keep = len < maxlen ? len : maxlen;
  const int64_t src0 = pre_trunc ? lo + (len - keep) : lo;          // which `keep` items survive
  const int64_t dst0 = pre_pad ? maxlen - keep : 0;                  // where they land
  int32_t v = pad;
  if (t >= dst0 && t < dst0 + keep) v = values[src0 + (t - dst0)];
  out[b * os + t] = v;
}

}  // namespace rec

using namespace rec;

static int fill_vocab(const uint32_t* const* vocabs, const int32_t* sizes, int F, VocabSet* vs, const char* who,
                      bool need_ptr) {
  REC_CHECK_ARG(F >= 1 && F <= REC_MAX_TABLES && sizes && (!need_ptr || vocabs), REC_ESHAPE, "%s: F=%d", who, F);
  for (int f = 0; f < REC_MAX_TABLES; ++f) {
    vs->vocab[f] = (need_ptr && f < F) ? vocabs[f] : nullptr;
    vs->size[f] = f < F ? sizes[f] : 1;
    if (f < F) {
      REC_CHECK_ARG(sizes[f] >= 1, REC_ESHAPE, "%s: sizes[%d]=%d", who, f, sizes[f]);
      REC_CHECK_ARG(!need_ptr || vocabs[f], REC_EINVAL, "%s: vocabs[%d] is NULL", who, f);
    }
  }
  return REC_OK;
}

extern "C" int rec_label_encode_u32(const uint32_t* const* vocabs, const int32_t* vocab_sizes, int32_t F,
                                    const uint32_t* tokens, int64_t tok_stride, int64_t B, int32_t* ids,
                                    int64_t ids_stride, int32_t* unseen_flag, void* stream) {
  const char* who = "rec_label_encode_u32";
  VocabSet vs;
  int rc = fill_vocab(vocabs, vocab_sizes, F, &vs, who, true);
  if (rc != REC_OK) return rc;
  REC_CHECK_ARG(B >= 0 && tok_stride >= F && ids_stride >= F, REC_ESHAPE, "%s: bad shape", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(tokens && ids, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t n = B * F;
  hipLaunchKernelGGL(label_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vs, tokens,
                     tok_stride, F, n, ids, ids_stride, unseen_flag);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_hash_ids_u32(const uint32_t* tokens, int64_t tok_stride, const int32_t* vocab_sizes, int32_t F,
                                int64_t B, uint32_t seed, int32_t* ids, int64_t ids_stride, void* stream) {
  const char* who = "rec_hash_ids_u32";
  VocabSet vs;
  int rc = fill_vocab(nullptr, vocab_sizes, F, &vs, who, false);
  if (rc != REC_OK) return rc;
  REC_CHECK_ARG(B >= 0 && tok_stride >= F && ids_stride >= F, REC_ESHAPE, "%s: bad shape", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(tokens && ids, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t n = B * F;
  hipLaunchKernelGGL(hash_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tokens,
                     tok_stride, F, n, vs, seed, ids, ids_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int64_t rec_minmax_workspace_bytes(int64_t M, int32_t N) {
  if (M < 0 || N < 0) return 0;
  const int64_t chunks = (M + 255) / 256;
  return (int64_t)sizeof(float) * 2 * (chunks > 0 ? chunks : 1) * (N > 0 ? N : 1);
}

extern "C" int rec_minmax_fit_f32(const float* x, int64_t x_stride, int64_t M, int32_t N, int32_t truncate_to_int,
                                  float* col_min, float* col_max, void* workspace, void* stream) {
  const char* who = "rec_minmax_fit_f32";
  REC_CHECK_ARG(M >= 1 && N >= 1 && x_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(x && col_min && col_max && workspace, REC_EINVAL, "%s: NULL pointer", who);
  const int64_t chunks = (M + 255) / 256;
  REC_CHECK_ARG(chunks <= 65535, REC_ESHAPE, "%s: too many rows per call (fit in slices and combine)", who);
  float* pmin = static_cast<float*>(workspace);
  float* pmax = pmin + chunks * N;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)chunks), dim3(256), 0, st, x,
                     x_stride, M, N, truncate_to_int, pmin, pmax);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(minmax_finish_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, pmin, pmax, chunks, N,
                     col_min, col_max);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_minmax_scale_f32(const float* x, int64_t x_stride, int64_t M, int32_t N, const float* col_min,
                                    const float* col_max, int32_t truncate_to_int, float* out, int64_t out_stride,
                                    void* stream) {
  const char* who = "rec_minmax_scale_f32";
  REC_CHECK_ARG(M >= 0 && N >= 1 && x_stride >= N && out_stride >= N, REC_ESHAPE, "%s: bad shape", who);
  if (M == 0) return REC_OK;
  REC_CHECK_ARG(x && col_min && col_max && out, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(minmax_scale_kernel, dim3((unsigned)((M * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     x_stride, M, N, col_min, col_max, truncate_to_int, out, out_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_pad_sequences_i32(const int32_t* values, const int64_t* offsets, int64_t B, int32_t maxlen,
                                     int32_t pad_value, int32_t pre_padding, int32_t pre_truncating, int32_t* out,
                                     int64_t out_stride, void* stream) {
  const char* who = "rec_pad_sequences_i32";
  REC_CHECK_ARG(B >= 0 && maxlen >= 1 && out_stride >= maxlen, REC_ESHAPE, "%s: bad shape", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(offsets && out, REC_EINVAL, "%s: NULL pointer", who);
  hipLaunchKernelGGL(pad_sequences_kernel, dim3((unsigned)((B * maxlen + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     values, offsets, B, maxlen, pad_value, pre_padding, pre_truncating, out, out_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}
