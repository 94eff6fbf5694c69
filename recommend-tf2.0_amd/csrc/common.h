// Shared host/device helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "recamd.h"

namespace rec {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

// Kernel selection is by SHAPE only: the library reads no environment variable.  Tests and A/B scripts can force a
// variant with rec_debug_force(key, value) (include/recamd.h: process-global, not for production); forced(key) is the
// forced value, or NULL.  Keys: "dense" b|f|s|t, "dense_pipe" 0|s|d|h, "mha" f|v, "mha_ctr" b, "din" l|s, "cross" l,
// "cross_bpc" n, "pairdot" v, "topk" f, "autoint_wg" n.
const char* forced(const char* key);

#define REC_CHECK_ARG(cond, code, ...)        \
  do {                                        \
    if (!(cond)) {                            \
      ::rec::set_error(__VA_ARGS__);          \
      return (code);                          \
    }                                         \
  } while (0)

// after a kernel launch: hipGetLastError only reports launch-time failures, it does not sync.
#define REC_CHECK_LAUNCH(name)                                                     \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      ::rec::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return REC_EHIP;                                                             \
    }                                                                              \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Per-launch table descriptors, passed BY VALUE in the kernarg segment (<= 1.5 KiB): no device
// side descriptor buffer, no allocation, graph-capturable.  Divergent indexing of a kernarg
// array compiles to a plain global_load from the kernarg segment (L1/K$-resident).
struct TableSet {
  const float* base[REC_MAX_TABLES];
  int32_t vocab[REC_MAX_TABLES];
  int32_t dim[REC_MAX_TABLES];
  int32_t out_col[REC_MAX_TABLES];
};

#ifdef __HIPCC__
// tf.nn.relu propagates NaN (relu(NaN) = NaN); fmaxf(NaN, 0) would return 0 and hide a poisoned input
__device__ __forceinline__ float relu_nan(float x) { return (x > 0.f || x != x) ? x : 0.f; }

__device__ __forceinline__ float act_apply(float x, int act, float alpha) {
  switch (act) {
    case REC_ACT_RELU: return relu_nan(x);
    case REC_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    case REC_ACT_TANH: return tanhf(x);
    case REC_ACT_PRELU: return x >= 0.f ? x : alpha * x;
    default: return x;
  }
}

// id fetch with the Keras Embedding cast: float ids truncate toward zero (tf.cast -> int32);
// NaN/inf map to -1 (treated as out of range instead of UB).
template <int IDS_F32>
__device__ __forceinline__ int32_t load_id(const void* ids, int64_t idx) {
  if (IDS_F32) {
    float f = reinterpret_cast<const float*>(ids)[idx];
    return (f > -2147483648.f && f < 2147483648.f) ? (int32_t)f : -1;
  }
  return reinterpret_cast<const int32_t*>(ids)[idx];
}

// 16-B piece of a table row that a launch reads exactly once.  NT = streaming (nt) policy: tools/exp/rowsize_ceiling.hip
// measured uniformly random rows of 128 B - 1 KiB at 6.0-6.1 TB/s with the default policy and 6.6-6.8 TB/s with nt,
// whatever the row size.  In the fused kernels (tools/exp/rows_nt_ab.sh, same box): SASRec one-launch 94.0 -> 89.3 us
// with nt (shipped), DIN pooling 67.4 -> 71.4 us (its three 256-B pieces per slot keep the default policy).
typedef float rec_f32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ rec_f32x4_t row_load(const rec_f32x4_t* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

}  // namespace rec
