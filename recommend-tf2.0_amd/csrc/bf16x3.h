// Exact three-term bf16 split of fp32 values ("bf16x3"), shared by the matrix-core kernels
// (dense_bf16x3.hip, attention_b3.hip, attention_ctr_b3.hip, topk.hip, pairwise_dot_gram.hip).
//
//   x = h + m + l,  h = x with the low 16 mantissa bits cleared, m = (x - h) likewise, l = x - h - m
// Every step is exact in fp32 (8 + 8 + 8 significand bits), each term is representable in bf16, and the product of two
// bf16 values is exact in an fp32 accumulator, so  a*b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh) + am*bm + O(2^-24|ab|):
// six v_mfma_*_bf16 per fp32-accurate product, 16/6 = 2.7x the fp32 MFMA rate.  Non-finite inputs: +-inf gives NaN
// (inf - inf), NaN stays NaN; fp32-denormal low parts flush.
#ifndef RECAMD_BF16X3_H_
#define RECAMD_BF16X3_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rec {
namespace bf16x3 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fbits(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ float bfloat(uint32_t u) { return __builtin_bit_cast(float, u); }
// one dword of two bf16: low half = top 16 bits of lo, high half = top 16 bits of hi
__device__ __forceinline__ uint32_t pack_top16(uint32_t lo, uint32_t hi) {
  return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

// NV (<= 8) fp32 values -> three bf16x8 fragments (h, m, l); elements NV..7 are zero
template <int NV = 8>
__device__ __forceinline__ void split(const float* x, u32x4& h, u32x4& m, u32x4& l) {
  uint32_t hb[8], mb[8], lb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < NV) {
      const float hf = bfloat(fbits(x[j]) & 0xffff0000u);
      const float r = x[j] - hf;  // exact
      const float mf = bfloat(fbits(r) & 0xffff0000u);
      hb[j] = fbits(x[j]);
      mb[j] = fbits(r);
      lb[j] = fbits(r - mf);      // exact, <= 8 significant bits
    } else {
      hb[j] = mb[j] = lb[j] = 0u;
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h[t] = pack_top16(hb[2 * t], hb[2 * t + 1]);
    m[t] = pack_top16(mb[2 * t], mb[2 * t + 1]);
    l[t] = pack_top16(lb[2 * t], lb[2 * t + 1]);
  }
}

__device__ __forceinline__ void split8(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) { split<8>(x, h, m, l); }

}  // namespace bf16x3
}  // namespace rec
#endif  // RECAMD_BF16X3_H_
