// K8 on the bf16 matrix cores with fp32 accuracy ("bf16x3", see csrc/dense_bf16x3.hip for the split): the
// row-masked attention of src/match/layers/modules.py:76-131 (SASRec), flash-style.
//
// One workgroup (8 waves, or 4 for <= 4 query tiles) per (sample, head); a wave owns one 32-query tile per round
// and all waves walk the head's keys together in 32-key tiles that are STREAMED through LDS (double-buffered), so
// LDS use is independent of the sequence length (the fp32 kernel keeps the whole K and V of the head resident).
//
// Orientation as in the fp32 kernel (attention.hip): scores are computed transposed, S^T = K Q^T, so a lane owns
// one query (column) and 16 of the tile's 32 keys in its accumulator; the softmax is register-local plus one
// cross-half exchange, and the probabilities feed O^T = V^T P^T as the B operand straight from the accumulator
// registers: registers 8s..8s+7 are k-step s, element j of lane half h being key 16s + 8(j>>2) + 4h + (j&3) of the
// tile (cdna guide §3).  The A operand (V^T) is staged in exactly that key order.
//
// Every fp32 operand (Q, K, V and the probabilities) is split exactly into three bf16 terms and each product is
// rebuilt from six v_mfma_f32_32x32x16_bf16 (hh, hm, mh, hl, lh, mm): fp32 accuracy at 2.7x the fp32 MFMA peak.
// (Tried: 4-wave workgroups per group of 4 query tiles so that two share a CU: 0.83 ms against 0.79 ms at config 5 --
// the second pass over K/V and its splits cost more than the overlap wins; forcing 128 VGPRs (30 spilled) to fit two
// 8-wave workgroups per CU: 0.90 ms.)
// Staging per key tile: K rows are split and written as MFMA fragments [plane][k-step][half][key]; V rows are split
// the same way and scattered as 16-bit elements into the transposed fragments [plane][k-step][half][d] (element =
// key), so one barrier per key tile suffices.
#include <math.h>

#include "bf16x3.h"
#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace ab3 {
using bf16x3::split8;
__device__ __forceinline__ f32x16 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), am = __builtin_bit_cast(bf16x8, a[1]),
               al = __builtin_bit_cast(bf16x8, a[2]);
  const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[1]),
               bl = __builtin_bit_cast(bf16x8, b[2]);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  return c;
}
}  // namespace ab3

template <int DK>
__global__ __launch_bounds__(512, 2) void mha_rowmask_b3_kernel(const float* __restrict__ q,
                                                                const float* __restrict__ k,
                                                                const float* __restrict__ v,
                                                                const float* __restrict__ mask, int Sq, int Sk, int H,
                                                                float* __restrict__ out, int64_t qs, int64_t ks_,
                                                                int64_t vs) {
  using namespace ab3;
  constexpr int NKS = DK / 16;  // k-steps of QK^T
  constexpr int NDT = DK / 32;  // 32-wide d tiles of the output
  constexpr int NCH = DK / 8;   // 8-float chunks per row
  __shared__ u32x4 Kf[2][3][NKS][2][32];   // [stage][plane][k-step][half][key]
  __shared__ u32x4 Vf[2][3][2][2][DK];     // [stage][plane][k-step][half][d]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwv = blockDim.x >> 6;
  const int ql = lane & 31, hf = lane >> 5;
  const int h = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int dm = H * DK;
  // qs / ks_ / vs: row strides (floats) of q / k / v -- dm when contiguous, larger for views of a fused projection
  const float* kb = k + b * (int64_t)Sk * ks_ + h * DK;
  const float* vb = v + b * (int64_t)Sk * vs + h * DK;
  const int nkt = (Sk + 31) >> 5;
  const int nqt = (Sq + 31) >> 5;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)DK);

  // staging roles: threads [0, 32*NCH) stage K, the next 32*NCH stage V rows; all of them (row = t & 31, chunk)
  const int s_role = tid / (32 * NCH);          // 0 = K, 1 = V, >= 2 idle in staging
  const int s_t = tid - s_role * (32 * NCH);
  const int s_key = s_t & 31, s_chunk = s_t >> 5;

  float sreg[8];
  auto gload = [&](int kt) {
    const int key = kt * 32 + s_key;
    if (s_role < 2) {
      if (key < Sk) {
        const float* p = (s_role == 0 ? kb + (int64_t)key * ks_ : vb + (int64_t)key * vs) + s_chunk * 8;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(p), a1 = *reinterpret_cast<const f32x4*>(p + 4);
        sreg[0] = a0.x, sreg[1] = a0.y, sreg[2] = a0.z, sreg[3] = a0.w;
        sreg[4] = a1.x, sreg[5] = a1.y, sreg[6] = a1.z, sreg[7] = a1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) sreg[j] = 0.f;  // pad keys: 0 * V stays 0, their logits are masked to -inf
      }
    }
  };
  auto stage_write = [&](int st) {  // K fragments, V^T fragments
    if (s_role == 0) {
      u32x4 hh, mm, ll;
      split8(sreg, hh, mm, ll);
      Kf[st][0][s_chunk >> 1][s_chunk & 1][s_key] = hh;
      Kf[st][1][s_chunk >> 1][s_chunk & 1][s_key] = mm;
      Kf[st][2][s_chunk >> 1][s_chunk & 1][s_key] = ll;
    } else if (s_role == 1) {
      // V^T fragments: this thread holds 8 feature columns of ONE key; element (key) j of the fragment of column d
      // is a 16-bit slot, key -> (k-step, half, j) by the accumulator-order map 16 ks + 8 (j >> 2) + 4 half + (j & 3)
      u32x4 hh, mm, ll;
      split8(sreg, hh, mm, ll);
      const int ks = s_key >> 4, rem = s_key & 15;
      const int vhf = (rem >> 2) & 1, j = ((rem >> 3) << 2) | (rem & 3);
      uint16_t* vh = reinterpret_cast<uint16_t*>(&Vf[st][0][ks][vhf][s_chunk * 8]) + j;
      uint16_t* vm = reinterpret_cast<uint16_t*>(&Vf[st][1][ks][vhf][s_chunk * 8]) + j;
      uint16_t* vl = reinterpret_cast<uint16_t*>(&Vf[st][2][ks][vhf][s_chunk * 8]) + j;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int sh = 16 * (e & 1);
        vh[e * 8] = (uint16_t)(hh[e >> 1] >> sh);
        vm[e * 8] = (uint16_t)(mm[e >> 1] >> sh);
        vl[e * 8] = (uint16_t)(ll[e >> 1] >> sh);
      }
    }
  };

  for (int q0 = 0; q0 < nqt; q0 += nwv) {
    const int qt = q0 + wv;
    const bool active = qt < nqt;  // wave-uniform
    const int qi = qt * 32 + ql;
    const int qc = active ? (qi < Sq ? qi : Sq - 1) : 0;
    // Q operand fragments: lane (query, half) holds Q[query][16 s + 8 half .. +8]
    u32x4 qf[NKS][3];
    {
      const float* qp = q + (b * Sq + qc) * qs + h * DK + hf * 8;
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(qp + 16 * s), a1 = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
        const float x[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        split8(x, qf[s][0], qf[s][1], qf[s][2]);
      }
    }
    const bool masked = mask[b * Sq + qc] == 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

    __syncthreads();  // previous round's last tile fully consumed before its stage is overwritten
    gload(0);
    stage_write(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < nkt) gload(kt + 1);
      if (active) {
        // ---- S^T tile = K_tile . Q_tile^T
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const u32x4 kf[3] = {Kf[st][0][ks][hf][ql], Kf[st][1][ks][hf][ql], Kf[st][2][ks][hf][ql]};
          s = mfma6(kf, qf[ks], s);
        }
        // ---- online softmax for this lane's query over its 16 keys (+ the other half's 16)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          float x = masked ? 0.f : s[r] * scale_log2e;  // masked query row: every logit equal
          x = key < Sk ? x : -INFINITY;                  // pad keys
          s[r] = x;
          tmax = fmaxf(tmax, x);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float resc = exp2f(m_run - m_new);  // first tile: exp2(-inf) = 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = exp2f(s[r] - m_new);
          s[r] = p;
          psum += p;
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * resc + psum;
        m_run = m_new;
        // ---- O^T += V_tile^T . P^T: the probabilities of registers 8 ks .. 8 ks + 7 are k-step ks of the B operand
        u32x4 pf[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const float x[8] = {s[8 * ks], s[8 * ks + 1], s[8 * ks + 2], s[8 * ks + 3],
                              s[8 * ks + 4], s[8 * ks + 5], s[8 * ks + 6], s[8 * ks + 7]};
          split8(x, pf[ks][0], pf[ks][1], pf[ks][2]);
        }
#pragma unroll
        for (int t = 0; t < NDT; ++t) {
#pragma unroll
          for (int r = 0; r < 16; ++r) o[t][r] *= resc;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const u32x4 vf[3] = {Vf[st][0][ks][hf][t * 32 + ql], Vf[st][1][ks][hf][t * 32 + ql],
                                 Vf[st][2][ks][hf][t * 32 + ql]};
            o[t] = mfma6(vf, pf[ks], o[t]);
          }
        }
      }
      if (kt + 1 < nkt) stage_write(st ^ 1);  // stage st^1 was last read in iteration kt-1 (one barrier ago)
      __syncthreads();
    }
    // ---- write O: lane = query, register r of tile t = column t*32 + (r&3) + 8(r>>2) + 4 hf
    if (active && qi < Sq) {
      const float inv = 1.f / l_run;
      float* orow = out + (b * Sq + qi) * (int64_t)dm + h * DK;
#pragma unroll
      for (int t = 0; t < NDT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 w = {o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv};
          *reinterpret_cast<f32x4*>(orow + t * 32 + 8 * g + 4 * hf) = w;
        }
    }
  }
}

// q, k, v, out 16-B aligned with dm % 4 == 0 is the caller's precondition (checked in rec_mha_rowmask_f32)
bool mha_rowmask_b3_dispatch(const float* q, const float* k, const float* v, const float* mask, int64_t B, int Sq,
                             int Sk, int dk, int H, float* out, int64_t qs, int64_t ks, int64_t vs, hipStream_t st) {
  if (B > 65535 || H > 65535 || Sq < 16) return false;
  const int nqt = (Sq + 31) / 32;
  const dim3 grid((unsigned)H, (unsigned)B);
  if (dk == 64) {
    // staging needs 2 * 32 * 8 = 512 threads for K + V rows
    hipLaunchKernelGGL((mha_rowmask_b3_kernel<64>), grid, dim3(512), 0, st, q, k, v, mask, Sq, Sk, H, out, qs, ks, vs);
    return true;
  }
  if (dk == 32) {
    hipLaunchKernelGGL((mha_rowmask_b3_kernel<32>), grid, dim3(nqt > 4 ? 512 : 256), 0, st, q, k, v, mask, Sq, Sk, H,
                       out, qs, ks, vs);
    return true;
  }
  return false;
}

}  // namespace rec
