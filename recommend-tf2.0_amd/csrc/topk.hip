// §8f-4 — the retrieval step after the two towers: exact inner-product top-k, the job
// faiss.IndexFlatIP(d).add(items).search(users, 10) does in src/match/dssm/dssm_train.py:74-78 and
// src/match/fm/train.py:71-75.  GEMM-shaped (Q x N x d), so it runs on the fp32 matrix cores; the Q x N score
// matrix is never written to HBM.
//
// A workgroup owns 128 queries and streams all N items through 128-item tiles:
//   * the query tile is loaded ONCE into registers in LDS-staging layout (d/2 floats per thread);
//   * per item tile: 16-wide k-steps through LDS; wave w computes query rows 32w..32w+31 against all 128 items
//     (1 x 4 v_mfma_f32_32x32x2_f32 tiles), so a row's 128 scores sit in the accumulators of ONE half-wave
//     (column on the lane, row in the register) and never leave registers;
//   * per accumulator register (= one query row per half-wave) the 4 scores of a lane are compared with the row's
//     current k-th best: one ballot, and after the first few tiles almost every row is skipped;
//   * rows with a candidate merge {running top-k, 128 new scores} by k rounds of half-wave arg-max
//     ((score, index) pairs, ties -> smaller index, so the result is deterministic).
// The running lists (k <= 32 per row) live in LDS (32 KiB) and are written once at the end, scores descending;
// 49 KiB of LDS per workgroup leaves room for 3 workgroups per CU, which is what hides the item-tile loads.
#include <math.h>

#include "bf16x3.h"
#include "common.h"

namespace rec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace topk {
constexpr int BM = 128, BN = 128, BK = 16, KMAX = 32;
constexpr int LDA = BM + 4, LDB = BN + 4;
constexpr int DMAX = 128;
// LDS: As, Bs, running scores, running indices
constexpr size_t lds_bytes() {
  return (size_t)(BK * LDA + BK * LDB + BM * KMAX) * sizeof(float) + (size_t)BM * KMAX * sizeof(int);
}
}  // namespace topk

// a "better" than b: larger score, ties -> smaller index; index -1 (empty) always loses
__device__ __forceinline__ bool better(float sa, int ia, float sb, int ib) {
  return sa > sb || (sa == sb && (unsigned)ia < (unsigned)ib);
}

template <int KS>  // k-steps of 16: d <= 16 * KS
__global__ __launch_bounds__(256, (KS <= 2 ? 3 : 2)) void topk_ip_kernel(const float* __restrict__ q, int64_t q_stride,
                                                      const float* __restrict__ items, int64_t items_stride,
                                                      int64_t Q, int N, int d, int k, float* __restrict__ out_scores,
                                                      int64_t* __restrict__ out_idx) {
  using namespace topk;
  __shared__ __attribute__((aligned(16))) float As[BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
  __shared__ float run_s[BM * KMAX];
  __shared__ int run_i[BM * KMAX];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;

  for (int e = tid; e < BM * KMAX; e += 256) {
    run_s[e] = -INFINITY;
    run_i[e] = -1;
  }

  // query tile in staging layout: thread (row = tid/2, 8 consecutive k of every 16-wide k-step)
  const int a_row = tid >> 1, a_k = (tid & 1) * 8;
  float aq[KS][8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int64_t gm = m0 + a_row;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kk = ks * BK + a_k + e;
      aq[ks][e] = (gm < Q && kk < d) ? q[gm * q_stride + kk] : 0.f;
    }
  }
  __syncthreads();

  for (int n0 = 0; n0 < N; n0 += BN) {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float bv[8];
      {
        const int gn = n0 + a_row;  // item row staged by this thread (same map as the queries)
        const float* pb = items + (int64_t)gn * items_stride + ks * BK + a_k;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (gn < N && ks * BK + a_k + e < d) ? pb[e] : 0.f;
      }
      __syncthreads();  // previous k-step's tiles consumed
#pragma unroll
      for (int e = 0; e < 8; ++e) As[(a_k + e) * LDA + a_row] = aq[ks][e];
#pragma unroll
      for (int e = 0; e < 8; ++e) Bs[(a_k + e) * LDB + a_row] = bv[e];
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        const int kr = kk + half;
        const float a = As[kr * LDA + wv * 32 + l32];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Bs[kr * LDB + j * 32 + l32], acc[j], 0, 0, 0);
      }
    }

    // top-k update in registers: accumulator register r of half-wave `half` is query row
    // wv*32 + (r&3) + 8*(r>>2) + 4*half, item columns n0 + 32 j + l32 (j = 0..3)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const bool live = m0 + row < Q;
      float v[5];
      int id[5];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = n0 + 32 * j + l32;
        v[j] = (live && c < N) ? acc[j][r] : -INFINITY;
        id[j] = (live && c < N) ? c : -1;
      }
      const float thr = run_s[row * KMAX + k - 1];
      const int thr_i = run_i[row * KMAX + k - 1];
      const bool cand = better(v[0], id[0], thr, thr_i) || better(v[1], id[1], thr, thr_i) ||
                        better(v[2], id[2], thr, thr_i) || better(v[3], id[3], thr, thr_i);
      if (!__any(cand)) continue;  // wave-uniform: neither of the two rows of this register has a candidate
      v[4] = l32 < k ? run_s[row * KMAX + l32] : -INFINITY;
      id[4] = l32 < k ? run_i[row * KMAX + l32] : -1;
      float my_s = -INFINITY;
      int my_i = -1;
      for (int t = 0; t < k; ++t) {
        float bs = v[0];
        int bi = id[0];
#pragma unroll
        for (int j = 1; j < 5; ++j)
          if (better(v[j], id[j], bs, bi)) bs = v[j], bi = id[j];
        float ws = bs;
        int wi = bi;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {  // stays inside the 32-lane half
          const float os = __shfl_xor(ws, off);
          const int oi = __shfl_xor(wi, off);
          if (better(os, oi, ws, wi)) ws = os, wi = oi;
        }
        if (l32 == t) my_s = ws, my_i = wi;
        if (wi >= 0 && wi == bi) {  // the owner retires the slot (indices are unique within a row)
#pragma unroll
          for (int j = 0; j < 5; ++j)
            if (id[j] == wi) v[j] = -INFINITY, id[j] = -1;
        }
      }
      if (l32 < k) {
        run_s[row * KMAX + l32] = my_s;
        run_i[row * KMAX + l32] = my_i;
      }
    }
  }
  __syncthreads();

  for (int e = tid; e < BM * k; e += 256) {
    const int row = e / k, t = e - row * k;
    if (m0 + row < Q) {
      out_scores[(m0 + row) * k + t] = run_s[row * KMAX + t];
      out_idx[(m0 + row) * k + t] = (int64_t)run_i[row * KMAX + t];
    }
  }
}


// ---- the same kernel on the bf16 matrix cores with fp32 accuracy (exact 3-term bf16 split, see dense_bf16x3.hip):
// scores = Qh Ih + Qh Im + Qm Ih + Qh Il + Ql Ih + Qm Im.  The query fragments of a wave's 32 rows are loaded
// straight from global memory in MFMA operand layout and stay in registers; an item tile is staged whole (all
// k-steps, three planes) as [k-step][plane][half][row] fragments, prefetched in registers one tile ahead.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4t __attribute__((ext_vector_type(4)));

namespace tb3 {
using bf16x3::split8;
}  // namespace tb3

template <int KS>
__global__ __launch_bounds__(256, 2) void topk_ip_b3_kernel(const float* __restrict__ q, int64_t q_stride,
                                                            const float* __restrict__ items, int64_t items_stride,
                                                            int64_t Q, int N, int d, int k,
                                                            float* __restrict__ out_scores,
                                                            int64_t* __restrict__ out_idx, int vec_ok, int n_chunk,
                                                            float* __restrict__ part_s, int* __restrict__ part_i) {
  // n_chunk > 0 (split-N, few queries): workgroup (x, y) scans items [y * n_chunk, (y+1) * n_chunk) and writes its
  // k best per query to part_s / part_i [y][Q][k]; topk_merge_kernel combines the gridDim.y partial lists.
  using namespace topk;
  using namespace tb3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* Bf = reinterpret_cast<u32x4*>(smem_raw);                       // [KS][3][2][128]
  float* run_s = reinterpret_cast<float*>(Bf + KS * 3 * 2 * 128);       // [BM][KMAX]
  int* run_i = reinterpret_cast<int*>(run_s + BM * KMAX);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;

  for (int e = tid; e < BM * KMAX; e += 256) {
    run_s[e] = -INFINITY;
    run_i[e] = -1;
  }

  auto load8 = [&](const float* base, int64_t stride, int64_t row, bool row_ok, int kb, float (&r)[8]) {
    const float* p = base + (row_ok ? row : 0) * stride + kb;
    if (vec_ok && row_ok && kb + 8 <= d) {
      const f32x4t a0 = *reinterpret_cast<const f32x4t*>(p), a1 = *reinterpret_cast<const f32x4t*>(p + 4);
      r[0] = a0.x, r[1] = a0.y, r[2] = a0.z, r[3] = a0.w;
      r[4] = a1.x, r[5] = a1.y, r[6] = a1.z, r[7] = a1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (row_ok && kb + j < d) ? p[j] : 0.f;
    }
  };

  // A operand: lane (row l32 of the wave's 32 queries, k-half) for every k-step
  u32x4 qf[KS][3];
  {
    const int64_t gm = m0 + wv * 32 + l32;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float x[8];
      load8(q, q_stride, gm, gm < Q, ks * BK + 8 * half, x);
      split8(x, qf[ks][0], qf[ks][1], qf[ks][2]);
    }
  }

  // item staging: thread (row = tid & 127, k-half = tid >> 7), all k-steps of the tile, one tile ahead
  const int srow = tid & 127, skh = tid >> 7;
  float bv[KS][8];
  auto gload = [&](int n0) {
    const int64_t gn = (int64_t)n0 + srow;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) load8(items, items_stride, gn, gn < N, ks * BK + 8 * skh, bv[ks]);
  };
  auto lwrite = [&]() {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      u32x4 h, m, l;
      split8(bv[ks], h, m, l);
      Bf[((ks * 3 + 0) * 2 + skh) * 128 + srow] = h;
      Bf[((ks * 3 + 1) * 2 + skh) * 128 + srow] = m;
      Bf[((ks * 3 + 2) * 2 + skh) * 128 + srow] = l;
    }
  };
  const int n_begin = n_chunk > 0 ? (int)blockIdx.y * n_chunk : 0;
  const int n_end = n_chunk > 0 ? (n_begin + n_chunk < N ? n_begin + n_chunk : N) : N;
  if (n_begin < n_end) gload(n_begin);
  __syncthreads();

  for (int n0 = n_begin; n0 < n_end; n0 += BN) {
    lwrite();
    __syncthreads();
    if (n0 + BN < n_end) gload(n0 + BN);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, qf[ks][0]), am = __builtin_bit_cast(bf16x8, qf[ks][1]),
                   al = __builtin_bit_cast(bf16x8, qf[ks][2]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, Bf[((ks * 3 + 0) * 2 + half) * 128 + j * 32 + l32]);
        const bf16x8 bm = __builtin_bit_cast(bf16x8, Bf[((ks * 3 + 1) * 2 + half) * 128 + j * 32 + l32]);
        const bf16x8 bl = __builtin_bit_cast(bf16x8, Bf[((ks * 3 + 2) * 2 + half) * 128 + j * 32 + l32]);
        f32x16 c = acc[j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
        acc[j] = c;
      }
    }
    __syncthreads();  // every wave has read the fragments; the next lwrite may overwrite them

    // top-k update in registers: accumulator register r of half-wave `half` is query row
    // wv*32 + (r&3) + 8*(r>>2) + 4*half, item columns n0 + 32 j + l32 (j = 0..3)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const bool live = m0 + row < Q;
      float v[5];
      int id[5];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = n0 + 32 * j + l32;
        v[j] = (live && c < N) ? acc[j][r] : -INFINITY;
        id[j] = (live && c < N) ? c : -1;
      }
      const float thr = run_s[row * KMAX + k - 1];
      const int thr_i = run_i[row * KMAX + k - 1];
      const bool cand = better(v[0], id[0], thr, thr_i) || better(v[1], id[1], thr, thr_i) ||
                        better(v[2], id[2], thr, thr_i) || better(v[3], id[3], thr, thr_i);
      if (!__any(cand)) continue;  // wave-uniform: neither of the two rows of this register has a candidate
      v[4] = l32 < k ? run_s[row * KMAX + l32] : -INFINITY;
      id[4] = l32 < k ? run_i[row * KMAX + l32] : -1;
      // Few candidates (the common case once the lists have warmed up): insert them one by one into the sorted list
      // held by lanes 0..k-1 of the half-wave -- position = number of entries that beat the candidate (one ballot),
      // entries behind it move down one lane.  ~25 instructions per candidate instead of k arg-max rounds.
      {
        int ncand = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) ncand += better(v[j], id[j], thr, thr_i) ? 1 : 0;
        const uint64_t anyc = __ballot(ncand > 0);
        const int cntA = __popcll(anyc & 0xffffffffull), cntB = __popcll(anyc >> 32);
        // lanes hold up to 4 candidates each; bound the work by the number of candidate LANES times 4
        if (cntA <= 4 && cntB <= 4) {   // wave-uniform
          float ls = v[4];
          int li = id[4];
          for (int it = 0; it < 16; ++it) {
            bool have = false;
            float cs = -INFINITY;
            int ci = -1, cj = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (id[j] >= 0 && better(v[j], id[j], cs, ci)) cs = v[j], ci = id[j], cj = j, have = true;
            // still a candidate against the CURRENT k-th entry of my half's list?
            const float kth_s = __shfl(ls, (lane & 32) + k - 1);
            const int kth_i = __shfl(li, (lane & 32) + k - 1);
            have = have && better(cs, ci, kth_s, kth_i);
            const uint64_t hm = __ballot(have);
            if (hm == 0) break;         // wave-uniform
            const uint64_t mine = half ? (hm >> 32) : (hm & 0xffffffffull);
            const int src = mine ? (int)__builtin_ctzll(mine) : 0;       // lowest lane of my half with a candidate
            const float xs = __shfl(cs, (lane & 32) + src);
            const int xi = __shfl(ci, (lane & 32) + src);
            const bool active = mine != 0;
            if (active && l32 == src && cj >= 0) {                        // the owner retires that slot
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (j == cj) v[j] = -INFINITY, id[j] = -1;
            }
            // position in the sorted list: entries that beat x
            const uint64_t bm = __ballot(l32 < k && better(ls, li, xs, xi));
            const int pos = __popcll(half ? (bm >> 32) : (bm & 0xffffffffull));
            const float up_s = __shfl_up(ls, 1);
            const int up_i = __shfl_up(li, 1);
            if (active && l32 < k) {
              if (l32 == pos) ls = xs, li = xi;
              else if (l32 > pos) ls = up_s, li = up_i;
            }
            // candidates of this lane that no longer beat the k-th entry drop out.  The k-th entry is read in
            // wave-uniform control flow: a shuffle from an EXEC-disabled source lane returns 0, which used to drop
            // true candidates whose score is <= 0 (lane k-1 often holds no candidate itself).
            {
              const float ks2 = __shfl(ls, (lane & 32) + k - 1);
              const int ki2 = __shfl(li, (lane & 32) + k - 1);
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (id[j] >= 0 && !better(v[j], id[j], ks2, ki2)) v[j] = -INFINITY, id[j] = -1;
            }
          }
          if (l32 < k) {
            run_s[row * KMAX + l32] = ls;
            run_i[row * KMAX + l32] = li;
          }
          continue;
        }
      }
      float my_s = -INFINITY;
      int my_i = -1;
      for (int t = 0; t < k; ++t) {
        float bs = v[0];
        int bi = id[0];
#pragma unroll
        for (int j = 1; j < 5; ++j)
          if (better(v[j], id[j], bs, bi)) bs = v[j], bi = id[j];
        float ws = bs;
        int wi = bi;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) {  // stays inside the 32-lane half
          const float os = __shfl_xor(ws, off);
          const int oi = __shfl_xor(wi, off);
          if (better(os, oi, ws, wi)) ws = os, wi = oi;
        }
        if (l32 == t) my_s = ws, my_i = wi;
        if (wi >= 0 && wi == bi) {  // the owner retires the slot (indices are unique within a row)
#pragma unroll
          for (int j = 0; j < 5; ++j)
            if (id[j] == wi) v[j] = -INFINITY, id[j] = -1;
        }
      }
      if (l32 < k) {
        run_s[row * KMAX + l32] = my_s;
        run_i[row * KMAX + l32] = my_i;
      }
    }
  }
  __syncthreads();

  for (int e = tid; e < BM * k; e += 256) {
    const int row = e / k, t = e - row * k;
    if (m0 + row < Q) {
      if (n_chunk > 0) {
        const int64_t o = ((int64_t)blockIdx.y * Q + m0 + row) * k + t;
        part_s[o] = run_s[row * KMAX + t];
        part_i[o] = run_i[row * KMAX + t];
      } else {
        out_scores[(m0 + row) * k + t] = run_s[row * KMAX + t];
        out_idx[(m0 + row) * k + t] = (int64_t)run_i[row * KMAX + t];
      }
    }
  }
}


// split-N merge: one wave per query, the nsplit * k partial candidates (<= 256) sit 4 per lane, k rounds of wave
// arg-max (ties -> smaller index) write the final list
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ part_s, const int* __restrict__ part_i,
                                                         int nsplit, int64_t Q, int k, float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_idx) {
  const int lane = threadIdx.x & 63;
  const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= Q) return;
  const int total = nsplit * k;
  float v[4];
  int id[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int e = lane + 64 * c;
    if (e < total) {
      const int sp = e / k, t = e - sp * k;
      v[c] = part_s[((int64_t)sp * Q + qi) * k + t];
      id[c] = part_i[((int64_t)sp * Q + qi) * k + t];
    } else {
      v[c] = -INFINITY;
      id[c] = -1;
    }
  }
  for (int t = 0; t < k; ++t) {
    float bs = v[0];
    int bi = id[0];
#pragma unroll
    for (int c = 1; c < 4; ++c)
      if (better(v[c], id[c], bs, bi)) bs = v[c], bi = id[c];
    float ws = bs;
    int wi = bi;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float os = __shfl_xor(ws, off);
      const int oi = __shfl_xor(wi, off);
      if (better(os, oi, ws, wi)) ws = os, wi = oi;
    }
    if (lane == 0) {
      out_scores[qi * k + t] = ws;
      out_idx[qi * k + t] = (int64_t)wi;
    }
    if (wi >= 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (id[c] == wi) v[c] = -INFINITY, id[c] = -1;
    }
  }
}

}  // namespace rec

using namespace rec;

// split-N plan: few query workgroups and many items -> spread the item range over nsplit workgroups per query tile
static int topk_nsplit(int64_t Q, int64_t N, int k) {
  const int64_t qblocks = (Q + topk::BM - 1) / topk::BM;
  if (qblocks >= 256 || N < 2048) return 1;
  int64_t ns = 512 / qblocks;              // aim at ~2 workgroups per CU
  const int64_t by_items = N / 1024;       // at least 1024 items per split
  if (ns > by_items) ns = by_items;
  if (ns > 256 / k) ns = 256 / k;          // merge kernel: nsplit * k <= 256 candidates
  return ns < 2 ? 1 : (int)ns;
}

extern "C" int64_t rec_topk_ip_workspace_bytes(int64_t Q, int64_t N, int32_t k) {
  if (Q < 1 || N < 1 || k < 1 || k > topk::KMAX) return 0;
  const int ns = topk_nsplit(Q, N, k);
  return ns > 1 ? (int64_t)ns * Q * k * 8 : 0;
}

extern "C" int rec_topk_ip_ws_f32(const float* queries, int64_t q_stride, int64_t Q, const float* items,
                                  int64_t items_stride, int64_t N, int32_t d, int32_t k, float* out_scores,
                                  int64_t* out_idx, void* workspace, void* stream) {
  const char* who = "rec_topk_ip_f32";
  // tile offsets (n0 + BN, n_begin + n_chunk, n0 + 32 j + lane) are int arithmetic: keep a tile of headroom below 2^31
  REC_CHECK_ARG(Q >= 0 && N >= 0 && N <= 0x7fffffffLL - 256, REC_ESHAPE, "%s: Q=%lld N=%lld (N must be <= 2^31 - 257)", who, (long long)Q, (long long)N);
  REC_CHECK_ARG(d >= 1 && d <= topk::DMAX, REC_ESHAPE, "%s: d=%d (1..%d)", who, d, topk::DMAX);
  REC_CHECK_ARG(k >= 1 && k <= topk::KMAX, REC_ESHAPE, "%s: k=%d (1..%d)", who, k, topk::KMAX);
  REC_CHECK_ARG(q_stride >= d && items_stride >= d, REC_ESHAPE, "%s: strides smaller than d", who);
  if (Q == 0) return REC_OK;
  REC_CHECK_ARG(queries && out_scores && out_idx && (items || N == 0), REC_EINVAL, "%s: NULL pointer", who);
  const int64_t blocks = (Q + topk::BM - 1) / topk::BM;
  REC_CHECK_ARG(blocks <= 0x7fffffffLL, REC_ESHAPE, "%s: too many queries", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    const char* e = forced("topk");  // "f": fp32-MFMA kernel (rec_debug_force: tests / A/B only)
    if (!(e && e[0] == 'f') && d <= 64) {  // d > 64: the bf16x3 form would spill (96 query + 64 staging VGPRs)
      const int vec_ok = (aligned16(queries) && aligned16(items) && q_stride % 4 == 0 && items_stride % 4 == 0) ? 1 : 0;
      const int nsplit = workspace ? topk_nsplit(Q, N, k) : 1;
      const int n_chunk = nsplit > 1 ? (int)(((N + nsplit - 1) / nsplit + 127) / 128 * 128) : 0;
      float* part_s = static_cast<float*>(workspace);
      int* part_i = nsplit > 1 ? reinterpret_cast<int*>(part_s + (int64_t)nsplit * Q * k) : nullptr;
#define REC_TOPK_B3(KS_)                                                                                          \
  do {                                                                                                            \
    const size_t lds = (size_t)KS_ * 3 * 2 * 128 * 16 + (size_t)topk::BM * topk::KMAX * 8;                        \
    hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void*>(topk_ip_b3_kernel<KS_>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
    REC_CHECK_ARG(he == hipSuccess, REC_EHIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(he));         \
    hipLaunchKernelGGL((topk_ip_b3_kernel<KS_>), dim3((unsigned)blocks, (unsigned)nsplit), dim3(256), lds, st,    \
                       queries, q_stride, items, items_stride, Q, (int)N, d, k, out_scores, out_idx, vec_ok,      \
                       n_chunk, part_s, part_i);                                                                  \
  } while (0)
      if (d <= 16) REC_TOPK_B3(1);
      else if (d <= 32) REC_TOPK_B3(2);
      else REC_TOPK_B3(4);
#undef REC_TOPK_B3
      if (nsplit > 1)
        hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, st, part_s, part_i, nsplit, Q, k,
                           out_scores, out_idx);
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
  }
#define REC_TOPK(KS_)                                                                                           \
  hipLaunchKernelGGL((topk_ip_kernel<KS_>), dim3((unsigned)blocks), dim3(256), 0, st, queries, q_stride, items, \
                     items_stride, Q, (int)N, d, k, out_scores, out_idx)
  if (d <= 16) REC_TOPK(1);
  else if (d <= 32) REC_TOPK(2);
  else if (d <= 64) REC_TOPK(4);
  else REC_TOPK(8);
#undef REC_TOPK
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_topk_ip_f32(const float* queries, int64_t q_stride, int64_t Q, const float* items,
                               int64_t items_stride, int64_t N, int32_t d, int32_t k, float* out_scores,
                               int64_t* out_idx, void* stream) {
  return rec_topk_ip_ws_f32(queries, q_stride, Q, items, items_stride, N, d, k, out_scores, out_idx, nullptr, stream);
}
