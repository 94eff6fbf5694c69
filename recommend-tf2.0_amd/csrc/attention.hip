// Attention-family interaction kernels (round-1 versions: exact fp32, LDS-staged, VALU math; the
// MFMA tilings for the QK^T / PV contractions are the next optimisation step):
//   K6  rec_mha_ctr_f32        ctr MultiHeadAttention (AutoInt)   src/ctr/layers/modules.py:285-325
//   K7  rec_din_attn_pool_f32  DIN AttentionLayer pooling         src/ctr/layers/modules.py:144-175
//   K8  rec_mha_rowmask_f32    match scaled-dot-product attention src/match/layers/modules.py:76-96,115-131
#include <stdlib.h>

#include "attention_ctr.h"
#include <type_traits>

#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The reference's padding value: python `-2 ** 32 + 1` = -4294967295 -> fp32 -4294967296.0
__device__ constexpr float kNegPad = -4294967296.0f;

// ------------------------------------------------------------------------------------------------
// K6 — one workgroup per sample; X, Q, K, V and the H x N x N probabilities live in LDS.
//   q = act(Xq Wq) etc. (no bias);  P = softmax(q k^T * sqrt(S));  out = merge(P v)
//   use_res: out = relu(out + act(Xv W0)).
// LDS floats: max(nx*N*din, H*N*N) + 3*N*HS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mha_ctr_kernel(const float* __restrict__ xq,
                                                      const float* __restrict__ xk,
                                                      const float* __restrict__ xv, int N, int din,
                                                      const float* __restrict__ Wq,
                                                      const float* __restrict__ Wk,
                                                      const float* __restrict__ Wv,
                                                      const float* __restrict__ W0, int H, int S,
                                                      int act, int nx, int regionA,
                                                      float* __restrict__ out) {
  // LDS: region A = the nx (1 if xq==xk==xv, else 3) input copies, later REUSED for the H*N*N
  // probabilities; region B = Q, K, V.  regionA = max(nx*N*din, H*N*N) floats.
  extern __shared__ float lds[];
  const int HS = H * S;
  float* Xq = lds;
  float* Xk = nx == 3 ? Xq + N * din : Xq;
  float* Xv = nx == 3 ? Xk + N * din : Xq;
  float* Q = lds + regionA;
  float* Kt = Q + N * HS;
  float* V = Kt + N * HS;
  float* Pm = lds;  // [H][N][N], valid after the projection phase
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t xoff = b * (int64_t)N * din;
  float* orow = out + b * (int64_t)N * HS;
  for (int e = tid; e < N * din; e += 256) {
    Xq[e] = xq[xoff + e];
    if (nx == 3) {
      Xk[e] = xk[xoff + e];
      Xv[e] = xv[xoff + e];
    }
  }
  __syncthreads();
  // projections: thread -> (n, c), c fastest (coalesced W reads).  The residual branch
  // act(Xv W0) is parked in the output row (same thread re-reads its own element later).
  for (int e = tid; e < N * HS; e += 256) {
    const int n = e / HS, c = e - n * HS;
    float aq = 0.f, ak = 0.f, av = 0.f, ar = 0.f;
    for (int k = 0; k < din; ++k) {
      aq = fmaf(Xq[n * din + k], Wq[k * HS + c], aq);
      ak = fmaf(Xk[n * din + k], Wk[k * HS + c], ak);
      av = fmaf(Xv[n * din + k], Wv[k * HS + c], av);
      if (W0) ar = fmaf(Xv[n * din + k], W0[k * HS + c], ar);
    }
    Q[e] = act_apply(aq, act, 0.f);
    Kt[e] = act_apply(ak, act, 0.f);
    V[e] = act_apply(av, act, 0.f);
    if (W0) orow[e] = act_apply(ar, act, 0.f);
  }
  __syncthreads();
  // scores + softmax: one thread per (h, i) row.  "/ (S ** -0.5)" == "* sqrt(S)" (modules.py:235-237)
  const float scale = sqrtf((float)S);
  for (int r = tid; r < H * N; r += 256) {
    const int h = r / N, i = r - h * N;
    float* prow = Pm + (size_t)r * N;
    float m = -INFINITY;
    for (int j = 0; j < N; ++j) {
      float s = 0.f;
      for (int k = 0; k < S; ++k) s = fmaf(Q[i * HS + h * S + k], Kt[j * HS + h * S + k], s);
      s *= scale;
      prow[j] = s;
      m = fmaxf(m, s);
    }
    float l = 0.f;
    for (int j = 0; j < N; ++j) {
      const float e = expf(prow[j] - m);
      prow[j] = e;
      l += e;
    }
    const float inv = 1.f / l;
    for (int j = 0; j < N; ++j) prow[j] *= inv;
  }
  __syncthreads();
  // out[i][h*S+s] = sum_j P[h][i][j] V[j][h*S+s]  (+ residual)
  for (int e = tid; e < N * HS; e += 256) {
    const int i = e / HS, c = e - i * HS;
    const int h = c / S;
    const float* prow = Pm + ((size_t)h * N + i) * N;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(prow[j], V[j * HS + c], acc);
    if (W0) acc = relu_nan(acc + orow[e]);
    orow[e] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// K7 — DIN pooling, one wave per sample, online softmax, k/v rows read exactly once (HBM bound:
// 2*T*d*4 B per sample, or T*d*4 when k == v).  The Dense(1) over [q, k, q-k, q*k] is affine in k:
//   score_t = act(k_t . (w2 - w3 + q*w4) + q . (w1 + w3) + bias)          (verified)
// lanes 0..d/4-1 each own 4 columns (d <= 256, d % 4 == 0).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void din_pool_kernel(const float* __restrict__ q,
                                                       const float* __restrict__ k,
                                                       const float* __restrict__ v,
                                                       const float* __restrict__ mask,
                                                       const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ alpha, int act,
                                                       int has_mask, int64_t B, int T, int d,
                                                       float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int nv = d >> 2;
  const bool on = lane < nv;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 qv = on ? reinterpret_cast<const f32x4*>(q + b * d)[lane] : z4;
  const f32x4 w1 = on ? reinterpret_cast<const f32x4*>(W)[lane] : z4;
  const f32x4 w2 = on ? reinterpret_cast<const f32x4*>(W + d)[lane] : z4;
  const f32x4 w3 = on ? reinterpret_cast<const f32x4*>(W + 2 * d)[lane] : z4;
  const f32x4 w4 = on ? reinterpret_cast<const f32x4*>(W + 3 * d)[lane] : z4;
  const f32x4 u = w2 - w3 + qv * w4;
  const f32x4 cw = qv * (w1 + w3);
  const float c0 = wave_sum(cw.x + cw.y + cw.z + cw.w) + bias[0];
  const float al = alpha ? alpha[0] : 0.f;
  const bool same = (k == v);
  const f32x4* pk = reinterpret_cast<const f32x4*>(k + b * (int64_t)T * d);
  const f32x4* pv = reinterpret_cast<const f32x4*>(v + b * (int64_t)T * d);
  float m = -INFINITY, l = 0.f;
  f32x4 acc = z4;
  constexpr int U = 4;
  for (int t0 = 0; t0 < T; t0 += U) {
    f32x4 kr[U], vr[U];
#pragma unroll
    for (int e = 0; e < U; ++e) {
      const int t = t0 + e < T ? t0 + e : T - 1;
      kr[e] = on ? pk[(int64_t)t * nv + lane] : z4;
      vr[e] = same ? kr[e] : (on ? pv[(int64_t)t * nv + lane] : z4);
    }
#pragma unroll
    for (int e = 0; e < U; ++e) {
      if (t0 + e >= T) break;
      const f32x4 pr = kr[e] * u;
      float s = wave_sum(pr.x + pr.y + pr.z + pr.w) + c0;
      s = act_apply(s, act, al);
      if (!has_mask || mask[b * T + t0 + e] == 0.f) s = kNegPad;  // modules.py:161-165
      const float mn = fmaxf(m, s);
      const float sc = expf(m - mn);  // first step: exp(-inf) = 0
      const float p = expf(s - mn);
      acc = acc * sc + vr[e] * p;
      l = l * sc + p;
      m = mn;
    }
  }
  if (on) reinterpret_cast<f32x4*>(out + b * d)[lane] = acc * (1.f / l);
}

// ------------------------------------------------------------------------------------------------
// K7 fused with the history gather (K1): same online-softmax pooling, but the T history rows are
// fetched straight from the embedding tables (n_tab tables x Dt columns, lane -> (table, 16-B
// chunk)); the (B, T, d) tensor never exists.  HBM: T*d*4 B of rows + T*n_tab*4 B of ids per sample.
// ------------------------------------------------------------------------------------------------
struct DinTables {
  const float* base[8];
  int32_t vocab[8];
};

template <int IDS_F32>
__global__ __launch_bounds__(256) void din_gather_pool_kernel(const float* __restrict__ q, DinTables tb,
                                                              int n_tab, int Dt, const void* __restrict__ ids,
                                                              const float* __restrict__ mask, int mask_mode,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ alpha, int act,
                                                              int64_t B, int T, float* __restrict__ out,
                                                              int* __restrict__ oob) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int d = n_tab * Dt;
  const int nv = d >> 2;
  const bool on = lane < nv;
  const int lpt = Dt >> 2;                    // lanes per table
  const int tab = on ? lane / lpt : 0;
  const int col = on ? (lane - tab * lpt) * 4 : 0;
  const float* tbase = tb.base[tab];
  const uint32_t tvocab = (uint32_t)tb.vocab[tab];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 qv = on ? reinterpret_cast<const f32x4*>(q + b * d)[lane] : z4;
  const f32x4 w1 = on ? reinterpret_cast<const f32x4*>(W)[lane] : z4;
  const f32x4 w2 = on ? reinterpret_cast<const f32x4*>(W + d)[lane] : z4;
  const f32x4 w3 = on ? reinterpret_cast<const f32x4*>(W + 2 * d)[lane] : z4;
  const f32x4 w4 = on ? reinterpret_cast<const f32x4*>(W + 3 * d)[lane] : z4;
  const f32x4 u = w2 - w3 + qv * w4;
  const f32x4 cw = qv * (w1 + w3);
  const float c0 = wave_sum(cw.x + cw.y + cw.z + cw.w) + bias[0];
  const float al = alpha ? alpha[0] : 0.f;
  const int64_t idbase = b * (int64_t)T * n_tab;
  float m = -INFINITY, l = 0.f;
  f32x4 acc = z4;
  // Padded slots get the logit -2^32+1: as soon as ONE real slot exists their softmax weight underflows to exactly 0
  // (exp(-4e9) = 0 in fp32), so their rows need not be fetched at all -- with pre-padded histories of random length
  // that is half of the traffic.  Only if every slot is padded do all rows count (uniform weights).  The slot flags
  // of 64 positions at a time come from one coalesced id / mask load and a ballot.
  auto is_pad = [&](int t) -> bool {
    if (mask_mode == 0) return true;                                      // non-tensor mask: all padded
    if (mask_mode == 1) return mask[b * T + t] == 0.f;
    return load_id<IDS_F32>(ids, idbase + (int64_t)t * n_tab) == 0;       // slot real iff first id != 0
  };
  bool any_real = false;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    any_real = any_real || __any(t < T && !is_pad(t));
  }
  constexpr int U = 4;  // 8 measured slower (0.197 vs 0.146 ms at config 4)
  for (int tb0 = 0; tb0 < T; tb0 += 64) {
    const int tl = tb0 + lane;
    const bool padl = tl < T ? is_pad(tl) : true;
    const uint64_t padmask = __ballot(padl);
    if (oob && tl < T) {  // out-of-range ids are reported for every slot, fetched or skipped (ids are 1.5 % of the bytes)
      bool bad = false;
      for (int c = 0; c < n_tab; ++c)
        bad = bad || (uint32_t)load_id<IDS_F32>(ids, idbase + (int64_t)tl * n_tab + c) >= (uint32_t)tb.vocab[c];
      if (bad) *oob = 1;
    }
    uint64_t todo = __ballot(tl < T && (!padl || !any_real));             // slots whose rows matter
    while (todo) {
      f32x4 kr[U];
      int tt[U];
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int bit = todo ? __builtin_ctzll(todo) : -1;                // wave-uniform
        tt[e] = bit;
        if (bit >= 0) todo &= todo - 1;
        const int t = tb0 + (bit >= 0 ? bit : 0);
        const int32_t id = load_id<IDS_F32>(ids, idbase + (int64_t)t * n_tab + tab);
        const bool ok = (uint32_t)id < tvocab;
        const f32x4 row = *reinterpret_cast<const f32x4*>(tbase + (int64_t)(ok ? id : 0) * Dt + col);
        kr[e] = (on && ok) ? row : z4;
      }
#pragma unroll
      for (int e = 0; e < U; ++e) {
        if (tt[e] < 0) break;
        const f32x4 pr = kr[e] * u;
        float s = wave_sum(pr.x + pr.y + pr.z + pr.w) + c0;
        s = act_apply(s, act, al);
        if ((padmask >> tt[e]) & 1) s = kNegPad;
        const float mn = fmaxf(m, s);
        const float sc = expf(m - mn);
        const float p = expf(s - mn);
        acc = acc * sc + kr[e] * p;
        l = l * sc + p;
        m = mn;
      }
    }
  }
  if (on) reinterpret_cast<f32x4*>(out + b * d)[lane] = acc * (1.f / l);
}

// Same op, latency-restructured (round 2): the kernel above walks the slots with an id load -> address -> row load
// chain per batch of 4 slots, i.e. two dependent memory latencies per batch (0.106 ms at config 4 = 0.40 of the HBM
// roofline on the bytes it needs).  Here a wave first stages ALL ids of its sample in LDS (coalesced), compacts the
// list of slots whose rows matter, and then streams the rows with the loads of batch i+1 in flight while batch i is
// reduced (register double buffer): one memory latency per batch, hidden behind the previous batch's arithmetic.
// Results are bit-identical to the streaming kernel (same per-slot arithmetic, same slot order).
template <int IDS_F32, int U>
__global__ __launch_bounds__(256) void din_gather_pool_lds_kernel(const float* __restrict__ q, DinTables tb, int n_tab,
                                                                  int Dt, const void* __restrict__ ids,
                                                                  const float* __restrict__ mask, int mask_mode,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ alpha, int act, int64_t B,
                                                                  int T, float* __restrict__ out, int* __restrict__ oob) {
  extern __shared__ int32_t din_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= B) return;
  int32_t* sid = din_lds + (size_t)w * T * (n_tab + 1);   // [T][n_tab] ids of the sample
  int32_t* slots = sid + (size_t)T * n_tab;                // [<= T] slots to fetch, in order; bit 31 = padded slot
  const int d = n_tab * Dt;
  const int nv = d >> 2;
  const bool on = lane < nv;
  const int lpt = Dt >> 2;
  const int tab = on ? lane / lpt : 0;
  const int col = on ? (lane - tab * lpt) * 4 : 0;
  const float* tbase = tb.base[tab];
  const uint32_t tvocab = (uint32_t)tb.vocab[tab];
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t idbase = b * (int64_t)T * n_tab;
  const int nid = T * n_tab;
  bool bad = false;
  for (int e = lane; e < nid; e += 64) {
    const int32_t id = load_id<IDS_F32>(ids, idbase + e);
    sid[e] = id;
    const int c = e % n_tab;
    bad = bad || (uint32_t)id >= (uint32_t)tb.vocab[c];
  }
  if (oob && bad) *oob = 1;
  const f32x4 qv = on ? reinterpret_cast<const f32x4*>(q + b * d)[lane] : z4;
  const f32x4 w1 = on ? reinterpret_cast<const f32x4*>(W)[lane] : z4;
  const f32x4 w2 = on ? reinterpret_cast<const f32x4*>(W + d)[lane] : z4;
  const f32x4 w3 = on ? reinterpret_cast<const f32x4*>(W + 2 * d)[lane] : z4;
  const f32x4 w4 = on ? reinterpret_cast<const f32x4*>(W + 3 * d)[lane] : z4;
  const f32x4 u = w2 - w3 + qv * w4;
  const f32x4 cw = qv * (w1 + w3);
  const float c0 = wave_sum(cw.x + cw.y + cw.z + cw.w) + bias[0];
  const float al = alpha ? alpha[0] : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  auto is_pad = [&](int t) -> bool {
    if (mask_mode == 0) return true;
    if (mask_mode == 1) return mask[b * T + t] == 0.f;
    return sid[t * n_tab] == 0;
  };
  bool any_real = false;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    any_real = any_real || __any(t < T && !is_pad(t));
  }
  int n = 0;  // wave-uniform count of listed slots
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const bool padl = t < T ? is_pad(t) : true;
    const bool take = t < T && (!padl || !any_real);
    const uint64_t m = __ballot(take);
    if (take) slots[n + __popcll(m & ((1ull << lane) - 1ull))] = t | (padl ? (int)0x80000000 : 0);
    n += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  auto load_batch = [&](int i0, f32x4 (&kr)[U]) {
#pragma unroll
    for (int e = 0; e < U; ++e) {
      const int i = i0 + e < n ? i0 + e : (n > 0 ? n - 1 : 0);   // clamped: the load stays unconditional
      const int t = n > 0 ? (slots[i] & 0x7fffffff) : 0;
      const int32_t id = sid[t * n_tab + tab];
      const bool ok = (uint32_t)id < tvocab;
      const f32x4 row = *reinterpret_cast<const f32x4*>(tbase + (int64_t)(ok ? id : 0) * Dt + col);
      kr[e] = (on && ok) ? row : z4;
    }
  };
  float m = -INFINITY, l = 0.f;
  f32x4 acc = z4;
  auto reduce_batch = [&](int i0, const f32x4 (&kr)[U]) {
#pragma unroll
    for (int e = 0; e < U; ++e) {
      if (i0 + e >= n) break;  // wave-uniform
      const f32x4 pr = kr[e] * u;
      float s = wave_sum(pr.x + pr.y + pr.z + pr.w) + c0;
      s = act_apply(s, act, al);
      if (slots[i0 + e] < 0) s = kNegPad;
      const float mn = fmaxf(m, s);
      const float sc = expf(m - mn);
      const float p = expf(s - mn);
      acc = acc * sc + kr[e] * p;
      l = l * sc + p;
      m = mn;
    }
  };
  f32x4 ka[U], kb[U];
  load_batch(0, ka);
  for (int i0 = 0; i0 < n; i0 += 2 * U) {
    load_batch(i0 + U, kb);
    reduce_batch(i0, ka);
    load_batch(i0 + 2 * U, ka);
    reduce_batch(i0 + U, kb);
  }
  if (on) reinterpret_cast<f32x4*>(out + b * d)[lane] = acc * (1.f / l);
}

// Same op, one 16-lane group per history slot (round 2, second pass).  The kernel above spends ~70 VALU instructions per
// slot: every slot is a 64-lane reduction (six cross-lane steps, two of them across DPP rows) plus two full expf and a
// rescale of the accumulator, for 768 B of row data — rocprofv3 put it at 102.8 us for config 4 (0.41 of the HBM
// roofline on the bytes it needs), about half of that VALU issue.  Here table rows are Dt = 64 floats = 16 lanes x 16 B,
// so lane group g of the wave takes slot 4i + g and every lane loads its 16-B piece of EACH of the slot's n_tab rows
// (one load instruction per table covers four slots: 1 KiB); a score is a 16-lane reduction (four DPP steps inside one
// row), the four groups keep independent online-softmax states that are merged once at the end, the rescale happens once
// per batch (batch maximum first), and exp is v_exp_f32 on log2(e)-scaled scores: ~14 VALU per slot.
// Round 3: the lanes per row are a template parameter — LPR = Dt / 4 in {4, 8, 16, 32} (tables 16 / 32 / 64 / 128 wide), 64 / LPR
// lane groups per wave, one slot each; Dt = 64 (LPR 16) is the BASELINE configs[3] instantiation and compiles as before.
template <int IDS_F32, int NTAB, int LPR = 16>
__global__ __launch_bounds__(256, (NTAB * LPR <= 48 ? 4 : 1)) void din_gather_pool_grp_kernel(const float* __restrict__ q, DinTables tb,
                                                                  const void* __restrict__ ids,
                                                                  const float* __restrict__ mask, int mask_mode,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ alpha, int act, int64_t B,
                                                                  int T, float* __restrict__ out, int* __restrict__ oob) {
  // batch depth (tools/exp/din_u_ab.sh, same box): U = 1 (94 VGPRs, five waves per SIMD, 120 KiB in flight per CU) and U = 2
  // (120 VGPRs, four waves, 192 KiB) both run configs[3] in 64.3 us: neither the bytes in flight nor the wave count bound it
#ifndef REC_DIN_U
#define REC_DIN_U 2
#endif
  constexpr int Dt = LPR * 4, d = NTAB * Dt, U = REC_DIN_U, NG = 64 / LPR;   // NG lane groups = slots per load step
  constexpr float kLog2e = 1.4426950408889634f;
  extern __shared__ int32_t din_lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + w;   // 4 waves per block, or 1 (see the dispatch)
  if (b >= B) return;
  int32_t* sid = din_lds + (size_t)w * T * (NTAB + 1);   // [T][NTAB] ids of the sample
  int32_t* slots = sid + (size_t)T * NTAB;                // [<= T] slots to fetch, in order; bit 31 = padded slot
  const int sub = lane % LPR, grp = lane / LPR;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int64_t idbase = b * (int64_t)T * NTAB;
  const int nid = T * NTAB;
  bool bad = false;
  for (int e = lane; e < nid; e += 64) {
    const int32_t id = load_id<IDS_F32>(ids, idbase + e);
    sid[e] = id;
    bad = bad || (uint32_t)id >= (uint32_t)tb.vocab[e % NTAB];
  }
  if (oob && bad) *oob = 1;
  // score(k) = k . (w2 - w3 + q o w4) + q . (w1 + w3) + bias   (affine in the history row k)
  f32x4 u[NTAB];
  float cpart = 0.f;
#pragma unroll
  for (int tt = 0; tt < NTAB; ++tt) {
    const int c = tt * Dt + sub * 4;
    const f32x4 qv = *reinterpret_cast<const f32x4*>(q + b * d + c);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(W + c), w2 = *reinterpret_cast<const f32x4*>(W + d + c);
    const f32x4 w3 = *reinterpret_cast<const f32x4*>(W + 2 * d + c), w4 = *reinterpret_cast<const f32x4*>(W + 3 * d + c);
    u[tt] = w2 - w3 + qv * w4;
    const f32x4 cw = qv * (w1 + w3);
    cpart += (cw.x + cw.y) + (cw.z + cw.w);
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) cpart += __shfl_xor(cpart, o, 64);   // every group holds all LPR x NTAB pieces
  const float c0 = cpart + bias[0];
  const float al = alpha ? alpha[0] : 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  auto is_pad = [&](int t) -> bool {
    if (mask_mode == 0) return true;
    if (mask_mode == 1) return mask[b * T + t] == 0.f;
    return sid[t * NTAB] == 0;
  };
  bool any_real = false;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    any_real = any_real || __any(t < T && !is_pad(t));
  }
  int n = 0;  // wave-uniform count of listed slots (>= 1: if no slot is real, all T are listed)
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    const bool padl = t < T ? is_pad(t) : true;
    const bool take = t < T && (!padl || !any_real);
    const uint64_t m = __ballot(take);
    if (take) slots[n + __popcll(m & ((1ull << lane) - 1ull))] = t | (padl ? (int)0x80000000 : 0);
    n += __popcll(m);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // batch = U steps of NG slots; slot index of (step e, group g) = i0 + NG e + g; indices past the end re-read the last
  // listed slot (a cache hit) and enter the softmax with the logit -inf
  // A/B builds (tools/exp/din_ab.sh, configs[3], same box): SKIP = lane groups without a slot issue nothing, NT =
  // streaming row loads.  SKIP 0 / NT 0 (shipped) 66.9-67.1 us; NT alone 71.0; SKIP (with or without NT) 99-101 us — the
  // exec-masked loads lose their batching (a wait lands behind every conditional load).
#ifndef REC_DIN_SKIP
#define REC_DIN_SKIP 0
#endif
#ifndef REC_DIN_NT
#define REC_DIN_NT 0
#endif
  // REC_DIN_NT = 2 (mixed; A/B: 64.5 vs 64.4 us — no gain, off): a batch in which EVERY lane group has its own slot is fetched with the streaming policy; the
  // tail batch and the prefetches past the end, whose spare groups re-read the last slot, keep the default policy (that
  // re-read must stay a cache hit).  The choice is wave-uniform.
  auto load_batch_p = [&](int i0, f32x4 (&kr)[U][NTAB], auto nt_tag) {
    constexpr bool NTV = decltype(nt_tag)::value;
#pragma unroll
    for (int e = 0; e < U; ++e) {
      const int i = i0 + NG * e + grp;
#if REC_DIN_SKIP
      // a lane group without a slot issues nothing (its rows enter the softmax with weight 0 whatever kr holds, but NaN
      // bits would survive the multiply by 0: zeros).  With streaming loads a clamped re-read of the last slot is an
      // HBM fetch, not a cache hit.
      if (i < n) {
        const int t = slots[i] & 0x7fffffff;
#pragma unroll
        for (int tt = 0; tt < NTAB; ++tt) {
          const int32_t id = sid[t * NTAB + tt];
          const bool ok = (uint32_t)id < (uint32_t)tb.vocab[tt];
          const f32x4 row = row_load<NTV>(reinterpret_cast<const f32x4*>(tb.base[tt] + (int64_t)(ok ? id : 0) * Dt + sub * 4));
          kr[e][tt] = ok ? row : z4;
        }
      } else {
#pragma unroll
        for (int tt = 0; tt < NTAB; ++tt) kr[e][tt] = z4;
      }
#else
      const int t = slots[i < n ? i : n - 1] & 0x7fffffff;
#pragma unroll
      for (int tt = 0; tt < NTAB; ++tt) {
        const int32_t id = sid[t * NTAB + tt];
        const bool ok = (uint32_t)id < (uint32_t)tb.vocab[tt];
        const f32x4 row = row_load<NTV>(reinterpret_cast<const f32x4*>(tb.base[tt] + (int64_t)(ok ? id : 0) * Dt + sub * 4));
        kr[e][tt] = ok ? row : z4;
      }
#endif
    }
  };
  auto load_batch = [&](int i0, f32x4 (&kr)[U][NTAB]) {
#if REC_DIN_NT == 2
    if (i0 + NG * U <= n) load_batch_p(i0, kr, std::true_type{});
    else load_batch_p(i0, kr, std::false_type{});
#elif REC_DIN_NT == 1
    load_batch_p(i0, kr, std::true_type{});
#else
    load_batch_p(i0, kr, std::false_type{});
#endif
  };
  float m = -INFINITY, l = 0.f;
  f32x4 acc[NTAB];
#pragma unroll
  for (int tt = 0; tt < NTAB; ++tt) acc[tt] = z4;
  auto reduce_batch = [&](int i0, const f32x4 (&kr)[U][NTAB]) {
    float s[U];
    float mb = m;
#pragma unroll
    for (int e = 0; e < U; ++e) {
      const int i = i0 + NG * e + grp;
      // the tables' products are chained as packed FMAs first, ONE horizontal sum afterwards (was: one per table)
      f32x4 pr = kr[e][0] * u[0];
#pragma unroll
      for (int tt = 1; tt < NTAB; ++tt) pr = __builtin_elementwise_fma(kr[e][tt], u[tt], pr);
      float dsum = (pr.x + pr.y) + (pr.z + pr.w);
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o, 64);
      float sv = act_apply(dsum + c0, act, al);
      if (slots[i < n ? i : n - 1] < 0) sv = kNegPad;
      s[e] = i < n ? sv * kLog2e : -INFINITY;
      mb = fmaxf(mb, s[e]);
    }
    const float mbs = mb == -INFINITY ? 0.f : mb;      // a lane group without a slot yet: all weights 0
    const float sc = __builtin_amdgcn_exp2f(m - mbs);   // m = -inf -> 0
    l *= sc;
#pragma unroll
    for (int tt = 0; tt < NTAB; ++tt) acc[tt] *= sc;
#pragma unroll
    for (int e = 0; e < U; ++e) {
      const float p = __builtin_amdgcn_exp2f(s[e] - mbs);
      l += p;
#pragma unroll
      for (int tt = 0; tt < NTAB; ++tt) acc[tt] += kr[e][tt] * p;
    }
    m = mb;
  };
  f32x4 ka[U][NTAB], kb[U][NTAB];
  load_batch(0, ka);
  for (int i0 = 0; i0 < n; i0 += 2 * NG * U) {
    load_batch(i0 + NG * U, kb);
    reduce_batch(i0, ka);
    load_batch(i0 + 2 * NG * U, ka);
    if (i0 + NG * U < n) reduce_batch(i0 + NG * U, kb);
  }
  // merge the group states
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    const float s1 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mn);
    const float s2 = m2 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
    for (int tt = 0; tt < NTAB; ++tt) {
      f32x4 a2;
      a2.x = __shfl_xor(acc[tt].x, o, 64);
      a2.y = __shfl_xor(acc[tt].y, o, 64);
      a2.z = __shfl_xor(acc[tt].z, o, 64);
      a2.w = __shfl_xor(acc[tt].w, o, 64);
      acc[tt] = acc[tt] * s1 + a2 * s2;
    }
    l = l * s1 + l2 * s2;
    m = mn;
  }
  if (grp == 0) {
    const float inv = 1.f / l;
#pragma unroll
    for (int tt = 0; tt < NTAB; ++tt) *reinterpret_cast<f32x4*>(out + b * d + tt * Dt + sub * 4) = acc[tt] * inv;
  }
}

// ------------------------------------------------------------------------------------------------
// K8 — match attention.  One workgroup per (sample, head, 256-query tile); the head's K and V
// (Sk x dk each) are staged in LDS once per workgroup and read by broadcast (every thread reads
// the same K_j / V_j address: conflict-free), one query row per thread, online softmax.
// Rows whose mask is 0 have EVERY logit replaced by -4294967296.0 (the reference's mask
// broadcasts along the key axis, modules.py:90-91) => uniform attention over all Sk keys.
// dk <= 64 (template), q/k/v laid out (B, S, H*dk) exactly as the Dense projections write them.
// ------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void mha_rowmask_kernel(const float* __restrict__ q,
                                                          const float* __restrict__ k,
                                                          const float* __restrict__ v,
                                                          const float* __restrict__ mask, int Sq,
                                                          int Sk, int H, float* __restrict__ out) {
  extern __shared__ float lds[];
  float* Ks = lds;                    // [Sk][DK]
  float* Vs = lds + (size_t)Sk * DK;  // [Sk][DK]
  const int tid = threadIdx.x;
  const int qt = blockIdx.x;  // query tile
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int dm = H * DK;
  const float* kb = k + b * (int64_t)Sk * dm + h * DK;
  const float* vb = v + b * (int64_t)Sk * dm + h * DK;
  for (int e = tid; e < Sk * (DK / 4); e += 256) {
    const int j = e / (DK / 4), c = e - j * (DK / 4);
    reinterpret_cast<f32x4*>(Ks)[e] = reinterpret_cast<const f32x4*>(kb + (int64_t)j * dm)[c];
    reinterpret_cast<f32x4*>(Vs)[e] = reinterpret_cast<const f32x4*>(vb + (int64_t)j * dm)[c];
  }
  __syncthreads();
  const int i = qt * 256 + tid;
  if (i >= Sq) return;
  const float* qrow = q + (b * Sq + i) * (int64_t)dm + h * DK;
  f32x4 qr[DK / 4], acc[DK / 4];
#pragma unroll
  for (int c = 0; c < DK / 4; ++c) {
    qr[c] = reinterpret_cast<const f32x4*>(qrow)[c];
    acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool masked = mask[b * Sq + i] == 0.f;
  const float inv_sqrt = 1.f / sqrtf((float)DK);
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Sk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DK / 4; ++c) {
      const f32x4 kk = reinterpret_cast<const f32x4*>(Ks + (size_t)j * DK)[c];
      s = fmaf(qr[c].x, kk.x, s);
      s = fmaf(qr[c].y, kk.y, s);
      s = fmaf(qr[c].z, kk.z, s);
      s = fmaf(qr[c].w, kk.w, s);
    }
    s = masked ? kNegPad : s * inv_sqrt;
    const float mn = fmaxf(m, s);
    const float sc = expf(m - mn);
    const float p = expf(s - mn);
    l = l * sc + p;
    m = mn;
#pragma unroll
    for (int c = 0; c < DK / 4; ++c) {
      const f32x4 vv = reinterpret_cast<const f32x4*>(Vs + (size_t)j * DK)[c];
      acc[c] = acc[c] * sc + vv * p;
    }
  }
  const float inv = 1.f / l;
  float* orow = out + (b * Sq + i) * (int64_t)dm + h * DK;
#pragma unroll
  for (int c = 0; c < DK / 4; ++c) reinterpret_cast<f32x4*>(orow)[c] = acc[c] * inv;
}

bool mha_rowmask_b3_dispatch(const float* q, const float* k, const float* v, const float* mask, int64_t B, int Sq,
                             int Sk, int dk, int H, float* out, int64_t qs, int64_t ks, int64_t vs, hipStream_t st);
bool mha_rowmask_smallq_dispatch(const float* q, const float* k, const float* v, const float* mask, int64_t B, int Sq,
                                 int Sk, int dk, int H, float* out, int64_t qs, int64_t ks, int64_t vs,
                                 hipStream_t st);
void mha_gather_fewq_dispatch(const float* q, int64_t q_stride, const float* table, int vocab, const void* ids,
                              bool ids_f32, const float* mask, int64_t B, int Sq, int Sk, int dk, int H, float* out,
                              hipStream_t st);
bool mha_rowmask_mfma_dispatch(const float* q, const float* k, const float* v, const float* mask, int64_t B,
                               int Sq, int Sk, int dk, int H, float* out, hipStream_t st);
bool mha_ctr_mfma_dispatch(const float* xq, const float* xk, const float* xv, int64_t B, int N, int din,
                           const float* Wq, const float* Wk, const float* Wv, const float* W0, int H, int S,
                           int act, float* out, hipStream_t st);
bool mha_ctr_b3_dispatch(const float* xq, const float* xk, const float* xv, int64_t B, int N, int din, const float* Wq,
                         const float* Wk, const float* Wv, const float* W0, int H, int S, int act, float* out,
                         hipStream_t st);

}  // namespace rec

using namespace rec;

extern "C" int rec_mha_ctr_stack_f32(const float* x, int64_t B, int32_t N, int32_t din, const float* const* Wq,
                                     const float* const* Wk, const float* const* Wv, const float* const* W0, int32_t L,
                                     int32_t H, int32_t S, int32_t act, float* out, void* stream) {
  using namespace rec;
  const char* who = "rec_mha_ctr_stack_f32";
  REC_CHECK_ARG(B >= 0 && N >= 1 && din >= 1 && H >= 1 && S >= 1 && L >= 1, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_TANH, REC_EINVAL, "%s: bad act %d", who, act);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(x && Wq && Wk && Wv && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(L <= 4, REC_ENOTIMPL, "%s: at most 4 stacked layers per launch (got %d): split the stack", who, L);
  CtrStackArgs wa{};
  for (int l = 0; l < L; ++l) {
    wa.Wq[l] = Wq[l], wa.Wk[l] = Wk[l], wa.Wv[l] = Wv[l];
    wa.W0[l] = W0 ? W0[l] : nullptr;
  }
  if (!mha_ctr_stack_dispatch(x, B, N, din, wa, L, H, S, act, out, reinterpret_cast<hipStream_t>(stream), nullptr)) {
    set_error("%s: stack not covered (needs S = 16, din in {16, 32}, H in {1, 2}, N <= 64, aligned x / out): run the "
              "layers one by one with rec_mha_ctr_f32", who);
    return REC_ENOTIMPL;
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_autoint_forward_f32(const rec_table_desc* tables, int32_t n_sparse, const int32_t* ids,
                                       int64_t ids_stride, const float* dense, int64_t dense_stride, int32_t n_dense,
                                       const float* dense_embed, int32_t D, const float* const* Wq, const float* const* Wk,
                                       const float* const* Wv, const float* const* W0, int32_t L, int32_t H, int32_t S,
                                       int32_t act, const float* head_w, const float* head_b, int64_t B, float* out_prob,
                                       float* out_fields, int32_t* oob_flag, void* stream) {
  using namespace rec;
  const char* who = "rec_autoint_forward_f32";
  REC_CHECK_ARG(B >= 0 && n_sparse >= 0 && n_dense >= 0 && n_sparse + n_dense >= 1 && D >= 1 && H >= 1 && S >= 1 && L >= 1,
                REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(n_sparse <= REC_MAX_TABLES, REC_ESHAPE, "%s: at most %d sparse fields", who, REC_MAX_TABLES);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_TANH, REC_EINVAL, "%s: bad act %d", who, act);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(Wq && Wk && Wv && head_w && out_prob, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(n_sparse == 0 || (tables && ids && ids_stride >= n_sparse), REC_EINVAL, "%s: tables / ids", who);
  REC_CHECK_ARG(n_dense == 0 || (dense && dense_embed && dense_stride >= n_dense), REC_EINVAL, "%s: dense / dense_embed", who);
  REC_CHECK_ARG(L <= 4, REC_ENOTIMPL, "%s: at most 4 stacked layers per launch (got %d)", who, L);
  CtrStackArgs wa{};
  for (int l = 0; l < L; ++l) {
    wa.Wq[l] = Wq[l], wa.Wk[l] = Wk[l], wa.Wv[l] = Wv[l];
    wa.W0[l] = W0 ? W0[l] : nullptr;
  }
  CtrFusedIo io{};
  for (int f = 0; f < n_sparse; ++f) {
    REC_CHECK_ARG(tables[f].base && aligned16(tables[f].base) && tables[f].dim == D && tables[f].vocab >= 1 &&
                      tables[f].vocab <= 0x7fffffffLL,
                  REC_ESHAPE, "%s: table %d must be 16-B aligned, %d wide, with 1 <= vocab < 2^31", who, f, D);
    io.ts.base[f] = tables[f].base, io.ts.vocab[f] = (int32_t)tables[f].vocab, io.ts.dim[f] = D, io.ts.out_col[f] = f * D;
  }
  io.ids = ids, io.ids_stride = ids_stride, io.n_sparse = n_sparse;
  io.dense = dense, io.dense_stride = dense_stride, io.dense_embed = dense_embed;
  io.head_w = head_w, io.head_b = head_b, io.head_out = out_prob, io.oob = reinterpret_cast<int*>(oob_flag);
  if (!mha_ctr_stack_dispatch(nullptr, B, n_sparse + n_dense, D, wa, L, H, S, act, out_fields,
                              reinterpret_cast<hipStream_t>(stream), &io)) {
    set_error("%s: not covered (needs S = 16, D in {16, 32}, H in {1, 2}, <= 64 fields, 16-B aligned weights): compose "
              "rec_gather_concat_f32 / rec_scale_embed_f32 / rec_mha_ctr_f32 / rec_dense_f32", who);
    return REC_ENOTIMPL;
  }
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_mha_ctr_f32(const float* xq, const float* xk, const float* xv, int64_t B,
                               int32_t N, int32_t din, const float* Wq, const float* Wk,
                               const float* Wv, const float* W0, int32_t H, int32_t S, int32_t act,
                               float* out, void* stream) {
  const char* who = "rec_mha_ctr_f32";
  REC_CHECK_ARG(B >= 0 && N >= 1 && din >= 1 && H >= 1 && S >= 1, REC_ESHAPE, "%s: bad shape", who);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_TANH, REC_EINVAL, "%s: bad act %d", who, act);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(xq && xk && xv && Wq && Wk && Wv && out, REC_EINVAL, "%s: NULL pointer", who);
  {
    // default: bf16x3 kernel (attention_ctr.hip) for the AutoInt shapes; rec_debug_force("mha", "f") keeps the fp32-MFMA
    // kernel, "v" the LDS/VALU kernel (tests / A/B only)
    const char* e = forced("mha");
    if (!(e && (e[0] == 'v' || e[0] == 'f')) && mha_ctr_b3_dispatch(xq, xk, xv, B, N, din, Wq, Wk, Wv, W0, H, S, act, out,
                                                                     reinterpret_cast<hipStream_t>(stream))) {
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
    if (!(e && e[0] == 'v') && mha_ctr_mfma_dispatch(xq, xk, xv, B, N, din, Wq, Wk, Wv, W0, H, S, act, out,
                                                     reinterpret_cast<hipStream_t>(stream))) {
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
  }
  const int nx = (xq == xk && xk == xv) ? 1 : 3;
  size_t regionA = (size_t)nx * N * din;
  if ((size_t)H * N * N > regionA) regionA = (size_t)H * N * N;
  regionA = (regionA + 3) & ~(size_t)3;
  const size_t floats = regionA + (size_t)3 * N * H * S;
  const size_t lds = floats * sizeof(float);
  REC_CHECK_ARG(lds <= 160 * 1024, REC_ESHAPE, "%s: N=%d din=%d H*S=%d needs %zu B of LDS (> 160 KiB)",
                who, N, din, H * S, lds);
  REC_CHECK_ARG(B <= 0x7fffffffLL, REC_ESHAPE, "%s: B too large", who);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mha_ctr_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mha_ctr_kernel, dim3((unsigned)B), dim3(256), lds, st, xq, xk, xv, N, din, Wq, Wk,
                     Wv, W0, H, S, act, nx, (int)regionA, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_din_attn_pool_f32(const float* q, const float* k, const float* v, const float* mask,
                                     const float* W, const float* bias, const float* alpha,
                                     int32_t act, int64_t B, int32_t T, int32_t d, float* out,
                                     void* stream) {
  const char* who = "rec_din_attn_pool_f32";
  REC_CHECK_ARG(B >= 0 && T >= 1 && d >= 4 && d % 4 == 0 && d <= 256, REC_ESHAPE,
                "%s: need d %% 4 == 0, 4 <= d <= 256 (got T=%d d=%d)", who, T, d);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_PRELU, REC_EINVAL, "%s: bad act %d", who, act);
  REC_CHECK_ARG(act != REC_ACT_PRELU || alpha, REC_EINVAL, "%s: PReLU needs alpha", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(q && k && v && W && bias && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(W) && aligned16(out),
                REC_EINVAL, "%s: q/k/v/W/out must be 16-B aligned", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(din_pool_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, q, k, v, mask, W,
                     bias, alpha, act, mask ? 1 : 0, B, T, d, out);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_din_attn_pool_f32(const float* q, const rec_table_desc* tables, int32_t n_tab,
                                            const void* ids, int32_t ids_dtype, const float* mask,
                                            int32_t mask_from_ids, const float* W, const float* bias,
                                            const float* alpha, int32_t act, int64_t B, int32_t T,
                                            float* out, int32_t* oob_flag, void* stream) {
  const char* who = "rec_gather_din_attn_pool_f32";
  REC_CHECK_ARG(tables && n_tab >= 1 && n_tab <= 8, REC_ESHAPE, "%s: n_tab=%d outside [1,8]", who, n_tab);
  const int Dt = tables[0].dim;
  DinTables tb;
  for (int t = 0; t < 8; ++t) {
    const rec_table_desc& s = tables[t < n_tab ? t : 0];
    REC_CHECK_ARG(s.base && aligned16(s.base) && s.dim == Dt && s.vocab >= 1 && s.vocab <= 0x7fffffffLL,
                  REC_ESHAPE, "%s: tables must be 16-B aligned and share one dim", who);
    tb.base[t] = s.base;
    tb.vocab[t] = (int32_t)s.vocab;
  }
  const int d = n_tab * Dt;
  REC_CHECK_ARG(B >= 0 && T >= 1 && Dt % 4 == 0 && d <= 256, REC_ESHAPE, "%s: need Dt %% 4 == 0 and d <= 256", who);
  REC_CHECK_ARG(act >= REC_ACT_NONE && act <= REC_ACT_PRELU && (act != REC_ACT_PRELU || alpha), REC_EINVAL,
                "%s: bad act", who);
  REC_CHECK_ARG(ids_dtype == REC_IDS_I32 || ids_dtype == REC_IDS_F32, REC_EINVAL, "%s: bad ids_dtype", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(q && ids && W && bias && out && aligned16(q) && aligned16(W) && aligned16(out), REC_EINVAL,
                "%s: NULL or unaligned pointer", who);
  const int mode = mask ? 1 : (mask_from_ids ? 2 : 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  // ids of a sample staged in LDS + pipelined row loads (din_gather_pool_lds_kernel) while the 4 waves' id lists fit
  // 48 KiB; longer histories keep the streaming kernel.  rec_debug_force("din", "s" | "l"): tests / A/B.
  const char* din_forced = forced("din");
  const bool lds_ok = !(din_forced && din_forced[0] == 's');
  const size_t lds = (size_t)4 * T * (n_tab + 1) * sizeof(int32_t);
  // 64-wide tables (16 lanes x 16 B per row): one lane group per history slot (din_gather_pool_grp_kernel);
  // REC_DIN_IMPL=lds keeps the wave-per-slot kernel for A/B
  const bool grp_ok = !(din_forced && (din_forced[0] == 's' || din_forced[0] == 'l'));
  // rows of 16 / 32 / 64 / 128 floats (4 / 8 / 16 / 32 lanes x 16 B): one lane group per history slot
  // (din_gather_pool_grp_kernel; 64 is the BASELINE configs[3] width).  Register budget: NTAB x LPR <= 64 (d <= 256)
  if (grp_ok && (Dt == 64 || Dt == 32 || Dt == 16 || Dt == 128) && n_tab <= 4 && n_tab * Dt <= 256 && lds <= 48 * 1024) {
    // ONE wave per workgroup (round 3): a sample's wave slot is free again as soon as that sample is done, instead of when
    // the workgroup's four histories are (history lengths differ up to 100 x) — 65.4 -> 64.4 us at configs[3], same box,
    // three interleaved repeats (profiles/r03_din_wpb_ab.txt)
    const dim3 ggrid((unsigned)B), gblock(64);
    const size_t glds = lds / 4;
#define REC_DIN_GRP(IDF_, NT_, LPR_)                                                                                      \
  hipLaunchKernelGGL((din_gather_pool_grp_kernel<IDF_, NT_, LPR_>), ggrid, gblock, glds, st, q, tb, ids, mask, mode, W, bias, \
                     alpha, act, B, T, out, oob_flag)
#define REC_DIN_GRP_NT(IDF_, LPR_)             \
  if (n_tab == 1) REC_DIN_GRP(IDF_, 1, LPR_);  \
  else if (n_tab == 2) REC_DIN_GRP(IDF_, 2, LPR_); \
  else if (n_tab == 3) REC_DIN_GRP(IDF_, 3, LPR_); \
  else REC_DIN_GRP(IDF_, 4, LPR_)
    if (Dt == 64) {
      if (ids_dtype == REC_IDS_F32) { REC_DIN_GRP_NT(1, 16); } else { REC_DIN_GRP_NT(0, 16); }
    } else if (Dt == 32) {
      if (ids_dtype == REC_IDS_F32) { REC_DIN_GRP_NT(1, 8); } else { REC_DIN_GRP_NT(0, 8); }
    } else if (Dt == 16) {
      if (ids_dtype == REC_IDS_F32) { REC_DIN_GRP_NT(1, 4); } else { REC_DIN_GRP_NT(0, 4); }
    } else {  // 128-wide: at most two tables (d <= 256)
      if (ids_dtype == REC_IDS_F32) {
        if (n_tab == 1) REC_DIN_GRP(1, 1, 32); else REC_DIN_GRP(1, 2, 32);
      } else {
        if (n_tab == 1) REC_DIN_GRP(0, 1, 32); else REC_DIN_GRP(0, 2, 32);
      }
    }
#undef REC_DIN_GRP_NT
#undef REC_DIN_GRP
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
  if (lds_ok && lds <= 48 * 1024) {
    if (ids_dtype == REC_IDS_F32)
      hipLaunchKernelGGL((din_gather_pool_lds_kernel<1, 4>), grid, block, lds, st, q, tb, n_tab, Dt, ids, mask, mode, W,
                         bias, alpha, act, B, T, out, oob_flag);
    else
      hipLaunchKernelGGL((din_gather_pool_lds_kernel<0, 4>), grid, block, lds, st, q, tb, n_tab, Dt, ids, mask, mode, W,
                         bias, alpha, act, B, T, out, oob_flag);
    REC_CHECK_LAUNCH(who);
    return REC_OK;
  }
  if (ids_dtype == REC_IDS_F32)
    hipLaunchKernelGGL((din_gather_pool_kernel<1>), grid, block, 0, st, q, tb, n_tab, Dt, ids, mask, mode, W, bias,
                       alpha, act, B, T, out, oob_flag);
  else
    hipLaunchKernelGGL((din_gather_pool_kernel<0>), grid, block, 0, st, q, tb, n_tab, Dt, ids, mask, mode, W, bias,
                       alpha, act, B, T, out, oob_flag);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_mha_rowmask_strided_f32(const float* q, int64_t q_stride, const float* k, int64_t k_stride,
                                           const float* v, int64_t v_stride, const float* mask, int64_t B,
                                           int32_t Sq, int32_t Sk, int32_t dm, int32_t H, float* out,
                                           void* stream) {
  const char* who = "rec_mha_rowmask_f32";
  REC_CHECK_ARG(q_stride >= dm && k_stride >= dm && v_stride >= dm && q_stride % 4 == 0 && k_stride % 4 == 0 &&
                    v_stride % 4 == 0,
                REC_ESHAPE, "%s: row strides must be >= dm and multiples of 4 floats", who);
  const bool contiguous = q_stride == dm && k_stride == dm && v_stride == dm;
  REC_CHECK_ARG(B >= 0 && Sq >= 1 && Sk >= 1 && H >= 1 && dm >= H && dm % H == 0, REC_ESHAPE,
                "%s: bad shape Sq=%d Sk=%d dm=%d H=%d", who, Sq, Sk, dm, H);
  const int dk = dm / H;
  REC_CHECK_ARG(dk == 8 || dk == 16 || dk == 32 || dk == 64, REC_ESHAPE,
                "%s: head depth %d not in {8,16,32,64}", who, dk);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(q && k && v && mask && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out), REC_EINVAL,
                "%s: q/k/v/out must be 16-B aligned", who);
  REC_CHECK_ARG(B <= 65535, REC_ESHAPE, "%s: B > 65535 per call (chunk the batch)", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  {
    // default for Sq > 8, dk in {32, 64}: bf16x3 matrix-core kernel streaming K/V tiles (attention_b3.hip; no
    // sequence-length limit).  rec_debug_force("mha", "f") keeps the fp32-MFMA kernel, "v" the round-1 VALU kernel.
    const char* e = forced("mha");
    if (!(e && (e[0] == 'v' || e[0] == 'f')) && Sq > 8 &&
        mha_rowmask_b3_dispatch(q, k, v, mask, B, Sq, Sk, dk, H, out, q_stride, k_stride, v_stride, st)) {
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
    if (!(e && e[0] == 'v') &&
        mha_rowmask_smallq_dispatch(q, k, v, mask, B, Sq, Sk, dk, H, out, q_stride, k_stride, v_stride, st)) {
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
  }
  REC_CHECK_ARG(contiguous, REC_ENOTIMPL, "%s: strided q/k/v need Sq <= 8 or (Sq >= 16 and dk in {32, 64})", who);
  const size_t lds = (size_t)2 * Sk * dk * sizeof(float);
  REC_CHECK_ARG(lds <= 160 * 1024, REC_ESHAPE, "%s: Sk=%d dk=%d needs %zu B of LDS", who, Sk, dk, lds);
  {
    const char* e = forced("mha");  // "v" forces the round-1 VALU kernel (tests / A/B only)
    if (!(e && e[0] == 'v') && mha_rowmask_mfma_dispatch(q, k, v, mask, B, Sq, Sk, dk, H, out, st)) {
      REC_CHECK_LAUNCH(who);
      return REC_OK;
    }
  }
  dim3 grid((unsigned)((Sq + 255) / 256), (unsigned)H, (unsigned)B);
#define REC_MHA(DK_)                                                                               \
  case DK_: {                                                                                      \
    if (lds > 64 * 1024) {                                                                         \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mha_rowmask_kernel<DK_>),   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: hipFuncSetAttribute: %s", who,                 \
                    hipGetErrorString(e));                                                         \
    }                                                                                              \
    hipLaunchKernelGGL((mha_rowmask_kernel<DK_>), grid, dim3(256), lds, st, q, k, v, mask, Sq, Sk, \
                       H, out);                                                                    \
    break;                                                                                         \
  }
  switch (dk) { REC_MHA(8) REC_MHA(16) REC_MHA(32) REC_MHA(64) }
#undef REC_MHA
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_gather_mha_fewq_f32(const float* q, int64_t q_stride, const float* table, int32_t vocab,
                                       const void* ids, int32_t ids_dtype, const float* mask, int64_t B,
                                       int32_t Sq, int32_t Sk, int32_t dm, int32_t H, float* out, void* stream) {
  const char* who = "rec_gather_mha_fewq_f32";
  REC_CHECK_ARG(B >= 0 && Sq >= 1 && Sq <= 8 && Sk >= 1 && H >= 1 && dm >= H && dm % H == 0 && vocab >= 1, REC_ESHAPE,
                "%s: bad shape Sq=%d (1..8) Sk=%d dm=%d H=%d vocab=%d", who, Sq, Sk, dm, H, vocab);
  const int dk = dm / H;
  REC_CHECK_ARG(dk == 16 || dk == 32 || dk == 64, REC_ESHAPE, "%s: head depth %d not in {16,32,64}", who, dk);
  REC_CHECK_ARG(ids_dtype == REC_IDS_I32 || ids_dtype == REC_IDS_F32, REC_EINVAL, "%s: bad ids_dtype", who);
  REC_CHECK_ARG(q_stride >= dm && q_stride % 4 == 0, REC_ESHAPE, "%s: bad q_stride", who);
  if (B == 0) return REC_OK;
  REC_CHECK_ARG(q && table && ids && mask && out, REC_EINVAL, "%s: NULL pointer", who);
  REC_CHECK_ARG(aligned16(q) && aligned16(table) && aligned16(out), REC_EINVAL, "%s: q/table/out must be 16-B aligned",
                who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  mha_gather_fewq_dispatch(q, q_stride, table, vocab, ids, ids_dtype == REC_IDS_F32, mask, B, Sq, Sk, dk, H, out, st);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

extern "C" int rec_mha_rowmask_f32(const float* q, const float* k, const float* v, const float* mask,
                                   int64_t B, int32_t Sq, int32_t Sk, int32_t dm, int32_t H,
                                   float* out, void* stream) {
  return rec_mha_rowmask_strided_f32(q, dm, k, dm, v, dm, mask, B, Sq, Sk, dm, H, out, stream);
}

// ================================================================================================
// K8 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32): flash-style, one workgroup of
// 4 waves per (sample, head); the head's K (row stride DK+4: conflict-free ds_read_b128) and V
// live in LDS; every wave owns 32-query tiles.
//
// Orientation (cdna guide §3 "accumulator tile as the next MFMA's operand"): the scores are
// computed TRANSPOSED, S^T = K Q^T, so a lane owns ONE query (column) and holds 16 of the 32 keys
// of a tile in its accumulator registers (the other 16 sit in lane ^ 32): the row softmax is
// register-local plus one cross-half exchange, and the probabilities are — with no data movement
// — the B operand of O^T = V^T P^T, whose accumulator again has the query on the lane, so the
// online-softmax rescale is a per-lane multiply.  k-index permutations are free as long as both
// operands agree: lane half h holds k in [h*DK/2, (h+1)*DK/2) for Q and K, and key
// (r&3) + 8(r>>2) + 4h of the tile for P and V.
// ================================================================================================
namespace rec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DK>
__global__ __launch_bounds__(512) void mha_rowmask_mfma_kernel(const float* __restrict__ q,
                                                               const float* __restrict__ k,
                                                               const float* __restrict__ v,
                                                               const float* __restrict__ mask, int Sq,
                                                               int Sk, int H, float* __restrict__ out) {
  constexpr int KH = DK / 2;        // k values per lane half = QK^T MFMA steps
  constexpr int LDK = DK + 4;       // K row stride in LDS (floats)
  constexpr int NDT = DK / 32;      // 32-wide output column tiles
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int SkP = (Sk + 31) & ~31;
  float* Ks = lds;                          // [SkP][LDK]
  float* Vs = lds + (size_t)SkP * LDK;      // [SkP][DK]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int dm = H * DK;
  const float* kb = k + b * (int64_t)Sk * dm + h * DK;
  const float* vb = v + b * (int64_t)Sk * dm + h * DK;
  // stage K and V (pad rows: zeros, so that 0 * V stays 0)
  for (int e = tid; e < SkP * (DK / 4); e += blockDim.x) {
    const int j = e / (DK / 4), c = e - j * (DK / 4);
    f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;
    if (j < Sk) {
      kk = reinterpret_cast<const f32x4*>(kb + (int64_t)j * dm)[c];
      vv = reinterpret_cast<const f32x4*>(vb + (int64_t)j * dm)[c];
    }
    *reinterpret_cast<f32x4*>(Ks + (size_t)j * LDK + c * 4) = kk;
    *reinterpret_cast<f32x4*>(Vs + (size_t)j * DK + c * 4) = vv;
  }
  __syncthreads();

  const int ql = lane & 31;   // query within the tile (and key row / d column for operand reads)
  const int hf = lane >> 5;   // lane half
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)DK);
  const int nqt = (Sq + 31) >> 5;
  // 8 waves = 2 per SIMD: one wave's softmax (VALU) hides under its partner's MFMAs
  const int nwv = blockDim.x >> 6;
  for (int qt = wv; qt < nqt; qt += nwv) {
    const int qi = qt * 32 + ql;
    const int qc = qi < Sq ? qi : Sq - 1;
    // Q operand: B[k][j = query]; lane holds Q[query][hf*KH + s], s = 0..KH-1
    float qreg[KH];
    {
      const f32x4* qp = reinterpret_cast<const f32x4*>(q + (b * Sq + qc) * (int64_t)dm + h * DK + hf * KH);
#pragma unroll
      for (int c = 0; c < KH / 4; ++c) {
        const f32x4 t = qp[c];
        qreg[4 * c] = t.x;
        qreg[4 * c + 1] = t.y;
        qreg[4 * c + 2] = t.z;
        qreg[4 * c + 3] = t.w;
      }
    }
    const bool masked = mask[b * Sq + qc] == 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[NDT];
#pragma unroll
    for (int t = 0; t < NDT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;

    for (int kt = 0; kt < SkP; kt += 32) {
      // ---- S^T tile = K_tile (32 keys x DK) . Q_tile^T
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const float* krow = Ks + (size_t)(kt + ql) * LDK + hf * KH;
#pragma unroll
      for (int c = 0; c < KH / 4; ++c) {
        const f32x4 kk = *reinterpret_cast<const f32x4*>(krow + 4 * c);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[4 * c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[4 * c + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[4 * c + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[4 * c + 3], s, 0, 0, 0);
      }
      // ---- online softmax for this lane's query over its 16 keys (+ the other half's 16)
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * hf;
        float x = masked ? 0.f : s[r] * scale_log2e;   // masked query row: every logit equal
        x = key < Sk ? x : -INFINITY;                   // LDS pad rows
        s[r] = x;
        tmax = fmaxf(tmax, x);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      const float resc = exp2f(m_run - m_new);          // first tile: exp2(-inf) = 0
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
      // ---- O^T += V_tile^T . P^T : A[i = d][k = key], B[k = key][j = query] = s[r]
#pragma unroll
      for (int t = 0; t < NDT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= resc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt + (r & 3) + 8 * (r >> 2) + 4 * hf;
          const float a = Vs[(size_t)key * DK + t * 32 + ql];
          o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[t], 0, 0, 0);
        }
      }
    }
    // ---- write O: lane = query, register r of tile t = column t*32 + (r&3) + 8(r>>2) + 4hf
    if (qi < Sq) {
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


// ------------------------------------------------------------------------------------------------
// K8, few query rows (Sq <= 8: SASRec's last encoder block only encodes x[:, -1],
// src/match/sasrec/model.py:88): "decode-style", HBM-bound — K and V of the head are read exactly
// once, straight from global memory.  One wave per (sample, head, query); LPK = DK/4 lanes share a
// key (16 B each), 64/LPK keys per wave step, one online-softmax state per lane group, merged at
// the end by a butterfly over the group index.
// ------------------------------------------------------------------------------------------------
// GATHER: k == v == rows of one embedding table addressed by ids (B, Sk) (`k` = table base, `ks` = its row stride
// = dm, `v` unused); ids outside [0, vocab) read as zero rows (the pad rows of `seq_embed * mask`,
// src/match/sasrec/model.py:81-82).  The (B, Sk, dm) sequence tensor is then never written or re-read.
template <int DK, bool GATHER = false, int IDS_F32 = 0>
__global__ __launch_bounds__(256) void mha_rowmask_smallq_kernel(const float* __restrict__ q,
                                                                 const float* __restrict__ k,
                                                                 const float* __restrict__ v,
                                                                 const float* __restrict__ mask, int Sq,
                                                                 int Sk, int H, int64_t total,
                                                                 float* __restrict__ out, int64_t qs, int64_t ks,
                                                                 int64_t vs, const void* __restrict__ ids = nullptr,
                                                                 int vocab = 0) {
  constexpr int LPK = DK / 4;       // lanes per key
  constexpr int KPS = 64 / LPK;     // keys per wave step
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, i, h) flattened
  if (w >= total) return;
  const int h = (int)(w % H);
  const int64_t bi = w / H;         // b * Sq + i
  const int64_t b = bi / Sq;
  const int dm = H * DK;
  const int sub = lane % LPK, grp = lane / LPK;
  const f32x4 qv = reinterpret_cast<const f32x4*>(q + bi * qs + h * DK)[sub];
  const bool masked = mask[bi] == 0.f;
  const float scale_log2e = 1.4426950408889634f / sqrtf((float)DK);
  const float* kb = GATHER ? k + h * DK + sub * 4 : k + b * (int64_t)Sk * ks + h * DK + sub * 4;
  const float* vb = GATHER ? nullptr : v + b * (int64_t)Sk * vs + h * DK + sub * 4;
  float m = -INFINITY, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 4;
  for (int j0 = 0; j0 < Sk; j0 += KPS * U) {
    f32x4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int j = j0 + u * KPS + grp;
      j = j < Sk ? j : Sk - 1;
      if constexpr (GATHER) {
        const int32_t id = load_id<IDS_F32>(ids, b * (int64_t)Sk + j);
        const bool ok = (uint32_t)id < (uint32_t)vocab;
        const f32x4 row = *reinterpret_cast<const f32x4*>(kb + (int64_t)(ok ? id : 0) * ks);
        kr[u] = ok ? row : f32x4{0.f, 0.f, 0.f, 0.f};
        vr[u] = kr[u];
      } else {
        kr[u] = *reinterpret_cast<const f32x4*>(kb + (int64_t)j * ks);
        vr[u] = *reinterpret_cast<const f32x4*>(vb + (int64_t)j * vs);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * KPS + grp;
      const f32x4 pr = kr[u] * qv;
      float s = pr.x + pr.y + pr.z + pr.w;
#pragma unroll
      for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      s = masked ? 0.f : s * scale_log2e;     // masked query row: all logits equal -> uniform
      if (j >= Sk) s = -INFINITY;
      const float mn = fmaxf(m, s);
      const float sc = mn == -INFINITY ? 0.f : exp2f(m - mn);
      const float p = mn == -INFINITY ? 0.f : exp2f(s - mn);
      acc = acc * sc + vr[u] * p;
      l = l * sc + p;
      m = mn;
    }
  }
  // merge the KPS group states (butterfly over lane bits >= log2(LPK))
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64);
    const float l2 = __shfl_xor(l, o, 64);
    f32x4 a2;
    a2.x = __shfl_xor(acc.x, o, 64);
    a2.y = __shfl_xor(acc.y, o, 64);
    a2.z = __shfl_xor(acc.z, o, 64);
    a2.w = __shfl_xor(acc.w, o, 64);
    const float mn = fmaxf(m, m2);
    const float s1 = m == -INFINITY ? 0.f : exp2f(m - mn);
    const float s2 = m2 == -INFINITY ? 0.f : exp2f(m2 - mn);
    acc = acc * s1 + a2 * s2;
    l = l * s1 + l2 * s2;
    m = mn;
  }
  if (grp == 0) reinterpret_cast<f32x4*>(out + bi * dm + h * DK)[sub] = acc * (1.f / l);
}

template <int DK>
static bool launch_mha_mfma(const float* q, const float* k, const float* v, const float* mask, int64_t B,
                            int Sq, int Sk, int H, float* out, hipStream_t st) {
  const int SkP = (Sk + 31) & ~31;
  const size_t lds = (size_t)SkP * (2 * DK + 4) * sizeof(float);
  if (lds > 160 * 1024) return false;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(mha_rowmask_mfma_kernel<DK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return false;
  }
  const int nqt = (Sq + 31) / 32;
  const int threads = nqt > 4 ? 512 : 256;
  hipLaunchKernelGGL((mha_rowmask_mfma_kernel<DK>), dim3((unsigned)H, (unsigned)B), dim3(threads), lds, st, q, k, v,
                     mask, Sq, Sk, H, out);
  return true;
}

// few query rows: HBM-bound decode-style kernel (takes row strides)
bool mha_rowmask_smallq_dispatch(const float* q, const float* k, const float* v, const float* mask, int64_t B, int Sq,
                                 int Sk, int dk, int H, float* out, int64_t qs, int64_t ks, int64_t vs,
                                 hipStream_t st) {
  if (!(Sq <= 8 && (dk == 64 || dk == 32 || dk == 16))) return false;
  const int64_t total = B * Sq * H;
  const dim3 grid((unsigned)((total + 3) / 4)), block(256);
  if (dk == 64)
    hipLaunchKernelGGL((mha_rowmask_smallq_kernel<64>), grid, block, 0, st, q, k, v, mask, Sq, Sk, H, total, out, qs,
                       ks, vs);
  else if (dk == 32)
    hipLaunchKernelGGL((mha_rowmask_smallq_kernel<32>), grid, block, 0, st, q, k, v, mask, Sq, Sk, H, total, out, qs,
                       ks, vs);
  else
    hipLaunchKernelGGL((mha_rowmask_smallq_kernel<16>), grid, block, 0, st, q, k, v, mask, Sq, Sk, H, total, out, qs,
                       ks, vs);
  return true;
}

void mha_gather_fewq_dispatch(const float* q, int64_t q_stride, const float* table, int vocab, const void* ids,
                              bool ids_f32, const float* mask, int64_t B, int Sq, int Sk, int dk, int H, float* out,
                              hipStream_t st) {
  const int64_t total = B * Sq * H;
  const dim3 grid((unsigned)((total + 3) / 4)), block(256);
  const int64_t dm = (int64_t)dk * H;
#define REC_GMHA(DK_, F_)                                                                                             \
  hipLaunchKernelGGL((mha_rowmask_smallq_kernel<DK_, true, F_>), grid, block, 0, st, q, table, (const float*)nullptr, \
                     mask, Sq, Sk, H, total, out, q_stride, dm, dm, ids, vocab)
  if (dk == 64) { if (ids_f32) REC_GMHA(64, 1); else REC_GMHA(64, 0); }
  else if (dk == 32) { if (ids_f32) REC_GMHA(32, 1); else REC_GMHA(32, 0); }
  else { if (ids_f32) REC_GMHA(16, 1); else REC_GMHA(16, 0); }
#undef REC_GMHA
}

bool mha_rowmask_mfma_dispatch(const float* q, const float* k, const float* v, const float* mask, int64_t B,
                               int Sq, int Sk, int dk, int H, float* out, hipStream_t st) {
  if (Sq < 16 || B > 65535) return false;
  if (dk == 64) return launch_mha_mfma<64>(q, k, v, mask, B, Sq, Sk, H, out, st);
  if (dk == 32) return launch_mha_mfma<32>(q, k, v, mask, B, Sq, Sk, H, out, st);
  return false;
}

}  // namespace rec

// ================================================================================================
// K6 on the fp32 matrix cores (v_mfma_f32_16x16x4_f32): AutoInt interacting layer with head size
// S = 16 (BASELINE config 3: 39 fields, d = 16, H = 2).  One WAVE per sample (4 per workgroup), all
// intermediates in a wave-private LDS region, no barriers after the weights are staged.
//   Q, K, V = act(X W)            16x16 output tiles, written to LDS
//   S^T     = K_h Q_h^T * sqrt(S) transposed scores: a lane owns one query, its 4 lane groups hold
//                                  the keys -> softmax = 12 local values + two cross-group shuffles
//   O^T     = V_h^T P^T            P^T accumulators are the B operand as they stand
//   R^T     = W0^T Xv^T            residual branch computed transposed so that it lands in the
//                                  same (query on lane, column in register) layout as O^T
//   out     = relu(O + act(R))    16-B stores
// ================================================================================================
namespace rec {

// NTc / KSc > 0: compile-time field-tile count and din/4 (full unrolling lets the compiler batch the
// ds_reads ahead of each MFMA chain); 0 = runtime values
template <int NTc, int KSc>
__global__ __launch_bounds__(256) void mha_ctr_mfma_kernel(const float* __restrict__ xq,
                                                           const float* __restrict__ xk,
                                                           const float* __restrict__ xv, int64_t B, int N,
                                                           int din, const float* __restrict__ Wq,
                                                           const float* __restrict__ Wk,
                                                           const float* __restrict__ Wv,
                                                           const float* __restrict__ W0, int H, int act,
                                                           int nx, float* __restrict__ out) {
  constexpr int S = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HS = H * S;
  const int NP = (N + 15) & ~15;   // padded field count
  const int NT = NTc > 0 ? NTc : (NP >> 4);
  const int LDX = din + 1;         // X row stride (bank spread for the row-per-lane operand reads)
  const int LDQ = S + 1;           // per-head Q/K/V tiles: 2+ workgroups fit a CU's LDS
  // block-shared weights [4][din][HS]
  float* Wsh = lds;
  const int wsz = din * HS;
  const int tid = threadIdx.x;
  for (int e = tid; e < wsz; e += 256) {
    Wsh[e] = Wq[e];
    Wsh[wsz + e] = Wk[e];
    Wsh[2 * wsz + e] = Wv[e];
    Wsh[3 * wsz + e] = W0 ? W0[e] : 0.f;
  }
  __syncthreads();
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = (int64_t)blockIdx.x * 4 + wv;
  if (b >= B) return;  // wave-uniform; no barrier below
  const int per_wave = nx * NP * LDX + 3 * NP * LDQ;
  float* Xs = lds + 4 * wsz + (size_t)wv * per_wave;   // [nx][NP][LDX]
  float* Qs = Xs + nx * NP * LDX;                       // [NP][LDQ]
  float* Ks = Qs + NP * LDQ;
  float* Vs = Ks + NP * LDQ;
  const float* xin[3] = {xq, xk, xv};
  for (int c = 0; c < nx; ++c)
    for (int e = lane; e < NP * din; e += 64) {
      const int n = e / din, kk = e - n * din;
      Xs[(c * NP + n) * LDX + kk] = n < N ? xin[c][(b * N + n) * (int64_t)din + kk] : 0.f;
    }
  const float* Xq = Xs;
  const float* Xk = nx == 3 ? Xs + NP * LDX : Xs;
  const float* Xv = nx == 3 ? Xs + 2 * NP * LDX : Xs;
  const int lr = lane & 15, g = lane >> 4;
  const int ksteps = KSc > 0 ? KSc : (din >> 2);

  const float scale = 4.0f * 1.4426950408889634f;  // "/ (S ** -0.5)" = x sqrt(16), folded with log2(e)
  for (int h = 0; h < H; ++h) {
    // ---- projections of head h: Q_h, K_h, V_h = act(X W[:, 16h:16h+16]) -> wave-private LDS
    for (int m = 0; m < 3; ++m) {
      const float* X = m == 0 ? Xq : (m == 1 ? Xk : Xv);
      const float* Wm = Wsh + m * wsz;
      float* dst = m == 0 ? Qs : (m == 1 ? Ks : Vs);
#pragma unroll
      for (int rt = 0; rt < NT; ++rt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < ksteps; ++st) {
          const float a = X[(rt * 16 + lr) * LDX + 4 * st + g];
          const float bw = Wm[(4 * st + g) * HS + h * 16 + lr];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(rt * 16 + 4 * g + r) * LDQ + lr] = act_apply(acc[r], act, 0.f);
      }
    }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      // ---- transposed scores for this query tile against every key tile (NT <= 4)
      f32x4 sc[4];
      float mloc = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        sc[kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < NT) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            const float a = Ks[(kt * 16 + lr) * LDQ + 4 * st + g];
            const float bq = Qs[(qt * 16 + lr) * LDQ + 4 * st + g];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * g + r;
            const float x = key < N ? acc[r] * scale : -INFINITY;
            sc[kt][r] = x;
            mloc = fmaxf(mloc, x);
          }
        }
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      float lsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = exp2f(sc[kt][r] - mloc);   // pad keys: exp2(-inf) = 0
          sc[kt][r] = p;
          lsum += p;
        }
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      const float inv = 1.f / lsum;
      // ---- O^T = V_h^T P^T
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        if (kt < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = Vs[(kt * 16 + 4 * g + r) * LDQ + lr];
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sc[kt][r] * inv, o, 0, 0, 0);
          }
        }
      // ---- residual, transposed: R^T[c][n] = sum_k W0[k][c] Xv[n][k]
      if (W0) {
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};
        const float* W0s = Wsh + 3 * wsz;
#pragma unroll
        for (int st = 0; st < ksteps; ++st) {
          const float a = W0s[(4 * st + g) * HS + h * 16 + lr];
          const float bx = Xv[(qt * 16 + lr) * LDX + 4 * st + g];
          rr = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx, rr, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = relu_nan(o[r] + act_apply(rr[r], act, 0.f));
      }
      const int qi = qt * 16 + lr;
      if (qi < N) *reinterpret_cast<f32x4*>(out + (b * N + qi) * (int64_t)HS + h * 16 + 4 * g) = o;
    }
  }
}

bool mha_ctr_mfma_dispatch(const float* xq, const float* xk, const float* xv, int64_t B, int N, int din,
                           const float* Wq, const float* Wk, const float* Wv, const float* W0, int H, int S,
                           int act, float* out, hipStream_t st) {
  if (S != 16 || N > 64 || din % 4 != 0 || din > 256 || !aligned16(out)) return false;
  const int nx = (xq == xk && xk == xv) ? 1 : 3;
  const int HS = H * 16, NP = (N + 15) & ~15;
  const size_t floats = (size_t)4 * din * HS + (size_t)4 * ((size_t)nx * NP * (din + 1) + (size_t)3 * NP * (16 + 1));
  const size_t lds = floats * sizeof(float);
  if (lds > 160 * 1024) return false;
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define REC_CTR(NT_, KS_)                                                                                   \
  {                                                                                                         \
    if (lds > 64 * 1024 &&                                                                                  \
        hipFuncSetAttribute(reinterpret_cast<const void*>(mha_ctr_mfma_kernel<NT_, KS_>),                   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)            \
      return false;                                                                                         \
    hipLaunchKernelGGL((mha_ctr_mfma_kernel<NT_, KS_>), grid, block, lds, st, xq, xk, xv, B, N, din, Wq, Wk, \
                       Wv, W0, H, act, nx, out);                                                            \
    return true;                                                                                            \
  }
  if (NP == 48 && din == 16) REC_CTR(3, 4)      // 39 fields x dim 16 (BASELINE config 3, layer 1)
  if (NP == 48 && din == 32) REC_CTR(3, 8)      // layers 2.. (H*S = 32 inputs)
  if (NP == 32 && din == 16) REC_CTR(2, 4)      // 26 sparse fields only
  REC_CTR(0, 0)
#undef REC_CTR
}

}  // namespace rec
