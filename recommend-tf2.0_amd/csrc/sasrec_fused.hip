// SASRec forward, one encoder block, last position only, in ONE launch (BASELINE configs[4]).
//
// What it replaces (src/match/sasrec/model.py:72-96 with blocks = 1, num_heads = 1; the encoder block is
// src/match/layers/modules.py:152-185, its attention :76-131, its FFN :134-149):
//     mask        = seq != 0                                      :72
//     seq_embed   = Embedding_seq(seq) * mask                     :75, :81-82
//     att         = softmax(wq(x) wk(x)^T / sqrt(d)) wv(x)        (query rows masked, keys never)
//     out1        = LN1(x + att);  out2 = LN2(out1 + FFN(out1));  att_outputs = out2 * mask       :85-86
//     seq_info    = att_outputs[:, -1]                            :88
//     logits      = [Embedding_pos(pos) . seq_info, Embedding_neg(neg) . seq_info]                :77-79, :88-96
// Only the LAST position of the block is consumed, so only that query row is encoded (exact), and with one head
// the K / V projections of the S positions fold away (exact algebra, see match/layers/modules.py in this repo):
//     q . (x_j Wk + bk) = x_j . (Wk q) + const,     sum_j p_j (x_j Wv + bv) = (sum_j p_j x_j) Wv + bv.
// Round 2 profile of the unfused path at B = 8192, S = 200, d = 64, 100 negatives: two HBM-bound kernels (~85 us)
// inside a 234-us forward made of ~10 short launches + ~8 elementwise launches.  Here one wave owns a sample from
// its ids to its logits: last-row lookup -> Wq -> Wk^T pull-back -> attention over the raw item rows fetched by
// id (pad / out-of-range slots are zero rows: not fetched, counted into the softmax denominator) -> Wv -> LN1 ->
// FFN -> LN2 * mask -> the 1 + n candidate rows fetched by id and dotted.  Bytes from HBM = the rows of the real
// slots + the candidate rows + the ids: nothing else is read or written (logits and seq_info excepted).
//
// Layout: a vector of d = 64 elements lives one element per lane.  The block's weights (112 KiB at ffn = 128) are
// staged once per workgroup into LDS in their Keras (in, out) layout — lane o reads W[i][o], conflict-free — Wk
// transposed with a padded stride; a workgroup is 16 waves (one per CU: the weights take most of the LDS) that
// walk the samples persistently and never synchronise again.  Row fetches use 16 lanes per 256-B row (16 B each),
// 4 rows per wave instruction, two landing buffers of kU instructions each per wave.
//
// Measured (MI355X, config 5): 234 us unfused -> 100.6 us (first version: one landing buffer of 8 loads, ids fetched
// when needed) -> 91.6 us (ids by LDS-DMA one sample ahead, two landing buffers of 4 loads, fetches issued before the
// matvec chains) = 4.8 TB/s of required bytes.  8 loads per buffer spill (10 VGPRs) and run at 112 us: a spilled
// landing register is a wait for its load.  With every sample at the mean length: 86 us, so the static sample ->
// wave assignment costs 6 %; the rest is bytes in flight: 16 waves x 8 KiB per CU is all the VGPR file gives at 128
// registers per wave, and the LDS that could land more is full of weights.  (Dynamic sample -> wave assignment through an
// LDS counter inside the workgroup was tried: 92.4 vs 92.0 us on the same box — the residual imbalance is between CUs.)
// Second session: fewer waves with deeper landing buffers LOSE (tools/exp/sasrec_waves_ab.sh, same box, waves x loads per
// buffer): 16 x 4 (128 KiB in flight per CU, shipped) 86.0 us; 12 x 6 (144 KiB) 93.4; 8 x 8 (128 KiB) 93.4; 8 x 12 (192 KiB)
// 103.3; 8 x 16 (256 KiB) 105.0 — the number of waves that can overlap each other's serial phases matters more than the
// bytes each keeps in flight.
#include "common.h"

namespace rec {
namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef REC_SASREC_ROWS_NT
#define REC_SASREC_ROWS_NT 1
#endif
constexpr bool kRowsNT = REC_SASREC_ROWS_NT != 0;   // streaming row loads: 94.0 -> 89.3 us at configs[4] (common.h row_load)

constexpr int kD = 64;       // d_model
#ifndef REC_SASREC_WAVES
#define REC_SASREC_WAVES 16
#endif
constexpr int kWaves = REC_SASREC_WAVES;   // per workgroup (one workgroup per CU: the weights fill the LDS)
#ifndef REC_SASREC_KU
#define REC_SASREC_KU 4
#endif
constexpr int kU = REC_SASREC_KU;   // row-load instructions per landing buffer (4 rows each)
// landing buffers per wave (16 rows each at kU = 4), attention phase x candidate phase.  Round 3, same-box A/Bs with every
// arm checked against the oracle (tools/exp/sasrec_bufs_ab.sh, profiles/r03_sasrec_bufs_ab*.txt): 1 x 2 (shipped) 81.3 /
// 84.3 us against 82.6 / 85.7 for 2 x 2; 1 x 3, 1 x 2 with 5 or 6 loads per buffer 81.5 - 82.0; 1 x 1 with 7 loads 85.8;
// 2 x 4, 3 x 4, 4 x 4 82.9 / 84.9 / 99.1.  More rows in flight per wave do not help: ~33 MB of row requests are outstanding
// chip-wide, the memory system is saturated for this pattern — touching the candidate rows ahead (LDS-DMA of 4 B per lane
// into a dead buffer while the matvec chain runs) costs 14 % (r03_sasrec_touch_ab.txt), and a batch with every sample at
// the mean length is only 6 % faster (r03_sasrec_lens_ab.txt); compacting the NEXT sample's id list while this sample's
// candidate rows are in flight (the 9 % of a sample spent there, moved under a memory wait) changes nothing: 83.1 - 86.2 vs
// 82.9 - 84.8 us (r03_sasrec_early_compact_ab.txt).  (An arm that indexes a landing buffer it does not have
// compiles, skips that buffer's loads and looks 10 % faster: hence the oracle check per arm.)
#ifndef REC_SASREC_ATT_BUFS
#define REC_SASREC_ATT_BUFS 1
#endif

#ifndef REC_SASREC_CAND_BUFS
#define REC_SASREC_CAND_BUFS 2
#endif
constexpr int kNA = REC_SASREC_ATT_BUFS, kNC = REC_SASREC_CAND_BUFS;
// Experiment builds only (-DREC_SASREC_STAMPS, tools/exp/sasrec_stamps.py): s_memtime at the phase boundaries of every
// sample, written over seq_info[b, 0..8] (lane 0) — where a sample's ~20 us go (profiles/r03_sasrec_stamps.txt: attention
// loop 29 %, matvec chain 27 %, candidate loop 28 %, id compaction 9 %, Wq / Wk 6 %; weight staging 3.2 us per workgroup).
#ifdef REC_SASREC_STAMPS
#define REC_STAMP(i) do { if (lane == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define REC_STAMP(i) do { } while (0)
#endif
// Round 3 also tried DYNAMIC sample -> wave assignment (ticket counters, tools/exp/sasrec_variants/sasrec_fused_tickets.hip):
// bit-identical results, but 86.0-86.2 us against 84.5-85.3 us for this static stride on the same box, whether a wave
// claims its next sample at the top of the current one or as late as its ids can still arrive
// (profiles/r03_sasrec_balanced_ab_*.txt) — and device-scope atomics on ONE address serialise at ~27 ns (a single
// counter plus a "last wave out" counter: 371 / 273 us).  The waves' busy times differ (mean 46, max 63 of the same
// units), yet evening them out buys nothing: what bounds the kernel is the rate at which the memory system serves 256-B
// row requests to all waves together, not the slowest wave.
constexpr int kNB = kNA > kNC ? kNA : kNC;

struct SasrecParams {
  const float *wq, *bq, *wk, *wv, *bv, *g1, *be1, *w1, *b1, *w2, *b2, *g2, *be2;
  float eps1, eps2;
  int fh;  // FFN hidden width: 64 or 128
};

__device__ __forceinline__ float bcast(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// LDS-DMA, 4 B per active lane: LDS[lds + 4 * lane] <- *g.  No VGPR destination: the wave does not hold (or spill,
// which would mean waiting for) what it prefetches.  The compiler does not see this load in its vmcnt bookkeeping; its
// own waits then only become stricter (loads retire in issue order), never too weak.
__device__ __forceinline__ void glds4(const void* g, uint32_t lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds)
      : "memory");
}

// y[lane] = sum_i x_i W[i][col].  W sits in LDS as [i / 4][NCOL columns][4]: lane `col` fetches W[i .. i+3][col] with ONE
// conflict-free ds_read_b128, x comes from a wave-private LDS vector as a broadcast ds_read_b128 (every lane the same
// address), and the four products are two v_pk_fma_f32 — 6 instructions (2 VALU) per four terms where the first version
// spent 12 (v_readlane + ds_read_b32 + v_fmac per term, 8 VALU): the SQ counters had this kernel at ~80 % VALU issue
// (4 waves per SIMD, 20 % VALU-active each), and the seven matvecs of a sample were ~900 of its ~3 400 VALU
// instructions.  Same four partial sums (i mod 4) and the same final combination as before: bit-identical results.
// (The column index is laundered through an empty asm: the weights do not depend on the sample, and without it the
// compiler hoists all 28 672 / 64 weight reads out of the persistent loop and spills them.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NI, int NCOL>
__device__ __forceinline__ float matvec_pk(const float* __restrict__ W4, const float* __restrict__ xb, int col, float init) {
  f32x2 a01 = {init, 0.f}, a23 = {0.f, 0.f};
  // a real loop over blocks of 16 rows (not unrolled): 4 weight quads in registers at a time — the row landing buffers
  // of the caller hold 64 of the wave's 128 VGPRs while this runs
#pragma unroll 1
  for (int i0 = 0; i0 < NI; i0 += 16) {
    asm volatile("" : "+v"(col));
    const float* Wb = W4 + ((i0 >> 2) * NCOL + col) * 4;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(Wb + (i >> 2) * NCOL * 4);
      const f32x4 x = *reinterpret_cast<const f32x4*>(xb + i0 + i);
      a01 = __builtin_elementwise_fma(f32x2{x.x, x.y}, f32x2{w.x, w.y}, a01);
      a23 = __builtin_elementwise_fma(f32x2{x.z, x.w}, f32x2{w.z, w.w}, a23);
    }
  }
  return (a01.x + a01.y) + (a23.x + a23.y);
}

// wave-private LDS hand-over: what the lanes wrote is visible to the whole wave afterwards (LDS operations of one wave
// execute in order, so a later overwrite cannot overtake these reads either)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float layer_norm64(float v, float g, float b, float eps) {
  const float mu = wave_sum(v) * (1.f / 64.f);
  const float c = v - mu;
  const float var = wave_sum(c * c) * (1.f / 64.f);
  return c * (1.f / sqrtf(var + eps)) * g + b;
}

__device__ __forceinline__ float group16_sum(float s) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}

template <int FH>
__global__ __launch_bounds__(kWaves * 64) void sasrec_last_row_kernel(
    SasrecParams P, const float* __restrict__ seq_table, int32_t seq_vocab, const int32_t* __restrict__ seq_ids,
    int64_t seq_stride, int S, int32_t pad_id, const int32_t* __restrict__ mask_ids, int64_t mask_stride,
    const float* __restrict__ pos_table, int32_t pos_vocab, const int32_t* __restrict__ pos_ids, int64_t pos_stride,
    int n_pos, const float* __restrict__ neg_table, int32_t neg_vocab, const int32_t* __restrict__ neg_ids,
    int64_t neg_stride, int n_neg, int64_t B, float* __restrict__ seq_info, float* __restrict__ logits,
    int64_t logits_stride, int* __restrict__ oob, int lst_cap, int cand_cap) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef REC_SASREC_STAMPS
  const long long t_entry = (long long)__builtin_amdgcn_s_memtime();
#endif
  // weights as [i / 4][columns][4] (matvec_pk): (i, c) at ((i >> 2) * NCOL + c) * 4 + (i & 3)
  float* const Wq = lds;                        // in 64, out 64
  float* const WkT = Wq + 64 * 64;              // "in" = o, column = c: WkT(o, c) = Wk[c][o]  (q_back = Wk q)
  float* const Wv = WkT + 64 * 64;
  float* const W1 = Wv + 64 * 64;               // in 64, out FH
  float* const W2 = W1 + 64 * FH;               // in FH, out 64
  float* const vec = W2 + FH * 64;              // bq bv g1 be1 b2 g2 be2 (64 each), b1 (FH)
  float* const xbuf = vec + 7 * 64 + FH + (threadIdx.x >> 6) * 64;    // per wave: the vector the next matvec consumes
  // per wave: two sequence-id buffers (the list of real rows is compacted in place in the one being consumed, the other
  // receives the next sample's ids), the candidate ids, two mask words
  int32_t* const wave_lds = reinterpret_cast<int32_t*>(vec + 7 * 64 + FH + kWaves * 64) + (threadIdx.x >> 6) * (2 * lst_cap + cand_cap + 2);
  const int tid = threadIdx.x;
  const int lane_c = tid & 63, wave = tid >> 6;
  const float scale_log2e = 1.4426950408889634f * 0.125f;   // log2(e) / sqrt(64)
  const int n_cand = n_pos + n_neg;
  int32_t* const cbuf = wave_lds + 2 * lst_cap;
  int32_t* const msk = cbuf + cand_cap;
  // LDS byte address of this wave's area (wave-uniform; readfirstlane tells the compiler so)
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)wave_lds);

  // the first sample's ids leave before the weights are staged: their latency hides under the staging (round 3)
  const int64_t bstep = (int64_t)gridDim.x * kWaves;
  int64_t b = (int64_t)blockIdx.x * kWaves + wave;
  auto prefetch_seq_ids = [&](int64_t bb, int which) {
    for (int c = 0; c * 64 < S; ++c)
      if (c * 64 + lane_c < S) glds4(seq_ids + bb * seq_stride + c * 64 + lane_c, lds0 + (uint32_t)(which * lst_cap + c * 64) * 4u);
    if (lane_c == 0) glds4(mask_ids + bb * mask_stride, lds0 + (uint32_t)(2 * lst_cap + cand_cap + which) * 4u);
  };
  if (b < B) prefetch_seq_ids(b, 0);
  auto w4 = [](int i, int c, int ncol) { return ((i >> 2) * ncol + c) * 4 + (i & 3); };
  for (int e = tid; e < 64 * 64; e += kWaves * 64) {       // e = i * 64 + o of the Keras (in, out) kernels
    const int i = e >> 6, o = e & 63;
    Wq[w4(i, o, 64)] = P.wq[e];
    Wv[w4(i, o, 64)] = P.wv[e];
    WkT[w4(o, i, 64)] = P.wk[e];
  }
  for (int e = tid; e < 64 * FH; e += kWaves * 64) {
    W1[w4(e / FH, e % FH, FH)] = P.w1[e];                  // (in 64, out FH)
    W2[w4(e >> 6, e & 63, 64)] = P.w2[e];                  // (in FH, out 64)
  }
  if (tid < 64) {
    vec[tid] = P.bq[tid];
    vec[64 + tid] = P.bv[tid];
    vec[128 + tid] = P.g1[tid];
    vec[192 + tid] = P.be1[tid];
    vec[256 + tid] = P.b2[tid];
    vec[320 + tid] = P.g2[tid];
    vec[384 + tid] = P.be2[tid];
  }
  if (tid < FH) vec[448 + tid] = P.b1[tid];
  __syncthreads();

  // ---- per-sample pipeline ---------------------------------------------------------------------------------------
  // A wave's samples are a serial chain of memory round trips (ids -> rows -> ... -> candidate rows); with 16 waves per
  // CU that chain, not the bandwidth, set the first version's time (100 us).  So every fetch is issued as early as its
  // address is known: the NEXT sample's ids and this sample's candidate ids by LDS-DMA at the top (no registers: a
  // prefetched value the compiler has to spill is a value it waits for); the last row and the first two row batches
  // before the Wq / Wk matvecs; row batch i + 2 as soon as batch i has been reduced (two landing buffers in
  // registers); the first two candidate batches before the Wv / LN / FFN chain (candidate rows do not depend on it).
  int cur = 0;
  for (; b < B; b += bstep, cur ^= 1) {
    // everything lane-dependent below derives from a laundered lane index: otherwise the compiler hoists dozens of
    // per-lane invariants (bias reads, table-select pointers of every unrolled slot) out of this loop and spills them
    int ln = lane_c;
#ifdef REC_SASREC_STAMPS
    long long stamps[8];
    if (ln == 0) stamps[0] = (long long)__builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ln) : : "memory");     // this sample's ids have landed in LDS
    const int lane = ln, sub = ln & 15, grp = ln >> 4;
    REC_STAMP(1);
    const float* tb = seq_table + sub * 4;
    int32_t* const lst = wave_lds + cur * lst_cap;
    // ---- DMA: this sample's candidate ids, the next sample's sequence ids + mask ------------------------------
    for (int c = 0; c * 64 < n_cand; ++c) {
      const int j = c * 64 + lane;
      if (j < n_cand)
        glds4(j < n_pos ? pos_ids + b * pos_stride + j : neg_ids + b * neg_stride + (j - n_pos),
              lds0 + (uint32_t)(2 * lst_cap + c * 64) * 4u);
    }
    if (b + bstep < B) prefetch_seq_ids(b + bstep, cur ^ 1);
    // ---- list of the slots that hold a real row, compacted in place; everything else is a zero row --------------
    const int32_t id_last = lst[S - 1];
    const float mask_last = msk[cur] != 0 ? 1.f : 0.f;
    int nr = 0;
    for (int c = 0; c * 64 < S; ++c) {
      const int j = c * 64 + lane;
      const int32_t id = j < S ? lst[j] : pad_id;
      const bool is_pad = id == pad_id || j >= S;
      const bool ok = !is_pad && (uint32_t)id < (uint32_t)seq_vocab;
      if (!ok && !is_pad && oob) *oob = 1;
      const uint64_t bal = __ballot(ok);
      if (ok) lst[nr + __popcll(bal & ((1ull << lane) - 1ull))] = id;
      nr += __popcll(bal);
    }
    if (nr == 0 && lane == 0) lst[0] = 0;          // the clamped fetches below need one valid row id
    const int nr1 = nr > 0 ? nr - 1 : 0;
    const bool last_ok = id_last != pad_id && (uint32_t)id_last < (uint32_t)seq_vocab;
    // ---- issue: last row, row batches 0 and 1 ---------------------------------------------------------------------
    float x_last = seq_table[(int64_t)(last_ok ? id_last : 0) * kD + lane];
    f32x4 kbuf[kNB][kU];
    // every issue is unconditional (rows past the end re-read the last real row: a cache hit): a conditional one makes
    // the landing buffers phi values and the register allocator answers with copies and spills
    auto issue_rows = [&](f32x4(&kr)[kU], int r0) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int r = r0 + 4 * u + grp;
        kr[u] = row_load<kRowsNT>(reinterpret_cast<const f32x4*>(tb + (int64_t)lst[r < nr ? r : nr1] * kD));
      }
    };
#pragma unroll
    for (int i = 0; i < kNA; ++i) issue_rows(kbuf[i], 4 * kU * i);
    REC_STAMP(2);   // ids compacted, first row batches issued
    // ---- last position: x -> q = x Wq + bq -> q_back = Wk q ---------------------------------------------------
    x_last = last_ok ? x_last : 0.f;
    xbuf[lane] = x_last;
    wave_lds_sync();
    const float q = matvec_pk<64, 64>(Wq, xbuf, lane, vec[lane]);
    xbuf[lane] = q;                       // the wave has finished reading x_last: its LDS operations are in order
    wave_lds_sync();
    const float q_back = matvec_pk<64, 64>(WkT, xbuf, lane, 0.f);
    xbuf[lane] = q_back;
    wave_lds_sync();
    const f32x4 qv = *reinterpret_cast<const f32x4*>(xbuf + sub * 4);     // elements 4 sub .. 4 sub + 3 (the row piece's)
    REC_STAMP(3);   // x_last arrived, q and q_back done
    // ---- attention over the raw rows of the real slots (online softmax per 16-lane group) ---------------------
    const bool masked = mask_last == 0.f;      // masked query row: all logits equal -> uniform over all S keys
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // rows past the end enter the softmax with the logit -inf (weight 0)
    auto reduce_rows = [&](f32x4(&kr)[kU], int r0) {
      float s[kU];
      float mb = m;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const f32x4 pr = kr[u] * qv;
        float d = group16_sum((pr.x + pr.y) + (pr.z + pr.w));
        d = masked ? 0.f : d * scale_log2e;
        s[u] = (r0 + 4 * u + grp < nr) ? d : -INFINITY;
        mb = fmaxf(mb, s[u]);
      }
      const float mbs = mb == -INFINITY ? 0.f : mb;      // a lane group without a real row yet: all weights 0
      const float sc = __builtin_amdgcn_exp2f(m - mbs);   // m = -inf -> 0
      acc *= sc;
      l *= sc;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const float p = __builtin_amdgcn_exp2f(s[u] - mbs);
        acc += kr[u] * p;
        l += p;
      }
      m = mb;
    };
    for (int r0 = 0; r0 < nr; r0 += 4 * kU * kNA) {
#pragma unroll
      for (int i = 0; i < kNA; ++i) {
        if (i == 0 || r0 + 4 * kU * i < nr) reduce_rows(kbuf[i], r0 + 4 * kU * i);
        issue_rows(kbuf[i], r0 + 4 * kU * (i + kNA));
      }
    }
    REC_STAMP(4);   // attention rows reduced
    // ---- candidate batches 0 and 1 leave now; they land while the Wv / LN / FFN chain runs -------------------
    const float* pos_t = n_pos > 0 ? pos_table : neg_table;     // slots past the end read row 0 of an existing table
    const float* neg_t = n_neg > 0 ? neg_table : pos_table;
    const int nc1 = n_cand - 1;
    auto cand_id = [&](int j) -> int32_t {     // id of candidate j, -1 when out of range or past the end
      const int32_t id = cbuf[j < n_cand ? j : nc1];
      const bool ok = j < n_cand && (uint32_t)id < (uint32_t)(j < n_pos ? pos_vocab : neg_vocab);
      return ok ? id : -1;
    };
    auto issue_cand = [&](f32x4(&kr)[kU], int jb) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int j = jb + 4 * u + grp;
        const int32_t id = cand_id(j);
        const float* tp = (j < n_pos ? pos_t : neg_t) + sub * 4;
        kr[u] = row_load<kRowsNT>(reinterpret_cast<const f32x4*>(tp + (int64_t)(id >= 0 ? id : 0) * kD));
      }
    };
    issue_cand(kbuf[0], 0);
    if constexpr (kNC > 1) issue_cand(kbuf[1 % kNB], 4 * kU);
    // merge the four group states
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
      f32x4 a2;
      a2.x = __shfl_xor(acc.x, o, 64);
      a2.y = __shfl_xor(acc.y, o, 64);
      a2.z = __shfl_xor(acc.z, o, 64);
      a2.w = __shfl_xor(acc.w, o, 64);
      const float mn = fmaxf(m, m2);
      const float s1 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mn);
      const float s2 = m2 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m2 - mn);
      acc = acc * s1 + a2 * s2;
      l = l * s1 + l2 * s2;
      m = mn;
    }
    // the S - nr zero rows: logit 0 each, value 0
    const int nz = S - nr;
    if (nz > 0) {
      const float mn = fmaxf(m, 0.f);
      const float s1 = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - mn);
      acc *= s1;
      l = l * s1 + (float)nz * __builtin_amdgcn_exp2f(-mn);
    }
    acc *= 1.f / l;
    // f32x4-per-sub layout -> the matvec's input vector (group 0 holds the merged state)
    if (grp == 0) *reinterpret_cast<f32x4*>(xbuf + sub * 4) = acc;
    wave_lds_sync();
    // ---- Wv, LN1, FFN, LN2 * mask -------------------------------------------------------------------------------
    const float att = matvec_pk<64, 64>(Wv, xbuf, lane, vec[64 + lane]);
    const float out1 = layer_norm64(x_last + att, vec[128 + lane], vec[192 + lane], P.eps1);
    float f = vec[256 + lane];
#pragma unroll
    for (int hb = 0; hb < FH / 64; ++hb) {      // one 64-float staging vector: out1 is re-staged per block of 64 hidden units
      xbuf[lane] = out1;
      wave_lds_sync();
      const float h = relu_nan(matvec_pk<64, FH>(W1, xbuf, hb * 64 + lane, vec[448 + hb * 64 + lane]));
      xbuf[lane] = h;
      wave_lds_sync();
      f = matvec_pk<64, 64>(W2 + hb * 64 * 64, xbuf, lane, f);
    }
    const float si = layer_norm64(out1 + f, vec[320 + lane], vec[384 + lane], P.eps2) * mask_last;
    if (seq_info) seq_info[b * kD + lane] = si;
    REC_STAMP(5);   // Wv, LN1, FFN, LN2 done
    // ---- candidates: logits[b, j] = table_j[id_j] . seq_info ----------------------------------------------------
    xbuf[lane] = si;
    wave_lds_sync();
    const f32x4 sv = *reinterpret_cast<const f32x4*>(xbuf + sub * 4);
    auto reduce_cand = [&](f32x4(&kr)[kU], int jb) {
      float mine = 0.f;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        float d = kr[u].x * sv.x;
        d = fmaf(kr[u].y, sv.y, d);
        d = fmaf(kr[u].z, sv.z, d);
        d = fmaf(kr[u].w, sv.w, d);
        d = group16_sum(d);
        mine = sub == u ? d : mine;
      }
      const int jw = jb + 4 * sub + grp;       // lanes sub < kU of group grp hold candidate jb + 4 sub + grp
      if (sub < kU && jw < n_cand) {
        const int32_t id = cbuf[jw];
        const bool ok = (uint32_t)id < (uint32_t)(jw < n_pos ? pos_vocab : neg_vocab);
        if (!ok && oob) *oob = 1;
        logits[b * logits_stride + jw] = ok ? mine : 0.f;      // out-of-range candidate: zero row
      }
    };
#pragma unroll
    for (int i = 2; i < kNC; ++i) issue_cand(kbuf[i], 4 * kU * i);   // the matvec chain's registers are free now
    for (int jb = 0; jb < n_cand; jb += 4 * kU * kNC) {
#pragma unroll
      for (int i = 0; i < kNC; ++i) {
        if (i == 0 || jb + 4 * kU * i < n_cand) reduce_cand(kbuf[i], jb + 4 * kU * i);
        issue_cand(kbuf[i], jb + 4 * kU * (i + kNC));
      }
    }
#ifdef REC_SASREC_STAMPS
    if (lane == 0) {
      stamps[6] = (long long)__builtin_amdgcn_s_memtime();
      if (seq_info) {
        for (int i = 1; i <= 6; ++i) seq_info[b * kD + i - 1] = (float)(stamps[i] - stamps[i - 1]);
        seq_info[b * kD + 6] = (float)nr;
        seq_info[b * kD + 7] = (float)(stamps[0] & 0xffffff);
        seq_info[b * kD + 8] = (float)(stamps[0] - t_entry);     // kernel entry -> this sample's start
      }
    }
#endif
  }
}

template <int FH>
size_t lds_bytes(int lst_cap, int cand_cap) {
  return (size_t)(3 * 64 * 64 + 2 * 64 * FH + 7 * 64 + FH + kWaves * 64) * 4 +
         (size_t)kWaves * (2 * lst_cap + cand_cap + 2) * 4;
}

}  // namespace
}  // namespace rec

using namespace rec;

extern "C" int rec_sasrec_last_row_supported(int32_t d, int32_t ffn_hidden, int32_t S, int32_t n_cand) {
  if (d != 64 || (ffn_hidden != 64 && ffn_hidden != 128) || S < 1 || n_cand < 1) return 0;
  const int lst_cap = (S + 3) & ~3, cand_cap = (n_cand + 3) & ~3;
  const size_t lds = ffn_hidden == 128 ? lds_bytes<128>(lst_cap, cand_cap) : lds_bytes<64>(lst_cap, cand_cap);
  return lds <= 160 * 1024 ? 1 : 0;
}

extern "C" int rec_sasrec_last_row_f32(const rec_sasrec_block* blk, const float* seq_table, int32_t seq_vocab,
                                       const int32_t* seq_ids, int64_t seq_ids_stride, int32_t S, int32_t pad_id,
                                       const int32_t* mask_ids, int64_t mask_stride, const float* pos_table,
                                       int32_t pos_vocab, const int32_t* pos_ids, int64_t pos_ids_stride, int32_t n_pos,
                                       const float* neg_table, int32_t neg_vocab, const int32_t* neg_ids,
                                       int64_t neg_ids_stride, int32_t n_neg, int64_t B, int32_t d, float* seq_info,
                                       float* logits, int64_t logits_stride, int32_t* oob_flag, void* stream) {
  REC_CHECK_ARG(blk && seq_table && seq_ids && mask_ids && logits, REC_EINVAL, "sasrec_last_row: NULL argument");
  REC_CHECK_ARG(blk->wq && blk->bq && blk->wk && blk->wv && blk->bv && blk->ln1_gamma && blk->ln1_beta && blk->w1 &&
                    blk->b1 && blk->w2 && blk->b2 && blk->ln2_gamma && blk->ln2_beta,
                REC_EINVAL, "sasrec_last_row: NULL weight pointer in rec_sasrec_block");
  REC_CHECK_ARG(d == 64, REC_ENOTIMPL, "sasrec_last_row: d_model = %d (this kernel: 64)", d);
  REC_CHECK_ARG(blk->ffn_hidden == 64 || blk->ffn_hidden == 128, REC_ENOTIMPL,
                "sasrec_last_row: ffn_hidden = %d (this kernel: 64 or 128)", blk->ffn_hidden);
  REC_CHECK_ARG(S >= 1, REC_ESHAPE, "sasrec_last_row: S = %d", S);
  REC_CHECK_ARG(B >= 0 && n_pos >= 0 && n_neg >= 0 && n_pos + n_neg >= 1 && seq_vocab > 0, REC_ESHAPE,
                "sasrec_last_row: negative size or no candidate");
  REC_CHECK_ARG(n_pos == 0 || (pos_table && pos_ids && pos_vocab > 0), REC_EINVAL, "sasrec_last_row: pos table/ids NULL");
  REC_CHECK_ARG(n_neg == 0 || (neg_table && neg_ids && neg_vocab > 0), REC_EINVAL, "sasrec_last_row: neg table/ids NULL");
  REC_CHECK_ARG(seq_ids_stride >= S && pos_ids_stride >= n_pos && neg_ids_stride >= n_neg &&
                    logits_stride >= n_pos + n_neg,
                REC_ESHAPE, "sasrec_last_row: a row stride is smaller than its row");
  REC_CHECK_ARG(aligned16(seq_table) && (!pos_table || aligned16(pos_table)) && (!neg_table || aligned16(neg_table)) &&
                    aligned16(blk->wq) && aligned16(blk->wv) && aligned16(blk->w1) && aligned16(blk->w2),
                REC_EINVAL, "sasrec_last_row: tables and weight matrices must be 16-byte aligned");
  if (B == 0) return REC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  SasrecParams P{blk->wq, blk->bq, blk->wk, blk->wv, blk->bv, blk->ln1_gamma, blk->ln1_beta, blk->w1, blk->b1,
                 blk->w2, blk->b2, blk->ln2_gamma, blk->ln2_beta, blk->ln1_eps, blk->ln2_eps, blk->ffn_hidden};
  const int lst_cap = (S + 3) & ~3;                       // the id DMAs only touch slots < S / < n_cand
  const int cand_cap = (n_pos + n_neg + 3) & ~3;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      set_error("sasrec_last_row: cannot query the device");
      return REC_EHIP;
    }
    cus = prop.multiProcessorCount;
  }
  const int64_t wgs = (B + kWaves - 1) / kWaves;
  const int grid = (int)(wgs < cus ? wgs : cus);
  auto launch = [&](auto kern, size_t lds) -> int {
    if (lds > 160 * 1024) {
      set_error("sasrec_last_row: S = %d with %d candidates needs %zu bytes of LDS (> 160 KiB)", S, n_pos + n_neg, lds);
      return REC_ENOTIMPL;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      set_error("sasrec_last_row: cannot raise the dynamic LDS limit to %zu bytes", lds);
      return REC_EHIP;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWaves * 64), lds, st, P, seq_table, seq_vocab, seq_ids, seq_ids_stride,
                       (int)S, pad_id, mask_ids, mask_stride, pos_table, pos_vocab, pos_ids, pos_ids_stride, (int)n_pos,
                       neg_table, neg_vocab, neg_ids, neg_ids_stride, (int)n_neg, B, seq_info, logits, logits_stride,
                       reinterpret_cast<int*>(oob_flag), lst_cap, cand_cap);
    return REC_OK;
  };
  int rc = blk->ffn_hidden == 128 ? launch(sasrec_last_row_kernel<128>, lds_bytes<128>(lst_cap, cand_cap))
                                  : launch(sasrec_last_row_kernel<64>, lds_bytes<64>(lst_cap, cand_cap));
  if (rc != REC_OK) return rc;
  REC_CHECK_LAUNCH("sasrec_last_row");
  return REC_OK;
}
