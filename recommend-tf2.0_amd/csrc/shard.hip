// C2 — device-side helpers of the row-sharded lookup (owner = id % G, local row = id / G).
// The exchange itself is an RCCL all-to-all issued by the host (recamd/dist.py) over xGMI; these
// kernels build the owner-sorted send buffer (a STABLE counting sort, so results are reproducible
// and identical to a CPU simulation) and un-permute the returned rows.
//
// Bucketing, 3 launches over n ids in chunks of 1024:
//   1. per-chunk histogram over the G owners                (ballot + popcount per wave)
//   2. exclusive scan of the (owner-major, chunk-minor) histogram -> chunk base offsets, counts[G]
//   3. scatter: stable rank inside the chunk (wave ballots in index order) + base offset
// Out-of-range (negative) ids are sent to owner 0 with local row -1 (the owner's gather returns
// a zero row and raises its oob flag).
#include "common.h"

namespace rec {

constexpr int kChunk = 1024;  // ids per block: 4 rounds x 256 threads, index order = round, thread
constexpr int kMaxG = 64;

constexpr int32_t kSkip = INT32_MIN;  // "not in the send list": duplicates and (dedup path) out-of-range ids

__device__ __forceinline__ void owner_of(int32_t id, int G, int& owner, int32_t& local) {
  if (id == kSkip) {
    owner = -1;
    local = -1;
  } else if (id < 0) {
    owner = 0;
    local = -1;
  } else {
    owner = id % G;
    local = id / G;
  }
}

__global__ __launch_bounds__(256) void shard_hist_kernel(const int32_t* __restrict__ ids, int64_t n,
                                                         int G, int32_t* __restrict__ hist /*[G][nchunks]*/,
                                                         int64_t nchunks) {
  __shared__ int32_t h[kMaxG];
  if ((int)threadIdx.x < G) h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < n) {
      int o;
      int32_t l;
      owner_of(ids[i], G, o, l);
      if (o >= 0) atomicAdd(&h[o], 1);  // LDS integer atomics: order-independent, deterministic result
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) hist[(int64_t)threadIdx.x * nchunks + blockIdx.x] = h[threadIdx.x];
}

// single block: exclusive scan over G*nchunks entries in (owner, chunk) order
__global__ __launch_bounds__(1024) void shard_scan_kernel(int32_t* __restrict__ hist, int64_t total,
                                                          int G, int64_t nchunks,
                                                          int32_t* __restrict__ counts) {
  __shared__ int32_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (total + 1023) / 1024;
  const int64_t lo = t * per, hi = lo + per < total ? lo + per : total;
  int32_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += hist[i];
  part[t] = s;
  __syncthreads();
  // Hillis-Steele inclusive scan of the 1024 partials
  for (int off = 1; off < 1024; off <<= 1) {
    int32_t v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int32_t run = t ? part[t - 1] : 0;
  for (int64_t i = lo; i < hi; ++i) {
    const int32_t c = hist[i];
    hist[i] = run;
    run += c;
  }
  __syncthreads();
  if (t < G) {
    // counts[g] = start[g+1] - start[g]; start of owner g = scanned hist[g*nchunks]
    const int32_t start = hist[(int64_t)t * nchunks];
    const int32_t end = t + 1 < G ? hist[(int64_t)(t + 1) * nchunks] : part[1023];
    counts[t] = end - start;
  }
}

__global__ __launch_bounds__(256) void shard_scatter_kernel(const int32_t* __restrict__ ids, int64_t n,
                                                            int G, const int32_t* __restrict__ base /*scanned hist*/,
                                                            int64_t nchunks, int32_t* __restrict__ perm,
                                                            int32_t* __restrict__ send_local) {
  __shared__ int32_t run[kMaxG];        // running offset of each owner inside this chunk
  __shared__ int32_t wcnt[4][kMaxG];    // per-wave counts of the current round
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  if ((int)threadIdx.x < G) run[threadIdx.x] = base[(int64_t)threadIdx.x * nchunks + blockIdx.x];
  __syncthreads();
  const int64_t cbase = (int64_t)blockIdx.x * kChunk;
  for (int r = 0; r < 4; ++r) {
    const int64_t i = cbase + r * 256 + threadIdx.x;
    int o = -1;
    int32_t l = 0;
    if (i < n) owner_of(ids[i], G, o, l);
    int rank_in_wave = 0;
    for (int g = 0; g < G; ++g) {
      const unsigned long long m = __ballot(o == g);
      if (o == g) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wv][g] = __popcll(m);
    }
    __syncthreads();
    if (o >= 0) {
      int32_t pos = run[o] + rank_in_wave;
      for (int w2 = 0; w2 < wv; ++w2) pos += wcnt[w2][o];
      perm[i] = pos;
      send_local[pos] = l;
    } else if (i < n) {
      perm[i] = -1;  // skipped id: not in the send list
    }
    __syncthreads();
    if ((int)threadIdx.x < G)
      run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] +
                          wcnt[3][threadIdx.x];
    __syncthreads();
  }
}

// ---- exact per-lookup de-duplication (before the exchange) -----------------------------------------
// rep[v] (one int32 per virtual row, INT32_MAX between calls) receives the smallest lookup index that asks for
// row v: deterministic representative.  Representatives enter the send list, every lookup then reads the returned
// row of its representative (uidx).  Out-of-range ids (negative after the caller's range check) are not sent at
// all: uidx = -1, the consumer reads a zero row and raises the REQUESTER's oob flag.
__global__ __launch_bounds__(256) void shard_first_kernel(const int32_t* __restrict__ vids, int64_t n,
                                                          int32_t* __restrict__ rep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t v = vids[i];
  if (v >= 0) atomicMin(&rep[v], (int32_t)i);
}

__global__ __launch_bounds__(256) void shard_uniq_kernel(const int32_t* __restrict__ vids, int64_t n,
                                                         const int32_t* __restrict__ rep, int32_t* __restrict__ first,
                                                         int32_t* __restrict__ uniq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t v = vids[i];
  const int32_t f = v >= 0 ? (rep ? rep[v] : (int32_t)i) : -1;
  first[i] = f;
  uniq[i] = (f == (int32_t)i) ? v : kSkip;
}

__global__ __launch_bounds__(256) void shard_uidx_kernel(const int32_t* __restrict__ vids, int64_t n,
                                                         const int32_t* __restrict__ first,
                                                         const int32_t* __restrict__ perm, int32_t* __restrict__ rep,
                                                         int32_t* __restrict__ uidx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t f = first[i];
  uidx[i] = f >= 0 ? perm[f] : -1;
  const int32_t v = vids[i];
  if (rep && v >= 0) rep[v] = INT32_MAX;  // leave the table clean for the next call (duplicates write the same value)
}

// a += b (the in-process test transport's all-reduce)
__global__ __launch_bounds__(256) void vec_add_kernel(float* __restrict__ a, const float* __restrict__ b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] += b[i];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[i, :] = rows[perm[i], :]; LPR = D/4 lanes per row when vectorisable
__global__ __launch_bounds__(256) void unpermute_kernel(const float* __restrict__ rows,
                                                        const int32_t* __restrict__ perm, int64_t n,
                                                        int D, int vec, float* __restrict__ out,
                                                        int64_t out_stride) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    const int lpr = D >> 2;
    const int64_t i = t / lpr;
    const int c = (int)(t - i * lpr);
    if (i >= n) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(rows + (int64_t)perm[i] * D)[c];
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + i * out_stride) + c);
  } else {
    const int64_t i = t / D;
    const int c = (int)(t - i * D);
    if (i >= n) return;
    out[i * out_stride + c] = rows[(int64_t)perm[i] * D + c];
  }
}

}  // namespace rec

using namespace rec;

static int64_t nchunks_of(int64_t n) { return (n + kChunk - 1) / kChunk; }

extern "C" int64_t rec_shard_bucket_workspace_bytes(int64_t n, int32_t G) {
  if (n < 0 || G < 1) return 0;
  return (int64_t)sizeof(int32_t) * G * (nchunks_of(n) > 0 ? nchunks_of(n) : 1);
}

extern "C" int rec_shard_bucket_i32(const int32_t* ids, int64_t n, int32_t G, int32_t* counts,
                                    int32_t* perm, int32_t* send_local, void* workspace,
                                    void* stream) {
  const char* who = "rec_shard_bucket_i32";
  REC_CHECK_ARG(G >= 1 && G <= kMaxG && n >= 0 && n <= 0x7fffffffLL, REC_ESHAPE, "%s: n=%lld G=%d",
                who, (long long)n, G);
  REC_CHECK_ARG(counts && workspace && (n == 0 || (ids && perm && send_local)), REC_EINVAL,
                "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n == 0) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * G, st);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: memset: %s", who, hipGetErrorString(e));
    return REC_OK;
  }
  const int64_t nch = nchunks_of(n);
  int32_t* hist = static_cast<int32_t*>(workspace);
  hipLaunchKernelGGL(shard_hist_kernel, dim3((unsigned)nch), dim3(256), 0, st, ids, n, G, hist, nch);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(shard_scan_kernel, dim3(1), dim3(1024), 0, st, hist, (int64_t)G * nch, G, nch,
                     counts);
  REC_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(shard_scatter_kernel, dim3((unsigned)nch), dim3(256), 0, st, ids, n, G, hist, nch,
                     perm, send_local);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

namespace rec {
// device steps of rec_shard_plan_ids (shard_exchange.cpp): dedup (rep may be NULL) + stable bucketing of the
// representatives.  ws_hist: rec_shard_bucket_workspace_bytes(n, G).
int shard_plan_device(const int32_t* vids, int64_t n, int32_t G, int32_t* rep, int32_t* first, int32_t* uniq,
                      int32_t* perm, int32_t* uidx, int32_t* send_local, int32_t* counts, void* ws_hist,
                      hipStream_t st) {
  const char* who = "rec_shard_plan_ids";
  if (n == 0) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * G, st);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: memset: %s", who, hipGetErrorString(e));
    return REC_OK;
  }
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (rep) {
    hipLaunchKernelGGL(shard_first_kernel, dim3(nb), dim3(256), 0, st, vids, n, rep);
    REC_CHECK_LAUNCH(who);
  }
  hipLaunchKernelGGL(shard_uniq_kernel, dim3(nb), dim3(256), 0, st, vids, n, (const int32_t*)rep, first, uniq);
  REC_CHECK_LAUNCH(who);
  int rc = rec_shard_bucket_i32(uniq, n, G, counts, perm, send_local, ws_hist, st);
  if (rc != REC_OK) return rc;
  hipLaunchKernelGGL(shard_uidx_kernel, dim3(nb), dim3(256), 0, st, vids, n, (const int32_t*)first,
                     (const int32_t*)perm, rep, uidx);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}

int shard_vec_add(float* a, const float* b, int64_t n, hipStream_t st) {
  if (n <= 0) return REC_OK;
  hipLaunchKernelGGL(vec_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, n);
  REC_CHECK_LAUNCH("rec_comm_allreduce_sum_f32");
  return REC_OK;
}
}  // namespace rec

extern "C" int rec_shard_dedup_bucket_i32(const int32_t* vids, int64_t n, int32_t G, int32_t* rep_table, int32_t* first,
                                          int32_t* uniq, int32_t* perm, int32_t* uidx, int32_t* send_local,
                                          int32_t* counts, void* workspace, void* stream) {
  const char* who = "rec_shard_dedup_bucket_i32";
  REC_CHECK_ARG(G >= 1 && G <= kMaxG && n >= 0 && n <= 0x7fffffffLL, REC_ESHAPE, "%s: n=%lld G=%d", who, (long long)n, G);
  REC_CHECK_ARG(counts && workspace && (n == 0 || (vids && first && uniq && perm && uidx && send_local)), REC_EINVAL,
                "%s: NULL pointer", who);
  return rec::shard_plan_device(vids, n, G, rep_table, first, uniq, perm, uidx, send_local, counts, workspace,
                                reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rec_unpermute_rows_f32(const float* rows, const int32_t* perm, int64_t n, int32_t D,
                                      float* out, int64_t out_stride, void* stream) {
  const char* who = "rec_unpermute_rows_f32";
  REC_CHECK_ARG(n >= 0 && D >= 1 && out_stride >= D, REC_ESHAPE, "%s: n=%lld D=%d", who, (long long)n, D);
  if (n == 0) return REC_OK;
  REC_CHECK_ARG(rows && perm && out, REC_EINVAL, "%s: NULL pointer", who);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int vec = D % 4 == 0 && aligned16(rows) && aligned16(out) && out_stride % 4 == 0;
  const int64_t threads = vec ? n * (D / 4) : n * (int64_t)D;
  hipLaunchKernelGGL(unpermute_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, rows,
                     perm, n, D, vec, out, out_stride);
  REC_CHECK_LAUNCH(who);
  return REC_OK;
}
