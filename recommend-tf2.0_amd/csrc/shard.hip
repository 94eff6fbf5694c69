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

// ---- resolve: where does each lookup read its row? (before the exchange) ---------------------------------------
// The consumers address ONE row space [this rank's shard | hot-row replica cache | rows returned by the exchange];
// resolve turns a virtual id v into a row of that space:
//   v < 0                         -> -1 (zero row; the consumer raises the REQUESTER's oob flag)
//   me >= 0 and v % G == me       -> v / G: the row is read in place from this rank's shard (never sent to itself)
//   cache_slot[v] >= 0            -> cache_base + cache_slot[v]: a replica of a hot remote row (exact copy)
//   otherwise                     -> recv_base + position of the row in the buffer the exchange returns
// Exact per-lookup de-duplication of the remote rows: rep[v] (one int32 per virtual row, INT32_MAX between calls)
// receives the smallest lookup index that asks for row v — a deterministic representative.  Representatives enter
// the send list, every lookup then reads the returned row of its representative.
// `first[i]` carries the classification between the kernels: >= 0 representative lookup, -1 zero row,
// <= -2 direct row (-2 - row).
__device__ __forceinline__ bool is_local(int32_t v, int G, int me) { return me >= 0 && (v % G) == me; }

__global__ __launch_bounds__(256) void shard_first_kernel(const int32_t* __restrict__ vids, int64_t n, int G, int me,
                                                          const int32_t* __restrict__ cache_slot,
                                                          int32_t* __restrict__ rep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t v = vids[i];
  if (v < 0 || is_local(v, G, me)) return;
  if (cache_slot && cache_slot[v] >= 0) return;
  atomicMin(&rep[v], (int32_t)i);
}

__global__ __launch_bounds__(256) void shard_uniq_kernel(const int32_t* __restrict__ vids, int64_t n, int G, int me,
                                                         const int32_t* __restrict__ rep,
                                                         const int32_t* __restrict__ cache_slot,
                                                         int32_t* __restrict__ hot_count, int32_t cache_base,
                                                         unsigned long long* __restrict__ stat,
                                                         int32_t* __restrict__ first, int32_t* __restrict__ uniq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int cls = 0;  // 1 local, 2 cached
  if (i < n) {
    const int32_t v = vids[i];
    int32_t f = -1, u = kSkip;
    if (v >= 0) {
      if (is_local(v, G, me)) {
        f = -2 - v / G;
        cls = 1;
      } else {
        if (hot_count) atomicAdd(&hot_count[v], 1);  // integer atomics: order-independent
        const int32_t cs = cache_slot ? cache_slot[v] : -1;
        if (cs >= 0) {
          f = -2 - (cache_base + cs);
          cls = 2;
        } else {
          f = rep ? rep[v] : (int32_t)i;
          u = (f == (int32_t)i) ? v : kSkip;
        }
      }
    }
    first[i] = f;
    uniq[i] = u;
  }
  if (stat) {  // lookups served without the exchange, accumulated across calls (reporting only)
    const unsigned long long ml = __ballot(cls == 1), mc = __ballot(cls == 2);
    if ((threadIdx.x & 63) == 0) {
      if (ml) atomicAdd(&stat[0], (unsigned long long)__popcll(ml));
      if (mc) atomicAdd(&stat[1], (unsigned long long)__popcll(mc));
    }
  }
}

__global__ __launch_bounds__(256) void shard_uidx_kernel(const int32_t* __restrict__ vids, int64_t n,
                                                         const int32_t* __restrict__ first,
                                                         const int32_t* __restrict__ perm, int32_t recv_base,
                                                         int32_t* __restrict__ rep, int32_t* __restrict__ uidx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int32_t f = first[i];
  uidx[i] = f >= 0 ? recv_base + perm[f] : (f == -1 ? -1 : -2 - f);
  // leave the table clean for the next call (duplicates write the same value; only remote lookups touched it)
  if (rep && f >= 0) rep[vids[i]] = INT32_MAX;
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
// device steps of rec_shard_plan_ids / rec_shard_resolve_i32: classification + dedup (rep may be NULL) + stable
// bucketing of the representatives.  ws_hist: rec_shard_bucket_workspace_bytes(n, G).
int shard_plan_device(const int32_t* vids, int64_t n, int32_t G, int32_t me, int32_t* rep, const int32_t* cache_slot,
                      int32_t* hot_count, int32_t cache_base, int32_t recv_base, uint64_t* stat, int32_t* first,
                      int32_t* uniq, int32_t* perm, int32_t* uidx, int32_t* send_local, int32_t* counts, void* ws_hist,
                      hipStream_t st) {
  const char* who = "rec_shard_plan_ids";
  if (n == 0) {
    hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * G, st);
    REC_CHECK_ARG(e == hipSuccess, REC_EHIP, "%s: memset: %s", who, hipGetErrorString(e));
    return REC_OK;
  }
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (rep) {
    hipLaunchKernelGGL(shard_first_kernel, dim3(nb), dim3(256), 0, st, vids, n, G, me, cache_slot, rep);
    REC_CHECK_LAUNCH(who);
  }
  hipLaunchKernelGGL(shard_uniq_kernel, dim3(nb), dim3(256), 0, st, vids, n, G, me, (const int32_t*)rep, cache_slot,
                     hot_count, cache_base, reinterpret_cast<unsigned long long*>(stat), first, uniq);
  REC_CHECK_LAUNCH(who);
  int rc = rec_shard_bucket_i32(uniq, n, G, counts, perm, send_local, ws_hist, st);
  if (rc != REC_OK) return rc;
  hipLaunchKernelGGL(shard_uidx_kernel, dim3(nb), dim3(256), 0, st, vids, n, (const int32_t*)first,
                     (const int32_t*)perm, recv_base, rep, uidx);
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

extern "C" int rec_shard_resolve_i32(const int32_t* vids, int64_t n, int32_t G, int32_t me, int32_t* rep_table,
                                     const int32_t* cache_slot, int32_t* hot_count, int32_t cache_base,
                                     int32_t recv_base, uint64_t* stat, int32_t* first, int32_t* uniq, int32_t* perm,
                                     int32_t* uidx, int32_t* send_local, int32_t* counts, void* workspace,
                                     void* stream) {
  const char* who = "rec_shard_resolve_i32";
  REC_CHECK_ARG(G >= 1 && G <= kMaxG && n >= 0 && n <= 0x7fffffffLL && me < G, REC_ESHAPE, "%s: n=%lld G=%d me=%d", who,
                (long long)n, G, me);
  REC_CHECK_ARG(cache_base >= 0 && recv_base >= 0, REC_ESHAPE, "%s: negative row base", who);
  REC_CHECK_ARG(counts && workspace && (n == 0 || (vids && first && uniq && perm && uidx && send_local)), REC_EINVAL,
                "%s: NULL pointer", who);
  return rec::shard_plan_device(vids, n, G, me, rep_table, cache_slot, hot_count, cache_base, recv_base, stat, first,
                                uniq, perm, uidx, send_local, counts, workspace, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int rec_shard_dedup_bucket_i32(const int32_t* vids, int64_t n, int32_t G, int32_t* rep_table, int32_t* first,
                                          int32_t* uniq, int32_t* perm, int32_t* uidx, int32_t* send_local,
                                          int32_t* counts, void* workspace, void* stream) {
  // every row through the exchange, rows counted from 0: resolve without a local shard, a cache or a row-space offset
  return rec_shard_resolve_i32(vids, n, G, -1, rep_table, nullptr, nullptr, 0, 0, nullptr, first, uniq, perm, uidx,
                               send_local, counts, workspace, stream);
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
