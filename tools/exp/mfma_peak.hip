// round 3: what the matrix pipe sustains on this chip for v_mfma_f32_32x32x16_bf16 — the Dense kernel's instruction — when
// nothing else is in the way, and with the Dense kernel's LDS operand traffic beside it.  Prints TFLOP/s (bf16 MFMA
// flops), the shader clock implied by s_memtime (ticks / wall time) and the pipe's busy share at that clock.
//   mode 0: MFMAs on register operands only (4 independent accumulators, 24 MFMAs per "k-step")
//   mode 1: + 12 ds_read_b128 per 24 MFMAs (the 64 x 64 wave tile of csrc/dense_bf16x3.hip)
//   mode 2: + 9 ds_read_b128 per 24 MFMAs (a 128 x 64 wave tile's rate: 18 per 48)
//   mode 3: mode 1 + 6 ds_write_b128 per thread per k-step and one barrier (the kernel's full LDS traffic, no global loads)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd=3] [ksteps=2048]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* ticks, int ksteps) {
  __shared__ u32x4 frag[2][2][3][2][128];   // 48 KiB, as the Dense kernel
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l32 = lane & 31, half = lane >> 5, wm = wv >> 1, wn = wv & 1;
  for (int i = tid; i < 2 * 2 * 3 * 2 * 128; i += 256) (&frag[0][0][0][0][0])[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 a[2][3], b[2][3];
  for (int t = 0; t < 2; ++t)
    for (int p = 0; p < 3; ++p) {
      a[t][p] = __builtin_bit_cast(bf16x8, frag[0][0][p][half][wm * 64 + t * 32 + l32]);
      b[t][p] = __builtin_bit_cast(bf16x8, frag[0][1][p][half][wn * 64 + t * 32 + l32]);
    }
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int ks = 0; ks < ksteps; ++ks) {
    const int st = ks & 1;
    asm volatile("" ::: "memory");   // the LDS reads stay in the loop
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[t][p] = __builtin_bit_cast(bf16x8, frag[st][0][p][half][wm * 64 + t * 32 + l32]);
          b[t][p] = __builtin_bit_cast(bf16x8, frag[st][1][p][half][wn * 64 + t * 32 + l32]);
        }
    } else if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[t][p] = __builtin_bit_cast(bf16x8, frag[st][0][p][half][wm * 64 + t * 32 + l32]);
#pragma unroll
      for (int p = 0; p < 3; ++p) b[0][p] = __builtin_bit_cast(bf16x8, frag[st][1][p][half][wn * 64 + l32]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
        acc[i][j] = c;
      }
    if (MODE == 3) {
      const int srow = tid & 127, skh = tid >> 7;
      const u32x4 v = __builtin_bit_cast(u32x4, a[0][0]);
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int p = 0; p < 3; ++p) frag[st ^ 1][o][p][skh][srow] = v;
      __syncthreads();
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(int wps, int ksteps, float* out, long long* ticks, int nblk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(256), 0, 0, out, ticks, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(256), 0, 0, out, ticks, ksteps);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nblk);
  hipMemcpy(h.data(), ticks, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
  double mean = 0;
  for (long long v : h) mean += (double)v;
  mean /= nblk;
  const double flops = (double)nblk * 4 * ksteps * 24 * 2.0 * 32 * 32 * 16;
  const double ghz = mean / (ms * 1e6);                 // all workgroups resident at once: a wave's ticks span the kernel
  const double pipe_cycles = (double)wps * ksteps * 24 * 32;   // matrix-pipe cycles a SIMD owes (8 passes x 4 clk per MFMA)
  printf("mode %d: %.3f ms  %.1f TFLOP/s bf16  (= %.1f TFLOP/s fp32-equivalent at 6 MFMAs per product)  clock ~%.2f GHz  pipe busy %.3f\n",
         MODE, ms, flops / ms / 1e9, flops / ms / 1e9 / 6, ghz, pipe_cycles / mean);
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 3;       // workgroups per CU = waves per SIMD
  const int ksteps = argc > 2 ? atoi(argv[2]) : 2048;
  const int nblk = 256 * wps;
  float* out;
  long long* ticks;
  hipMalloc(&out, sizeof(float) * 256 * nblk);
  hipMalloc(&ticks, sizeof(long long) * nblk);
  printf("%d workgroups (%d per CU), %d k-steps of 24 MFMAs per wave\n", nblk, wps, ksteps);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(wps, ksteps, out, ticks, nblk);
    run<1>(wps, ksteps, out, ticks, nblk);
    run<2>(wps, ksteps, out, ticks, nblk);
    run<3>(wps, ksteps, out, ticks, nblk);
  }
  return 0;
}
