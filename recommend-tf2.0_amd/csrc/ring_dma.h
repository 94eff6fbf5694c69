// LDS-DMA (global_load_lds) and counted-vmcnt helpers shared by the ring kernels (pairwise_dot_ring.hip: the tuned
// 27 x 128 headline shape; pairwise_dot_ring_gen.hip: D in {64, 128, 256}, 17 <= n <= 32).  gfx950 only.
#pragma once
#include "common.h"

namespace rec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA burst, 16 B per lane and piece: LDS[lds + 1024*t + 16*lane] <- *g[t], t < NP, back to back (one asm
// statement: nothing is scheduled between the pieces, M0 = destination base is stepped in place; the s_nop 0 is the
// SALU-writes-M0 -> LDS-DMA wait state).  NT: streaming policy for rows that are read once.
// LM: cache policy of the row loads — 0 default, 1 nt (streaming; shipped); experiment builds: 2 sc1 nt, 3 sc0 sc1 nt,
// 4 sc1, 5 sc0 sc1
#define REC_BURST_PIECE_(i, SUF) "s_nop 0\n\tglobal_load_lds_dwordx4 %" #i ", off" SUF "\n\ts_add_u32 m0, m0, 0x400\n\t"
#define REC_BURST14_(SUF)                                                                                              \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\t" REC_BURST_PIECE_(2, SUF) REC_BURST_PIECE_(3, SUF)             \
                   REC_BURST_PIECE_(4, SUF) REC_BURST_PIECE_(5, SUF) REC_BURST_PIECE_(6, SUF) REC_BURST_PIECE_(7, SUF)  \
                       REC_BURST_PIECE_(8, SUF) REC_BURST_PIECE_(9, SUF) REC_BURST_PIECE_(10, SUF)                      \
                           REC_BURST_PIECE_(11, SUF) REC_BURST_PIECE_(12, SUF) REC_BURST_PIECE_(13, SUF)                \
                               REC_BURST_PIECE_(14, SUF) "s_nop 0\n\tglobal_load_lds_dwordx4 %15, off" SUF "\n\t"      \
                                                         "s_mov_b32 m0, %0"                                            \
               : "=&s"(keep)                                                                                           \
               : "s"(lds), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]),     \
                 "v"(g[8]), "v"(g[9]), "v"(g[10]), "v"(g[11]), "v"(g[12]), "v"(g[13])                                  \
               : "memory")
#define REC_BURST1_(SUF)                                                                                               \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" SUF               \
               "\n\ts_mov_b32 m0, %0"                                                                                  \
               : "=&s"(keep)                                                                                           \
               : "v"(g[t]), "s"(lds + 1024u * t)                                                                       \
               : "memory")
template <int NP, int LM>
__device__ __forceinline__ void glds16_burst(const uint64_t (&g)[16], uint32_t lds) {
  static_assert(NP >= 1 && NP <= 16, "");
  // operand 0 = saved M0, 1 = lds base, 2.. = piece addresses
  if constexpr (NP == 14) {
    unsigned keep;
    if constexpr (LM == 1) REC_BURST14_(" nt");
    else if constexpr (LM == 0) REC_BURST14_("");
#ifdef REC_RING_EXPERIMENTS
    else if constexpr (LM == 2) REC_BURST14_(" sc1 nt");
    else if constexpr (LM == 3) REC_BURST14_(" sc0 sc1 nt");
    else if constexpr (LM == 4) REC_BURST14_(" sc1");
    else REC_BURST14_(" sc0 sc1");
#endif
  } else {
    // other row counts: one statement per piece (same instructions, the compiler may schedule between them)
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      unsigned keep;
      if constexpr (LM == 0) REC_BURST1_("");
      else REC_BURST1_(" nt");
    }
  }
}
#undef REC_BURST1_
#undef REC_BURST14_
#undef REC_BURST_PIECE_
// LDS-DMA, 4 B per lane
__device__ __forceinline__ void glds4(const void* g, uint32_t lds) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds)
      : "memory");
}
__device__ __forceinline__ void gstore16_nt(void* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void gstore16(void* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// result stores with scope bits (POL bits 2-4).  Shipped: sc0 sc1 = system-scope write-through — the 126 MB of results
// leave the L2 as they are written instead of sitting there as dirty lines that are evicted in bursts between the
// streaming row reads: 184.8 -> 176 us on one box, 174.6 -> 168 us on another (profiles/r02_ring_store_ab.txt);
// sc1 alone 177; sc0 alone = plain; any of them with nt 188.
template <int SP>
__device__ __forceinline__ void gstore16_scope(void* p, f32x4 v) {
  if constexpr (SP == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else if constexpr (SP == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else if constexpr (SP == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else if constexpr (SP == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
#define REC_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(n) : "memory")
#define REC_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory")

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)p;
}

}  // namespace rec
