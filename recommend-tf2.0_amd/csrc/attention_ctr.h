// Shared declarations of the AutoInt interacting-layer stack (attention_ctr.hip) and its C entry points (attention.hip).
#pragma once
#include "common.h"

namespace rec {

struct CtrStackArgs {
  const float* Wq[4];
  const float* Wk[4];
  const float* Wv[4];
  const float* W0[4];
};

// AutoInt.call around the stack (src/ctr/autoint/model.py:46-55) folded into the same launch: the field tensor is read
// straight from its sources — fields < n_sparse are embedding rows fetched by id (:46), the others dense values times
// their embedding rows (:47-50) — and the flattened output meets its Dense(1) + sigmoid (:54-55) in the wave's
// registers, so neither the (B, N, din) input nor the (B, N, 16 H) output exists in memory.
struct CtrFusedIo {
  TableSet ts;                 // n_sparse tables of width din
  const int32_t* ids;          // (B, n_sparse)
  int64_t ids_stride;
  int32_t n_sparse;
  const float* dense;          // (B, N - n_sparse) values
  int64_t dense_stride;
  const float* dense_embed;    // (N - n_sparse, din)
  const float* head_w;         // (N * 16 H)
  const float* head_b;         // (1) or NULL
  float* head_out;             // (B)
  int* oob;
};

// returns false when the stack is not covered (the caller runs the layers one by one); io == nullptr: plain stack
bool mha_ctr_stack_dispatch(const float* x, int64_t B, int N, int din, const CtrStackArgs& wa, int L, int H, int S, int act,
                            float* out, hipStream_t st, const CtrFusedIo* io);

}  // namespace rec
