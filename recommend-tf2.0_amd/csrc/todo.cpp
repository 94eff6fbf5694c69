// Entry points declared in include/recamd.h that are not implemented yet: they fail loudly with
// REC_ENOTIMPL (never a silent fallback).  Each one moves to its own .hip file when written.
#include "common.h"
#define REC_TODO(name) rec::set_error(name ": not implemented in this build"); return REC_ENOTIMPL

extern "C" {
int64_t rec_fm_layer_workspace_floats(int64_t) { return 0; }
int rec_fm_layer_f32(const float*, int64_t, int32_t, const float*, const float*, int64_t, int32_t, int64_t, float*, float*, void*) { REC_TODO("rec_fm_layer_f32"); }
int rec_cross_f32(const float*, int64_t, int32_t, const float*, const float*, int32_t, int64_t, float*, int64_t, void*) { REC_TODO("rec_cross_f32"); }
int rec_fm_onehot_f32(const float*, int64_t, int32_t, const int32_t*, int64_t, int32_t, const int64_t*, const float*, const float*, const float*, int32_t, int64_t, float*, void*) { REC_TODO("rec_fm_onehot_f32"); }
int rec_dense_f32(const float*, int64_t, const float*, const float*, const float*, int32_t, int64_t, int32_t, int32_t, float*, int64_t, void*) { REC_TODO("rec_dense_f32"); }
int rec_mha_ctr_f32(const float*, const float*, const float*, int64_t, int32_t, int32_t, const float*, const float*, const float*, const float*, int32_t, int32_t, int32_t, float*, void*) { REC_TODO("rec_mha_ctr_f32"); }
int rec_din_attn_pool_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*, int32_t, int64_t, int32_t, int32_t, float*, void*) { REC_TODO("rec_din_attn_pool_f32"); }
int rec_mha_rowmask_f32(const float*, const float*, const float*, const float*, int64_t, int32_t, int32_t, int32_t, float*, void*) { REC_TODO("rec_mha_rowmask_f32"); }
int rec_layernorm_residual_f32(const float*, const float*, const float*, const float*, float, const float*, int64_t, int32_t, float*, void*) { REC_TODO("rec_layernorm_residual_f32"); }
int rec_gather_dot_scores_f32(const float*, int64_t, const rec_table_desc*, const int32_t*, int64_t, int32_t, int64_t, float*, int64_t, int32_t*, void*) { REC_TODO("rec_gather_dot_scores_f32"); }
int64_t rec_shard_bucket_workspace_bytes(int64_t, int32_t) { return 0; }
int rec_shard_bucket_i32(const int32_t*, int64_t, int32_t, int32_t*, int32_t*, int32_t*, void*, void*) { REC_TODO("rec_shard_bucket_i32"); }
int rec_unpermute_rows_f32(const float*, const int32_t*, int64_t, int32_t, float*, int64_t, void*) { REC_TODO("rec_unpermute_rows_f32"); }
}
