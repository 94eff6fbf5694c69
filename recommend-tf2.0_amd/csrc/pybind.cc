// Thin pybind11 shim over the C ABI (include/recamd.h).  No torch types: device pointers and the
// HIP stream cross as Python ints (tensor.data_ptr(), torch.cuda.current_stream().cuda_stream).
// Every function releases the GIL around the (asynchronous) enqueue and raises RuntimeError with
// rec_last_error() on a negative status, so failures are loud.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "recamd.h"

namespace py = pybind11;
using ptr_t = uintptr_t;
using TableTuple = std::tuple<ptr_t, int64_t, int32_t, int32_t>;  // (base, vocab, dim, out_col)

static void check(int rc, const char* fn) {
  if (rc >= 0) return;
  char buf[512];
  rec_last_error(buf, sizeof(buf));
  throw std::runtime_error(std::string(fn) + " failed (" + std::to_string(rc) + "): " + buf);
}

static std::vector<rec_table_desc> to_descs(const std::vector<TableTuple>& t) {
  std::vector<rec_table_desc> d(t.size());
  for (size_t i = 0; i < t.size(); ++i) {
    d[i].base = reinterpret_cast<const float*>(std::get<0>(t[i]));
    d[i].vocab = std::get<1>(t[i]);
    d[i].dim = std::get<2>(t[i]);
    d[i].out_col = std::get<3>(t[i]);
  }
  return d;
}

template <typename T>
static T* P(ptr_t p) { return reinterpret_cast<T*>(p); }

PYBIND11_MODULE(_C, m) {
  m.doc() = "pybind11 shim over librecamd.so (C ABI, include/recamd.h)";
  m.def("version", []() { return rec_version(); });
  m.attr("MAX_TABLES") = REC_MAX_TABLES;

  m.def("gather_concat_f32",
        [](const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, int64_t ids_stride,
           int64_t B, ptr_t out, int64_t out_stride, ptr_t oob, ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_concat_f32(d.data(), (int32_t)d.size(), P<const void>(ids), ids_dtype,
                                      ids_stride, B, P<float>(out), out_stride, P<int32_t>(oob),
                                      P<void>(stream)),
                "rec_gather_concat_f32");
        });

  m.def("pairwise_dot_f32", [](ptr_t x, int64_t B, int n, int D, ptr_t out, int64_t out_stride,
                               ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_pairwise_dot_f32(P<const float>(x), B, n, D, P<float>(out), out_stride,
                               P<void>(stream)),
          "rec_pairwise_dot_f32");
  });

  m.def("gather_pairwise_dot_f32",
        [](const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, int64_t ids_stride,
           ptr_t dense, int64_t dense_stride, int64_t B, ptr_t out, int64_t out_stride,
           int append_dense, ptr_t oob, ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_pairwise_dot_f32(d.data(), (int32_t)d.size(), P<const void>(ids),
                                            ids_dtype, ids_stride, P<const float>(dense),
                                            dense_stride, B, P<float>(out), out_stride,
                                            append_dense, P<int32_t>(oob), P<void>(stream)),
                "rec_gather_pairwise_dot_f32");
        });

  m.def("fm_layer_workspace_floats", [](int64_t B) { return rec_fm_layer_workspace_floats(B); });
  m.def("fm_layer_f32", [](ptr_t first, int64_t first_stride, int L1, ptr_t w, ptr_t second,
                           int64_t second_stride, int M, int64_t B, ptr_t out, ptr_t ws,
                           ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_fm_layer_f32(P<const float>(first), first_stride, L1, P<const float>(w),
                           P<const float>(second), second_stride, M, B, P<float>(out),
                           P<float>(ws), P<void>(stream)),
          "rec_fm_layer_f32");
  });

  m.def("gather_fm_f32",
        [](const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, int64_t ids_stride, ptr_t dense,
           int64_t dense_stride, int nd, ptr_t w, int64_t B, ptr_t emb_out, int64_t emb_stride, ptr_t fm_out,
           ptr_t ws, ptr_t oob, ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_fm_f32(d.data(), (int32_t)d.size(), P<const void>(ids), ids_dtype, ids_stride,
                                  P<const float>(dense), dense_stride, nd, P<const float>(w), B,
                                  P<float>(emb_out), emb_stride, P<float>(fm_out), P<float>(ws),
                                  P<int32_t>(oob), P<void>(stream)),
                "rec_gather_fm_f32");
        });

  m.def("gather_fm_absmax_f32",
        [](const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, int64_t ids_stride, ptr_t dense,
           int64_t dense_stride, int nd, ptr_t w, int64_t B, ptr_t emb_out, int64_t emb_stride, ptr_t fm_out,
           ptr_t ws, ptr_t oob, ptr_t row_absmax, ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_fm_absmax_f32(d.data(), (int32_t)d.size(), P<const void>(ids), ids_dtype, ids_stride,
                                         P<const float>(dense), dense_stride, nd, P<const float>(w), B,
                                         P<float>(emb_out), emb_stride, P<float>(fm_out), P<float>(ws),
                                         P<int32_t>(oob), P<float>(row_absmax), P<void>(stream)),
                "rec_gather_fm_absmax_f32");
        });

  m.def("cross_f32", [](ptr_t x, int64_t x_stride, int dim, ptr_t w, ptr_t b, int L, int64_t B,
                        ptr_t out, int64_t out_stride, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_cross_f32(P<const float>(x), x_stride, dim, P<const float>(w), P<const float>(b), L,
                        B, P<float>(out), out_stride, P<void>(stream)),
          "rec_cross_f32");
  });

  m.def("fm_onehot_f32",
        [](ptr_t dense, int64_t dense_stride, int nd, ptr_t ids, int64_t ids_stride,
           const std::vector<int64_t>& vocab, ptr_t w0, ptr_t w, ptr_t V, int k, int64_t B,
           ptr_t out, ptr_t stream) {
          py::gil_scoped_release nogil;
          check(rec_fm_onehot_f32(P<const float>(dense), dense_stride, nd, P<const int32_t>(ids),
                                  ids_stride, (int32_t)vocab.size(), vocab.data(),
                                  P<const float>(w0), P<const float>(w), P<const float>(V), k, B,
                                  P<float>(out), P<void>(stream)),
                "rec_fm_onehot_f32");
        });

  m.def("dense_f32", [](ptr_t x, int64_t x_stride, ptr_t W, ptr_t bias, ptr_t alpha, int act,
                        int64_t M, int K, int N, ptr_t out, int64_t out_stride, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dense_f32(P<const float>(x), x_stride, P<const float>(W), P<const float>(bias),
                        P<const float>(alpha), act, M, K, N, P<float>(out), out_stride,
                        P<void>(stream)),
          "rec_dense_f32");
  });

  m.def("dense_prepared_bytes", [](int K, int N) { return rec_dense_prepared_bytes(K, N); });
  m.def("dense_prepare_f32", [](ptr_t W, int K, int N, ptr_t prepared, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dense_prepare_f32(P<const float>(W), K, N, P<void>(prepared), P<void>(stream)), "rec_dense_prepare_f32");
  });
  m.def("dense_prep_f32", [](ptr_t x, int64_t x_stride, ptr_t W, ptr_t prepared, ptr_t bias, ptr_t alpha, int act,
                             int64_t M, int K, int N, ptr_t out, int64_t out_stride, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dense_prep_f32(P<const float>(x), x_stride, P<const float>(W), P<const void>(prepared),
                             P<const float>(bias), P<const float>(alpha), act, M, K, N, P<float>(out), out_stride,
                             P<void>(stream)),
          "rec_dense_prep_f32");
  });
  m.def("dense_prep_rs_f32", [](ptr_t x, int64_t x_stride, ptr_t W, ptr_t prepared, ptr_t bias, ptr_t alpha, int act,
                                int64_t M, int K, int N, ptr_t out, int64_t out_stride, ptr_t row_absmax, int absmax_valid,
                                ptr_t out_absmax, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dense_prep_rs_f32(P<const float>(x), x_stride, P<const float>(W), P<const void>(prepared),
                                P<const float>(bias), P<const float>(alpha), act, M, K, N, P<float>(out), out_stride,
                                P<float>(row_absmax), absmax_valid, P<float>(out_absmax), P<void>(stream)),
          "rec_dense_prep_rs_f32");
  });
  m.def("mha_ctr_f32", [](ptr_t xq, ptr_t xk, ptr_t xv, int64_t B, int N, int din, ptr_t Wq,
                          ptr_t Wk, ptr_t Wv, ptr_t W0, int H, int S, int act, ptr_t out,
                          ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_mha_ctr_f32(P<const float>(xq), P<const float>(xk), P<const float>(xv), B, N, din,
                          P<const float>(Wq), P<const float>(Wk), P<const float>(Wv),
                          P<const float>(W0), H, S, act, P<float>(out), P<void>(stream)),
          "rec_mha_ctr_f32");
  });

  m.def("din_attn_pool_f32", [](ptr_t q, ptr_t k, ptr_t v, ptr_t mask, ptr_t W, ptr_t bias,
                                ptr_t alpha, int act, int64_t B, int T, int d, ptr_t out,
                                ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_din_attn_pool_f32(P<const float>(q), P<const float>(k), P<const float>(v),
                                P<const float>(mask), P<const float>(W), P<const float>(bias),
                                P<const float>(alpha), act, B, T, d, P<float>(out),
                                P<void>(stream)),
          "rec_din_attn_pool_f32");
  });

  m.def("gather_din_attn_pool_f32",
        [](ptr_t q, const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, ptr_t mask,
           int mask_from_ids, ptr_t W, ptr_t bias, ptr_t alpha, int act, int64_t B, int T, ptr_t out,
           ptr_t oob, ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_din_attn_pool_f32(P<const float>(q), d.data(), (int32_t)d.size(),
                                             P<const void>(ids), ids_dtype, P<const float>(mask),
                                             mask_from_ids, P<const float>(W), P<const float>(bias),
                                             P<const float>(alpha), act, B, T, P<float>(out),
                                             P<int32_t>(oob), P<void>(stream)),
                "rec_gather_din_attn_pool_f32");
        });

  m.def("gather_mha_fewq_f32", [](ptr_t q, int64_t qs, ptr_t table, int vocab, ptr_t ids, int ids_dtype, ptr_t mask,
                                  int64_t B, int Sq, int Sk, int dm, int H, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_gather_mha_fewq_f32(P<const float>(q), qs, P<const float>(table), vocab, P<const void>(ids), ids_dtype,
                                  P<const float>(mask), B, Sq, Sk, dm, H, P<float>(out), P<void>(stream)),
          "rec_gather_mha_fewq_f32");
  });
  m.def("mha_rowmask_strided_f32", [](ptr_t q, int64_t qs, ptr_t k, int64_t ks, ptr_t v, int64_t vs, ptr_t mask,
                                      int64_t B, int Sq, int Sk, int dm, int H, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_mha_rowmask_strided_f32(P<const float>(q), qs, P<const float>(k), ks, P<const float>(v), vs,
                                      P<const float>(mask), B, Sq, Sk, dm, H, P<float>(out), P<void>(stream)),
          "rec_mha_rowmask_strided_f32");
  });
  m.def("mha_rowmask_f32", [](ptr_t q, ptr_t k, ptr_t v, ptr_t mask, int64_t B, int Sq, int Sk,
                              int dm, int H, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_mha_rowmask_f32(P<const float>(q), P<const float>(k), P<const float>(v),
                              P<const float>(mask), B, Sq, Sk, dm, H, P<float>(out),
                              P<void>(stream)),
          "rec_mha_rowmask_f32");
  });

  m.def("layernorm_residual_f32", [](ptr_t x, ptr_t r, ptr_t gamma, ptr_t beta, float eps,
                                     ptr_t row_mask, int64_t rows, int d, ptr_t out,
                                     ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_layernorm_residual_f32(P<const float>(x), P<const float>(r), P<const float>(gamma),
                                     P<const float>(beta), eps, P<const float>(row_mask), rows, d,
                                     P<float>(out), P<void>(stream)),
          "rec_layernorm_residual_f32");
  });

  m.def("gather_dot_scores_f32",
        [](ptr_t seq_info, int64_t seq_stride, const TableTuple& table, ptr_t ids,
           int64_t ids_stride, int n, int64_t B, ptr_t out, int64_t out_stride, ptr_t oob,
           ptr_t stream) {
          auto d = to_descs({table});
          py::gil_scoped_release nogil;
          check(rec_gather_dot_scores_f32(P<const float>(seq_info), seq_stride, d.data(),
                                          P<const int32_t>(ids), ids_stride, n, B, P<float>(out),
                                          out_stride, P<int32_t>(oob), P<void>(stream)),
                "rec_gather_dot_scores_f32");
        });

  m.def("gather_dots_f32",
        [](const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, int64_t ids_stride, ptr_t Wd, int nv,
           int width, int64_t B, ptr_t emb_out, int64_t emb_stride, ptr_t out_dots, ptr_t oob, ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_dots_f32(d.data(), (int32_t)d.size(), P<const void>(ids), ids_dtype, ids_stride,
                                    P<const float>(Wd), nv, width, B, P<float>(emb_out), emb_stride,
                                    P<float>(out_dots), P<int32_t>(oob), P<void>(stream)),
                "rec_gather_dots_f32");
        });
  m.def("gather_dots_absmax_f32",
        [](const std::vector<TableTuple>& tables, ptr_t ids, int ids_dtype, int64_t ids_stride, ptr_t Wd, int nv,
           int width, int64_t B, ptr_t emb_out, int64_t emb_stride, ptr_t out_dots, ptr_t oob, ptr_t row_absmax,
           ptr_t stream) {
          auto d = to_descs(tables);
          py::gil_scoped_release nogil;
          check(rec_gather_dots_absmax_f32(d.data(), (int32_t)d.size(), P<const void>(ids), ids_dtype, ids_stride,
                                           P<const float>(Wd), nv, width, B, P<float>(emb_out), emb_stride,
                                           P<float>(out_dots), P<int32_t>(oob), P<float>(row_absmax), P<void>(stream)),
                "rec_gather_dots_absmax_f32");
        });
  m.def("dcn_logit_f32", [](ptr_t dots, int L, ptr_t G, float c, ptr_t extra, int64_t B, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dcn_logit_f32(P<const float>(dots), L, P<const float>(G), c, P<const float>(extra), B, P<float>(out),
                            P<void>(stream)),
          "rec_dcn_logit_f32");
  });
  m.def("add_sigmoid_f32", [](ptr_t a, ptr_t b, int64_t n, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_add_sigmoid_f32(P<const float>(a), P<const float>(b), n, P<float>(out), P<void>(stream)),
          "rec_add_sigmoid_f32");
  });
  m.def("axpby_act_f32", [](ptr_t a, float alpha, ptr_t b, float beta, int64_t n, int act, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_axpby_act_f32(P<const float>(a), alpha, P<const float>(b), beta, n, act, P<float>(out),
                            P<void>(stream)),
          "rec_axpby_act_f32");
  });
  m.def("mul_act_f32", [](ptr_t a, ptr_t b, int64_t n, int act, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_mul_act_f32(P<const float>(a), P<const float>(b), n, act, P<float>(out), P<void>(stream)),
          "rec_mul_act_f32");
  });
  m.def("cosine_flat_workspace_bytes", [](int64_t n) { return rec_cosine_flat_workspace_bytes(n); });
  m.def("cosine_flat_f32", [](ptr_t a, ptr_t b, int64_t n, int sig, ptr_t out, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_cosine_flat_f32(P<const float>(a), P<const float>(b), n, sig, P<float>(out), P<void>(ws),
                              P<void>(stream)),
          "rec_cosine_flat_f32");
  });
  m.def("scale_rows_f32", [](ptr_t x, ptr_t sc, int64_t rows, int d, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_scale_rows_f32(P<const float>(x), P<const float>(sc), rows, d, P<float>(out),
                             P<void>(stream)),
          "rec_scale_rows_f32");
  });
  m.def("dice_f32", [](ptr_t x, ptr_t alpha, ptr_t mean, ptr_t var, float eps, int64_t rows, int d,
                       ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dice_f32(P<const float>(x), P<const float>(alpha), P<const float>(mean),
                       P<const float>(var), eps, rows, d, P<float>(out), P<void>(stream)),
          "rec_dice_f32");
  });

  m.def("embedding_grad_f32",
        [](const std::vector<TableTuple>& grads, ptr_t ids, int ids_dtype, int64_t ids_stride, ptr_t dy,
           int64_t dy_stride, int64_t B, ptr_t stream) {
          auto d = to_descs(grads);
          py::gil_scoped_release nogil;
          check(rec_embedding_grad_f32(d.data(), (int32_t)d.size(), P<const void>(ids), ids_dtype, ids_stride,
                                       P<const float>(dy), dy_stride, B, P<void>(stream)),
                "rec_embedding_grad_f32");
        });
  m.def("adam_f32", [](ptr_t var, ptr_t mm, ptr_t vv, ptr_t grad, int64_t n, float lr, float b1, float b2,
                       float eps, int64_t step, float l2, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_adam_f32(P<float>(var), P<float>(mm), P<float>(vv), P<const float>(grad), n, lr, b1, b2, eps, step,
                       l2, P<void>(stream)),
          "rec_adam_f32");
  });

  m.def("metrics_workspace_bytes", [](int64_t n) { return rec_metrics_workspace_bytes(n); });
  m.def("binary_crossentropy_f32", [](ptr_t y, ptr_t p, int64_t n, ptr_t out, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_binary_crossentropy_f32(P<const float>(y), P<const float>(p), n, P<float>(out), P<void>(ws),
                                      P<void>(stream)),
          "rec_binary_crossentropy_f32");
  });
  m.def("auc_f32", [](ptr_t y, ptr_t p, int64_t n, ptr_t out, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_auc_f32(P<const float>(y), P<const float>(p), n, P<float>(out), P<void>(ws), P<void>(stream)),
          "rec_auc_f32");
  });
  m.def("topk_ip_workspace_bytes", [](int64_t Q, int64_t N, int k) { return rec_topk_ip_workspace_bytes(Q, N, k); });
  m.def("topk_ip_ws_f32", [](ptr_t q, int64_t q_stride, int64_t Q, ptr_t items, int64_t items_stride, int64_t N, int d,
                             int k, ptr_t out_scores, ptr_t out_idx, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_topk_ip_ws_f32(P<const float>(q), q_stride, Q, P<const float>(items), items_stride, N, d, k,
                             P<float>(out_scores), P<int64_t>(out_idx), P<void>(ws), P<void>(stream)),
          "rec_topk_ip_ws_f32");
  });
  m.def("topk_ip_f32", [](ptr_t q, int64_t q_stride, int64_t Q, ptr_t items, int64_t items_stride, int64_t N, int d,
                          int k, ptr_t out_scores, ptr_t out_idx, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_topk_ip_f32(P<const float>(q), q_stride, Q, P<const float>(items), items_stride, N, d, k,
                          P<float>(out_scores), P<int64_t>(out_idx), P<void>(stream)),
          "rec_topk_ip_f32");
  });

  m.def("shard_bucket_workspace_bytes",
        [](int64_t n, int G) { return rec_shard_bucket_workspace_bytes(n, G); });
  m.def("shard_bucket_i32", [](ptr_t ids, int64_t n, int G, ptr_t counts, ptr_t perm,
                               ptr_t send_local, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_bucket_i32(P<const int32_t>(ids), n, G, P<int32_t>(counts), P<int32_t>(perm),
                               P<int32_t>(send_local), P<void>(ws), P<void>(stream)),
          "rec_shard_bucket_i32");
  });
  m.def("shard_dedup_bucket_i32", [](ptr_t vids, int64_t n, int G, ptr_t rep, ptr_t first, ptr_t uniq, ptr_t perm,
                                     ptr_t uidx, ptr_t send_local, ptr_t counts, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_dedup_bucket_i32(P<const int32_t>(vids), n, G, P<int32_t>(rep), P<int32_t>(first), P<int32_t>(uniq),
                                     P<int32_t>(perm), P<int32_t>(uidx), P<int32_t>(send_local), P<int32_t>(counts),
                                     P<void>(ws), P<void>(stream)),
          "rec_shard_dedup_bucket_i32");
  });
  m.def("unpermute_rows_f32", [](ptr_t rows, ptr_t perm, int64_t n, int D, ptr_t out,
                                 int64_t out_stride, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_unpermute_rows_f32(P<const float>(rows), P<const int32_t>(perm), n, D,
                                 P<float>(out), out_stride, P<void>(stream)),
          "rec_unpermute_rows_f32");
  });
  // ---- communicator + sharded-lookup plan (opaque handles cross as integers) ----------------------------
  m.def("debug_force", [](const std::string& key, py::object value) {   // tests / A-B only (rec_debug_force)
    if (value.is_none()) check(rec_debug_force(key.c_str(), nullptr), "rec_debug_force");
    else check(rec_debug_force(key.c_str(), value.cast<std::string>().c_str()), "rec_debug_force");
  });
  m.def("comm_unique_id", []() {
    char id[128];
    check(rec_comm_unique_id(id), "rec_comm_unique_id");
    return py::bytes(id, 128);
  });
  m.def("comm_init_rank", [](const std::string& id, int world, int rank) {
    if (id.size() != 128) throw std::runtime_error("comm_init_rank: the unique id must be 128 bytes");
    rec_comm* c = nullptr;
    {
      py::gil_scoped_release nogil;
      check(rec_comm_init_rank(&c, id.data(), world, rank), "rec_comm_init_rank");
    }
    return reinterpret_cast<ptr_t>(c);
  });
  m.def("comm_from_nccl", [](ptr_t nccl_comm, int world, int rank) {   // borrow an existing ncclComm_t
    rec_comm* c = nullptr;
    check(rec_comm_from_nccl(&c, P<void>(nccl_comm), world, rank), "rec_comm_from_nccl");
    return reinterpret_cast<ptr_t>(c);
  });
  // a transport struct built by another native module (address of a rec_transport); Python itself cannot supply callbacks
  m.def("comm_create_with_transport", [](ptr_t transport, int world, int rank) {
    rec_comm* c = nullptr;
    check(rec_comm_create_with_transport(&c, P<const rec_transport>(transport), world, rank), "rec_comm_create_with_transport");
    return reinterpret_cast<ptr_t>(c);
  });
  m.def("comm_create_local", [](int world) {
    std::vector<rec_comm*> cs(world > 0 ? world : 0, nullptr);
    check(rec_comm_create_local(world, cs.data()), "rec_comm_create_local");
    std::vector<ptr_t> out;
    for (rec_comm* c : cs) out.push_back(reinterpret_cast<ptr_t>(c));
    return out;
  });
  m.def("comm_destroy", [](ptr_t c) { check(rec_comm_destroy(P<rec_comm>(c)), "rec_comm_destroy"); });
  m.def("comm_world", [](ptr_t c) { return rec_comm_world(P<rec_comm>(c)); });
  m.def("comm_rank", [](ptr_t c) { return rec_comm_rank(P<rec_comm>(c)); });
  m.def("comm_transport_name", [](ptr_t c) { return std::string(rec_comm_transport_name(P<rec_comm>(c))); });
  m.def("comm_allreduce_sum_f32", [](ptr_t c, ptr_t buf, int64_t n, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_comm_allreduce_sum_f32(P<rec_comm>(c), P<float>(buf), n, P<void>(stream)), "rec_comm_allreduce_sum_f32");
  });
  m.def("shard_plan_workspace_bytes", [](int64_t max_ids, int world) { return rec_shard_plan_workspace_bytes(max_ids, world); });
  m.def("shard_plan_create", [](ptr_t comm, int64_t max_ids) {
    rec_shard_plan* p = nullptr;
    check(rec_shard_plan_create(P<rec_comm>(comm), max_ids, &p), "rec_shard_plan_create");
    return reinterpret_cast<ptr_t>(p);
  });
  m.def("shard_plan_destroy", [](ptr_t p) { check(rec_shard_plan_destroy(P<rec_shard_plan>(p)), "rec_shard_plan_destroy"); });
  m.def("shard_plan_ids", [](ptr_t p, ptr_t vids, int64_t n, ptr_t rep, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_plan_ids(P<rec_shard_plan>(p), P<const int32_t>(vids), n, P<int32_t>(rep), P<void>(ws), P<void>(stream)),
          "rec_shard_plan_ids");
  });
  m.def("shard_plan_ids_ex", [](ptr_t p, ptr_t vids, int64_t n, ptr_t rep, int bypass_local, ptr_t cache_slot,
                                ptr_t hot_count, int cache_base, int recv_base, ptr_t stat, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    rec_shard_resolve_opts o{bypass_local, P<const int32_t>(cache_slot), P<int32_t>(hot_count), cache_base, recv_base,
                             P<uint64_t>(stat)};
    check(rec_shard_plan_ids_ex(P<rec_shard_plan>(p), P<const int32_t>(vids), n, P<int32_t>(rep), &o, P<void>(ws),
                                P<void>(stream)),
          "rec_shard_plan_ids_ex");
  });
  m.def("shard_resolve_i32", [](ptr_t vids, int64_t n, int G, int me, ptr_t rep, ptr_t cache_slot, ptr_t hot_count,
                                int cache_base, int recv_base, ptr_t stat, ptr_t first, ptr_t uniq, ptr_t perm, ptr_t uidx,
                                ptr_t send_local, ptr_t counts, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_resolve_i32(P<const int32_t>(vids), n, G, me, P<int32_t>(rep), P<const int32_t>(cache_slot),
                                P<int32_t>(hot_count), cache_base, recv_base, P<uint64_t>(stat), P<int32_t>(first),
                                P<int32_t>(uniq), P<int32_t>(perm), P<int32_t>(uidx), P<int32_t>(send_local),
                                P<int32_t>(counts), P<void>(ws), P<void>(stream)),
          "rec_shard_resolve_i32");
  });
  m.def("shard_plan_finish", [](ptr_t p) {
    int64_t nu = 0, nr = 0;
    {
      py::gil_scoped_release nogil;
      check(rec_shard_plan_finish(P<rec_shard_plan>(p), &nu, &nr), "rec_shard_plan_finish");
    }
    return std::make_tuple(nu, nr);
  });
  m.def("shard_plan_uidx", [](ptr_t p) { return reinterpret_cast<ptr_t>(rec_shard_plan_uidx(P<rec_shard_plan>(p))); });
  m.def("shard_exchange_ids", [](ptr_t p, ptr_t recv_local, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_exchange_ids(P<rec_shard_plan>(p), P<int32_t>(recv_local), P<void>(stream)), "rec_shard_exchange_ids");
  });
  m.def("shard_serve_f32", [](ptr_t p, ptr_t arena, int64_t arena_rows, int D, ptr_t recv_local, ptr_t served, ptr_t oob,
                              ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_serve_f32(P<rec_shard_plan>(p), P<const float>(arena), arena_rows, D, P<const int32_t>(recv_local),
                              P<float>(served), P<int32_t>(oob), P<void>(stream)),
          "rec_shard_serve_f32");
  });
  m.def("shard_exchange_rows_f32", [](ptr_t p, ptr_t src, int D, ptr_t dst, int reverse, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_exchange_rows_f32(P<rec_shard_plan>(p), P<const float>(src), D, P<float>(dst), reverse, P<void>(stream)),
          "rec_shard_exchange_rows_f32");
  });
  m.def("shard_lookup_f32", [](ptr_t p, ptr_t arena, int64_t arena_rows, int D, ptr_t recv_local, int64_t recv_cap,
                               ptr_t served, ptr_t rows_out, int64_t rows_cap, ptr_t oob, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_shard_lookup_f32(P<rec_shard_plan>(p), P<const float>(arena), arena_rows, D, P<int32_t>(recv_local), recv_cap,
                               P<float>(served), P<float>(rows_out), rows_cap, P<int32_t>(oob), P<void>(stream)),
          "rec_shard_lookup_f32");
  });
  // ---- backward kernels ------------------------------------------------------------------------------------
  m.def("transpose_f32", [](ptr_t x, int64_t M, int64_t N, int64_t xs, ptr_t out, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_transpose_f32(P<const float>(x), M, N, xs, P<float>(out), P<void>(stream)), "rec_transpose_f32");
  });
  m.def("act_grad_f32", [](ptr_t dy, int64_t dys, ptr_t y, int64_t ys, int64_t M, int64_t N, int act, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_act_grad_f32(P<float>(dy), dys, P<const float>(y), ys, M, N, act, P<void>(stream)), "rec_act_grad_f32");
  });
  m.def("colsum_workspace_bytes", [](int64_t M, int64_t N) { return rec_colsum_workspace_bytes(M, N); });
  m.def("colsum_f32", [](ptr_t a, int64_t as, ptr_t b, int64_t bs, ptr_t rw, int64_t M, int64_t N, ptr_t out, ptr_t ws,
                         ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_colsum_f32(P<const float>(a), as, P<const float>(b), bs, P<const float>(rw), M, N, P<float>(out), P<void>(ws),
                         P<void>(stream)),
          "rec_colsum_f32");
  });
  m.def("bn_train_f32", [](ptr_t x, int64_t xs, int64_t M, int64_t N, ptr_t gamma, ptr_t beta, float eps, float momentum,
                           ptr_t mm, ptr_t mv, ptr_t y, int64_t ys, ptr_t save_mean, ptr_t save_inv, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_bn_train_f32(P<const float>(x), xs, M, N, P<const float>(gamma), P<const float>(beta), eps, momentum,
                           P<float>(mm), P<float>(mv), P<float>(y), ys, P<float>(save_mean), P<float>(save_inv), P<void>(ws),
                           P<void>(stream)),
          "rec_bn_train_f32");
  });
  m.def("bn_train_grad_workspace_bytes", [](int64_t M, int64_t N) { return rec_bn_train_grad_workspace_bytes(M, N); });
  m.def("bn_train_grad_f32", [](ptr_t x, int64_t xs, ptr_t dy, int64_t dys, int64_t M, int64_t N, ptr_t gamma, ptr_t save_mean,
                                ptr_t save_inv, ptr_t dx, int64_t dxs, ptr_t dgamma, ptr_t dbeta, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_bn_train_grad_f32(P<const float>(x), xs, P<const float>(dy), dys, M, N, P<const float>(gamma),
                                P<const float>(save_mean), P<const float>(save_inv), P<float>(dx), dxs, P<float>(dgamma),
                                P<float>(dbeta), P<void>(ws), P<void>(stream)),
          "rec_bn_train_grad_f32");
  });
  m.def("bce_sigmoid_grad_f32", [](ptr_t y, ptr_t p, int64_t n, float scale, ptr_t dz, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_bce_sigmoid_grad_f32(P<const float>(y), P<const float>(p), n, scale, P<float>(dz), P<void>(stream)),
          "rec_bce_sigmoid_grad_f32");
  });
  m.def("gather_pairwise_dot_grad_f32", [](const std::vector<TableTuple>& tables, const std::vector<TableTuple>& grads,
                                           ptr_t ids, int64_t ids_stride, ptr_t dense, int64_t dense_stride, int64_t B,
                                           ptr_t dz, int64_t dz_stride, int append, ptr_t d_dense, int64_t dd_stride,
                                           ptr_t stream) {
    auto t = to_descs(tables), g = to_descs(grads);
    py::gil_scoped_release nogil;
    check(rec_gather_pairwise_dot_grad_f32(t.data(), g.data(), (int32_t)t.size(), P<const int32_t>(ids), ids_stride,
                                           P<const float>(dense), dense_stride, B, P<const float>(dz), dz_stride, append,
                                           P<float>(d_dense), dd_stride, P<void>(stream)),
          "rec_gather_pairwise_dot_grad_f32");
  });
  m.def("fm_layer_grad_workspace_bytes", [](int64_t B, int64_t L1) { return rec_fm_layer_grad_workspace_bytes(B, L1); });
  m.def("fm_layer_grad_f32", [](ptr_t first, int64_t fs, int64_t L1, ptr_t second, int64_t ss, int64_t M, ptr_t w, ptr_t dout,
                                int64_t B, ptr_t d_first, int64_t dfs, ptr_t d_second, int64_t dss, ptr_t dw, ptr_t ws,
                                ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_fm_layer_grad_f32(P<const float>(first), fs, L1, P<const float>(second), ss, M, P<const float>(w),
                                P<const float>(dout), B, P<float>(d_first), dfs, P<float>(d_second), dss, P<float>(dw),
                                P<void>(ws), P<void>(stream)),
          "rec_fm_layer_grad_f32");
  });
  m.def("cross_layer_grad_f32", [](ptr_t x0, ptr_t xl, ptr_t w, int64_t dim, int64_t B, ptr_t g, ptr_t dx0, ptr_t ds,
                                   ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_cross_layer_grad_f32(P<const float>(x0), P<const float>(xl), P<const float>(w), dim, B, P<float>(g),
                                   P<float>(dx0), P<float>(ds), P<void>(stream)),
          "rec_cross_layer_grad_f32");
  });
  m.def("adam_rows_f32", [](const std::vector<TableTuple>& var, const std::vector<TableTuple>& mm,
                            const std::vector<TableTuple>& vv, const std::vector<TableTuple>& grad,
                            const std::vector<TableTuple>& stamp, ptr_t ids, int64_t ids_stride, int64_t B, float lr, float b1,
                            float b2, float eps, int64_t step, float l2, ptr_t stream) {
    auto a = to_descs(var), b = to_descs(mm), c = to_descs(vv), d = to_descs(grad), e = to_descs(stamp);
    py::gil_scoped_release nogil;
    check(rec_adam_rows_f32(a.data(), b.data(), c.data(), d.data(), e.data(), (int32_t)a.size(), P<const int32_t>(ids),
                            ids_stride, B, lr, b1, b2, eps, step, l2, P<void>(stream)),
          "rec_adam_rows_f32");
  });
  // ---- input pipeline --------------------------------------------------------------------------------------
  m.def("label_encode_u32", [](const std::vector<ptr_t>& vocabs, const std::vector<int32_t>& sizes, ptr_t tokens,
                               int64_t tok_stride, int64_t B, ptr_t ids, int64_t ids_stride, ptr_t unseen, ptr_t stream) {
    std::vector<const uint32_t*> vp;
    for (ptr_t v : vocabs) vp.push_back(P<const uint32_t>(v));
    py::gil_scoped_release nogil;
    check(rec_label_encode_u32(vp.data(), sizes.data(), (int32_t)vp.size(), P<const uint32_t>(tokens), tok_stride, B,
                               P<int32_t>(ids), ids_stride, P<int32_t>(unseen), P<void>(stream)),
          "rec_label_encode_u32");
  });
  m.def("hash_ids_u32", [](ptr_t tokens, int64_t tok_stride, const std::vector<int32_t>& sizes, int64_t B, uint32_t seed,
                           ptr_t ids, int64_t ids_stride, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_hash_ids_u32(P<const uint32_t>(tokens), tok_stride, sizes.data(), (int32_t)sizes.size(), B, seed,
                           P<int32_t>(ids), ids_stride, P<void>(stream)),
          "rec_hash_ids_u32");
  });
  m.def("minmax_workspace_bytes", [](int64_t M, int N) { return rec_minmax_workspace_bytes(M, N); });
  m.def("minmax_fit_f32", [](ptr_t x, int64_t xs, int64_t M, int N, int trunc, ptr_t mn, ptr_t mx, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_minmax_fit_f32(P<const float>(x), xs, M, N, trunc, P<float>(mn), P<float>(mx), P<void>(ws), P<void>(stream)),
          "rec_minmax_fit_f32");
  });
  m.def("minmax_scale_f32", [](ptr_t x, int64_t xs, int64_t M, int N, ptr_t mn, ptr_t mx, int trunc, ptr_t out, int64_t os,
                               ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_minmax_scale_f32(P<const float>(x), xs, M, N, P<const float>(mn), P<const float>(mx), trunc, P<float>(out), os,
                               P<void>(stream)),
          "rec_minmax_scale_f32");
  });
  m.def("pad_sequences_i32", [](ptr_t values, ptr_t offsets, int64_t B, int maxlen, int pad, int pre_pad, int pre_trunc,
                                ptr_t out, int64_t os, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_pad_sequences_i32(P<const int32_t>(values), P<const int64_t>(offsets), B, maxlen, pad, pre_pad, pre_trunc,
                                P<int32_t>(out), os, P<void>(stream)),
          "rec_pad_sequences_i32");
  });
  m.def("copy2d_f32", [](ptr_t src, int64_t ss, int is_f32, int64_t M, int64_t N, ptr_t dst, int64_t ds, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_copy2d_f32(P<const void>(src), ss, is_f32, M, N, P<float>(dst), ds, P<void>(stream)), "rec_copy2d_f32");
  });
  m.def("scale_embed_f32", [](ptr_t x, int64_t xs, ptr_t E, int64_t B, int nd, int D, ptr_t out, int64_t os, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_scale_embed_f32(P<const float>(x), xs, P<const float>(E), B, nd, D, P<float>(out), os, P<void>(stream)),
          "rec_scale_embed_f32");
  });
  m.def("mha_ctr_stack_f32", [](ptr_t x, int64_t B, int N, int din, const std::vector<ptr_t>& Wq, const std::vector<ptr_t>& Wk,
                                const std::vector<ptr_t>& Wv, const std::vector<ptr_t>& W0, int H, int S, int act, ptr_t out,
                                ptr_t stream) {
    const size_t L = Wq.size();
    if (Wk.size() != L || Wv.size() != L || (!W0.empty() && W0.size() != L))
      throw std::runtime_error("mha_ctr_stack_f32: weight lists must have one entry per layer");
    std::vector<const float*> q, k, v, r;
    for (size_t l = 0; l < L; ++l) {
      q.push_back(P<const float>(Wq[l])), k.push_back(P<const float>(Wk[l])), v.push_back(P<const float>(Wv[l]));
      r.push_back(W0.empty() ? nullptr : P<const float>(W0[l]));
    }
    py::gil_scoped_release nogil;
    check(rec_mha_ctr_stack_f32(P<const float>(x), B, N, din, q.data(), k.data(), v.data(), r.data(), (int32_t)L, H, S, act,
                                P<float>(out), P<void>(stream)),
          "rec_mha_ctr_stack_f32");
  });
  m.def("sasrec_last_row_supported", [](int d, int ffn_hidden, int S, int n_cand) {
    return rec_sasrec_last_row_supported(d, ffn_hidden, S, n_cand) != 0;
  });
  m.def("sasrec_last_row_f32", [](const std::vector<ptr_t>& w, float eps1, float eps2, int ffn_hidden, ptr_t seq_table,
                                  int seq_vocab, ptr_t seq_ids, int64_t seq_stride, int S, int pad_id, ptr_t mask_ids,
                                  int64_t mask_stride, ptr_t pos_table, int pos_vocab, ptr_t pos_ids, int64_t pos_stride,
                                  int n_pos, ptr_t neg_table, int neg_vocab, ptr_t neg_ids, int64_t neg_stride, int n_neg,
                                  int64_t B, int d, ptr_t seq_info, ptr_t logits, int64_t logits_stride, ptr_t oob,
                                  ptr_t stream) {
    if (w.size() != 13) throw std::runtime_error("sasrec_last_row_f32: 13 weight pointers expected");
    rec_sasrec_block blk{P<const float>(w[0]), P<const float>(w[1]), P<const float>(w[2]), P<const float>(w[3]),
                         P<const float>(w[4]), P<const float>(w[5]), P<const float>(w[6]), P<const float>(w[7]),
                         P<const float>(w[8]), P<const float>(w[9]), P<const float>(w[10]), P<const float>(w[11]),
                         P<const float>(w[12]), eps1, eps2, ffn_hidden};
    py::gil_scoped_release nogil;
    check(rec_sasrec_last_row_f32(&blk, P<const float>(seq_table), seq_vocab, P<const int32_t>(seq_ids), seq_stride, S,
                                  pad_id, P<const int32_t>(mask_ids), mask_stride, P<const float>(pos_table), pos_vocab,
                                  P<const int32_t>(pos_ids), pos_stride, n_pos, P<const float>(neg_table), neg_vocab,
                                  P<const int32_t>(neg_ids), neg_stride, n_neg, B, d, P<float>(seq_info), P<float>(logits),
                                  logits_stride, P<int32_t>(oob), P<void>(stream)),
          "rec_sasrec_last_row_f32");
  });
  m.def("pairwise_rank_loss_f32", [](ptr_t logits, int64_t stride, int64_t B, int n_neg, ptr_t out, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_pairwise_rank_loss_f32(P<const float>(logits), stride, B, n_neg, P<float>(out), P<void>(ws), P<void>(stream)),
          "rec_pairwise_rank_loss_f32");
  });
  m.def("autoint_forward_f32", [](const std::vector<TableTuple>& tables, ptr_t ids, int64_t ids_stride, ptr_t dense,
                                  int64_t dense_stride, int n_dense, ptr_t dense_embed, int D, const std::vector<ptr_t>& Wq,
                                  const std::vector<ptr_t>& Wk, const std::vector<ptr_t>& Wv, const std::vector<ptr_t>& W0,
                                  int H, int S, int act, ptr_t head_w, ptr_t head_b, int64_t B, ptr_t out_prob,
                                  ptr_t out_fields, ptr_t oob, ptr_t stream) {
    const size_t L = Wq.size();
    if (Wk.size() != L || Wv.size() != L || (!W0.empty() && W0.size() != L))
      throw std::runtime_error("autoint_forward_f32: weight lists must have one entry per layer");
    auto d = to_descs(tables);
    std::vector<const float*> q, k, v, r;
    for (size_t l = 0; l < L; ++l) {
      q.push_back(P<const float>(Wq[l])), k.push_back(P<const float>(Wk[l])), v.push_back(P<const float>(Wv[l]));
      r.push_back(W0.empty() ? nullptr : P<const float>(W0[l]));
    }
    py::gil_scoped_release nogil;
    check(rec_autoint_forward_f32(d.data(), (int32_t)d.size(), P<const int32_t>(ids), ids_stride, P<const float>(dense),
                                  dense_stride, n_dense, P<const float>(dense_embed), D, q.data(), k.data(), v.data(),
                                  r.data(), (int32_t)L, H, S, act, P<const float>(head_w), P<const float>(head_b), B,
                                  P<float>(out_prob), P<float>(out_fields), P<int32_t>(oob), P<void>(stream)),
          "rec_autoint_forward_f32");
  });

  // ---- T3: attention-shaped backward kernels (csrc/train_attn.hip) ------------------------------------------
  m.def("attn_core_f32", [](ptr_t q, int64_t ldq, ptr_t k, int64_t ldk, ptr_t v, int64_t ldv, ptr_t mask, int64_t B, int Nq,
                            int Nk, int H, int S, float scale, ptr_t out, int64_t ldo, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_attn_core_f32(P<const float>(q), ldq, P<const float>(k), ldk, P<const float>(v), ldv, P<const float>(mask), B,
                            Nq, Nk, H, S, scale, P<float>(out), ldo, P<void>(stream)),
          "rec_attn_core_f32");
  });
  m.def("attn_core_grad_workspace_bytes", [](int64_t B, int Nq, int Nk, int H) {
    return rec_attn_core_grad_workspace_bytes(B, Nq, Nk, H);
  });
  m.def("attn_core_grad_f32", [](ptr_t q, int64_t ldq, ptr_t k, int64_t ldk, ptr_t v, int64_t ldv, ptr_t mask, ptr_t dout,
                                 int64_t lddo, int64_t B, int Nq, int Nk, int H, int S, float scale, ptr_t dq, int64_t lddq,
                                 ptr_t dk, int64_t lddk, ptr_t dv, int64_t lddv, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_attn_core_grad_f32(P<const float>(q), ldq, P<const float>(k), ldk, P<const float>(v), ldv,
                                 P<const float>(mask), P<const float>(dout), lddo, B, Nq, Nk, H, S, scale, P<float>(dq), lddq,
                                 P<float>(dk), lddk, P<float>(dv), lddv, P<void>(ws), P<void>(stream)),
          "rec_attn_core_grad_f32");
  });
  m.def("din_attn_pool_grad_f32", [](ptr_t q, ptr_t k, ptr_t v, ptr_t mask, int mask_is_none, ptr_t W, ptr_t bias, int act,
                                     ptr_t alpha, ptr_t dout, int64_t B, int T, int d, ptr_t dq, ptr_t dk, ptr_t dv,
                                     ptr_t partials, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_din_attn_pool_grad_f32(P<const float>(q), P<const float>(k), P<const float>(v), P<const float>(mask),
                                     mask_is_none, P<const float>(W), P<const float>(bias), act, P<const float>(alpha),
                                     P<const float>(dout), B, T, d, P<float>(dq), P<float>(dk), P<float>(dv),
                                     P<float>(partials), P<void>(stream)),
          "rec_din_attn_pool_grad_f32");
  });
  m.def("prelu_f32", [](ptr_t z, int64_t zs, ptr_t alpha, int64_t M, int64_t N, ptr_t y, int64_t ys, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_prelu_f32(P<const float>(z), zs, P<const float>(alpha), M, N, P<float>(y), ys, P<void>(stream)), "rec_prelu_f32");
  });
  m.def("prelu_grad_f32", [](ptr_t z, int64_t zs, ptr_t alpha, ptr_t dy, int64_t dys, int64_t M, int64_t N, ptr_t dz,
                             ptr_t neg_part, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_prelu_grad_f32(P<const float>(z), zs, P<const float>(alpha), P<const float>(dy), dys, M, N, P<float>(dz),
                             P<float>(neg_part), P<void>(stream)),
          "rec_prelu_grad_f32");
  });
  m.def("dice_train_f32", [](ptr_t x, ptr_t xn, ptr_t alpha, int64_t n, ptr_t y, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dice_train_f32(P<const float>(x), P<const float>(xn), P<const float>(alpha), n, P<float>(y), P<void>(stream)),
          "rec_dice_train_f32");
  });
  m.def("dice_train_grad_f32", [](ptr_t x, ptr_t xn, ptr_t alpha, ptr_t dy, int64_t n, ptr_t dx, ptr_t dxn, ptr_t da, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dice_train_grad_f32(P<const float>(x), P<const float>(xn), P<const float>(alpha), P<const float>(dy), n, P<float>(dx),
                            P<float>(dxn), P<float>(da), P<void>(stream)),
          "rec_dice_train_grad_f32");
  });
  m.def("layernorm_residual_grad_f32", [](ptr_t x, ptr_t r, ptr_t gamma, ptr_t mask, ptr_t dy, int64_t M, int d, float eps,
                                          ptr_t ds, ptr_t xhat, ptr_t dym, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_layernorm_residual_grad_f32(P<const float>(x), P<const float>(r), P<const float>(gamma), P<const float>(mask),
                                          P<const float>(dy), M, d, eps, P<float>(ds), P<float>(xhat), P<float>(dym),
                                          P<void>(stream)),
          "rec_layernorm_residual_grad_f32");
  });
  m.def("pairwise_rank_loss_grad_f32", [](ptr_t logits, int64_t ls, int64_t B, int n_neg, float scale, ptr_t dl, int64_t ds,
                                          ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_pairwise_rank_loss_grad_f32(P<const float>(logits), ls, B, n_neg, scale, P<float>(dl), ds, P<void>(stream)),
          "rec_pairwise_rank_loss_grad_f32");
  });
  m.def("gather_dot_scores_grad_f32", [](ptr_t seq, ptr_t table, ptr_t gtable, int64_t vocab, int d, ptr_t ids,
                                         int64_t ids_stride, int n, ptr_t dl, int64_t dls, int64_t B, ptr_t dseq,
                                         int accumulate, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_gather_dot_scores_grad_f32(P<const float>(seq), P<const float>(table), P<float>(gtable), vocab, d,
                                         P<const int32_t>(ids), ids_stride, n, P<const float>(dl), dls, B, P<float>(dseq),
                                         accumulate, P<void>(stream)),
          "rec_gather_dot_scores_grad_f32");
  });
  m.def("fm_onehot_grad_f32", [](ptr_t dense, int64_t dense_stride, int n_dense, ptr_t ids, int64_t ids_stride,
                                 const std::vector<int32_t>& vocab, ptr_t V, int k, ptr_t dlogit, int64_t B, ptr_t dw, ptr_t dV,
                                 ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_fm_onehot_grad_f32(P<const float>(dense), dense_stride, n_dense, P<const int32_t>(ids), ids_stride,
                                 (int32_t)vocab.size(), vocab.data(), P<const float>(V), k, P<const float>(dlogit), B,
                                 P<float>(dw), P<float>(dV), P<void>(stream)),
          "rec_fm_onehot_grad_f32");
  });
  m.def("wgrad_small_workspace_bytes", [](int64_t M, int K, int N) { return rec_wgrad_small_workspace_bytes(M, K, N); });
  m.def("wgrad_small_f32", [](ptr_t x, int64_t xs, ptr_t dy, int64_t dys, int64_t M, int K, int N, ptr_t out, ptr_t ws,
                              ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_wgrad_small_f32(P<const float>(x), xs, P<const float>(dy), dys, M, K, N, P<float>(out), P<void>(ws),
                              P<void>(stream)),
          "rec_wgrad_small_f32");
  });
  m.def("bce_prob_grad_f32", [](ptr_t y, ptr_t p, int64_t n, float scale, ptr_t dp, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_bce_prob_grad_f32(P<const float>(y), P<const float>(p), n, scale, P<float>(dp), P<void>(stream)),
          "rec_bce_prob_grad_f32");
  });
  m.def("dense_splitk_workspace_bytes", [](int64_t M, int K, int N) { return rec_dense_splitk_workspace_bytes(M, K, N); });
  m.def("dense_splitk_f32", [](ptr_t x, int64_t xs, ptr_t W, int64_t M, int K, int N, ptr_t out, ptr_t ws, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dense_splitk_f32(P<const float>(x), xs, P<const float>(W), M, K, N, P<float>(out), P<void>(ws), P<void>(stream)),
          "rec_dense_splitk_f32");
  });
  m.def("dropout_f32", [](ptr_t x, int64_t n, float rate, uint64_t seed, ptr_t y, ptr_t stream) {
    py::gil_scoped_release nogil;
    check(rec_dropout_f32(P<const float>(x), n, rate, seed, P<float>(y), P<void>(stream)), "rec_dropout_f32");
  });
}
