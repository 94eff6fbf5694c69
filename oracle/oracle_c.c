/*
 * oracle_c.c — plain-C CPU restatement of the hot path (gather+concat, DLRM pairwise dot, FM
 * layer, CrossNetwork).  TEST INFRASTRUCTURE ONLY: used by tests/ as a second, independent checker
 * next to oracle/ref_numpy.py and by bench.py's `cpu_baseline` leg (kind "port").  The product
 * (recommend-tf2.0_amd/) never links or loads it.
 *
 * PARITY UNPINNED: the reference cannot be executed here (TensorFlow absent) and ships no golden
 * vectors; see oracle/ref_numpy.py.  Citations: file:line relative to the reference repo root.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared)  ->  oracle/_build/liboracle_c.so
 */
#include <stdint.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* tf.concat([Embedding_f(ids[:, f])], axis=-1): src/ctr/deep_fm/model.py:53, dcn/model.py:47,
 * dlrm/model.py:45.  tables[f] -> (vocab[f], dim[f]) row-major; out (B, sum dim) row-major.
 * Out-of-range id -> zero row (TF-GPU semantics); returns the number of such ids. */
int64_t orc_gather_concat_f32(const float* const* tables, const int64_t* vocab, const int32_t* dim,
                              int32_t F, const int32_t* ids, int64_t B, float* out) {
  int64_t width = 0, bad = 0;
  for (int f = 0; f < F; ++f) width += dim[f];
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (int64_t b = 0; b < B; ++b) {
    float* o = out + b * width;
    for (int f = 0; f < F; ++f) {
      const int64_t id = ids[b * F + f];
      if (id >= 0 && id < vocab[f]) {
        memcpy(o, tables[f] + id * dim[f], sizeof(float) * dim[f]);
      } else {
        memset(o, 0, sizeof(float) * dim[f]);
        ++bad;
      }
      o += dim[f];
    }
  }
  return bad;
}

/* DLRM dot interaction (paper cited at src/ctr/dlrm/model.py:7): x (B, n, D) -> out (B, n(n-1)/2),
 * out[b, i(i-1)/2 + j] = <x[b,i], x[b,j]>, i > j; k-ordered fp32 accumulation. */
void orc_pairwise_dot_f32(const float* x, int64_t B, int32_t n, int32_t D, float* out,
                          int64_t out_stride) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    const float* xb = x + b * (int64_t)n * D;
    float* o = out + b * out_stride;
    for (int i = 1; i < n; ++i)
      for (int j = 0; j < i; ++j) {
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc += xb[i * D + k] * xb[j * D + k];
        o[i * (i - 1) / 2 + j] = acc;
      }
  }
}

/* The reference op sequence of the DLRM sparse stage, unfused like TF would run it:
 * gather+concat into emb (B, F*D) [src/ctr/dlrm/model.py:45], append dense_fea as vector F
 * [:48], pairwise dots, out = [dots (P), dense (D)].  All tables share dim D (<= 256).
 * emb_scratch: (B, (F+1)*D) caller-provided. */
void orc_dlrm_gather_dot_f32(const float* const* tables, const int64_t* vocab, int32_t F, int32_t D,
                             const int32_t* ids, const float* dense, int64_t B, float* emb_scratch,
                             float* out) {
  const int n = F + 1;
  const int P = n * (n - 1) / 2;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    float* x = emb_scratch + b * (int64_t)n * D;
    for (int f = 0; f < F; ++f) {
      const int64_t id = ids[b * F + f];
      if (id >= 0 && id < vocab[f])
        memcpy(x + f * D, tables[f] + id * D, sizeof(float) * D);
      else
        memset(x + f * D, 0, sizeof(float) * D);
    }
    memcpy(x + F * D, dense + b * D, sizeof(float) * D);
    float* o = out + b * (int64_t)(P + D);
    for (int i = 1; i < n; ++i)
      for (int j = 0; j < i; ++j) {
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc += x[i * D + k] * x[j * D + k];
        o[i * (i - 1) / 2 + j] = acc;
      }
    memcpy(o + P, dense + b * D, sizeof(float) * D);
  }
}

/* FM layer, src/ctr/layers/modules.py:57-72 (2-D second input).  Accumulates in double. */
void orc_fm_layer_f32(const float* first, int32_t L1, const float* w, const float* second,
                      int32_t M, int64_t B, float* out) {
  double first_order = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : first_order)
  for (int64_t b = 0; b < B; ++b) {
    double s = 0.0;
    for (int j = 0; j < L1; ++j) s += (double)first[b * L1 + j] * w[j];
    first_order += s;
  }
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    double s = 0.0, q = 0.0;
    for (int j = 0; j < M; ++j) {
      const double v = second[b * M + j];
      s += v;
      q += v * v;
    }
    out[b] = (float)(first_order + 0.5 * (s * s - q));
  }
}

/* CrossNetwork, src/ctr/layers/modules.py:105-112.  w, bv: (L, dim). */
void orc_cross_f32(const float* x, int32_t dim, const float* w, const float* bv, int32_t L,
                   int64_t B, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    const float* x0 = x + b * (int64_t)dim;
    float* xl = out + b * (int64_t)dim;
    memcpy(xl, x0, sizeof(float) * dim);
    for (int l = 0; l < L; ++l) {
      double s = 0.0;
      for (int k = 0; k < dim; ++k) s += (double)xl[k] * w[l * dim + k];
      for (int k = 0; k < dim; ++k) xl[k] = (float)((double)x0[k] * s + bv[l * dim + k] + xl[k]);
    }
  }
}
