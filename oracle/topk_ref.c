/* oracle/topk_ref.c -- TEST INFRASTRUCTURE (see oracle/__init__.py), never linked into the product.
 *
 * CPU restatement of exact dot-product top-k, the search the reference delegates to Milvus
 * (verbatim_rag/vector_stores/milvus_base.py:239-259; third-party pymilvus 2.6.17 / milvus-lite
 * 2.5.1 absent from the reference tree -> parity unpinned at the reference level; anchored on
 * the call sites: COSINE/IP metric (milvus_local.py:109-129), `limit=top_k`, hits ordered by
 * distance).  Total order (score desc, id asc).  fp32 accumulation:
 *   dense : acc = fmaf(x[c], q[c], acc) for c ascending  (exact for dyadic-grid data in any order)
 *   sparse: acc = fmaf(value_j, q[term_j], acc) in CSR order; hits need a shared term (score > 0).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void insert_hit(float* s, int64_t* id, int k, float score, int64_t doc) {
  /* list sorted by (score desc, id asc); empty slots have id = -1 */
  int i = k - 1;
  if (id[i] >= 0 && !(score > s[i] || (score == s[i] && doc < id[i]))) return;
  while (i > 0 && (id[i - 1] < 0 || score > s[i - 1] || (score == s[i - 1] && doc < id[i - 1]))) {
    s[i] = s[i - 1];
    id[i] = id[i - 1];
    --i;
  }
  s[i] = score;
  id[i] = doc;
}

void dense_topk_ref(const float* rows, int64_t n, int dim, const float* queries, int nq, int k, float* scores,
                    int64_t* ids) {
#pragma omp parallel for schedule(static)
  for (int q = 0; q < nq; ++q) {
    float* s = scores + (size_t)q * k;
    int64_t* id = ids + (size_t)q * k;
    for (int i = 0; i < k; ++i) {
      s[i] = -INFINITY;
      id[i] = -1;
    }
    const float* qv = queries + (size_t)q * dim;
    for (int64_t r = 0; r < n; ++r) {
      const float* x = rows + (size_t)r * dim;
      float acc = 0.f;
      for (int c = 0; c < dim; ++c) acc = fmaf(x[c], qv[c], acc);
      insert_hit(s, id, k, acc, r);
    }
  }
}

void sparse_topk_ref(int64_t n_docs, const int64_t* indptr, const int32_t* indices, const float* values, int vocab,
                     const int64_t* q_indptr, const int32_t* q_indices, const float* q_values, int nq, int k,
                     float* scores, int64_t* ids) {
#pragma omp parallel for schedule(static)
  for (int q = 0; q < nq; ++q) {
    float* qd = (float*)calloc((size_t)vocab, sizeof(float));
    for (int64_t j = q_indptr[q]; j < q_indptr[q + 1]; ++j) qd[q_indices[j]] = q_values[j];
    float* s = scores + (size_t)q * k;
    int64_t* id = ids + (size_t)q * k;
    for (int i = 0; i < k; ++i) {
      s[i] = -INFINITY;
      id[i] = -1;
    }
    for (int64_t d = 0; d < n_docs; ++d) {
      float acc = 0.f;
      for (int64_t j = indptr[d]; j < indptr[d + 1]; ++j) acc = fmaf(values[j], qd[indices[j]], acc);
      if (acc > 0.f) insert_hit(s, id, k, acc, d);
    }
    free(qd);
  }
}

/* ---- Blocked forms for the full-size parity tests (BASELINE configs[2] / [3]: 10^6 .. 10^7 rows, 10^2 .. 10^3 queries).
 * Same arithmetic, bit for bit: every (row, query) accumulator is still the chain fmaf(x[c], q[c], acc) in ascending c
 * (CSR order for sparse rows); only the loop nest changes -- QB queries share one pass over a row (their accumulators are
 * independent lanes of a vector FMA) and the rows are cut over the OpenMP threads, each thread keeping private top-k lists
 * that are merged by the same insert_hit at the end (the total order makes the merge independent of the thread count).
 * tests/test_oracle_golden.py checks blocked == scalar. */
#define QB 16

static void merge_lists(float* s, int64_t* id, const float* ls, const int64_t* lid, int nq, int k) {
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < k; ++i)
      if (lid[(size_t)q * k + i] >= 0) insert_hit(s + (size_t)q * k, id + (size_t)q * k, k, ls[(size_t)q * k + i], lid[(size_t)q * k + i]);
}

void dense_topk_ref_blocked(const float* rows, int64_t n, int dim, const float* queries, int nq, int k, float* scores,
                            int64_t* ids) {
  for (size_t i = 0; i < (size_t)nq * k; ++i) {
    scores[i] = -INFINITY;
    ids[i] = -1;
  }
#pragma omp parallel
  {
    float* ls = (float*)malloc((size_t)nq * k * sizeof(float));
    int64_t* lid = (int64_t*)malloc((size_t)nq * k * sizeof(int64_t));
    float* qt = (float*)aligned_alloc(64, (size_t)dim * QB * sizeof(float));
    for (size_t i = 0; i < (size_t)nq * k; ++i) {
      ls[i] = -INFINITY;
      lid[i] = -1;
    }
    for (int q0 = 0; q0 < nq; q0 += QB) {
      const int qb = nq - q0 < QB ? nq - q0 : QB;
      for (int c = 0; c < dim; ++c)
        for (int j = 0; j < QB; ++j) qt[(size_t)c * QB + j] = j < qb ? queries[(size_t)(q0 + j) * dim + c] : 0.f;
#pragma omp for schedule(static) nowait
      for (int64_t r = 0; r < n; ++r) {
        const float* x = rows + (size_t)r * dim;
        float acc[QB];
        for (int j = 0; j < QB; ++j) acc[j] = 0.f;
        for (int c = 0; c < dim; ++c) {
          const float xv = x[c];
          const float* qc = qt + (size_t)c * QB;
          for (int j = 0; j < QB; ++j) acc[j] = fmaf(xv, qc[j], acc[j]);
        }
        for (int j = 0; j < qb; ++j) insert_hit(ls + (size_t)(q0 + j) * k, lid + (size_t)(q0 + j) * k, k, acc[j], r);
      }
    }
#pragma omp critical
    merge_lists(scores, ids, ls, lid, nq, k);
    free(ls);
    free(lid);
    free(qt);
  }
}

void sparse_topk_ref_blocked(int64_t n_docs, const int64_t* indptr, const int32_t* indices, const float* values, int vocab,
                             const int64_t* q_indptr, const int32_t* q_indices, const float* q_values, int nq, int k,
                             float* scores, int64_t* ids) {
  for (size_t i = 0; i < (size_t)nq * k; ++i) {
    scores[i] = -INFINITY;
    ids[i] = -1;
  }
#pragma omp parallel
  {
    float* ls = (float*)malloc((size_t)nq * k * sizeof(float));
    int64_t* lid = (int64_t*)malloc((size_t)nq * k * sizeof(int64_t));
    float* qd = (float*)aligned_alloc(64, (size_t)vocab * QB * sizeof(float));
    for (size_t i = 0; i < (size_t)nq * k; ++i) {
      ls[i] = -INFINITY;
      lid[i] = -1;
    }
    for (int q0 = 0; q0 < nq; q0 += QB) {
      const int qb = nq - q0 < QB ? nq - q0 : QB;
      memset(qd, 0, (size_t)vocab * QB * sizeof(float));
      for (int j = 0; j < qb; ++j)
        for (int64_t t = q_indptr[q0 + j]; t < q_indptr[q0 + j + 1]; ++t) qd[(size_t)q_indices[t] * QB + j] = q_values[t];
#pragma omp for schedule(static) nowait
      for (int64_t d = 0; d < n_docs; ++d) {
        float acc[QB];
        for (int j = 0; j < QB; ++j) acc[j] = 0.f;
        for (int64_t t = indptr[d]; t < indptr[d + 1]; ++t) {
          const float v = values[t];
          const float* qc = qd + (size_t)indices[t] * QB;
          for (int j = 0; j < QB; ++j) acc[j] = fmaf(v, qc[j], acc[j]);
        }
        for (int j = 0; j < qb; ++j)
          if (acc[j] > 0.f) insert_hit(ls + (size_t)(q0 + j) * k, lid + (size_t)(q0 + j) * k, k, acc[j], d);
      }
    }
#pragma omp critical
    merge_lists(scores, ids, ls, lid, nq, k);
    free(ls);
    free(lid);
    free(qd);
  }
}
