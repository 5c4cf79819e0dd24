/* oracle/synth_corpus.c -- TEST INFRASTRUCTURE (see oracle/__init__.py), never linked into the product.
 *
 * Seeded synthetic corpora at BASELINE.json's retrieval sizes (configs[2]: 10^6 SPLADE rows, configs[3]: 1.25 * 10^6 x 768
 * dense rows per GPU) for the full-size parity tests.  Counter-based (every value is a hash of (seed, row, column)), so the
 * output does not depend on the number of OpenMP threads. */
#include <stdint.h>
#include <stddef.h>

static inline uint64_t mix(uint64_t x) { /* splitmix64 finaliser */
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

/* Row lengths L_i in [lo, hi]; returns through indptr[0..n] (prefix sums). */
void synth_sparse_lengths(int64_t n_docs, int lo, int hi, uint64_t seed, int64_t* indptr) {
  indptr[0] = 0;
  for (int64_t i = 0; i < n_docs; ++i) indptr[i + 1] = indptr[i] + lo + (int64_t)(mix(seed ^ mix((uint64_t)i)) % (uint64_t)(hi - lo + 1));
}

/* Document i takes one term from each of L_i equal strata of the vocabulary (unique, strictly ascending); weights on a
 * 1/64 grid in (0, 3]. */
void synth_sparse_fill(int64_t n_docs, int vocab, uint64_t seed, const int64_t* indptr, int32_t* terms, float* values) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n_docs; ++i) {
    const int64_t a = indptr[i], len = indptr[i + 1] - a;
    const int64_t width = vocab / len;
    const uint64_t h = mix(seed * 0x2545F4914F6CDD1Dull + (uint64_t)i);
    for (int64_t p = 0; p < len; ++p) {
      const uint64_t r = mix(h + (uint64_t)p);
      terms[a + p] = (int32_t)(p * width + (int64_t)((r >> 32) % (uint64_t)width));
      values[a + p] = (float)((r & 0xFFFF) % 192 + 1) * (1.0f / 64.0f);
    }
  }
}

/* [n, dim] fp32 rows with arbitrary values in about (-1, 1): 16 random bits / 32768 plus a per-column offset, so
 * products and partial sums round in fp32 like real embeddings do. */
void synth_dense_rows(int64_t n, int dim, uint64_t seed, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = mix(seed * 0x2545F4914F6CDD1Dull + (uint64_t)i);
    float* x = out + (size_t)i * dim;
    for (int c = 0; c < dim; c += 4) {
      uint64_t r = mix(h + (uint64_t)c);
      for (int j = 0; j < 4 && c + j < dim; ++j, r >>= 16)
        x[c + j] = (float)(int16_t)(r & 0xFFFF) * (1.0f / 32768.0f) + (float)((c + j) % 7 - 3) * 0.0137f;
    }
  }
}

/* [n, dim] rows on the grid {-8 .. 8} / 8: bf16-exact values whose products and sums are exact in fp32 in any order, so
 * a bf16-row shard must reproduce the oracle bit for bit -- and exact score ties are frequent, so the (score desc,
 * id asc) order is exercised at scale. */
void synth_dense_grid(int64_t n, int dim, uint64_t seed, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t h = mix(seed * 0x2545F4914F6CDD1Dull + (uint64_t)i);
    float* x = out + (size_t)i * dim;
    for (int c = 0; c < dim; c += 8) {
      uint64_t r = mix(h + (uint64_t)c);
      for (int j = 0; j < 8 && c + j < dim; ++j, r >>= 8) x[c + j] = (float)((int)((r & 0xFF) % 17) - 8) * 0.125f;
    }
  }
}
