/* vrag_amd.h -- C ABI of the MI355X (gfx950) hot-path library `libvrag_amd.so`.
 *
 * The reference (KRLabsOrg/verbatim-rag) is pure Python and has no FFI of its own; these
 * entry points are what a Python binding (ctypes) calls in place of the
 * third-party arithmetic the reference delegates to:
 *
 *   vrag_encoder_*         <- transformers ModernBertModel.forward, called from
 *                             packages/core/verbatim_core/extractor_models/model.py:75
 *                             (QAModel.forward) and extractors.py:260-268 (legacy qa_model path)
 *   vrag_encoder_*qa*      <- QAModel sentence head, extractor_models/model.py:82-113
 *   vrag_encoder_*token*   <- ModernBertForTokenClassification head used by the v2 highlighter
 *                             `.process()` (extractors.py:213-221)
 *   vrag_encoder_*splade*  <- SparseEncoder.encode (MLM logits -> max_s log1p(relu))
 *                             verbatim_rag/embedding_providers.py:127-166
 *   vrag_encoder_*pool*    <- SentenceTransformer.encode (pool + L2 normalise)
 *                             verbatim_rag/embedding_providers.py:73-77
 *
 * Conventions: every function returns 0 (VRAG_OK) or a negative status; the message for the
 * last failure on the calling thread is vrag_last_error().  Handles are thread-safe (one
 * internal mutex per handle; callers may hit one handle from asyncio.to_thread workers like
 * extractors.py:48-54 does).  `stream` arguments are a hipStream_t passed as void* (NULL = the
 * handle's own stream).  load_* = host->device upload, run_* = device kernels only,
 * read_* = device->host (synchronises the stream).  No callbacks, no global state.
 */
#ifndef VRAG_AMD_H
#define VRAG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRAG_OK 0
#define VRAG_ERR_INVALID (-1) /* bad argument / shape / state */
#define VRAG_ERR_HIP (-2)     /* a HIP runtime call failed */
#define VRAG_ERR_CAPACITY (-3)/* batch does not fit the workspace the handle was created with */
#define VRAG_ERR_NO_DEVICE (-4)

#define VRAG_ABI_VERSION 6

/* MFMA operand type of an encoder handle.  bf16: fp32's exponent range (safe for any checkpoint), 8 significant bits --
 * sentence logits within 3e-4 of the fp32 reference.  fp16: 11 significant bits at the same matrix-core rate, values
 * saturate at +-65504 -- what the per-token logits of the v2 highlighter need to stay within 1e-3. */
#define VRAG_OPERAND_BF16 0
#define VRAG_OPERAND_F16 1

typedef struct vrag_encoder vrag_encoder;

typedef struct vrag_encoder_config {
  int32_t vocab_size;
  int32_t hidden_size;         /* multiple of 128, <= 1024; head_dim is fixed at 64 */
  int32_t num_layers;
  int32_t num_heads;           /* hidden_size / 64 */
  int32_t intermediate_size;   /* multiple of 64 */
  int32_t global_every;        /* layer l is global attention iff l % global_every == 0 */
  int32_t sliding_window;      /* local layers keep |i-j| <= sliding_window (ModernBERT: 64) */
  float rope_theta_global;     /* 160000 */
  float rope_theta_local;      /* 10000 */
  float norm_eps;              /* 1e-5 */
  int32_t pad_token_id;
  int32_t max_seq_len;         /* longest sequence (RoPE table rows) */
  int32_t max_tokens;          /* workspace capacity: packed tokens per batch */
  int32_t max_seqs;            /* sequences per batch */
  int32_t max_ranges;          /* sentence / pooling ranges per batch */
  int32_t micro_batch_tokens;  /* 0 = whole batch per kernel; else split (cache blocking) */
  int32_t device;              /* HIP device ordinal */
  int32_t operand_dtype;       /* VRAG_OPERAND_BF16 (default) or VRAG_OPERAND_F16: type of the MFMA operands (weights,
                                  LayerNorm outputs, q/k/v/P, GeGLU output); accumulation, the residual stream, LayerNorm,
                                  softmax, RoPE and the heads are fp32 either way */
} vrag_encoder_config;

/* Host fp32 arrays, HF layouts ([out,in] row-major for nn.Linear weights). Per-layer arrays
 * have num_layers entries; attn_norm[0] is ignored (layer 0 has no attn_norm). */
typedef struct vrag_encoder_weights {
  const float* tok_embeddings;   /* [V, H]   embeddings.tok_embeddings.weight */
  const float* emb_norm;         /* [H]      embeddings.norm.weight */
  const float* const* attn_norm; /* [L][H]   layers.i.attn_norm.weight */
  const float* const* wqkv;      /* [L][3H,H] layers.i.attn.Wqkv.weight */
  const float* const* wo;        /* [L][H,H] layers.i.attn.Wo.weight */
  const float* const* mlp_norm;  /* [L][H]   layers.i.mlp_norm.weight */
  const float* const* wi;        /* [L][2I,H] layers.i.mlp.Wi.weight */
  const float* const* wo_mlp;    /* [L][H,I] layers.i.mlp.Wo.weight */
  const float* final_norm;       /* [H]      final_norm.weight */
} vrag_encoder_weights;

/* BERT-family encoders (BERT, DistilBERT: post-LN, biased linears, learned absolute positions, GELU MLP) --
 * the checkpoints the reference names for its embedding providers (`naver/splade-v3`,
 * `opensearch-neural-sparse-encoding-doc-v2-distill`: verbatim_rag/embedding_providers.py:120, README.md:122-125;
 * BAAI/bge-base: embedding_providers.py:55).  Replaces transformers BertModel / DistilBertModel.forward
 * (models/bert/modeling_bert.py, models/distilbert/modeling_distilbert.py) underneath
 * sentence-transformers' SparseEncoder / SentenceTransformer .encode.  head_dim must be 64 or 32 (32, e.g.
 * all-MiniLM-L6-v2 -- embedding_providers.py:55 -- runs zero-padded on the head_dim-64 kernels). */
typedef struct vrag_bert_config {
  int32_t vocab_size;
  int32_t hidden_size;              /* multiple of 128, <= 1024 */
  int32_t num_layers;
  int32_t num_heads;                /* hidden_size / 64 or hidden_size / 32 */
  int32_t intermediate_size;        /* multiple of 128 */
  int32_t max_position_embeddings;  /* rows of position_embeddings */
  float norm_eps;                   /* 1e-12 */
  int32_t pad_token_id;
  int32_t max_seq_len;              /* <= max_position_embeddings */
  int32_t max_tokens;
  int32_t max_seqs;
  int32_t max_ranges;
  int32_t micro_batch_tokens;
  int32_t device;
  int32_t operand_dtype;            /* VRAG_OPERAND_BF16 / VRAG_OPERAND_F16 */
} vrag_bert_config;

/* Host fp32 arrays, HF layouts. wqkv/bqkv are the query, key, value matrices / biases concatenated
 * along the output dimension ([3H,H] / [3H]).  DistilBERT: token_type_row = NULL. */
typedef struct vrag_bert_weights {
  const float* word_embeddings;      /* [V, H] */
  const float* position_embeddings;  /* [P, H] */
  const float* token_type_row;       /* [H] = token_type_embeddings[0] (single-segment inputs) or NULL */
  const float* emb_norm_w;           /* [H] embeddings.LayerNorm.weight */
  const float* emb_norm_b;           /* [H] */
  const float* const* wqkv;          /* [L][3H,H] */
  const float* const* bqkv;          /* [L][3H] */
  const float* const* wo;            /* [L][H,H]  attention.output.dense / out_lin */
  const float* const* bo;            /* [L][H] */
  const float* const* attn_norm_w;   /* [L][H]    attention.output.LayerNorm / sa_layer_norm */
  const float* const* attn_norm_b;   /* [L][H] */
  const float* const* w1;            /* [L][I,H]  intermediate.dense / ffn.lin1 */
  const float* const* b1;            /* [L][I] */
  const float* const* w2;            /* [L][H,I]  output.dense / ffn.lin2 */
  const float* const* b2;            /* [L][H] */
  const float* const* out_norm_w;    /* [L][H]    output.LayerNorm / output_layer_norm */
  const float* const* out_norm_b;    /* [L][H] */
} vrag_bert_weights;

const char* vrag_last_error(void);
int vrag_abi_version(void);
/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int vrag_device_count(void);

int vrag_encoder_create(const vrag_encoder_config* cfg, const vrag_encoder_weights* w, vrag_encoder** out);
/* Same opaque handle type: load_batch / run / load_ranges / run_pool / run_splade / read_* work on it
 * unchanged (there is no final LayerNorm to apply; pooling averages the hidden states directly). */
int vrag_bert_encoder_create(const vrag_bert_config* cfg, const vrag_bert_weights* w, vrag_encoder** out);
void vrag_encoder_destroy(vrag_encoder* enc);

/* Heads (host fp32, HF layouts). */
int vrag_encoder_set_qa_head(vrag_encoder* enc, const float* w /*[labels,H]*/, const float* b /*[labels]*/,
                             int32_t num_labels);
int vrag_encoder_set_token_head(vrag_encoder* enc, const float* dense_w /*[H,H]*/, const float* norm_w /*[H]*/,
                                const float* cls_w /*[labels,H]*/, const float* cls_b /*[labels]*/,
                                int32_t num_labels);
/* decoder_w == NULL ties the decoder to tok_embeddings (ModernBertForMaskedLM). */
int vrag_encoder_set_mlm_head(vrag_encoder* enc, const float* dense_w /*[H,H]*/, const float* norm_w /*[H]*/,
                              const float* decoder_w /*[V,H] or NULL*/, const float* decoder_b /*[V]*/);

/* MLM head with biases (BertForMaskedLM cls.predictions / DistilBertForMaskedLM vocab_transform,
 * vocab_layer_norm, vocab_projector): logits = decoder(LN(gelu(dense(h) + dense_b)) * norm_w + norm_b) + decoder_b.
 * dense_b / norm_b may be NULL (ModernBERT); decoder_w == NULL ties the decoder to the word embeddings. */
int vrag_encoder_set_mlm_head_ex(vrag_encoder* enc, const float* dense_w /*[H,H]*/, const float* dense_b /*[H]*/,
                                 const float* norm_w /*[H]*/, const float* norm_b /*[H]*/,
                                 const float* decoder_w /*[V,H] or NULL*/, const float* decoder_b /*[V]*/);

/* Operand precision of the MLM / SPLADE head GEMMs; takes effect at the NEXT vrag_encoder_set_mlm_head* (which may be called
 * again on a handle: the previous head's images are released and rebuilt in the chosen form): split_operands != 0
 * (the default) carries activations and weights as (value, remainder) pairs of the operand type -- the dense layer as
 * three accumulating GEMMs, the decoder as one GEMM over K = 3H -- so that every SPLADE weight max_s log1p(relu(logit))
 * stays within 2e-3 of the fp32 arithmetic of SparseEncoder.encode (embedding_providers.py:127-166; nothing averages
 * operand rounding away under a max); 0 = plain 16-bit operands: a third of the decoder work, weights within ~1e-2. */
int vrag_encoder_set_head_precision(vrag_encoder* enc, int32_t split_operands);

/* Sentence-pair inputs (cross-encoder reranking: sentence-transformers CrossEncoder over
 * BertForSequenceClassification, verbatim_rag/rerankers.py:109-134).  BERT-family handles only.
 * set_token_types: the whole token_type_embeddings table [n_types, H] (the creator only takes row 0).
 * load_token_types: per-token segment ids of the batch loaded last (concatenation order of load_batch);
 *   stays in effect until the next load_batch.
 * set_pair_head / run_pair_head: logits[s] = cls_w . tanh(pooler_w . h[first token of s] + pooler_b) + cls_b
 *   (BertPooler + classifier, transformers models/bert/modeling_bert.py:451-463,1115-1119). */
int vrag_encoder_set_token_types(vrag_encoder* enc, const float* table /*[n_types, H]*/, int32_t n_types);
int vrag_encoder_load_token_types(vrag_encoder* enc, const int32_t* types /*[n_tokens]*/, void* stream);
int vrag_encoder_set_pair_head(vrag_encoder* enc, const float* pooler_w /*[H,H]*/, const float* pooler_b /*[H]*/,
                               const float* cls_w /*[labels,H]*/, const float* cls_b /*[labels]*/, int32_t num_labels);
int vrag_encoder_run_pair_head(vrag_encoder* enc, void* stream);
int vrag_encoder_read_pair_logits(vrag_encoder* enc, float* logits /*[n_seqs, labels]*/, void* stream);

/* Packed batch: `ids` is the plain concatenation of n_seqs unpadded sequences of lengths
 * seq_lens[i] (positions restart at 0 per sequence, like the reference's B=1 forward). */
int vrag_encoder_load_batch(vrag_encoder* enc, const int32_t* ids, const int32_t* seq_lens, int32_t n_seqs,
                            void* stream);
/* Embedding + all encoder layers; leaves the fp32 residual stream on the device. */
int vrag_encoder_run(vrag_encoder* enc, void* stream);
/* Same, stopping after `n_layers` layers (debug / per-layer parity). */
int vrag_encoder_run_layers(vrag_encoder* enc, int32_t n_layers, void* stream);

/* Inclusive token ranges inside sequences (sentence boundaries or pooling spans). */
int vrag_encoder_load_ranges(vrag_encoder* enc, const int32_t* seq_idx, const int32_t* start, const int32_t* end,
                             int32_t n_ranges, void* stream);
/* final LayerNorm + mean over each range + Linear(H, labels)  -> device logits [n_ranges, labels] */
int vrag_encoder_run_qa_head(vrag_encoder* enc, void* stream);
int vrag_encoder_read_qa_logits(vrag_encoder* enc, float* logits /*[n_ranges, labels]*/, void* stream);
/* final LayerNorm + mean over each range (+ L2 normalise) -> [n_ranges, H] */
int vrag_encoder_run_pool(vrag_encoder* enc, int32_t normalize, void* stream);
int vrag_encoder_read_pool(vrag_encoder* enc, float* out /*[n_ranges, H]*/, void* stream);

/* Token-classification head: logits for every packed token, in the caller's concatenation order. */
int vrag_encoder_run_token_head(vrag_encoder* enc, void* stream);
int vrag_encoder_read_token_logits(vrag_encoder* enc, float* logits /*[n_tokens, labels]*/, void* stream);

/* SPLADE head: rows[s][v] = max over the tokens of sequence s of log1p(relu(mlm_logit)). */
int vrag_encoder_run_splade(vrag_encoder* enc, void* stream);
int vrag_encoder_read_splade(vrag_encoder* enc, float* rows /*[n_seqs, V]*/, void* stream);
/* The same rows compacted on the device: for sequence s, counts[s] entries (vocabulary index ascending, weight >
 * threshold; threshold 0 = every non-zero, the `embed_batch` rule of embedding_providers.py:161-163; 1e-6 = the
 * `embed_text` rule :141-145) at indices/values[s*cap_per_seq ...].  counts[] is always exact; if any count exceeds
 * cap_per_seq the call returns VRAG_ERR_CAPACITY (nothing is truncated silently) and the caller re-reads with a
 * larger capacity or falls back to vrag_encoder_read_splade. */
int vrag_encoder_read_splade_sparse(vrag_encoder* enc, float threshold, int32_t cap_per_seq, int32_t* counts /*[n_seqs]*/,
                                    int32_t* indices /*[n_seqs, cap_per_seq]*/, float* values /*[n_seqs, cap_per_seq]*/,
                                    void* stream);

/* Debug / parity: final-LayerNorm hidden states (or the raw residual stream), caller order. */
int vrag_encoder_read_hidden(vrag_encoder* enc, int32_t apply_final_norm, float* out /*[n_tokens, H]*/,
                             void* stream);

/* One-call convenience for the extractor path: load_batch + load_ranges + run + run_qa_head + read. */
int vrag_encoder_extract_qa(vrag_encoder* enc, const int32_t* ids, const int32_t* seq_lens, int32_t n_seqs,
                            const int32_t* rng_seq, const int32_t* rng_start, const int32_t* rng_end,
                            int32_t n_ranges, float* logits);

/* Launch-bound batches (at most 8 192 packed rows in one micro-batch -- a query's handful of chunks, the reference's call
 * shape verbatim_rag/core.py:238-255 with k = 5): the layer schedule of a (rows, attention blocks, layers) geometry is
 * captured into a HIP graph the second time it is seen and replayed afterwards; results are bit-identical to the eager
 * launches (same kernels, same arguments).  enable: 0 = eager only, > 0 = row limit for graph replay, < 0 = leave as is;
 * *replays = graph launches so far, *cached = instantiated graphs (either may be NULL).  VRAG_GRAPHS=0 disables at create. */
int vrag_encoder_graph_stats(vrag_encoder* enc, int32_t enable, int64_t* replays, int32_t* cached);

/* fp16 operands (VRAG_OPERAND_F16) saturate at +-65504 instead of overflowing.  *saturated = 1 if any fp32 -> fp16
 * operand conversion on this device (weights at load time, LayerNorm-fold copies, q / k / v, GeGLU outputs, attention
 * outputs) has had to clamp since the last reset: the logits computed meanwhile are not to be trusted -- re-run with
 * VRAG_OPERAND_BF16 (checkpoints with activation outliers beyond fp16's range; the flag is per process and device, not
 * per handle).  Synchronises the device. */
int vrag_encoder_f16_saturated(vrag_encoder* enc, int32_t reset, int32_t* saturated);

/* Per-kernel-class timing with HIP events recorded on the launch stream.
 * classes: see VRAG_PROF_* ; ms[i] = summed event time, launches[i] = launch count since reset. */
#define VRAG_PROF_EMBED 0
#define VRAG_PROF_LAYERNORM 1
#define VRAG_PROF_GEMM_QKV 2
#define VRAG_PROF_ATTN_GLOBAL 3
#define VRAG_PROF_ATTN_LOCAL 4
#define VRAG_PROF_GEMM_WO 5
#define VRAG_PROF_GEMM_WI 6
#define VRAG_PROF_GEMM_WO_MLP 7
#define VRAG_PROF_HEAD 8
#define VRAG_PROF_QKV_ATTN_GLOBAL 9 /* fused Wqkv + RoPE + attention kernel (csrc/qkv_attn.hip), global layers */
#define VRAG_PROF_QKV_ATTN_LOCAL 10 /* the same, banded layers */
#define VRAG_PROF_COUNT 11
/* Micro-batches (cfg.micro_batch_tokens) are issued on 2 internal streams by default so that the
 * HBM-bound kernels of one overlap the MFMA-bound kernels of the other; 1 serialises them. */
int vrag_encoder_set_concurrency(vrag_encoder* enc, int32_t n_streams);
/* enabled: 0 = off, 1 = a HIP event pair around every launch, n > 1 = around every n-th launch of each class (the totals
 * vrag_encoder_read_profile returns are then the timed launches' mean x the launches issued). */
int vrag_encoder_set_profiling(vrag_encoder* enc, int32_t enabled);
/* Launch-bound configuration: GEMMs over at most `rows` token rows (a query's handful of chunks) use 128x128 / 64x64 tiles with
 * deeper LDS rings and a K split over waves instead of the 256x256 throughput tiles; 0 disables it, a negative value only
 * reads.  Process-wide; returns the threshold in effect (default 8192).  The two configurations sum fp32 partial products in
 * different orders, so tests pin one or the other through this call. */
int vrag_set_small_batch_rows(int32_t rows);
int vrag_encoder_read_profile(vrag_encoder* enc, float* ms /*[VRAG_PROF_COUNT]*/,
                              int64_t* launches /*[VRAG_PROF_COUNT]*/, int32_t reset);

/* ------------------------------------------------------------------------------------------
 * Exact dot-product top-k (what the reference delegates to Milvus:
 * verbatim_rag/vector_stores/milvus_base.py:239-259, metric types milvus_local.py:109-129).
 * Order: (score desc, id asc), ties included.  Missing hits: id = -1, score = -inf.  1 <= k <= 1024 for the search
 * calls (Milvus' `limit`, milvus_base.py:244-277: the reference asks for top_k or 2*top_k): lists of up to 64 come from
 * one device pass, longer ones as exact pages of 64 (page p+1 admits only keys below the last key of page p) on the
 * scalar kernels with fp32 queries; `*_run_resident` re-runs a single pass (k <= 64).
 * Dense rows are stored bf16 (dtype 0) or fp32 (dtype 1); COSINE == IP on rows/queries the caller
 * L2-normalised.  ids are row numbers in insertion order (the caller adds its shard base).
 * dtype 2 = fp32 rows (every result bit-identical to dtype 1: scores are the sequential fp32 fmaf chain over the fp32 rows)
 * plus a bf16 prefilter image of them (+50 % memory): vrag_dense_index_search, for k <= 16, ranks the image first, proves
 * from the image's MEASURED error bound that its candidates contain the exact top-k (else the full fp32 scan answers that
 * query) and re-scores them exactly -- half the bytes for a small batch, one read of the shard per batch instead of one per 32
 * queries.  Round 6 routes every batch through it: one to four queries share ONE pass over the image that collects every row
 * within twice the bound of an entry threshold (more than 4096 of them = the full scan); 5 .. 256 queries take the same idea
 * on the tiled score GEMM (thresholds from 65 536 rows -- a sample of the shard's tiles --, one appending pass, exact re-score of the lists); larger
 * batches rank the image for 64 candidates per query with a sufficiency test.  vrag_dense_index_search_device takes the same
 * routes with the full scan enqueued behind per-query flags instead of a host decision.
 * bf16 rows (dtype 0): one query streams the shard on the scalar kernel; two or more take the tiled score GEMM (shards of >= 4 096
 * rows, dim % 64 == 0) -- the shard read once per batch -- whose scores are exact on bf16-representable data and otherwise differ
 * from the fp32 chain by summation order.
 */
typedef struct vrag_dense_index vrag_dense_index;
int vrag_dense_index_create(int32_t dim, int64_t capacity, int32_t dtype, int32_t device, vrag_dense_index** out);
void vrag_dense_index_destroy(vrag_dense_index* ix);
int64_t vrag_dense_index_size(vrag_dense_index* ix);
int vrag_dense_index_add(vrag_dense_index* ix, const float* rows /*[n,dim] host fp32*/, int64_t n);
/* The same from DEVICE memory (rows: fp32 [n, dim] on the index's device, e.g. vrag_encoder pooled embeddings that never
 * visited the host): converted / copied on `stream` (NULL = the legacy default stream); returns when the rows are in place. */
int vrag_dense_index_add_device(vrag_dense_index* ix, const float* rows /*[n,dim] device fp32*/, int64_t n, void* stream);
int vrag_dense_index_search(vrag_dense_index* ix, const float* queries /*[nq,dim] host*/, int32_t nq, int32_t k,
                            float* scores /*[nq,k]*/, int64_t* ids /*[nq,k]*/, void* stream);
/* Re-runs the kernels of the last search on the device-resident queries (no copies, no sync). */
int vrag_dense_index_run_resident(vrag_dense_index* ix, int32_t nq, int32_t k, void* stream);
/* The same search (k <= 64) with the result lists left in DEVICE memory -- what a rank contributes to the cross-GPU
 * exchange (serves verbatim_rag/vector_stores/milvus_base.py:239-259 on a row-sharded corpus):
 * out_ids[q][j] = row_map[row] when `row_map` (device int64[n_map], the caller's local row -> global row table) is
 * given -- rows at or beyond n_map become -1 -- else id_base + row; missing hits -1 / -inf.  Queries are host memory
 * (the call returns once they have been uploaded); the kernels are only enqueued on `stream`: no synchronisation and no
 * device->host copy.  `stream` = NULL means the legacy default stream here (not the handle's own stream): whatever
 * consumes the lists next -- the all-gather -- is ordered against the stream the caller named. */
int vrag_dense_index_search_device(vrag_dense_index* ix, const float* queries /*[nq,dim] host*/, int32_t nq, int32_t k,
                                   const int64_t* row_map /*device or NULL*/, int64_t n_map, int64_t id_base,
                                   float* out_scores /*[nq,k] device*/, int64_t* out_ids /*[nq,k] device*/, void* stream);

/* Sparse (SPLADE) rows in CSR, term ids < vocab <= 65536; only documents sharing a term with the
 * query (score > 0) are hits, like an inverted index.  ids are CSR row numbers. */
typedef struct vrag_sparse_index vrag_sparse_index;
int vrag_sparse_index_create(int32_t vocab, int64_t n_docs, const int64_t* indptr, const int32_t* indices,
                             const float* values, int32_t device, vrag_sparse_index** out);
void vrag_sparse_index_destroy(vrag_sparse_index* ix);
int vrag_sparse_index_stats(vrag_sparse_index* ix, int64_t* n_docs, int64_t* nnz, int64_t* padded_nnz);
int vrag_sparse_index_search(vrag_sparse_index* ix, const int64_t* q_indptr, const int32_t* q_indices,
                             const float* q_values, int32_t nq, int32_t k, float* scores /*[nq,k]*/,
                             int64_t* ids /*[nq,k]*/, void* stream);
int vrag_sparse_index_run_resident(vrag_sparse_index* ix, int32_t nq, int32_t k, void* stream);
/* Device-resident result lists, as vrag_dense_index_search_device. */
int vrag_sparse_index_search_device(vrag_sparse_index* ix, const int64_t* q_indptr, const int32_t* q_indices,
                                    const float* q_values, int32_t nq, int32_t k, const int64_t* row_map /*device or NULL*/,
                                    int64_t n_map, int64_t id_base, float* out_scores /*[nq,k] device*/,
                                    int64_t* out_ids /*[nq,k] device*/, void* stream);

/* Cross-shard merge of per-shard top-k lists (the reference has no sharding -- this is the step after the
 * all-gather of `[n_lists][nq][k_in]` (fp32 score, global row id) lists, each sorted by (score desc, id asc) with
 * id = -1 entries as a tail).  Writes the first k_out entries of the merged order per query (-inf / -1 padded).
 * Global ids must be < 2^32 and unique across lists.  `*_list_stride`: bytes between consecutive lists (0 = dense,
 * nq*k_in elements) -- lets the gathered buffer of one packed all-gather ([ids | scores] per rank) be merged in
 * place.  on_device = 1: all four pointers are device memory on `device` and the kernel is only enqueued on
 * `stream`; on_device = 0: host pointers, the call copies in and out and synchronises. */
int vrag_topk_merge(const float* scores, const int64_t* ids, int32_t n_lists, int32_t nq, int32_t k_in, int32_t k_out,
                    int64_t score_list_stride, int64_t id_list_stride, float* out_scores, int64_t* out_ids,
                    int32_t on_device, int32_t device, void* stream);

/* ------------------------------------------------------------------------------------------
 * The exchange step of sharded retrieval: one process per GPU, the corpus row-sharded, every rank answers the
 * (replicated) query batch on its own shard with vrag_*_index_search_device and contributes its packed lists
 *     payload = [ global ids  int64 x nq*k_in | scores fp32 x nq*k_in | pad to 8 bytes ]       (device memory)
 * to ONE RCCL all-gather over xGMI; every rank then merges the world's lists on its GPU.  RCCL is bound at run time (the
 * copy already mapped into the process -- e.g. torch's -- else the system librccl.so.1; $VRAG_RCCL_LIB overrides).
 * The host framework only has to carry the unique id from rank 0 to the other ranks (torch.distributed broadcast, MPI, a
 * file): vrag_comm_get_unique_id on rank 0, vrag_comm_create (collective: every rank calls it) everywhere.
 * Calls on one communicator must be issued in the same order on every rank (RCCL's rule). */
#define VRAG_COMM_ID_BYTES 128
typedef struct vrag_comm vrag_comm;
int vrag_comm_get_unique_id(uint8_t* id /*[VRAG_COMM_ID_BYTES]*/);
int vrag_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, vrag_comm** out);
void vrag_comm_destroy(vrag_comm* comm);
int vrag_comm_info(vrag_comm* comm, int32_t* rank, int32_t* world, int32_t* rccl_version /* any may be NULL */);
/* ncclAllGather of nbytes per rank, enqueued on `stream` (NULL = the legacy default stream); recv holds world * nbytes. */
int vrag_comm_allgather(vrag_comm* comm, const void* send /*device*/, void* recv /*device*/, int64_t nbytes, void* stream);
/* The whole exchange: all-gather of `payload` into `gathered` (device scratch of world * payload bytes) + vrag_topk_merge in
 * place, both enqueued on `stream`; out_* are device [nq, k_out], identical on every rank.  No synchronisation. */
int vrag_topk_allgather_merge(vrag_comm* comm, const void* payload, void* gathered, int32_t nq, int32_t k_in, int32_t k_out,
                              float* out_scores /*device*/, int64_t* out_ids /*device*/, void* stream);

/* Sentence boundaries of a batch of chunk texts (the GPU-side sentence split): for every document the parts of
 *   re.split(r"(?<=[.!?])\s+", text), each stripped of surrounding white space, empty parts dropped
 * (packages/core/verbatim_core/extractors.py:190-195), as [start, end) BYTE offsets into the document's UTF-8 text.
 * `text` holds the documents back to back, document d is bytes doc_off[d] .. doc_off[d+1] (doc_off[0] = 0).  counts[d] is
 * the exact number of sentences; only the first `cap` of a document are stored at starts/ends[d * cap ...] -- the caller
 * re-splits a document with counts[d] > cap itself.  White space = Python's str.isspace set.  Host pointers; synchronous. */
int vrag_split_sentences(const uint8_t* text, const int64_t* doc_off /*[n_docs+1]*/, int32_t n_docs, int32_t cap,
                         int32_t* counts /*[n_docs]*/, int32_t* starts /*[n_docs, cap]*/, int32_t* ends /*[n_docs, cap]*/,
                         int32_t device);

/* Host-side packer of the (question, chunk) pairs of ONE question for the sentence-classification extractor -- the layout
 *   [CLS] q [SEP] s1 [SEP] s2 ... ([SEP])   with inclusive token ranges per sentence, budget = max_length - 2, sentences that do
 * not fit dropped from the first one that does not (extractor_models/dataset.py:127-243).  The question-independent pieces of a
 * chunk are cached by the caller: tail = [SEP] s1 [SEP] s2 ... (int32) and cum[i] = tokens of the first i + 1 `[SEP] sentence`
 * groups (int64); tails[p] / cums[p] are their ADDRESSES for pair p, n_groups[p] the sentence count.  q_ids = the question's ids
 * without a trailing [SEP] (q_len of them).  Per pair: kept[p] sentences fit (0 = none: the caller's general routine decides),
 * seq_lens[p] ids are appended to ids_out, kept[p] ranges to starts_out / ends_out; totals_out = {ids, ranges} written.
 * Pure host code (no device, no allocation); VRAG_ERR_CAPACITY when an output capacity would be exceeded. */
int vrag_pack_qa_pairs(const int32_t* q_ids, int32_t q_len, int32_t n_pairs, const uint64_t* tails, const uint64_t* cums,
                       const int32_t* n_groups, int32_t budget, int32_t sep_id, int32_t* ids_out, int64_t ids_cap,
                       int64_t* starts_out, int64_t* ends_out, int64_t ranges_cap, int32_t* seq_lens /*[n_pairs]*/,
                       int32_t* kept /*[n_pairs]*/, int64_t* totals_out /*[2]*/);

/* -inf / -1 lists in device memory: the contribution of a rank that holds none of the rows. */
int vrag_topk_fill_empty(float* scores /*[n] device*/, int64_t* ids /*[n] device*/, int64_t n, int32_t device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VRAG_AMD_H */
