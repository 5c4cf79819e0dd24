// C ABI (include/vrag_amd.h) over the gfx950 kernels: handle lifetime, weight packing,
// batch layout, the encoder schedule and the heads.  No torch types cross this boundary.
#include "../../include/vrag_amd.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "attention.h"
#include "qkv_attn.h"

#include <array>
#include <map>
#include "common.h"
#include "gemm_bf16.h"
#include "norm_heads.h"

namespace vrag {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------- weight packing kernels
template <typename T>
__global__ void cvt_rows_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst_, int rows_dst,
                                     int rows_src, int cols, int interleave_I,
                                     const float* __restrict__ col_scale, float* __restrict__ row_sum,
                                     bf16_t* __restrict__ dst_lo_ = nullptr) {
  // dst row r <- src row perm(r) (* col_scale per input column: a LayerNorm gain folded into the
  // weight); rows beyond rows_src are zero.  interleave_I > 0 applies the GeGLU interleave: each
  // 64-row group = 32 input rows (x1) then the 32 matching gate rows (x2).  row_sum[r] = sum over
  // the ROUNDED row (what the MFMA will actually multiply), fp32.  T = operand type (bf16 / fp16);
  // dst_lo (optional) receives the remainder  v - float(T(v))  in the same type (split-operand head GEMMs).
  T* dst = reinterpret_cast<T*>(dst_);
  T* dst_lo = reinterpret_cast<T*>(dst_lo_);
  __shared__ float red[4];
  const int r = blockIdx.x;
  int s = r;
  bool valid = true;
  if (interleave_I > 0) {
    const int g = r >> 6, w = r & 63;
    const int f = g * 32 + (w & 31);       // feature index; rows with f >= I are zero padding
    valid = f < interleave_I;
    s = w < 32 ? f : interleave_I + f;
  }
  float acc = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    float v = (valid && r < rows_dst && s < rows_src) ? src[(size_t)s * cols + c] : 0.f;
    if (col_scale) v *= col_scale[c];
    const T b = Op<T>::to(v);
    dst[(size_t)r * cols + c] = b;
    if (dst_lo) dst_lo[(size_t)r * cols + c] = Op<T>::to(v - (float)b);
    acc += (float)b;
  }
  if (row_sum) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) row_sum[r] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

static void launch_cvt_rows(int op_dtype, dim3 grid, hipStream_t st, const float* src, bf16_t* dst, int rows_dst, int rows_src,
                            int cols, int interleave_I, const float* col_scale, float* row_sum, bf16_t* dst_lo = nullptr) {
  if (op_dtype == kOpF16)
    hipLaunchKernelGGL(cvt_rows_bf16_kernel<f16_t>, grid, dim3(256), 0, st, src, dst, rows_dst, rows_src, cols, interleave_I,
                       col_scale, row_sum, dst_lo);
  else
    hipLaunchKernelGGL(cvt_rows_bf16_kernel<bf16_t>, grid, dim3(256), 0, st, src, dst, rows_dst, rows_src, cols, interleave_I,
                       col_scale, row_sum, dst_lo);
}

// fp32 matrix rows [rows_src, cols] -> [rows_dst, 3 cols] operand image [hi | hi | lo] (zero rows beyond rows_src): the
// weight side of a K = 3 cols split-operand GEMM whose A operand is [xhi | xlo | xhi] (norm_heads.hip layernorm_kernel,
// split3): sum = xhi.Whi + xlo.Whi + xhi.Wlo, the three products of (xhi + xlo).(Whi + Wlo) above fp32 resolution.
template <typename T>
__global__ void cvt_split3_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst_, int rows_dst, int rows_src, int cols) {
  T* dst = reinterpret_cast<T*>(dst_);
  const int r = blockIdx.x;
  if (r >= rows_dst) return;
  T* row = dst + (size_t)r * 3 * cols;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) {
    const float v = r < rows_src ? src[(size_t)r * cols + c] : 0.f;
    const T hi = Op<T>::to(v);
    row[c] = hi;
    row[cols + c] = hi;
    row[2 * cols + c] = Op<T>::to(v - (float)hi);
  }
}
static void launch_cvt_split3(int op_dtype, hipStream_t st, const float* src, bf16_t* dst, int rows_dst, int rows_src, int cols) {
  if (op_dtype == kOpF16) hipLaunchKernelGGL(cvt_split3_kernel<f16_t>, dim3(rows_dst), dim3(256), 0, st, src, dst, rows_dst, rows_src, cols);
  else hipLaunchKernelGGL(cvt_split3_kernel<bf16_t>, dim3(rows_dst), dim3(256), 0, st, src, dst, rows_dst, rows_src, cols);
}

// Row statistics from the per-segment partial sums left by the residual GEMM epilogue.  The sums are over (h - c)
// with c = shift[r] (the row's previous mean; null = 0):  d = mean(h - c),  var = E[(h-c)^2] - d^2  -- no
// cancellation however large |mean(h)| is --, mu_rel[r] = d (what the consumer GEMM's fold subtracts from its
// bf16(h - c) operand), shift[r] <- c + d (the absolute mean: next shift, and the post-LN rebuild's mean).
__global__ void ln_stats_finalize_kernel(const float* __restrict__ part, int ld, int np, int H, float eps, int rows,
                                         float* __restrict__ mu_rel, float* __restrict__ rstd, const float* shift_in,
                                         float* shift_out, float* __restrict__ shift_prev_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < np; ++i) {  // fixed order: deterministic
    s1 += part[((size_t)i * ld + r) * 2];      // slice-major partials [np][ld rows][2]: coalesced over the rows of a wave
    s2 += part[((size_t)i * ld + r) * 2 + 1];
  }
  const float d = s1 / (float)H;
  const float var = fmaxf(s2 / (float)H - d * d, 0.f);
  const float c = shift_in ? shift_in[r] : 0.f;
  mu_rel[r] = d;
  rstd[r] = 1.0f / sqrtf(var + eps);
  if (shift_prev_out) shift_prev_out[r] = c;   // what the split residual planes written before this call are relative to
  shift_out[r] = c + d;
}

// Packing metadata on the device: the host hands over the batch as it received it -- ids back to back plus
// (first row, first id, length) per sequence -- and this kernel lays the padding-free row image out: ids / position inside
// the sequence / sequence index for every row of the packed buffer, pad tokens in the alignment gaps, between micro-batches
// and over the stale rows of a longer previous batch.  One lane per row, sequence found by bisection.
__global__ void pack_layout_kernel(const int* __restrict__ packed, const int* __restrict__ seq_row, const int* __restrict__ seq_src,
                                   const int* __restrict__ seq_len, int n_seqs, int rows, int pad_id, int* __restrict__ ids,
                                   int* __restrict__ pos, int* __restrict__ tok_seq) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int lo = 0, hi = n_seqs;            // last sequence whose first row is <= r
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seq_row[mid] <= r) lo = mid;
    else hi = mid;
  }
  const int i = r - seq_row[lo];
  const bool in = i >= 0 && i < seq_len[lo];
  ids[r] = in ? packed[seq_src[lo] + i] : pad_id;
  pos[r] = in ? i : 0;
  tok_seq[r] = in ? lo : -1;
}

// One workgroup per sequence: stable compaction of a SPLADE row (weights are >= 0) into (index, value) pairs.
// Round 6: every thread owns a CONTIGUOUS range of the vocabulary (counts its survivors, one workgroup-wide exclusive scan,
// writes them at its offset) -- the first form walked the row 256 entries at a time with three barriers per step, 120 dependent
// round trips: 70 us for one query's row, 5 % of a single-question embedding call (profiles/r06_embed_latency.txt).
__global__ __launch_bounds__(256) void splade_compact_kernel(const float* __restrict__ rows, int V, int ld, float thr, int cap,
                                                             int* __restrict__ counts, int* __restrict__ idx,
                                                             float* __restrict__ val) {
  __shared__ int wsum[4];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = rows + (size_t)s * ld;
  const int per = ((V + 255) / 256 + 3) & ~3;   // entries per thread, a multiple of 4 (rows are 16-byte aligned: ld % 4 == 0)
  const int v_lo = tid * per, v_hi = min(V, v_lo + per);
  int mine = 0;
  for (int v = v_lo; v < v_hi; v += 4) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(row + v);   // reads up to 3 entries past V inside the padded row: masked below
#pragma unroll
    for (int j = 0; j < 4; ++j) mine += (v + j < v_hi && w[j] > thr) ? 1 : 0;
  }
  // exclusive scan over the 256 threads: wave scan by shuffles, then the four wave totals
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int off = incl - mine;
  for (int w2 = 0; w2 < wave; ++w2) off += wsum[w2];
  for (int v = v_lo; v < v_hi; v += 4) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(row + v);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v + j < v_hi && w[j] > thr) {
        if (off < cap) {
          idx[(size_t)s * cap + off] = v + j;
          val[(size_t)s * cap + off] = w[j];
        }
        ++off;
      }
  }
  if (tid == 255) counts[s] = off;   // the last thread's end offset = the row's total (its range may be empty: off = everything before it)
}

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct Layer {
  float* attn_norm = nullptr;
  bf16_t* wqkv = nullptr;
  bf16_t* wo = nullptr;
  float* mlp_norm = nullptr;
  bf16_t* wi = nullptr;  // interleaved rows
  bf16_t* wo_mlp = nullptr;
  // LayerNorm gains are folded into wqkv (attn_norm, layers > 0) and wi (mlp_norm); s_* = row sums
  float* s_qkv = nullptr;
  float* s_wi = nullptr;
  // the same Wqkv rows grouped per head, q(64) k(64) v(64), for the fused QKV + attention kernel (qkv_attn.hip)
  bf16_t* wqkv_h = nullptr;
  float* s_qkv_h = nullptr;
};

// BERT-family layer (post-LN, biased linears, GELU MLP): TF:models/bert/modeling_bert.py:282-286,340-344,
// models/distilbert/modeling_distilbert.py:236-239.
struct BertLayer {
  bf16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr;
  float *ln1_w = nullptr, *ln1_b = nullptr, *ln2_w = nullptr, *ln2_b = nullptr;
  // LayerNorm fold (default): wqkv (layers > 0) / w1 carry the gain of the LayerNorm that feeds them, bqkv / b1 carry
  // W . ln_bias on top of the linear bias, s_* are the row sums of the bf16 folded weights
  float *s_qkv = nullptr, *s_w1 = nullptr;
};

struct MicroBatch {
  int row0, row1;  // multiples of kRowPad (256): GEMM tiles never straddle micro-batches
  int blk0, blk1;    // global-layer q-block range
  int lblk0, lblk1;  // banded-layer q-block range
  int seq0 = 0, seq1 = 0;   // sequences of the micro-batch
  int max_len = 0;          // longest of them
  int tokens = 0;           // their lengths summed
  int grp0 = 0, grp1 = 0;   // work items of the fused QKV + attention kernel (groups of sequences, qkv_attn.h); empty = not packed
};

struct ProfRec {
  int cls;
  hipEvent_t a, b;
};

}  // namespace vrag

using namespace vrag;

struct vrag_encoder {
  vrag_encoder_config cfg{};
  std::recursive_mutex mu;
  hipStream_t own_stream = nullptr;
  hipStream_t aux_streams[4] = {nullptr, nullptr, nullptr, nullptr};  // micro-batch concurrency (up to 4 ways)
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  int n_streams = 1;
  bool ln_fold = true;   // LayerNorm folded into the producer/consumer GEMM epilogues (VRAG_LN_FOLD=0: separate LN kernels)
  int fused_qkv_attn = 1;   // Wqkv GEMM + RoPE + attention in one kernel per (sequence, head) when every sequence of the micro-batch has <= 512 tokens (VRAG_FUSED_QKV_ATTN)
  std::vector<void*> dev_allocs;
  std::vector<void*> host_allocs;
  // fp16 clamp reports (vrag_encoder_f16_saturated): the device addresses of the five translation units' flags and one pinned,
  // device-mapped word they are gathered into by one launch
  unsigned* sat_addr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  unsigned *sat_host = nullptr, *sat_host_dev = nullptr;

  // weights
  int op_dtype = kOpBf16;   // MFMA operand type of every 16-bit buffer of this handle (cfg.operand_dtype)
  int arch = 0;  // 0 = ModernBERT (pre-LN, RoPE, GeGLU, no biases); 1 = BERT family (post-LN, biases, learned positions)
  float* tok_emb = nullptr;
  float* emb_norm = nullptr;
  float* final_norm = nullptr;   // null for arch 1 (the stream is already normalised)
  std::vector<Layer> layers;
  // arch 1
  float *pos_emb = nullptr, *type_row = nullptr, *emb_norm_b = nullptr;
  std::vector<BertLayer> blayers;
  float *mlm_dense_b = nullptr, *mlm_norm_b = nullptr;
  // sentence-pair inputs + cross-encoder head (arch 1)
  float* type_table = nullptr;   // [n_types, H] token_type_embeddings
  int n_types = 0;
  int* d_types = nullptr;        // [cap_rows] per-token segment id of the current batch
  int* h_types = nullptr;
  bool types_loaded = false;
  float *pr_wp = nullptr, *pr_bp = nullptr, *pr_wc = nullptr, *pr_bc = nullptr;
  int pr_labels = 0;
  int *d_first_row = nullptr, *h_first_row = nullptr;   // [max_seqs]
  float* d_pair_out = nullptr;                          // [max_seqs, pr_labels]
  int i_pad = 0;        // GeGLU width padded to a multiple of 128 (2*i_pad = whole 256-wide GEMM tiles)
  int attn_w = 0;       // width of the q / k / v^T / o buffers = num_heads * 64 (> hidden_size when head_dim is 32)
  float q_scale = 0.125f * 1.4426950408889634f;  // head_dim^-1/2 * log2(e)
  float *cos_g = nullptr, *sin_g = nullptr, *cos_l = nullptr, *sin_l = nullptr;
  // heads
  float *qa_w = nullptr, *qa_b = nullptr;
  int qa_labels = 0;
  bf16_t *tk_dense = nullptr, *tk_dense_lo = nullptr;
  float *tk_norm = nullptr, *tk_w = nullptr, *tk_b = nullptr;
  int tk_labels = 0;
  bf16_t *mlm_dense = nullptr, *mlm_dec = nullptr;
  float *mlm_norm = nullptr, *mlm_bias = nullptr;
  int vpad = 0;
  // split-operand MLM head (vrag_encoder_set_head_precision, the default): dense weight remainder, the decoder as a
  // [vpad, 3H] image [Whi | Whi | Wlo] and the [cap_rows, 3H] operand image [xhi | xlo | xhi] its K = 3H GEMM reads
  bool mlm_split = true;
  bf16_t *mlm_dense_lo = nullptr, *mlm_dec3 = nullptr, *splade_a3 = nullptr;

  // workspace
  int cap_rows = 0;
  int *d_ids = nullptr, *d_pos = nullptr, *d_tokseq = nullptr;
  int *d_packed = nullptr, *d_seq_meta = nullptr;   // the batch as handed over: ids back to back; [3][max_seqs] first row / first id / length
  int* h_seq_meta = nullptr;
  int4 *d_groups = nullptr, *h_groups = nullptr;   // [max_seqs * 8] wave descriptors of the fused kernel's groups
  float* h = nullptr;
  bf16_t *a = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *o = nullptr, *act = nullptr;
  float* f32tmp = nullptr;  // [cap_rows, H] final hidden / head dense output
  float *st_part = nullptr, *ln_mu = nullptr, *ln_rstd = nullptr;  // folded-LayerNorm row statistics (ln_mu relative to ln_shift's previous value)
  float* ln_shift = nullptr;                                        // absolute row means = the next residual epilogue's shift
  float* ln_shift_prev = nullptr;                                   // the shift before the last advance: what the split planes in HBM are relative to
  unsigned char* lo8 = nullptr;                                     // [cap_rows * H] byte remainder plane of the split residual stream (gemm_bf16.h: 64 x 64 blocks)
  bool split_resid = true;                                          // VRAG_SPLIT_RESID=0: fp32 rows between all sub-layers (A/B)
  int *d_blk_start = nullptr, *d_blk_len = nullptr, *d_blk_q0 = nullptr;     // global-layer q-blocks
  int *d_lblk_start = nullptr, *d_lblk_len = nullptr, *d_lblk_q0 = nullptr;  // banded-layer q-blocks
  int cap_blocks = 0;
  int *d_rng_start = nullptr, *d_rng_end = nullptr;
  float* d_rng_out = nullptr;  // [max_ranges, max(H, labels)]
  float* d_tok_logits = nullptr;
  unsigned* d_splade = nullptr;  // allocated with the MLM head
  int *d_sp_cnt = nullptr, *d_sp_idx = nullptr;   // device-side compaction of the SPLADE rows (grown on demand)
  float* d_sp_val = nullptr;
  int sp_cap = 0;

  // pinned staging
  int* h_ids = nullptr;   // the batch's ids back to back (the row image is laid out on the device, pack_layout_kernel)
  int *h_blk_start = nullptr, *h_blk_len = nullptr, *h_blk_q0 = nullptr;
  int *h_lblk_start = nullptr, *h_lblk_len = nullptr, *h_lblk_q0 = nullptr;
  int *h_rng_start = nullptr, *h_rng_end = nullptr;
  float* h_out = nullptr;  // generic read-back staging
  size_t h_out_bytes = 0;

  // current batch
  int n_seqs = 0, n_tokens = 0, rows = 0, n_blocks = 0, n_ranges = 0;
  std::vector<int> seq_start, seq_len;
  std::vector<MicroBatch> mbs;
  bool ran = false;

  // Launch-bound batches (one query's handful of chunks: ~180 launches of a few microseconds each): the layer schedule of a
  // (rows, q-blocks, layers) geometry is captured into a HIP graph the second time the geometry is seen and replayed from
  // then on -- every per-batch quantity the kernels read (ids, positions, block descriptors) lives in device arrays that
  // load_batch refreshes, every pointer is a fixed workspace address, so only the geometry is baked into the graph.
  struct GraphEntry {
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    int seen = 0;
    uint64_t last_use = 0;
  };
  std::map<std::array<int, 7>, GraphEntry> graphs;
  uint64_t graph_clock = 0;
  int graph_rows_max = 8192;   // 0 disables (VRAG_GRAPHS=0); batches above it are throughput-bound, not launch-bound
  int64_t graph_replays = 0;

  // profiling
  bool prof_on = false;
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> prof_free;
  float prof_ms[VRAG_PROF_COUNT] = {0};
  int64_t prof_n[VRAG_PROF_COUNT] = {0};      // launches timed
  int64_t prof_seen[VRAG_PROF_COUNT] = {0};   // launches issued while profiling was on
  int prof_period = 1;                        // vrag_encoder_set_profiling(enabled): every enabled-th launch of a class is timed
};

namespace {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));          \
      return VRAG_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

#define ARG_CHECK(cond, ...)      \
  do {                            \
    if (!(cond)) {                \
      set_error(__VA_ARGS__);     \
      return VRAG_ERR_INVALID;    \
    }                             \
  } while (0)

template <typename T>
int dev_alloc(vrag_encoder* e, T** out, size_t count, bool zero = true) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
  HIP_TRY(hipMalloc(&p, bytes));
  e->dev_allocs.push_back(p);
  if (zero) {
    // hipMemset runs on the null stream, which does NOT order against the handle's non-blocking streams: wait for it
    // here (allocations are rare), or a kernel launched right after could have its output zeroed under it.
    HIP_TRY(hipMemset(p, 0, bytes));
    HIP_TRY(hipDeviceSynchronize());
  }
  *out = reinterpret_cast<T*>(p);
  return VRAG_OK;
}

// frees one tracked device allocation (a head that is set again) and clears the pointer
template <typename T>
void dev_release(vrag_encoder* e, T** ptr) {
  if (!*ptr) return;
  void* p = reinterpret_cast<void*>(*ptr);
  for (size_t i = 0; i < e->dev_allocs.size(); ++i)
    if (e->dev_allocs[i] == p) {
      e->dev_allocs.erase(e->dev_allocs.begin() + (long)i);
      (void)hipFree(p);
      break;
    }
  *ptr = nullptr;
}

template <typename T>
int host_alloc(vrag_encoder* e, T** out, size_t count) {
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, std::max<size_t>(count * sizeof(T), 256), hipHostMallocDefault));
  e->host_allocs.push_back(p);
  *out = reinterpret_cast<T*>(p);
  return VRAG_OK;
}

int upload_f32(vrag_encoder* e, float** out, const float* src, size_t count) {
  int rc = dev_alloc(e, out, count, false);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(*out, src, count * sizeof(float), hipMemcpyHostToDevice));
  return VRAG_OK;
}

// fp32 host matrix [rows_src, cols] -> bf16 device matrix [rows_dst, cols] (zero padded rows).
int upload_bf16(vrag_encoder* e, bf16_t** out, const float* src, int rows_src, int cols, int rows_dst,
                int interleave_I, float* stage, size_t stage_elems, const float* d_col_scale = nullptr,
                float** row_sum_out = nullptr, bf16_t** out_lo = nullptr) {
  int rc = dev_alloc(e, out, (size_t)rows_dst * cols, false);
  if (rc) return rc;
  bf16_t* d_lo = nullptr;
  if (out_lo) {
    if ((rc = dev_alloc(e, &d_lo, (size_t)rows_dst * cols, false))) return rc;
    *out_lo = d_lo;
  }
  float* d_row_sum = nullptr;
  if (row_sum_out) {
    if ((rc = dev_alloc(e, &d_row_sum, rows_dst, true))) return rc;
    *row_sum_out = d_row_sum;
  }
  // stream the fp32 source through the staging buffer in row chunks
  const int chunk_rows_max = (int)std::max<size_t>(1, stage_elems / cols);
  if (interleave_I > 0 || d_col_scale || row_sum_out || out_lo || rows_src <= chunk_rows_max) {
    ARG_CHECK((size_t)rows_src * cols <= stage_elems, "internal: staging buffer too small");
    HIP_TRY(hipMemcpy(stage, src, (size_t)rows_src * cols * sizeof(float), hipMemcpyHostToDevice));
    launch_cvt_rows(e->op_dtype, dim3(rows_dst), 0, stage, *out, rows_dst, rows_src, cols, interleave_I, d_col_scale, d_row_sum,
                    d_lo);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return VRAG_OK;
  }
  for (int r0 = 0; r0 < rows_dst; r0 += chunk_rows_max) {
    const int nr_dst = std::min(chunk_rows_max, rows_dst - r0);
    const int nr_src = std::max(0, std::min(chunk_rows_max, rows_src - r0));
    if (nr_src > 0)
      HIP_TRY(hipMemcpy(stage, src + (size_t)r0 * cols, (size_t)nr_src * cols * sizeof(float),
                        hipMemcpyHostToDevice));
    launch_cvt_rows(e->op_dtype, dim3(nr_dst), 0, stage, *out + (size_t)r0 * cols, nr_dst, nr_src, cols, 0, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  return VRAG_OK;
}

void rope_table(int max_pos, float theta, std::vector<float>& cs, std::vector<float>& sn) {
  // inv_freq_j = 1 / theta^(2j/64), j = 0..31; fp32 like the reference (TF:141,150-163)
  cs.resize((size_t)max_pos * 32);
  sn.resize((size_t)max_pos * 32);
  for (int j = 0; j < 32; ++j) {
    const float inv = 1.0f / powf(theta, (float)(2 * j) / 64.0f);
    for (int p = 0; p < max_pos; ++p) {
      const float f = (float)p * inv;
      cs[(size_t)p * 32 + j] = cosf(f);
      sn[(size_t)p * 32 + j] = sinf(f);
    }
  }
}

hipStream_t pick_stream(vrag_encoder* e, void* stream) {
  return stream ? reinterpret_cast<hipStream_t>(stream) : e->own_stream;
}

struct ProfScope {
  vrag_encoder* e;
  hipStream_t st;
  ProfRec rec{};
  bool on;
  ProfScope(vrag_encoder* e_, int cls, hipStream_t st_) : e(e_), st(st_), on(e_->prof_on) {
    if (!on) return;
    // sampling: every launch is counted, every prof_period-th launch of a class is bracketed by events (an event pair between
    // two launches keeps the second from starting under the first one's tail: 1.1 % of the bench step when every launch has one)
    on = (e->prof_seen[cls]++ % e->prof_period) == 0;
    if (!on) return;
    auto get = [&]() {
      hipEvent_t ev;
      if (!e->prof_free.empty()) {
        ev = e->prof_free.back();
        e->prof_free.pop_back();
      } else {
        (void)hipEventCreate(&ev);
      }
      return ev;
    };
    rec.cls = cls;
    rec.a = get();
    rec.b = get();
    (void)hipEventRecord(rec.a, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.b, st);
    e->prof_pending.push_back(rec);
  }
};

int check_ready(vrag_encoder* e) {
  ARG_CHECK(e != nullptr, "null encoder handle");
  ARG_CHECK(e->n_seqs > 0, "no batch loaded (call vrag_encoder_load_batch first)");
  return VRAG_OK;
}

// Groups of the fused QKV + attention kernel for sequences seq0 .. seq1 - 1: best-fit-decreasing bins of eight 64-token wave
// slots over ALL of the micro-batch's sequences (a wave's descriptor carries its own first row, so the sequences of a group need
// not be neighbours in the packed buffer -- fused_pack_groups of qkv_attn.hip, first fit over consecutive sequences, is the
// simple form the diagnostics use).  A workgroup costs what a full one costs, and ceil(S / 64) slots per sequence is all the
// kernel wastes then: 86 % fill for pairs of 64-512 tokens in ANY order (70 % with consecutive first fit).
static int pack_groups_best_fit(const int* seq_row, const int* seq_len, int seq0, int seq1, int4* out) {
  std::vector<int> by_need[9];
  for (int s = seq0; s < seq1; ++s) by_need[std::min(8, (seq_len[s] + 63) / 64)].push_back(s);
  std::vector<int> open_with[9];   // open groups by free slots
  std::vector<int> used_of;        // slots taken per group
  int n = 0;
  for (int need = 8; need >= 1; --need)
    for (int s : by_need[need]) {
      int g = -1;
      for (int room = need; room <= 8 && g < 0; ++room)   // best fit: the open group with the least room that still takes it
        if (!open_with[room].empty()) {
          g = open_with[room].back();
          open_with[room].pop_back();
        }
      if (g < 0) {
        g = n++;
        used_of.push_back(0);
        for (int w = 0; w < 8; ++w) out[g * 8 + w] = int4{0, 0, 0, 0};
      }
      const int used = used_of[g];
      for (int j = 0; j < need; ++j) out[g * 8 + used + j] = int4{seq_row[s] + 64 * j, seq_len[s], used, 0};
      used_of[g] = used + need;
      open_with[8 - used_of[g]].push_back(g);
    }
  return n;
}

// Does this micro-batch take the fused Wqkv + RoPE + attention kernel (qkv_attn.hip)?  One rule for the schedule and for the
// HIP-graph cache key: a graph captured on one path must never be replayed for a batch that takes the other.
static bool fused_attention_for(const vrag_encoder* e, const MicroBatch& mb) {
  if (!e->fused_qkv_attn || mb.max_len > kFusedMaxSeq || mb.grp1 <= mb.grp0) return false;
  if (e->fused_qkv_attn == 2) return true;
  const int M = mb.row1 - mb.row0;
  // a workgroup costs what a full 512-token one costs: the kernel pays when its groups are FULL enough (tokens per 512-token
  // workgroup; kFusedMinFill, measured: bench.py ragged_leg with VRAG_FUSED_QKV_ATTN = 0 / 2)
  return !gemm_consumer_finalizes(M) && (int64_t)mb.tokens * 100 >= (int64_t)kFusedMinFillPct * kFusedMaxSeq * (mb.grp1 - mb.grp0);
}

int run_layers_locked(vrag_encoder* e, int n_layers, hipStream_t user_st) {
  const auto& c = e->cfg;
  const int H = c.hidden_size, I = e->i_pad;   // padded GeGLU width (zero rows of Wi / zero columns of mlp.Wo)
  const int Tp = e->cap_rows;
  // Micro-batches touch disjoint rows of every buffer, so with n_streams == 2 they are issued on two
  // internal streams: the HBM-bound kernels of one (LayerNorm, epilogue tails) overlap the MFMA-bound
  // GEMM main loops of the other.  Fork/join with events on the caller's stream.
  const bool fork = e->n_streams > 1 && e->mbs.size() > 1;
  if (fork) {
    HIP_TRY(hipEventRecord(e->ev_fork, user_st));
    for (int i = 0; i < e->n_streams; ++i) HIP_TRY(hipStreamWaitEvent(e->aux_streams[i], e->ev_fork, 0));
  }
  for (size_t mbi = 0; mbi < e->mbs.size(); ++mbi) {
    const MicroBatch& mb = e->mbs[mbi];
    hipStream_t st = fork ? e->aux_streams[mbi % e->n_streams] : user_st;
    const int r0 = mb.row0, M = mb.row1 - mb.row0;
    {
      ProfScope ps(e, VRAG_PROF_EMBED, st);
      HIP_TRY(launch_embed_ln(e->d_ids + r0, e->tok_emb, e->emb_norm, c.norm_eps, H, M, e->h + (size_t)r0 * H,
                              e->a + (size_t)r0 * H, st, nullptr, nullptr, nullptr, nullptr, nullptr, e->op_dtype));
    }
    for (int l = 0; l < n_layers; ++l) {
      const Layer& L = e->layers[l];
      const bool global = (l % c.global_every) == 0;
      // Default: LayerNorm is its own HBM-streaming kernel (it overlaps the other stream's GEMM main
      // loops).  With ln_fold the residual GEMM epilogues leave bf16(h) in `a` plus per-row partial sums,
      // a tiny kernel turns those into (mu, rstd) and the consumer GEMM (gain folded into its weight)
      // normalises in its epilogue -- fewer bytes, but epilogue work is not overlapped (measured slower).
      const bool fold = e->ln_fold;
      // Split residual stream (gemm_bf16.h): between the sub-layers of layers >= 1 the stream lives as the operand copy `a` plus
      // a 16-bit low plane; the fp32 rows of `h` are written by layer 0 and by the LAST sub-layer of the run only (what the heads
      // and vrag_encoder_read_hidden read).
      const bool split = fold && e->split_resid;
      const bool more_layers = l + 1 < n_layers;   // a further layer of THIS run consumes the fold outputs
      auto layer_norm = [&](const float* gain) -> int {
        ProfScope ps(e, VRAG_PROF_LAYERNORM, st);
        HIP_TRY(launch_layernorm(e->h + (size_t)r0 * H, gain, c.norm_eps, H, M, e->a + (size_t)r0 * H, nullptr, st, nullptr, nullptr,
                                 e->op_dtype));
        return VRAG_OK;
      };
      if (!fold && l > 0) {
        int rc = layer_norm(L.attn_norm);
        if (rc) return rc;
      }
      // `first`: no previous mean to shift by (c = 0)
      // launch-bound batches: the consumer GEMM (small-row configuration) finishes the statistics itself
      const bool consumer_stats = fold && gemm_consumer_finalizes(M);
      // sequences of <= 512 tokens, throughput-sized micro-batch: one kernel per (sequence, head) instead of the QKV GEMM and the
      // attention launch -- Q, K and V^T never leave the CU (qkv_attn.hip)
      // The fused kernel spends a 512-token workgroup per (sequence, head) whatever the sequence's length (waves past its end
      // idle): it wins from a mean length of ~270 tokens up and loses below (tools/bench_seq_len.py: 42.7 vs 38.6 ms per
      // 131 072-token step at 192 tokens, a tie at 256, 35.8 vs 39.9 at 512), so short-chunk batches keep the packed two-kernel path.
      const bool fused_attn = L.wqkv_h && fused_attention_for(e, mb);
      auto finalize_stats = [&](bool first, bool for_qkv = false) -> int {
        if (consumer_stats && !(for_qkv && fused_attn)) return VRAG_OK;
        ProfScope ps(e, VRAG_PROF_LAYERNORM, st);
        hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, st,
                           e->st_part + (size_t)r0 * 2, e->cap_rows, H / 64, H, c.norm_eps, M, e->ln_mu + r0,
                           e->ln_rstd + r0, first ? (const float*)nullptr : e->ln_shift + r0, e->ln_shift + r0, e->ln_shift_prev + r0);
        HIP_TRY(hipGetLastError());
        return VRAG_OK;
      };
      auto stats_for_consumer = [&](GemmParams& g) {
        if (!consumer_stats) return;
        g.stats_in = e->st_part + (size_t)r0 * 2;
        g.stats_ld = e->cap_rows;
        g.ln_shift = e->ln_shift + r0;
        g.ln_shift_prev = e->ln_shift_prev + r0;
        g.fin_eps = c.norm_eps;
      };
      if (fused_attn) {
        QkvAttnParams f{};
        f.op_dtype = e->op_dtype;
        f.x = e->a;
        f.w = L.wqkv_h;
        if (fold && l > 0) {
          f.ln_mu = e->ln_mu;
          f.ln_rstd = e->ln_rstd;
          f.ln_s = L.s_qkv_h;
        }
        f.rope_cos = global ? e->cos_g : e->cos_l;
        f.rope_sin = global ? e->sin_g : e->sin_l;
        f.rope_rows = c.max_seq_len;
        f.o = e->o;
        f.groups = e->d_groups + (size_t)mb.grp0 * 8;
        f.n_groups = mb.grp1 - mb.grp0;
        f.H = H;
        f.nh = c.num_heads;
        f.Tp = Tp;
        f.window = c.sliding_window;
        f.q_scale = 0.125f * 1.4426950408889634f;
        ProfScope ps(e, global ? VRAG_PROF_QKV_ATTN_GLOBAL : VRAG_PROF_QKV_ATTN_LOCAL, st);
        HIP_TRY(launch_qkv_attention(f, !global, st));
      } else {
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = e->a + (size_t)r0 * H;
        g.W = L.wqkv;
        if (fold && l > 0) {  // layer 0 consumes the embedding LayerNorm output directly (no attn_norm)
          g.ln_mu = e->ln_mu + r0;
          g.ln_rstd = e->ln_rstd + r0;
          g.ln_s = L.s_qkv;
          stats_for_consumer(g);
        }
        g.M = M;
        g.N = 3 * H;
        g.K = H;
        g.q = e->q + (size_t)r0 * H;
        g.k = e->k + (size_t)r0 * H;
        g.vt = e->vt + r0;
        g.vt_ld = Tp;
        g.rope_cos = global ? e->cos_g : e->cos_l;
        g.rope_sin = global ? e->sin_g : e->sin_l;
        g.pos = e->d_pos + r0;
        g.hidden = H;
        g.q_scale = 0.125f * 1.4426950408889634f;  // head_dim^-1/2 * log2(e): attention softmax runs in exp2 units
        ProfScope ps(e, VRAG_PROF_GEMM_QKV, st);
        HIP_TRY(launch_gemm(EPI_QKV_ROPE, g, st));
      }
      {
        AttnParams ap{};
        ap.op_dtype = e->op_dtype;
        ap.q = e->q;
        ap.k = e->k;
        ap.vt = e->vt;
        ap.o = e->o;
        ap.blk_seq_start = global ? e->d_blk_start + mb.blk0 : e->d_lblk_start + mb.lblk0;
        ap.blk_seq_len = global ? e->d_blk_len + mb.blk0 : e->d_lblk_len + mb.lblk0;
        ap.blk_q0 = global ? e->d_blk_q0 + mb.blk0 : e->d_lblk_q0 + mb.lblk0;
        ap.n_blocks = global ? mb.blk1 - mb.blk0 : mb.lblk1 - mb.lblk0;
        ap.H = H;
        ap.nh = c.num_heads;
        ap.Tp = Tp;
        ap.window = c.sliding_window;
        ProfScope ps(e, global ? VRAG_PROF_ATTN_GLOBAL : VRAG_PROF_ATTN_LOCAL, st);
        HIP_TRY(launch_attention(ap, !global, st));
      }
      }
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = e->o + (size_t)r0 * H;
        g.W = L.wo;
        g.M = M;
        g.N = H;
        g.K = H;
        g.out_f32 = e->h + (size_t)r0 * H;
        // The fold's per-row shift is the row's previous mean; the very first sub-layer has none, so layer 0's
        // mlp_norm runs as a stand-alone (two-pass) LayerNorm that also records the exact row means.
        const bool fold_here = fold && l > 0;
        if (fold_here) {
          g.resid_bf16 = e->a + (size_t)r0 * H;
          g.stats_part = e->st_part + (size_t)r0 * 2;
          g.stats_ld = e->cap_rows;
          g.ln_shift = e->ln_shift + r0;
          if (split) {   // layers >= 1: the stream arrives split (mlp Wo of the previous layer) and leaves split (this layer's mlp Wo reads it)
            g.lo_in = e->lo8 + (size_t)r0 * H;
            g.lo_out = e->lo8 + (size_t)r0 * H;
            g.ln_shift_prev = e->ln_shift_prev + r0;
          }
        }
        {
          ProfScope ps(e, VRAG_PROF_GEMM_WO, st);
          HIP_TRY(launch_gemm(EPI_RESIDUAL, g, st));
        }
        if (fold && l == 0) {   // a = bf16(normalised h) without the gain (folded into Wi), ln_shift = mean(h)
          ProfScope ps(e, VRAG_PROF_LAYERNORM, st);
          HIP_TRY(launch_layernorm(e->h + (size_t)r0 * H, nullptr, c.norm_eps, H, M, e->a + (size_t)r0 * H, nullptr, st, nullptr,
                                   e->ln_shift + r0, e->op_dtype));
        } else {
          int rc = fold ? finalize_stats(false) : layer_norm(L.mlp_norm);
          if (rc) return rc;
        }
      }
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = e->a + (size_t)r0 * H;
        g.W = L.wi;
        if (fold && l > 0) {
          g.ln_mu = e->ln_mu + r0;
          g.ln_rstd = e->ln_rstd + r0;
          g.ln_s = L.s_wi;
          stats_for_consumer(g);
        }
        g.M = M;
        g.N = 2 * I;
        g.K = H;
        g.out_bf16 = e->act + (size_t)r0 * I;
        ProfScope ps(e, VRAG_PROF_GEMM_WI, st);
        HIP_TRY(launch_gemm(EPI_GEGLU, g, st));
      }
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = e->act + (size_t)r0 * I;
        g.W = L.wo_mlp;
        g.M = M;
        g.N = H;
        g.K = I;
        g.out_f32 = e->h + (size_t)r0 * H;
        if (fold && more_layers) {  // the next layer's attn_norm is folded into its QKV GEMM
          g.resid_bf16 = e->a + (size_t)r0 * H;
          g.stats_part = e->st_part + (size_t)r0 * 2;
          g.stats_ld = e->cap_rows;
          g.ln_shift = e->ln_shift + r0;
          if (split) g.lo_out = e->lo8 + (size_t)r0 * H;
        }
        if (split && l > 0) {   // this layer's attention Wo left the stream split; the last layer of a run writes the fp32 rows again
          g.lo_in = e->lo8 + (size_t)r0 * H;
          g.ln_shift_prev = e->ln_shift_prev + r0;
          if (!g.resid_bf16) g.resid_bf16 = e->a + (size_t)r0 * H;   // the high plane is read from there
        }
        {
          ProfScope ps(e, VRAG_PROF_GEMM_WO_MLP, st);
          HIP_TRY(launch_gemm(EPI_RESIDUAL, g, st));
        }
        if (fold && more_layers) {
          int rc = finalize_stats(false, true);
          if (rc) return rc;
        }
      }
    }
  }
  if (fork) {
    for (int i = 0; i < e->n_streams; ++i) {
      HIP_TRY(hipEventRecord(e->ev_join[i], e->aux_streams[i]));
      HIP_TRY(hipStreamWaitEvent(user_st, e->ev_join[i], 0));
    }
  }
  e->ran = true;
  return VRAG_OK;
}

// BERT-family schedule (arch 1).  Post-LN: every sub-layer is  h <- LN(h + f(h) + b),  so the residual
// GEMM adds the linear bias in its epilogue and is followed by an in-place LayerNorm kernel that also refreshes
// the bf16 copy the next GEMM reads.  Opt-in (VRAG_BERT_LN_FOLD=1; round 3 measured it 2-3 % slower, on round 6's residual epilogues it is
// 4.5-6.4 % FASTER at 256 x 512 tokens and 6 % slower for one 20-token question: profiles/r06_embed_latency.txt): no LayerNorm kernels --
// the stream keeps the pre-LayerNorm sums t, the residual epilogues emit bf16(t) and the row statistics, the
// consumer GEMMs (gain folded into their weights, W . ln_bias into their biases) normalise in their epilogues,
// the next residual epilogue rebuilds LN(t) on the fly as its residual input, and only the output of the last
// layer of a run is materialised.
// The QKV GEMM reuses EPI_QKV_ROPE with identity rotation tables.
// TF:models/bert/modeling_bert.py:141-205 (attention), 282-286 / 340-344 (post-LN), 326-336 (GELU MLP);
// models/distilbert/modeling_distilbert.py:131-239.
int run_layers_bert_locked(vrag_encoder* e, int n_layers, hipStream_t user_st) {
  const auto& c = e->cfg;
  const int H = c.hidden_size, I = c.intermediate_size, Ha = e->attn_w;
  const int Tp = e->cap_rows;
  const bool fork = e->n_streams > 1 && e->mbs.size() > 1;
  if (fork) {
    HIP_TRY(hipEventRecord(e->ev_fork, user_st));
    for (int i = 0; i < e->n_streams; ++i) HIP_TRY(hipStreamWaitEvent(e->aux_streams[i], e->ev_fork, 0));
  }
  for (size_t mbi = 0; mbi < e->mbs.size(); ++mbi) {
    const MicroBatch& mb = e->mbs[mbi];
    hipStream_t st = fork ? e->aux_streams[mbi % e->n_streams] : user_st;
    const int r0 = mb.row0, M = mb.row1 - mb.row0;
    float* h = e->h + (size_t)r0 * H;
    bf16_t* a = e->a + (size_t)r0 * H;
    {
      ProfScope ps(e, VRAG_PROF_EMBED, st);
      HIP_TRY(launch_embed_ln(e->d_ids + r0, e->tok_emb, e->emb_norm, c.norm_eps, H, M, h, a, st, e->pos_emb,
                              e->d_pos + r0, e->types_loaded ? e->type_table : e->type_row, e->emb_norm_b,
                              e->types_loaded ? e->d_types + r0 : nullptr, e->op_dtype));
    }
    const bool fold = e->ln_fold;
    float* st_part = e->st_part + (size_t)r0 * 2;
    auto finalize_stats = [&](bool first) -> int {
      ProfScope ps(e, VRAG_PROF_LAYERNORM, st);
      hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, st, st_part, e->cap_rows, H / 64, H, c.norm_eps, M,
                         e->ln_mu + r0, e->ln_rstd + r0, first ? (const float*)nullptr : e->ln_shift + r0, e->ln_shift + r0, (float*)nullptr);
      HIP_TRY(hipGetLastError());
      return VRAG_OK;
    };
    for (int l = 0; l < n_layers; ++l) {
      const BertLayer& L = e->blayers[l];
      // fold mode: the stream `h` holds the PRE-LayerNorm sums t (layer 0 input excepted: the embedding LayerNorm
      // is eager); (ln_mu, ln_rstd) are the row statistics of the most recent t, `a` = bf16(t).
      const bool lazy_in = fold && l > 0;           // this layer's input is LN2 of layer l-1, still lazy
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = a;
        g.W = L.wqkv;
        g.bias = L.bqkv;
        if (lazy_in) {
          g.ln_mu = e->ln_mu + r0;
          g.ln_rstd = e->ln_rstd + r0;
          g.ln_s = L.s_qkv;
        }
        g.M = M;
        g.N = 3 * Ha;
        g.K = H;
        g.q = e->q + (size_t)r0 * Ha;
        g.k = e->k + (size_t)r0 * Ha;
        g.vt = e->vt + r0;
        g.vt_ld = Tp;
        g.rope_cos = e->cos_g;   // all ones
        g.rope_sin = e->sin_g;   // all zeros
        g.pos = e->d_pos + r0;
        g.hidden = Ha;
        g.q_scale = e->q_scale;
        ProfScope ps(e, VRAG_PROF_GEMM_QKV, st);
        HIP_TRY(launch_gemm(EPI_QKV_ROPE, g, st));
      }
      {
        AttnParams ap{};
        ap.op_dtype = e->op_dtype;
        ap.q = e->q;
        ap.k = e->k;
        ap.vt = e->vt;
        ap.o = e->o;
        ap.blk_seq_start = e->d_blk_start + mb.blk0;
        ap.blk_seq_len = e->d_blk_len + mb.blk0;
        ap.blk_q0 = e->d_blk_q0 + mb.blk0;
        ap.n_blocks = mb.blk1 - mb.blk0;
        ap.H = Ha;
        ap.nh = c.num_heads;
        ap.Tp = Tp;
        ap.window = 0;
        ProfScope ps(e, VRAG_PROF_ATTN_GLOBAL, st);
        HIP_TRY(launch_attention(ap, false, st));
      }
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = e->o + (size_t)r0 * Ha;
        g.W = L.wo;
        g.M = M;
        g.N = H;
        g.K = Ha;
        g.out_f32 = h;
        g.bias = L.bo;
        if (lazy_in) {
          const BertLayer& P = e->blayers[l - 1];
          g.res_mu = e->ln_shift + r0;   // absolute mean of t2 of the previous layer
          g.res_rstd = e->ln_rstd + r0;
          g.res_g = P.ln2_w;
          g.res_b = P.ln2_b;
        }
        if (fold) {
          g.resid_bf16 = a;
          g.stats_part = st_part;   // no shift here: a post-LN stream is re-centred by every LayerNorm (mean = O(1) sigma)
          g.stats_ld = e->cap_rows;
        }
        ProfScope ps(e, VRAG_PROF_GEMM_WO, st);
        HIP_TRY(launch_gemm(EPI_RESIDUAL, g, st));
      }
      if (fold) {
        int rc = finalize_stats(true);   // statistics of t1 = attention sub-layer sum (LN1 stays lazy)
        if (rc) return rc;
      } else {
        ProfScope ps(e, VRAG_PROF_LAYERNORM, st);
        HIP_TRY(launch_layernorm(h, L.ln1_w, c.norm_eps, H, M, a, h, st, L.ln1_b, nullptr, e->op_dtype));
      }
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = a;
        g.W = L.w1;
        g.M = M;
        g.N = I;
        g.K = H;
        g.out_bf16 = e->act + (size_t)r0 * I;
        g.bias = L.b1;
        g.act_gelu = 1;
        if (fold) {
          g.ln_mu = e->ln_mu + r0;
          g.ln_rstd = e->ln_rstd + r0;
          g.ln_s = L.s_w1;
        }
        ProfScope ps(e, VRAG_PROF_GEMM_WI, st);
        HIP_TRY(launch_gemm(EPI_BF16, g, st));
      }
      {
        GemmParams g{};
        g.op_dtype = e->op_dtype;
        g.A = e->act + (size_t)r0 * I;
        g.W = L.w2;
        g.M = M;
        g.N = H;
        g.K = I;
        g.out_f32 = h;
        g.bias = L.b2;
        if (fold) {
          g.res_mu = e->ln_shift + r0;   // absolute mean of t1
          g.res_rstd = e->ln_rstd + r0;
          g.res_g = L.ln1_w;
          g.res_b = L.ln1_b;
          g.resid_bf16 = a;
          g.stats_part = st_part;
          g.stats_ld = e->cap_rows;
        }
        ProfScope ps(e, VRAG_PROF_GEMM_WO_MLP, st);
        HIP_TRY(launch_gemm(EPI_RESIDUAL, g, st));
      }
      if (fold && l + 1 < n_layers) {
        int rc = finalize_stats(true);   // statistics of t2: the next layer consumes LN2 lazily
        if (rc) return rc;
      } else {
        // materialise the layer output (last layer of this run, or un-folded mode): h <- LN2(h), a <- bf16(h)
        ProfScope ps(e, VRAG_PROF_LAYERNORM, st);
        HIP_TRY(launch_layernorm(h, L.ln2_w, c.norm_eps, H, M, a, h, st, L.ln2_b, nullptr, e->op_dtype));
      }
    }
  }
  if (fork) {
    for (int i = 0; i < e->n_streams; ++i) {
      HIP_TRY(hipEventRecord(e->ev_join[i], e->aux_streams[i]));
      HIP_TRY(hipStreamWaitEvent(user_st, e->ev_join[i], 0));
    }
  }
  e->ran = true;
  return VRAG_OK;
}

int init_streams(vrag_encoder* e) {
  HIP_TRY(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
  e->n_streams = 2;  // micro-batches alternate between two internal streams (VRAG_STREAMS=1 disables)
  if (const char* ns = getenv("VRAG_STREAMS")) e->n_streams = std::min(4, std::max(1, atoi(ns)));
  if (const char* gr = getenv("VRAG_GRAPHS")) e->graph_rows_max = std::max(0, atoi(gr)) == 1 ? 8192 : std::max(0, atoi(gr));   // 0 = eager only
  for (int i = 0; i < 4; ++i) HIP_TRY(hipStreamCreateWithFlags(&e->aux_streams[i], hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming));
  return VRAG_OK;
}

// Device workspace + pinned staging, sized from e->cfg (shared by both encoder families).
int init_workspace(vrag_encoder* e) {
  const vrag_encoder_config* cfg = &e->cfg;
  const int H = cfg->hidden_size, I = cfg->intermediate_size;
  int rc;
#define TRY(x)            \
  do {                    \
    rc = (x);             \
    if (rc) return rc;    \
  } while (0)
  int n_mb = 1;
  // load_batch cuts a batch into micro-batches of PACKED rows (tokens + the 8-row alignment gaps between sequences): budget
  // the pad rows between micro-batches from the same quantity (8 * max_seqs can exceed a micro-batch)
  if (cfg->micro_batch_tokens > 0)
    n_mb = (int)(((int64_t)cfg->max_tokens + (int64_t)kSeqAlign * cfg->max_seqs) / std::max(1, cfg->micro_batch_tokens)) + 2;
  e->cap_rows = (int)align_up((int64_t)cfg->max_tokens + (int64_t)kSeqAlign * cfg->max_seqs + (int64_t)kRowPad * (n_mb + 1),
                              kRowPad);
  const size_t R = e->cap_rows;
  e->cap_blocks = cfg->max_tokens / 128 + cfg->max_seqs + 1;  // >= blocks of either granularity
  TRY(dev_alloc(e, &e->d_ids, R));
  TRY(dev_alloc(e, &e->d_pos, R));
  TRY(dev_alloc(e, &e->d_tokseq, R));
  TRY(dev_alloc(e, &e->h, R * H));
  TRY(dev_alloc(e, &e->a, R * H));
  if (e->attn_w <= 0) e->attn_w = H;
  const size_t Ha = e->attn_w;
  TRY(dev_alloc(e, &e->q, R * Ha));
  TRY(dev_alloc(e, &e->k, R * Ha));
  TRY(dev_alloc(e, &e->vt, R * Ha));
  TRY(dev_alloc(e, &e->o, R * Ha));
  if (e->i_pad <= 0) e->i_pad = I;
  TRY(dev_alloc(e, &e->act, R * (size_t)e->i_pad));
  TRY(dev_alloc(e, &e->f32tmp, R * H));
  TRY(dev_alloc(e, &e->st_part, R * (H / 64) * 2));
  TRY(dev_alloc(e, &e->ln_mu, R));
  TRY(dev_alloc(e, &e->ln_rstd, R));
  TRY(dev_alloc(e, &e->ln_shift, R));
  TRY(dev_alloc(e, &e->ln_shift_prev, R));
  TRY(dev_alloc(e, &e->lo8, (size_t)R * H));
  // the six q-block descriptor arrays are slices of ONE buffer (host and device alike): one upload per batch
  TRY(dev_alloc(e, &e->d_blk_start, (size_t)6 * e->cap_blocks));
  e->d_blk_len = e->d_blk_start + e->cap_blocks;
  e->d_blk_q0 = e->d_blk_start + 2 * e->cap_blocks;
  e->d_lblk_start = e->d_blk_start + 3 * e->cap_blocks;
  e->d_lblk_len = e->d_blk_start + 4 * e->cap_blocks;
  e->d_lblk_q0 = e->d_blk_start + 5 * e->cap_blocks;
  TRY(dev_alloc(e, &e->d_packed, (size_t)cfg->max_tokens));
  TRY(dev_alloc(e, &e->d_seq_meta, (size_t)3 * cfg->max_seqs));
  TRY(dev_alloc(e, &e->d_groups, (size_t)8 * cfg->max_seqs));
  TRY(dev_alloc(e, &e->d_rng_start, cfg->max_ranges));
  TRY(dev_alloc(e, &e->d_rng_end, cfg->max_ranges));
  TRY(dev_alloc(e, &e->d_rng_out, (size_t)cfg->max_ranges * H));
  TRY(host_alloc(e, &e->h_ids, R));
  TRY(host_alloc(e, &e->h_blk_start, (size_t)6 * e->cap_blocks));
  e->h_blk_len = e->h_blk_start + e->cap_blocks;
  e->h_blk_q0 = e->h_blk_start + 2 * e->cap_blocks;
  e->h_lblk_start = e->h_blk_start + 3 * e->cap_blocks;
  e->h_lblk_len = e->h_blk_start + 4 * e->cap_blocks;
  e->h_lblk_q0 = e->h_blk_start + 5 * e->cap_blocks;
  TRY(host_alloc(e, &e->h_seq_meta, (size_t)3 * cfg->max_seqs));
  TRY(host_alloc(e, &e->h_groups, (size_t)8 * cfg->max_seqs));
  TRY(host_alloc(e, &e->h_rng_start, cfg->max_ranges));
  TRY(host_alloc(e, &e->h_rng_end, cfg->max_ranges));
  // pad ids everywhere so never-loaded rows embed a valid token
  for (size_t i = 0; i < R; ++i) e->h_ids[i] = cfg->pad_token_id;
  HIP_TRY(hipMemcpy(e->d_ids, e->h_ids, R * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(hipDeviceSynchronize());
#undef TRY
  return VRAG_OK;
}

int ensure_h_out(vrag_encoder* e, size_t bytes) {
  if (bytes <= e->h_out_bytes) return VRAG_OK;
  void* p = nullptr;
  HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  e->host_allocs.push_back(p);
  e->h_out = reinterpret_cast<float*>(p);
  e->h_out_bytes = bytes;
  return VRAG_OK;
}

// Copies a padded device matrix [rows, cols] to the host and gathers the real tokens of each
// sequence into the caller's concatenation order.
int read_rows(vrag_encoder* e, const float* dsrc, int cols, float* out, hipStream_t st) {
  const size_t bytes = (size_t)e->rows * cols * sizeof(float);
  int rc = ensure_h_out(e, bytes);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(e->h_out, dsrc, bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  size_t o = 0;
  for (int s = 0; s < e->n_seqs; ++s) {
    memcpy(out + o * cols, e->h_out + (size_t)e->seq_start[s] * cols, (size_t)e->seq_len[s] * cols * sizeof(float));
    o += e->seq_len[s];
  }
  return VRAG_OK;
}

struct SatFlags {
  unsigned* a[5];
};
// one thread: OR of the translation units' clamp flags into the pinned word; a raised flag is cleared when `reset`
__global__ void f16_sat_gather_kernel(SatFlags f, int reset, unsigned* __restrict__ out) {
  unsigned any = 0u;
  for (int i = 0; i < 5; ++i) {
    const unsigned v = *f.a[i];
    any |= v;
    if (reset && v) *f.a[i] = 0u;
  }
  *out = any;
}

}  // namespace

extern "C" {

const char* vrag_last_error(void) { return g_last_error.c_str(); }
int vrag_abi_version(void) { return VRAG_ABI_VERSION; }

int vrag_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int vrag_encoder_create(const vrag_encoder_config* cfg, const vrag_encoder_weights* w, vrag_encoder** out) {
  ARG_CHECK(cfg && w && out, "null argument");
  *out = nullptr;
  const int H = cfg->hidden_size, I = cfg->intermediate_size, L = cfg->num_layers, V = cfg->vocab_size;
  ARG_CHECK(H > 0 && H % 128 == 0 && H <= 1024, "hidden_size must be a multiple of 128 and <= 1024 (got %d)", H);
  ARG_CHECK(cfg->num_heads * 64 == H, "head_dim must be 64: num_heads*64 != hidden_size (%d, %d)", cfg->num_heads, H);
  ARG_CHECK(I > 0 && I % 64 == 0, "intermediate_size must be a multiple of 64 (got %d)", I);
  ARG_CHECK(L > 0 && V > 0 && cfg->global_every > 0, "bad layer/vocab configuration");
  ARG_CHECK(cfg->max_tokens > 0 && cfg->max_seqs > 0 && cfg->max_seq_len > 0 && cfg->max_ranges > 0,
            "max_tokens/max_seqs/max_seq_len/max_ranges must be positive");
  ARG_CHECK(cfg->pad_token_id >= 0 && cfg->pad_token_id < V, "pad_token_id outside the vocabulary");
  ARG_CHECK(cfg->micro_batch_tokens >= 0, "micro_batch_tokens must be >= 0");
  ARG_CHECK(cfg->operand_dtype == VRAG_OPERAND_BF16 || cfg->operand_dtype == VRAG_OPERAND_F16, "operand_dtype: 0 = bf16, 1 = fp16");
  if (vrag_device_count() <= cfg->device) {
    set_error("no HIP device %d visible (the gfx950 library has no CPU fallback)", cfg->device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(cfg->device));

  vrag_encoder* e = new vrag_encoder();
  e->cfg = *cfg;
  e->op_dtype = cfg->operand_dtype;
  auto fail = [&](int rc) {
    vrag_encoder_destroy(e);
    return rc;
  };
  int rc;
#define TRY(x)                 \
  do {                         \
    rc = (x);                  \
    if (rc) return fail(rc);   \
  } while (0)

  TRY(init_streams(e));

  if (const char* lf = getenv("VRAG_LN_FOLD")) e->ln_fold = atoi(lf) != 0;
  if (const char* sr = getenv("VRAG_SPLIT_RESID")) e->split_resid = atoi(sr) != 0;
  // 0: always the QKV GEMM + attention launch; 1 (default): fused kernel for throughput-sized micro-batches; 2: launch-bound batches too
  if (const char* fq = getenv("VRAG_FUSED_QKV_ATTN")) e->fused_qkv_attn = atoi(fq);
  if (cfg->hidden_size != cfg->num_heads * 64) e->fused_qkv_attn = 0;

  // ---- weights
  const int Ip = (int)align_up(I, 128);
  e->i_pad = Ip;
  const size_t stage_elems = std::max<size_t>({(size_t)3 * H * H, (size_t)2 * Ip * H, (size_t)1 << 22});
  float* stage = nullptr;
  TRY(dev_alloc(e, &stage, stage_elems, false));
  std::vector<float> wo_pad;  // mlp.Wo with zero columns for the padded GeGLU features
  TRY(upload_f32(e, &e->tok_emb, w->tok_embeddings, (size_t)V * H));
  TRY(upload_f32(e, &e->emb_norm, w->emb_norm, H));
  TRY(upload_f32(e, &e->final_norm, w->final_norm, H));
  e->layers.resize(L);
  for (int l = 0; l < L; ++l) {
    Layer& ly = e->layers[l];
    if (l > 0) TRY(upload_f32(e, &ly.attn_norm, w->attn_norm[l], H));
    TRY(upload_f32(e, &ly.mlp_norm, w->mlp_norm[l], H));
    // LayerNorm folded into the consumer GEMMs: W' = W * gain, s = row sums of bf16(W')
    const bool fold = e->ln_fold;
    TRY(upload_bf16(e, &ly.wqkv, w->wqkv[l], 3 * H, H, 3 * H, 0, stage, stage_elems,
                    fold && l > 0 ? ly.attn_norm : nullptr, fold && l > 0 ? &ly.s_qkv : nullptr));
    if (e->fused_qkv_attn) {
      TRY(dev_alloc(e, &ly.wqkv_h, (size_t)3 * H * H, false));
      if (ly.s_qkv) TRY(dev_alloc(e, &ly.s_qkv_h, (size_t)3 * H + 64));   // the kernel's 256-float DMA of a head's 192 sums reads 64 floats on
      HIP_TRY(permute_qkv_heads(ly.wqkv, ly.s_qkv, H, cfg->num_heads, ly.wqkv_h, ly.s_qkv_h, nullptr));
      HIP_TRY(hipDeviceSynchronize());   // null-stream work does not order against the handle's non-blocking streams
    }
    TRY(upload_bf16(e, &ly.wo, w->wo[l], H, H, H, 0, stage, stage_elems));
    TRY(upload_bf16(e, &ly.wi, w->wi[l], 2 * I, H, 2 * Ip, I, stage, stage_elems, fold ? ly.mlp_norm : nullptr,
                    fold ? &ly.s_wi : nullptr));
    const float* wo2 = w->wo_mlp[l];
    if (Ip != I) {
      wo_pad.assign((size_t)H * Ip, 0.f);
      for (int r = 0; r < H; ++r) memcpy(&wo_pad[(size_t)r * Ip], wo2 + (size_t)r * I, (size_t)I * sizeof(float));
      wo2 = wo_pad.data();
    }
    TRY(upload_bf16(e, &ly.wo_mlp, wo2, H, Ip, H, 0, stage, stage_elems));
  }
  {
    std::vector<float> cs, sn;
    rope_table(cfg->max_seq_len, cfg->rope_theta_global, cs, sn);
    TRY(upload_f32(e, &e->cos_g, cs.data(), cs.size()));
    TRY(upload_f32(e, &e->sin_g, sn.data(), sn.size()));
    rope_table(cfg->max_seq_len, cfg->rope_theta_local, cs, sn);
    TRY(upload_f32(e, &e->cos_l, cs.data(), cs.size()));
    TRY(upload_f32(e, &e->sin_l, sn.data(), sn.size()));
  }

  TRY(init_workspace(e));
#undef TRY
  *out = e;
  return VRAG_OK;
}

int vrag_bert_encoder_create(const vrag_bert_config* cfg, const vrag_bert_weights* w, vrag_encoder** out) {
  ARG_CHECK(cfg && w && out, "null argument");
  *out = nullptr;
  const int H = cfg->hidden_size, I = cfg->intermediate_size, L = cfg->num_layers, V = cfg->vocab_size;
  const int P = cfg->max_position_embeddings;
  ARG_CHECK(H > 0 && H % 128 == 0 && H <= 1024, "hidden_size must be a multiple of 128 and <= 1024 (got %d)", H);
  ARG_CHECK(cfg->num_heads > 0 && (cfg->num_heads * 64 == H || cfg->num_heads * 32 == H),
            "head_dim must be 32 or 64 (num_heads %d, hidden_size %d)", cfg->num_heads, H);
  ARG_CHECK(I > 0 && I % 128 == 0, "intermediate_size must be a multiple of 128 (got %d)", I);
  ARG_CHECK(L > 0 && V > 0 && P > 0, "bad layer/vocab/position configuration");
  ARG_CHECK(cfg->max_tokens > 0 && cfg->max_seqs > 0 && cfg->max_seq_len > 0 && cfg->max_ranges > 0,
            "max_tokens/max_seqs/max_seq_len/max_ranges must be positive");
  ARG_CHECK(cfg->max_seq_len <= P, "max_seq_len %d exceeds max_position_embeddings %d", cfg->max_seq_len, P);
  ARG_CHECK(cfg->pad_token_id >= 0 && cfg->pad_token_id < V, "pad_token_id outside the vocabulary");
  ARG_CHECK(cfg->micro_batch_tokens >= 0, "micro_batch_tokens must be >= 0");
  ARG_CHECK(cfg->operand_dtype == VRAG_OPERAND_BF16 || cfg->operand_dtype == VRAG_OPERAND_F16, "operand_dtype: 0 = bf16, 1 = fp16");
  ARG_CHECK(w->word_embeddings && w->position_embeddings && w->emb_norm_w && w->emb_norm_b && w->wqkv && w->bqkv &&
                w->wo && w->bo && w->attn_norm_w && w->attn_norm_b && w->w1 && w->b1 && w->w2 && w->b2 &&
                w->out_norm_w && w->out_norm_b,
            "null weight array");
  if (vrag_device_count() <= cfg->device) {
    set_error("no HIP device %d visible (the gfx950 library has no CPU fallback)", cfg->device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(cfg->device));

  // head_dim 32 (all-MiniLM-L6-v2, the reference's default dense model: embedding_providers.py:55) runs on the
  // head_dim-64 kernels with every head zero-padded to 64: padded q/k dims add 0 to every score, padded v dims
  // produce zeros that meet zero columns of the padded output projection -- the arithmetic is unchanged.
  const int hd = H / cfg->num_heads, Ha = cfg->num_heads * 64;
  vrag_encoder* e = new vrag_encoder();
  e->arch = 1;
  e->op_dtype = cfg->operand_dtype;
  e->ln_fold = false;   // opt-in (VRAG_BERT_LN_FOLD=1): +4.5-6.4 % texts/s at 256 x 512 tokens, but one question's embedding 1.23 -> 1.30 ms (the consumer GEMMs finalise the statistics); the providers answer single questions, so the LayerNorm kernels stay the default
  if (const char* lf = getenv("VRAG_BERT_LN_FOLD")) e->ln_fold = atoi(lf) != 0;
  e->attn_w = Ha;
  e->q_scale = (1.0f / sqrtf((float)hd)) * 1.4426950408889634f;
  {
    vrag_encoder_config& c = e->cfg;
    c.vocab_size = V;
    c.hidden_size = H;
    c.num_layers = L;
    c.num_heads = cfg->num_heads;
    c.intermediate_size = I;
    c.global_every = 1;
    c.sliding_window = 0;
    c.rope_theta_global = c.rope_theta_local = 0.f;
    c.norm_eps = cfg->norm_eps;
    c.pad_token_id = cfg->pad_token_id;
    c.max_seq_len = cfg->max_seq_len;
    c.max_tokens = cfg->max_tokens;
    c.max_seqs = cfg->max_seqs;
    c.max_ranges = cfg->max_ranges;
    c.micro_batch_tokens = cfg->micro_batch_tokens;
    c.device = cfg->device;
  }
  auto fail = [&](int rc) {
    vrag_encoder_destroy(e);
    return rc;
  };
  int rc;
#define TRY(x)                 \
  do {                         \
    rc = (x);                  \
    if (rc) return fail(rc);   \
  } while (0)
  TRY(init_streams(e));
  const size_t stage_elems = std::max<size_t>({(size_t)3 * Ha * H, (size_t)I * H, (size_t)1 << 22});
  float* stage = nullptr;
  TRY(dev_alloc(e, &stage, stage_elems, false));
  std::vector<float> pw, pb, po;   // head-padded copies (head_dim 32 only)
  TRY(upload_f32(e, &e->tok_emb, w->word_embeddings, (size_t)V * H));
  TRY(upload_f32(e, &e->pos_emb, w->position_embeddings, (size_t)P * H));
  if (w->token_type_row) TRY(upload_f32(e, &e->type_row, w->token_type_row, H));
  TRY(upload_f32(e, &e->emb_norm, w->emb_norm_w, H));
  TRY(upload_f32(e, &e->emb_norm_b, w->emb_norm_b, H));
  e->blayers.resize(L);
  for (int l = 0; l < L; ++l) {
    BertLayer& ly = e->blayers[l];
    ARG_CHECK(w->wqkv[l] && w->bqkv[l] && w->wo[l] && w->bo[l] && w->attn_norm_w[l] && w->attn_norm_b[l] && w->w1[l] &&
                  w->b1[l] && w->w2[l] && w->b2[l] && w->out_norm_w[l] && w->out_norm_b[l],
              "null weight pointer in layer %d", l);
    const float *wqkv = w->wqkv[l], *bqkv = w->bqkv[l], *wo = w->wo[l];
    if (hd == 32) {
      pw.assign((size_t)3 * Ha * H, 0.f);
      pb.assign((size_t)3 * Ha, 0.f);
      po.assign((size_t)H * Ha, 0.f);
      for (int part = 0; part < 3; ++part)
        for (int hh = 0; hh < cfg->num_heads; ++hh)
          for (int d = 0; d < 32; ++d) {
            const size_t src = (size_t)part * H + hh * 32 + d, dst = (size_t)part * Ha + hh * 64 + d;
            memcpy(&pw[dst * H], wqkv + src * H, (size_t)H * sizeof(float));
            pb[dst] = bqkv[src];
          }
      for (int r = 0; r < H; ++r)
        for (int hh = 0; hh < cfg->num_heads; ++hh)
          memcpy(&po[(size_t)r * Ha + hh * 64], wo + (size_t)r * H + hh * 32, 32 * sizeof(float));
      wqkv = pw.data();
      bqkv = pb.data();
      wo = po.data();
    }
    TRY(upload_f32(e, &ly.ln1_w, w->attn_norm_w[l], H));
    TRY(upload_f32(e, &ly.ln1_b, w->attn_norm_b[l], H));
    // folded bias: b + W . ln_bias (the LayerNorm bias pushed through the linear layer), fp32 on the host
    auto folded_bias = [&](const float* W, const float* bvec, const float* ln_b, int rows, std::vector<float>& out) {
      out.resize(rows);
      for (int r = 0; r < rows; ++r) {
        double acc = 0.0;
        const float* wr = W + (size_t)r * H;
        for (int c2 = 0; c2 < H; ++c2) acc += (double)wr[c2] * (double)ln_b[c2];
        out[r] = bvec[r] + (float)acc;
      }
    };
    std::vector<float> cb;
    if (e->ln_fold && l > 0) {   // QKV consumes LN2 of the previous layer
      folded_bias(wqkv, bqkv, w->out_norm_b[l - 1], 3 * Ha, cb);
      TRY(upload_bf16(e, &ly.wqkv, wqkv, 3 * Ha, H, 3 * Ha, 0, stage, stage_elems, e->blayers[l - 1].ln2_w, &ly.s_qkv));
      TRY(upload_f32(e, &ly.bqkv, cb.data(), (size_t)3 * Ha));
    } else {
      TRY(upload_bf16(e, &ly.wqkv, wqkv, 3 * Ha, H, 3 * Ha, 0, stage, stage_elems));
      TRY(upload_f32(e, &ly.bqkv, bqkv, (size_t)3 * Ha));
    }
    TRY(upload_bf16(e, &ly.wo, wo, H, Ha, H, 0, stage, stage_elems));
    TRY(upload_f32(e, &ly.bo, w->bo[l], H));
    if (e->ln_fold) {            // W1 consumes LN1 of this layer
      folded_bias(w->w1[l], w->b1[l], w->attn_norm_b[l], I, cb);
      TRY(upload_bf16(e, &ly.w1, w->w1[l], I, H, I, 0, stage, stage_elems, ly.ln1_w, &ly.s_w1));
      TRY(upload_f32(e, &ly.b1, cb.data(), I));
    } else {
      TRY(upload_bf16(e, &ly.w1, w->w1[l], I, H, I, 0, stage, stage_elems));
      TRY(upload_f32(e, &ly.b1, w->b1[l], I));
    }
    TRY(upload_bf16(e, &ly.w2, w->w2[l], H, I, H, 0, stage, stage_elems));
    TRY(upload_f32(e, &ly.b2, w->b2[l], H));
    TRY(upload_f32(e, &ly.ln2_w, w->out_norm_w[l], H));
    TRY(upload_f32(e, &ly.ln2_b, w->out_norm_b[l], H));
  }
  {
    // identity rotation: the QKV epilogue's RoPE becomes a no-op
    std::vector<float> cs((size_t)cfg->max_seq_len * 32, 1.0f), sn((size_t)cfg->max_seq_len * 32, 0.0f);
    TRY(upload_f32(e, &e->cos_g, cs.data(), cs.size()));
    TRY(upload_f32(e, &e->sin_g, sn.data(), sn.size()));
  }
  TRY(init_workspace(e));
#undef TRY
  *out = e;
  return VRAG_OK;
}

void vrag_encoder_destroy(vrag_encoder* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& r : e->prof_pending) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto ev : e->prof_free) (void)hipEventDestroy(ev);
  for (auto& kv : e->graphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
  }
  for (void* p : e->dev_allocs) (void)hipFree(p);
  for (void* p : e->host_allocs) (void)hipHostFree(p);
  for (int i = 0; i < 4; ++i) {
    if (e->aux_streams[i]) (void)hipStreamDestroy(e->aux_streams[i]);
    if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
  }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

int vrag_encoder_set_qa_head(vrag_encoder* e, const float* w, const float* b, int32_t num_labels) {
  ARG_CHECK(e && w && b && num_labels > 0 && num_labels <= e->cfg.hidden_size, "bad qa head arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  int rc = upload_f32(e, &e->qa_w, w, (size_t)num_labels * e->cfg.hidden_size);
  if (rc) return rc;
  rc = upload_f32(e, &e->qa_b, b, num_labels);
  if (rc) return rc;
  e->qa_labels = num_labels;
  return VRAG_OK;
}

int vrag_encoder_set_token_head(vrag_encoder* e, const float* dense_w, const float* norm_w, const float* cls_w,
                                const float* cls_b, int32_t num_labels) {
  ARG_CHECK(e && dense_w && norm_w && cls_w && cls_b && num_labels > 0 && num_labels <= 64,
            "bad token head arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int H = e->cfg.hidden_size;
  float* stage = nullptr;
  int rc = dev_alloc(e, &stage, (size_t)H * H, false);
  if (rc) return rc;
  // dense weight as (value, remainder) operand pairs: the head GEMM then carries ~fp32 precision (see run_token_head)
  rc = upload_bf16(e, &e->tk_dense, dense_w, H, H, H, 0, stage, (size_t)H * H, nullptr, nullptr, &e->tk_dense_lo);
  if (rc) return rc;
  if ((rc = upload_f32(e, &e->tk_norm, norm_w, H))) return rc;
  if ((rc = upload_f32(e, &e->tk_w, cls_w, (size_t)num_labels * H))) return rc;
  if ((rc = upload_f32(e, &e->tk_b, cls_b, num_labels))) return rc;
  if ((rc = dev_alloc(e, &e->d_tok_logits, (size_t)e->cap_rows * num_labels))) return rc;
  e->tk_labels = num_labels;
  return VRAG_OK;
}

int vrag_encoder_set_mlm_head_ex(vrag_encoder* e, const float* dense_w, const float* dense_b, const float* norm_w,
                                 const float* norm_b, const float* decoder_w, const float* decoder_b) {
  ARG_CHECK(e && dense_w && norm_w, "bad mlm head arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int H = e->cfg.hidden_size, V = e->cfg.vocab_size;
  const int vpad = (int)align_up(V, 256);  // whole 256-wide GEMM tiles
  const size_t stage_elems = std::max<size_t>((size_t)H * H, (size_t)1 << 22);
  float* stage = nullptr;
  // setting the head again (another checkpoint's, or the other operand form): the previous decoder images go first
  HIP_TRY(hipDeviceSynchronize());
  dev_release(e, &e->mlm_dec);
  dev_release(e, &e->mlm_dec3);
  dev_release(e, &e->splade_a3);
  dev_release(e, &e->mlm_dense_lo);
  dev_release(e, &e->mlm_dense);
  dev_release(e, &e->mlm_bias);
  dev_release(e, &e->d_splade);
  int rc = dev_alloc(e, &stage, stage_elems, false);
  if (rc) return rc;
  if ((rc = upload_bf16(e, &e->mlm_dense, dense_w, H, H, H, 0, stage, stage_elems, nullptr, nullptr,
                        e->mlm_split ? &e->mlm_dense_lo : nullptr)))
    return rc;
  if ((rc = upload_f32(e, &e->mlm_norm, norm_w, H))) return rc;
  if (dense_b && (rc = upload_f32(e, &e->mlm_dense_b, dense_b, H))) return rc;
  if (norm_b && (rc = upload_f32(e, &e->mlm_norm_b, norm_b, H))) return rc;
  if (e->mlm_split) {
    // decoder as [Whi | Whi | Wlo] rows of 3H operand values; host weights stream through the staging buffer in row chunks
    if ((rc = dev_alloc(e, &e->mlm_dec3, (size_t)vpad * 3 * H, false))) return rc;
    if ((rc = dev_alloc(e, &e->splade_a3, (size_t)e->cap_rows * 3 * H, true))) return rc;
    if (decoder_w) {
      const int chunk = (int)std::max<size_t>(1, stage_elems / H);
      for (int r0 = 0; r0 < vpad; r0 += chunk) {
        const int nr_dst = std::min(chunk, vpad - r0), nr_src = std::max(0, std::min(chunk, V - r0));
        if (nr_src > 0) HIP_TRY(hipMemcpy(stage, decoder_w + (size_t)r0 * H, (size_t)nr_src * H * sizeof(float), hipMemcpyHostToDevice));
        launch_cvt_split3(e->op_dtype, 0, stage, e->mlm_dec3 + (size_t)r0 * 3 * H, nr_dst, nr_src, H);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
      }
    } else {   // tied decoder: the device-resident fp32 embedding table
      launch_cvt_split3(e->op_dtype, 0, e->tok_emb, e->mlm_dec3, vpad, V, H);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
    }
  } else if (decoder_w) {
    if ((rc = upload_bf16(e, &e->mlm_dec, decoder_w, V, H, vpad, 0, stage, stage_elems))) return rc;
  } else {
    // tied decoder: convert the device-resident fp32 embedding table
    if ((rc = dev_alloc(e, &e->mlm_dec, (size_t)vpad * H, false))) return rc;
    launch_cvt_rows(e->op_dtype, dim3(vpad), 0, e->tok_emb, e->mlm_dec, vpad, V, H, 0, (const float*)nullptr, (float*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  }
  if ((rc = dev_alloc(e, &e->mlm_bias, vpad, true))) return rc;
  if (decoder_b) HIP_TRY(hipMemcpy(e->mlm_bias, decoder_b, (size_t)V * sizeof(float), hipMemcpyHostToDevice));
  if ((rc = dev_alloc(e, &e->d_splade, (size_t)e->cfg.max_seqs * vpad))) return rc;
  e->vpad = vpad;
  HIP_TRY(hipDeviceSynchronize());
  dev_release(e, &stage);
  return VRAG_OK;
}

int vrag_encoder_set_head_precision(vrag_encoder* e, int32_t split_operands) {
  ARG_CHECK(e, "null handle");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  // A head that exists keeps its form until it is set again: vrag_encoder_set_mlm_head* rebuilds the decoder images for the
  // mode chosen here (and releases the other form's), so the setter is legal at any time.
  e->mlm_split = split_operands != 0;
  return VRAG_OK;
}

int vrag_encoder_set_mlm_head(vrag_encoder* e, const float* dense_w, const float* norm_w, const float* decoder_w,
                              const float* decoder_b) {
  return vrag_encoder_set_mlm_head_ex(e, dense_w, nullptr, norm_w, nullptr, decoder_w, decoder_b);
}

int vrag_encoder_set_token_types(vrag_encoder* e, const float* table, int32_t n_types) {
  ARG_CHECK(e && table && n_types > 0, "bad token type arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->arch == 1, "token types exist on BERT-family handles only");
  HIP_TRY(hipSetDevice(e->cfg.device));
  int rc = upload_f32(e, &e->type_table, table, (size_t)n_types * e->cfg.hidden_size);
  if (rc) return rc;
  if (!e->d_types) {
    if ((rc = dev_alloc(e, &e->d_types, e->cap_rows))) return rc;
    if ((rc = host_alloc(e, &e->h_types, e->cap_rows))) return rc;
  }
  e->n_types = n_types;
  return VRAG_OK;
}

int vrag_encoder_load_token_types(vrag_encoder* e, const int32_t* types, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(types, "null token types");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->type_table != nullptr, "token type table not set (vrag_encoder_set_token_types)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  HIP_TRY(hipStreamSynchronize(st));   // a previous upload may still read the pinned staging buffer
  memset(e->h_types, 0, (size_t)e->rows * sizeof(int));
  size_t src = 0;
  for (int s = 0; s < e->n_seqs; ++s) {
    for (int i = 0; i < e->seq_len[s]; ++i) {
      const int t = types[src + i];
      ARG_CHECK(t >= 0 && t < e->n_types, "token type %d outside the table (sequence %d)", t, s);
      e->h_types[e->seq_start[s] + i] = t;
    }
    src += e->seq_len[s];
  }
  HIP_TRY(hipMemcpyAsync(e->d_types, e->h_types, (size_t)e->rows * sizeof(int), hipMemcpyHostToDevice, st));
  e->types_loaded = true;
  e->ran = false;
  return VRAG_OK;
}

int vrag_encoder_set_pair_head(vrag_encoder* e, const float* pooler_w, const float* pooler_b, const float* cls_w,
                               const float* cls_b, int32_t num_labels) {
  ARG_CHECK(e && pooler_w && pooler_b && cls_w && cls_b && num_labels > 0 && num_labels <= 64, "bad pair head arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int H = e->cfg.hidden_size;
  int rc;
  if ((rc = upload_f32(e, &e->pr_wp, pooler_w, (size_t)H * H))) return rc;
  if ((rc = upload_f32(e, &e->pr_bp, pooler_b, H))) return rc;
  if ((rc = upload_f32(e, &e->pr_wc, cls_w, (size_t)num_labels * H))) return rc;
  if ((rc = upload_f32(e, &e->pr_bc, cls_b, num_labels))) return rc;
  if ((rc = dev_alloc(e, &e->d_pair_out, (size_t)e->cfg.max_seqs * num_labels))) return rc;
  if (!e->d_first_row) {
    if ((rc = dev_alloc(e, &e->d_first_row, e->cfg.max_seqs))) return rc;
    if ((rc = host_alloc(e, &e->h_first_row, e->cfg.max_seqs))) return rc;
  }
  e->pr_labels = num_labels;
  return VRAG_OK;
}

int vrag_encoder_run_pair_head(vrag_encoder* e, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->pr_labels > 0, "pair head not set (vrag_encoder_set_pair_head)");
  ARG_CHECK(e->ran, "encoder has not run on this batch");
  ARG_CHECK(e->final_norm == nullptr, "the pair head reads an already-normalised stream (BERT-family handles)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  HIP_TRY(hipStreamSynchronize(st));
  for (int s = 0; s < e->n_seqs; ++s) e->h_first_row[s] = e->seq_start[s];
  HIP_TRY(hipMemcpyAsync(e->d_first_row, e->h_first_row, (size_t)e->n_seqs * sizeof(int), hipMemcpyHostToDevice, st));
  ProfScope ps(e, VRAG_PROF_HEAD, st);
  HIP_TRY(launch_pooler_classifier(e->h, e->cfg.hidden_size, e->d_first_row, e->n_seqs, e->pr_wp, e->pr_bp, e->pr_wc,
                                   e->pr_bc, e->pr_labels, e->d_pair_out, st));
  return VRAG_OK;
}

int vrag_encoder_read_pair_logits(vrag_encoder* e, float* logits, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(logits, "null output");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->pr_labels > 0, "pair head not set");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  HIP_TRY(hipMemcpyAsync(logits, e->d_pair_out, (size_t)e->n_seqs * e->pr_labels * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return VRAG_OK;
}

int vrag_encoder_load_batch(vrag_encoder* e, const int32_t* ids, const int32_t* seq_lens, int32_t n_seqs,
                            void* stream) {
  ARG_CHECK(e && ids && seq_lens && n_seqs > 0, "bad batch arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  const auto& c = e->cfg;
  if (n_seqs > c.max_seqs) {
    set_error("batch has %d sequences, handle capacity is %d", n_seqs, c.max_seqs);
    return VRAG_ERR_CAPACITY;
  }
  HIP_TRY(hipSetDevice(c.device));
  hipStream_t st = pick_stream(e, stream);
  // a previous batch's async uploads read the pinned staging buffers
  HIP_TRY(hipStreamSynchronize(st));

  e->seq_start.assign(n_seqs, 0);
  e->seq_len.assign(seq_lens, seq_lens + n_seqs);
  e->mbs.clear();
  int64_t total = 0;
  for (int s = 0; s < n_seqs; ++s) {
    ARG_CHECK(seq_lens[s] > 0 && seq_lens[s] <= c.max_seq_len, "sequence %d has length %d (max_seq_len %d)", s,
              seq_lens[s], c.max_seq_len);
    total += seq_lens[s];
  }
  if (total > c.max_tokens) {
    set_error("batch has %lld tokens, handle capacity is %d", (long long)total, c.max_tokens);
    return VRAG_ERR_CAPACITY;
  }
  // Host side: O(sequences) geometry only -- first row of every sequence (multiples of kSeqAlign; a micro-batch starts on a
  // multiple of kRowPad), micro-batch cuts, q-block descriptors -- plus ONE pass over the ids that copies them into the
  // pinned staging and range-checks them.  The per-row image (ids in their rows, positions, sequence index, pad rows) is
  // laid out on the device by pack_layout_kernel.
  // Micro-batch cuts: as many micro-batches as the packed rows (tokens + alignment gaps) make at ~micro_batch_tokens each, of EQUAL
  // size -- never a full one followed by a sliver: a remainder of a few hundred rows is a latency-bound launch chain (2 ms for 22
  // layers) that runs alone at the end of the step, and a micro-batch a few rows past a whole number of GEMM tile rounds pays a
  // round for them.  (Round 3 cut greedily at micro_batch_tokens: a ragged 131 072-token batch became 65 5xx + 65 5xx + ~300.)
  int n_mb_target = 1;
  int64_t mb_rows_target = 0;
  if (c.micro_batch_tokens > 0) {
    int64_t packed = 0;
    for (int s = 0; s < n_seqs; ++s) packed = align_up(packed, kSeqAlign) + seq_lens[s];
    n_mb_target = (int)std::max<int64_t>(1, (packed + c.micro_batch_tokens / 2) / c.micro_batch_tokens);
    if (packed * 8 > (int64_t)n_mb_target * c.micro_batch_tokens * 9) ++n_mb_target;   // more than 1.125 x the nominal size each: one more
    mb_rows_target = (packed + n_mb_target - 1) / n_mb_target;
  }
  int t = 0, mb_row0 = 0, mb_blk0 = 0, nblk = 0, mb_lblk0 = 0, nlblk = 0, mb_tokens = 0, mb_seq0 = 0, mb_max_len = 0;
  size_t src = 0;
  const int prev_rows = e->rows;
  int *seq_row = e->h_seq_meta, *seq_src = e->h_seq_meta + c.max_seqs, *seq_ln = e->h_seq_meta + 2 * c.max_seqs;
  for (int s = 0; s < n_seqs; ++s) {
    const int Ls = seq_lens[s];
    t = (int)align_up(t, kSeqAlign);
    if (c.micro_batch_tokens > 0 && mb_tokens > 0 && (int)e->mbs.size() + 1 < n_mb_target && (t - mb_row0) + Ls > mb_rows_target) {
      const int row1 = (int)align_up(t, kRowPad);
      e->mbs.push_back({mb_row0, row1, mb_blk0, nblk, mb_lblk0, nlblk, mb_seq0, s, mb_max_len, mb_tokens});
      mb_seq0 = s;
      mb_max_len = 0;
      t = row1;
      mb_row0 = row1;
      mb_blk0 = nblk;
      mb_lblk0 = nlblk;
      mb_tokens = 0;
    }
    e->seq_start[s] = t;
    seq_row[s] = t;
    seq_src[s] = (int)src;
    seq_ln[s] = Ls;
    for (int q0 = 0; q0 < Ls; q0 += attention_q_block(false)) {
      if (nblk < e->cap_blocks) {
        e->h_blk_start[nblk] = t;
        e->h_blk_len[nblk] = Ls;
        e->h_blk_q0[nblk] = q0;
      }
      ++nblk;
    }
    for (int q0 = 0; q0 < Ls; q0 += attention_q_block(true)) {
      if (nlblk < e->cap_blocks) {
        e->h_lblk_start[nlblk] = t;
        e->h_lblk_len[nlblk] = Ls;
        e->h_lblk_q0[nlblk] = q0;
      }
      ++nlblk;
    }
    src += Ls;
    t += Ls;
    mb_tokens += Ls;
    mb_max_len = std::max(mb_max_len, Ls);
  }
  const int rows = (int)align_up(t, kRowPad);
  if (rows > e->cap_rows || nblk > e->cap_blocks || nlblk > e->cap_blocks) {
    set_error("internal: packed layout (%d rows, %d blocks) exceeds the workspace (%d rows, %d blocks)", rows, nblk,
              e->cap_rows, e->cap_blocks);
    return VRAG_ERR_CAPACITY;
  }
  {
    unsigned bad = 0;
    const unsigned V = (unsigned)c.vocab_size;
    for (int64_t i = 0; i < total; ++i) {   // one vectorisable pass: copy + range check
      const int id = ids[i];
      bad |= (unsigned)((unsigned)id >= V);
      e->h_ids[i] = id;
    }
    if (bad) {
      size_t o = 0;
      for (int s = 0; s < n_seqs; o += seq_lens[s], ++s)
        for (int i = 0; i < seq_lens[s]; ++i)
          ARG_CHECK(ids[o + i] >= 0 && ids[o + i] < c.vocab_size, "token id %d outside the vocabulary (sequence %d)", ids[o + i], s);
    }
  }
  // rows up to max(rows, previous rows) get pad ids so stale tokens of an older batch vanish
  const int fill_to = std::min(e->cap_rows, std::max(rows, prev_rows));
  e->mbs.push_back({mb_row0, rows, mb_blk0, nblk, mb_lblk0, nlblk, mb_seq0, n_seqs, mb_max_len, mb_tokens});
  // Work items of the fused QKV + attention kernel: per micro-batch that could take it (every sequence <= 512 tokens, rows above
  // the launch-bound threshold), groups of consecutive sequences packed onto the eight 64-token waves of a workgroup.
  int n_groups = 0;
  for (auto& mb : e->mbs) {
    mb.grp0 = mb.grp1 = n_groups;
    if (e->arch == 1 || !e->fused_qkv_attn || mb.max_len > kFusedMaxSeq) continue;
    if (e->fused_qkv_attn != 2 && gemm_consumer_finalizes(mb.row1 - mb.row0)) continue;
    n_groups += pack_groups_best_fit(seq_row, seq_ln, mb.seq0, mb.seq1, e->h_groups + (size_t)n_groups * 8);
    mb.grp1 = n_groups;
  }
  e->n_seqs = n_seqs;
  e->n_tokens = (int)total;
  e->rows = rows;
  e->n_blocks = nblk;
  e->n_ranges = 0;
  e->ran = false;
  e->types_loaded = false;   // segment ids belong to one batch
  HIP_TRY(hipMemcpyAsync(e->d_packed, e->h_ids, (size_t)total * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(e->d_seq_meta, e->h_seq_meta, (size_t)3 * c.max_seqs * sizeof(int), hipMemcpyHostToDevice, st));
  if (n_groups > 0) HIP_TRY(hipMemcpyAsync(e->d_groups, e->h_groups, (size_t)n_groups * 8 * sizeof(int4), hipMemcpyHostToDevice, st));
  // one upload for the six descriptor arrays: from the first used entry of the first to the last used entry of the last
  HIP_TRY(hipMemcpyAsync(e->d_blk_start, e->h_blk_start, ((size_t)5 * e->cap_blocks + nlblk) * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(pack_layout_kernel, dim3((fill_to + 255) / 256), dim3(256), 0, st, e->d_packed, e->d_seq_meta,
                     e->d_seq_meta + c.max_seqs, e->d_seq_meta + 2 * c.max_seqs, n_seqs, fill_to, c.pad_token_id, e->d_ids, e->d_pos,
                     e->d_tokseq);
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

// Eager or graph-replayed layer schedule (see vrag_encoder::graphs).  First two sights of a geometry: eager (the first also runs
// every first-use hipFuncSetAttribute of the instantiations involved, which must not happen inside a capture); third: capture,
// instantiate, launch; afterwards: one hipGraphLaunch.
constexpr size_t kGraphCacheMax = 48;      // entries of vrag_encoder::graphs (sighted geometries, instantiated or not)
constexpr int kGraphCaptureAfter = 2;      // eager sightings of a geometry before it is captured
static int run_layers_maybe_graphed(vrag_encoder* e, int n_layers, hipStream_t st) {
  auto eager = [&]() { return e->arch == 1 ? run_layers_bert_locked(e, n_layers, st) : run_layers_locked(e, n_layers, st); };
  const bool eligible = e->graph_rows_max > 0 && e->mbs.size() == 1 && e->rows <= e->graph_rows_max && !e->prof_on &&
                        st != nullptr && n_layers > 0;
  if (!eligible) return eager();
  const MicroBatch& mb = e->mbs[0];
  // The key holds everything that selects kernels or launch geometry: row count, attention block counts, sequences, and the
  // attention path (it depends on the longest sequence and the mean length, which the other fields do not determine).
  const std::array<int, 7> key = {e->rows, mb.blk1 - mb.blk0, mb.lblk1 - mb.lblk0, n_layers, e->types_loaded ? 1 : 0, mb.seq1 - mb.seq0,
                                  e->arch != 1 && fused_attention_for(e, mb) ? 1 + (mb.grp1 - mb.grp0) : 0};   // + the fused kernel's grid
  // Bound the whole map (instantiated or not): query traffic has many geometries, most of them seen once.  The least recently
  // used entry goes first, whatever it holds.
  while (e->graphs.size() >= kGraphCacheMax && e->graphs.find(key) == e->graphs.end()) {
    auto victim = e->graphs.begin();
    for (auto it = e->graphs.begin(); it != e->graphs.end(); ++it)
      if (it->second.last_use < victim->second.last_use) victim = it;
    if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
    if (victim->second.graph) (void)hipGraphDestroy(victim->second.graph);
    e->graphs.erase(victim);
  }
  auto& g = e->graphs[key];
  g.last_use = ++e->graph_clock;
  if (g.exec) {
    HIP_TRY(hipGraphLaunch(g.exec, st));
    ++e->graph_replays;
    e->ran = true;
    return VRAG_OK;
  }
  // capture + instantiate costs milliseconds: only a geometry that keeps coming back (third sight) is worth it
  if (g.seen++ < kGraphCaptureAfter) return eager();
  auto& slot = g;
  hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (ce != hipSuccess) return eager();                    // a stream that cannot be captured: stay eager
  const int rc = eager();
  hipGraph_t graph = nullptr;
  ce = hipStreamEndCapture(st, &graph);
  if (rc != VRAG_OK || ce != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    slot.seen = -1000000;                                   // never try this geometry again
    return rc != VRAG_OK ? rc : eager();
  }
  hipGraphExec_t exec = nullptr;
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    slot.seen = -1000000;
    return eager();
  }
  slot.graph = graph;
  slot.exec = exec;
  HIP_TRY(hipGraphLaunch(exec, st));
  ++e->graph_replays;
  e->ran = true;
  return VRAG_OK;
}

int vrag_encoder_run_layers(vrag_encoder* e, int32_t n_layers, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(n_layers >= 0 && n_layers <= e->cfg.num_layers, "n_layers out of range");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  return run_layers_maybe_graphed(e, n_layers, pick_stream(e, stream));
}

int vrag_encoder_graph_stats(vrag_encoder* e, int32_t enable, int64_t* replays, int32_t* cached) {
  ARG_CHECK(e, "null encoder handle");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  if (enable == 0) e->graph_rows_max = 0;
  else if (enable > 0) e->graph_rows_max = enable;
  if (replays) *replays = e->graph_replays;
  if (cached) {
    int n = 0;
    for (auto& kv : e->graphs) n += kv.second.exec != nullptr;
    *cached = n;
  }
  return VRAG_OK;
}

int vrag_encoder_run(vrag_encoder* e, void* stream) {
  if (!e) {
    set_error("null encoder handle");
    return VRAG_ERR_INVALID;
  }
  return vrag_encoder_run_layers(e, e->cfg.num_layers, stream);
}

int vrag_encoder_load_ranges(vrag_encoder* e, const int32_t* seq_idx, const int32_t* start, const int32_t* end,
                             int32_t n_ranges, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(seq_idx && start && end && n_ranges > 0, "bad range arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  if (n_ranges > e->cfg.max_ranges) {
    set_error("%d ranges, handle capacity is %d", n_ranges, e->cfg.max_ranges);
    return VRAG_ERR_CAPACITY;
  }
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < n_ranges; ++i) {
    const int s = seq_idx[i];
    ARG_CHECK(s >= 0 && s < e->n_seqs, "range %d: sequence index %d out of range", i, s);
    ARG_CHECK(start[i] >= 0 && end[i] >= start[i] && end[i] < e->seq_len[s],
              "range %d: [%d, %d] is not inside sequence %d of length %d", i, start[i], end[i], s, e->seq_len[s]);
    e->h_rng_start[i] = e->seq_start[s] + start[i];
    e->h_rng_end[i] = e->seq_start[s] + end[i];
  }
  e->n_ranges = n_ranges;
  HIP_TRY(hipMemcpyAsync(e->d_rng_start, e->h_rng_start, (size_t)n_ranges * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(e->d_rng_end, e->h_rng_end, (size_t)n_ranges * sizeof(int), hipMemcpyHostToDevice, st));
  return VRAG_OK;
}

int vrag_encoder_run_qa_head(vrag_encoder* e, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->qa_labels > 0, "qa head not set (vrag_encoder_set_qa_head)");
  ARG_CHECK(e->n_ranges > 0, "no ranges loaded (vrag_encoder_load_ranges)");
  ARG_CHECK(e->ran, "encoder has not run on this batch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  ProfScope ps(e, VRAG_PROF_HEAD, st);
  HIP_TRY(launch_range_pool(e->h, e->final_norm, e->cfg.norm_eps, e->cfg.hidden_size, e->d_rng_start, e->d_rng_end,
                            e->n_ranges, 0, e->qa_w, e->qa_b, e->qa_labels, e->d_rng_out, st));
  return VRAG_OK;
}

int vrag_encoder_read_qa_logits(vrag_encoder* e, float* logits, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(logits, "null output");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  HIP_TRY(hipMemcpyAsync(logits, e->d_rng_out, (size_t)e->n_ranges * e->qa_labels * sizeof(float),
                         hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return VRAG_OK;
}

int vrag_encoder_run_pool(vrag_encoder* e, int32_t normalize, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->n_ranges > 0, "no ranges loaded (vrag_encoder_load_ranges)");
  ARG_CHECK(e->ran, "encoder has not run on this batch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  ProfScope ps(e, VRAG_PROF_HEAD, st);
  HIP_TRY(launch_range_pool(e->h, e->final_norm, e->cfg.norm_eps, e->cfg.hidden_size, e->d_rng_start, e->d_rng_end,
                            e->n_ranges, normalize ? 1 : 2, nullptr, nullptr, 0, e->d_rng_out, st));
  return VRAG_OK;
}

int vrag_encoder_read_pool(vrag_encoder* e, float* out, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(out, "null output");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  HIP_TRY(hipMemcpyAsync(out, e->d_rng_out, (size_t)e->n_ranges * e->cfg.hidden_size * sizeof(float),
                         hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return VRAG_OK;
}

int vrag_encoder_run_token_head(vrag_encoder* e, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->tk_labels > 0, "token head not set (vrag_encoder_set_token_head)");
  ARG_CHECK(e->arch == 0, "the token-classification head is defined for the ModernBERT encoder only");
  ARG_CHECK(e->ran, "encoder has not run on this batch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  const int H = e->cfg.hidden_size;
  ProfScope ps(e, VRAG_PROF_HEAD, st);
  // Per-token logits see the un-averaged operand rounding of the head's dense layer (2e-2 with plain bf16 operands,
  // against north_star's 1e-3): the head runs with split operands -- x = x_hi + x_lo, W = W_hi + W_lo in the operand
  // type, three MFMA GEMMs accumulated in fp32 (hi.hi + lo.hi + hi.lo; lo.lo is below fp32 resolution) -- which costs
  // 3 x 2.T.H^2 FLOP, under 2 % of the encoder (tests/probes/precision_probe.py: 6.3e-3 -> 8.8e-4 on bf16 encoders).
  bf16_t* x_lo = e->o;   // the attention output buffer is free once the layers have run (row stride H for arch 0)
  HIP_TRY(launch_layernorm(e->h, e->final_norm, e->cfg.norm_eps, H, e->rows, e->a, nullptr, st, nullptr, nullptr, e->op_dtype, x_lo));
  HIP_TRY(hipMemsetAsync(e->f32tmp, 0, (size_t)e->rows * H * sizeof(float), st));
  const bf16_t* parts[3][2] = {{e->a, e->tk_dense}, {x_lo, e->tk_dense}, {e->a, e->tk_dense_lo}};
  for (auto& pr : parts) {
    GemmParams g{};
    g.op_dtype = e->op_dtype;
    g.A = pr[0];
    g.W = pr[1];
    g.M = e->rows;
    g.N = H;
    g.K = H;
    g.out_f32 = e->f32tmp;
    HIP_TRY(launch_gemm(EPI_RESIDUAL, g, st));
  }
  HIP_TRY(launch_ln_classifier(e->f32tmp, e->tk_norm, e->cfg.norm_eps, H, e->rows, e->tk_w, e->tk_b, e->tk_labels,
                               e->d_tok_logits, st, nullptr, /*gelu_first=*/1));
  return VRAG_OK;
}

int vrag_encoder_read_token_logits(vrag_encoder* e, float* logits, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(logits, "null output");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->tk_labels > 0, "token head not set");
  HIP_TRY(hipSetDevice(e->cfg.device));
  return read_rows(e, e->d_tok_logits, e->tk_labels, logits, pick_stream(e, stream));
}

int vrag_encoder_run_splade(vrag_encoder* e, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->mlm_dec != nullptr || e->mlm_dec3 != nullptr, "mlm head not set (vrag_encoder_set_mlm_head)");
  ARG_CHECK(e->ran, "encoder has not run on this batch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  const int H = e->cfg.hidden_size;
  ProfScope ps(e, VRAG_PROF_HEAD, st);
  const bool split = e->mlm_dec3 != nullptr;
  // Split operands (default): every SPLADE weight is a max over tokens of log1p(relu(logit)) -- nothing averages the
  // operand rounding of the two head GEMMs away (1e-2 on a weight with plain bf16 operands, and these rows feed a sparse
  // index whose top-k is held to bit-exactness).  x = xhi + xlo and W = Whi + Wlo in the operand type; the dense layer
  // runs as three accumulating GEMMs like the token head's, the decoder as ONE GEMM over K = 3H on the operand images
  // [xhi | xlo | xhi] x [Whi | Whi | Wlo] with the SPLADE epilogue on the sum (lo.lo is below fp32 resolution).
  bf16_t* x_lo = split ? e->o : nullptr;   // the attention output buffer is free once the layers have run
  if (e->arch == 1) {  // post-LN stream: no final LayerNorm, just the operand copy
    launch_cvt_rows(e->op_dtype, dim3(e->rows), st, e->h, e->a, e->rows, e->rows, H, 0, (const float*)nullptr, (float*)nullptr, x_lo);
    HIP_TRY(hipGetLastError());
  } else {
    HIP_TRY(launch_layernorm(e->h, e->final_norm, e->cfg.norm_eps, H, e->rows, e->a, nullptr, st, nullptr, nullptr, e->op_dtype, x_lo));
  }
  HIP_TRY(hipMemsetAsync(e->d_splade, 0, (size_t)e->n_seqs * e->vpad * sizeof(unsigned), st));
  GemmParams d{};
  d.op_dtype = e->op_dtype;
  d.M = e->rows;
  d.N = e->vpad;
  d.bias = e->mlm_bias;
  d.tok_seq = e->d_tokseq;
  d.splade_rows = e->d_splade;
  if (split) {
    HIP_TRY(hipMemsetAsync(e->f32tmp, 0, (size_t)e->rows * H * sizeof(float), st));
    const bf16_t* parts[3][2] = {{e->a, e->mlm_dense}, {x_lo, e->mlm_dense}, {e->a, e->mlm_dense_lo}};
    for (int i = 0; i < 3; ++i) {
      GemmParams g{};
      g.op_dtype = e->op_dtype;
      g.A = parts[i][0];
      g.W = parts[i][1];
      g.M = e->rows;
      g.N = H;
      g.K = H;
      g.out_f32 = e->f32tmp;
      g.bias = i == 0 ? e->mlm_dense_b : nullptr;   // BERT-family heads: the bias rides in with the first partial product
      HIP_TRY(launch_gemm(EPI_RESIDUAL, g, st));
    }
    HIP_TRY(launch_layernorm(e->f32tmp, e->mlm_norm, e->cfg.norm_eps, H, e->rows, e->splade_a3, nullptr, st, e->mlm_norm_b, nullptr,
                             e->op_dtype, nullptr, /*gelu_first=*/1, /*split3=*/1));
    d.A = e->splade_a3;
    d.W = e->mlm_dec3;
    d.K = 3 * H;
  } else {
    GemmParams g{};
    g.op_dtype = e->op_dtype;
    g.A = e->a;
    g.W = e->mlm_dense;
    g.M = e->rows;
    g.N = H;
    g.K = H;
    g.out_f32 = e->f32tmp;
    g.bias = e->mlm_dense_b;
    HIP_TRY(launch_gemm(EPI_F32_GELU, g, st));
    HIP_TRY(launch_layernorm(e->f32tmp, e->mlm_norm, e->cfg.norm_eps, H, e->rows, e->a, nullptr, st, e->mlm_norm_b, nullptr,
                             e->op_dtype));
    d.A = e->a;
    d.W = e->mlm_dec;
    d.K = H;
  }
  HIP_TRY(launch_gemm(EPI_SPLADE, d, st));
  return VRAG_OK;
}

int vrag_encoder_read_splade(vrag_encoder* e, float* rows, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(rows, "null output");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->mlm_dec != nullptr || e->mlm_dec3 != nullptr, "mlm head not set");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  const int V = e->cfg.vocab_size;
  HIP_TRY(hipMemcpy2DAsync(rows, (size_t)V * sizeof(float), e->d_splade, (size_t)e->vpad * sizeof(float),
                           (size_t)V * sizeof(float), e->n_seqs, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return VRAG_OK;
}

int vrag_encoder_read_splade_sparse(vrag_encoder* e, float threshold, int32_t cap_per_seq, int32_t* counts,
                                    int32_t* indices, float* values, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(counts && indices && values && cap_per_seq > 0 && threshold >= 0.f, "bad arguments");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->mlm_dec != nullptr || e->mlm_dec3 != nullptr, "mlm head not set");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  if (cap_per_seq > e->sp_cap) {   // (re)allocate; the old buffers stay owned by the handle until destroy
    if ((rc = dev_alloc(e, &e->d_sp_idx, (size_t)e->cfg.max_seqs * cap_per_seq, false))) return rc;
    if ((rc = dev_alloc(e, &e->d_sp_val, (size_t)e->cfg.max_seqs * cap_per_seq, false))) return rc;
    if (!e->d_sp_cnt && (rc = dev_alloc(e, &e->d_sp_cnt, e->cfg.max_seqs))) return rc;
    e->sp_cap = cap_per_seq;
  }
  const int n = e->n_seqs, cap = cap_per_seq;
  hipLaunchKernelGGL(splade_compact_kernel, dim3(n), dim3(256), 0, st, reinterpret_cast<const float*>(e->d_splade),
                     e->cfg.vocab_size, e->vpad, threshold, cap, e->d_sp_cnt, e->d_sp_idx, e->d_sp_val);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(counts, e->d_sp_cnt, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int worst = 0;
  for (int i = 0; i < n; ++i) worst = std::max(worst, counts[i]);
  if (worst > cap) {
    set_error("a SPLADE row has %d non-zero weights, capacity per sequence is %d", worst, cap);
    return VRAG_ERR_CAPACITY;
  }
  // only the used prefix of every row travels: rows are cap apart on the device and in the caller's buffers
  const size_t width = (size_t)std::max(worst, 1);
  HIP_TRY(hipMemcpy2DAsync(indices, (size_t)cap * sizeof(int), e->d_sp_idx, (size_t)cap * sizeof(int), width * sizeof(int), n,
                           hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpy2DAsync(values, (size_t)cap * sizeof(float), e->d_sp_val, (size_t)cap * sizeof(float),
                           width * sizeof(float), n, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return VRAG_OK;
}

int vrag_encoder_read_hidden(vrag_encoder* e, int32_t apply_final_norm, float* out, void* stream) {
  int rc = check_ready(e);
  if (rc) return rc;
  ARG_CHECK(out, "null output");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  ARG_CHECK(e->ran, "encoder has not run on this batch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t st = pick_stream(e, stream);
  const int H = e->cfg.hidden_size;
  const float* src = e->h;
  if (apply_final_norm && e->final_norm) {
    HIP_TRY(launch_layernorm(e->h, e->final_norm, e->cfg.norm_eps, H, e->rows, nullptr, e->f32tmp, st));
    src = e->f32tmp;
  }
  return read_rows(e, src, H, out, st);
}

int vrag_encoder_extract_qa(vrag_encoder* e, const int32_t* ids, const int32_t* seq_lens, int32_t n_seqs,
                            const int32_t* rng_seq, const int32_t* rng_start, const int32_t* rng_end,
                            int32_t n_ranges, float* logits) {
  ARG_CHECK(e != nullptr, "null encoder handle");
  std::lock_guard<std::recursive_mutex> lk(e->mu);  // the whole sequence is one critical section
  int rc;
  if ((rc = vrag_encoder_load_batch(e, ids, seq_lens, n_seqs, nullptr))) return rc;
  if ((rc = vrag_encoder_load_ranges(e, rng_seq, rng_start, rng_end, n_ranges, nullptr))) return rc;
  if ((rc = vrag_encoder_run(e, nullptr))) return rc;
  if ((rc = vrag_encoder_run_qa_head(e, nullptr))) return rc;
  return vrag_encoder_read_qa_logits(e, logits, nullptr);
}

int vrag_set_small_batch_rows(int32_t rows) { return gemm_small_m_threshold(rows < 0 ? -1 : rows); }

int vrag_encoder_f16_saturated(vrag_encoder* e, int32_t reset, int32_t* saturated) {
  ARG_CHECK(e && saturated, "null argument");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  // One launch + one stream wait instead of five synchronous symbol copies (each ~12 us: 70 us of a single question's 1.35 ms
  // embedding call sat here, tools/probes/embed_latency_probe.py).  The flags are per device, not per engine, as before.
  if (!e->sat_host) {
    void* h = nullptr;
    if (hipHostMalloc(&h, sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
      void* d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
        e->host_allocs.push_back(h);
        e->sat_host = reinterpret_cast<unsigned*>(h);
        e->sat_host_dev = reinterpret_cast<unsigned*>(d);
        e->sat_addr[0] = f16_sat_flag_address();   // conversions in this file (weight packing)
        e->sat_addr[1] = gemm_f16_flag_address();
        e->sat_addr[2] = attention_f16_flag_address();
        e->sat_addr[3] = qkv_attn_f16_flag_address();
        e->sat_addr[4] = norm_heads_f16_flag_address();
      } else {
        (void)hipHostFree(h);
      }
    }
    (void)hipGetLastError();
  }
  if (e->sat_host && e->sat_addr[0] && e->sat_addr[1] && e->sat_addr[2] && e->sat_addr[3] && e->sat_addr[4]) {
    SatFlags f;
    for (int i = 0; i < 5; ++i) f.a[i] = e->sat_addr[i];
    hipLaunchKernelGGL(f16_sat_gather_kernel, dim3(1), dim3(1), 0, e->own_stream, f, reset != 0 ? 1 : 0, e->sat_host_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->own_stream));
    *saturated = *reinterpret_cast<volatile unsigned*>(e->sat_host) ? 1 : 0;
    return VRAG_OK;
  }
  unsigned any = f16_sat_take(reset != 0);                 // (no mapped word or no symbol address: the per-file copies)
  any |= gemm_f16_saturated(reset != 0);
  any |= attention_f16_saturated(reset != 0);
  any |= qkv_attn_f16_saturated(reset != 0);
  any |= norm_heads_f16_saturated(reset != 0);
  *saturated = any ? 1 : 0;
  return VRAG_OK;
}

// Diagnostics / unit tests of the attention kernels alone: n_seqs sequences of S tokens, host operands in the kernels' own
// layouts (q, k: [T, H] operand-type bits; vt: [H, Tp] with Tp = T rounded up to 256), output o [T, H] operand-type bits.
int vrag_encoder_set_concurrency(vrag_encoder* e, int32_t n_streams) {
  ARG_CHECK(e && n_streams >= 1 && n_streams <= 4, "n_streams must be in [1, 4]");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  e->n_streams = n_streams;
  return VRAG_OK;
}

int vrag_encoder_set_profiling(vrag_encoder* e, int32_t enabled) {
  ARG_CHECK(e, "null encoder handle");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  e->prof_on = enabled != 0;
  e->prof_period = enabled > 1 ? enabled : 1;
  return VRAG_OK;
}

int vrag_encoder_read_profile(vrag_encoder* e, float* ms, int64_t* launches, int32_t reset) {
  ARG_CHECK(e && ms && launches, "null argument");
  std::lock_guard<std::recursive_mutex> lk(e->mu);
  HIP_TRY(hipSetDevice(e->cfg.device));
  for (auto& r : e->prof_pending) {
    HIP_TRY(hipEventSynchronize(r.b));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    e->prof_ms[r.cls] += t;
    e->prof_n[r.cls] += 1;
    e->prof_free.push_back(r.a);
    e->prof_free.push_back(r.b);
  }
  e->prof_pending.clear();
  for (int i = 0; i < VRAG_PROF_COUNT; ++i) {
    // sampled classes: the timed launches' mean duration times the launches issued (ms / launches stays the measured mean)
    ms[i] = e->prof_n[i] > 0 ? e->prof_ms[i] * (float)((double)e->prof_seen[i] / (double)e->prof_n[i]) : 0.f;
    launches[i] = e->prof_seen[i];
    if (reset) {
      e->prof_ms[i] = 0.f;
      e->prof_n[i] = 0;
      e->prof_seen[i] = 0;
    }
  }
  return VRAG_OK;
}

}  // extern "C"
