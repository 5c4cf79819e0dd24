// bf16 MFMA GEMM  C[M,N] = A[M,K] . W[N,K]^T  with fused epilogues (gfx950).
// A = token-major activations (K contiguous), W = HF nn.Linear weight [out,in].
#pragma once
#include "common.h"

namespace vrag {

enum GemmEpi {
  EPI_F32 = 0,       // out_f32[m][n] = acc (+bias)
  EPI_BF16 = 1,      // out_bf16[m][n] = bf16(acc (+bias))
  EPI_F32_GELU = 2,  // out_f32[m][n] = gelu_erf(acc)                   (prediction-head dense)
  EPI_RESIDUAL = 3,  // out_f32[m][n] += acc                             (fp32 residual stream)
  EPI_GEGLU = 4,     // out_bf16[m][f] = bf16(gelu_erf(x1[f]) * x2[f])   (Wi rows pre-interleaved)
  EPI_QKV_ROPE = 5,  // RoPE(q,k) in fp32, q *= d^-1/2, write Q,K [T,nh,64] and V^T [nh,64,T]
  EPI_SPLADE = 6,    // rows[seq(m)][n] = max(rows, log1p(relu(acc+bias)))  via ordered-uint atomicMax
  EPI_NONE = 7,      // diagnostics: main loop only (accumulators kept alive, nothing stored)
  EPI_TOPK = 8,      // batched dense search: A = corpus rows, W = queries; scores above the query's entry threshold become candidate keys
  EPI_COUNT
};

struct GemmParams {
  const bf16_t* A;     // [Mpad, K]
  const bf16_t* W;     // [N, K]
  int M, N, K;         // M valid rows; Mpad = roundup(M, kRowPad) rows are readable
  float* out_f32;      // EPI_F32 / EPI_F32_GELU / EPI_RESIDUAL  [M, N]
  bf16_t* out_bf16;    // EPI_BF16 [M, N]; EPI_GEGLU [M, N/2]
  const float* bias;   // optional [N] (EPI_F32, EPI_F32_GELU, EPI_BF16, EPI_RESIDUAL, EPI_QKV_ROPE, EPI_SPLADE)
  // EPI_QKV_ROPE
  bf16_t* q;           // [Mpad, hidden]
  bf16_t* k;           // [Mpad, hidden]
  bf16_t* vt;          // [hidden, vt_ld]   (row = head*64+d, col = token)
  const float* rope_cos;  // [max_pos, 32]
  const float* rope_sin;  // [max_pos, 32]
  const int* pos;      // [Mpad] position of each packed token inside its sequence
  int hidden;          // H (= N/3)
  int vt_ld;           // Mpad
  float q_scale;       // head_dim^-0.5 (exact power of two for d=64)
  // EPI_SPLADE
  const int* tok_seq;  // [Mpad] sequence index of each token, -1 for padding tokens
  unsigned* splade_rows;  // [n_seqs, N] float bits (values >= 0 so uint order == float order)
  // LayerNorm folded into the GEMM (EPI_QKV_ROPE, EPI_GEGLU): A = bf16(h - c) (un-normalised residual minus a per-row
  // shift c), W' = W * ln_weight (per input column), ln_s[n] = sum_k W'[n][k]; the epilogue applies
  //   out[m][n] = ln_rstd[m] * (acc[m][n] - ln_mu[m] * ln_s[n])  ==  (LayerNorm(h) . W^T)[m][n],   ln_mu = mean(h) - c.
  const float* ln_mu;     // [Mpad] row means of (h - c) (null = no fold)
  const float* ln_rstd;   // [Mpad] 1/sqrt(var + eps)
  const float* ln_s;      // [N]
  // EPI_RESIDUAL extras: bf16 copy of the updated residual rows (the next GEMM's A operand) and the
  // per-row partial sums (sum x, sum x^2) over each 64-column segment, for the next fold.
  // EPI_RESIDUAL, post-LN encoders: the residual input is LN(out_f32 row) = (t - res_mu) * res_rstd * res_g + res_b
  const float* res_mu;    // [Mpad] or null (= plain residual add)
  const float* res_rstd;  // [Mpad]
  const float* res_g;     // [N] LayerNorm gain
  const float* res_b;     // [N] LayerNorm bias
  const float* ln_shift;  // [Mpad] or null: per-row shift c subtracted before the bf16 copy / the statistics (see the epilogue)
  bf16_t* resid_bf16;     // [Mpad, N] or null  = bf16(updated residual - c)
  float* stats_part;      // [N/64][stats_ld][2] or null: slice-major, so that the rows a store instruction covers are contiguous
  int stats_ld;           // rows per slice of stats_part / stats_in (the allocation's row count)
  // EPI_RESIDUAL, split residual stream (round 4; byte remainders since round 6): between two sub-layers the stream is kept as the
  // operand plane plus ONE byte per element, h = c + float(hi) * (1 + (byte - 128) * step): hi = op16(h - c) is `resid_bf16` -- the
  // very operand copy the next GEMM reads -- and the byte carries the next 8 bits of (h - c) / hi (16 significant bits together
  // for bf16 operands, 19 for fp16; the fp32 rows are not written at all).  6 instead of 10 bytes per element and sub-layer.
  // lo_in: the stream arrives split (hi is read from resid_bf16, relative to ln_shift_prev); lo_out: it leaves split (relative
  // to ln_shift).  Both null = the fp32 rows of out_f32 on both sides.  The byte plane's layout is the epilogue's own (64 x 64
  // blocks of 4 KiB, gemm_bf16.hip lo8_offset); a [Mpad, N]-byte allocation holds it, Mpad % 64 == 0, and the plane of a row range
  // that starts at a multiple of 64 rows starts at row0 * N bytes.
  const unsigned char* lo_in;     // [Mpad * N] bytes or null
  unsigned char* lo_out;          // [Mpad * N] bytes or null (may alias lo_in: every byte is read and written by the same lane)
  float* ln_shift_prev;   // [Mpad]: the shift the arriving planes are relative to (written by whoever advances ln_shift)
  // Consumer-side finalisation of the row statistics, small-row configuration only (gemm_consumer_finalizes()): the LayerNorm-folding GEMM
  // (EPI_QKV_ROPE / EPI_GEGLU / EPI_BF16) finishes the producer's partial statistics itself -- every wave for its own 64 rows,
  // before its epilogue reads them -- and the wave of column 0 advances ln_shift.  One launch fewer per sub-layer where a
  // launch costs as much as the kernel (a query's handful of chunks).
  const float* stats_in;  // [K/64][stats_ld][2] or null (then ln_mu / ln_rstd were written by ln_stats_finalize_kernel)
  float fin_eps;          // LayerNorm eps of that finalisation
  // EPI_TOPK (csrc/topk.hip, the tiled batched dense search): output column n is query n (topk_pairs: queries ride as
  // (bf16 value, bf16 remainder) column pairs 2q, 2q + 1 and the score is the pair's sum); row m is corpus row topk_row_base + m.
  // A score enters query q's candidate list when its key (score, row) is above topk_thr_key[q]; topk_thr_score[q] is that
  // key's score (-inf while the list is short), the cheap pre-filter every accumulator is compared with.
  // topk_direct: first stage, every row is a candidate -- slot = m, no counter traffic.
  const float* topk_thr_score;      // [n queries]
  const unsigned long long* topk_thr_key;
  unsigned* topk_cnt;               // [n queries] candidates appended so far (may exceed topk_cap: the overflow is detected by the selection kernel)
  unsigned long long* topk_buf;     // [n queries][topk_cap]
  int topk_cap, topk_nq, topk_pairs, topk_direct, topk_tile;   // topk_tile: 1 = 256 x 128 tiles on a three-stage ring (<= 128 query columns), 2 = 256 x 64 tiles on a four-stage ring (<= 64)
  unsigned topk_row_base;
  int topk_tile_skip, topk_tile0;   // EPI_TOPK, appending stages behind a sampled first stage: the launch walks the corpus tiles the sample did NOT take --
                                    // launch tile t is tile d = topk_tile0 + t of that sequence = corpus tile d + d / (skip - 1) + 1 (d < 256 (skip - 1)), d + 256 beyond;
                                    // A = the shard's first row, keys carry the corpus row (0 / 1 = contiguous rows from A)
  int topk_tile_stride;   // EPI_TOPK, first (one-key-per-row) stage only: row tile t of the launch reads corpus rows [t * stride * 256, + 256) -- a SAMPLE of
                          // the shard's 256-row tiles instead of its first rows (0 / 1 = contiguous; the 256-row tile forms only); keys carry the corpus row
  int op_dtype;           // kOpBf16 (0) or kOpF16 (1): what A, W and every 16-bit output hold (pointers stay typed bf16_t*)
  int n_tiles;            // filled by the launcher: output tiles walked by the persistent grid
  int act_gelu;           // EPI_BF16: apply GELU(erf) after the bias
};

// Launches on `stream`. Requirements: N % 128 == 0, K % 64 == 0.
hipError_t launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream);
const char* gemm_kernel_name(GemmEpi epi);
// GEMMs with M <= threshold rows use the small-batch configuration (128x128 tiles, four LDS stages).
// set_to >= 0 changes the threshold (0 disables the configuration); returns the current value.
int gemm_small_m_threshold(int set_to);

// True when a LayerNorm-folding GEMM over `rows` token rows takes the small-row configuration and therefore finishes the row
// statistics itself when given `stats_in` (the caller then skips ln_stats_finalize_kernel).
bool gemm_consumer_finalizes(int rows);

// 1 if an fp32 -> fp16 operand conversion in this file's kernels clamped since the last reset (common.h).
unsigned gemm_f16_saturated(bool reset);
unsigned* gemm_f16_flag_address();   // device address of this file's flag on the current device (common.h)

}  // namespace vrag
