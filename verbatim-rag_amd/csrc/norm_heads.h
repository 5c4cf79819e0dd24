// HBM-bound row kernels: embedding gather + LayerNorm, LayerNorm, and the fused
// final-LayerNorm + range-mean heads (sentence classifier / dense pooling), gfx950.
#pragma once
#include "common.h"

namespace vrag {

// h[t] = LN(E[ids[t]]) (fp32 residual stream) and a = bf16(h) (layer 0 has no attn_norm).
// BERT family: P/pos/type_row add learned position rows and the token-type row before the LN, `bias` is
// the LayerNorm bias (TF:models/bert/modeling_bert.py:53-62, models/distilbert/modeling_distilbert.py:82-116).
hipError_t launch_embed_ln(const int* ids, const float* E, const float* w, float eps, int H, int rows,
                           float* h, bf16_t* a, hipStream_t stream, const float* P = nullptr,
                           const int* pos = nullptr, const float* type_row = nullptr, const float* bias = nullptr,
                           const int* type_ids = nullptr,    // type_ids: per-token row of the table at type_row
                           int op_dtype = kOpBf16);          // what `a` holds (kOpBf16 / kOpF16)

// out = LN(h) * w (+ bias); writes bf16 and/or fp32 (either pointer may be null; out_f32 may be h itself).
// w == nullptr: no gain (it is folded into the consumer GEMM's weight); row_mean != nullptr: also mean(h[row]).
// op_dtype: what out_bf16 holds (kOpBf16 / kOpF16); out_lo != nullptr: also the remainder  x - float(out_bf16)  in the
// same type (the split-operand head GEMM multiplies both parts).  gelu_first: h is a raw dense output, gelu_erf is applied
// before the normalisation.  split3: out_bf16 is a [rows, 3H] image [hi | lo | hi] (out_lo must be null): the A operand of a
// K = 3H split-operand GEMM.
hipError_t launch_layernorm(const float* h, const float* w, float eps, int H, int rows,
                            bf16_t* out_bf16, float* out_f32, hipStream_t stream, const float* bias = nullptr,
                            float* row_mean = nullptr, int op_dtype = kOpBf16, bf16_t* out_lo = nullptr, int gelu_first = 0,
                            int split3 = 0);

// For each range r: v = mean_{t in [start[r], end[r]]} LN(h[t]) * lnw   (inclusive token range; lnw == nullptr:
// no LayerNorm, v = mean of h -- post-LN encoders)
//   mode 0: out[r][c] = v . Wc[c] + bc[c]          (reference QAModel head, model.py:82-113)
//   mode 1: out[r][:] = v / max(||v||, 1e-12)       (sentence-transformers mean/CLS pooling + Normalize)
//   mode 2: out[r][:] = v                           (pooling without normalisation)
hipError_t launch_range_pool(const float* h, const float* lnw, float eps, int H, const int* start,
                             const int* end, int n_ranges, int mode, const float* Wc, const float* bc,
                             int num_labels, float* out, hipStream_t stream);

// Token-classification tail: logits[t][c] = LN(x[t]) * lnw . Wc[c] + bc[c]   (x fp32 = gelu(dense(h));
// gelu_first: x holds the raw dense output and the kernel applies gelu_erf itself)
hipError_t launch_ln_classifier(const float* x, const float* lnw, float eps, int H, int rows,
                                const float* Wc, const float* bc, int num_labels, float* logits,
                                hipStream_t stream, const float* lnb = nullptr, int gelu_first = 0);

// logits[s][c] = Wc[c] . tanh(Wp . h[first[s]] + bp) + bc[c]   (BertPooler + classifier; one workgroup per sequence)
hipError_t launch_pooler_classifier(const float* h, int H, const int* first_row, int n_seqs, const float* Wp,
                                    const float* bp, const float* Wc, const float* bc, int num_labels, float* logits,
                                    hipStream_t stream);

// 1 if an fp32 -> fp16 operand conversion in this file's kernels clamped since the last reset (common.h).
unsigned norm_heads_f16_saturated(bool reset);
unsigned* norm_heads_f16_flag_address();   // device address of this file's flag on the current device (common.h)

}  // namespace vrag
