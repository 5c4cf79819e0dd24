// Wqkv GEMM + RoPE + attention in ONE kernel per (sequence, head), for sequences of at most 512 tokens (gfx950).
#pragma once
#include "common.h"

namespace vrag {

constexpr int kFusedMaxSeq = 512;   // tokens one workgroup holds: 8 waves x 64 rows
constexpr int kFusedMinFillPct = 60;   // tokens per 512-token workgroup (%) from which the kernel beats the two-kernel path (a workgroup costs what a full one costs; measured: profiles/r04_fused_by_sequence_length.txt -- it still wins at 62.5 %, break-even ~58 %)

struct QkvAttnParams {
  const bf16_t* x;         // [Tp, H] the Wqkv GEMM's A operand rows (op16(h - c) under the LayerNorm fold, else LN(h))
  const bf16_t* w;         // [nh][192][H] per-head weight rows: q(64) k(64) v(64) (permute_qkv_heads), LayerNorm gain folded in
  const float* ln_mu;      // [Tp] or null (no fold): see GemmParams
  const float* ln_rstd;    // [Tp]
  const float* ln_s;       // [nh * 192 (+ 64 readable floats behind the last head)] row sums of w, same permutation
  const float* rope_cos;   // [rope_rows, 32]
  const float* rope_sin;
  int rope_rows;           // positions the rotary tables hold (>= the longest sequence)
  bf16_t* o;               // [Tp, H] attention output (the Wo GEMM's A operand)
  // Work items: GROUPS of consecutive sequences, each sequence on ceil(S / 64) consecutive waves of the group's workgroup
  // (fused_pack_groups below).  Per wave: x = packed row of the wave's first token, y = its sequence's length (0 = unused wave),
  // z = the first wave of its sequence, w unused.
  const int4* groups;      // [n_groups][8]
  int n_groups;
  int H, nh, Tp;
  int window;              // banded layers: keep |i - j| <= window
  int op_dtype;
  float q_scale;           // head_dim^-1/2 * log2(e): the softmax runs in exp2 units
  int debug_flags;         // tuning probe (vrag_debug_qkv_attn_ms): 1 = no attention phase, 2 = no main-loop MFMAs, 4 = no operand DMA
};

hipError_t launch_qkv_attention(const QkvAttnParams& p, bool local, hipStream_t stream);

// Host: packs sequences seq0 .. seq1 - 1 (first rows `seq_row`, lengths `seq_len` <= kFusedMaxSeq) into groups, first fit over
// consecutive sequences -- the simple form the diagnostics use; the engine's packer (capi.hip) is best fit decreasing over the
// whole micro-batch.  Writes 8 descriptors per group to `out` (room for 8 * (seq1 - seq0) of them), returns the number of groups.
int fused_pack_groups(const int* seq_row, const int* seq_len, int seq0, int seq1, int4* out);

// out[(h * 3 + part) * 64 + d][:] = w[part * H + h * 64 + d][:]  (and the same for the optional row-sum / bias vectors)
hipError_t permute_qkv_heads(const bf16_t* w, const float* s, int H, int nh, bf16_t* w_out, float* s_out, hipStream_t stream);

unsigned qkv_attn_f16_saturated(bool reset);
unsigned* qkv_attn_f16_flag_address();   // device address of this file's flag on the current device (common.h)

}  // namespace vrag
