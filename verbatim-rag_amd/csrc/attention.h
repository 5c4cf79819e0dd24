// Packed (padding-free) bidirectional attention, global or banded |i-j| <= window (gfx950).
#pragma once
#include "common.h"

namespace vrag {

struct AttnParams {
  const bf16_t* q;   // [Tp, H]  RoPE'd and pre-scaled by head_dim^-1/2 * log2(e)
  const bf16_t* k;   // [Tp, H]  RoPE'd
  const bf16_t* vt;  // [H, Tp]  V transposed (row = head*64+d, col = token)
  bf16_t* o;         // [Tp, H]
  const int* blk_seq_start;  // [n_blocks] first packed token of the q-block's sequence
  const int* blk_seq_len;    // [n_blocks] sequence length S
  const int* blk_q0;         // [n_blocks] first query row of the block inside its sequence (x attention_q_block(local))
  int n_blocks;
  int H;       // hidden = nh * 64
  int nh;
  int Tp;      // padded token rows (leading dimension of vt)
  int window;  // banded layers: keep |i-j| <= window; ignored for global layers
  int op_dtype;  // kOpBf16 / kOpF16: what q, k, vt and o hold
};

hipError_t launch_attention(const AttnParams& p, bool local, hipStream_t stream);
int attention_q_block(bool local);  // query rows per work item (256 global / 128 banded)

// 1 if an fp32 -> fp16 operand conversion in this file's kernels clamped since the last reset (common.h).
unsigned attention_f16_saturated(bool reset);
unsigned* attention_f16_flag_address();   // device address of this file's flag on the current device (common.h)

}  // namespace vrag
