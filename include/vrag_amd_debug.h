/* vrag_amd_debug.h -- tuning / unit-test harness of the gfx950 kernels.  NOT part of the product ABI (include/vrag_amd.h):
 * these entry points exist only in libvrag_amd_dbg.so, the harness build of the same sources (verbatim-rag_amd/build.py,
 * -DVRAG_DEBUG_API: it also keeps the phase-decomposition branches of the fused kernel that the product build compiles out).
 * tools/ and tests/test_attention_unit_gpu.py load it beside the product library. */
#ifndef VRAG_AMD_DEBUG_H
#define VRAG_AMD_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Average ms of one GEMM instantiation (epilogue id as in csrc/gemm_bf16.h, 7 = no epilogue) on synthetic [-1,1) operands. */
int vrag_debug_gemm_ms(int32_t epi, int32_t M, int32_t N, int32_t K, int32_t iters, int32_t device, float* ms_out);
/* Same for one attention launch: n_seqs sequences of S tokens (S a multiple of 8), hidden H = 64 * heads, local != 0 =
 * the banded kernel with |i - j| <= window. */
int vrag_debug_attn_ms(int32_t local, int32_t n_seqs, int32_t S, int32_t H, int32_t window, int32_t iters, int32_t device,
                       float* ms_out);
/* Same for the fused Wqkv + RoPE + attention kernel (csrc/qkv_attn.hip; S <= 512).  flags: 1 = no attention phase, 2 = no
 * main-loop MFMAs, 4 = no operand DMA (phase decomposition of the kernel's time). */
int vrag_debug_qkv_attn_ms(int32_t local, int32_t n_seqs, int32_t S, int32_t H, int32_t window, int32_t iters, int32_t flags,
                           int32_t device, float* ms_out);
/* Unit-test hook of the attention kernels alone: host operands in the kernels' layouts (q, k: [T, H] bf16 / fp16 bits, q
 * pre-scaled by head_dim^-1/2 * log2 e; vt: [H, Tp], Tp = T rounded up to 256), o [T, H] out; T = n_seqs * S. */
int vrag_debug_attn_run(int32_t local, int32_t n_seqs, int32_t S, int32_t H, int32_t window, int32_t f16, const uint16_t* q,
                        const uint16_t* k, const uint16_t* vt, uint16_t* o, int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* VRAG_AMD_DEBUG_H */
