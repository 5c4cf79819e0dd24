// Flash-style packed attention for gfx950 (head_dim 64), replacing the reference's
// eager/SDPA attention inside ModernBertAttention.forward
// (transformers modeling_modernbert.py:166-185,286-299; mask: masking_utils.py:141-151).
//
// Work item = (sequence, 128-row query block, head); 4 waves x 32 query rows.
// Per 64-key step:
//   S^T[key][q] = K . Q^T        v_mfma_f32_32x32x16_bf16, A = K tile (LDS), B = Q (registers)
//     -> lane (q = lane&31) holds 16 keys of ITS query row per 32-key tile, so the softmax
//        row max / sum are in-lane + one cross-half exchange (lane ^ 32).
//   O^T[d][q] += V^T . P^T       A = V^T tile (LDS, key-contiguous), B = P straight from the
//        S^T accumulators (the accumulator row map IS the B-operand k-slot order once the
//        V^T fragment is read with the same key permutation) -> no LDS round trip for P and
//        the per-row rescale factor lives in the lane that owns the O^T column.
// K tile rows are 128 B, 16-byte chunks XOR-swizzled by ((key>>1)&7) (ds_read_b128 conflict free);
// V^T tile rows are 128 B, 8-byte chunks XOR-swizzled by ((d>>1)&15) (ds_read_b64 conflict free).
// Softmax is fp32 (exp2 domain); P and the MFMA operands are bf16, accumulation fp32.
#include "attention.h"

namespace vrag {

template <bool LOCAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[16384];
  char* sK = smem;
  char* sV = smem + 8192;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int blk = blockIdx.x, head = blockIdx.y;
  const int H = p.H, Tp = p.Tp;

  const int t0 = p.blk_seq_start[blk];
  const int S = p.blk_seq_len[blk];
  const int qb0 = p.blk_q0[blk];
  const int qw0 = qb0 + wave * 32;  // first query row of this wave (inside the sequence)
  const int qi = qw0 + l31;
  const int W = p.window;

  // Q fragments (B operand): k-slot (8*hi + j) of step s <-> d = 16*s + 8*hi + j
  bf16x8 qf[4];
  {
    const int row = min(t0 + qi, Tp - 1);
    const bf16_t* qrow = p.q + (size_t)row * H + head * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
  }

  int kb_lo = 0, kb_hi = (S - 1) >> 6;
  if constexpr (LOCAL) {
    kb_lo = max(0, qb0 - W) >> 6;
    kb_hi = min(S - 1, qb0 + 127 + W) >> 6;
  }

  // staging roles
  const int srow = tid >> 2;        // K: key row 0..63 ; V^T: d row 0..63
  const int spiece = (tid & 3) * 2; // two 16-byte pieces per thread
  const int ksw = (srow >> 1) & 7;
  const int vsw = (srow >> 1) & 15;

  f32x4 kreg[2], vreg[2];  // raw 16-byte payloads in flight
  auto load_tile = [&](int kb) {
    const int krow = min(t0 + kb * 64 + srow, Tp - 1);
    const bf16_t* ksrc = p.k + (size_t)krow * H + head * 64;
    const bf16_t* vsrc = p.vt + (size_t)(head * 64 + srow) * Tp;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      kreg[i] = *reinterpret_cast<const f32x4*>(ksrc + (spiece + i) * 8);
      const int col = min(t0 + kb * 64 + (spiece + i) * 8, Tp - 8);
      vreg[i] = *reinterpret_cast<const f32x4*>(vsrc + col);
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = spiece + i;
      *reinterpret_cast<f32x4*>(sK + srow * 128 + ((c ^ ksw) << 4)) = kreg[i];
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 lo, hi2;
      lo[0] = vreg[i][0]; lo[1] = vreg[i][1];
      hi2[0] = vreg[i][2]; hi2[1] = vreg[i][3];
      *reinterpret_cast<f32x2*>(sV + srow * 128 + (((2 * c) ^ vsw) << 3)) = lo;
      *reinterpret_cast<f32x2*>(sV + srow * 128 + (((2 * c + 1) ^ vsw) << 3)) = hi2;
    }
  };

  f32x16 ot[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[n][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;

  const int fsw = (l31 >> 1) & 7;   // K fragment swizzle
  const int vfs = (l31 >> 1) & 15;  // V^T fragment swizzle

  load_tile(kb_lo);
  for (int kb = kb_lo; kb <= kb_hi; ++kb) {
    __syncthreads();  // everyone finished reading the previous tile
    write_tile();
    __syncthreads();
    if (kb < kb_hi) load_tile(kb + 1);  // in flight during the MFMA work below

    bool active = qw0 < S;
    if constexpr (LOCAL) {
      active = active && (kb * 64 + 63 >= qw0 - W) && (kb * 64 <= qw0 + 31 + W);
    }
    if (!active) continue;  // wave-uniform

    // ---- S^T = K . Q^T
    f32x16 st[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + (t * 32 + l31) * 128 + (((2 * s + hi) ^ fsw) << 4));
        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[t], 0, 0, 0);
      }
    }
    // ---- mask + online softmax (exp2 domain)
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = kb * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        bool ok = kk < S;
        if constexpr (LOCAL) ok = ok && (kk - qi <= W) && (qi - kk <= W);
        const float x = ok ? st[t][r] * LOG2E : -INFINITY;
        st[t][r] = x;
        mx = fmaxf(mx, x);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(st[t][r] - m_new);
        psum += pv;
        pf[t][r >> 3][r & 7] = (bf16_t)pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[n][r] *= alpha;

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c8 = 8 * t + 4 * hf + hi;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const char* vrow = sV + (n * 32 + l31) * 128;
          bf16x4 a0 = *reinterpret_cast<const bf16x4*>(vrow + ((c8 ^ vfs) << 3));
          bf16x4 a1 = *reinterpret_cast<const bf16x4*>(vrow + (((c8 + 2) ^ vfs) << 3));
          bf16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = a0[j]; vf[4 + j] = a1[j]; }
          ot[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[t][hf], ot[n], 0, 0, 0);
        }
      }
  }

  // ---- normalise and store O[q][head*64 + d]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qi < S) {
    const float inv = 1.0f / l_tot;
    bf16_t* orow = p.o + (size_t)(t0 + qi) * H + head * 64;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(ot[n][4 * g + j] * inv);
        *reinterpret_cast<bf16x4*>(orow + n * 32 + 8 * g + 4 * hi) = o;
      }
  }
}

hipError_t launch_attention(const AttnParams& p, bool local, hipStream_t stream) {
  if (p.n_blocks <= 0) return hipSuccess;
  dim3 grid(p.n_blocks, p.nh);
  if (local)
    hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, dim3(256), 0, stream, p);
  return hipGetLastError();
}

}  // namespace vrag
