// Flash-style packed attention for gfx950 (head_dim 64), replacing the reference's
// eager/SDPA attention inside ModernBertAttention.forward
// (transformers modeling_modernbert.py:166-185,286-299; mask: masking_utils.py:141-151).
//
// Work item = (sequence, query block, head): 256 rows / 4 waves on global layers, 128 rows / 2 waves
// on banded layers; every wave owns 64 query rows (two 32-row sub-tiles that SHARE every K / V^T
// fragment read from LDS).  64-key tiles stream
// through a 3-slot LDS ring by 16-byte LDS-DMA (global_load_lds), two tiles in flight, one raw
// s_barrier per tile and a counted s_waitcnt vmcnt (never a full drain in steady state).
// Per tile and sub-tile:
//   S^T[key][q] = K . Q^T        v_mfma_f32_32x32x16_bf16, A = K tile (LDS), B = Q (registers)
//     -> lane (q = lane&31) holds 16 keys of ITS query row per 32-key tile, so the softmax
//        row max / sum are in-lane + one cross-half exchange (lane ^ 32).
//   O^T[d][q] += V^T . P^T       A = V^T tile (LDS, key-contiguous), B = P straight from the
//        S^T accumulators (the accumulator row map IS the B-operand k-slot order once the
//        V^T fragment is read with the same key permutation) -> no LDS round trip for P and
//        the per-row rescale factor lives in the lane that owns the O^T column.
// LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled on the DMA source address and on
// the read address: K by ((key>>1)&7) (ds_read_b128 conflict free), V^T by ((d>>1)&7)
// (ds_read_b64, 2-way).  Softmax is fp32 (exp2 domain); P and the MFMA operands are bf16.
// Banded layers (|i-j| <= window) visit only the key tiles inside the block's band and each
// wave skips tiles outside its own 64 rows' band.
#include "attention.h"

#include <cstdlib>

namespace vrag {

constexpr int ATT_QB_GLOBAL = 256;  // query rows per workgroup, global layers (4 waves)
constexpr int ATT_QB_LOCAL = 256;   // banded layers (128 rows / 2 waves measured slower: 4.8 vs 4.4 ms per step)
constexpr int ATT_TILE = 16384;  // bytes per LDS ring slot (K 8 KiB + V^T 8 KiB)
constexpr int ATT_SLOTS = 3;

template <bool LOCAL, int NW, typename T>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const AttnParams p) {
  typedef typename Op<T>::v4 V4;
  typedef typename Op<T>::v8 V8;
  constexpr int ATT_QB = NW * 64;
  constexpr int NI = 8 / NW;  // LDS-DMA instructions per wave per operand per tile (8 rows each)
  __shared__ __attribute__((aligned(16))) char smem[ATT_SLOTS * ATT_TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int blk = blockIdx.x, head = blockIdx.y;
  const int H = p.H, Tp = p.Tp;

  const int t0 = p.blk_seq_start[blk];
  const int S = p.blk_seq_len[blk];
  const int qb0 = p.blk_q0[blk];
  const int qw0 = qb0 + wave * 64;  // first query row of this wave (inside the sequence), global layers
  const int W = p.window;
  // First query row of sub-tile u.  Global layers: the wave's 64 consecutive rows.  Banded layers: rows 32*(wave + NW*u) of
  // the block, i.e. the wave's two sub-tiles sit half a block apart -- their bands cover different key tiles, so in every
  // tile step each wave has about half a tile of work instead of some waves a whole tile and the others none (the
  // workgroup walks its 6-7 tiles in lock-step; with consecutive rows a wave had work in only 3 of them).
  auto qlo = [&](int u) { return LOCAL ? qb0 + 32 * (wave + NW * u) : qw0 + 32 * u; };

  // Q fragments (B operand) for both sub-tiles: k-slot (8*hi + j) of step s <-> d = 16*s + 8*hi + j
  V8 qf[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int row = min(t0 + qlo(u) + l31, Tp - 1);
    const T* qrow = reinterpret_cast<const T*>(p.q) + (size_t)row * H + head * 64 + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[u][s] = *reinterpret_cast<const V8*>(qrow + 16 * s);
  }
  // Retire the Q loads HERE: an ordinary load still pending when the loop starts makes hipcc wait
  // vmcnt(0) at its first use inside the loop on every iteration, which drains the LDS-DMA
  // prefetch ring (two tiles ahead) each tile.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[u][s]));

  int kb_lo = 0, kb_hi = (S - 1) >> 6;
  if constexpr (LOCAL) {
    kb_lo = max(0, qb0 - W) >> 6;
    kb_hi = min(S - 1, qb0 + ATT_QB - 1 + W) >> 6;
  }

  // LDS-DMA roles: instruction i of this wave covers tile rows wave*(64/NW) + i*8 + (lane>>3)
  const int drow0 = wave * (64 / NW) + (lane >> 3);
  const int dchunk = lane & 7;
  auto stage = [&](int kb) {
    char* slot = smem + (kb % ATT_SLOTS) * ATT_TILE;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = drow0 + i * 8;  // K: key row, V^T: d row
      const int lc = dchunk ^ ((row >> 1) & 7);
      const int krow = min(t0 + kb * 64 + row, Tp - 1);
      glds16(p.k + (size_t)krow * H + head * 64 + lc * 8, slot + (wave * (64 / NW) + i * 8) * 128);
      const int col = min(t0 + kb * 64 + lc * 8, Tp - 8);
      glds16(p.vt + (size_t)(head * 64 + row) * Tp + col, slot + 8192 + (wave * (64 / NW) + i * 8) * 128);
    }
  };

  f32x16 ot[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[u][n][r] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  const int fsw = (l31 >> 1) & 7;  // fragment swizzle (same for K rows and V^T rows)

  stage(kb_lo);
  if (kb_lo + 1 <= kb_hi) stage(kb_lo + 1);
  for (int kb = kb_lo; kb <= kb_hi; ++kb) {
    // tile kb landed (this wave's part); the next tile (2*NI DMA instructions) may stay in flight
    if (kb < kb_hi) {
      if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kb + 2 <= kb_hi) stage(kb + 2);  // slot of tile kb-1: every wave finished it before the barrier

    bool active = qw0 < S;
    if constexpr (LOCAL) {
      active = false;
#pragma unroll
      for (int u = 0; u < 2; ++u) active = active || (qlo(u) < S && kb * 64 + 63 >= qlo(u) - W && kb * 64 <= qlo(u) + 31 + W);
    }
    if (!active) continue;  // wave-uniform

    const char* sK = smem + (kb % ATT_SLOTS) * ATT_TILE;
    const char* sV = sK + 8192;

    // ---- S^T = K . Q^T for both sub-tiles (K fragments read once)
    f32x16 st[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[u][t][r] = 0.f;
    // Banded layers: a 32-key half of the tile is, for a 32-row sub-tile, either completely outside the band (no row sees any
    // of its keys: skip its MFMAs and its softmax), completely inside (no mask) or triangular.  Of the six halves a sub-tile
    // meets in its three tiles one is outside, three inside, two triangular (window 64, 32-aligned rows).
    bool skip[2][2] = {{false, false}, {false, false}}, tri[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int q_lo = qlo(u), kh = kb * 64 + t * 32;
        tri[u][t] = kh + 31 >= S;
        if constexpr (LOCAL) {
          skip[u][t] = (kh + 31 < q_lo - W) || (kh > q_lo + 31 + W) || kh >= S || q_lo >= S;
          tri[u][t] = tri[u][t] || (kh < q_lo + 31 - W) || (kh + 31 > q_lo + W);
        }
      }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (skip[0][t] && skip[1][t]) continue;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const V8 kf = *reinterpret_cast<const V8*>(sK + (t * 32 + l31) * 128 + (((2 * s + hi) ^ fsw) << 4));
        if (!skip[0][t]) st[0][t] = Op<T>::mfma32(kf, qf[0][s], st[0][t]);
        if (!skip[1][t]) st[1][t] = Op<T>::mfma32(kf, qf[1][s], st[1][t]);
      }
    }

    // ---- mask + online softmax (scores arrive in log2 units: q was pre-scaled by d^-1/2 * log2 e),
    //      P -> bf16 B-operand fragments.  Interior tiles (every key valid and inside every row's
    //      band) take the mask-free path: the VALU, not the MFMA, is the busier pipe at head_dim 64.
    V8 pf[2][2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q_lo = qlo(u);
      float mx = -1e30f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (skip[u][t]) continue;   // wave-uniform
        if (tri[u][t]) {
          // visible keys of row qi form one interval [lo, hi_]: key(r) = kbase + c_r is visible iff
          // (unsigned)(kbase + c_r - lo) <= hi_ - lo  -> one add + one unsigned compare per element
          int kbase = kb * 64 + t * 32 + 4 * hi;
          asm volatile("" : "+v"(kbase));  // keep the predicate arithmetic inside this branch
          const int qi = q_lo + l31;
          int lo = 0, hi_ = S - 1;
          if constexpr (LOCAL) {
            lo = max(0, qi - W);
            hi_ = min(S - 1, qi + W);
          }
          const unsigned span = hi_ >= lo ? (unsigned)(hi_ - lo) : 0u;
          const int d0 = hi_ >= lo ? kbase - lo : -100000;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = (unsigned)(d0 + ((r & 3) + 8 * (r >> 2))) <= span;
            const float x = ok ? st[u][t][r] : -INFINITY;
            st[u][t][r] = x;
            mx = fmaxf(mx, x);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[u][t][r]);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[u], mx);
      const bool grew = !__all(m_new == m_run[u]);  // wave-uniform
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (skip[u][t]) continue;   // its P.V MFMAs are skipped as well
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(st[u][t][r] - m_new);
          psum += pv;
          pf[u][t][r >> 3][r & 7] = (T)pv;   // p in [0, 1]: no saturation needed
        }
      }
      if (grew) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
        m_run[u] = m_new;
        l_run[u] = l_run[u] * alpha + psum;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[u][n][r] *= alpha;
      } else {
        l_run[u] += psum;
      }
    }

    // ---- O^T += V^T . P^T (V^T fragments read once for both sub-tiles)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (skip[0][t] && skip[1][t]) continue;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int c16 = 4 * t + 2 * hf;  // 16-byte chunk of keys t*32 + hf*16 .. ; +1 = the next 8 keys
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const char* vrow = sV + (n * 32 + l31) * 128 + (hi << 3);
          const V4 a0 = *reinterpret_cast<const V4*>(vrow + ((c16 ^ fsw) << 4));
          const V4 a1 = *reinterpret_cast<const V4*>(vrow + (((c16 + 1) ^ fsw) << 4));
          V8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = a0[j];
            vf[4 + j] = a1[j];
          }
          if (!skip[0][t]) ot[0][n] = Op<T>::mfma32(vf, pf[0][t][hf], ot[0][n]);
          if (!skip[1][t]) ot[1][n] = Op<T>::mfma32(vf, pf[1][t][hf], ot[1][n]);
        }
      }
    }
  }

  // ---- normalise and store O[q][head*64 + d].  A lane holds 4 consecutive dims of ITS row per register group, i.e.
  // 8-byte pieces at a 2H-byte row stride; staged through the (now idle) LDS ring -- 8 KiB per wave, 16-byte chunk index
  // XOR (row & 7) -- every store instruction writes 8 whole 128-byte head rows instead of 64 scattered 8-byte pieces.
  __builtin_amdgcn_s_barrier();   // every wave is past its last K / V^T tile read before the ring is reused
  asm volatile("" ::: "memory");
  char* stg = smem + wave * 8192;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
    const float inv = 1.0f / l_tot;
    const int row = u * 32 + l31;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        V4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = Op<T>::to(ot[u][n][4 * g + j] * inv);
        *reinterpret_cast<V4*>(stg + row * 128 + (((n * 4 + g) ^ (row & 7)) << 4) + (hi << 3)) = o;
      }
  }
  // the staging area is private to the wave: program order + the compiler's lgkmcnt wait order the reads below
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), c16 = lane & 7;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 128 + ((c16 ^ (row & 7)) << 4));
    const int grow = qlo(row >> 5) + (row & 31);   // row of the sequence this staged row belongs to
    if (grow < S)
      *reinterpret_cast<f32x4*>(reinterpret_cast<T*>(p.o) + (size_t)(t0 + grow) * H + head * 64 + c16 * 8) = v;
  }
}


int attention_q_block(bool local) { return local ? ATT_QB_LOCAL : ATT_QB_GLOBAL; }

hipError_t launch_attention(const AttnParams& p, bool local, hipStream_t stream) {
  if (p.n_blocks <= 0) return hipSuccess;
  dim3 grid(p.n_blocks, p.nh);
  if (p.op_dtype == kOpF16) {
    if (local) hipLaunchKernelGGL((attn_fwd_kernel<true, ATT_QB_LOCAL / 64, f16_t>), grid, dim3(ATT_QB_LOCAL), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, ATT_QB_GLOBAL / 64, f16_t>), grid, dim3(ATT_QB_GLOBAL), 0, stream, p);
  } else {
    if (local) hipLaunchKernelGGL((attn_fwd_kernel<true, ATT_QB_LOCAL / 64, bf16_t>), grid, dim3(ATT_QB_LOCAL), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, ATT_QB_GLOBAL / 64, bf16_t>), grid, dim3(ATT_QB_GLOBAL), 0, stream, p);
  }
  return hipGetLastError();
}

unsigned attention_f16_saturated(bool reset) { return f16_sat_take(reset); }
unsigned* attention_f16_flag_address() { return f16_sat_flag_address(); }

}  // namespace vrag
