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


// ---------------------------------------------------------------------------------------------------------------
// Second-generation kernel (round 3).  Same work decomposition, LDS ring and DMA as attn_fwd_kernel above -- (sequence,
// 256-row block, head) per workgroup, 4 waves x 64 query rows, 64-key tiles -- with the arithmetic re-laid for the VALU
// budget (the first kernel spends ~5.5 VALU issue slots per score, the matrix pipe idles at 0.10-0.28):
//   * v_mfma_f32_16x16x32: a wave's 64 rows are four 16-query column blocks c; S^T[key][q] tiles are 16 keys x 16 queries,
//     lane (q = lane & 15, g = lane >> 4) holds keys 16 kt + 4 g + r of ITS query -- every K / V^T fragment read from LDS
//     feeds four MFMAs, and each fp32 accumulator is touched once per 32 k (the GEMMs' energy argument);
//   * the running maximum is LAZY: the S^T accumulators are seeded with -m_run (the MFMA's C operand), so scores arrive
//     relative to the reference already and p = exp2(s) needs no subtraction; the reference is only moved when some
//     score exceeds it by more than 2^8 (or a row meets its first visible key) -- a wave-uniform, rare branch that pays
//     the cross-lane exchange, the subtraction and the O rescale.  Softmax is
//     invariant to the reference; p <= 256 is exact enough in bf16 / far inside fp16's range;
//   * row sums come from the matrix pipe: one extra MFMA per 32 keys with an all-ones A operand accumulates
//     l[q] = sum_k bf16(p) -- the very operands the P.V product sums, so numerator and denominator stay consistent;
//   * the in-lane maximum uses v_max3.
// -> per score: 1/2 (max3) + 1 (exp2, ~2 slots) + 1/2 (packed convert) issue slots instead of ~5.5.
// Cross-lane maxima over the four lane groups of a query (lanes l, l^16, l^32, l^48).  They run only when the reference moves
// (rare), so they take the plain LDS-crossbar shuffle: the v_permlane16/32_swap forms tried first returned the even row's /
// lower half's value instead of the maximum here (tools/probes/permlane_probe.hip) and broke the banded layers whenever a lane
// group had no visible key in a row's first tile.
__device__ __forceinline__ float xor16_max(float v) { return fmaxf(v, __shfl_xor(v, 16, 64)); }
__device__ __forceinline__ float xor32_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // one v_max3_f32

constexpr float ATT_LAZY = 8.0f;   // log2 units: the reference moves when a score exceeds it by more than 2^8

template <bool LOCAL, typename T>
__global__ __launch_bounds__(256, 2) void attn2_fwd_kernel(const AttnParams p) {
  typedef typename Op<T>::v4 V4;
  typedef typename Op<T>::v8 V8;
  constexpr int NW = 4, NI = 2;
  __shared__ __attribute__((aligned(16))) char smem[ATT_SLOTS * ATT_TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int blk = blockIdx.x, head = blockIdx.y;
  const int H = p.H, Tp = p.Tp;
  const int t0 = p.blk_seq_start[blk];
  const int S = p.blk_seq_len[blk];
  const int qb0 = p.blk_q0[blk];
  const int W = p.window;
  // first query row of column block c (16 rows).  Global layers: the wave's 64 consecutive rows.  Banded layers: two 32-row
  // groups half a block apart (rows 32 * (wave + 4 u)), as in the first kernel: every wave has work in almost every tile.
  auto crow = [&](int c) { return LOCAL ? qb0 + 32 * (wave + NW * (c >> 1)) + 16 * (c & 1) : qb0 + wave * 64 + 16 * c; };

  // Q fragments (B operand of S^T = K . Q^T): lane (query l15, group g) holds d = 32 s + 8 g .. + 8
  V8 qf[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int row = min(t0 + crow(c) + l15, Tp - 1);
    const T* qrow = reinterpret_cast<const T*>(p.q) + (size_t)row * H + head * 64 + 8 * g;
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[c][s] = *reinterpret_cast<const V8*>(qrow + 32 * s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // see attn_fwd_kernel: no plain load may be pending inside the tile loop
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(qf[c][s]));

  int kb_lo = 0, kb_hi = (S - 1) >> 6;
  if constexpr (LOCAL) {
    kb_lo = max(0, qb0 - W) >> 6;
    kb_hi = min(S - 1, qb0 + NW * 64 - 1 + W) >> 6;
  }
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

  f32x4 ot[4][4];   // O^T[d = 16 dt + 4 g + r][query l15] of column block c
  f32x4 lo[4];      // row sums of column block c (all four registers hold the same value)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lo[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float m_run[4] = {0.f, 0.f, 0.f, 0.f};   // the reference of the lane's query in block c (log2 units); meaningful once seen
  bool seen[4] = {false, false, false, false};
  V8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;
  const int fsw = (l15 >> 1) & 7;   // fragment swizzle of a row 16 x + l15 (K rows and V^T rows alike)

  stage(kb_lo);
  if (kb_lo + 1 <= kb_hi) stage(kb_lo + 1);
  for (int kb = kb_lo; kb <= kb_hi; ++kb) {
    if (kb < kb_hi) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kb landed; the next one (2 * NI DMA instructions) stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kb + 2 <= kb_hi) stage(kb + 2);

    // 32-row group u = c >> 1 against the 32-key half t2 of the tile: outside the band / beyond the sequence -> skipped
    // (no MFMA, p = 0); cut by the band edge or the sequence end -> masked element by element; else mask-free
    bool skip[2][2], tri[2][2];
    bool any_work = false;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int q_lo = crow(2 * u), kh = kb * 64 + t2 * 32;
        skip[u][t2] = kh >= S || q_lo >= S;
        tri[u][t2] = kh + 31 >= S;
        if constexpr (LOCAL) {
          skip[u][t2] = skip[u][t2] || (kh + 31 < q_lo - W) || (kh > q_lo + 31 + W);
          tri[u][t2] = tri[u][t2] || (kh < q_lo + 31 - W) || (kh + 31 > q_lo + W);
        }
        any_work = any_work || !skip[u][t2];
      }
    if (!any_work) continue;   // wave-uniform

    const char* sK = smem + (kb % ATT_SLOTS) * ATT_TILE;
    const char* sV = sK + 8192;

    // ---- per 32-row group u (column blocks 2u, 2u + 1; half of the score registers live at a time):
    //      S^T - m_run = K . Q^T - m_run (the reference rides in as the C operand), masks, lazy reference, p = exp2(.)
    V8 pf[4][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (skip[u][0] && skip[u][1]) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[2 * u + cc][t2][j] = (T)0.f;
        continue;
      }
      f32x4 st[2][4];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const float seed = p.v2_noseed ? 0.f : -m_run[2 * u + cc];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) st[cc][kt] = f32x4{seed, seed, seed, seed};
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if (skip[u][kt >> 1]) continue;
        const char* krow = sK + (kt * 16 + l15) * 128;
        const V8 k0 = *reinterpret_cast<const V8*>(krow + ((g ^ fsw) << 4));
        const V8 k1 = *reinterpret_cast<const V8*>(krow + (((4 + g) ^ fsw) << 4));
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          st[cc][kt] = Op<T>::mfma16(k0, qf[2 * u + cc][0], st[cc][kt]);
          st[cc][kt] = Op<T>::mfma16(k1, qf[2 * u + cc][1], st[cc][kt]);
        }
      }
      if (p.v2_noseed) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[cc][kt][r] -= m_run[2 * u + cc];
      }
      // masks and in-lane maxima (scores are relative to m_run already)
      float mx[2];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = 2 * u + cc;
        mx[cc] = -INFINITY;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          if (skip[u][t2]) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) st[cc][2 * t2 + h2] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            continue;
          }
          if (tri[u][t2]) {
            const int qi = crow(c) + l15;
            int lo_k = 0, hi_k = S - 1;
            if constexpr (LOCAL) {
              lo_k = max(0, qi - W);
              hi_k = min(S - 1, qi + W);
            }
            const unsigned span = hi_k >= lo_k ? (unsigned)(hi_k - lo_k) : 0u;
            int d0 = hi_k >= lo_k ? kb * 64 + t2 * 32 + 4 * g - lo_k : -100000;
            asm volatile("" : "+v"(d0));
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool ok = (unsigned)(d0 + 16 * h2 + r) <= span;
                st[cc][2 * t2 + h2][r] = ok ? st[cc][2 * t2 + h2][r] : -INFINITY;
              }
          }
          const f32x4& a = st[cc][2 * t2];
          const f32x4& b = st[cc][2 * t2 + 1];
          mx[cc] = max3f(mx[cc], max3f(a[0], a[1], a[2]), max3f(a[3], b[0], b[1]));
          mx[cc] = max3f(mx[cc], b[2], b[3]);
        }
      }
      // move the reference?  (wave-uniform; steady state: no)
      bool move = false;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) move = move || mx[cc] > p.v2_lazy || (!seen[2 * u + cc] && mx[cc] > -INFINITY);
      if (__any(move)) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c = 2 * u + cc;
          const float m_all = xor32_max(xor16_max(mx[cc]));          // over the four lane groups of the query
          const bool has = m_all > -INFINITY;
          const float delta = seen[c] ? fmaxf(m_all, 0.f) : (has ? m_all : 0.f);
          // a row's first reference may sit far below zero: nothing has been accumulated yet, so nothing is rescaled
          // (2^-delta would overflow); afterwards delta >= 0 and alpha <= 1
          const float alpha = seen[c] ? __builtin_amdgcn_exp2f(-delta) : 1.0f;
          seen[c] = seen[c] || has;
          m_run[c] += delta;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[cc][kt][r] -= delta;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[c][dt][r] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) lo[c][r] *= alpha;
        }
      }
      // p = exp2(s - m), straight into the B-operand fragments of O^T += V^T . P^T: k-slot (g, j) of the 32-key step t2 is
      // key 32 t2 + 4 g + j (j < 4) or 32 t2 + 16 + 4 g + (j - 4) -- the accumulator rows of key tiles 2 t2 and 2 t2 + 1
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[2 * u + cc][t2][j] = (T)__builtin_amdgcn_exp2f(st[cc][2 * t2 + (j >> 2)][j & 3]);
    }

    // ---- O^T += V^T . P^T, l += 1 . P^T (V^T fragments read once for the four column blocks)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      if (skip[0][t2] && skip[1][t2]) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (!skip[c >> 1][t2]) lo[c] = Op<T>::mfma16(ones, pf[c][t2], lo[c]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // keys 32 t2 + 4 g .. + 4 and 32 t2 + 16 + 4 g .. + 4 of V^T row d = 16 dt + l15: 16-byte chunks 4 t2 + (g >> 1) and + 2, half g & 1
        const char* vrow = sV + (dt * 16 + l15) * 128 + ((g & 1) << 3);
        const V4 a0 = *reinterpret_cast<const V4*>(vrow + (((4 * t2 + (g >> 1)) ^ fsw) << 4));
        const V4 a1 = *reinterpret_cast<const V4*>(vrow + (((4 * t2 + 2 + (g >> 1)) ^ fsw) << 4));
        V8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = a0[j];
          vf[4 + j] = a1[j];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (!skip[c >> 1][t2]) ot[c][dt] = Op<T>::mfma16(vf, pf[c][t2], ot[c][dt]);
      }
    }
  }

  // ---- normalise and store (staged through the idle ring: whole 128-byte head rows per store instruction)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  char* stg = smem + wave * 8192;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float inv = 1.0f / lo[c][0];
    const int row = 16 * c + l15;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      V4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = Op<T>::to(ot[c][dt][r] * inv);
      // dims 16 dt + 4 g .. + 4 -> 16-byte chunk 2 dt + (g >> 1), half g & 1
      *reinterpret_cast<V4*>(stg + row * 128 + (((2 * dt + (g >> 1)) ^ (row & 7)) << 4) + ((g & 1) << 3)) = o;
    }
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3), c16 = lane & 7;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 128 + ((c16 ^ (row & 7)) << 4));
    const int grow = crow(row >> 4) + (row & 15);
    if (grow < S)
      *reinterpret_cast<f32x4*>(reinterpret_cast<T*>(p.o) + (size_t)(t0 + grow) * H + head * 64 + c16 * 8) = v;
  }
}

int attention_q_block(bool local) { return local ? ATT_QB_LOCAL : ATT_QB_GLOBAL; }

hipError_t launch_attention(const AttnParams& p, bool local, hipStream_t stream) {
  if (p.n_blocks <= 0) return hipSuccess;
  dim3 grid(p.n_blocks, p.nh);
  // The second-generation kernel is opt-in (VRAG_ATTN_V2=1): measured slower than the first one (global 191-202 vs 177 us,
  // banded 118-124 vs 115 us per 65 536-token launch -- 45 % fewer VALU slots per score but 12 % more MFMA issue and 1.5x
  // the LDS fragment reads; the kernel is stall-bound, not VALU-bound).  Its accuracy equals the first kernel's since the
  // cross-lane maxima use the plain shuffle (tests/test_attention_unit_gpu.py checks both kernels against a float64 softmax).
  static const bool v2 = getenv("VRAG_ATTN_V2") != nullptr;
  if (v2) {
    static const float lazy = getenv("VRAG_ATTN_V2_LAZY") ? (float)atof(getenv("VRAG_ATTN_V2_LAZY")) : ATT_LAZY;
    static const int noseed = getenv("VRAG_ATTN_V2_NOSEED") ? 1 : 0;
    AttnParams pv = p;
    pv.v2_lazy = lazy;
    pv.v2_noseed = noseed;
    const AttnParams& p = pv;
    if (p.op_dtype == kOpF16) {
      if (local) hipLaunchKernelGGL((attn2_fwd_kernel<true, f16_t>), grid, dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((attn2_fwd_kernel<false, f16_t>), grid, dim3(256), 0, stream, p);
    } else {
      if (local) hipLaunchKernelGGL((attn2_fwd_kernel<true, bf16_t>), grid, dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((attn2_fwd_kernel<false, bf16_t>), grid, dim3(256), 0, stream, p);
    }
    return hipGetLastError();
  }
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

}  // namespace vrag
