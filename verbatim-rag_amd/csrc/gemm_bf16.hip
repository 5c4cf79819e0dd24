// 16-bit MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T, fp32 accumulate, fused epilogues.
//
// Replaces the nn.Linear calls of the reference's ModernBERT forward
// (transformers modeling_modernbert.py: Wqkv :271, attn Wo :300, mlp Wi :90, mlp Wo :91,
//  prediction-head dense :487, MLM decoder :550, token classifier :697).
//
// Two tile configurations of one template (v_mfma_f32_16x16x32_{bf16,f16}, BK = 64):
//   256(M) x 256(N): 512 threads = 8 waves as 2(M) x 4(N), 128x64 per wave (128 accumulator registers),
//                    128 KiB LDS (2 stages x (32 KiB A + 32 KiB W)), 1 workgroup / CU  -- used when N % 256 == 0
//   128(M) x 128(N): 256 threads = 4 waves as 2x2, 64x64 per wave, 64 KiB LDS, 2 workgroups / CU
// The larger tile halves the L2->LDS operand traffic per FLOP (M*N*K*2*(1/BM+1/BN) bytes), which is
// the co-limiter of the small tile at these K (768/1152).  Operands go HBM -> LDS by 16-byte LDS-DMA (global_load_lds),
// double buffered.  LDS rows are 128 B (64 k-values); the 16-byte chunk index is XOR-swizzled with
// ((row>>1)&7) on the *source address* (LDS-DMA writes lane-linear) and on the ds_read_b128
// address, which makes each 16-lane read group of a 16-row x 32-k fragment hit the 64 banks exactly once.
//
// Why the 16x16x32 instruction and not 32x32x16 (same FLOP/s on paper, and the 32x32 shape reads each operand fragment
// half as often): the step is power-limited, and the wider-K shape reads and writes each fp32
// accumulator once per 32 k instead of once per 16 -- measured on the main loop alone, same operands, fragments and
// accumulator count: 196 vs 217 us at M = 65536 / N = 2304 / K = 768, 67 vs 75 us at N = 768, 101 vs 112 us at
// N = 768 / K = 1152 (profiles/r02_mfma_shape_probe.txt).
//
// The MFMA is issued "swapped" (weights as the A operand, activations as B) so that a lane's
// accumulator registers hold 4 *consecutive output features* of ONE token row:
//   token   m = m0 + wm*WROWS + rt*16 + (lane&15)            rt = 16-row tile of the wave
//   feature n = n0 + wn*64 + nj*16 + 4*(lane>>4) + r         nj = 16-column tile, r = register 0..3
// -> 8-byte 16-bit / 16-byte fp32 row-major stores, and RoPE's (d, d+32) / GeGLU's (x1, x2)
// partners sit in the same lane and register of acc[nj][rt] / acc[nj+2][rt].
#include "gemm_bf16.h"

#include <atomic>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace vrag {

constexpr int BK = 64;   // K-step: one LDS stage holds 64 k-values of every tile row

__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// sum over each aligned group of 8 lanes (DPP: xor 1, xor 2, mirror within 8), result in every lane of the group
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  return v;
}

// EPI_RESIDUAL with the residual stream split into the operand plane and a BYTE remainder plane on one or both sides
// (GemmParams::lo_in / lo_out).  Same staging as the fp32 form (a pass = 64 rows x 64 columns of accumulators through the wave's
// 16 KiB, half-passes of 32 rows, the arriving rows of a half prefetched one half ahead); a lane takes EIGHT consecutive columns of
// its row, so the high plane moves in 16-byte pieces per lane and whole 128-byte lines per 8 lanes.
// Round 6: the low plane is ONE byte per element, relative to the high plane:  v = hi * (1 + (b - 128) * kLoStep),
// b = rne(((v / hi) - 1) / kLoStep) + 128 saturated to [0, 255]  (|v / hi - 1| <= 2^-8 for bf16, 2^-11 for fp16: 16 / 19
// significant bits together; hi = 0 decodes to 0 whatever the byte).  One v_cvt_f32_ubyteN + one packed FMA + one packed
// multiply to decode two elements, one v_rcp + one multiply + one FMA + one v_cvt_pk_u8_f32 to encode one.  The byte plane is
// touched by this epilogue only, so its layout is the epilogue's own (lo8_offset below): per 64 x 64 block of the stream 4 KiB,
// and inside it the 16 bytes a lane needs for TWO consecutive instructions of a half-pass are contiguous -- every access to the
// plane is a 16-byte access of a lane-linear 1 KiB run.  (Round 4 probed a byte plane twice and dropped it: with 8 columns per
// lane in row-major order its accesses are 8-byte pieces, which cost the memory pipeline what 16-byte pieces cost, and the
// bit-field packing took ~20 integer instructions per element: profiles/r04_split_lo8_*_probe.txt.)
// Arithmetic per element: v = (acc + h_in) + off, off = [c_prev if arriving split] - [c if leaving split];
// leaving split: hi = op16(v), byte as above, statistics of v (sum, sum of squares per 64-column slice) as the fp32 form.
template <typename T> struct LoStep;
template <> struct LoStep<bf16_t> { static constexpr float step = 1.0f / 32768.0f, inv = 32768.0f; };      // 2^-15: 2^-8 / 128
template <> struct LoStep<f16_t> { static constexpr float step = 1.0f / 262144.0f, inv = 262144.0f; };    // 2^-18: 2^-11 / 128

// byte offset of the 16-byte piece of (64 x 64 block at rows rb64 * 64, columns cb64 * 64; half-pass half; instruction pair ip; lane)
__device__ __forceinline__ size_t lo8_offset(int rb64, int cb64, int n_cb, int half, int ip, int lane) {
  return ((size_t)rb64 * n_cb + cb64) * 4096 + (size_t)(half * 2048 + ip * 1024 + lane * 16);
}

// The optional sides (arriving split, leaving split, row statistics) are wave-uniform launch parameters: the body is compiled per
// combination and entered through ONE dispatch per tile (residual_split_epilogue below).  Left as run-time tests inside the body
// the compiler merges the two arms' loads into one load with a run-time destination slot -- the prefetch sets then live in
// scratch, every load followed by s_waitcnt vmcnt(0) and a scratch store.
template <int RT, typename T, bool in_split, bool out_split, bool STATS>
__device__ __forceinline__ void residual_split_body(const GemmParams& p, f32x4 (&acc)[4][RT], char* stg, int lane, int mw, int nw) {
  typedef typename Op<T>::v8 V8;
  constexpr int NI = 4;   // instructions per half-pass: 8 rows x (8 lanes x 8 columns)
  constexpr float kStep = LoStep<T>::step, kInv = LoStep<T>::inv;
  const int q = lane >> 4, l15 = lane & 15, gq = lane >> 3, l7 = lane & 7;
  float off[RT / 4];   // lane L: the offset of row ps * 64 + L
#pragma unroll
  for (int ps = 0; ps < RT / 4; ++ps) {
    const int row = mw + ps * 64 + lane;
    off[ps] = (in_split ? p.ln_shift_prev[row] : 0.f) - (out_split ? p.ln_shift[row] : 0.f);
  }
  // per half-pass and set: four 16-byte pieces of the high plane (or eight of the fp32 rows) + two of the byte plane
  f32x4 hA[NI][2], hB[NI][2];
  // addresses as (wave-uniform row base) + (one 32-bit lane offset): a 64-bit lane address per plane and instruction would cost
  // the epilogue the registers its two prefetch sets need
  unsigned loff = (unsigned)(gq * p.N + 8 * l7);   // elements, inside the 8-row group of an instruction
  unsigned boff = (unsigned)(lane * 16);           // bytes, inside the 1 KiB run of an instruction pair
  asm volatile("" : "+v"(loff));
  asm volatile("" : "+v"(boff));
  const int n_cb = p.N >> 6;
  auto row_base = [&](int hp, int i) {   // uniform: first element of the instruction's first row
    return ((size_t)(mw + (hp >> 1) * 64 + (hp & 1) * 32 + i * 8)) * p.N + nw;
  };
  auto byte_base = [&](int hp, int ip) {   // uniform: first byte of the pair's 1 KiB run
    return lo8_offset((mw >> 6) + (hp >> 1), nw >> 6, n_cb, hp & 1, ip, 0);
  };
  auto prefetch = [&](f32x4 (&dst)[NI][2], int hp) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const size_t rb = row_base(hp, i);
      if constexpr (in_split) {
        dst[i][0] = load16_nt(p.resid_bf16 + rb + loff);
        if ((i & 1) == 0) dst[i >> 1][1] = load16_nt(p.lo_in + byte_base(hp, i >> 1) + boff);   // bytes of instructions i and i + 1
      } else {
        dst[i][0] = load16_nt(p.out_f32 + rb + loff);
        dst[i][1] = load16_nt(p.out_f32 + rb + loff + 4);
      }
    }
  };
  unsigned pk0 = 0u, pk1 = 0u;   // the bytes of an even instruction, waiting for its odd partner's
  auto finish_half = [&](const f32x4 (&hv)[NI][2], int hp) {
    const int ps = hp >> 1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rowp = (hp & 1) * 32 + i * 8 + gq;   // row inside the pass
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(stg + rowp * 256 + (((2 * l7) ^ (rowp & 15)) << 4));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(stg + rowp * 256 + (((2 * l7 + 1) ^ (rowp & 15)) << 4));
      const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rowp << 2, __builtin_bit_cast(int, off[ps])));
      // the additions and the statistics as packed fp32 (two columns per issue slot)
      f32x2 v[4] = {{a0[0], a0[1]}, {a0[2], a0[3]}, {a1[0], a1[1]}, {a1[2], a1[3]}};
      if constexpr (in_split) {
        const V8 hi = __builtin_bit_cast(V8, hv[i][0]);
        const f32x4 bytes = hv[i >> 1][1];
        const unsigned w_lo = f2u((i & 1) ? bytes[2] : bytes[0]), w_hi = f2u((i & 1) ? bytes[3] : bytes[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned ww = j < 2 ? w_lo : w_hi;
          const int sh = (j & 1) * 16;
          const f32x2 b = {(float)((ww >> sh) & 0xffu), (float)((ww >> (sh + 8)) & 0xffu)};   // v_cvt_f32_ubyteN
          const f32x2 m = pk_fma(b, splat2(kStep), splat2(1.0f - 128.0f * kStep));
          v[j] = pk_fma(f32x2{(float)hi[2 * j], (float)hi[2 * j + 1]}, m, v[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v[j] += f32x2{hv[i][0][2 * j], hv[i][0][2 * j + 1]};
          v[2 + j] += f32x2{hv[i][1][2 * j], hv[i][1][2 * j + 1]};
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += o;
      const size_t rb = row_base(hp, i);
      if constexpr (out_split) {
        V8 ho;
        unsigned w0 = 0u, w1 = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = v[j >> 1][j & 1];
          ho[j] = Op<T>::to(x);
          // (x / hi - 1) / step + 128; hi = 0 (x = 0, or an fp16 underflow): 0 * inf = NaN -> byte 0 -> decodes to hi * (...) = 0;
          // an fp16 hi clamped at +-65504 saturates the byte (and Op<T>::to has raised the saturation flag)
          const float t = fmaf(x * __builtin_amdgcn_rcpf((float)ho[j]), kInv, 128.0f - kInv);
          if (j < 4) w0 = __builtin_amdgcn_cvt_pk_u8_f32(t, (unsigned)j, w0);
          else w1 = __builtin_amdgcn_cvt_pk_u8_f32(t, (unsigned)(j - 4), w1);
        }
        store16_nt(p.resid_bf16 + rb + loff, __builtin_bit_cast(f32x4, ho));
        if (i & 1) {
          store16_nt(p.lo_out + byte_base(hp, i >> 1) + boff, f32x4{__builtin_bit_cast(float, pk0), __builtin_bit_cast(float, pk1),
                                                                    __builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1)});
        } else {
          pk0 = w0;
          pk1 = w1;
        }
        if constexpr (STATS) {
          const f32x2 t1 = (v[0] + v[1]) + (v[2] + v[3]);
          const f32x2 t2 = pk_fma(v[3], v[3], pk_fma(v[2], v[2], pk_fma(v[1], v[1], v[0] * v[0])));
          float s1 = t1[0] + t1[1], s2 = t2[0] + t2[1];
          s1 = row8_sum(s1);
          s2 = row8_sum(s2);
          if (l7 == 0) {
            // slice-major: the 8 rows of an instruction are 64 contiguous bytes (row-major, 8 bytes in each of 8 lines: as many
            // write requests per instruction as a whole high-plane store, for 1/16 of its bytes)
            float* sp = p.stats_part + ((size_t)(nw >> 6) * p.stats_ld + (mw + ps * 64 + rowp)) * 2;
            sp[0] = s1;
            sp[1] = s2;
          }
        }
      } else {
        store16_nt(p.out_f32 + rb + loff, f32x4{v[0][0], v[0][1], v[1][0], v[1][1]});
        store16_nt(p.out_f32 + rb + loff + 4, f32x4{v[2][0], v[2][1], v[3][0], v[3][1]});
      }
    }
  };
  prefetch(hA, 0);
#pragma unroll
  for (int ps = 0; ps < RT / 4; ++ps) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int row = rq * 16 + l15;
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) *reinterpret_cast<f32x4*>(stg + row * 256 + (((nj * 4 + q) ^ l15) << 4)) = acc[nj][ps * 4 + rq];
    }
    prefetch(hB, 2 * ps + 1);
    finish_half(hA, 2 * ps);
    if (ps + 1 < RT / 4) prefetch(hA, 2 * ps + 2);
    finish_half(hB, 2 * ps + 1);
  }
}

template <int RT, typename T>
__device__ __forceinline__ void residual_split_epilogue(const GemmParams& p, f32x4 (&acc)[4][RT], char* stg, int lane, int mw, int nw) {
  const bool in_split = p.lo_in != nullptr, out_split = p.lo_out != nullptr, stats = p.stats_part != nullptr;
  if (in_split && out_split) {            // layers >= 1 of the ModernBERT schedule
    if (stats) residual_split_body<RT, T, true, true, true>(p, acc, stg, lane, mw, nw);
    else residual_split_body<RT, T, true, true, false>(p, acc, stg, lane, mw, nw);
  } else if (out_split) {                 // layer 0's mlp Wo: fp32 rows in, planes out
    if (stats) residual_split_body<RT, T, false, true, true>(p, acc, stg, lane, mw, nw);
    else residual_split_body<RT, T, false, true, false>(p, acc, stg, lane, mw, nw);
  } else {                                // the last sub-layer of a run: planes in, fp32 rows out
    residual_split_body<RT, T, true, false, false>(p, acc, stg, lane, mw, nw);
  }
}

// Epilogue shared by the GEMM kernels. `acc[nj][rt]` are this wave's accumulators (swapped layout:
// lane = token row, registers = 4 consecutive features; un-swapped for the V third of QKV).
// XPREF (GeGLU, 256 x 256 throughput form): the epilogue stages through the UPPER half of the operand ring and calls `mid()`
// between its arithmetic and its stores -- the kernel issues the NEXT tile's first operand stage into the lower half there.
template <int EPI, int RT, int WROWS, typename T, bool SMALL = false, bool XPREF = false, typename Mid>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][RT], char* smem, int wave, int lane,
                                              int mw, int nw, bool v_block, Mid&& mid) {
  typedef typename Op<T>::v4 V4;   // 4 operand-type values (8 bytes)
  // The lane index is re-read through an opaque asm: every per-lane address of the epilogue then depends on a value
  // defined inside the tile loop, so none of them is hoisted out of it to sit in (spilled) registers across the K loop.
  asm volatile("" : "+v"(lane));
  const int q = lane >> 4, l15 = lane & 15;
  // ------------------------------------------------------------------ epilogues
  // All operand-tile reads are done (the loop ends with a barrier), so the LDS is reused as a
  // per-wave 16 KiB staging area: accumulators are written in their natural (row-per-lane)
  // layout with an XOR-swizzled 16-byte chunk index and read back row-contiguous, so every
  // global store/RMW instruction covers whole 128/256-byte row segments (full cache lines)
  // instead of 16 scattered 16-byte pieces.
  char* stg = smem + wave * 16384;

  if constexpr (SMALL && (EPI == EPI_QKV_ROPE || EPI == EPI_GEGLU || EPI == EPI_BF16)) {
    // Small-row configuration: finish the producer's row statistics here instead of in a launch of their own
    // (ln_stats_finalize_kernel's arithmetic, partials in the same fixed order -> the same bits).  Lane L takes row mw + L of
    // the wave's 64; the values go through ln_mu / ln_rstd in global memory (every wave that covers these rows writes the
    // same numbers) and are read back by this wave below: same CU, same L1, ordered by the vmcnt wait.
    static_assert(WROWS == 64 || WROWS == 32, "one row per lane (the first WROWS lanes)");
    if (p.stats_in) {
      const int row = mw + (WROWS == 64 ? lane : (lane & (WROWS - 1))), np = p.K >> 6;   // WROWS = 32: both halves of the wave take the same 32 rows (same values written twice)
      const float* part = p.stats_in + (size_t)row * 2;
      float s1 = 0.f, s2 = 0.f;
      // slice-major partials: [K / 64][stats_ld rows][2].  Sixteen slices are fetched per round trip and added in slice order (the
      // finalize kernel's order: same bits).  The plain loop compiled to load -> s_waitcnt vmcnt(0) -> add per slice, twelve dependent
      // round trips -- L2 hits, as it turned out: batching them moved the launch by 0.05-0.3 us (round 6).
      for (int i0 = 0; i0 < np; i0 += 16) {
        f32x2 pv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) pv[u] = *reinterpret_cast<const f32x2*>(part + (size_t)min(i0 + u, np - 1) * p.stats_ld * 2);
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (i0 + u < np) {
            s1 += pv[u][0];
            s2 += pv[u][1];
          }
      }
      const float d = s1 / (float)p.K;
      const float var = fmaxf(s2 / (float)p.K - d * d, 0.f);
      const_cast<float*>(p.ln_mu)[row] = d;
      const_cast<float*>(p.ln_rstd)[row] = 1.0f / sqrtf(var + p.fin_eps);
      if (nw == 0) {   // exactly one wave per row sits on column 0
        const float c = p.ln_shift[row];
        if (p.ln_shift_prev) p.ln_shift_prev[row] = c;
        const_cast<float*>(p.ln_shift)[row] = c + d;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }

  // 16-bit tile [R rows][C cols] (C = 64 or 32): lane writes 4 consecutive columns of its row.
  auto put_bf16 = [&](int row, int col, const V4& v, int row_bytes) {
    const int c16 = col >> 3, half = (col >> 2) & 1;
    const int sw16 = row_bytes == 128 ? (row & 7) : (row_bytes == 64 ? (row & 3) : (row & 15));
    *reinterpret_cast<V4*>(stg + row * row_bytes + ((c16 ^ sw16) << 4) + (half << 3)) = v;
  };
  // read back 16 bytes: chunk c16 of `row`
  auto get16 = [&](int row, int c16, int row_bytes) -> f32x4 {
    const int sw16 = row_bytes == 128 ? (row & 7) : (row_bytes == 64 ? (row & 3) : (row & 15));
    return *reinterpret_cast<const f32x4*>(stg + row * row_bytes + ((c16 ^ sw16) << 4));
  };

  if constexpr (EPI == EPI_NONE) {
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) asm volatile("" ::"v"(acc[nj][rt]));
  } else if constexpr (EPI == EPI_F32 || EPI == EPI_F32_GELU || EPI == EPI_RESIDUAL) {
    if constexpr (EPI == EPI_RESIDUAL) {
      if (p.lo_in || p.lo_out) {   // wave-uniform: the stream is split on at least one side (ModernBERT schedule: no bias, no post-LN rebuild)
        residual_split_epilogue<RT, T>(p, acc, stg, lane, mw, nw);
        return;
      }
    }
    // fp32 [64 rows][64 cols] per pass (256-byte rows, 16 chunks), RT/4 passes; a pass is read back in two halves of
    // 32 rows (8 row-segment instructions each).  Residual: the h rows of a half are prefetched one half ahead
    // (two 8 x 16-byte register sets, 8 KiB per wave in flight), so the fp32 read-modify-write is one HBM round trip
    // per pass that runs under the previous half's LDS reads and stores.
    constexpr int HALF = 8;
    f32x4 hA[HALF], hB[HALF];
    float cw[RT / 4];   // residual + LayerNorm fold: lane L holds the shift of row ps*64 + L (fetched by bpermute below)
    auto h_ptr = [&](int hp, int i) {   // half-pass hp = 2*ps + half, instruction i: 4 rows x 16 lanes
      const int row = (hp >> 1) * 64 + (hp & 1) * 32 + i * 4 + (lane >> 4);
      return p.out_f32 + (size_t)(mw + row) * p.N + nw + (lane & 15) * 4;
    };
    auto prefetch = [&](f32x4 (&dst)[HALF], int hp) {
#pragma unroll
      for (int i = 0; i < HALF; ++i) dst[i] = load16_nt(h_ptr(hp, i));
    };
    if constexpr (EPI == EPI_RESIDUAL) {
      prefetch(hA, 0);
#pragma unroll
      for (int ps = 0; ps < RT / 4; ++ps) cw[ps] = p.ln_shift ? p.ln_shift[mw + ps * 64 + lane] : 0.f;
    }
    auto finish_half = [&](const f32x4 (&hv)[HALF], int hp) {
      const int ps = hp >> 1;
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const int row = (hp & 1) * 32 + i * 4 + (lane >> 4), c16 = lane & 15;   // row inside the pass
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 256 + ((c16 ^ (row & 15)) << 4));
        float* dst = p.out_f32 + (size_t)(mw + ps * 64 + row) * p.N + nw + c16 * 4;
        if constexpr (EPI == EPI_RESIDUAL) {
          f32x4 hin = hv[i];
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + nw + c16 * 4);   // BERT-family linears have biases
          if (p.res_mu) {
            // post-LN encoders keep the PRE-LayerNorm sum in the stream; the residual input LN(t) is rebuilt here
            // from the row statistics instead of being written and re-read by a LayerNorm kernel
            const int grow = mw + ps * 64 + row;
            const float m_ = p.res_mu[grow], r_ = p.res_rstd[grow];
            const f32x4 g_ = *reinterpret_cast<const f32x4*>(p.res_g + nw + c16 * 4);
            const f32x4 b_ = *reinterpret_cast<const f32x4*>(p.res_b + nw + c16 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) hin[j] = (hin[j] - m_) * r_ * g_[j] + b_[j];
          }
          v += hin;
        } else if constexpr (EPI == EPI_F32) {
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + nw + c16 * 4);
        }
        store16_nt(dst, v);
        if constexpr (EPI == EPI_RESIDUAL) {
          // LayerNorm fold: the bf16 operand copy and the row statistics are taken RELATIVE to a per-row shift c (the
          // row's mean after the previous sub-layer, a close estimate of its new mean): bf16(h - c) spends its 8
          // mantissa bits on the deviation instead of on a common offset, and sum / sum of squares of (h - c) do
          // not cancel in  var = E[(h-c)^2] - (mu-c)^2  however large |mean| / sigma is.
          if (p.ln_shift) {
            // lane group g = lane >> 4 works on row r0 + g of the pass: four scalar lane reads and a select by group
            // (a ds_bpermute would be one more trip through the LDS crossbar per instruction)
            const int r0 = (hp & 1) * 32 + i * 4;
            const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cw[ps]), r0));
            const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cw[ps]), r0 + 1));
            const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cw[ps]), r0 + 2));
            const float c3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cw[ps]), r0 + 3));
            const int g4 = lane >> 4;
            const float c_ = g4 == 0 ? c0 : (g4 == 1 ? c1 : (g4 == 2 ? c2 : c3));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] -= c_;
          }
          if (p.resid_bf16) {
            V4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = Op<T>::to(v[j]);
            store8_nt(p.resid_bf16 + (size_t)(mw + ps * 64 + row) * p.N + nw + c16 * 4, o);
          }
          if (p.stats_part) {
            // the 16 lanes of a row segment reduce (sum, sum of squares) of the UPDATED residual (minus the shift)
            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
            float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            s1 = row16_sum(s1);
            s2 = row16_sum(s2);
            if (c16 == 0) {
              float* sp = p.stats_part + ((size_t)(nw >> 6) * p.stats_ld + (mw + ps * 64 + row)) * 2;
              sp[0] = s1;
              sp[1] = s2;
            }
          }
        }
      }
    };
#pragma unroll
    for (int ps = 0; ps < RT / 4; ++ps) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int row = rq * 16 + l15;   // row inside the pass
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          f32x4 v = acc[nj][ps * 4 + rq];
          if constexpr (EPI == EPI_F32_GELU) {
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + nw + nj * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
          }
          const int c16 = nj * 4 + q;
          *reinterpret_cast<f32x4*>(stg + row * 256 + ((c16 ^ l15) << 4)) = v;
        }
      }
      // half 0 of this pass came in under the staging writes; fetch half 1 now, and the next pass's half 0 under half 1
      if constexpr (EPI == EPI_RESIDUAL) prefetch(hB, 2 * ps + 1);
      finish_half(hA, 2 * ps);
      if constexpr (EPI == EPI_RESIDUAL) {
        if (ps + 1 < RT / 4) prefetch(hA, 2 * ps + 2);
      }
      finish_half(hB, 2 * ps + 1);
    }
  } else if constexpr (EPI == EPI_BF16) {
    // The optional terms (LayerNorm fold, bias, GELU) are wave-uniform: one dispatch per tile into a body compiled for
    // the combination, so that the per-row / per-column loads of a tile are independent of any branch and batch up.
    auto body = [&](auto fold_tag, auto bias_tag, auto gelu_tag) {
      constexpr bool FOLD = decltype(fold_tag)::value, BIAS = decltype(bias_tag)::value, GELU = decltype(gelu_tag)::value;
      f32x4 ls[4] = {}, bs[4] = {};
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        if constexpr (FOLD) ls[nj] = *reinterpret_cast<const f32x4*>(p.ln_s + nw + nj * 16 + 4 * q);
        if constexpr (BIAS) bs[nj] = *reinterpret_cast<const f32x4*>(p.bias + nw + nj * 16 + 4 * q);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float mu = 0.f, rs = 1.f;
        if constexpr (FOLD) {
          mu = p.ln_mu[mw + rt * 16 + l15];
          rs = p.ln_rstd[mw + rt * 16 + l15];
        }
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          V4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[nj][rt][j];
            if constexpr (FOLD) v = rs * (v - mu * ls[nj][j]);
            if constexpr (BIAS) v += bs[nj][j];
            if constexpr (GELU) v = gelu_fast(v);   // BERT-family MLP: gelu(x W1^T + b1)
            o[j] = Op<T>::to(v);
          }
          put_bf16(rt * 16 + l15, nj * 16 + 4 * q, o, 128);
        }
      }
    };
    typedef std::true_type Y;
    typedef std::false_type N_;
    if (p.act_gelu) {                              // BERT-family MLP up-projection (always biased)
      if (p.ln_mu) body(Y{}, Y{}, Y{});
      else body(N_{}, Y{}, Y{});
    } else if (p.ln_mu) {
      if (p.bias) body(Y{}, Y{}, N_{});
      else body(Y{}, N_{}, N_{});
    } else {
      if (p.bias) body(N_{}, Y{}, N_{});
      else body(N_{}, N_{}, N_{});
    }
#pragma unroll
    for (int it = 0; it < WROWS / 8; ++it) {
      const int row = it * 8 + (lane >> 3), c16 = lane & 7;
      store16_nt(p.out_bf16 + (size_t)(mw + row) * p.N + nw + c16 * 8, get16(row, c16, 128));
    }
  } else if constexpr (EPI == EPI_GEGLU) {
    // Wi rows were interleaved at load time: each 64-row group = 32 "input" rows (x1)
    // followed by the 32 matching "gate" rows (x2).  A wave's output is [rows][32 features] = 64-byte rows, i.e. HALF cache
    // lines: stored per wave, every line of the output would be touched by two waves.  So the two waves of a column pair
    // (wn, wn^1) stage into ONE tile of [rows][64 features] (128-byte rows, 16-byte chunk index XOR (row & 7)) and each
    // stores one half of its ROWS as whole lines (257 -> 251 us per launch; the epilogue itself is VALU-bound, ~16
    // instructions per output).
    const int NO = p.N >> 1;
    const int odd = wave & 1;                               // WN is even: wave parity == column parity of the pair
    char* stgp = XPREF ? smem + 65536 + (wave >> 1) * 16384  // the pair's staging tile: upper half of the ring (the lower takes the next tile's stage 0)
                       : smem + (wave & ~1) * 16384;         // (16 KiB of the pair's 32)
    auto put_pair = [&](int row, int col, const V4& v) {    // col = output feature 0..31 of this wave
      const int c16 = odd * 4 + (col >> 3), half = (col >> 2) & 1;
      *reinterpret_cast<V4*>(stgp + row * 128 + ((c16 ^ (row & 7)) << 4) + (half << 3)) = v;
    };
    auto body = [&](auto fold_tag) {
      constexpr bool FOLD = decltype(fold_tag)::value;
      f32x4 s1[2] = {}, s2[2] = {};
#pragma unroll
      for (int nj = 0; nj < 2; ++nj) {
        if constexpr (FOLD) {
          s1[nj] = *reinterpret_cast<const f32x4*>(p.ln_s + nw + nj * 16 + 4 * q);
          s2[nj] = *reinterpret_cast<const f32x4*>(p.ln_s + nw + 32 + nj * 16 + 4 * q);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float mu = 0.f, rs = 1.f;
        if constexpr (FOLD) {
          mu = p.ln_mu[mw + rt * 16 + l15];
          rs = p.ln_rstd[mw + rt * 16 + l15];
        }
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          V4 o;
#pragma unroll
          for (int h = 0; h < 2; ++h) {   // two outputs at a time: packed fp32 (gelu_fast2, common.h)
            f32x2 x1 = {acc[nj][rt][2 * h], acc[nj][rt][2 * h + 1]}, x2 = {acc[nj + 2][rt][2 * h], acc[nj + 2][rt][2 * h + 1]};
            if constexpr (FOLD) {
              x1 = pk_fma(splat2(-mu), f32x2{s1[nj][2 * h], s1[nj][2 * h + 1]}, x1) * rs;
              x2 = pk_fma(splat2(-mu), f32x2{s2[nj][2 * h], s2[nj][2 * h + 1]}, x2) * rs;
            }
            const f32x2 g = gelu_fast2(x1) * x2;
            o[2 * h] = Op<T>::to(g[0]);
            o[2 * h + 1] = Op<T>::to(g[1]);
          }
          put_pair(rt * 16 + l15, nj * 16 + 4 * q, o);
        }
      }
    };
    if (p.ln_mu) body(std::true_type{});
    else body(std::false_type{});
    if constexpr (XPREF) {
      mid();   // LDS-DMA of the next tile's first stage: must stay in flight across the barrier, so no __syncthreads (its fence drains vmcnt)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      __syncthreads();   // the partner wave's half of every row is staged (workgroup-uniform path: every wave is here)
    }
    const int f0 = ((nw - odd * 64) >> 6) * 32;   // first output feature of the pair
#pragma unroll
    for (int it = 0; it < WROWS / 16; ++it) {
      const int row = odd * (WROWS / 2) + it * 8 + (lane >> 3), c16 = lane & 7;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stgp + row * 128 + ((c16 ^ (row & 7)) << 4));
      store16_nt(p.out_bf16 + (size_t)(mw + row) * NO + f0 + c16 * 8, v);
    }
  } else if constexpr (EPI == EPI_QKV_ROPE) {
    const int H = p.hidden;
    const int which = nw / H;  // 0 = q, 1 = k, 2 = v   (wave-uniform)
    const int head = (nw - which * H) >> 6;
    typedef std::true_type Y;
    typedef std::false_type N_;
    if (!v_block) {
      bf16_t* dst = which == 0 ? p.q : p.k;
      const float scale = which == 0 ? p.q_scale : 1.0f;
      auto body = [&](auto fold_tag, auto bias_tag) {
        constexpr bool FOLD = decltype(fold_tag)::value, BIAS = decltype(bias_tag)::value;
        f32x4 ls1[2] = {}, ls2[2] = {}, b1[2] = {}, b2[2] = {};
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const int dd = nj * 16 + 4 * q;
          if constexpr (FOLD) {
            ls1[nj] = *reinterpret_cast<const f32x4*>(p.ln_s + nw + dd);
            ls2[nj] = *reinterpret_cast<const f32x4*>(p.ln_s + nw + 32 + dd);
          }
          if constexpr (BIAS) {
            b1[nj] = *reinterpret_cast<const f32x4*>(p.bias + nw + dd);
            b2[nj] = *reinterpret_cast<const f32x4*>(p.bias + nw + 32 + dd);
          }
        }
        // rotary tables are gathered per token row (L2 hits): two row tiles per group, the next group's gathers issued
        // before this group's arithmetic (two register sets), scheduling fenced per group so the set stays at 64 registers
        constexpr int G = 2, NG = RT / G;
        f32x4 cz[2][G][2], sz[2][G][2];
        float mu_[2][G], rs_[2][G];
        auto gather = [&](int set, int g) {
#pragma unroll
          for (int i = 0; i < G; ++i) {
            const int row = mw + (g * G + i) * 16 + l15;
            const int pos = p.pos[row];
            mu_[set][i] = 0.f;
            rs_[set][i] = 1.f;
            if constexpr (FOLD) {
              mu_[set][i] = p.ln_mu[row];
              rs_[set][i] = p.ln_rstd[row];
            }
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
              cz[set][i][nj] = *reinterpret_cast<const f32x4*>(p.rope_cos + (size_t)pos * 32 + nj * 16 + 4 * q);
              sz[set][i][nj] = *reinterpret_cast<const f32x4*>(p.rope_sin + (size_t)pos * 32 + nj * 16 + 4 * q);
            }
          }
        };
        gather(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) gather((g + 1) & 1, g + 1);
#pragma unroll
          for (int i = 0; i < G; ++i) {
            const int rt = g * G + i;
            const float mu = mu_[g & 1][i], rs = rs_[g & 1][i];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
              const int dd = nj * 16 + 4 * q;
              const f32x4 c = cz[g & 1][i][nj], sv = sz[g & 1][i][nj];
              V4 o1, o2;
#pragma unroll
              for (int h = 0; h < 2; ++h) {   // two features at a time as packed fp32 (see gelu_fast2, common.h)
                const int j0 = 2 * h;
                f32x2 x1 = {acc[nj][rt][j0], acc[nj][rt][j0 + 1]}, x2 = {acc[nj + 2][rt][j0], acc[nj + 2][rt][j0 + 1]};
                if constexpr (FOLD) {
                  x1 = pk_fma(splat2(-mu), f32x2{ls1[nj][j0], ls1[nj][j0 + 1]}, x1) * rs;
                  x2 = pk_fma(splat2(-mu), f32x2{ls2[nj][j0], ls2[nj][j0 + 1]}, x2) * rs;
                }
                if constexpr (BIAS) {
                  x1 += f32x2{b1[nj][j0], b1[nj][j0 + 1]};
                  x2 += f32x2{b2[nj][j0], b2[nj][j0 + 1]};
                }
                const f32x2 c2 = {c[j0], c[j0 + 1]}, s2 = {sv[j0], sv[j0 + 1]};
                // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)  (TF:188-219)
                const f32x2 r1 = pk_fma(-x2, s2, x1 * c2) * scale, r2 = pk_fma(x1, s2, x2 * c2) * scale;
                o1[j0] = Op<T>::to(r1[0]);
                o1[j0 + 1] = Op<T>::to(r1[1]);
                o2[j0] = Op<T>::to(r2[0]);
                o2[j0 + 1] = Op<T>::to(r2[1]);
              }
              put_bf16(rt * 16 + l15, dd, o1, 128);
              put_bf16(rt * 16 + l15, dd + 32, o2, 128);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (p.ln_mu) {
        if (p.bias) body(Y{}, Y{});
        else body(Y{}, N_{});
      } else {
        if (p.bias) body(N_{}, Y{});
        else body(N_{}, N_{});
      }
#pragma unroll
      for (int it = 0; it < WROWS / 8; ++it) {
        const int row = it * 8 + (lane >> 3), c16 = lane & 7;
        store16_nt(dst + (size_t)(mw + row) * H + head * 64 + c16 * 8, get16(row, c16, 128));
      }
    } else {
      // un-swapped accumulators: lane = feature d (nj*16 + l15), registers = tokens
      //   token = rt*16 + 4*q + r.   Stage V^T tile [64 d][WROWS tokens].
      constexpr int RB = WROWS * 2;  // row bytes (256 for 128 tokens, 128 for 64)
      auto body = [&](auto fold_tag, auto bias_tag) {
        constexpr bool FOLD = decltype(fold_tag)::value, BIAS = decltype(bias_tag)::value;
        f32x4 mu4[RT] = {}, rs4[RT] = {};
        if constexpr (FOLD) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            mu4[rt] = *reinterpret_cast<const f32x4*>(p.ln_mu + mw + rt * 16 + 4 * q);
            rs4[rt] = *reinterpret_cast<const f32x4*>(p.ln_rstd + mw + rt * 16 + 4 * q);
          }
        }
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          float sn_ = 0.f, bv_ = 0.f;
          if constexpr (FOLD) sn_ = p.ln_s[nw + nj * 16 + l15];
          if constexpr (BIAS) bv_ = p.bias[nw + nj * 16 + l15];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            V4 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // packed fp32, two tokens at a time
              const int j0 = 2 * h;
              f32x2 v = {acc[nj][rt][j0], acc[nj][rt][j0 + 1]};
              if constexpr (FOLD) v = pk_fma(f32x2{-mu4[rt][j0], -mu4[rt][j0 + 1]}, splat2(sn_), v) * f32x2{rs4[rt][j0], rs4[rt][j0 + 1]};
              if constexpr (BIAS) v += bv_;
              o[j0] = Op<T>::to(v[0]);
              o[j0 + 1] = Op<T>::to(v[1]);
            }
            put_bf16(nj * 16 + l15, rt * 16 + 4 * q, o, RB);
          }
        }
      };
      if (p.ln_mu) {
        if (p.bias) body(Y{}, Y{});
        else body(Y{}, N_{});
      } else {
        if (p.bias) body(N_{}, Y{});
        else body(N_{}, N_{});
      }
      constexpr int LPR = RB / 16;        // lanes per row
      constexpr int RPI = 64 / LPR;       // rows per instruction
#pragma unroll
      for (int it = 0; it < 64 / RPI; ++it) {
        const int row = it * RPI + lane / LPR, c16 = lane % LPR;
        store16_nt(p.vt + (size_t)(head * 64 + row) * p.vt_ld + mw + c16 * 8, get16(row, c16, RB));
      }
    }
  } else if constexpr (EPI == EPI_SPLADE) {
    // rows[seq][n] = max over the tokens of `seq` of log1p(relu(logit + bias[n])).  log1p(relu(. + b)) is
    // monotone non-decreasing, so the max is taken over the raw accumulators first and the transcendental is
    // evaluated once per (sequence, column) -- bit-identical to transforming every element.  A wave's
    // WROWS token rows are a few contiguous runs of sequences: one pass per distinct sequence id.
    int sq[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) sq[rt] = p.tok_seq[mw + rt * 16 + l15];
    int done_below = 0;   // sequence ids < done_below are finished (ids are non-negative and ascending along rows)
    while (true) {
      int m = 0x7fffffff;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        if (sq[rt] >= done_below) m = min(m, sq[rt]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o, 64));
      const int cur = uniform(m);
      if (cur == 0x7fffffff) break;
      bool mine[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) mine[rt] = sq[rt] == cur;
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = -INFINITY;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) v = mine[rt] ? fmaxf(v, acc[nj][rt][r]) : v;
          v = row16_max(v);   // over the 16 token lanes of this quarter
          if (l15 == 0) {
            const int n = nw + nj * 16 + 4 * q + r;
            const float w = log1pf(fmaxf(v + (p.bias ? p.bias[n] : 0.f), 0.f));
            if (w > 0.f) atomicMax(p.splade_rows + (size_t)cur * p.N + n, f2u(w));
          }
        }
      done_below = cur + 1;
    }
  } else if constexpr (EPI == EPI_TOPK) {
    // Lane = corpus row (per row tile rt), registers = 4 consecutive query columns per column tile nj.  After the first
    // rows of a shard almost nothing passes the entry threshold: the common case is 128 compares per lane and one branch.
    f32x4 thr[4];
    if (p.topk_pairs) {
      // columns (2j, 2j + 1) = (value, remainder) of query (n >> 1): both columns of a pair hold the pair's threshold
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int qa = (nw + nj * 16 + 4 * q) >> 1;
        const float t0 = qa < p.topk_nq ? p.topk_thr_score[qa] : INFINITY, t1 = qa + 1 < p.topk_nq ? p.topk_thr_score[qa + 1] : INFINITY;
        thr[nj] = f32x4{t0, t0, t1, t1};
      }
    } else {
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const int n = nw + nj * 16 + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) thr[nj][r] = n + r < p.topk_nq ? p.topk_thr_score[n + r] : INFINITY;
      }
    }
    bool any = false;
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        if (p.topk_pairs) {
          any |= (acc[nj][rt][0] + acc[nj][rt][1] >= thr[nj][0]) | (acc[nj][rt][2] + acc[nj][rt][3] >= thr[nj][2]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) any |= (acc[nj][rt][r] >= thr[nj][r]);
        }
      }
    if (p.topk_direct) {
      // first stage: slot = row, no counters.  EVERY slot of a live row is written -- its key, or "no key" for a score that fails
      // the threshold test (the stage runs with -inf: a NaN) -- so the buffer needs no clearing in front of the launch
      // one 64-bit slot address per (column tile, column) at a time, the row tiles at constant offsets from it: all sixteen
      // addresses live at once cost the 256 x 256 form two spilled registers (tests/test_kernel_resources.py)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (p.topk_pairs && (r & 1)) continue;
          const int n = nw + nj * 16 + 4 * q + r;
          const int query = p.topk_pairs ? n >> 1 : n;
          if (query >= p.topk_nq) continue;
          unsigned long long* slot = p.topk_buf + (size_t)query * p.topk_cap + (mw + l15);
          asm volatile("" : "+v"(slot));
          // sampled stage: the key carries the corpus row, rows of launch tile t sit (stride - 1) * 256 * t further on
          const unsigned row_bias = p.topk_tile_stride > 1 ? (unsigned)(mw >> 8) * (unsigned)(p.topk_tile_stride - 1) * 256u : 0u;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int row = mw + rt * 16 + l15;
            if (row >= p.M) continue;   // padding rows of the last tile hold whatever the allocation does
            const float sc = p.topk_pairs ? acc[nj][rt][r] + acc[nj][rt][r + 1] : acc[nj][rt][r];
            slot[rt * 16] = sc >= thr[nj][r] ? make_key(sc, p.topk_row_base + row_bias + (unsigned)row) : 0ull;
          }
        }
    } else if (any) {
      // Three straight-line passes instead of one branch per accumulator: a branch that holds an atomic with a returned slot and a
      // store is a memory round trip per branch a wave enters, and with k (ratio - 1) candidates per query and stage a wave entered
      // tens of its 128 branches one after the other.  (a) per (lane, query column): rows above the threshold score counted, ONE
      // atomic reserves their slots -- all of a lane's atomics are in flight together; per column tile: (b) the threshold keys
      // of its columns, (c) the keys go out; a reserved slot whose key fails the key test (equal score, higher row) is written as "no key".
      unsigned base[16];
      unsigned have = 0u;
      unsigned skip_bias = 0u;   // appending stage behind a sampled first stage: rows of launch tile t are corpus rows (see GemmParams::topk_tile_skip)
      if (p.topk_tile_skip > 1) {
        const int d = p.topk_tile0 + (mw >> 8), s1 = p.topk_tile_skip - 1;
        skip_bias = (unsigned)((d < 256 * s1 ? d + d / s1 + 1 : d + 256) - (mw >> 8)) * 256u;
      }
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nj * 4 + r;
          base[j] = 0u;
          if (p.topk_pairs && (r & 1)) continue;
          unsigned c = 0u;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const float sc = p.topk_pairs ? acc[nj][rt][r] + acc[nj][rt][(r + 1) & 3] : acc[nj][rt][r];
            c += (mw + rt * 16 + l15 < p.M && sc >= thr[nj][r]) ? 1u : 0u;
          }
          const int n = nw + nj * 16 + 4 * q + r, query = p.topk_pairs ? n >> 1 : n;
          if (c > 0u && query < p.topk_nq) {
            base[j] = atomicAdd(p.topk_cnt + query, c);
            have |= 1u << j;
          }
        }
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        if (!((have >> (nj * 4)) & 15u)) continue;
        unsigned long long tk[4];   // per column tile: four keys in flight, 8 registers (all 16 at once cost scratch at 256 x 256)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nw + nj * 16 + 4 * q + r, query = p.topk_pairs ? n >> 1 : n;
          tk[r] = (have >> (nj * 4 + r)) & 1u ? p.topk_thr_key[query] : ~0ull;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nj * 4 + r;
          if (!((have >> j) & 1u)) continue;
          const int n = nw + nj * 16 + 4 * q + r, query = p.topk_pairs ? n >> 1 : n;
          unsigned slot = base[j];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const int row = mw + rt * 16 + l15;
            const float sc = p.topk_pairs ? acc[nj][rt][r] + acc[nj][rt][(r + 1) & 3] : acc[nj][rt][r];
            if (row < p.M && sc >= thr[nj][r]) {
              const unsigned long long key = make_key(sc, p.topk_row_base + skip_bias + (unsigned)row);
              if (slot < (unsigned)p.topk_cap) p.topk_buf[(size_t)query * p.topk_cap + slot] = key > tk[r] ? key : 0ull;
              ++slot;
            }
          }
        }
      }
    }
  }
}

// BM x BN tile, WM x WN waves; every wave owns (BM/WM) x 64 outputs (RT = BM/WM/16 row tiles, 4 column tiles).
// NS = LDS stages.  2: the throughput configuration (one stage in flight, plain barriers).  4: the
// small-batch configuration -- with a handful of tiles the K loop is a chain of memory round trips, and three
// stages in flight (counted vmcnt, raw barriers) cut that chain to a third.
// HW = K-split waves (round 4, the launch-bound configurations): HW / (WM x WN) further GROUPS of WM x WN waves that compute the
// SAME tile over OTHER K-steps.  At this size a launch is a chain of latencies, and the chain is a wave's own K-step (fragment
// reads -> MFMAs, ~0.55 us, whatever the ring depth: profiles/r04_small_gemm_probes.txt).  With KG groups the ring holds 2 KG
// stages, a super-step holds KG landed K-steps, group g computes K-step KG j + g of super-step j, and the groups' partial tiles
// meet in LDS in group order (group 0 adds groups 1, 2, ...) before group 0 runs the epilogue.  Every wave takes its share of the
// DMA.  64 x 64 tiles: four single-wave groups on an eight-stage ring; 128 x 128 tiles: two groups of 2 x 2 waves on four stages.
// KCH (round 6; the launch-bound residual configurations WITHOUT a K-split): the K-steps are visited chain by chain -- first
// k-steps 0, 4, 8, ..., then 1, 5, 9, ..., ... -- each chain accumulated from zero and the four chain sums added in chain order,
// i.e. exactly the fp32 operations, in exactly the order, of the four-wave K-split (group g = chain g, group 0 adds 1, 2, 3).
// A sequence's bits then no longer depend on which launch-bound configuration its batch happened to take (VERDICT r5 item 3:
// a lone chunk took the K-split, the same chunk inside a few thousand rows the one-wave tiles, and the two summed in different
// orders).  Costs one more accumulator set (the running sum): these forms take one wave per SIMD, so the registers are there.
template <int EPI, int BM, int BN, int WM, int WN, int NS = 2, typename T = bf16_t, int HW = 0, bool KCH = false>
__global__ __launch_bounds__((WM * WN + HW) * 64, KCH ? 1 : 2) void gemm_bf16_kernel(const GemmParams p) {
  typedef typename Op<T>::v8 V8;   // one MFMA operand fragment (8 operand-type values, 16 bytes)
  static_assert(BN == WN * 64, "a wave spans exactly 64 output features (one head / one GeGLU group)");
  static_assert(NS >= 2 && NS <= 8, "LDS stages");
  constexpr int NWAVE = WM * WN + HW;       // waves that share the operand DMA
  constexpr int ROWB = BK * 2;                // bytes per LDS row
  constexpr int RPI = 1024 / ROWB;            // tile rows filled by one LDS-DMA instruction (64 lanes x 16 bytes)
  constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
  constexpr int KS = BK / 32;                 // MFMA k-substeps per stage
  constexpr int RT = BM / WM / 16;            // 16-row accumulator tiles per wave
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / RPI / NWAVE; // LDS-DMA instructions per wave per stage
  constexpr int W_INSTR = BN / RPI / NWAVE;
  static_assert(A_INSTR * RPI * NWAVE == BM && W_INSTR * RPI * NWAVE == BN, "tile rows must split over the waves");
  static_assert((NS - 1) * (A_INSTR + W_INSTR) <= 63, "a wave's vmcnt counts at most 63 DMA instructions in flight");
  static_assert(NS * STAGE_BYTES >= WM * WN * 16384 || EPI == EPI_NONE, "the epilogues stage 16 KiB per wave through the operand ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  constexpr int KG = 1 + HW / (WM * WN);     // K-groups
  static_assert(HW % (WM * WN) == 0 && (HW == 0 || NS == 2 * KG), "K-split: whole groups of waves, a ring of two super-steps");
  const int kg = wave / (WM * WN), wl = wave % (WM * WN);   // K-group of this wave, its place in the group
  const int wm = wl / WN, wn = wl % WN;
  const int q = lane >> 4, l15 = lane & 15;
  constexpr int WROWS = BM / WM;              // rows of the A tile owned by one wave

  // Persistent tile loop: the grid is at most one workgroup per CU (p.n_tiles tiles in total).  Tiles
  // are dealt so that each XCD (workgroup w runs on XCD w % 8) walks a contiguous range of logical
  // tiles, n fastest: the column tiles of an A row panel run together on one XCD and share its L2.
  // Staying resident lets a tile's epilogue stores drain while the next tile's operands stream in.
  const int nbn = p.N / BN;
  const int K = p.K;
  const int n_tiles = p.n_tiles;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int per_xcd_wgs = (gridDim.x + 7 - xcd) >> 3;  // workgroups of this grid that sit on my XCD
  const int tq = n_tiles >> 3, tr = n_tiles & 7;
  const int range_lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int range_len = xcd < tr ? tq + 1 : tq;
  // Cross-tile prefetch (round 6, GeGLU throughput form): the epilogue needs half of the ring, so the next tile's first operand
  // stage is issued from inside it and lands under its stores and the end-of-tile barrier instead of after them (a tile used to
  // start with: wait for the store acknowledgements, THEN issue stage 0 and wait for its round trip).
  constexpr bool XPREF = EPI == EPI_GEGLU && NS == 2 && HW == 0 && BM == 256 && BN == 256;
  bool stage0_in_flight = false;   // workgroup-uniform
  for (int tix = slot; tix < range_len; tix += per_xcd_wgs) {
  const int b = range_lo + tix;
  const int m0 = (b / nbn) * BM, n0 = (b % nbn) * BN;
  const bf16_t* __restrict__ Ab = p.A + (size_t)m0 * K;
  if constexpr (EPI == EPI_TOPK && BM == 256) {   // sampled first stage: tile t of the launch = corpus tile t * stride
    if (p.topk_tile_stride > 1) Ab = p.A + (size_t)(b / nbn) * p.topk_tile_stride * BM * K;
    else if (p.topk_tile_skip > 1) {   // the tiles the sample left: dense tile d -> corpus tile
      const int d = p.topk_tile0 + b / nbn, s1 = p.topk_tile_skip - 1;
      Ab = p.A + (size_t)(d < 256 * s1 ? d + d / s1 + 1 : d + 256) * BM * K;
    }
  }
  const bf16_t* __restrict__ Wb = p.W + (size_t)n0 * K;

  // LDS-DMA staging: instruction i of this wave fills tile rows wave*RPI*INSTR + i*RPI .. +RPI.
  // Throughput residual form: the lane index is re-read through an opaque asm per tile, so the ten lane-dependent offsets below
  // are recomputed per tile (a dozen instructions) instead of being hoisted out of the tile loop as invariants -- hoisted they
  // are live across the epilogue, which has no register to spare, and were kept in scratch (round 6: 56-108 B -> see
  // tests/test_kernel_resources.py).
  int lane_t = lane;
  if constexpr ((EPI == EPI_RESIDUAL || EPI == EPI_QKV_ROPE) && BM == 256 && NS == 2) asm volatile("" : "+v"(lane_t));
  auto swz = [](int row) { return (row >> 1) & 7; };
  int soffA[A_INSTR], soffW[W_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = wave * (RPI * A_INSTR) + i * RPI + lane_t / CPR;
    soffA[i] = row * K + (((lane_t % CPR) ^ swz(row)) << 3);
  }
#pragma unroll
  for (int i = 0; i < W_INSTR; ++i) {
    const int row = wave * (RPI * W_INSTR) + i * RPI + lane_t / CPR;
    soffW[i] = row * K + (((lane_t % CPR) ^ swz(row)) << 3);
  }
  // fragment read offsets (bytes inside a 16-row sub-tile): lane = row l15, k-values 32 s + 8 q .. + 8
  const int l15t = lane_t & 15, qt = lane_t >> 4;
  const int sw = swz(l15t);
  int fo[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) fo[s] = l15t * ROWB + ((((4 * s + qt) ^ sw)) << 4);

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE_BYTES;
    char* sW = sA + A_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) glds16(Ab + soffA[i] + k0, sA + (wave * (RPI * A_INSTR) + i * RPI) * ROWB);
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) glds16(Wb + soffW[i] + k0, sW + (wave * (RPI * W_INSTR) + i * RPI) * ROWB);
  };
  auto stage0_of = [&](const bf16_t* A_, const bf16_t* W_) {   // XPREF only: stage 0 of ANOTHER tile into slot 0
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) glds16(A_ + soffA[i], smem + (wave * (RPI * A_INSTR) + i * RPI) * ROWB);
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) glds16(W_ + soffW[i], smem + A_BYTES + (wave * (RPI * W_INSTR) + i * RPI) * ROWB);
  };

  f32x4 acc[4][RT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < RT; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nw = n0 + wn * 64;      // first feature of this wave's 64-wide range
  const int mw = m0 + wm * WROWS;   // first token row of this wave
  const int KT = K / BK;
  // The V third of the QKV product is computed un-swapped (activations as the A operand): then a
  // lane holds 4 consecutive TOKENS of one feature, which is the V^T row layout attention wants.
  bool v_block = false;
  if constexpr (EPI == EPI_QKV_ROPE) v_block = n0 >= 2 * p.hidden;  // workgroup-uniform (BN divides hidden)

  auto mainloop = [&](auto swapped_tag) {
    constexpr bool SWAPPED = decltype(swapped_tag)::value;
    constexpr int NDMA = A_INSTR + W_INSTR;   // LDS-DMA instructions per wave per stage
    auto wait_allow = [&](int stages_in_flight) {   // this wave's DMA is retired except the newest `stages_in_flight` stages
      if constexpr (NS > 4) {
        switch (stages_in_flight) {
          case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NDMA) : "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * NDMA) : "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NDMA) : "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory"); break;
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      } else if (NS > 2 && stages_in_flight >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
      else if (NS > 2 && stages_in_flight == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto step_barrier = [&]() {
      if constexpr (NS == 2) {
        __syncthreads();
      } else {   // DMA of later stages stays in flight across the barrier: no implicit vmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    // one K-step's MFMAs on LDS slot `buf`
    auto compute = [&](int buf) {
      const char* sA = smem + buf * STAGE_BYTES + (wm * WROWS) * ROWB;
      const char* sW = smem + buf * STAGE_BYTES + A_BYTES + (wn * 64) * ROWB;
      // MFMA order: the KS k-substeps of one accumulator tile back to back (a dependent pair), row tile by row tile -- measured
      // 4-5 % faster on the main loop than one substep over all 32 tiles and then the next (201.6 vs 210.3 us at N = 2304,
      // 67.1 vs 70.1, 101.1 vs 107.3: tools/gpu_r2ae.sh).  The W fragments of the whole K-step stay in registers (KS x 4), the
      // A fragments stream per row tile (KS at a time): 12 fragments live, as before.
      V8 wf[KS][4] = {};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[s][i] = *reinterpret_cast<const V8*>(sW + i * 16 * ROWB + fo[s]);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        V8 af[KS] = {};
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = *reinterpret_cast<const V8*>(sA + rt * 16 * ROWB + fo[s]);
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            if constexpr (SWAPPED)
              acc[nj][rt] = Op<T>::mfma16(wf[s][nj], af[s], acc[nj][rt]);
            else
              acc[nj][rt] = Op<T>::mfma16(af[s], wf[s][nj], acc[nj][rt]);
          }
      }
    };
    if constexpr (HW > 0) {
      // K-split: super-step j = K-steps G j .. G j + G - 1, all landed before its barrier; wave w computes K-step G j + w.
      constexpr int G = KG;
      int issued = min(NS - 1, KT);
#pragma unroll
      for (int i = 0; i < NS - 1; ++i)
        if (i < KT) stage(i, i);
      for (int first = 0; first < KT; first += G) {
        const int last = min(KT, first + G) - 1;
        wait_allow(issued - 1 - last);   // this wave's share of K-steps first .. last has landed; later ones stay in flight
        step_barrier();                  // ... and every other wave's
        if (first + kg <= last) compute((first + kg) % NS);
        step_barrier();                  // all reads of this super-step's slots are done: they are refilled now
#pragma unroll
        for (int i = 0; i < G; ++i)
          if (issued < KT) {
            stage(issued, issued % NS);
            ++issued;
          }
      }
      // the partial tiles meet in LDS (the ring is free): the waves of groups 1 .. store theirs lane-linear, the wave of group 0 in
      // the same place of the tile adds them in group order
      constexpr int PART = RT * 4096;   // bytes of one wave's accumulators
      if (kg > 0) {
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            *reinterpret_cast<f32x4*>(smem + ((kg - 1) * (WM * WN) + wl) * PART + ((nj * RT + rt) * 64 + lane) * 16) = acc[nj][rt];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int g = 0; g < KG - 1; ++g)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[nj][rt] += *reinterpret_cast<const f32x4*>(smem + (g * (WM * WN) + wl) * PART + ((nj * RT + rt) * 64 + lane) * 16);
      }
      __syncthreads();   // group 0's epilogue stages through the same bytes
      return;
    }
    if constexpr (KCH) {
      static_assert(HW == 0 && NS > 2, "chain order: the multi-stage launch-bound forms");
      struct Cursor {   // visiting order of the K-steps: chain g = k-steps g, g + 4, ...
        int g, kt;
        __device__ __forceinline__ void next(int KT_) {
          kt += 4;
          while (kt >= KT_ && g < 4) {
            ++g;
            kt = g;
          }
        }
      };
      Cursor cs{0, 0}, cc{0, 0};
      f32x4 sum[4][RT];
      int issued = 0;
#pragma unroll
      for (int i = 0; i < NS - 1; ++i)
        if (issued < KT) {
          stage(cs.kt, i);
          cs.next(KT);
          ++issued;
        }
      wait_allow(min(NS - 1, KT) - 1);
      step_barrier();
      int buf = 0;
      for (int i = 0; i < KT; ++i) {
        const int nbuf = buf == 0 ? NS - 1 : buf - 1;
        if (issued < KT) {
          stage(cs.kt, nbuf);
          cs.next(KT);
          ++issued;
        }
        compute(buf);
        const int g = cc.g;
        cc.next(KT);
        if (i + 1 == KT || cc.g != g) {   // chain g is complete (wave-uniform)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < RT; ++c) {
              sum[a][c] = g == 0 ? acc[a][c] : sum[a][c] + acc[a][c];
              acc[a][c] = i + 1 == KT ? sum[a][c] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        wait_allow(max(0, min(NS - 2, KT - 2 - i)));
        step_barrier();
        buf = buf + 1 == NS ? 0 : buf + 1;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
      if (i < KT && !(XPREF && i == 0 && stage0_in_flight)) stage(i, i);
    bool counted = false;
    if constexpr (XPREF) counted = stage0_in_flight;
    if (counted) {
      // stage 0 was issued from inside the previous tile's epilogue, ahead of that epilogue's WROWS / 16 output stores (the only
      // vector-memory operations behind it): a counted wait retires the DMA and leaves the stores in flight -- their
      // acknowledgements are not on this tile's critical path (loads and stores retire in order on this counter: the compiler's
      // own counted waits in the residual epilogue rely on the same)
      static_assert(!XPREF || WROWS / 16 == 8, "the GeGLU epilogue's store count");
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // raw: __syncthreads' fence would drain the stores after all
      asm volatile("" ::: "memory");
    } else {
      wait_allow(min(NS - 1, KT) - 1);
      step_barrier();
    }
    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const int nxt = kt + NS - 1, nbuf = buf == 0 ? NS - 1 : buf - 1;   // the slot read in step kt-1: every wave passed the barrier since
      const bool do_stage = nxt < KT;
      if (do_stage) stage(nxt, nbuf);   // right after the barrier (issuing half of the waves' share a substep later: no difference, r2s)
      compute(buf);
      wait_allow(max(0, min(NS - 2, KT - 2 - kt)));   // step kt+1 has landed (this wave's share)
      step_barrier();
      buf = buf + 1 == NS ? 0 : buf + 1;
    }
  };
  if (v_block) mainloop(std::false_type{});
  else mainloop(std::true_type{});

  auto prefetch_next = [&]() {
    if constexpr (XPREF) {
      const int ntix = tix + per_xcd_wgs;
      stage0_in_flight = ntix < range_len;
      if (stage0_in_flight) {
        const int nb = range_lo + ntix;
        stage0_of(p.A + (size_t)((nb / nbn) * BM) * K, p.W + (size_t)((nb % nbn) * BN) * K);
      }
    }
  };
  if (kg == 0) gemm_epilogue<EPI, RT, WROWS, T, (BM == 128 && BN == 128 && (NS == 4 || HW > 0)), XPREF>(p, acc, smem, wave, lane, mw, nw, v_block, prefetch_next);
  else if constexpr (EPI == EPI_GEGLU) __syncthreads();   // the GeGLU epilogue's pair-staging barrier is a workgroup barrier
  if constexpr (XPREF) {   // the next tile's stage 0 is in flight: a raw barrier (the staging reads are in registers: the stores consumed them)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    __syncthreads();  // staging area is reused as operand slots by the next tile
  }
  }  // tile loop
}

int gemm_small_m_threshold(int set_to) {
  static std::atomic<int> thr{8192};
  if (set_to >= 0) thr.store(set_to);
  return thr.load();
}

// One instantiation: dynamic-LDS attribute on first use, persistent grid of at most `grid_cap` workgroups.
template <int EPI, int BM, int BN, int WM, int WN, int NS, typename T, int HW = 0, bool KCH = false>
static hipError_t launch_cfg(GemmParams p, int grid_cap, hipStream_t stream) {
  constexpr int SMEM = NS * (BM + BN) * BK * 2;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, BM, BN, WM, WN, NS, T, HW, KCH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
  p.n_tiles = nbm * nbn;
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, WM, WN, NS, T, HW, KCH>), dim3(std::min(nbm * nbn, grid_cap)), dim3((WM * WN + HW) * 64), SMEM,
                     stream, p);
  return hipGetLastError();
}

template <int EPI, typename T>
hipError_t launch_t(const GemmParams& p, hipStream_t stream) {
  // Small batches (a query's handful of chunks): few tiles, so the K loop's chain of memory round trips is the
  // whole kernel -- 128x128 tiles (4x the workgroups) with four LDS stages (three K-steps in flight), 1 workgroup per CU.
  if constexpr (EPI == EPI_TOPK) {
    // <= 64 query columns (round 6): 256 x 64 tiles on a FOUR-stage ring -- 96 KB of rows in flight per CU (all 160 KiB of LDS:
    // the EPI_TOPK epilogue stages nothing), one wave per SIMD with the same 64 x 64 accumulators per wave.  N = 64 has no other
    // tile, so the first (small) stages of a search take it too; the shard's allocation is padded past the last tile's rows.
    if (p.topk_tile == 2) return launch_cfg<EPI, 256, 64, 4, 1, 4, T>(p, 256, stream);
  }
  if (p.M <= gemm_small_m_threshold(-1) && EPI != EPI_NONE) {
    if constexpr (EPI == EPI_RESIDUAL) {
      // N = 768 at ~1 000 rows is 48 tiles of 128 x 128 on 256 CUs, and a K-step there is bound by what ONE CU's LDS-DMA
      // brings in (32 KiB per step: 1.2 us measured, profiles/r03_latency_kernel_stats.txt), not by its 0.25 us of MFMAs:
      // 64 x 64 tiles (one wave each, 16 KiB per step) spread the same bytes over four times the CUs
      // (extract_spans(question, 5 chunks) 1.83 -> 1.74 ms, profiles/r03_small_residual_tiles_ab.txt).
      if ((int64_t)((p.M + 127) / 128) * (p.N / 128) <= 128) {
        // at most one 64 x 64 tile per CU: four waves split K over an eight-stage ring (see the kernel): 12.4 -> 9.8 us at K = 768,
        // 15.6 -> 11.5 us at K = 1152, extract_spans(question, 5 chunks) 1.68 -> 1.59 ms (profiles/r04_small_gemm_probes.txt)
        if ((int64_t)((p.M + 63) / 64) * (p.N / 64) <= 256) return launch_cfg<EPI, 64, 64, 1, 1, 8, T, 3>(p, 1024, stream);
        return launch_cfg<EPI, 64, 64, 1, 1, 4, T, 0, true>(p, 1024, stream);   // same summation order as the K-split (KCH)
      }
      return launch_cfg<EPI, 128, 128, 2, 2, 4, T, 0, true>(p, 256, stream);    // ... and here: one order for every launch-bound residual GEMM
    }
    // (two K-groups of 2 x 2 waves for these 128 x 128 tiles measured SLOWER: extract_spans(question, 5 chunks) 1.71 vs 1.56 ms,
    //  profiles/r04_small_gemm_probes.txt -- eight waves per CU, a 64 KiB reduction and twice the barriers cost more than the
    //  halved K-step chain returns; the template keeps the general form)
    // Round 6: EIGHT waves (4 x 2, 32 x 64 outputs each) on the same 128 x 128 tile and four-stage ring for the Wqkv / GeGLU / bf16-out
    // GEMMs of a launch-bound batch: a K-step there is a chain -- barrier, fragment reads, 32 MFMAs -- on the one wave a SIMD holds;
    // with two waves per SIMD one's fragment reads run under the other's MFMAs.  Worth less than the chain suggested: Wqkv 15.3 ->
    // 15.1 us, GeGLU 14.1 -> 13.5 us per launch at ~1 000 rows, extract_spans(question, 5 chunks) 1.56 -> 1.52 ms
    // (profiles/r06_latency_kernel_stats.txt); same accumulation order per output element, same bits.
    if constexpr (EPI == EPI_QKV_ROPE || EPI == EPI_GEGLU || EPI == EPI_BF16)
      return launch_cfg<EPI, 128, 128, 4, 2, 4, T>(p, 256, stream);
    return launch_cfg<EPI, 128, 128, 2, 2, 4, T>(p, 256, stream);
  }
  if constexpr (EPI == EPI_TOPK) {
    // the batched dense search streams its A operand (the corpus rows) from HBM, not from L2 like the encoder's activations: with
    // few query columns it is bound by the row bytes in flight per CU -- 256 x 128 tiles on a THREE-stage ring keep 64 KB of rows in
    // flight instead of 32 (GemmParams::topk_tile, chosen by the search for <= 128 query columns)
    if (p.topk_tile == 1 && p.M >= 256) return launch_cfg<EPI, 256, 128, 4, 2, 3, T>(p, 256, stream);
  }
  if (p.N % 256 == 0 && p.M >= 256 && (EPI != EPI_QKV_ROPE || p.hidden % 256 == 0))
    return launch_cfg<EPI, 256, 256, 2, 4, 2, T>(p, 256, stream);   // one persistent workgroup per CU
  return launch_cfg<EPI, 128, 128, 2, 2, 2, T>(p, 512, stream);
}

template <typename T>
static hipError_t launch_typed(GemmEpi epi, const GemmParams& p, hipStream_t stream) {
  switch (epi) {
    case EPI_F32: return launch_t<EPI_F32, T>(p, stream);
    case EPI_BF16: return launch_t<EPI_BF16, T>(p, stream);
    case EPI_F32_GELU: return launch_t<EPI_F32_GELU, T>(p, stream);
    case EPI_RESIDUAL: return launch_t<EPI_RESIDUAL, T>(p, stream);
    case EPI_GEGLU: return launch_t<EPI_GEGLU, T>(p, stream);
    case EPI_QKV_ROPE: return launch_t<EPI_QKV_ROPE, T>(p, stream);
    case EPI_SPLADE: return launch_t<EPI_SPLADE, T>(p, stream);
    case EPI_NONE: return launch_t<EPI_NONE, T>(p, stream);
    case EPI_TOPK: return launch_t<EPI_TOPK, T>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream) {
  if (p.M <= 0) return hipSuccess;
  if ((p.N % 128 != 0 && !(epi == EPI_TOPK && p.topk_tile == 2 && p.N == 64)) || p.K % BK != 0) return hipErrorInvalidValue;
  if ((p.lo_in || p.lo_out) && (epi != EPI_RESIDUAL || p.bias || p.res_mu || !p.resid_bf16 || (p.lo_in && !p.ln_shift_prev) || (p.lo_out && !p.ln_shift)))
    return hipErrorInvalidValue;   // the split stream exists for the pre-LN schedule's plain residual add only
  return p.op_dtype == kOpF16 ? launch_typed<f16_t>(epi, p, stream) : launch_typed<bf16_t>(epi, p, stream);
}

bool gemm_consumer_finalizes(int rows) {
  return rows > 0 && rows <= gemm_small_m_threshold(-1);               // launch_t's choice of the 128 x 128, NS = 4 configuration
}

const char* gemm_kernel_name(GemmEpi epi) {
  static const char* names[] = {"gemm_bf16_kernel<EPI_F32>",      "gemm_bf16_kernel<EPI_BF16>",
                                "gemm_bf16_kernel<EPI_F32_GELU>", "gemm_bf16_kernel<EPI_RESIDUAL>",
                                "gemm_bf16_kernel<EPI_GEGLU>",     "gemm_bf16_kernel<EPI_QKV_ROPE>",
                                "gemm_bf16_kernel<EPI_SPLADE>",    "gemm_bf16_kernel<EPI_NONE>",
                                "gemm_bf16_kernel<EPI_TOPK>"};
  return epi >= 0 && epi < EPI_COUNT ? names[epi] : "?";
}

unsigned gemm_f16_saturated(bool reset) { return f16_sat_take(reset); }
unsigned* gemm_f16_flag_address() { return f16_sat_flag_address(); }

}  // namespace vrag
