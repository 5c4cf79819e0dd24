// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T, fp32 accumulate, fused epilogues.
//
// Replaces the nn.Linear calls of the reference's ModernBERT forward
// (transformers modeling_modernbert.py: Wqkv :271, attn Wo :300, mlp Wi :90, mlp Wo :91,
//  prediction-head dense :487, MLM decoder :550, token classifier :697).
//
// Two tile configurations of one template (v_mfma_f32_32x32x16_bf16, BK = 64):
//   256(M) x 256(N): 512 threads = 8 waves as 2(M) x 4(N), 128x64 per wave (128 accumulator VGPRs),
//                    128 KiB LDS (2 stages x (32 KiB A + 32 KiB W)), 1 workgroup / CU  -- used when N % 256 == 0
//   128(M) x 128(N): 256 threads = 4 waves as 2x2, 64x64 per wave, 64 KiB LDS, 2 workgroups / CU
// The larger tile halves the L2->LDS operand traffic per FLOP (M*N*K*2*(1/BM+1/BN) bytes), which is
// the co-limiter of the small tile at these K (768/1152).  Operands go HBM -> LDS by 16-byte LDS-DMA (global_load_lds),
// double buffered.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-swizzled with
// ((row>>1)&7) on the *source address* (LDS-DMA writes lane-linear) and on the ds_read_b128
// address, which makes each 16-lane read group hit 16 distinct 16-byte slots of the 256-byte
// bank row (conflict free).
//
// The MFMA is issued "swapped" (weights as the A operand, activations as B) so that a lane's
// accumulator registers hold 4 *consecutive output features* of ONE token row:
//   token  m = m0 + wm*64 + mi*32 + (lane&31)
//   feature n = n0 + wn*64 + ni*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
// -> 8-byte bf16 / 16-byte fp32 row-major stores, and RoPE's (d, d+32) / GeGLU's (x1, x2)
// partners sit in the same lane and register index of acc[0][mi] / acc[1][mi].
#include "gemm_bf16.h"

#include <type_traits>

namespace vrag {

constexpr int BK = 64;

__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

// BM x BN tile, WM x WN waves; every wave owns (BM/WM) x 64 outputs (MI = BM/WM/32 row tiles, 2 column tiles).
template <int EPI, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(const GemmParams p) {
  static_assert(BN == WN * 64, "a wave spans exactly 64 output features (one head / one GeGLU group)");
  constexpr int NT = WM * WN * 64;            // threads
  constexpr int MI = BM / WM / 32;            // 32-row accumulator tiles per wave
  constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int A_INSTR = BM / 8 / (WM * WN); // LDS-DMA instructions per wave per stage (8 rows each)
  constexpr int W_INSTR = BN / 8 / (WM * WN);
  static_assert(A_INSTR * 8 * WM * WN == BM && W_INSTR * 8 * WM * WN == BN, "tile rows must split over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int hi = lane >> 5, l31 = lane & 31;
  constexpr int WROWS = BM / WM;              // rows of the A tile owned by one wave

  // XCD-aware, bijective block remap: consecutive logical tiles (same A row panel, n fastest)
  // land on the same XCD so the panel is served from that XCD's L2.
  const int nbn = p.N / BN;
  const int nblk = gridDim.x;
  int b = blockIdx.x;
  {
    const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (b / nbn) * BM, n0 = (b % nbn) * BN;
  const int K = p.K;

  const bf16_t* __restrict__ Ab = p.A + (size_t)m0 * K;
  const bf16_t* __restrict__ Wb = p.W + (size_t)n0 * K;

  // LDS-DMA staging: instruction i of this wave fills tile rows wave*8*INSTR + i*8 .. +8.
  int soffA[A_INSTR], soffW[W_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = wave * (8 * A_INSTR) + i * 8 + (lane >> 3);
    soffA[i] = row * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int i = 0; i < W_INSTR; ++i) {
    const int row = wave * (8 * W_INSTR) + i * 8 + (lane >> 3);
    soffW[i] = row * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
  // fragment read offsets (bytes inside a 32-row sub-tile)
  const int sw = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fo[s] = l31 * 128 + ((((2 * s + hi) ^ sw)) << 4);

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE_BYTES;
    char* sW = sA + A_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) glds16(Ab + soffA[i] + k0, sA + (wave * (8 * A_INSTR) + i * 8) * 128);
#pragma unroll
    for (int i = 0; i < W_INSTR; ++i) glds16(Wb + soffW[i] + k0, sW + (wave * (8 * W_INSTR) + i * 8) * 128);
  };

  f32x16 acc[2][MI];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < MI; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  const int KT = K / BK;
  // The V third of the QKV product is computed un-swapped (activations as the A operand): then a
  // lane holds 4 consecutive TOKENS of one feature, which is the V^T row layout attention wants.
  bool v_block = false;
  if constexpr (EPI == EPI_QKV_ROPE) v_block = n0 >= 2 * p.hidden;  // workgroup-uniform (BN divides hidden)

  auto mainloop = [&](auto swapped_tag) {
    constexpr bool SWAPPED = decltype(swapped_tag)::value;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < KT) stage(kt + 1, buf ^ 1);
      const char* sA = smem + buf * STAGE_BYTES + (wm * WROWS) * 128;
      const char* sW = smem + buf * STAGE_BYTES + A_BYTES + (wn * 64) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[MI], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sW + i * 32 * 128 + fo[s]);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sA + i * 32 * 128 + fo[s]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr (SWAPPED)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            else
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], wf[ni], acc[ni][mi], 0, 0, 0);
          }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  };
  if (v_block) mainloop(std::false_type{});
  else mainloop(std::true_type{});

  // ------------------------------------------------------------------ epilogues
  // All operand-tile reads are done (the loop ends with a barrier), so the LDS is reused as a
  // per-wave 16 KiB staging area: accumulators are written in their natural (row-per-lane)
  // layout with an XOR-swizzled 16-byte chunk index and read back row-contiguous, so every
  // global store/RMW instruction covers whole 128/256-byte row segments (full cache lines)
  // instead of 32 scattered 16-byte pieces.
  const int nw = n0 + wn * 64;  // first feature of this wave's 64-wide range
  char* stg = smem + wave * 16384;
  const int mw = m0 + wm * WROWS;  // first token row of this wave

  // bf16 tile [R rows][C cols] (C = 64 or 32): lane writes 4 consecutive columns of its row.
  auto put_bf16 = [&](int row, int col, const bf16x4& v, int row_bytes) {
    const int c16 = col >> 3, half = (col >> 2) & 1;
    const int sw16 = row_bytes == 128 ? (row & 7) : (row_bytes == 64 ? (row & 3) : (row & 15));
    *reinterpret_cast<bf16x4*>(stg + row * row_bytes + ((c16 ^ sw16) << 4) + (half << 3)) = v;
  };
  // read back 16 bytes: chunk c16 of `row`
  auto get16 = [&](int row, int c16, int row_bytes) -> f32x4 {
    const int sw16 = row_bytes == 128 ? (row & 7) : (row_bytes == 64 ? (row & 3) : (row & 15));
    return *reinterpret_cast<const f32x4*>(stg + row * row_bytes + ((c16 ^ sw16) << 4));
  };

  if constexpr (EPI == EPI_NONE) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[ni][mi][r]));
  } else if constexpr (EPI == EPI_F32 || EPI == EPI_F32_GELU || EPI == EPI_RESIDUAL) {
    // fp32 [64 rows][64 cols] per pass (256-byte rows, 16 chunks), MI/2 passes
#pragma unroll
    for (int ps = 0; ps < MI / 2; ++ps) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh) {
        const int mi = ps * 2 + mh;
        const int row = mh * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
            if constexpr (EPI == EPI_F32_GELU) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            }
            const int c16 = ni * 8 + 2 * g + hi;
            *reinterpret_cast<f32x4*>(stg + row * 256 + ((c16 ^ (row & 15)) << 4)) = v;
          }
      }
      // read back: 16 lanes per row, 4 rows per instruction
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + (lane >> 4), c16 = lane & 15;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 256 + ((c16 ^ (row & 15)) << 4));
        float* dst = p.out_f32 + (size_t)(mw + ps * 64 + row) * p.N + nw + c16 * 4;
        if constexpr (EPI == EPI_RESIDUAL) {
          v += *reinterpret_cast<const f32x4*>(dst);
        } else if constexpr (EPI == EPI_F32) {
          if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + nw + c16 * 4);
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  } else if constexpr (EPI == EPI_BF16) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = ni * 32 + 8 * g + 4 * hi;
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[ni][mi][4 * g + j];
            if (p.bias) v += p.bias[nw + col + j];
            o[j] = (bf16_t)v;
          }
          put_bf16(mi * 32 + l31, col, o, 128);
        }
#pragma unroll
    for (int it = 0; it < MI * 4; ++it) {
      const int row = it * 8 + (lane >> 3), c16 = lane & 7;
      *reinterpret_cast<f32x4*>(p.out_bf16 + (size_t)(mw + row) * p.N + nw + c16 * 8) = get16(row, c16, 128);
    }
  } else if constexpr (EPI == EPI_GEGLU) {
    // Wi rows were interleaved at load time: each 64-row group = 32 "input" rows (x1)
    // followed by the 32 matching "gate" rows (x2).  Output tile: [rows][32 features], 64-byte rows.
    const int NO = p.N >> 1;
    const int f0 = (nw >> 6) * 32;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(gelu_erf(acc[0][mi][4 * g + j]) * acc[1][mi][4 * g + j]);
        put_bf16(mi * 32 + l31, 8 * g + 4 * hi, o, 64);
      }
#pragma unroll
    for (int it = 0; it < MI * 2; ++it) {
      const int row = it * 16 + (lane >> 2), c16 = lane & 3;
      *reinterpret_cast<f32x4*>(p.out_bf16 + (size_t)(mw + row) * NO + f0 + c16 * 8) = get16(row, c16, 64);
    }
  } else if constexpr (EPI == EPI_QKV_ROPE) {
    const int H = p.hidden;
    const int which = nw / H;  // 0 = q, 1 = k, 2 = v   (wave-uniform)
    const int head = (nw - which * H) >> 6;
    if (!v_block) {
      bf16_t* dst = which == 0 ? p.q : p.k;
      const float scale = which == 0 ? p.q_scale : 1.0f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int pos = p.pos[mw + mi * 32 + l31];
        const float* cs = p.rope_cos + (size_t)pos * 32;
        const float* sn = p.rope_sin + (size_t)pos * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 8 * g + 4 * hi;
          const f32x4 c = *reinterpret_cast<const f32x4*>(cs + dd);
          const f32x4 sv = *reinterpret_cast<const f32x4*>(sn + dd);
          bf16x4 o1, o2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x1 = acc[0][mi][4 * g + j], x2 = acc[1][mi][4 * g + j];
            // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)  (TF:188-219)
            o1[j] = (bf16_t)((x1 * c[j] - x2 * sv[j]) * scale);
            o2[j] = (bf16_t)((x2 * c[j] + x1 * sv[j]) * scale);
          }
          put_bf16(mi * 32 + l31, dd, o1, 128);
          put_bf16(mi * 32 + l31, dd + 32, o2, 128);
        }
      }
#pragma unroll
      for (int it = 0; it < MI * 4; ++it) {
        const int row = it * 8 + (lane >> 3), c16 = lane & 7;
        *reinterpret_cast<f32x4*>(dst + (size_t)(mw + row) * H + head * 64 + c16 * 8) = get16(row, c16, 128);
      }
    } else {
      // un-swapped accumulators: lane = feature d (ni*32 + l31), registers = tokens
      //   token = mi*32 + 8*(r>>2) + 4*hi + (r&3).   Stage V^T tile [64 d][WROWS tokens].
      constexpr int RB = WROWS * 2;  // row bytes (256 for 128 tokens, 128 for 64)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16_t)acc[ni][mi][4 * g + j];
            put_bf16(ni * 32 + l31, mi * 32 + 8 * g + 4 * hi, o, RB);
          }
      constexpr int LPR = RB / 16;        // lanes per row
      constexpr int RPI = 64 / LPR;       // rows per instruction
#pragma unroll
      for (int it = 0; it < 64 / RPI; ++it) {
        const int row = it * RPI + lane / LPR, c16 = lane % LPR;
        *reinterpret_cast<f32x4*>(p.vt + (size_t)(head * 64 + row) * p.vt_ld + mw + c16 * 8) = get16(row, c16, RB);
      }
    }
  } else if constexpr (EPI == EPI_SPLADE) {
    // max over the tokens of each sequence of log1p(relu(logit + bias)).
    int sq[MI];
    bool same_l = true;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) sq[mi] = p.tok_seq[mw + mi * 32 + l31];
    const int s0 = uniform(sq[0]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) same_l = same_l && (sq[mi] == s0);
    const bool same = __all(same_l);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nw + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float bias = p.bias ? p.bias[n] : 0.f;
        float vv[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) vv[mi] = log1pf(fmaxf(acc[ni][mi][r] + bias, 0.f));
        if (same) {
          if (s0 < 0) continue;
          float v = vv[0];
#pragma unroll
          for (int mi = 1; mi < MI; ++mi) v = fmaxf(v, vv[mi]);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
          if (l31 == 0 && v > 0.f) atomicMax(p.splade_rows + (size_t)s0 * p.N + n, f2u(v));
        } else {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            if (sq[mi] >= 0 && vv[mi] > 0.f) atomicMax(p.splade_rows + (size_t)sq[mi] * p.N + n, f2u(vv[mi]));
        }
      }
  }
}

template <int EPI>
hipError_t launch_t(const GemmParams& p, hipStream_t stream) {
  if (p.N % 256 == 0 && p.M >= 256 && (EPI != EPI_QKV_ROPE || p.hidden % 256 == 0)) {
    constexpr int BM = 256, BN = 256, SMEM = 2 * (BM + BN) * BK * 2;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<EPI, BM, BN, 2, 4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
      if (e != hipSuccess) return e;
      attr = true;
    }
    const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, 2, 4>), dim3(nbm * nbn), dim3(512), SMEM, stream, p);
  } else {
    constexpr int BM = 128, BN = 128, SMEM = 2 * (BM + BN) * BK * 2;
    const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, BM, BN, 2, 2>), dim3(nbm * nbn), dim3(256), SMEM, stream, p);
  }
  return hipGetLastError();
}

hipError_t launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream) {
  if (p.M <= 0) return hipSuccess;
  if (p.N % 128 != 0 || p.K % BK != 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_F32: return launch_t<EPI_F32>(p, stream);
    case EPI_BF16: return launch_t<EPI_BF16>(p, stream);
    case EPI_F32_GELU: return launch_t<EPI_F32_GELU>(p, stream);
    case EPI_RESIDUAL: return launch_t<EPI_RESIDUAL>(p, stream);
    case EPI_GEGLU: return launch_t<EPI_GEGLU>(p, stream);
    case EPI_QKV_ROPE: return launch_t<EPI_QKV_ROPE>(p, stream);
    case EPI_SPLADE: return launch_t<EPI_SPLADE>(p, stream);
    case EPI_NONE: return launch_t<EPI_NONE>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

const char* gemm_kernel_name(GemmEpi epi) {
  static const char* names[] = {"gemm_bf16_kernel<EPI_F32>",      "gemm_bf16_kernel<EPI_BF16>",
                                "gemm_bf16_kernel<EPI_F32_GELU>", "gemm_bf16_kernel<EPI_RESIDUAL>",
                                "gemm_bf16_kernel<EPI_GEGLU>",     "gemm_bf16_kernel<EPI_QKV_ROPE>",
                                "gemm_bf16_kernel<EPI_SPLADE>",    "gemm_bf16_kernel<EPI_NONE>"};
  return epi >= 0 && epi < EPI_COUNT ? names[epi] : "?";
}

}  // namespace vrag
