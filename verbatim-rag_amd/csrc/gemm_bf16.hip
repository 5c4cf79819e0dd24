// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] . W[N,K]^T, fp32 accumulate, fused epilogues.
//
// Replaces the nn.Linear calls of the reference's ModernBERT forward
// (transformers modeling_modernbert.py: Wqkv :271, attn Wo :300, mlp Wi :90, mlp Wo :91,
//  prediction-head dense :487, MLM decoder :550, token classifier :697).
//
// Tile: 128(M) x 128(N) x 64(K) per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// v_mfma_f32_32x32x16_bf16).  Operands go HBM -> LDS by 16-byte LDS-DMA (global_load_lds),
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

namespace vrag {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;  // 16 KiB per operand tile
constexpr int SMEM_BYTES = 4 * TILE_BYTES;

__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int hi = lane >> 5, l31 = lane & 31;

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

  // LDS-DMA staging: instruction i of this wave fills tile rows wave*32 + i*8 .. +8.
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);
    soff[i] = row * K + lc * 8;
  }
  // fragment read offsets (bytes inside a 32-row sub-tile)
  const int sw = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fo[s] = l31 * 128 + ((((2 * s + hi) ^ sw)) << 4);

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * (2 * TILE_BYTES);
    char* sW = sA + TILE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(Ab + soff[i] + k0, sA + (wave * 32 + i * 8) * 128);
      glds16(Wb + soff[i] + k0, sW + (wave * 32 + i * 8) * 128);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  const int KT = K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) stage(kt + 1, buf ^ 1);
    const char* sA = smem + buf * (2 * TILE_BYTES) + (wm * 64) * 128;
    const char* sW = smem + buf * (2 * TILE_BYTES) + TILE_BYTES + (wn * 64) * 128;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[2], wf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sA + i * 32 * 128 + fo[s]);
        wf[i] = *reinterpret_cast<const bf16x8*>(sW + i * 32 * 128 + fo[s]);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogues
  const int nw = n0 + wn * 64;  // first feature of this wave's 64-wide range

  if constexpr (EPI == EPI_F32 || EPI == EPI_F32_GELU || EPI == EPI_RESIDUAL) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + l31;
      float* row = p.out_f32 + (size_t)m * p.N;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nw + ni * 32 + 8 * g + 4 * hi;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g + j];
          if constexpr (EPI == EPI_RESIDUAL) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(row + n);
            v += o;
          } else if constexpr (EPI == EPI_F32_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
          } else if (p.bias) {
            v += *reinterpret_cast<const f32x4*>(p.bias + n);
          }
          *reinterpret_cast<f32x4*>(row + n) = v;
        }
    }
  } else if constexpr (EPI == EPI_BF16) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + l31;
      bf16_t* row = p.out_bf16 + (size_t)m * p.N;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nw + ni * 32 + 8 * g + 4 * hi;
          bf16x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[ni][mi][4 * g + j];
            if (p.bias) v += p.bias[n + j];
            o[j] = (bf16_t)v;
          }
          *reinterpret_cast<bf16x4*>(row + n) = o;
        }
    }
  } else if constexpr (EPI == EPI_GEGLU) {
    // Wi rows were interleaved at load time: each 64-row group = 32 "input" rows (x1)
    // followed by the 32 matching "gate" rows (x2).
    const int NO = p.N >> 1;
    const int f0 = (nw >> 6) * 32;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm * 64 + mi * 32 + l31;
      bf16_t* row = p.out_bf16 + (size_t)m * NO + f0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = (bf16_t)(gelu_erf(acc[0][mi][4 * g + j]) * acc[1][mi][4 * g + j]);
        *reinterpret_cast<bf16x4*>(row + 8 * g + 4 * hi) = o;
      }
    }
  } else if constexpr (EPI == EPI_QKV_ROPE) {
    const int H = p.hidden;
    const int which = nw / H;              // 0 = q, 1 = k, 2 = v   (wave-uniform)
    const int head = (nw - which * H) >> 6;
    if (which < 2) {
      bf16_t* dst = which == 0 ? p.q : p.k;
      const float scale = which == 0 ? p.q_scale : 1.0f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 64 + mi * 32 + l31;
        const int pos = p.pos[m];
        const float* cs = p.rope_cos + (size_t)pos * 32;
        const float* sn = p.rope_sin + (size_t)pos * 32;
        bf16_t* row = dst + (size_t)m * H + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dd = 8 * g + 4 * hi;
          const f32x4 c = *reinterpret_cast<const f32x4*>(cs + dd);
          const f32x4 s = *reinterpret_cast<const f32x4*>(sn + dd);
          bf16x4 o1, o2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x1 = acc[0][mi][4 * g + j], x2 = acc[1][mi][4 * g + j];
            // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)  (TF:188-219)
            o1[j] = (bf16_t)((x1 * c[j] - x2 * s[j]) * scale);
            o2[j] = (bf16_t)((x2 * c[j] + x1 * s[j]) * scale);
          }
          *reinterpret_cast<bf16x4*>(row + dd) = o1;
          *reinterpret_cast<bf16x4*>(row + dd + 32) = o2;
        }
      }
    } else {
      // V^T: row = head*64 + d, column = token (key-contiguous for the PV MFMA operand)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 64 + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            p.vt[(size_t)(head * 64 + d) * p.vt_ld + m] = (bf16_t)acc[ni][mi][r];
          }
      }
    }
  } else if constexpr (EPI == EPI_SPLADE) {
    // max over the tokens of each sequence of log1p(relu(logit + bias)).
    int sq[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) sq[mi] = p.tok_seq[m0 + wm * 64 + mi * 32 + l31];
    const int s0 = uniform(sq[0]);
    const bool same = __all(sq[0] == s0 && sq[1] == s0);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nw + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float bias = p.bias ? p.bias[n] : 0.f;
        float v0 = log1pf(fmaxf(acc[ni][0][r] + bias, 0.f));
        float v1 = log1pf(fmaxf(acc[ni][1][r] + bias, 0.f));
        if (same) {
          if (s0 < 0) continue;
          float v = fmaxf(v0, v1);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
          if (l31 == 0 && v > 0.f) atomicMax(p.splade_rows + (size_t)s0 * p.N + n, f2u(v));
        } else {
          if (sq[0] >= 0 && v0 > 0.f) atomicMax(p.splade_rows + (size_t)sq[0] * p.N + n, f2u(v0));
          if (sq[1] >= 0 && v1 > 0.f) atomicMax(p.splade_rows + (size_t)sq[1] * p.N + n, f2u(v1));
        }
      }
  }
}

template <int EPI>
hipError_t launch_t(const GemmParams& p, hipStream_t stream) {
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN;
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(nbm * nbn), dim3(256), SMEM_BYTES, stream, p);
  return hipGetLastError();
}

hipError_t launch_gemm(GemmEpi epi, const GemmParams& p, hipStream_t stream) {
  if (p.M <= 0) return hipSuccess;
  if (p.N % BN != 0 || p.K % BK != 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_F32: return launch_t<EPI_F32>(p, stream);
    case EPI_BF16: return launch_t<EPI_BF16>(p, stream);
    case EPI_F32_GELU: return launch_t<EPI_F32_GELU>(p, stream);
    case EPI_RESIDUAL: return launch_t<EPI_RESIDUAL>(p, stream);
    case EPI_GEGLU: return launch_t<EPI_GEGLU>(p, stream);
    case EPI_QKV_ROPE: return launch_t<EPI_QKV_ROPE>(p, stream);
    case EPI_SPLADE: return launch_t<EPI_SPLADE>(p, stream);
    default: return hipErrorInvalidValue;
  }
}

const char* gemm_kernel_name(GemmEpi epi) {
  static const char* names[] = {"gemm_bf16_kernel<EPI_F32>",      "gemm_bf16_kernel<EPI_BF16>",
                                "gemm_bf16_kernel<EPI_F32_GELU>", "gemm_bf16_kernel<EPI_RESIDUAL>",
                                "gemm_bf16_kernel<EPI_GEGLU>",     "gemm_bf16_kernel<EPI_QKV_ROPE>",
                                "gemm_bf16_kernel<EPI_SPLADE>"};
  return epi >= 0 && epi < EPI_COUNT ? names[epi] : "?";
}

}  // namespace vrag
