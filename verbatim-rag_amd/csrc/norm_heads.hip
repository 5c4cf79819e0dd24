// Row-wise kernels (one 64-lane wave per token row, float4 accesses, wave-shuffle reductions).
// Restates nn.LayerNorm(bias=False) (transformers modeling_modernbert.py:61,70,312,314,476),
// the embedding gather (:64-71), the reference QAModel sentence head
// (packages/core/verbatim_core/extractor_models/model.py:82-113) and the
// ModernBertPredictionHead norm + classifier (modeling_modernbert.py:481-490,697-699).
#include "norm_heads.h"

namespace vrag {

namespace {

constexpr int MAXV = 4;  // float4 per lane: H <= 1024

// Loads one row (fp32) into registers: lane owns elements [lane*4 + 256*i, +4).
__device__ __forceinline__ void load_row(const float* row, int H, int lane, f32x4 (&x)[MAXV]) {
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < H) x[i] = *reinterpret_cast<const f32x4*>(row + c);
    else x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// In-register LayerNorm (biased variance, two-pass), result overwrites x.
__device__ __forceinline__ void ln_row(f32x4 (&x)[MAXV], const f32x4 (&w)[MAXV], int H, int lane, float eps,
                                       const float* bias = nullptr, float* mean_out = nullptr) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
  const float mean = wave_sum(s) / (float)H;
  if (mean_out) *mean_out = mean;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = x[i][j] - mean;
        x[i][j] = d;
        v += d * d;
      }
    }
  }
  const float var = wave_sum(v) / (float)H;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) x[i][j] = x[i][j] * rstd * w[i][j];
  if (bias) {  // nn.LayerNorm with bias (BERT family, TF:models/bert/modeling_bert.py:62)
    f32x4 b[MAXV];
    load_row(bias, H, lane, b);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) x[i] += b[i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int* __restrict__ ids, const float* __restrict__ E,
                                                        const float* __restrict__ w, float eps, int H, int rows,
                                                        float* __restrict__ h, bf16_t* __restrict__ a,
                                                        const float* __restrict__ P, const int* __restrict__ pos,
                                                        const float* __restrict__ type_row,
                                                        const float* __restrict__ bias,
                                                        const int* __restrict__ type_ids) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 x[MAXV], wv[MAXV];
  load_row(E + (size_t)ids[row] * H, H, lane, x);
  if (P) {  // (word + token_type) + position, the reference's order of additions (TF:models/bert/modeling_bert.py:100-104)
    f32x4 y[MAXV];
    if (type_row) {
      load_row(type_row + (type_ids ? (size_t)type_ids[row] * H : 0), H, lane, y);
#pragma unroll
      for (int i = 0; i < MAXV; ++i) x[i] += y[i];
    }
    load_row(P + (size_t)pos[row] * H, H, lane, y);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) x[i] += y[i];
  }
  load_row(w, H, lane, wv);
  ln_row(x, wv, H, lane, eps, bias);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < H) {
      *reinterpret_cast<f32x4*>(h + (size_t)row * H + c) = x[i];
      typename Op<T>::v4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = Op<T>::to(x[i][j]);
      *reinterpret_cast<typename Op<T>::v4*>(a + (size_t)row * H + c) = o;
    }
  }
}

// `of` may alias `h` (in place): a row is fully in registers before anything is written.
// gelu_first: h holds a raw dense output and the activation is applied here (split-operand heads accumulate their three
// partial GEMMs before the non-linearity).  split3: ob is a [rows, 3H] operand image  [hi | lo | hi]  of the normalised row
// (hi = T(x), lo = T(x - hi)) -- the A operand of a K = 3H GEMM against a [Whi | Whi | Wlo] weight, i.e. the three
// significant products of (hi + lo) . (Whi + Wlo) summed in ONE fp32 accumulation (SPLADE decoder, capi.hip run_splade).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* h, const float* __restrict__ w,
                                                         float eps, int H, int rows, bf16_t* __restrict__ ob,
                                                         float* of, const float* __restrict__ bias,
                                                         float* __restrict__ row_mean, bf16_t* __restrict__ ob_lo,
                                                         int gelu_first, int split3) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 x[MAXV], wv[MAXV];
  load_row(h + (size_t)row * H, H, lane, x);
  if (gelu_first) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[i][j] = gelu_erf(x[i][j]);
  }
  if (w) {
    load_row(w, H, lane, wv);
  } else {   // gain already folded into the consumer's weight
#pragma unroll
    for (int i = 0; i < MAXV; ++i) wv[i] = f32x4{1.f, 1.f, 1.f, 1.f};
  }
  float mean = 0.f;
  ln_row(x, wv, H, lane, eps, bias, &mean);
  if (row_mean && lane == 0) row_mean[row] = mean;
  const size_t ld = split3 ? (size_t)3 * H : (size_t)H;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < H) {
      if (of) *reinterpret_cast<f32x4*>(of + (size_t)row * H + c) = x[i];
      if (ob) {
        typename Op<T>::v4 o, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = Op<T>::to(x[i][j]);
          lo[j] = Op<T>::to(x[i][j] - (float)o[j]);   // remainder: x = o + lo to twice the operand precision
        }
        *reinterpret_cast<typename Op<T>::v4*>(ob + (size_t)row * ld + c) = o;
        if (split3) {
          *reinterpret_cast<typename Op<T>::v4*>(ob + (size_t)row * ld + H + c) = lo;
          *reinterpret_cast<typename Op<T>::v4*>(ob + (size_t)row * ld + 2 * H + c) = o;
        } else if (ob_lo) {
          *reinterpret_cast<typename Op<T>::v4*>(ob_lo + (size_t)row * H + c) = lo;
        }
      }
    }
  }
}

// One workgroup (4 waves) per range; waves stride over the range's tokens.
__global__ __launch_bounds__(256) void range_pool_kernel(const float* __restrict__ h, const float* __restrict__ lnw,
                                                          float eps, int H, const int* __restrict__ start,
                                                          const int* __restrict__ end, int mode,
                                                          const float* __restrict__ Wc, const float* __restrict__ bc,
                                                          int num_labels, float* __restrict__ out) {
  __shared__ float red[4][MAXV * 256];
  __shared__ float nrm[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x;
  const int s = start[r], e = end[r];
  f32x4 acc[MAXV], wv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (lnw) load_row(lnw, H, lane, wv);
  for (int t = s + wave; t <= e; t += 4) {
    f32x4 x[MAXV];
    load_row(h + (size_t)t * H, H, lane, x);
    if (lnw) ln_row(x, wv, H, lane, eps);   // post-LN encoders (BERT family) have no final LayerNorm
#pragma unroll
    for (int i = 0; i < MAXV; ++i) acc[i] += x[i];
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    *reinterpret_cast<f32x4*>(&red[wave][lane * 4 + 256 * i]) = acc[i];
  __syncthreads();
  const float inv_n = 1.0f / (float)(e - s + 1);
  // every wave rebuilds the mean vector (cheap) in registers
  f32x4 mean[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane * 4 + 256 * i;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[0][c]);
    v += *reinterpret_cast<const f32x4*>(&red[1][c]);
    v += *reinterpret_cast<const f32x4*>(&red[2][c]);
    v += *reinterpret_cast<const f32x4*>(&red[3][c]);
    mean[i] = v * inv_n;
  }
  if (mode == 0) {
    for (int c = wave; c < num_labels; c += 4) {
      f32x4 wc[MAXV];
      load_row(Wc + (size_t)c * H, H, lane, wc);
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d += mean[i][j] * wc[i][j];
      d = wave_sum(d);
      if (lane == 0) out[(size_t)r * num_labels + c] = d + bc[c];
    }
  } else {
    float scale = 1.0f;
    if (mode == 1) {
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) q += mean[i][j] * mean[i][j];
      q = wave_sum(q);
      scale = 1.0f / fmaxf(sqrtf(q), 1e-12f);
    }
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < H) *reinterpret_cast<f32x4*>(out + (size_t)r * H + c) = mean[i] * scale;
      }
    }
  }
  (void)nrm;
}

__global__ __launch_bounds__(256) void ln_classifier_kernel(const float* __restrict__ x_in, const float* __restrict__ lnw,
                                                             float eps, int H, int rows, const float* __restrict__ Wc,
                                                             const float* __restrict__ bc, int num_labels,
                                                             float* __restrict__ logits, const float* __restrict__ lnb,
                                                             int gelu_first) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 x[MAXV], wv[MAXV];
  load_row(x_in + (size_t)row * H, H, lane, x);
  if (gelu_first) {   // x_in is the raw dense output (split-operand head): the activation was not applied by a GEMM epilogue
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[i][j] = gelu_erf(x[i][j]);
  }
  load_row(lnw, H, lane, wv);
  ln_row(x, wv, H, lane, eps, lnb);
  for (int c = 0; c < num_labels; ++c) {
    f32x4 wc[MAXV];
    load_row(Wc + (size_t)c * H, H, lane, wc);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d += x[i][j] * wc[i][j];
    d = wave_sum(d);
    if (lane == 0) logits[(size_t)row * num_labels + c] = d + bc[c];
  }
}

// One workgroup per sequence.  Phase 1: the 4 waves split the H pooler rows, a wave owns a row at a time
// (lanes stride the H inputs, wave-shuffle reduction) and leaves tanh(.) in LDS; phase 2: one wave per label.
__global__ __launch_bounds__(256) void pooler_classifier_kernel(const float* __restrict__ h, int H,
                                                                 const int* __restrict__ first_row,
                                                                 const float* __restrict__ Wp, const float* __restrict__ bp,
                                                                 const float* __restrict__ Wc, const float* __restrict__ bc,
                                                                 int num_labels, float* __restrict__ logits) {
  __shared__ float pooled[MAXV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x;
  f32x4 x[MAXV];
  load_row(h + (size_t)first_row[s] * H, H, lane, x);
  for (int r = wave; r < H; r += 4) {
    f32x4 wv[MAXV];
    load_row(Wp + (size_t)r * H, H, lane, wv);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d += x[i][j] * wv[i][j];
    d = wave_sum(d);
    if (lane == 0) pooled[r] = tanhf(d + bp[r]);
  }
  __syncthreads();
  f32x4 pv[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane * 4 + 256 * i;
    pv[i] = c < H ? *reinterpret_cast<const f32x4*>(&pooled[c]) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int c = wave; c < num_labels; c += 4) {
    f32x4 wc[MAXV];
    load_row(Wc + (size_t)c * H, H, lane, wc);
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d += pv[i][j] * wc[i][j];
    d = wave_sum(d);
    if (lane == 0) logits[(size_t)s * num_labels + c] = d + bc[c];
  }
}

}  // namespace

hipError_t launch_embed_ln(const int* ids, const float* E, const float* w, float eps, int H, int rows, float* h,
                           bf16_t* a, hipStream_t stream, const float* P, const int* pos, const float* type_row,
                           const float* bias, const int* type_ids, int op_dtype) {
  if (rows <= 0) return hipSuccess;
  if (H > MAXV * 256 || (H & 3) || (P && !pos)) return hipErrorInvalidValue;
  if (op_dtype == kOpF16)
    hipLaunchKernelGGL(embed_ln_kernel<f16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, E, w, eps, H, rows, h, a, P, pos,
                       type_row, bias, type_ids);
  else
    hipLaunchKernelGGL(embed_ln_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, ids, E, w, eps, H, rows, h, a, P, pos,
                       type_row, bias, type_ids);
  return hipGetLastError();
}

hipError_t launch_layernorm(const float* h, const float* w, float eps, int H, int rows, bf16_t* ob, float* of,
                            hipStream_t stream, const float* bias, float* row_mean, int op_dtype, bf16_t* ob_lo, int gelu_first,
                            int split3) {
  if (rows <= 0) return hipSuccess;
  if (H > MAXV * 256 || (H & 3) || (split3 && (!ob || ob_lo))) return hipErrorInvalidValue;
  if (op_dtype == kOpF16)
    hipLaunchKernelGGL(layernorm_kernel<f16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, h, w, eps, H, rows, ob, of, bias, row_mean, ob_lo,
                       gelu_first, split3);
  else
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, stream, h, w, eps, H, rows, ob, of, bias, row_mean, ob_lo,
                       gelu_first, split3);
  return hipGetLastError();
}

hipError_t launch_range_pool(const float* h, const float* lnw, float eps, int H, const int* start, const int* end,
                             int n_ranges, int mode, const float* Wc, const float* bc, int num_labels, float* out,
                             hipStream_t stream) {
  if (n_ranges <= 0) return hipSuccess;
  if (H > MAXV * 256 || (H & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(range_pool_kernel, dim3(n_ranges), dim3(256), 0, stream, h, lnw, eps, H, start, end, mode, Wc,
                     bc, num_labels, out);
  return hipGetLastError();
}

hipError_t launch_ln_classifier(const float* x, const float* lnw, float eps, int H, int rows, const float* Wc,
                                const float* bc, int num_labels, float* logits, hipStream_t stream, const float* lnb,
                                int gelu_first) {
  if (rows <= 0) return hipSuccess;
  if (H > MAXV * 256 || (H & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ln_classifier_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, lnw, eps, H, rows, Wc, bc,
                     num_labels, logits, lnb, gelu_first);
  return hipGetLastError();
}

hipError_t launch_pooler_classifier(const float* h, int H, const int* first_row, int n_seqs, const float* Wp,
                                    const float* bp, const float* Wc, const float* bc, int num_labels, float* logits,
                                    hipStream_t stream) {
  if (n_seqs <= 0) return hipSuccess;
  if (H > MAXV * 256 || (H & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pooler_classifier_kernel, dim3(n_seqs), dim3(256), 0, stream, h, H, first_row, Wp, bp, Wc, bc, num_labels,
                     logits);
  return hipGetLastError();
}

unsigned norm_heads_f16_saturated(bool reset) { return f16_sat_take(reset); }
unsigned* norm_heads_f16_flag_address() { return f16_sat_flag_address(); }

}  // namespace vrag
