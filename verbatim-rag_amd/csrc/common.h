// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrag {

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MFMA operand type of an encoder: bf16 (default; fp32's exponent range) or fp16 (11 significant bits instead of 8 at
// the same MFMA rate -- what the per-token logits of the v2 highlighter need to stay within 1e-3 of the fp32 reference,
// tests/probes/precision_probe.py).  Both are 2 bytes: host-side buffers are typed bf16_t* whatever they hold and the
// kernels, templated on the operand type, reinterpret them.  fp16 conversions saturate at +-65504 instead of
// producing inf.
constexpr int kOpBf16 = 0, kOpF16 = 1;
// Set by any fp32 -> fp16 operand conversion that had to clamp (one copy per translation unit: no relocatable device
// code in this build; f16_sat_take() below reads and clears the copy of the file that includes it).
static __device__ unsigned vrag_f16_sat_flag;
template <typename T> struct Op;
template <> struct Op<bf16_t> {
  typedef bf16x4 v4;
  typedef bf16x8 v8;
  static __device__ __forceinline__ bf16_t to(float v) { return (bf16_t)v; }
  static __device__ __forceinline__ f32x16 mfma32(const v8& a, const v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(const v8& a, const v8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Op<f16_t> {
  typedef f16x4 v4;
  typedef f16x8 v8;
  static __device__ __forceinline__ f16_t to(float v) {
    // a value outside fp16's range is stored as +-65504 AND reported: silently clamped activations would come back as
    // plausible, wrong logits (vrag_encoder_f16_saturated; the branch is never taken on healthy checkpoints)
    if (__builtin_fabsf(v) > 65504.f) vrag_f16_sat_flag = 1u;
    return (f16_t)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  }
  static __device__ __forceinline__ f32x16 mfma32(const v8& a, const v8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(const v8& a, const v8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kWave = 64;

// Token rows of every activation buffer are padded to a multiple of this many
// rows so that GEMM/attention tiles never read outside an allocation.
constexpr int kRowPad = 256;
// Packed sequences start at token offsets that are multiples of this (keeps the
// key-contiguous V^T rows 16-byte aligned for vector loads).
constexpr int kSeqAlign = 8;

// Cross-lane steps as DPP modifiers (VALU only).  `__shfl_xor` compiles to ds_bpermute_b32 -- a trip through the LDS
// crossbar plus an s_waitcnt per step -- which is what the residual epilogue's row statistics used to spend their time on
// (288 bpermutes and as many waits per tile and wave).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum / max over each aligned group of 16 lanes, result in every lane of the group: xor 1, xor 2 (quad permutes), then
// mirror within 8 and within 16 (every lane already holds its quad's / its 8's total, so any cross pairing completes it)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// wave-uniform value the compiler can prove uniform (scalar register).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// exact (erf-form) GELU, matching torch.nn.functional.gelu default.
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// GELU for 16-bit-rounded outputs (GeGLU epilogue, BERT-family MLP).  gelu(x) = relu(x) - 0.5 |x| erfc(|x| / sqrt 2) and
// erfc(z) = 2^P(z), P a degree-6 polynomial through the origin (minimax fit on [0, 4.2], |erf error| < 2.4e-7; beyond 4.2
// erfc < 3e-9): 7 FMA-class operations and ONE transcendental (v_exp_f32), absolute error < 6e-7 in fp32
// (tools/fit_gelu.py).  The Abramowitz-Stegun 7.1.26 form used before needed v_rcp_f32 as well and ~13 operations; libm's
// erff ~40.  That matters when 1152 activations per token sit in a GEMM epilogue that is VALU-bound.  (The same arithmetic
// on 2-vectors -- v_pk_fma_f32 / v_pk_mul_f32 -- measured slower: 48.5 vs 45.5 us of epilogue per launch.)
__device__ __forceinline__ float gelu_fast(float x) {
  const float a = fabsf(x);
  const float z = fminf(a * 0.70710678118654752440f, 4.2f);
  float t = fmaf(1.420304104e-04f, z, -3.664225454e-03f);
  t = fmaf(t, z, 3.089610590e-02f);
  t = fmaf(t, z, -1.496993778e-01f);
  t = fmaf(t, z, -9.181654851e-01f);
  t = fmaf(t, z, -1.627925069e+00f);
  const float e = __builtin_amdgcn_exp2f(t * z);   // erfc(|x| / sqrt 2)
  return fmaf(-0.5f * a, e, fmaxf(x, 0.f));
}

// The same on two values at a time (round 4): the FMA-class operations as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two results
// per issue slot), abs / min / max / exp2 per component -- 14 issue slots per output instead of ~20 in the GeGLU epilogue, which is
// VALU-bound.  Same operations in the same order as gelu_fast: bit-identical results.  (Round 2 measured a packed form slower; that
// one built its pairs with moves -- here the pairs are the accumulators' own consecutive registers.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  f32x2 z = a * 0.70710678118654752440f;
  z = f32x2{fminf(z[0], 4.2f), fminf(z[1], 4.2f)};
  f32x2 t = pk_fma(splat2(1.420304104e-04f), z, splat2(-3.664225454e-03f));
  t = pk_fma(t, z, splat2(3.089610590e-02f));
  t = pk_fma(t, z, splat2(-1.496993778e-01f));
  t = pk_fma(t, z, splat2(-9.181654851e-01f));
  t = pk_fma(t, z, splat2(-1.627925069e+00f));
  const f32x2 tz = t * z;
  const f32x2 e = {__builtin_amdgcn_exp2f(tz[0]), __builtin_amdgcn_exp2f(tz[1])};   // erfc(|x| / sqrt 2)
  const f32x2 m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return pk_fma(a * -0.5f, e, m);
}

// Streaming (non-temporal) 16/8-byte accesses for data that is written once and consumed by a LATER
// kernel (GEMM / attention outputs, the residual read-modify-write): keeps the XCD's 4 MiB L2 for
// the operand panels that co-running tiles share.  Measured on the bf16-out GEMM: 594 -> 548 us.
// -DVRAG_PLAIN_STREAMS (probe build, tools/probes/mall_probe.py): ordinary accesses instead -- does the non-temporal hint
// keep the data out of the Infinity Cache as well?
#ifdef VRAG_PLAIN_STREAMS
__device__ __forceinline__ void store16_nt(void* dst, const f32x4& v) { *reinterpret_cast<f32x4*>(dst) = v; }
template <typename V4>
__device__ __forceinline__ void store8_nt(void* dst, const V4& v) { *reinterpret_cast<V4*>(dst) = v; }
__device__ __forceinline__ f32x4 load16_nt(const void* src) { return *reinterpret_cast<const f32x4*>(src); }
#else
__device__ __forceinline__ void store16_nt(void* dst, const f32x4& v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
}
template <typename V4>
__device__ __forceinline__ void store8_nt(void* dst, const V4& v) {
  __builtin_nontemporal_store(v, reinterpret_cast<V4*>(dst));
}
__device__ __forceinline__ f32x4 load16_nt(const void* src) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
}
#endif

// Host side of vrag_f16_sat_flag for THIS translation unit: 1 if a conversion clamped since the last reset (synchronises
// the device: callers use it on the read-back path, never between launches).
static inline unsigned f16_sat_take(bool reset) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(vrag_f16_sat_flag), sizeof(v)) != hipSuccess) return 0;
  if (reset && v) {
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vrag_f16_sat_flag), &zero, sizeof(zero));
  }
  return v;
}

// Device address of THIS translation unit's vrag_f16_sat_flag on the current device: the encoder gathers the flags of all its
// translation units with ONE launch into a pinned word (capi.hip: vrag_encoder_f16_saturated) instead of one synchronous
// symbol copy per file (five ~12 us copies on every fp16 call's read-back path).
static inline unsigned* f16_sat_flag_address() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(vrag_f16_sat_flag)) != hipSuccess) return nullptr;
  return reinterpret_cast<unsigned*>(p);
}

// Top-k candidate keys (csrc/topk.hip; also built by the EPI_TOPK epilogue of csrc/gemm_bf16.hip): one u64
//   [ orderable(score) : 32 | 0xFFFFFFFF - local_row : 32 ]      (max key == best hit under (score desc, id asc))
typedef unsigned long long u64;
__device__ __forceinline__ unsigned orderable(float s) {
  const unsigned b = __builtin_bit_cast(unsigned, s);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float unorderable(unsigned k) {
  const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, b);
}
__device__ __forceinline__ u64 make_key(float s, unsigned row) {
  return ((u64)orderable(s) << 32) | (u64)(0xFFFFFFFFu - row);
}

// 16-byte global -> LDS DMA. LDS destination = wave-uniform `lds` + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

}  // namespace vrag
