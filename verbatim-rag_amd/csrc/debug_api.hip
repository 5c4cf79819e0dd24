// Tuning / unit-test harness of the kernels (include/vrag_amd_debug.h): synthetic-operand timing loops and the attention
// kernels' unit-test hook.  NOT part of the product library: compiled only into libvrag_amd_dbg.so (build.py, -DVRAG_DEBUG_API),
// which tools/ and the attention unit test load beside libvrag_amd.so.
#include "../../include/vrag_amd.h"
#include "../../include/vrag_amd_debug.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "attention.h"
#include "gemm_bf16.h"
#include "norm_heads.h"
#include "qkv_attn.h"

namespace vrag {
void set_error(const char* fmt, ...);
}
using namespace vrag;

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));          \
      return VRAG_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

#define ARG_CHECK(cond, ...)      \
  do {                            \
    if (!(cond)) {                \
      set_error(__VA_ARGS__);     \
      return VRAG_ERR_INVALID;    \
    }                             \
  } while (0)

extern "C" {

int vrag_debug_gemm_ms(int32_t epi, int32_t M, int32_t N, int32_t K, int32_t iters, int32_t device, float* ms_out) {
  ARG_CHECK(ms_out && M > 0 && N % 128 == 0 && K % 64 == 0 && iters > 0, "bad arguments");
  ARG_CHECK(epi == EPI_F32 || epi == EPI_BF16 || epi == EPI_RESIDUAL || epi == EPI_GEGLU || epi == EPI_QKV_ROPE ||
                epi == EPI_F32_GELU || epi == EPI_NONE,
            "unsupported epilogue for the diagnostic");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  const size_t Mp = (size_t)align_up(M, kRowPad);
  void *A = nullptr, *W = nullptr, *outf = nullptr, *outb = nullptr, *q = nullptr, *kk = nullptr, *vt = nullptr;
  float *cs = nullptr, *sn = nullptr;
  int* pos = nullptr;
  auto cleanup = [&]() {
    for (void* p : {A, W, outf, outb, q, kk, vt, (void*)cs, (void*)sn, (void*)pos})
      if (p) (void)hipFree(p);
  };
  hipError_t e = hipMalloc(&A, Mp * K * 2);
  if (e == hipSuccess) e = hipMalloc(&W, (size_t)N * K * 2);
  if (e == hipSuccess) e = hipMalloc(&outf, Mp * N * 4);
  if (e == hipSuccess) e = hipMalloc(&outb, Mp * N * 2);
  if (e == hipSuccess) e = hipMalloc(&q, Mp * N * 2);
  if (e == hipSuccess) e = hipMalloc(&kk, Mp * N * 2);
  if (e == hipSuccess) e = hipMalloc(&vt, Mp * N * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&cs, 512 * 32 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&sn, 512 * 32 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&pos, Mp * 4);
  if (e != hipSuccess) {
    cleanup();
    set_error("debug gemm allocation failed: %s", hipGetErrorString(e));
    return VRAG_ERR_HIP;
  }
  // pseudo-random bf16 operands in [-1, 1): 0x3f80 | 7 mantissa bits = [1,2), minus 1.5, times 2
  {
    std::vector<unsigned short> h(std::max(Mp * K, (size_t)N * K));
    unsigned x = 12345u;
    for (auto& v : h) {
      x = x * 1664525u + 1013904223u;
      const unsigned m = (x >> 9) & 0x7f, s = (x >> 31) << 15, ex = 0x3e80u + (((x >> 20) & 1) << 7);
      v = (unsigned short)(s | ex | m);
    }
    if (getenv("VRAG_DEBUG_GEMM_ZERO")) std::fill(h.begin(), h.end(), (unsigned short)0);   // probe: operand-data dependence of the clock
    (void)hipMemcpy(A, h.data(), Mp * K * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  }
  (void)hipMemset(outf, 0, Mp * N * 4);
  (void)hipMemset(pos, 0, Mp * 4);
  (void)hipMemset(cs, 0, 512 * 32 * 4);
  (void)hipMemset(sn, 0, 512 * 32 * 4);
  GemmParams g{};
  g.op_dtype = getenv("VRAG_DEBUG_GEMM_F16") ? kOpF16 : kOpBf16;   // same bit patterns read as fp16: finite values in [2^-15, 2^-7)
  g.A = (const bf16_t*)A;
  g.W = (const bf16_t*)W;
  g.M = M;
  g.N = N;
  g.K = K;
  g.out_f32 = (float*)outf;
  g.out_bf16 = (bf16_t*)outb;
  g.q = (bf16_t*)q;
  g.k = (bf16_t*)kk;
  g.vt = (bf16_t*)vt;
  g.vt_ld = (int)Mp;
  g.rope_cos = cs;
  g.rope_sin = sn;
  g.pos = pos;
  g.hidden = N / 3;
  g.q_scale = 0.125f;
  if (epi == EPI_RESIDUAL && !getenv("VRAG_DEBUG_GEMM_PLAIN_RESID")) {   // as the encoder launches it with the LayerNorm fold
    g.resid_bf16 = (bf16_t*)outb;
    g.stats_part = (float*)q;                                              // Mp * N/64 * 2 floats <= Mp * N * 2 bytes
    g.stats_ld = (int)Mp;
    if (getenv("VRAG_DEBUG_GEMM_SPLIT")) {   // the split residual stream on both sides (layers >= 1 of the encoder schedule)
      g.lo_in = (const unsigned char*)kk;
      g.lo_out = (unsigned char*)kk;
      g.ln_shift = (const float*)pos;        // zeros
      g.ln_shift_prev = (float*)pos;
      (void)hipMemset(kk, 0, Mp * N * 2);
      (void)hipMemset(outb, 0, Mp * N * 2);
    }
  }
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipError_t le = hipSuccess;
  for (int i = 0; i < 3 && le == hipSuccess; ++i) le = launch_gemm((GemmEpi)epi, g, 0);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters && le == hipSuccess; ++i) le = launch_gemm((GemmEpi)epi, g, 0);
  (void)hipEventRecord(b, 0);
  hipError_t se = hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  cleanup();
  if (le != hipSuccess || se != hipSuccess) {
    set_error("debug gemm failed: %s", hipGetErrorString(le != hipSuccess ? le : se));
    return VRAG_ERR_HIP;
  }
  *ms_out = ms / iters;
  return VRAG_OK;
}

int vrag_debug_attn_ms(int32_t local, int32_t n_seqs, int32_t S, int32_t H, int32_t window, int32_t iters, int32_t device,
                       float* ms_out) {
  ARG_CHECK(ms_out && n_seqs > 0 && S > 0 && S % kSeqAlign == 0 && H % 64 == 0 && iters > 0, "bad arguments");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  const size_t T = (size_t)n_seqs * S, Tp = (size_t)align_up((int)T, kRowPad);
  const int qb = attention_q_block(local != 0);
  std::vector<int> bs, bl, bq;
  for (int s = 0; s < n_seqs; ++s)
    for (int q0 = 0; q0 < S; q0 += qb) {
      bs.push_back(s * S);
      bl.push_back(S);
      bq.push_back(q0);
    }
  void *q = nullptr, *k = nullptr, *vt = nullptr, *o = nullptr;
  int *d_bs = nullptr, *d_bl = nullptr, *d_bq = nullptr;
  auto cleanup = [&]() {
    for (void* p : {q, k, vt, o, (void*)d_bs, (void*)d_bl, (void*)d_bq})
      if (p) (void)hipFree(p);
  };
  hipError_t e = hipMalloc(&q, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&k, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&vt, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&o, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&d_bs, bs.size() * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_bl, bs.size() * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_bq, bs.size() * 4);
  if (e != hipSuccess) {
    cleanup();
    set_error("debug attention allocation failed: %s", hipGetErrorString(e));
    return VRAG_ERR_HIP;
  }
  {
    std::vector<unsigned short> h(Tp * H);
    unsigned x = 777u;
    for (auto& v : h) {   // pseudo-random bf16 in about [-1, 1), as vrag_debug_gemm_ms
      x = x * 1664525u + 1013904223u;
      v = (unsigned short)(((x >> 31) << 15) | (0x3e80u + (((x >> 20) & 1) << 7)) | ((x >> 9) & 0x7f));
    }
    (void)hipMemcpy(q, h.data(), Tp * H * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(k, h.data(), Tp * H * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(vt, h.data(), Tp * H * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_bs, bs.data(), bs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_bl, bl.data(), bs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_bq, bq.data(), bs.size() * 4, hipMemcpyHostToDevice);
  }
  AttnParams ap{};
  ap.q = (const bf16_t*)q;
  ap.k = (const bf16_t*)k;
  ap.vt = (const bf16_t*)vt;
  ap.o = (bf16_t*)o;
  ap.blk_seq_start = d_bs;
  ap.blk_seq_len = d_bl;
  ap.blk_q0 = d_bq;
  ap.n_blocks = (int)bs.size();
  ap.H = H;
  ap.nh = H / 64;
  ap.Tp = (int)Tp;
  ap.window = window;
  ap.op_dtype = getenv("VRAG_DEBUG_GEMM_F16") ? kOpF16 : kOpBf16;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipError_t le = hipSuccess;
  for (int i = 0; i < 3 && le == hipSuccess; ++i) le = launch_attention(ap, local != 0, 0);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters && le == hipSuccess; ++i) le = launch_attention(ap, local != 0, 0);
  (void)hipEventRecord(b, 0);
  hipError_t se = hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  cleanup();
  if (le != hipSuccess || se != hipSuccess) {
    set_error("debug attention failed: %s", hipGetErrorString(le != hipSuccess ? le : se));
    return VRAG_ERR_HIP;
  }
  *ms_out = ms / iters;
  return VRAG_OK;
}

int vrag_debug_qkv_attn_ms(int32_t local, int32_t n_seqs, int32_t S, int32_t H, int32_t window, int32_t iters, int32_t flags,
                           int32_t device, float* ms_out) {
  ARG_CHECK(ms_out && n_seqs > 0 && S > 0 && S <= kFusedMaxSeq && S % kSeqAlign == 0 && H % 64 == 0 && iters > 0, "bad arguments");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  const size_t T = (size_t)n_seqs * S, Tp = (size_t)align_up((int)T, kRowPad);
  const int nh = H / 64;
  std::vector<int> row(n_seqs), len(n_seqs, S);
  for (int s = 0; s < n_seqs; ++s) row[s] = s * S;
  void *x = nullptr, *w = nullptr, *o = nullptr;
  float *mu = nullptr, *rstd = nullptr, *lns = nullptr, *cs = nullptr;
  int *d_row = nullptr, *d_len = nullptr;
  auto cleanup = [&]() {
    for (void* p : {x, w, o, (void*)mu, (void*)rstd, (void*)lns, (void*)cs, (void*)d_row, (void*)d_len})
      if (p) (void)hipFree(p);
  };
  hipError_t e = hipMalloc(&x, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&w, (size_t)3 * H * H * 2);
  if (e == hipSuccess) e = hipMalloc(&o, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&mu, Tp * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&rstd, Tp * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&lns, ((size_t)3 * H + 64) * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&cs, (size_t)kFusedMaxSeq * 32 * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_row, (size_t)n_seqs * 8 * sizeof(int4));   // the groups' wave descriptors
  if (e == hipSuccess) e = hipMalloc((void**)&d_len, n_seqs * 4);
  if (e != hipSuccess) {
    cleanup();
    set_error("debug allocation failed: %s", hipGetErrorString(e));
    return VRAG_ERR_HIP;
  }
  {
    std::vector<unsigned short> h(std::max(Tp * H, (size_t)3 * H * H));
    unsigned xs = 777u;
    for (auto& v : h) {   // pseudo-random bf16 in about [-1, 1), as vrag_debug_gemm_ms
      xs = xs * 1664525u + 1013904223u;
      v = (unsigned short)(((xs >> 31) << 15) | (0x3e80u + (((xs >> 20) & 1) << 7)) | ((xs >> 9) & 0x7f));
    }
    (void)hipMemcpy(x, h.data(), Tp * H * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, h.data(), (size_t)3 * H * H * 2, hipMemcpyHostToDevice);
    std::vector<float> f(std::max(Tp, (size_t)kFusedMaxSeq * 32), 0.05f);
    (void)hipMemcpy(mu, f.data(), Tp * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(rstd, f.data(), Tp * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(lns, f.data(), (size_t)3 * H * 4, hipMemcpyHostToDevice);
    std::fill(f.begin(), f.end(), 0.7071f);
    (void)hipMemcpy(cs, f.data(), (size_t)kFusedMaxSeq * 32 * 4, hipMemcpyHostToDevice);
  }
  std::vector<int4> groups((size_t)n_seqs * 8);
  const int n_groups = fused_pack_groups(row.data(), len.data(), 0, n_seqs, groups.data());
  (void)hipMemcpy(d_row, groups.data(), (size_t)n_groups * 8 * sizeof(int4), hipMemcpyHostToDevice);
  QkvAttnParams f{};
  f.x = (const bf16_t*)x;
  f.w = (const bf16_t*)w;
  f.ln_mu = mu;
  f.ln_rstd = rstd;
  f.ln_s = lns;
  f.rope_cos = cs;
  f.rope_sin = cs;
  f.rope_rows = kFusedMaxSeq;
  f.o = (bf16_t*)o;
  f.groups = reinterpret_cast<const int4*>(d_row);
  f.n_groups = n_groups;
  f.H = H;
  f.nh = nh;
  f.Tp = (int)Tp;
  f.window = window;
  f.op_dtype = getenv("VRAG_DEBUG_GEMM_F16") ? kOpF16 : kOpBf16;
  f.q_scale = 0.125f * 1.4426950408889634f;
  f.debug_flags = flags;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  hipError_t le = hipSuccess;
  for (int i = 0; i < 3 && le == hipSuccess; ++i) le = launch_qkv_attention(f, local != 0, 0);
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters && le == hipSuccess; ++i) le = launch_qkv_attention(f, local != 0, 0);
  (void)hipEventRecord(b, 0);
  hipError_t se = hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  cleanup();
  if (le != hipSuccess || se != hipSuccess) {
    set_error("debug fused attention failed: %s", hipGetErrorString(le != hipSuccess ? le : se));
    return VRAG_ERR_HIP;
  }
  *ms_out = ms / iters;
  return VRAG_OK;
}

int vrag_debug_attn_run(int32_t local, int32_t n_seqs, int32_t S, int32_t H, int32_t window, int32_t f16, const uint16_t* q,
                        const uint16_t* k, const uint16_t* vt, uint16_t* o, int32_t device) {
  ARG_CHECK(q && k && vt && o && n_seqs > 0 && S > 0 && S % kSeqAlign == 0 && H % 64 == 0, "bad arguments");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  const size_t T = (size_t)n_seqs * S, Tp = (size_t)align_up((int)T, kRowPad);
  const int qb = attention_q_block(local != 0);
  std::vector<int> bs, bl, bq;
  for (int s = 0; s < n_seqs; ++s)
    for (int q0 = 0; q0 < S; q0 += qb) {
      bs.push_back(s * S);
      bl.push_back(S);
      bq.push_back(q0);
    }
  void *dq = nullptr, *dk = nullptr, *dv = nullptr, *dout = nullptr;
  int *d_bs = nullptr, *d_bl = nullptr, *d_bq = nullptr;
  auto cleanup = [&]() {
    for (void* p : {dq, dk, dv, dout, (void*)d_bs, (void*)d_bl, (void*)d_bq})
      if (p) (void)hipFree(p);
  };
  hipError_t e = hipMalloc(&dq, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&dk, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&dv, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc(&dout, Tp * H * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&d_bs, bs.size() * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_bl, bs.size() * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_bq, bs.size() * 4);
  if (e == hipSuccess) e = hipMemset(dq, 0, Tp * H * 2);
  if (e == hipSuccess) e = hipMemset(dk, 0, Tp * H * 2);
  if (e == hipSuccess) e = hipMemset(dout, 0, Tp * H * 2);
  if (e == hipSuccess) e = hipMemcpy(dq, q, T * H * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dk, k, T * H * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv, vt, (size_t)H * Tp * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_bs, bs.data(), bs.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_bl, bl.data(), bs.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_bq, bq.data(), bs.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) {
    AttnParams ap{};
    ap.q = (const bf16_t*)dq;
    ap.k = (const bf16_t*)dk;
    ap.vt = (const bf16_t*)dv;
    ap.o = (bf16_t*)dout;
    ap.blk_seq_start = d_bs;
    ap.blk_seq_len = d_bl;
    ap.blk_q0 = d_bq;
    ap.n_blocks = (int)bs.size();
    ap.H = H;
    ap.nh = H / 64;
    ap.Tp = (int)Tp;
    ap.window = window;
    ap.op_dtype = f16 ? kOpF16 : kOpBf16;
    e = launch_attention(ap, local != 0, 0);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(o, dout, T * H * 2, hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) {
    set_error("debug attention run failed: %s", hipGetErrorString(e));
    return VRAG_ERR_HIP;
  }
  return VRAG_OK;
}

}  // extern "C"
