// Wqkv GEMM + RoPE + attention fused per (sequence, head) for gfx950 -- ModernBertAttention.forward
// (transformers modeling_modernbert.py:271 Wqkv, :188-219 rotary, :166-185 eager attention, mask masking_utils.py:141-151)
// without the round trip of Q, K and V^T through HBM.
//
// Unfused, a 65 536-token micro-batch writes 300 MB of Q / K / V^T per layer (the QKV GEMM's epilogue: 95 of its 268 us) and
// attention reads them back; that is 26 of the step's 101 GB.  Here one workgroup (8 waves) owns one
// sequence of <= 512 tokens and one head:
//   1. main loop: C^T[192 features][512 tokens] = W_head[192, H] . X_seq[512, H]^T with v_mfma_f32_16x16x32, wave w owns
//      tokens 64 w .. 64 w + 63 and ALL 192 features (192 accumulator registers).  Operands by 16-byte LDS-DMA in 64-k
//      stages of 128-byte rows (16-byte chunk index XOR-swizzled by (row >> 1) & 7, as in gemm_bf16.hip): the weight rows
//      double-buffered and shared (one raw barrier per stage), the token rows in a buffer private to their wave.
//      q and k products are issued "swapped" (lane = token, registers = 4 consecutive features), v
//      un-swapped (lane = feature, registers = 4 consecutive tokens) -- the layouts the attention operands want.
//   2. epilogue in registers: LayerNorm fold, RoPE (the (d, d + 32) partners sit in the same lane), q scaled by
//      head_dim^-1/2 log2 e.  K rows and V^T rows go to LDS (64 KiB each, over the operand buffers), Q stays in registers as
//      the B operand of S^T = K . Q^T.  The k-slot <-> feature map of an accumulator pair, slot j of step s <-> feature
//      (2 s + j / 4) * 16 + 4 g + j % 4 (g = lane >> 4), is used for Q and K alike (a dot product does not care), and the
//      same map over keys pairs P (straight from the S^T accumulators) with V^T: no shuffles, no transposes.
//      Row statistics, fold sums and rotary rows (48 table rows, angle addition) arrive by DMA at kernel start and are read
//      from LDS: no global round trip between the main loop and attention.
//   3. attention: every wave walks the 64-key tiles of its band (global: all of them) on the LDS-resident K / V^T -- no
//      DMA, no barriers, waves run free; softmax in exp2 units with a lazily moved reference that rides into the S^T MFMAs
//      as their C operand, row sums from an all-ones MFMA (first built as a stand-alone second-generation attention kernel in round 3, since folded in here).
//   4. O rows are staged through LDS and stored as whole 128-byte head rows.
// Round 4, wave-slot packing: a workgroup holds a GROUP of consecutive sequences, each on ceil(S / 64) consecutive waves (per-wave
// descriptors: first row, sequence length, first wave of the sequence); K / V^T rows stay indexed by workgroup slot, positions,
// masks and the key-tile walk are relative to the sequence.  A workgroup costs what a full one costs, so the schedule takes the
// kernel when the groups are full enough (kFusedMinFillPct of 512 tokens per workgroup).  A wave's descriptor carries its own first
// row, so the sequences of a group need not be neighbours in the packed buffer: the engine packs a micro-batch's sequences best fit
// decreasing (capi.hip), whatever order they arrived in.
// Sequences longer than 512 tokens, BERT-family encoders, launch-bound batches and poorly filled batches keep the two-kernel path.
#include "qkv_attn.h"

#include <cstdlib>
#include <type_traits>

// Phase-decomposition probes of the fused kernel (vrag_debug_qkv_attn_ms, include/vrag_amd_debug.h): present in the harness build
// only; the product build compiles every one of these branches out.
#ifdef VRAG_DEBUG_API
#define VRAG_DBG(bit) ((p.debug_flags & (bit)) != 0)
#else
#define VRAG_DBG(bit) false
#endif

namespace vrag {

constexpr int QA_WB = 192 * 128;             // one weight stage: 192 rows x 64 k-values
constexpr int QA_XOFF = 2 * QA_WB;           // the waves' private token stages (64 rows x 128 B each) behind the two weight buffers
constexpr int QA_TAB = 131072;               // main loop: 48 + 64 KiB; afterwards K and V^T of the whole sequence: 2 x 64 KiB; then the tables:
constexpr int QA_MU = QA_TAB, QA_RS = QA_TAB + 2048, QA_LS = QA_TAB + 4096;            // row statistics of the 512 tokens, fold sums of the head
constexpr int QA_CA = QA_TAB + 5120, QA_SA = QA_CA + 4096, QA_CB = QA_SA + 4096, QA_SB = QA_CB + 2048;   // rotary rows 16 b (b < 32) and 0 .. 15
constexpr int QA_SMEM = QA_SB + 2048;        // 145 KiB
constexpr int QA_V_OFF = 65536;              // K rows [512][128 B] at 0, V^T rows [64][1024 B] behind them

constexpr float QA_LAZY = 8.0f;   // log2 units: the softmax reference moves when a score exceeds it by more than 2^8
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // one v_max3_f32

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool LOCAL, bool FOLD, typename T>
__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(const QkvAttnParams p) {
  typedef typename Op<T>::v4 V4;
  typedef typename Op<T>::v8 V8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int H = p.H, Tp = p.Tp, nh = p.nh;
  // workgroup b runs on XCD b % 8: the heads of one sequence are dealt to ONE XCD back to back, so its token rows are
  // fetched from HBM once and re-read from that XCD's L2 by the other heads
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = (slot / nh) * 8 + xcd, head = slot % nh;
  if (grp >= p.n_groups) return;
  // Wave-slot packing (round 4): the workgroup holds a GROUP of consecutive sequences, each on ceil(S / 64) consecutive waves.
  // Per wave: the packed row of its first token, its sequence's length and the first wave (= 64-key LDS slot) of its sequence.
  // K / V^T rows stay indexed by workgroup slot (wave * 64 + row); positions, masks and the key-tile walk are relative to the
  // sequence.  A 512-token sequence is the group of one: row0 = first row + 64 w, kbase = 0.
  const int4 wd = p.groups[grp * 8 + wave];
  const int wrow0 = uniform(wd.x), S = uniform(wd.y), kbase = uniform(wd.z);
  const int srow0 = wave * 64;                 // first K / V^T row (LDS slot position) of this wave
  const int qrow0 = (wave - kbase) * 64;       // first token of this wave inside its sequence
  const bool active = qrow0 < S;               // wave-uniform (S = 0: an unused wave): such a wave only helps with the weight DMA
  const T* X = reinterpret_cast<const T*>(p.x);
  const T* Wh = reinterpret_cast<const T*>(p.w) + (size_t)head * 192 * H;

  // ---------------------------------------------------------------- 1. main loop
  // 64-k stages with 128-byte operand rows (whole cache lines per DMA row: 64-byte rows -- a 32-k ring -- halve the L1's
  // effective rate).  Weight rows are shared by the eight waves: two 24 KiB buffers, the next stage's DMA issued right
  // after the barrier that retires the previous one.  Token rows are PRIVATE to their wave (8 KiB each): no barrier guards
  // them -- a wave reads its eight fragments of the stage into registers, waits for those reads, and refills its own buffer
  // for the next stage while the MFMAs of this one run.
  // DMA sources as (uniform base) + (32-bit lane offset): the scalar-base addressing form, one VGPR per distinct lane pattern
  unsigned voffX[2], voffW[3];   // bytes
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int r = par * 8 + (lane >> 3);   // row inside a 16-row block of the wave's 64
    voffX[par] = (unsigned)(r * H + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = wave * 24 + i * 8 + (lane >> 3);
    voffW[i] = (unsigned)(row * H + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u;
  }
  const char* const Xw = reinterpret_cast<const char*>(X + (size_t)wrow0 * H);   // this wave's first token row (uniform)
  const char* const Wb = reinterpret_cast<const char*>(Wh);
  char* const xbuf = smem + QA_XOFF + wave * 8192;
  // one DMA instruction of the next stage: weight rows (i < 3, 8 rows each) or this wave's token rows (i < 8)
  // The lane offset is re-read through an opaque asm at every use: otherwise the eleven 64-bit lane addresses are hoisted out
  // of the K loop as loop invariants -- 22 registers the loop does not have, i.e. spills and reloads in front of every DMA.
  auto dma_w = [&](int kt, int buf, int i) {
    if (VRAG_DBG(4)) return;
    unsigned vo = voffW[i];
    asm volatile("" : "+v"(vo));
    glds16(Wb + kt * 128 + vo, smem + buf * QA_WB + (wave * 24 + i * 8) * 128);
  };
  auto dma_x = [&](int kt, int i) {
    if (!active || VRAG_DBG(4)) return;
    unsigned vo = voffX[i & 1];
    asm volatile("" : "+v"(vo));
    glds16(Xw + ((size_t)(i >> 1) * 16 * H + kt * 64) * 2 + vo, xbuf + i * 1024);
  };

  // Everything the epilogue gathers -- the row statistics of the sequence, the head's fold sums, rotary rows -- comes in by DMA
  // NOW, behind the K / V^T area, and is read from LDS when the main loop is over: no global round trip sits between the main
  // loop and attention.  Rotary rows by angle addition: pos = 16 b + i, cos(pos f) = cos(16 b f) cos(i f) - sin(16 b f) sin(i f):
  // table rows 16 b (b < 32) and rows 0 .. 15 are all a 512-token sequence needs (12 KiB instead of 128).
  if (!VRAG_DBG(4)) {
    if (FOLD && lane < 16) {   // every wave brings the statistics of ITS 64 rows (16 lanes x 4 floats each)
      glds16(p.ln_mu + min(wrow0 + lane * 4, Tp - 4), smem + QA_MU + srow0 * 4);
      glds16(p.ln_rstd + min(wrow0 + lane * 4, Tp - 4), smem + QA_RS + srow0 * 4);
    }
    if (FOLD && wave == 2) glds16(p.ln_s + head * 192 + lane * 4, smem + QA_LS);
    if (wave == 2 || wave == 3) {   // rows 0 .. 15 (8 rows of 128 bytes per instruction)
      const float* tab = wave == 2 ? p.rope_cos : p.rope_sin;
#pragma unroll
      for (int i = 0; i < 2; ++i) {   // 16-byte chunk index XOR (row >> 1) & 7: the epilogue reads one chunk of 16 DIFFERENT rows per lane group
        const int r = i * 8 + (lane >> 3);
        glds16(tab + min(r, p.rope_rows - 1) * 32 + (((lane & 7) ^ ((r >> 1) & 7)) << 2), smem + (wave == 2 ? QA_CB : QA_SB) + i * 1024);   // clamped like the 16 b rows: a handle may hold fewer than 16 positions
      }
    }
    if (wave == 4 || wave == 5) {   // rows 16 b
      const float* tab = wave == 4 ? p.rope_cos : p.rope_sin;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        glds16(tab + (size_t)min(16 * (i * 8 + (lane >> 3)), p.rope_rows - 1) * 32 + (lane & 7) * 4, smem + (wave == 4 ? QA_CA : QA_SA) + i * 1024);
    }
  }

  f32x4 acc[12][4];
#pragma unroll
  for (int a = 0; a < 12; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = H >> 6;
  int fo[2];   // fragment of a 16-row block: row l15, k-values 32 s + 8 g .. + 7
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) fo[s2] = l15 * 128 + ((((4 * s2 + g) ^ ((l15 >> 1) & 7))) << 4);
  // `mm` must not be provably equal to `active`: when the compiler can see that, it threads the main loop's `if (mm)` into the
  // later `if (active)` phases and the register allocation of the non-banded instantiations collapses -- 650 VGPRs in scratch,
  // 3.4 ms per launch instead of 0.3 (round 5: found when the harness build's phase probe `p.debug_flags & 2`, which used to
  // sit here, was compiled out of the product kernel).  The product build keeps the expression's shape with an opaque scalar
  // zero in the probe's place (one s_and + s_cmp per launch).
#ifdef VRAG_DEBUG_API
  const int probe = p.debug_flags;
#else
  int probe = 0;
  asm volatile("" : "+s"(probe));
#endif
  const bool mm = active && !(probe & 2);
#pragma unroll
  for (int i = 0; i < 3; ++i) dma_w(0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_x(0, i);
  for (int kt = 0; kt < KT; ++kt) {
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // stage kt has landed everywhere; every wave is past its reads of weight buffer (kt + 1) & 1
    asm volatile("" ::: "memory");
    const char* sW = smem + (kt & 1) * QA_WB;
    const bool more = kt + 1 < KT;
    V8 xf[2][4], wa;
    if (mm) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) xf[s2][rt] = *reinterpret_cast<const V8*>(xbuf + rt * 16 * 128 + fo[s2]);
      wa = *reinterpret_cast<const V8*>(sW + fo[0]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my token fragments are in registers: the buffer may be refilled
    __builtin_amdgcn_sched_barrier(0);
    if (mm) {
      // Weight fragments are software-pipelined by hand (the read of block i + 1 sits in front of the four MFMAs of block i),
      // and the 11 DMA instructions of the next stage are dealt out one per block behind the barrier, the shared weight rows first
      // (round 3: dealt out instead of one burst, 392 vs 409 us per launch inside the bench step; round 6: one per block with the
      // weight rows first instead of one per two blocks with the token rows first, 279.5 -> 274.7 us alone, step +0.5 %,
      // profiles/r06_fused_dma_order_ab.txt -- a three-slot weight ring issued two stages ahead measured no better than that).
#pragma unroll
      for (int it = 0; it < 24; ++it) {   // it = 12 s + nj
        const int s2 = it / 12, nj = it % 12;
        V8 na = wa;
        if (it < 23) na = *reinterpret_cast<const V8*>(sW + ((it + 1) % 12) * 16 * 128 + fo[(it + 1) / 12]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          if (nj < 8) acc[nj][rt] = Op<T>::mfma16(wa, xf[s2][rt], acc[nj][rt]);   // q, k: lane = token, registers = features
          else acc[nj][rt] = Op<T>::mfma16(xf[s2][rt], wa, acc[nj][rt]);          // v: lane = feature, registers = tokens
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more && it < 11) {   // one DMA instruction per block: the shared weight rows first, then this wave's token rows
          if (it < 3) dma_w(kt + 1, (kt + 1) & 1, it);
          else dma_x(kt + 1, it - 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        wa = na;
      }
    } else if (more) {
#pragma unroll
      for (int i = 0; i < 3; ++i) dma_w(kt + 1, (kt + 1) & 1, i);
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_x(kt + 1, i);
    }
  }
  __syncthreads();   // every wave is past its last operand read: the ring becomes the K / V^T store
  if (VRAG_DBG(16)) return;

  // ---------------------------------------------------------------- 2. epilogue: fold, RoPE, K / V^T -> LDS, Q -> registers
  const float* l_mu = reinterpret_cast<const float*>(smem + QA_MU) + srow0;
  const float* l_rs = reinterpret_cast<const float*>(smem + QA_RS) + srow0;
  const float* l_ls = reinterpret_cast<const float*>(smem + QA_LS);
  // ---- V^T third (un-swapped accumulators: lane = feature d, registers = tokens)
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    const int d = db * 16 + l15;
    const float lsv = FOLD ? l_ls[128 + d] : 0.f;
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
      V8 vv;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int rt = 2 * tp + a;
        f32x4 m4 = {0.f, 0.f, 0.f, 0.f}, r4 = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FOLD) {
          m4 = *reinterpret_cast<const f32x4*>(l_mu + rt * 16 + 4 * g);
          r4 = *reinterpret_cast<const f32x4*>(l_rs + rt * 16 + 4 * g);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // packed fp32, two tokens at a time
          const int r0 = 2 * h;
          const f32x2 v = pk_fma(f32x2{-m4[r0], -m4[r0 + 1]}, splat2(lsv), f32x2{acc[8 + db][rt][r0], acc[8 + db][rt][r0 + 1]}) * f32x2{r4[r0], r4[r0 + 1]};
#pragma unroll
          for (int e = 0; e < 2; ++e)   // keys past the end are masked to probability 0: their V must be a finite number for 0 * v to stay 0
            vv[a * 4 + r0 + e] = qrow0 + rt * 16 + 4 * g + r0 + e < S ? Op<T>::to(v[e]) : (T)0.f;
        }
      }
      const int c = wave * 8 + tp * 4 + g;   // 16-byte chunk of the row: keys 64 w + 32 tp .. + 31, slot order (see top)
      *reinterpret_cast<V8*>(smem + QA_V_OFF + d * 1024 + ((c ^ l15) << 4)) = vv;
    }
  }
  // ---- q and k thirds: rotate accumulators a0 .. a0 + 3 of row tile rt, out[s] = the two 8-value operand fragments
  auto rope_rows = [&](int a0, int rt, int ls_off, float scale, V8 (&out)[2]) {
    const bool live = qrow0 + rt * 16 + l15 < S;
    const float mu = FOLD ? l_mu[rt * 16 + l15] : 0.f, rs = FOLD ? l_rs[rt * 16 + l15] : 1.f;
#pragma unroll
    for (int np = 0; np < 2; ++np) {
      const int dd = np * 16 + 4 * g;
      const f32x4 ca = *reinterpret_cast<const f32x4*>(smem + QA_CA + (((qrow0 >> 4) + rt) * 32 + dd) * 4);
      const f32x4 sa = *reinterpret_cast<const f32x4*>(smem + QA_SA + (((qrow0 >> 4) + rt) * 32 + dd) * 4);
      const int bsw = l15 * 128 + ((((dd >> 2) ^ ((l15 >> 1) & 7))) << 4);
      const f32x4 cb = *reinterpret_cast<const f32x4*>(smem + QA_CB + bsw);
      const f32x4 sb = *reinterpret_cast<const f32x4*>(smem + QA_SB + bsw);
      f32x4 l1 = {0.f, 0.f, 0.f, 0.f}, l2 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (FOLD) {
        l1 = *reinterpret_cast<const f32x4*>(l_ls + ls_off + dd);
        l2 = *reinterpret_cast<const f32x4*>(l_ls + ls_off + 32 + dd);
      }
      // two columns at a time as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 on the accumulators' own register pairs; round 4):
      // this epilogue is VALU-bound, and every operation of it is FMA-class -- same operations, same order as the scalar form
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j0 = 2 * h;
        f32x2 x1 = {acc[a0 + np][rt][j0], acc[a0 + np][rt][j0 + 1]}, x2 = {acc[a0 + 2 + np][rt][j0], acc[a0 + 2 + np][rt][j0 + 1]};
        if constexpr (FOLD) {
          x1 = pk_fma(splat2(-mu), f32x2{l1[j0], l1[j0 + 1]}, x1) * rs;
          x2 = pk_fma(splat2(-mu), f32x2{l2[j0], l2[j0 + 1]}, x2) * rs;
        }
        const f32x2 ca2 = {ca[j0], ca[j0 + 1]}, sa2 = {sa[j0], sa[j0 + 1]}, cb2 = {cb[j0], cb[j0 + 1]}, sb2 = {sb[j0], sb[j0 + 1]};
        const f32x2 c = pk_fma(-sa2, sb2, ca2 * cb2), sn = pk_fma(ca2, sb2, sa2 * cb2);   // cos / sin of (16 b + i) f
        // q*cos + rotate_half(q)*sin, rotate_half = cat(-x2, x1)  (TF:188-219)
        const f32x2 o1 = pk_fma(-x2, sn, x1 * c) * scale, o2 = pk_fma(x1, sn, x2 * c) * scale;
        // rows past the end of the sequence are masked (keys) or never stored (queries); only the fp16 conversion cares,
        // because it reports values it has to clamp
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if constexpr (!std::is_same<T, bf16_t>::value) {
            out[0][np * 4 + j0 + e] = live ? Op<T>::to(o1[e]) : (T)0.f;
            out[1][np * 4 + j0 + e] = live ? Op<T>::to(o2[e]) : (T)0.f;
          } else {
            out[0][np * 4 + j0 + e] = Op<T>::to(o1[e]);
            out[1][np * 4 + j0 + e] = Op<T>::to(o2[e]);
          }
        }
      }
    }
  };
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    V8 kf[2];
    rope_rows(4, rt, 64, 1.0f, kf);
    const int row = srow0 + rt * 16 + l15;   // LDS slot position
#pragma unroll
    for (int s = 0; s < 2; ++s) *reinterpret_cast<V8*>(smem + row * 128 + (((4 * s + g) ^ ((row >> 1) & 7)) << 4)) = kf[s];
  }
  V8 qf[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) rope_rows(0, rt, 0, p.q_scale, qf[rt]);
  __syncthreads();   // K and V^T of the whole sequence are in LDS

  // ---------------------------------------------------------------- 3. attention over the LDS-resident keys
  // The running reference rides into S^T = K . Q^T as the MFMA's C operand,
  // so p = exp2(s) needs no subtraction; the reference moves (cross-lane maximum, rescale) only when a score exceeds it by
  // more than 2^8 or a row meets its first key -- a rare wave-uniform branch; row sums come from the matrix pipe (an all-ones
  // A operand against the very P fragments the P . V product uses).  Per score: half a v_max3, one v_exp, half a packed
  // convert.  A 32-row group (u) is scored at a time: half of the score registers live, K fragments read twice from LDS
  // (no DMA competes for it here).
  f32x4 ot[4][4];   // O^T[d = 16 dt + 4 g + r][query l15] of column block c
  f32x4 lo[4];      // row sums of column block c (every register holds the same value)
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
  const int W = p.window;
  int kt_lo = 0, kt_hi = (S - 1) >> 6;
  if constexpr (LOCAL) {
    kt_lo = max(0, qrow0 - W) >> 6;
    kt_hi = min(S - 1, qrow0 + 63 + W) >> 6;
  }
  const int ksw = (l15 >> 1) & 7;
  if (active && !VRAG_DBG(1)) {
    for (int kt = kt_lo; kt <= kt_hi; ++kt) {
      // 32-row group u against the 32-key half t2 of the tile: `cut` = some element of the 32 x 32 block is outside the band or
      // beyond the sequence (masked element by element; a block wholly outside is simply all -inf); `dead` = the whole 64-key
      // tile is outside for the group (no MFMA, p = 0).  Coarse on purpose: one wave-uniform branch per group, straight-line
      // MFMA runs inside -- a branch per skipped 16 x 16 block (as attention.hip does between barriers) chops the matrix work
      // into pieces the scheduler cannot interleave with the softmax arithmetic.
      bool cut[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q_lo = qrow0 + 32 * u;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int kh = kt * 64 + t2 * 32;
          bool outside = kh >= S || q_lo >= S;
          cut[u][t2] = kh + 31 >= S;
          if constexpr (LOCAL) {
            outside = outside || (kh + 31 < q_lo - W) || (kh > q_lo + 31 + W);
            cut[u][t2] = cut[u][t2] || (kh < q_lo + 31 - W) || (kh + 31 > q_lo + W);
          }
          cut[u][t2] = cut[u][t2] || outside;
        }
      }
      V8 pf[4][2];
      {
        f32x4 st[4][4];   // S^T - reference: lane (l15 = query of column block c, g) holds keys 16 kb + 4 g + r
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float seed = -m_run[c];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) st[c][kb] = f32x4{seed, seed, seed, seed};
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const char* krow = smem + ((kbase + kt) * 64 + kb * 16 + l15) * 128;   // key tile kt of THIS sequence = LDS slot kbase + kt
          const V8 k0 = *reinterpret_cast<const V8*>(krow + ((g ^ ksw) << 4));
          const V8 k1 = *reinterpret_cast<const V8*>(krow + (((4 + g) ^ ksw) << 4));
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            st[c][kb] = Op<T>::mfma16(k0, qf[c][0], st[c][kb]);
            st[c][kb] = Op<T>::mfma16(k1, qf[c][1], st[c][kb]);
          }
        }
        // masks and in-lane maxima (scores are relative to m_run already)
        float mx[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int u = c >> 1;
          mx[c] = -INFINITY;
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            if (cut[u][t2]) {
              const int qi = qrow0 + 16 * c + l15;
              int lo_k = 0, hi_k = S - 1;
              if constexpr (LOCAL) {
                lo_k = max(0, qi - W);
                hi_k = min(S - 1, qi + W);
              }
              const unsigned span = hi_k >= lo_k ? (unsigned)(hi_k - lo_k) : 0u;
              int d0 = hi_k >= lo_k ? kt * 64 + t2 * 32 + 4 * g - lo_k : -100000;
              asm volatile("" : "+v"(d0));   // keep the predicate arithmetic inside this branch
#pragma unroll
              for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const bool ok = (unsigned)(d0 + 16 * h2 + r) <= span;
                  st[c][2 * t2 + h2][r] = ok ? st[c][2 * t2 + h2][r] : -INFINITY;
                }
            }
            const f32x4& a = st[c][2 * t2];
            const f32x4& b2 = st[c][2 * t2 + 1];
            mx[c] = max3f(mx[c], max3f(a[0], a[1], a[2]), max3f(a[3], b2[0], b2[1]));
            mx[c] = max3f(mx[c], b2[2], b2[3]);
          }
        }
        // move the reference?  (wave-uniform; steady state: no)
        bool move = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) move = move || mx[c] > QA_LAZY || (!seen[c] && mx[c] > -INFINITY);
        if (__any(move)) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float m_all = fmaxf(mx[c], __shfl_xor(mx[c], 16, 64));   // over the four lane groups of the query
            m_all = fmaxf(m_all, __shfl_xor(m_all, 32, 64));
            const bool has = m_all > -INFINITY;
            const float delta = seen[c] ? fmaxf(m_all, 0.f) : (has ? m_all : 0.f);
            // a row's first reference may sit far below zero: nothing has been accumulated yet, so nothing is rescaled
            // (2^-delta would overflow); afterwards delta >= 0 and alpha <= 1
            const float alpha = seen[c] ? __builtin_amdgcn_exp2f(-delta) : 1.0f;
            seen[c] = seen[c] || has;
            m_run[c] += delta;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
              for (int r = 0; r < 4; ++r) st[c][kb][r] -= delta;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
              for (int r = 0; r < 4; ++r) ot[c][dt][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo[c][r] *= alpha;
          }
        }
        // p = exp2(s - m) straight into the B-operand fragments of O^T += V^T . P^T: k-slot (g, j) of the 32-key step t2 is the
        // accumulator row j & 3 of key block 2 t2 + (j >> 2)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[c][t2][j] = (T)__builtin_amdgcn_exp2f(st[c][2 * t2 + (j >> 2)][j & 3]);
      }
      // ---- O^T += V^T . P^T, l += 1 . P^T (V^T fragments read once for the four column blocks)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) lo[c] = Op<T>::mfma16(ones, pf[c][t2], lo[c]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const V8 vf = *reinterpret_cast<const V8*>(smem + QA_V_OFF + (dt * 16 + l15) * 1024 + ((((kbase + kt) * 8 + t2 * 4 + g) ^ l15) << 4));
#pragma unroll
          for (int c = 0; c < 4; ++c) ot[c][dt] = Op<T>::mfma16(vf, pf[c][t2], ot[c][dt]);
        }
      }
    }
  }
  __syncthreads();   // every wave is done with K / V^T: the K area becomes the output staging (8 KiB per wave)

  // ---------------------------------------------------------------- 4. normalise, stage, store whole head rows
  if (active) {
    char* stg = smem + wave * 8192;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = rt * 16 + l15;
      const bool live = qrow0 + row < S;
      const float inv = live ? 1.0f / lo[rt][0] : 0.f;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        V4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = live ? Op<T>::to(ot[rt][db][j] * inv) : (T)0.f;
        const int c16 = db * 2 + (g >> 1);
        *reinterpret_cast<V4*>(stg + row * 128 + ((c16 ^ (row & 7)) << 4) + ((g & 1) << 3)) = o;
      }
    }
    // private to the wave: program order + the compiler's lgkmcnt wait order the reads below
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), c16 = lane & 7;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 128 + ((c16 ^ (row & 7)) << 4));
      if (qrow0 + row < S && !VRAG_DBG(8)) store16_nt(reinterpret_cast<T*>(p.o) + (size_t)(wrow0 + row) * H + head * 64 + c16 * 8, v);
    }
  }
}

template <bool LOCAL, bool FOLD, typename T>
static hipError_t launch_t(const QkvAttnParams& p, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn_kernel<LOCAL, FOLD, T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, QA_SMEM);
    if (e != hipSuccess) return e;
    attr = true;
  }
  const int grid = ((p.n_groups + 7) / 8) * 8 * p.nh;
  hipLaunchKernelGGL((qkv_attn_kernel<LOCAL, FOLD, T>), dim3(grid), dim3(512), QA_SMEM, stream, p);
  return hipGetLastError();
}

hipError_t launch_qkv_attention(const QkvAttnParams& p, bool local, hipStream_t stream) {
  if (p.n_groups <= 0) return hipSuccess;
  if (p.H % 64 != 0 || p.H != p.nh * 64 || (size_t)p.Tp * p.H >= (size_t)1 << 31) return hipErrorInvalidValue;
  const bool fold = p.ln_mu != nullptr;
  if (p.op_dtype == kOpF16) {
    if (fold) return local ? launch_t<true, true, f16_t>(p, stream) : launch_t<false, true, f16_t>(p, stream);
    return local ? launch_t<true, false, f16_t>(p, stream) : launch_t<false, false, f16_t>(p, stream);
  }
  if (fold) return local ? launch_t<true, true, bf16_t>(p, stream) : launch_t<false, true, bf16_t>(p, stream);
  return local ? launch_t<true, false, bf16_t>(p, stream) : launch_t<false, false, bf16_t>(p, stream);
}

__global__ void permute_qkv_heads_kernel(const bf16_t* __restrict__ w, const float* __restrict__ s, int H, int nh,
                                         bf16_t* __restrict__ w_out, float* __restrict__ s_out) {
  const int orow = blockIdx.x;   // (head * 3 + part) * 64 + d
  const int head = orow / 192, part = (orow % 192) / 64, d = orow % 64;
  const int src = part * H + head * 64 + d;
  for (int i = threadIdx.x; i < H; i += blockDim.x) w_out[(size_t)orow * H + i] = w[(size_t)src * H + i];
  if (threadIdx.x == 0 && s && s_out) s_out[orow] = s[src];
}

hipError_t permute_qkv_heads(const bf16_t* w, const float* s, int H, int nh, bf16_t* w_out, float* s_out, hipStream_t stream) {
  hipLaunchKernelGGL(permute_qkv_heads_kernel, dim3(3 * H), dim3(256), 0, stream, w, s, H, nh, w_out, s_out);
  return hipGetLastError();
}

int fused_pack_groups(const int* seq_row, const int* seq_len, int seq0, int seq1, int4* out) {
  int n = 0, used = 8;   // waves taken in the current group (8 = none open)
  for (int s = seq0; s < seq1; ++s) {
    const int need = (seq_len[s] + 63) / 64;
    if (used + need > 8) {   // open a new group
      for (int w = 0; w < 8; ++w) out[n * 8 + w] = int4{0, 0, 0, 0};
      ++n;
      used = 0;
    }
    for (int j = 0; j < need; ++j) out[(n - 1) * 8 + used + j] = int4{seq_row[s] + 64 * j, seq_len[s], used, 0};
    used += need;
  }
  return n;
}

unsigned qkv_attn_f16_saturated(bool reset) { return f16_sat_take(reset); }
unsigned* qkv_attn_f16_flag_address() { return f16_sat_flag_address(); }

}  // namespace vrag
