// Brute-force dot-product top-k on gfx950 (dense rows and SPLADE sparse rows) + C ABI.
//
// Replaces what the reference delegates to Milvus (verbatim_rag/vector_stores/milvus_base.py:
// dense `client.search(anns_field="dense_vector")` :239-248, sparse `anns_field="sparse_vector"`
// :250-259; metric types COSINE / IP in milvus_local.py:109-129).  Exact search: the reference's
// IVF_FLAT is approximate, so "recall vs CPU ref" is measured against exact brute force.
//
// Total order everywhere: (score desc, id asc).  A candidate is one u64 key
//   [ orderable(score) : 32 | 0xFFFFFFFF - local_row : 32 ]      (max key == best hit)
//
// Routes as of the end of round 6 (every one pinned to oracle/topk_ref.c by tests/test_topk_gpu.py, test_full_size_gpu.py):
//   dense, bf16 rows
//     1 query                      dense_topk_kernel: one 16-lane group per row, 16-byte loads, per-workgroup lists + merge (HBM-bound)
//     >= 2 queries                 the TILED search (shards >= 4 096 rows, dim % 64 == 0): scores = rows x queries^T on the encoder's GEMM
//                                  kernel (csrc/gemm_bf16.hip, EPI_TOPK: the epilogue keeps only keys above the query's entry threshold);
//                                  first stage = one key per row over a SAMPLE of the shard's 256-row tiles (a whole tile round when its
//                                  keys fit 128 MB) picked over by tiled_select_direct_kernel, later stages whole rounds that append;
//                                  an overflowing candidate buffer flags its query: pass kernels behind the flags (host call) or the
//                                  sliced rescue pass (device-resident / sharded search)
//     other shards, VRAG_TOPK_NO_TILED   4 / 32 queries per pass: dense_topk_kernel, dense_topk_mfma_kernel, dense_topk_mfma2_kernel
//   dense, fp32 rows (the store's default; scores = the oracle's sequential fmaf chain, bit for bit)
//     with the bf16 prefilter image (index dtype 2): candidates from the image with a proven error bound, exact re-score --
//       1-4 queries ONE streaming pass (prefilter_collect[_multi]_kernel), >= 5 the tiled search over the image (collect form up to 256
//       queries, 64-candidate form above), flagged queries re-answered by the gated full scan
//     full scan: dense_topk_exact[2]_kernel (32 queries per pass on the fp32 matrix instruction, the chain's bits)
//   sparse (SELL-64 in groups of four terms; fmaf in the document's term order = the oracle's bits)
//     1 query sparse_topk_kernel (dense query vector in LDS / L2, scattered on the device); >= 2 queries sparse_topk_multi_kernel
//     (8 / 16 queries per pass: LDS term map + weight table of the pass's term union)
//   k > 64: exact pages of 64; topk_merge_*: per-query merges, shard merge behind the all-gather (csrc/comm.hip)
#include "../../include/vrag_amd.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <numeric>
#include <type_traits>
#include <vector>

#include "common.h"
#include "gemm_bf16.h"

namespace vrag {
void set_error(const char* fmt, ...);

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// u64 / orderable / unorderable / make_key: common.h (the tiled batched search builds the same keys in a GEMM epilogue)

// Paged search (k > KMAX): page p+1 only admits keys strictly below the last key of page p; keys are unique per
// (score, row), so the pages are disjoint and their concatenation is the exact top-(pages * KMAX).
__device__ __forceinline__ u64 make_key_below(float s, unsigned row, u64 bound) {
  const u64 key = make_key(s, row);
  return key < bound ? key : 0ull;
}

constexpr int KMAX = 64;        // list length of one device pass
constexpr int KPAGED_MAX = 1024;  // largest k of a search call (ceil(k / KMAX) passes)

// Sorted (descending) insert into list[0..k) held in LDS; called by ONE lane.
__device__ __forceinline__ void insert_key(u64* list, int k, u64 key) {
  if (key <= list[k - 1]) return;
  int i = k - 1;
  while (i > 0 && list[i - 1] < key) {
    list[i] = list[i - 1];
    --i;
  }
  list[i] = key;
}

// Maximum of one u64 per lane over the wave, the same value in every lane: DPP steps inside each row of 16 lanes (quad permutes,
// half-row and row mirrors: VALU only), then the four row maxima through scalar lane reads -- ~40 instructions.  The
// __shfl_xor form is 12 ds_bpermute_b32 with a wait each (~700 cycles), and this sits on the insertion path of every list.
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  u64 o = dpp_u64<0xB1>(v);    // quad_perm [1,0,3,2]
  v = o > v ? o : v;
  o = dpp_u64<0x4E>(v);        // quad_perm [2,3,0,1]
  v = o > v ? o : v;
  o = dpp_u64<0x141>(v);       // row_half_mirror
  v = o > v ? o : v;
  o = dpp_u64<0x140>(v);       // row_mirror
  v = o > v ? o : v;
  u64 m = 0ull;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * r), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * r);
    const u64 x = ((u64)hi << 32) | lo;
    m = x > m ? x : m;
  }
  return m;
}

// One key per lane (0 = none) into a wave-private sorted list of k <= 64 keys (LDS).  The lanes' keys are taken in DESCENDING
// order and the loop ends as soon as the largest one left cannot enter, so a call inserts at most k keys -- not one per
// qualifying lane.  Round 5: during a call the list lives in REGISTERS, entry i in lane i -- an insertion is a ballot (how many
// entries are larger), one DPP shift (wave_shr:1) and a select; the k-th entry is a lane read.  No LDS round trip per inserted
// key (the round-4 form had lane 0 shift the list through LDS: ~10 dependent LDS operations per key, and with one list per
// wave and query and a few hundred documents per list the batched pass spent half its time there).
__device__ __forceinline__ void wave_insert_topk(u64* list, int k, u64 key, int lane) {
  u64 kth = list[k - 1];
  if (!__ballot(key > kth)) return;
  u64 mine = lane < k ? list[lane] : 0ull;
  do {
    const u64 mx = wave_max_u64(key);
    const int pos = __popcll(__ballot(mine > mx));            // entries that stay in front of the new key (lanes >= k hold 0)
    const u64 up = dpp_u64<0x138>(mine);                        // wave_shr:1 -- lane i receives lane i - 1's entry
    mine = lane < pos ? mine : (lane == pos ? mx : up);
    if (key == mx) key = 0ull;                                  // keys are unique (they carry the row id)
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, k - 1), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), k - 1);
    kth = ((u64)hi << 32) | lo;
  } while (__ballot(key > kth));
  if (lane < k) list[lane] = mine;
}

// ------------------------------------------------------------------------------------ dense
constexpr int DQT = 4;        // queries per pass
constexpr int DROWS_MIN = 32;    // rows per workgroup iteration; rows per workgroup = a multiple of this, chosen at launch
constexpr int DENSE_WGS = 1536;  // target grid: 256 CUs x 6 resident workgroups -> one balanced round, no tail

// QT queries per pass (1: query slice lives in registers; 4: in LDS), DIMC = dim/128 16-byte chunks per
// lane per row when known at compile time (0 = runtime loop).  Two rows per 16-lane group are in
// flight per iteration so every lane has 2*DIMC independent 16-byte loads outstanding.
template <bool F32, int QT, int DIMC>
__global__ __launch_bounds__(256) void dense_topk_kernel(const void* __restrict__ rows_v, long long n_rows, int dim,
                                                          const float* __restrict__ queries, int nq, int q0,
                                                          int k, u64* __restrict__ cand, int rows_per_wg,
                                                          const u64* __restrict__ bound) {
  // LDS: queries [QT][dim] fp32, per 16-lane group lists [16 groups][QT][k]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = reinterpret_cast<float*>(smem);
  u64* lists = reinterpret_cast<u64*>(smem + (size_t)QT * dim * sizeof(float));
  const int tid = threadIdx.x;
  const int grp = tid >> 4, gl = tid & 15;  // 16 groups of 16 lanes
  const int nqt = min(QT, nq - q0);
  for (int i = tid; i < QT * dim; i += 256) {
    const int q = i / dim;
    sq[i] = q < nqt ? queries[(size_t)(q0 + q) * dim + (i - q * dim)] : 0.f;
  }
  for (int i = tid; i < 16 * QT * k; i += 256) lists[i] = 0ull;
  __syncthreads();

  constexpr int EPC = F32 ? 4 : 8;       // elements per 16-byte chunk
  constexpr int CSTR = 16 * EPC;         // element stride between a lane's chunks
  // QT == 1 with a compile-time dim: keep this lane's query slice in registers
  float qreg[(QT == 1 && DIMC > 0) ? DIMC * EPC : 1];
  if constexpr (QT == 1 && DIMC > 0) {
#pragma unroll
    for (int i = 0; i < DIMC; ++i)
#pragma unroll
      for (int j = 0; j < EPC; ++j) qreg[i * EPC + j] = sq[gl * EPC + i * CSTR + j];
  }

  // rows_per_wg > 0: this workgroup owns the contiguous range [b*rows_per_wg, +rows_per_wg);
  // rows_per_wg == 0: grid-stride over 32-row blocks (all workgroups sweep the shard front to back together)
  const bool strided = rows_per_wg == 0;
  const long long r_begin = (long long)blockIdx.x * (strided ? DROWS_MIN : rows_per_wg);
  const long long r_end = strided ? n_rows : min(n_rows, r_begin + rows_per_wg);
  u64* mylist = lists + (size_t)grp * QT * k;
  const size_t esz = F32 ? 4 : 2;
  constexpr int RSTEP = QT == 1 ? 32 : 16;  // the LDS-query path keeps one row per group in flight (LDS-bound)
  const long long r_step = strided ? (long long)gridDim.x * DROWS_MIN : RSTEP;
  for (long long r = r_begin + grp; r < r_end; r += r_step) {
    const bool has2 = QT == 1 && r + 16 < r_end;
    const char* row0 = reinterpret_cast<const char*>(rows_v) + (size_t)r * dim * esz + (size_t)gl * 16;
    const char* row1 = has2 ? row0 + (size_t)16 * dim * esz : row0;
    float acc[2][QT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < QT; ++q) acc[u][q] = 0.f;

    auto accumulate = [&](const f32x4& raw, int u, int i) {
      float x[EPC];
      if constexpr (F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = raw[j];
      } else {
        const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (float)v[j];
      }
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        if constexpr (QT == 1 && DIMC > 0) {
          acc[u][0] = fmaf(x[j], qreg[i * EPC + j], acc[u][0]);
        } else {
#pragma unroll
          for (int q = 0; q < QT; ++q) acc[u][q] = fmaf(x[j], sq[q * dim + gl * EPC + i * CSTR + j], acc[u][q]);
        }
      }
    };
    if constexpr (DIMC > 0) {
      f32x4 raw[2][DIMC];
#pragma unroll
      for (int i = 0; i < DIMC; ++i) {
        raw[0][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row0 + (size_t)i * 256));
        if constexpr (QT == 1) raw[1][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row1 + (size_t)i * 256));
      }
#pragma unroll
      for (int i = 0; i < DIMC; ++i) {
        accumulate(raw[0][i], 0, i);
        if constexpr (QT == 1) accumulate(raw[1][i], 1, i);
      }
    } else {
      for (int i = 0; (gl + 16 * i) * EPC < dim; ++i) {  // any dim that is a multiple of 8
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(row0 + (size_t)i * 256);
        accumulate(a0, 0, i);
        if constexpr (QT == 1) {
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(row1 + (size_t)i * 256);
          accumulate(a1, 1, i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < QT; ++q) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc[u][q] += __shfl_xor(acc[u][q], o, 64);
      }
    if (gl == 0) {
      for (int q = 0; q < nqt; ++q) {
        const u64 b = bound ? bound[q0 + q] : ~0ull;
        insert_key(mylist + q * k, k, make_key_below(acc[0][q], (unsigned)r, b));
        if (has2) insert_key(mylist + q * k, k, make_key_below(acc[1][q], (unsigned)(r + 16), b));
      }
    }
  }
  __syncthreads();
  // merge the 16 group lists per query: thread q does a k-way selection (lists are sorted)
  if (tid < nqt) {
    int head[16];
    for (int g = 0; g < 16; ++g) head[g] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + (q0 + tid)) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int g = 0; g < 16; ++g) {
        if (head[g] < k) {
          const u64 v = lists[((size_t)g * QT + tid) * k + head[g]];
          if (v > best) {
            best = v;
            bg = g;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// rows per workgroup of the streaming (non-MFMA) kernels: the shard is cut into <= DENSE_WGS equal contiguous ranges
static int dense_rows_per_wg(long long n) {
  const long long per = (n + DENSE_WGS - 1) / DENSE_WGS;
  return (int)std::max<long long>(DROWS_MIN, (per + DROWS_MIN - 1) / DROWS_MIN * DROWS_MIN);
}

template <bool F32>
static hipError_t dense_launch_pass(const void* rows, long long n, int dim, const float* dq, int nq, int q0, int qt, int k,
                                    u64* cand, int n_wg, hipStream_t st, const u64* bound) {
  const int per = dense_rows_per_wg(n);
  const size_t lds = (size_t)qt * dim * sizeof(float) + (size_t)16 * qt * k * sizeof(u64);
  const int cstr = F32 ? 64 : 128;
  const int dimc = dim % cstr == 0 ? dim / cstr : -1;
#define VRAG_DENSE_CASE(QT_, DC_)                                                                               \
  hipLaunchKernelGGL((dense_topk_kernel<F32, QT_, DC_>), dim3(n_wg), dim3(256), lds, st, rows, n, dim, dq, nq, q0, k, cand, per, bound)
  if (qt == 1) {
    if (dimc == 6) VRAG_DENSE_CASE(1, 6);
    else if (dimc == 3) VRAG_DENSE_CASE(1, 3);
    else if (dimc == 8) VRAG_DENSE_CASE(1, 8);
    else if (dimc == 12) VRAG_DENSE_CASE(1, 12);
    else VRAG_DENSE_CASE(1, 0);
  } else {
    if (dimc == 6) VRAG_DENSE_CASE(4, 6);
    else if (dimc == 3) VRAG_DENSE_CASE(4, 3);
    else if (dimc == 8) VRAG_DENSE_CASE(4, 8);
    else VRAG_DENSE_CASE(4, 0);
  }
#undef VRAG_DENSE_CASE
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------ dense, batched queries
// 32 queries per pass on the matrix cores: the corpus is still read once per pass (HBM-bound) and the
// MFMA is ~10 % busy.  Rows stream through a 5-slot LDS ring (128 rows x 64 dims per slot, LDS-DMA,
// 4 tiles in flight; a register-streamed variant with row-strided loads measured 10x slower); queries
// sit in LDS as bf16 (rounded: the single-query path keeps fp32 queries) with a 16-byte-chunk XOR swizzle.
// D[row][query] = rows . q^T  -> lane (query = lane&31) owns 16 row scores of ITS query per 32-row
// group and keeps a private sorted top-k list in LDS.
constexpr int MQ = 32;           // queries per pass
constexpr int MROWS_WG = 2048;   // rows per workgroup
constexpr int MSLOTS = 5;
constexpr int MKMAX = 16;        // private list length limit (LDS budget)

// `split` (fp32 queries that are not exactly bf16): a pass carries 16 queries, column j holds bf16(q_j) and column
// j + 16 the remainder bf16(q_j - bf16(q_j)); the rows are bf16, so both products are exact and their sum carries the
// query to 16 significant bits -- the scores then agree with the fp32-query oracle to fp32 summation noise.
__global__ __launch_bounds__(256) void dense_topk_mfma_kernel(const bf16_t* __restrict__ rows, long long n_rows, int dim,
                                                               const float* __restrict__ queries, int nq, int q0, int k,
                                                               u64* __restrict__ cand, int split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;                                         // [32][dim] bf16, chunk-swizzled
  char* ring = smem + (size_t)MQ * dim * 2;                // MSLOTS x 16 KiB
  u64* lists = reinterpret_cast<u64*>(ring + MSLOTS * 16384);  // [256 lanes][k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int row_bytes = dim * 2;

  const long long r_begin = (long long)blockIdx.x * MROWS_WG;
  const int n_groups = (int)((min(n_rows, r_begin + MROWS_WG) - r_begin + 127) / 128);
  const int KT = dim / 64;
  const int T = n_groups * KT;

  // LDS-DMA roles; (g, kt) of the next tile to stage are carried incrementally (no divisions in the loop)
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    soff[i] = ((lane & 7) ^ ((row >> 1) & 7)) << 3;
  }
  int sg = 0, skt = 0, st_t = 0;
  auto stage_next = [&]() {
    char* slot = ring + (st_t % MSLOTS) * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const long long gr = min(r_begin + (long long)sg * 128 + row, n_rows - 1);
      glds16(rows + (size_t)gr * dim + skt * 64 + soff[i], slot + (wave * 32 + i * 8) * 128);
    }
    ++st_t;
    if (++skt == KT) {
      skt = 0;
      ++sg;
    }
  };
  for (int t = 0; t < min(T, MSLOTS - 1); ++t) stage_next();

  // queries -> bf16 LDS image (while the first tiles fly): element (j, c) at
  // j*row_bytes + ((c/8) ^ (j & 15))*16 + (c%8)*2
  const int dim4 = dim >> 2;
  const int qpp = split ? MQ / 2 : MQ;   // queries per pass
  for (int i = tid; i < MQ * dim4; i += 256) {
    const int j = i / dim4, c = (i - j * dim4) << 2;
    const int jq = split ? (j & 15) : j;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q0 + jq < nq) v = *reinterpret_cast<const f32x4*>(queries + (size_t)(q0 + jq) * dim + c);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (bf16_t)v[e];
      if (split && j >= 16) o[e] = (bf16_t)(v[e] - (float)o[e]);
    }
    *reinterpret_cast<bf16x4*>(sQ + j * row_bytes + ((((c >> 3) ^ (j & 15))) << 4) + ((c & 7) << 1)) = o;
  }
  u64* mylist = lists + (size_t)tid * k;
  for (int i = 0; i < k; ++i) mylist[i] = 0ull;
  u64 kth = 0ull;
  // retire the plain query loads so hipcc does not drain the DMA ring inside the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // sQ and lists visible

  const int fsw = (l31 >> 1) & 7;
  const char* sB = sQ + l31 * row_bytes;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int g = 0, kt = 0;
  for (int t = 0; t < T; ++t) {
    const int ahead = min(MSLOTS - 2, T - 1 - t);  // tiles allowed to stay in flight
    if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st_t < T) stage_next();
    const char* sA = ring + (t % MSLOTS) * 16384 + (wave * 32 + l31) * 128;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + (((2 * s2 + hi) ^ fsw) << 4));
      const int c16 = kt * 8 + 2 * s2 + hi;
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(sB + ((c16 ^ (l31 & 15)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf, acc, 0, 0, 0);
    }
    if (++kt == KT) {
      kt = 0;
      // 16 row scores of query (q0 + l31): private filtered insertion
      const long long rb = r_begin + (long long)g * 128 + wave * 32 + 4 * hi;
      if (split) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += __shfl_xor(acc[r], 16, 64);   // hi part + remainder part of the same query
      }
      if (q0 + l31 < nq && l31 < qpp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long row = rb + (r & 3) + 8 * (r >> 2);
          if (row < n_rows) {
            const u64 key = make_key(acc[r], (unsigned)row);
            if (key > kth) {
              insert_key(mylist, k, key);
              kth = mylist[k - 1];
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      ++g;
    }
  }
  __syncthreads();
  // per query: 8 sorted lists (4 waves x 2 halves) -> k best
  if (tid < qpp && q0 + tid < nq) {
    int head[8];
    for (int gg = 0; gg < 8; ++gg) head[gg] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + (q0 + tid)) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int gg = 0; gg < 8; ++gg) {
        if (head[gg] < k) {
          const int src_lane = (gg >> 1) * 64 + (gg & 1) * 32 + tid;  // wave gg/2, half gg&1, query column tid
          const u64 v = lists[(size_t)src_lane * k + head[gg]];
          if (v > best) {
            best = v;
            bg = gg;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// Second generation of the batched kernel (dim = 384 / 768 / 1024): the 32 query columns live in REGISTERS as
// MFMA B-operand fragments (KT*4 x 16 bytes per lane; one wave per SIMD, so up to 512 VGPRs are available), which
// frees the whole LDS for the row ring: 8 slots x 16 KiB, six 128-row x 64-dim tiles (96 KiB) in flight per CU
// instead of three -- the kernel is bound by bytes in flight, not by the matrix pipe (~10 % busy).  The shard is
// cut into <= 256 equal ranges (one resident workgroup per CU, one balanced round).
constexpr int M2SLOTS = 8;
template <int KT>
__global__ __launch_bounds__(256, 1) void dense_topk_mfma2_kernel(const bf16_t* __restrict__ rows, long long row_lo,
                                                                   long long n_rows /* = end of this launch's row range */,
                                                                   const float* __restrict__ queries, int nq, int q0, int k,
                                                                   u64* __restrict__ cand, int rows_per_wg,
                                                                   u64* __restrict__ thr, int split) {
  constexpr int DIM = KT * 64;
  const int qpp = split ? MQ / 2 : MQ;   // queries per pass (see dense_topk_mfma_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                               // M2SLOTS x 16 KiB
  u64* lists = reinterpret_cast<u64*>(ring + M2SLOTS * 16384);      // [256 lanes][k]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  // query (q0 + l31) as B-operand fragments: k-slots 8*hi .. 8*hi+7 of step s <-> dims 16*s + 8*hi + j (bf16-rounded)
  bf16x8 qf[KT * 4];
  const int lq = split ? (l31 & 15) : l31;     // query of this lane's column
  {
    const bool live = q0 + lq < nq;
    const float* qrow = queries + (size_t)(live ? q0 + lq : 0) * DIM + 8 * hi;
#pragma unroll
    for (int s = 0; s < KT * 4; ++s) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      if (live) {
        a = *reinterpret_cast<const f32x4*>(qrow + 16 * s);
        b = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qf[s][j] = (bf16_t)a[j];
        qf[s][4 + j] = (bf16_t)b[j];
        if (split && l31 >= 16) {   // remainder column
          qf[s][j] = (bf16_t)(a[j] - (float)qf[s][j]);
          qf[s][4 + j] = (bf16_t)(b[j] - (float)qf[s][4 + j]);
        }
      }
    }
  }
  // retire the plain loads before any LDS-DMA is in flight (a later compiler-inserted vmcnt(0) would drain the ring)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < KT * 4; ++s) asm volatile("" : "+v"(qf[s]));

  const long long r_begin = row_lo + (long long)blockIdx.x * rows_per_wg;
  const long long r_end = min(n_rows, r_begin + rows_per_wg);
  const int n_groups = (int)((r_end - r_begin + 127) / 128);
  const int T = n_groups * KT;

  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    soff[i] = ((lane & 7) ^ ((row >> 1) & 7)) << 3;
  }
  int sg = 0, skt = 0, st_t = 0;
  auto stage_next = [&]() {
    char* slot = ring + (st_t & (M2SLOTS - 1)) * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const long long gr = min(r_begin + (long long)sg * 128 + row, n_rows - 1);
      glds16(rows + (size_t)gr * DIM + skt * 64 + soff[i], slot + (wave * 32 + i * 8) * 128);
    }
    ++st_t;
    if (++skt == KT) {
      skt = 0;
      ++sg;
    }
  };
  for (int t = 0; t < min(T, M2SLOTS - 1); ++t) stage_next();

  u64* mylist = lists + (size_t)tid * k;
  for (int i = 0; i < k; ++i) mylist[i] = 0ull;
  u64 kth = 0ull;
  __syncthreads();

  const int fsw = (l31 >> 1) & 7;
  int t = 0;
  for (int g = 0; g < n_groups; ++g) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      // tile t landed (this wave's share); up to M2SLOTS-2 later tiles (4 DMA instructions each) stay in flight
      const int ahead = min(M2SLOTS - 2, T - 1 - t);
      if (ahead >= 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (ahead == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else if (ahead == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (st_t < T) stage_next();   // into the slot of tile t-1, which every wave finished before the barrier
      const char* sA = ring + (t & (M2SLOTS - 1)) * 16384 + (wave * 32 + l31) * 128;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + (((2 * s2 + hi) ^ fsw) << 4));
        if (s2 & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[kt * 4 + s2], acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, qf[kt * 4 + s2], acc0, 0, 0, 0);
      }
      ++t;
    }
    // 16 row scores of query (q0 + l31): private filtered insertion.  The entry threshold is shared through
    // thr[query] (global, atomicMax of every FULL list's k-th key, all workgroups): a key below the k-th key of
    // any list of its query cannot be in the query's top-k, so the union of the lists still contains it, while
    // the number of (divergent, LDS read-modify-write) insertions drops from ~k ln(n) per list to per shard.
    const long long rb = r_begin + (long long)g * 128 + wave * 32 + 4 * hi;
    if (split) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] += acc1[r];
        acc1[r] = 0.f;
        acc0[r] += __shfl_xor(acc0[r], 16, 64);   // hi part + remainder part of the same query
      }
    }
    if (q0 + l31 < nq && l31 < qpp) {
      const u64 shared = __hip_atomic_load(thr + q0 + l31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      u64 bar = shared > kth ? shared : kth;
      bool changed = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = rb + (r & 3) + 8 * (r >> 2);
        if (row < r_end) {
          const u64 key = make_key(acc0[r] + acc1[r], (unsigned)row);
          if (key > bar) {
            insert_key(mylist, k, key);
            kth = mylist[k - 1];
            bar = kth > bar ? kth : bar;
            changed = true;
          }
        }
      }
      if (changed && kth > shared) atomicMax(reinterpret_cast<unsigned long long*>(thr + q0 + l31), (unsigned long long)kth);
    }
  }
  __syncthreads();
  // per query: 8 sorted lists (4 waves x 2 halves) -> k best
  if (tid < qpp && q0 + tid < nq) {
    int head[8];
    for (int gg = 0; gg < 8; ++gg) head[gg] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + (q0 + tid)) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int gg = 0; gg < 8; ++gg) {
        if (head[gg] < k) {
          const int src_lane = (gg >> 1) * 64 + (gg & 1) * 32 + tid;
          const u64 v = lists[(size_t)src_lane * k + head[gg]];
          if (v > best) {
            best = v;
            bg = gg;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// ------------------------------------------------------------------------------------ dense, fp32 rows, bit-exact
// fp32 rows x fp32 queries on the fp32 matrix instruction v_mfma_f32_32x32x2_f32: its result is, bit for bit, the k-ordered
// chain  D = fma(a_k1, b_k1, fma(a_k0, b_k0, C))  (cdna_hip_programming.md section 3), i.e. exactly the sequential
// `acc = fmaf(x[c], q[c], acc)` of the CPU restatement when instruction t carries dims (2t, 2t+1).  Scores -- and
// therefore the top-k order -- equal oracle/topk_ref.c on ARBITRARY data, for every batch size, not only on
// dyadic-grid data where any summation order is exact.  At 64 FLOP/clk/SIMD (157 TFLOP/s) a pass of 32 queries over
// N x 768 fp32 rows needs 0.31 us of matrix time per 1000 rows against 0.49 us of HBM time (6.3 TB/s): still a
// streaming kernel.  One wave per SIMD; rows arrive as 128-row x 32-dim tiles (16 KiB) by LDS-DMA, the 32 query rows sit
// in LDS as fp32 (row stride dim*4 + 16 bytes: conflict-free ds_read_b128), both operands are read 16 bytes at a time
// and feed two MFMAs each (lane (i, kk) takes elements kk and 2 + kk).  Every lane keeps the top-KL keys of its
// (query, 16-row stripe) in REGISTERS (branch-free insertion network, entered only by keys above the shared threshold).
constexpr int XQ = 32;      // queries per pass
constexpr int XKL = 16;     // register list length (k <= 16)
template <int NSLOT>
__global__ __launch_bounds__(256, 1) void dense_topk_exact_kernel(const float* __restrict__ rows, long long row_lo,
                                                                   long long n_rows /* end of this launch's row range */,
                                                                   int dim, const float* __restrict__ queries, int nq, int q0,
                                                                   int k, u64* __restrict__ cand, int rows_per_wg,
                                                                   u64* __restrict__ thr, const unsigned* __restrict__ gate) {
  if (gate) {   // behind the prefilter route of a device-resident search: only query groups with a flagged query are re-answered
    bool any = false;
    for (int i = 0; i < XQ && q0 + i < nq; ++i) any = any || gate[q0 + i] != 0u;
    if (!any) return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int qstride = dim * 4 + 16;                                  // bytes between query rows in LDS
  char* sQ = smem;
  char* ring = smem + ((XQ * qstride + 1023) / 1024) * 1024;          // NSLOT x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;

  const long long r_begin = row_lo + (long long)blockIdx.x * rows_per_wg;
  const long long r_end = min(n_rows, r_begin + rows_per_wg);
  const int n_groups = (int)((r_end - r_begin + 127) / 128);
  const int KT = dim >> 5;
  const int T = n_groups * KT;

  // queries -> LDS (plain loads, retired before any LDS-DMA is in flight)
  for (int i = tid; i < XQ * (dim >> 2); i += 256) {
    const int j = i / (dim >> 2), c = (i - j * (dim >> 2)) << 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q0 + j < nq) v = *reinterpret_cast<const f32x4*>(queries + (size_t)(q0 + j) * dim + c);
    *reinterpret_cast<f32x4*>(sQ + j * qstride + c * 4) = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int soff[4];   // element offset of this lane's 16-byte chunk inside a tile row (source-side swizzle)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + (lane >> 3);
    soff[i] = ((lane & 7) ^ ((row >> 1) & 7)) << 2;
  }
  int sg = 0, skt = 0, st_t = 0, st_slot = 0;
  auto stage_next = [&]() {
    char* slot = ring + st_slot * 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 32 + i * 8 + (lane >> 3);
      const long long gr = min(r_begin + (long long)sg * 128 + row, n_rows - 1);
      glds16(rows + (size_t)gr * dim + skt * 32 + soff[i], slot + (wave * 32 + i * 8) * 128);
    }
    ++st_t;
    st_slot = st_slot + 1 == NSLOT ? 0 : st_slot + 1;
    if (++skt == KT) {
      skt = 0;
      ++sg;
    }
  };
  for (int t = 0; t < min(T, NSLOT - 1); ++t) stage_next();

  u64 lst[XKL];
#pragma unroll
  for (int i = 0; i < XKL; ++i) lst[i] = 0ull;
  u64 kth = 0ull;
  __syncthreads();   // sQ visible

  const int fsw = (l31 >> 1) & 7;
  const char* qrow = sQ + l31 * qstride;
  int t = 0, slot_r = 0;
  for (int g = 0; g < n_groups; ++g) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kt = 0; kt < KT; ++kt) {
      // tile t landed (this wave's share); up to NSLOT-2 later tiles (4 DMA instructions each) stay in flight
      const int ahead = min(NSLOT - 2, T - 1 - t);
      if (ahead >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (st_t < T) stage_next();   // into the slot of tile t-1, which every wave finished before the barrier
      const char* sA = ring + slot_r * 16384 + (wave * 32 + l31) * 128;
      const char* sB = qrow + kt * 128;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(sA + ((u ^ fsw) << 4));
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + (u << 4));
        // instruction 2u carries dims (4u, 4u+1), instruction 2u+1 dims (4u+2, 4u+3): ascending, like the oracle's loop
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a4[1] : a4[0], hi ? b4[1] : b4[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? a4[3] : a4[2], hi ? b4[3] : b4[2], acc, 0, 0, 0);
      }
      ++t;
      slot_r = slot_r + 1 == NSLOT ? 0 : slot_r + 1;
    }
    // 16 row scores of query (q0 + l31): filtered insertion (threshold shared through thr[], see dense_topk_mfma2_kernel)
    const long long rb = r_begin + (long long)g * 128 + wave * 32 + 4 * hi;
    if (q0 + l31 < nq) {
      const u64 shared = __hip_atomic_load(thr + q0 + l31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      u64 bar = shared > kth ? shared : kth;
      bool changed = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = rb + (r & 3) + 8 * (r >> 2);
        const u64 key = row < r_end ? make_key(acc[r], (unsigned)row) : 0ull;
        if (key > bar) {
          u64 cur = key;   // branch-free sorted insertion: every slot keeps the larger of (itself, the carried key)
#pragma unroll
          for (int i = 0; i < XKL; ++i) {
            const u64 a = lst[i];
            const bool up = cur > a;
            lst[i] = up ? cur : a;
            cur = up ? a : cur;
          }
          kth = k == XKL ? lst[XKL - 1] : 0ull;
#pragma unroll
          for (int i = 0; i < XKL - 1; ++i)
            if (i == k - 1) kth = lst[i];
          bar = kth > bar ? kth : bar;
          changed = true;
        }
      }
      if (changed && kth > shared) atomicMax(reinterpret_cast<unsigned long long*>(thr + q0 + l31), (unsigned long long)kth);
    }
  }
  __syncthreads();   // every wave is done with the ring: reuse it for the per-lane lists
  u64* lists = reinterpret_cast<u64*>(ring);   // [256 lanes][k]
#pragma unroll
  for (int i = 0; i < XKL; ++i)
    if (i < k) lists[(size_t)tid * k + i] = lst[i];
  __syncthreads();
  // per query: 8 sorted lists (4 waves x 2 halves) -> k best
  if (tid < XQ && q0 + tid < nq) {
    int head[8];
    for (int gg = 0; gg < 8; ++gg) head[gg] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + (q0 + tid)) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int gg = 0; gg < 8; ++gg) {
        if (head[gg] < k) {
          const int src_lane = (gg >> 1) * 64 + (gg & 1) * 32 + tid;
          const u64 v = lists[(size_t)src_lane * k + head[gg]];
          if (v > best) {
            best = v;
            bg = gg;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// Second form of the bit-exact kernel for dim = 384 / 768 (KT = dim / 32): the query operands live in REGISTERS -- lane
// (query j, k-parity kk) holds the dim/2 elements of its query with that parity, 384 VGPRs at dim 768; one wave per SIMD
// owns the whole 512-entry register file -- so the LDS is all row ring.  The ring is private per wave (8 slots of one
// 32-row x 32-dim tile = 4 KiB): a wave reads only rows it staged itself, its own counted vmcnt orders DMA -> ds_read
// (MI355X_MICROARCH.md: only OTHER waves' reads need a barrier), and the main loop has no s_barrier at all; 7 tiles =
// 28 KiB per wave (112 KiB per CU) are in flight.  Same arithmetic, same results as dense_topk_exact_kernel.
constexpr int X2SLOTS = 8;
template <int KT>
__global__ __launch_bounds__(256, 1) void dense_topk_exact2_kernel(const float* __restrict__ rows, long long row_lo,
                                                                    long long n_rows /* end of this launch's row range */,
                                                                    const float* __restrict__ queries, int nq, int q0, int k,
                                                                    u64* __restrict__ cand, int rows_per_wg,
                                                                    u64* __restrict__ thr, const unsigned* __restrict__ gate) {
  if (gate) {   // behind the prefilter route of a device-resident search: only query groups with a flagged query are re-answered
    bool any = false;
    for (int i = 0; i < XQ && q0 + i < nq; ++i) any = any || gate[q0 + i] != 0u;
    if (!any) return;
  }
  constexpr int DIM = KT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  char* ring = smem + wave * (X2SLOTS * 4096);                         // this wave's slots
  u64* lists = reinterpret_cast<u64*>(smem + 4 * X2SLOTS * 4096);      // [256 lanes][k]

  // query (q0 + l31): elements of parity kk = hi, in the order the MFMAs consume them
  float qreg[KT][16];
  {
    const bool live = q0 + l31 < nq;
    const float* qrow = queries + (size_t)(live ? q0 + l31 : 0) * DIM;
    // one 32-dim slice at a time, retired before the next: 8 x 16-byte loads in flight, not 8 KT of them (the
    // compiler would otherwise hoist every load of the unrolled loop and spill the whole query)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(qrow + kt * 32 + 4 * u);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        qreg[kt][2 * u] = live ? (hi ? v[u][1] : v[u][0]) : 0.f;
        qreg[kt][2 * u + 1] = live ? (hi ? v[u][3] : v[u][2]) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(qreg[kt][i]));
    }
  }

  const long long r_begin = row_lo + (long long)blockIdx.x * rows_per_wg;
  const long long r_end = min(n_rows, r_begin + rows_per_wg);
  const int n_groups = (int)((r_end - r_begin + 127) / 128);
  const int T = n_groups * KT;

  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    soff[i] = ((lane & 7) ^ ((row >> 1) & 7)) << 2;
  }
  int sg = 0, skt = 0, st_t = 0;
  auto stage_next = [&]() {
    char* slot = ring + (st_t & (X2SLOTS - 1)) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 8 + (lane >> 3);
      const long long gr = min(r_begin + (long long)sg * 128 + wave * 32 + row, n_rows - 1);
      glds16(rows + (size_t)gr * DIM + skt * 32 + soff[i], slot + i * 1024);
    }
    ++st_t;
    if (++skt == KT) {
      skt = 0;
      ++sg;
    }
  };
  for (int t = 0; t < min(T, X2SLOTS - 1); ++t) stage_next();

  u64* mylist = lists + (size_t)tid * k;
  for (int i = 0; i < k; ++i) mylist[i] = 0ull;
  u64 kth = 0ull;
  const int fsw = (l31 >> 1) & 7;
  // wait until this wave's tile `tt` has landed: up to `ahead` younger tiles (4 DMA instructions each) may stay in flight
  auto wait_tile = [&](int tt) {
    const int ahead = min(st_t - 1 - tt, X2SLOTS - 1);
    if (ahead >= 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (ahead == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (ahead == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // (Measured r2i: a two-deep register pipeline over half tiles -- next half's reads under the current half's MFMAs, each
  // element selected right in front of its MFMA -- ran 1.76 ms per pass against 0.93 ms for this plain form: the
  // select -> MFMA-operand dependency stalls cost more than the exposed LDS latency.)
  int t = 0;
  for (int g = 0; g < n_groups; ++g) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      wait_tile(t);
      const char* sA = ring + (t & (X2SLOTS - 1)) * 4096 + l31 * 128;
      float a2[16];   // this lane's 16 row elements of the tile (parity kk), selected out of eight 16-byte reads
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sA + ((u ^ fsw) << 4));
        a2[2 * u] = hi ? v[1] : v[0];
        a2[2 * u + 1] = hi ? v[3] : v[2];
      }
      // the slot of tile t-1 is free: its reads were consumed by the MFMAs issued in the previous iteration
      if (st_t < T) stage_next();
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[i], qreg[kt][i], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);   // keep later tiles' reads from being hoisted over this one (384 live query registers)
      ++t;
    }
    const long long rb = r_begin + (long long)g * 128 + wave * 32 + 4 * hi;
    if (q0 + l31 < nq) {
      const u64 shared = __hip_atomic_load(thr + q0 + l31, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      u64 bar = shared > kth ? shared : kth;
      bool changed = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = rb + (r & 3) + 8 * (r >> 2);
        if (row < r_end) {
          const u64 key = make_key(acc[r], (unsigned)row);
          if (key > bar) {
            insert_key(mylist, k, key);
            kth = mylist[k - 1];
            bar = kth > bar ? kth : bar;
            changed = true;
          }
        }
      }
      if (changed && kth > shared) atomicMax(reinterpret_cast<unsigned long long*>(thr + q0 + l31), (unsigned long long)kth);
    }
  }
  __syncthreads();
  // per query: 8 sorted lists (4 waves x 2 halves) -> k best
  if (tid < XQ && q0 + tid < nq) {
    int head[8];
    for (int gg = 0; gg < 8; ++gg) head[gg] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + (q0 + tid)) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int gg = 0; gg < 8; ++gg) {
        if (head[gg] < k) {
          const int src_lane = (gg >> 1) * 64 + (gg & 1) * 32 + tid;
          const u64 v = lists[(size_t)src_lane * k + head[gg]];
          if (v > best) {
            best = v;
            bg = gg;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// thr[q] = k-th key of query q's exact top-k over a prefix of the shard: a lower bound of the k-th key over the
// whole shard, so the main pass may discard everything below it.
__global__ void topk_seed_threshold_kernel(const u64* __restrict__ out, int nq, int k, u64* __restrict__ thr) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) thr[q] = out[(size_t)q * k + (k - 1)];
}

static hipError_t launch_topk_merge(const u64* cand, int n_wg, int nq, int k, u64* out, hipStream_t st);

struct Mfma2Plan {
  long long prefix;   // rows of the threshold-seeding pass (0 = single pass)
  int n_wg0, per1, n_wg1;
};
static Mfma2Plan dense_mfma2_plan(long long n) {
  constexpr int target = 256;           // workgroups of the main pass (one per CU)
  constexpr long long pre = 32768;      // rows of the threshold-seeding prefix pass
  Mfma2Plan p{};
  p.prefix = (pre > 0 && n >= 4 * pre) ? pre / 128 * 128 : 0;
  p.n_wg0 = (int)(p.prefix / 128);
  const long long rest = n - p.prefix;
  const long long per = (rest + target - 1) / target;
  p.per1 = (int)std::max<long long>(128, (per + 127) / 128 * 128);
  p.n_wg1 = (int)((rest + p.per1 - 1) / p.per1);
  return p;
}

static bool dense_use_mfma2(int dim) {
  return dim == 384 || dim == 768 || dim == 1024;
}
static bool dense_use_exact(int dtype, int dim, int k) {
  static const bool off = getenv("VRAG_TOPK_NO_EXACT") != nullptr;   // tuning: fp32 rows on the scalar kernels
  return !off && dtype == 1 && dim % 32 == 0 && dim <= 768 && k <= XKL;
}
static bool dense_use_mfma(int dtype, int dim, int nq, int k) {
  return dtype == 0 && dim % 128 == 0 && nq >= 3 && k <= MKMAX &&
         (size_t)MQ * dim * 2 + MSLOTS * 16384 + (size_t)256 * k * 8 <= 160 * 1024;
}
static int dense_n_wg(int dtype, int dim, int nq, int k, long long size) {
  if (dense_use_exact(dtype, dim, k) || (dense_use_mfma(dtype, dim, nq, k) && dense_use_mfma2(dim))) {
    const Mfma2Plan pl = dense_mfma2_plan(size);
    return std::max(1, pl.n_wg0 + pl.n_wg1);
  }
  const int per = dense_use_mfma(dtype, dim, nq, k) ? MROWS_WG : dense_rows_per_wg(size);
  return (int)std::max<long long>(1, (size + per - 1) / per);
}

// all passes of one search: query tiles of 4 (a final tile of 1 query uses the register path)
static hipError_t dense_launch_all(int dtype, const void* rows, long long n, int dim, const float* dq, int nq, int k,
                                   u64* cand, int n_wg, hipStream_t st, u64* thr, u64* out, const u64* bound = nullptr,
                                   int split = 0, const unsigned* gate = nullptr) {
  const int qpp = split ? MQ / 2 : MQ;
  if (bound && dense_use_mfma(dtype, dim, nq, k)) return hipErrorInvalidValue;   // pages run with k = KMAX: scalar path only
  if (!bound && dense_use_exact(dtype, dim, k)) {
    hipError_t me = hipMemsetAsync(thr, 0, (size_t)nq * sizeof(u64), st);
    if (me != hipSuccess) return me;
    const int qbytes = ((XQ * (dim * 4 + 16) + 1023) / 1024) * 1024;
    const bool deep = qbytes + 6 * 16384 <= 160 * 1024;               // dim <= 384: six ring slots, else three
    const size_t ldsx = (size_t)qbytes + (deep ? 6 : 3) * 16384;
    static bool attrx = false;
    if (!attrx) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_exact_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_exact_kernel<6>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attrx = true;
    }
    const float* r32 = reinterpret_cast<const float*>(rows);
    const bool regq = dim == 768 || dim == 384;   // queries in registers; other dims keep them in LDS
    const size_t lds2 = (size_t)4 * X2SLOTS * 4096 + (size_t)256 * k * 8;
    if (regq) {
      static bool attrx2 = false;
      if (!attrx2) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_exact2_kernel<24>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_exact2_kernel<12>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attrx2 = true;
      }
    }
    auto pass = [&](long long lo, long long hi, int per, int wgs, u64* cand_base) -> hipError_t {
      for (int q0 = 0; q0 < nq; q0 += XQ) {
        if (regq && dim == 768) hipLaunchKernelGGL(dense_topk_exact2_kernel<24>, dim3(wgs), dim3(256), lds2, st, r32, lo, hi, dq, nq, q0, k, cand_base, per, thr, gate);
        else if (regq) hipLaunchKernelGGL(dense_topk_exact2_kernel<12>, dim3(wgs), dim3(256), lds2, st, r32, lo, hi, dq, nq, q0, k, cand_base, per, thr, gate);
        else if (deep) hipLaunchKernelGGL(dense_topk_exact_kernel<6>, dim3(wgs), dim3(256), ldsx, st, r32, lo, hi, dim, dq, nq, q0, k, cand_base, per, thr, gate);
        else hipLaunchKernelGGL(dense_topk_exact_kernel<3>, dim3(wgs), dim3(256), ldsx, st, r32, lo, hi, dim, dq, nq, q0, k, cand_base, per, thr, gate);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    };
    const Mfma2Plan pl = dense_mfma2_plan(n);
    if (pl.prefix > 0) {   // seeding pass: exact top-k of the first rows -> per-query entry threshold for the main pass
      hipError_t e = pass(0, pl.prefix, 128, pl.n_wg0, cand);
      if (e == hipSuccess) e = launch_topk_merge(cand, pl.n_wg0, nq, k, out, st);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(topk_seed_threshold_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, out, nq, k, thr);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    return pass(pl.prefix, n, pl.per1, pl.n_wg1, cand + (size_t)pl.n_wg0 * nq * k);
  }
  if (dense_use_mfma(dtype, dim, nq, k) && dense_use_mfma2(dim)) {
    hipError_t me = hipMemsetAsync(thr, 0, (size_t)nq * sizeof(u64), st);
    if (me != hipSuccess) return me;
    const size_t lds2 = (size_t)M2SLOTS * 16384 + (size_t)256 * k * 8;
    static bool attr2 = false;
    if (!attr2) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_mfma2_kernel<6>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_mfma2_kernel<12>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_mfma2_kernel<16>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attr2 = true;
    }
    const bf16_t* r16 = reinterpret_cast<const bf16_t*>(rows);
    auto pass = [&](long long lo, long long hi, int per, int wgs, u64* cand_base) -> hipError_t {
      for (int q0 = 0; q0 < nq; q0 += qpp) {
        if (dim == 384) hipLaunchKernelGGL(dense_topk_mfma2_kernel<6>, dim3(wgs), dim3(256), lds2, st, r16, lo, hi, dq, nq, q0, k, cand_base, per, thr, split);
        else if (dim == 768) hipLaunchKernelGGL(dense_topk_mfma2_kernel<12>, dim3(wgs), dim3(256), lds2, st, r16, lo, hi, dq, nq, q0, k, cand_base, per, thr, split);
        else hipLaunchKernelGGL(dense_topk_mfma2_kernel<16>, dim3(wgs), dim3(256), lds2, st, r16, lo, hi, dq, nq, q0, k, cand_base, per, thr, split);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    };
    const Mfma2Plan pl = dense_mfma2_plan(n);
    if (pl.prefix > 0) {
      // seeding pass: exact top-k of the first `prefix` rows -> per-query entry threshold for the main pass
      hipError_t e = pass(0, pl.prefix, 128, pl.n_wg0, cand);
      if (e == hipSuccess) e = launch_topk_merge(cand, pl.n_wg0, nq, k, out, st);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(topk_seed_threshold_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, out, nq, k, thr);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    } else if (pl.prefix > 0) {
      hipError_t e = pass(0, pl.prefix, 128, pl.n_wg0, cand);
      if (e != hipSuccess) return e;
    }
    return pass(pl.prefix, n, pl.per1, pl.n_wg1, cand + (size_t)pl.n_wg0 * nq * k);
  }
  if (dense_use_mfma(dtype, dim, nq, k)) {
    const size_t lds = (size_t)MQ * dim * 2 + MSLOTS * 16384 + (size_t)256 * k * 8;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_topk_mfma_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      attr = true;
    }
    for (int q0 = 0; q0 < nq; q0 += qpp) {
      hipLaunchKernelGGL(dense_topk_mfma_kernel, dim3(n_wg), dim3(256), lds, st, reinterpret_cast<const bf16_t*>(rows), n,
                         dim, dq, nq, q0, k, cand, split);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  for (int q0 = 0; q0 < nq;) {
    const int qt = (nq - q0 == 1) ? 1 : DQT;
    hipError_t e = dtype == 0 ? dense_launch_pass<false>(rows, n, dim, dq, nq, q0, qt, k, cand, n_wg, st, bound)
                              : dense_launch_pass<true>(rows, n, dim, dq, nq, q0, qt, k, cand, n_wg, st, bound);
    if (e != hipSuccess) return e;
    q0 += qt;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------ dense, tiled batched search
// Batches over bf16 rows (>= 33 queries, >= 17 when fp32 queries ride as column pairs, >= 3 on the prefilter image of an fp32 index) (SURVEY 8d "Dense top-k, Q-query batch": MFMA-bound once the batch is large, the shard's
// bytes read ONCE per batch): the scores are a GEMM  rows [N, dim] x queries [Q, dim]^T  on the encoder's own kernel
// (csrc/gemm_bf16.hip, 256 x 256 tiles) with the EPI_TOPK epilogue -- nothing is stored but the (score, row) keys above each
// query's entry threshold, appended to a per-query candidate buffer.  The passes of 32 queries above re-stream the shard
// Q / 32 times (256 queries over 1.25 M x 768 rows: 8 passes; 10 240 queries: 320).
// The shard is walked in stages of geometrically growing row ranges [0, 256), [256, 4096), [4096, 65536), ...: after each stage
// one workgroup per query sorts that query's candidates, keeps the best k and publishes the k-th key as the next stage's
// threshold.  With rows in random order a stage that multiplies the rows seen by 16 admits ~15 k candidates per query
// (k ln 16 if the threshold moved inside the stage; it does not); the buffer holds 2048.  A query whose buffer overflows
// (rows sorted by similarity to it) is flagged and re-answered by dense_tiled_rescue_kernel, which walks the shard for that
// query alone -- exactness never depends on the order of the rows.
constexpr int TCAP = 2048;        // candidate slots per query and stage
constexpr int TSTAGE0 = 4096;     // least rows of the first stage (every row is a candidate: one key per row in a buffer of its own, no counters)
constexpr int TDIRECT_KEYS = 1 << 24;   // budget of that buffer in keys (128 MB: a whole tile round for every power-of-two batch up to 4 096 queries): the first stage takes up to min(one tile round, this / queries) rows
constexpr int TRATIO = 16;        // growth of the rows seen per stage (k <= 16)
constexpr int TRATIO_WIDE = 4;    // the same for longer lists

// fp32 queries -> the GEMM's W operand [n_cols_pad, dim] bf16.  pairs: rows (2q, 2q + 1) = (bf16(q), bf16(q - bf16(q)));
// rows beyond the queries are zero.
__global__ void tiled_queries_kernel(const float* __restrict__ q, int nq, int dim, int pairs, int n_cols_pad, bf16_t* __restrict__ w) {
  const int r = blockIdx.x;
  if (r >= n_cols_pad) return;
  const int qi = pairs ? r >> 1 : r;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    float v = qi < nq ? q[(size_t)qi * dim + c] : 0.f;
    if (pairs && (r & 1)) v -= (float)(bf16_t)v;
    w[(size_t)r * dim + c] = (bf16_t)v;
  }
}

// One workgroup per query: the candidates of the stage just finished (cnt[q] keys, the first `carry` of them the best k of
// the earlier stages) -> sorted, best k kept in place, k-th key published as the next entry threshold; last stage: the
// list goes to out[q][0..k).  A counter beyond the capacity marks the query for the rescue pass.
__global__ __launch_bounds__(256) void tiled_select_kernel(u64* __restrict__ buf, unsigned* __restrict__ cnt, int cap, int k,
                                                            u64* __restrict__ thr_key, float* __restrict__ thr_score,
                                                            u64* __restrict__ out, unsigned* __restrict__ ovf, int direct_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* sk = reinterpret_cast<u64*>(smem);
  const int q = blockIdx.x, tid = threadIdx.x;
  u64* mine = buf + (size_t)q * cap;
  const unsigned raw = direct_n > 0 ? (unsigned)direct_n : cnt[q];
  if (raw > (unsigned)cap && tid == 0) ovf[q] = 1u;
  const int n = (int)min(raw, (unsigned)cap);
  int P = 2;
  while (P < n) P <<= 1;
  for (int i = tid; i < P; i += 256) sk[i] = i < n ? mine[i] : 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (P >> 1); i += 256) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const u64 a = sk[lo], b = sk[hi];
        if ((a < b) == desc) {
          sk[lo] = b;
          sk[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < k; i += 256) {
    const u64 v = i < P ? sk[i] : 0ull;
    mine[i] = v;
    if (out) out[(size_t)q * k + i] = v;
  }
  if (tid == 0) {
    const bool full = n >= k;
    cnt[q] = (unsigned)min(n, k);
    thr_key[q] = full ? sk[k - 1] : 0ull;
    thr_score[q] = full ? unorderable((unsigned)(sk[k - 1] >> 32)) : -INFINITY;
  }
}

// The first stage's selection (round 6): the stage wrote one key per row (0 = no key) into a buffer of its own, `n` keys per query
// at `src_stride`; the best k go to the front of the query's candidate buffer, the k-th becomes the next entry threshold.  Sorting
// thousands of keys is slow (a 4 096-key bitonic sort: ~80 us); instead the selection runs the staged search's own idea inside the
// LDS: the first `s0` keys are sorted, their k-th is a cut, the next window (16x / 4x the keys seen so far) is filtered against the
// cut -- ~k (growth - 1) survivors with rows in any but an adversarial order -- the survivors are sorted together with the best k,
// and so on.  The survivors' order of arrival does not matter (the sort orders them); more survivors than the LDS holds flag the
// query for the rescue pass / the gated full scan, like an overflowing candidate buffer.
constexpr int TSEL_NT = 1024;   // threads of the first stage's selection: its windows are scanned with sixteen keys per thread in flight
__global__ __launch_bounds__(TSEL_NT) void tiled_select_direct_kernel(const u64* __restrict__ src, int src_stride, int n, u64* __restrict__ buf,
                                                                   unsigned* __restrict__ cnt, int cap, int k, u64* __restrict__ thr_key,
                                                                   float* __restrict__ thr_score, u64* __restrict__ out,
                                                                   unsigned* __restrict__ ovf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ unsigned n_surv;
  u64* sk = reinterpret_cast<u64*>(smem);   // cap keys
  const int q = blockIdx.x, tid = threadIdx.x;
  const u64* in = src + (size_t)q * src_stride;
  u64* mine = buf + (size_t)q * cap;
  auto sort_desc = [&](int P) {   // bitonic, P a power of two <= cap
    for (int size = 2; size <= P; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < (P >> 1); i += TSEL_NT) {
          const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
          const bool desc = (lo & size) == 0;
          const u64 a = sk[lo], b = sk[hi];
          if ((a < b) == desc) {
            sk[lo] = b;
            sk[hi] = a;
          }
        }
        __syncthreads();
      }
  };
  const int growth = k > 16 ? 4 : 16;
  int seen = min(n, k > 16 ? 1024 : 256), P = 2;
  while (P < seen) P <<= 1;
  for (int i = tid; i < P; i += TSEL_NT) sk[i] = i < seen ? in[i] : 0ull;
  __syncthreads();
  sort_desc(P);
  while (seen < n) {
    const int chunk = min(n - seen, seen * (growth - 1));
    const u64 cut = P >= k ? sk[k - 1] : 0ull;   // 0: fewer than k keys so far, every key survives
    if (tid == 0) n_surv = 0u;
    __syncthreads();
    // sixteen keys per thread in flight: one dependent load per key made a 65 536-key window a 100 us chain of round trips
    for (int i0 = seen + tid; i0 < seen + chunk; i0 += TSEL_NT * 16) {
      u64 kk[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) kk[u] = i0 + u * TSEL_NT < seen + chunk ? __builtin_nontemporal_load(in + i0 + u * TSEL_NT) : 0ull;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (kk[u] > cut) {
          const unsigned pos = atomicAdd(&n_surv, 1u);
          if (k + (int)pos < cap) sk[k + pos] = kk[u];
        }
    }
    __syncthreads();
    const unsigned ns = n_surv;
    if (ns > (unsigned)(cap - k) && tid == 0) ovf[q] = 1u;
    const int m = k + (int)min(ns, (unsigned)(cap - k));
    // slots [min(P, k), k) (a first window shorter than k) hold nothing yet
    for (int i = P + tid; i < k; i += TSEL_NT) sk[i] = 0ull;
    P = 2;
    while (P < m) P <<= 1;
    for (int i = m + tid; i < P; i += TSEL_NT) sk[i] = 0ull;
    __syncthreads();
    sort_desc(P);
    seen += chunk;
  }
  for (int i = tid; i < k; i += TSEL_NT) {
    const u64 v = i < P ? sk[i] : 0ull;
    mine[i] = v;
    if (out) out[(size_t)q * k + i] = v;
  }
  if (tid == 0) {
    const bool full = n >= k && sk[k - 1] != 0ull;   // fewer than k keys (rows with NaN scores): everything passes the next stage
    cnt[q] = (unsigned)min(n, k);
    thr_key[q] = full ? sk[k - 1] : 0ull;
    thr_score[q] = full ? unorderable((unsigned)(sk[k - 1] >> 32)) : -INFINITY;
  }
}

// Queries whose candidate buffer overflowed: one workgroup per query walks the whole shard (16 rows in flight, one per
// 16-lane group, fp32 query in LDS) and rewrites out[q].  Launched with one workgroup per query after every tiled search;
// all of them leave at once unless a flag is set.
constexpr int TRESCUE_SLICES = 64;   // workgroups per flagged query (at most: gridDim.y = max(8, min(64, 8192 / queries)) -- thousands of queries launch fewer empty workgroups)
__global__ __launch_bounds__(256) void dense_tiled_rescue_kernel(const bf16_t* __restrict__ rows, long long n_rows, int dim,
                                                                  const float* __restrict__ queries, int nq, int k,
                                                                  const unsigned* __restrict__ ovf, u64* __restrict__ part,
                                                                  unsigned* __restrict__ done, u64* __restrict__ out) {
  // grid (queries, TRESCUE_SLICES).  Late round 6: the flagged query's shard walk is cut into slices -- one workgroup per slice,
  // four rows per 16-lane group in flight -- and the slice that finishes last (a counter per query) merges the slices' lists.  One
  // workgroup per query with one dependent row load per group and step took ~150 ms over 1.25 M x 768 rows: since every batch
  // of two or more queries over bf16 rows takes the tiled search, a topic-ordered corpus (a query's few thousand relevant rows
  // in one run, beyond the first stage) must not fall off that cliff.  The per-row arithmetic is unchanged.
  const int q = blockIdx.x, sl = blockIdx.y;
  if (!ovf[q]) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ bool last;
  float* sq = reinterpret_cast<float*>(smem);
  u64* lists = reinterpret_cast<u64*>(smem + (size_t)dim * sizeof(float));   // [16 groups][k]
  const int tid = threadIdx.x, grp = tid >> 4, gl = tid & 15;
  for (int i = tid; i < dim; i += 256) sq[i] = queries[(size_t)q * dim + i];
  for (int i = tid; i < 16 * k; i += 256) lists[i] = 0ull;
  __syncthreads();
  u64* mylist = lists + (size_t)grp * k;
  const int n_sl = (int)gridDim.y;
  const long long per = ((n_rows + n_sl - 1) / n_sl + 63) / 64 * 64;
  const long long lo = (long long)sl * per, hi = lo + per < n_rows ? lo + per : n_rows;
  for (long long r0 = lo + grp; r0 < hi; r0 += 64) {   // rows r0, r0 + 16, r0 + 32, r0 + 48 of this group in flight together
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = gl * 8; c < dim; c += 128) {
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long r = r0 + 16 * u < hi ? r0 + 16 * u : r0;
        v[u] = *reinterpret_cast<const bf16x8*>(rows + (size_t)r * dim + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) acc[u] = fmaf((float)v[u][j2], sq[c + j2], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float a = acc[u];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (gl == 0 && r0 + 16 * u < hi) insert_key(mylist, k, make_key(a, (unsigned)(r0 + 16 * u)));
    }
  }
  __syncthreads();
  auto merge_lists = [&](const u64* src, int n_lists, size_t stride, u64* dst) {   // n_lists sorted lists of k -> the best k (thread 0)
    int head[TRESCUE_SLICES];
    for (int g = 0; g < n_lists; ++g) head[g] = 0;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int g = 0; g < n_lists; ++g)
        if (head[g] < k) {
          const u64 v = src[(size_t)g * stride + head[g]];
          if (v > best) {
            best = v;
            bg = g;
          }
        }
      dst[i] = best;
      if (bg >= 0) ++head[bg];
    }
  };
  u64* mine = part + ((size_t)sl * nq + q) * k;
  if (tid == 0) {
    merge_lists(lists, 16, (size_t)k, mine);
    __threadfence();
    last = atomicAdd(done + q, 1u) == (unsigned)(n_sl - 1);
  }
  __syncthreads();
  if (last && tid == 0) {
    __threadfence();
    merge_lists(part + (size_t)q * k, n_sl, (size_t)nq * k, out + (size_t)q * k);
    done[q] = 0u;   // ready for the next search
  }
}

// ------------------------------------------------------------------------------------ fp32 rows, bf16 prefilter
// The store's default rows are fp32 and its contract bit-exact (scores = the oracle's sequential fmaf chain), which costs a
// full 4-byte-per-element scan per query batch (1.03 ms for one query over 1.25 M x 768 rows).  With a bf16 image of the rows
// beside them (dtype 2) a search first ranks the IMAGE -- half the bytes, and the batched routes above -- for the 64 best
// approximate scores a_r per query, then re-scores those 64 rows exactly.  Why that is still exact: the image row x~_r differs
// from x_r by a vector whose norm is MEASURED when the image is built (E = max_r ||x~_r - x_r||, prefilter_image_kernel; bf16
// keeps 8 significant bits, so E <= 2^-8 max||x|| in the worst case and about 0.45 of that on ordinary data), hence
//     |a_r - e_r| <= eps = E ||q|| + Dq max||x~|| + 4 dim 2^-24 max||x|| ||q||
// (Cauchy-Schwarz; Dq = the norm of what the query loses on its way into the pass: 0 for the fp32 query of the one-pass route,
// ||q~ - q|| measured on the host for the bf16-rounded queries of the batch route; the last term covers the fp32 accumulation
// of a_r and of the oracle's chain e_r with a factor 2 to spare).
// Every row outside the 64 has a_r <= a_64, hence e_r <= a_64 + eps; the k best approximate rows have e_r >= a_k - eps.  If
//     a_64 + eps < a_k - eps
// every outside row is strictly below k candidates in exact score, so the exact top-k lies inside the 64 -- found by the exact
// chain on 64 rows.  If the inequality fails for a query (scores bunched within 2 eps: near-duplicate rows) its flag is set and
// the full fp32 scan re-answers it: correctness never rests on the data.
constexpr int PFK = 64;   // candidates per query of a batch (the tiled search delivers 64 at no extra cost)

// fp32 rows -> bf16 image + the maximum squared row norm (one wave per row)
__global__ __launch_bounds__(256) void prefilter_image_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n_rows,
                                                               int dim, float* __restrict__ stats /*[0] max ||x||^2, [1] max ||x~ - x||^2*/) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n_rows) return;
  const float* x = src + (size_t)r * dim;
  bf16_t* y = dst + (size_t)r * dim;
  float s2 = 0.f, d2 = 0.f;
  for (int c = lane * 4; c < dim; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = (bf16_t)v[j];
      s2 = fmaf(v[j], v[j], s2);
      const float d = v[j] - (float)o[j];   // exact in fp32: the two are within a factor two of each other
      d2 = fmaf(d, d, d2);
    }
    *reinterpret_cast<bf16x4*>(y + c) = o;
  }
  s2 = wave_sum(s2) * 1.0001f;   // the sums' own rounding
  d2 = wave_sum(d2) * 1.0001f;
  if (lane == 0 && s2 == s2) atomicMax(reinterpret_cast<unsigned*>(stats), __builtin_bit_cast(unsigned, s2));
  if (lane == 0 && d2 == d2) atomicMax(reinterpret_cast<unsigned*>(stats) + 1, __builtin_bit_cast(unsigned, d2));
}

// The oracle's chain  acc = fmaf(x[c], q[c], acc), c ascending from acc = 0,  for one row per lane.  The chain itself is serial;
// the row is fetched eight 16-byte pieces ahead of it (one lane walks 3 KB alone: without the batch the loop ran at one memory
// round trip per piece -- 90 us for a few hundred candidates).
__device__ __forceinline__ float exact_chain(const float* __restrict__ x, const float* __restrict__ q, int dim) {
  float acc = 0.f;
  int c = 0;
  for (; c + 32 <= dim; c += 32) {
    f32x4 xv[8], qq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      xv[u] = *reinterpret_cast<const f32x4*>(x + c + 4 * u);
      qq[u] = *reinterpret_cast<const f32x4*>(q + c + 4 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __fmaf_rn(xv[u][j], qq[u][j], acc);
  }
  for (; c < dim; c += 4) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + c);
    const f32x4 qq = *reinterpret_cast<const f32x4*>(q + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __fmaf_rn(xv[j], qq[j], acc);
  }
  return acc;
}

// One wave per query: the 64 approximate candidates (keys sorted descending, 0 = none) -> sufficiency test, exact chain per
// candidate (lane = candidate; c ascending from acc = 0: the oracle's arithmetic), exact keys ranked by counting.
__global__ __launch_bounds__(64) void prefilter_rescore_kernel(const u64* __restrict__ approx, const float* __restrict__ rows, int dim,
                                                                const float* __restrict__ queries, const float* __restrict__ eps, int k,
                                                                int pfk, u64* __restrict__ out, unsigned* __restrict__ flag) {
  __shared__ u64 keys[PFK];
  const int q = blockIdx.x, lane = threadIdx.x;
  const u64 ak = lane < pfk ? approx[(size_t)q * pfk + lane] : 0ull;
  const float a_k = unorderable((unsigned)(approx[(size_t)q * pfk + (k - 1)] >> 32));
  const u64 last = approx[(size_t)q * pfk + (pfk - 1)];
  const bool k_full = approx[(size_t)q * pfk + (k - 1)] != 0ull;
  // a short list holds every row of the shard: nothing is outside it
  const bool ok = last == 0ull || (k_full && unorderable((unsigned)(last >> 32)) + eps[q] < a_k - eps[q]);
  if (lane == 0) flag[q] = ok ? 0u : 1u;
  u64 key = 0ull;
  if (ak != 0ull) {
    const unsigned row = 0xFFFFFFFFu - (unsigned)(ak & 0xFFFFFFFFu);
    key = make_key(exact_chain(rows + (size_t)row * dim, queries + (size_t)q * dim, dim), row);
  }
  keys[lane] = key;
  __syncthreads();
  int rank = 0;
  for (int i = 0; i < PFK; ++i) rank += keys[i] > key ? 1 : 0;
  if (key != 0ull && rank < k) out[(size_t)q * k + rank] = key;
  int n_live = 0;
  for (int i = 0; i < PFK; ++i) n_live += keys[i] != 0ull ? 1 : 0;
  if (lane >= n_live && lane < k) out[(size_t)q * k + lane] = 0ull;   // fewer rows than k
}

// One query (two to four: prefilter_collect_multi_kernel below, one pass for all of them): ONE streaming pass over the image collects
// every row that can be in the exact top-k.  A prefix of the
// shard is ranked first (exact top-k of the IMAGE scores over the first rows: t0 = its k-th score).  The exact k-th score over
// the whole shard is at least the exact k-th over the prefix, which is at least t0 - eps; so a row of the exact top-k has
// e_r >= t0 - eps, hence a_r >= t0 - 2 eps: the pass appends exactly the rows with a_r >= t0 - 2 eps to a candidate list (no
// lists to maintain, no merge), the candidates are re-scored with the exact chain and sorted.  More candidates than the
// list holds (flat score distributions) = fall back to the full fp32 scan.
constexpr int PFCAP = 4096;           // candidate slots per query
constexpr long long PFPREFIX = 32768; // rows ranked first for the entry threshold
template <int DIMC>
__global__ __launch_bounds__(256) void prefilter_collect_kernel(const bf16_t* __restrict__ rows, long long n_rows, int dim,
                                                                 const float* __restrict__ query, const u64* __restrict__ kth_key,
                                                                 const float* __restrict__ eps, unsigned* __restrict__ cnt,
                                                                 unsigned* __restrict__ cand_rows, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, grp = tid >> 4, gl = tid & 15;
  for (int i = tid; i < dim; i += 256) sq[i] = query[i];
  __syncthreads();
  float qreg[DIMC > 0 ? DIMC * 8 : 1];
  if constexpr (DIMC > 0) {
#pragma unroll
    for (int i = 0; i < DIMC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) qreg[i * 8 + j] = sq[gl * 8 + i * 128 + j];
  }
  const u64 kk = *kth_key;
  const float tau = kk ? unorderable((unsigned)(kk >> 32)) - 2.f * eps[0] : -INFINITY;
  const long long r_begin = (long long)blockIdx.x * rows_per_wg, r_end = min(n_rows, r_begin + rows_per_wg);
  for (long long r = r_begin + grp; r < r_end; r += 32) {
    const bool has2 = r + 16 < r_end;
    const char* row0 = reinterpret_cast<const char*>(rows + (size_t)r * dim) + (size_t)gl * 16;
    const char* row1 = has2 ? row0 + (size_t)16 * dim * 2 : row0;
    float acc0 = 0.f, acc1 = 0.f;
    if constexpr (DIMC > 0) {
      f32x4 raw0[DIMC], raw1[DIMC];
#pragma unroll
      for (int i = 0; i < DIMC; ++i) {
        raw0[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row0 + (size_t)i * 256));
        raw1[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row1 + (size_t)i * 256));
      }
#pragma unroll
      for (int i = 0; i < DIMC; ++i) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, raw0[i]), b = __builtin_bit_cast(bf16x8, raw1[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc0 = fmaf((float)a[j], qreg[i * 8 + j], acc0);
          acc1 = fmaf((float)b[j], qreg[i * 8 + j], acc1);
        }
      }
    } else {
      for (int c = gl * 8; c < dim; c += 128) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(rows + (size_t)r * dim + c);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(rows + (size_t)(has2 ? r + 16 : r) * dim + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc0 = fmaf((float)a[j], sq[c + j], acc0);
          acc1 = fmaf((float)b[j], sq[c + j], acc1);
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      acc0 += __shfl_xor(acc0, o, 64);
      acc1 += __shfl_xor(acc1, o, 64);
    }
    if (gl == 0) {
      if (acc0 >= tau) {
        const unsigned slot = atomicAdd(cnt, 1u);
        if (slot < (unsigned)PFCAP) cand_rows[slot] = (unsigned)r;
      }
      if (has2 && acc1 >= tau) {
        const unsigned slot = atomicAdd(cnt, 1u);
        if (slot < (unsigned)PFCAP) cand_rows[slot] = (unsigned)(r + 16);
      }
    }
  }
}

// Two to PFQ queries in ONE streaming pass (round 6; the route above ran once per query: two queries cost two passes, 0.70 ms, and
// three or four went to the tiled search at 0.61-0.65 ms -- the size a handful of coalesced VerbatimRAG.query calls produces).
// 32 lanes per row: a lane holds dim / 32 elements of EVERY query in registers (24 per query at dim 768) and fetches its 16-byte
// pieces of a row at a 512-byte stride; a half-wave takes four rows per step (12 loads in flight per lane), 32 rows per workgroup
// and step.  The image scores' summation order is free (the bound's accumulation term covers any order).  Same append as the
// single-query kernel, one counter and one list per query.
constexpr int PFQ = 4;   // queries the one-pass route takes together (eight in one pass -- 192 query registers, two rows per step -- measured 0.60-0.62 ms for 5-8 queries: what the tiled search takes; profiles/r06_onepass_multi_probe.txt)
template <int DIMC32, int QN>   // dim / 256; queries compiled in (2 or 4; slots beyond nq never append)
__global__ __launch_bounds__(256) void prefilter_collect_multi_kernel(const bf16_t* __restrict__ rows, long long n_rows, int dim,
                                                                       const float* __restrict__ queries, int nq,
                                                                       const u64* __restrict__ kth_key, const float* __restrict__ eps,
                                                                       unsigned* __restrict__ cnt, unsigned* __restrict__ cand_rows,
                                                                       int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = reinterpret_cast<float*>(smem);   // [QN][dim]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, hl = lane & 31;
  for (int i = tid; i < QN * dim; i += 256) sq[i] = i < nq * dim ? queries[i] : 0.f;
  __syncthreads();
  float qreg[QN][DIMC32 * 8];
#pragma unroll
  for (int q = 0; q < QN; ++q)
#pragma unroll
    for (int i = 0; i < DIMC32; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) qreg[q][i * 8 + j] = sq[q * dim + i * 256 + hl * 8 + j];
  float tau[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    tau[q] = INFINITY;   // an unused slot appends nothing
    if (q < nq) {
      const u64 kk = kth_key[q];
      tau[q] = kk ? unorderable((unsigned)(kk >> 32)) - 2.f * eps[q] : -INFINITY;
    }
  }
  constexpr int RPH = 4;   // rows per half-wave and step
  const long long r_begin = (long long)blockIdx.x * rows_per_wg, r_end = min(n_rows, r_begin + rows_per_wg);
  for (long long r0 = r_begin + wave * 2 + half; r0 < r_end; r0 += 8 * RPH) {
    f32x4 raw[RPH][DIMC32];
#pragma unroll
    for (int rr = 0; rr < RPH; ++rr) {
      const long long r = min(r0 + 8 * rr, r_end - 1);   // past the end: a row read again, never appended
      const char* p = reinterpret_cast<const char*>(rows + (size_t)r * dim) + (size_t)hl * 16;
#pragma unroll
      for (int i = 0; i < DIMC32; ++i) raw[rr][i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (size_t)i * 512));
    }
    float acc[QN][RPH];
#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int rr = 0; rr < RPH; ++rr) acc[q][rr] = 0.f;
#pragma unroll
    for (int rr = 0; rr < RPH; ++rr)
#pragma unroll
      for (int i = 0; i < DIMC32; ++i) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, raw[rr][i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)a[j];
#pragma unroll
          for (int q = 0; q < QN; ++q) acc[q][rr] = fmaf(f, qreg[q][i * 8 + j], acc[q][rr]);
        }
      }
#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int rr = 0; rr < RPH; ++rr) {
        float v = row16_sum(acc[q][rr]);
        acc[q][rr] = v + __shfl_xor(v, 16, 64);   // the other 16 lanes of this half-wave
      }
    if (hl == 0) {
#pragma unroll
      for (int rr = 0; rr < RPH; ++rr) {
        const long long r = r0 + 8 * rr;
        if (r >= r_end) continue;
#pragma unroll
        for (int q = 0; q < QN; ++q)
          if (acc[q][rr] >= tau[q]) {
            const unsigned slot = atomicAdd(cnt + q, 1u);
            if (slot < (unsigned)PFCAP) cand_rows[(size_t)q * PFCAP + slot] = (unsigned)r;
          }
      }
    }
  }
}

// Entry threshold of the one-pass route in one launch: 128 prefix rows per workgroup, image scores as in the collect kernel, the
// two best keys of every workgroup written out.  ANY t0 with at least k rows at a_r >= t0 is a valid threshold (the exact k-th
// score of the shard is then >= t0 - eps), so the k-th largest of these keys -- the k-th over a SUBSET of the prefix rows -- serves
// as well as the prefix's own k-th and needs no per-workgroup top-k lists and no merge; tiled_select_kernel picks it.
// Workgroup 0 also clears the candidate counter and the overflow flag of this query.
constexpr int PFROWS = 128;   // prefix rows per workgroup
constexpr int PFBEST = 2;     // keys kept per workgroup
template <int DIMC>
__global__ __launch_bounds__(256) void prefilter_prefix_kernel(const bf16_t* __restrict__ rows, long long n_rows, int dim,
                                                                const float* __restrict__ query, u64* __restrict__ best,
                                                                unsigned* __restrict__ cnt, unsigned* __restrict__ flag, int best_stride,
                                                                long long block_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = reinterpret_cast<float*>(smem);
  u64* keys = reinterpret_cast<u64*>(smem + (size_t)dim * sizeof(float));   // [PFROWS]
  const int tid = threadIdx.x, grp = tid >> 4, gl = tid & 15;
  query += (size_t)blockIdx.y * dim;   // blockIdx.y = query of a multi-query launch (round 6)
  best += (size_t)blockIdx.y * best_stride;
  cnt += blockIdx.y;
  flag += blockIdx.y;
  if (blockIdx.x == 0 && tid == 0) {
    *cnt = 0u;
    *flag = 0u;
  }
  for (int i = tid; i < dim; i += 256) sq[i] = query[i];
  __syncthreads();
  float qreg[DIMC > 0 ? DIMC * 8 : 1];
  if constexpr (DIMC > 0) {
#pragma unroll
    for (int i = 0; i < DIMC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) qreg[i * 8 + j] = sq[gl * 8 + i * 128 + j];
  }
  // block_stride = PFROWS: a prefix; larger: the workgroups' 128-row blocks are a SAMPLE spread over the shard (any subset of rows gives a
  // valid threshold; a sample gives one that does not depend on the order of the rows -- see dense_tiled_search)
  const long long r_begin = (long long)blockIdx.x * block_stride;
  for (int it = 0; it < PFROWS / 32; ++it) {
    const long long r = r_begin + grp + 32 * it;
    const bool has0 = r < n_rows, has1 = r + 16 < n_rows;
    const bf16_t* row0 = rows + (size_t)(has0 ? r : r_begin) * dim;
    const bf16_t* row1 = rows + (size_t)(has1 ? r + 16 : r_begin) * dim;
    float acc0 = 0.f, acc1 = 0.f;
    if constexpr (DIMC > 0) {
      bf16x8 a[DIMC], b[DIMC];
#pragma unroll
      for (int i = 0; i < DIMC; ++i) {
        a[i] = *reinterpret_cast<const bf16x8*>(row0 + gl * 8 + i * 128);
        b[i] = *reinterpret_cast<const bf16x8*>(row1 + gl * 8 + i * 128);
      }
#pragma unroll
      for (int i = 0; i < DIMC; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc0 = fmaf((float)a[i][j], qreg[i * 8 + j], acc0);
          acc1 = fmaf((float)b[i][j], qreg[i * 8 + j], acc1);
        }
    } else {
      for (int c = gl * 8; c < dim; c += 128) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(row0 + c);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(row1 + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc0 = fmaf((float)a[j], sq[c + j], acc0);
          acc1 = fmaf((float)b[j], sq[c + j], acc1);
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      acc0 += __shfl_xor(acc0, o, 64);
      acc1 += __shfl_xor(acc1, o, 64);
    }
    if (gl == 0) {
      keys[grp + 32 * it] = has0 ? make_key(acc0, (unsigned)r) : 0ull;
      keys[grp + 32 * it + 16] = has1 ? make_key(acc1, (unsigned)(r + 16)) : 0ull;
    }
  }
  __syncthreads();
  if (tid < PFROWS) {
    const u64 key = keys[tid];
    int rank = 0, n_live = 0;
    for (int i = 0; i < PFROWS; ++i) {
      rank += keys[i] > key ? 1 : 0;
      n_live += keys[i] != 0ull ? 1 : 0;
    }
    if (key != 0ull && rank < PFBEST) best[(size_t)blockIdx.x * PFBEST + rank] = key;
    if (tid >= n_live && tid < PFBEST) best[(size_t)blockIdx.x * PFBEST + tid] = 0ull;
  }
}

// Exact keys of the collected rows, zero keys behind them: 16 candidates per workgroup.  The oracle's chain is serial per row, so
// what costs is fetching the row: all 256 threads stage the 16 rows (coalesced 16-byte pieces, 512 columns per round) and the
// query through LDS, then 16 lanes run one chain each out of LDS (row stride 516 words: the 16 b128 reads of a step fall in
// 16 different bank quartets).  One lane walking its 3 KB row alone took 24 us for a few hundred candidates; this form 8.
constexpr int RCH = 512;   // columns staged per round
// FROM_KEYS (round 6, the batch collect route): the candidates are the approximate KEYS the score GEMM's epilogue appended to
// keys[] itself (row = the key's low word, complemented); the exact keys replace them in place.
template <bool FROM_KEYS>
__global__ __launch_bounds__(256) void prefilter_rescore_list_kernel(const unsigned* __restrict__ cand_rows, const unsigned* __restrict__ cnt,
                                                                       const float* __restrict__ rows, int dim,
                                                                       const float* __restrict__ query, u64* __restrict__ keys) {
  __shared__ __attribute__((aligned(16))) float srow[16][RCH + 4];
  __shared__ __attribute__((aligned(16))) float sq[RCH];
  __shared__ unsigned srid[16];
  const int tid = threadIdx.x, base = blockIdx.x * 16;
  if constexpr (!FROM_KEYS) cand_rows += (size_t)blockIdx.y * PFCAP;   // blockIdx.y = query of a multi-query launch (round 6)
  cnt += blockIdx.y;
  query += (size_t)blockIdx.y * dim;
  keys += (size_t)blockIdx.y * PFCAP;
  const unsigned n = min(*cnt, (unsigned)PFCAP);
  if ((unsigned)base >= n) {
    if (tid < 16) keys[base + tid] = 0ull;
    return;
  }
  if (tid < 16) {
    const unsigned src = (unsigned)(base + tid) < n ? base + tid : base;
    if constexpr (FROM_KEYS) {
      const u64 ak = keys[src];   // 0 = a reserved slot whose key failed the epilogue's key test: row 0 is re-scored, the key dropped below
      srid[tid] = ak ? 0xFFFFFFFFu - (unsigned)(ak & 0xFFFFFFFFu) : 0xFFFFFFFFu;
    } else {
      srid[tid] = cand_rows[src];
    }
  }
  __syncthreads();
  __shared__ unsigned dead[16];
  if (tid < 16) {
    dead[tid] = srid[tid] == 0xFFFFFFFFu ? 1u : 0u;
    if (dead[tid]) srid[tid] = 0u;
  }
  __syncthreads();
  float acc = 0.f;
  for (int c0 = 0; c0 < dim; c0 += RCH) {
    const int w = min(RCH, dim - c0), ppr = w >> 2, total = 16 * ppr;   // dim % 4 == 0
    for (int p0 = 0; p0 < total; p0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * 256 + tid;
        if (p < total) v[u] = *reinterpret_cast<const f32x4*>(rows + (size_t)srid[p / ppr] * dim + c0 + 4 * (p % ppr));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + u * 256 + tid;
        if (p < total) *reinterpret_cast<f32x4*>(&srow[p / ppr][4 * (p % ppr)]) = v[u];
      }
    }
    for (int i = tid; i < ppr; i += 256) *reinterpret_cast<f32x4*>(&sq[4 * i]) = *reinterpret_cast<const f32x4*>(query + c0 + 4 * i);
    __syncthreads();
    if (tid < 16) {
#pragma unroll 8
      for (int c = 0; c < w; c += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&srow[tid][c]);
        const f32x4 qq = *reinterpret_cast<const f32x4*>(&sq[c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __fmaf_rn(xv[j], qq[j], acc);
      }
    }
    __syncthreads();
  }
  if (tid < 16) keys[base + tid] = ((unsigned)(base + tid) < n && !dead[tid]) ? make_key(acc, srid[tid]) : 0ull;
}

// Device-resident searches cannot fall back through the host: the gated full scan re-answers flagged queries into `exact`, and this
// picks, per query, the scan's list where the flag is set and the re-scored candidates' list elsewhere (in place in `pf`).
__global__ void prefilter_combine_kernel(u64* __restrict__ pf, const u64* __restrict__ exact, const unsigned* __restrict__ flag, int nq, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * k) return;
  if (flag[i / k]) pf[i] = exact[i];
}

// Smallest batch that takes the tiled search (round 6; profiles/r06_dense_midbatch_probe.txt, 1.25 M x 768 rows):
//  * the bf16 image of an fp32 index: 3 -- every batch the one-pass route (1-2 queries) does not take.  The 32-queries-per-pass
//    exact scan costs 0.93-1.02 ms per pass (3-32 queries; 1.96-2.1 ms for 33-63), the image route 0.65-0.81 (0.88-0.95);
//  * bf16 rows: 2 (late round 6).  One pass of the 32-query kernel is 0.43-0.48 ms (two queries ran as two single-query passes:
//    0.95 ms); with its first stage a whole round of one-key-per-row writes the tiled search takes 0.37-0.41 ms for anything
//    up to 64 query columns -- 2-32 exact-bf16 queries 0.38-0.40, 2-32 generic fp32 queries (column pairs) 0.37-0.40
//    (tools/probes/bf16_small_batch_route.py, profiles/r06_search_timeline.txt).  The pass kernels keep the shards the tiled
//    search does not take (fewer than 4 096 rows, dim not a multiple of 64) and VRAG_TOPK_NO_TILED.
constexpr int kTiledMinImage = 3, kTiledMinBf16 = 2, kTiledMinBf16Pairs = 2;
static bool dense_use_tiled(int dtype, int dim, int nq, int k, long long size, int min_nq = kTiledMinBf16) {
  static const bool off = getenv("VRAG_TOPK_NO_TILED") != nullptr;   // A/B against the 32-query passes
  return !off && dtype == 0 && dim % 64 == 0 && nq >= min_nq && k <= KMAX && size >= 4096;
}

// ------------------------------------------------------------------------------------ sparse
template <bool LDSQ>
__global__ __launch_bounds__(1024) void sparse_topk_kernel(const unsigned short* __restrict__ cols,
                                                            const float* __restrict__ vals,
                                                            const long long* __restrict__ slice_off,
                                                            const int* __restrict__ slice_len, int n_slices,
                                                            long long n_docs, const float* __restrict__ qdense, int vocab,
                                                            int nq, int q, int k, int slices_per_wg,
                                                            u64* __restrict__ cand, const unsigned* __restrict__ docid,
                                                            const u64* __restrict__ bound) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [16 waves][k] lists, then (LDSQ) the dense query vector
  u64* lists = reinterpret_cast<u64*>(smem);
  float* sq = reinterpret_cast<float*>(smem + (size_t)16 * k * sizeof(u64));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* qv = qdense + (size_t)q * vocab;
  if constexpr (LDSQ) {
    for (int i = tid; i < vocab; i += 1024) sq[i] = qv[i];
  }
  for (int i = tid; i < 16 * k; i += 1024) lists[i] = 0ull;
  __syncthreads();
  u64* mylist = lists + (size_t)wave * k;
  // Slices are dealt round-robin over the waves of the whole grid, not as one contiguous range per workgroup: the documents are
  // stored by length, so contiguous ranges hand the last workgroups the longest documents (+29 % terms over the mean for
  // Poisson(128) lengths) and the pass waits for them.
  (void)slices_per_wg;
  for (int s = blockIdx.x + gridDim.x * wave; s < n_slices; s += gridDim.x * 16) {
    const long long off = slice_off[s];
    const int ng = slice_len[s];   // groups of 4 terms
    const u32x2* c = reinterpret_cast<const u32x2*>(cols + off) + lane;     // group g of this lane's document: c[g * 64]
    const f32x4* v = reinterpret_cast<const f32x4*>(vals + off) + lane;
    float acc = 0.f;
    auto qw = [&](unsigned t) { return LDSQ ? sq[t] : qv[t]; };
    auto consume = [&](const u32x2& cg, const f32x4& vg) {   // strictly sequential, term-order fmaf chain
      acc = __fmaf_rn(vg[0], qw(cg[0] & 0xFFFFu), acc);
      acc = __fmaf_rn(vg[1], qw(cg[0] >> 16), acc);
      acc = __fmaf_rn(vg[2], qw(cg[1] & 0xFFFFu), acc);
      acc = __fmaf_rn(vg[3], qw(cg[1] >> 16), acc);
    };
    // 8 groups (32 terms) per step: all 16 loads of a step (8 + 16 bytes per lane each) are issued before the chain consumes them
    int g = 0;
    for (; g + 8 <= ng; g += 8) {
      u32x2 cg[8];
      f32x4 vg[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cg[u] = __builtin_nontemporal_load(c + (size_t)(g + u) * 64);
        vg[u] = __builtin_nontemporal_load(v + (size_t)(g + u) * 64);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) consume(cg[u], vg[u]);
    }
    for (; g < ng; ++g) consume(c[(size_t)g * 64], v[(size_t)g * 64]);
    const long long doc = (long long)s * 64 + lane;  // position in nnz-sorted order
    const bool hit = doc < n_docs && acc > 0.f;      // inverted-index semantics: no shared term => not a hit
    // the key carries the caller's document index (not the sorted position): ties order by id ascending
    const u64 key = hit ? make_key_below(acc, docid[doc], bound ? bound[q] : ~0ull) : 0ull;
    wave_insert_topk(mylist, k, key, lane);
  }
  __syncthreads();
  if (tid == 0) {
    int head[16];
    for (int g = 0; g < 16; ++g) head[g] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + q) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int g = 0; g < 16; ++g) {
        if (head[g] < k) {
          const u64 v = lists[(size_t)g * k + head[g]];
          if (v > best) {
            best = v;
            bg = g;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// (term, weight) pairs -> dense query vectors [nq][vocab] (cleared by the caller): one thread per query, the query's own term order
__global__ void sparse_scatter_queries_kernel(const long long* __restrict__ indptr, const int* __restrict__ terms, const float* __restrict__ weights,
                                              int nq, int vocab, float* __restrict__ dense) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  for (long long j = indptr[q]; j < indptr[q + 1]; ++j) dense[(size_t)q * vocab + terms[j]] = weights[j];
}

// ------------------------------------------------------------------------------------ sparse, batched queries
// QB queries per pass over the shard.  The QB queries of a pass touch at most a few hundred distinct terms
// ("union").  Per pass the LDS holds ONE u16 map term -> union id (0 = in none of the queries; 60 KiB for a
// 30 522-term vocabulary) and a small dense weight table W[q][union id] (W[q][0] = 0), so a document term
// costs one 2-byte LDS read shared by all queries plus the union id's QB weights (stored [union id][q]: QB/4 16-byte
// reads, and the ~97 % of document terms that are in no query all read entry 0 -- a broadcast), with no divergence and
// no probing; absent terms contribute fmaf(v, 0, acc) == acc and the per-document sum keeps the document's term
// order: bit-identical to the single-query kernel and the CPU restatement.  The document stream is read once
// for QB queries.
constexpr int SUW = 1024;                   // weight-table stride: union ids 0 .. SUW-1

// NW = waves per workgroup (16; an 8-wave form with 256 registers and two 40-term load sets measured 10 % slower for 16 queries:
// profiles/r05_sparse_probes.txt).
// PAD (round 6): weight rows at a stride of QB + PAD floats.  PAD = 4 spreads the row starts over all sixteen 4-bank groups instead of
// four (a row of 16 floats starts on bank 16 (uid % 4)): the ~8 active lanes of a 16-lane read group then collide 2 deep instead of
// 3-4.  Round 5 measured this as "no gain" when the pass was still bound by its list maintenance and its unbalanced slices; with
// those gone it is 8 % (1 000 queries 14.95 -> 13.70 ms, 16-query pass 0.264 -> 0.242 ms: profiles/r06_sparse_probes.txt).  Taken
// for 16-query passes whenever the padded table still fits the LDS beside the term map and the lists.
template <int QB, int NW, int PAD>
__global__ __launch_bounds__(NW * 64) void sparse_topk_multi_kernel(const unsigned short* __restrict__ cols,
                                                                 const float* __restrict__ vals,
                                                                 const long long* __restrict__ slice_off,
                                                                 const int* __restrict__ slice_len, int n_slices,
                                                                 long long n_docs, const unsigned short* __restrict__ qmap,
                                                                 const float* __restrict__ qw, int vocab, int n_union,
                                                                 int nq, int q0, int k, int slices_per_wg,
                                                                 u64* __restrict__ cand, const unsigned* __restrict__ docid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int vpad = (vocab + 7) & ~7;
  unsigned short* tmap = reinterpret_cast<unsigned short*>(smem);                 // [vpad]
  constexpr int WS = QB + PAD;   // weight-row stride in floats
  float* tw = reinterpret_cast<float*>(smem + (size_t)vpad * 2);                  // [SUW][WS] (first n_union+1 rows used)
  u64* lists = reinterpret_cast<u64*>(smem + (size_t)vpad * 2 + (size_t)WS * SUW * 4);   // [NW waves][QB][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < vpad / 8; i += NW * 64)   // 16-byte copies
    reinterpret_cast<f32x4*>(tmap)[i] = reinterpret_cast<const f32x4*>(qmap)[i];
  for (int i = tid; i < QB * (n_union + 1); i += NW * 64) {
    const int q = i / (n_union + 1), u = i - q * (n_union + 1);
    tw[u * WS + q] = qw[(size_t)q * SUW + u];
  }
  for (int i = tid; i < NW * QB * k; i += NW * 64) lists[i] = 0ull;
  __syncthreads();
  u64* mylists = lists + (size_t)wave * QB * k;
  (void)slices_per_wg;   // slices dealt round-robin over the grid's waves (see sparse_topk_kernel)
  for (int s = blockIdx.x + gridDim.x * wave; s < n_slices; s += gridDim.x * NW) {
    const long long off = slice_off[s];
    const int ng = slice_len[s];   // groups of 4 terms
    const u32x2* c = reinterpret_cast<const u32x2*>(cols + off) + lane;
    const f32x4* v = reinterpret_cast<const f32x4*>(vals + off) + lane;
    f32x2 acc2[QB / 2];
#pragma unroll
    for (int q = 0; q < QB / 2; ++q) acc2[q] = f32x2{0.f, 0.f};
    auto term_uid = [&](unsigned uid, float v1) {   // acc[q] = fma(v1, W[uid][q], acc[q]) for every query, two per v_pk_fma_f32
      // Only the lanes whose term is in the pass's union (a few per cent) read weight rows: the other lanes would add
      // v * 0 -- exactly nothing.  The branch is per lane (exec mask); a wave with no hit skips.
      if (uid != 0u) {
        const f32x4* wrow = reinterpret_cast<const f32x4*>(tw + uid * WS);
        const f32x2 vv = splat2(v1);
#pragma unroll
        for (int g4 = 0; g4 < QB / 4; ++g4) {
          const f32x4 w4 = wrow[g4];
          acc2[2 * g4] = pk_fma(vv, f32x2{w4[0], w4[1]}, acc2[2 * g4]);
          acc2[2 * g4 + 1] = pk_fma(vv, f32x2{w4[2], w4[3]}, acc2[2 * g4 + 1]);
        }
      }
    };
    auto consume = [&](const u32x2& cg, const f32x4& vg) {   // the document's term order; weight rows of two terms in flight
      // (the four map reads of a group issued together, ahead of the weight rows: measured equal -- the pass is bound by LDS
      // throughput, not by the dependent round trips: profiles/r06_sparse_probes.txt)
      term_uid(tmap[cg[0] & 0xFFFFu], vg[0]);
      term_uid(tmap[cg[0] >> 16], vg[1]);
      if constexpr (QB > 8 && NW == 16) __builtin_amdgcn_sched_barrier(0);   // 128 registers: weight rows of two terms in flight at most
      term_uid(tmap[cg[1] & 0xFFFFu], vg[2]);
      term_uid(tmap[cg[1] >> 16], vg[3]);
    };
    // Two register sets of G groups (4 G terms): the loads of step i + 1 are in flight while step i is consumed -- one 16-wave
    // workgroup per CU (the term map fills the LDS), so the bytes in flight per lane are what covers the memory latency.
    constexpr int G = NW == 8 ? 10 : (QB > 8 ? 3 : 5);   // groups per load set: what the register budget of the configuration holds twice
    u32x2 ca[G], cb[G];
    f32x4 va[G], vb[G];
    auto load = [&](u32x2 (&cd)[G], f32x4 (&vd)[G], int g0) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int gi = min(g0 + u, ng - 1);   // the tail re-reads the last group; its copies are not consumed
        cd[u] = __builtin_nontemporal_load(c + (size_t)gi * 64);
        vd[u] = __builtin_nontemporal_load(v + (size_t)gi * 64);
      }
    };
    auto eat = [&](const u32x2 (&cd)[G], const f32x4 (&vd)[G], int g0) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (g0 + u < ng) consume(cd[u], vd[u]);   // wave-uniform
        if constexpr (NW == 16 || (QB > 8)) { if ((u & (NW == 8 ? 1 : 0)) == (NW == 8 ? 1 : 0)) __builtin_amdgcn_sched_barrier(0); }   // bound the LDS reads in flight: every group (128 registers) / every other group
      }
    };
    // Every load is issued unconditionally (group indices clamp to the slice's last group; what a tail re-reads is not consumed):
    // the number of loads in flight is then a compile-time fact and the wait in front of a set leaves the OTHER set's loads
    // outstanding (s_waitcnt vmcnt(2 G)).  With the loads under `if (more)` the compiler had to drain everything at each join
    // -- no overlap at all, and a pass ran at the pace of one set per memory round trip (profiles/r05_sparse_lines.json).
    if (ng > 0) {
      load(ca, va, 0);
      load(cb, vb, G);
      for (int g0 = 0; g0 < ng; g0 += 2 * G) {
        eat(ca, va, g0);
        load(ca, va, g0 + 2 * G);
        eat(cb, vb, g0 + G);
        load(cb, vb, g0 + 3 * G);
      }
    }
    float acc[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) acc[q] = acc2[q >> 1][q & 1];
    const long long doc = (long long)s * 64 + lane;
    const unsigned did = doc < n_docs ? docid[doc] : 0u;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const bool hit = doc < n_docs && acc[q] > 0.f;
      const u64 key = hit ? make_key(acc[q], did) : 0ull;
      wave_insert_topk(mylists + q * k, k, key, lane);
    }
  }
  __syncthreads();
  if (tid < QB && q0 + tid < nq) {
    int head[NW];
    for (int g = 0; g < NW; ++g) head[g] = 0;
    u64* out = cand + ((size_t)blockIdx.x * nq + (q0 + tid)) * k;
    for (int i = 0; i < k; ++i) {
      u64 best = 0ull;
      int bg = -1;
      for (int g = 0; g < NW; ++g) {
        if (head[g] < k) {
          const u64 vv = lists[((size_t)g * QB + tid) * k + head[g]];
          if (vv > best) {
            best = vv;
            bg = g;
          }
        }
      }
      out[i] = best;
      if (bg >= 0) ++head[bg];
    }
  }
}

// ------------------------------------------------------------------------------------ merge
// One workgroup per query: k rounds of workgroup-wide arg-max over the candidate keys.
__global__ __launch_bounds__(256) void topk_merge_scan_kernel(const u64* __restrict__ cand, int n_wg, int nq, int k,
                                                          u64* __restrict__ out) {
  __shared__ u64 red[4];
  __shared__ u64 last;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long total = (long long)n_wg * k;
  u64 bound = ~0ull;  // keys are unique per (score,row); select strictly below the previous pick
  for (int i = 0; i < k; ++i) {
    u64 best = 0ull;
    for (long long c = tid; c < total; c += 256) {
      const long long wg = c / k, j = c - wg * k;
      const u64 v = cand[((size_t)wg * nq + q) * k + j];
      if (v < bound && v > best) best = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const u64 other = __shfl_xor(best, o, 64);
      best = other > best ? other : best;
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tid == 0) {
      u64 b = red[0];
      for (int w = 1; w < 4; ++w) b = red[w] > b ? red[w] : b;
      last = b;
      out[(size_t)q * k + i] = b;
    }
    __syncthreads();
    bound = last;
    if (bound == 0ull) {  // exhausted: remaining slots stay empty
      for (int j = i + 1 + tid; j < k; j += 256) out[(size_t)q * k + j] = 0ull;
      break;
    }
  }
}

// Fast merge for k <= MERGE_KMAX: every thread folds whole per-workgroup lists (sorted descending, so a list is
// abandoned at its first key that cannot enter) into a private sorted top-k in LDS -- the candidates are read
// once -- then k rounds of "largest list head wins" across the 256 private lists.
constexpr int MERGE_KMAX = 32;
__global__ __launch_bounds__(256) void topk_merge_lists_kernel(const u64* __restrict__ cand, int n_wg, int nq, int k,
                                                                u64* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char msm[];
  u64* lists = reinterpret_cast<u64*>(msm);          // [256][k]
  __shared__ u64 red[4];
  __shared__ int red_t[4];
  __shared__ int winner;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u64* mine = lists + (size_t)tid * k;
  for (int i = 0; i < k; ++i) mine[i] = 0ull;
  for (int wg = tid; wg < n_wg; wg += 256) {
    const u64* src = cand + ((size_t)wg * nq + q) * k;
    for (int j = 0; j < k; ++j) {
      const u64 v = src[j];
      if (v <= mine[k - 1]) break;     // the rest of this (sorted) list is smaller still
      insert_key(mine, k, v);
    }
  }
  int head = 0;
  for (int i = 0; i < k; ++i) {
    u64 best = head < k ? mine[head] : 0ull;
    int bt = tid;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const u64 ob = __shfl_xor(best, o, 64);
      const int ot = __shfl_xor(bt, o, 64);
      if (ob > best) {
        best = ob;
        bt = ot;
      }
    }
    if (lane == 0) {
      red[wave] = best;
      red_t[wave] = bt;
    }
    __syncthreads();
    if (tid == 0) {
      u64 b = red[0];
      int t = red_t[0];
      for (int w = 1; w < 4; ++w)
        if (red[w] > b) {
          b = red[w];
          t = red_t[w];
        }
      out[(size_t)q * k + i] = b;          // 0 once every list is exhausted
      winner = b ? t : -1;
    }
    __syncthreads();
    if (winner == tid) ++head;              // keys are unique per (score, row): exactly one owner
    __syncthreads();
  }
}

static hipError_t launch_topk_merge(const u64* cand, int n_wg, int nq, int k, u64* out, hipStream_t st) {
  if (k <= MERGE_KMAX)
    hipLaunchKernelGGL(topk_merge_lists_kernel, dim3(nq), dim3(256), (size_t)256 * k * sizeof(u64), st, cand, n_wg, nq, k, out);
  else
    hipLaunchKernelGGL(topk_merge_scan_kernel, dim3(nq), dim3(256), 0, st, cand, n_wg, nq, k, out);
  return hipGetLastError();
}

// Cross-shard merge (the step after the all-gather of per-shard top-k lists).  One wave per query: lane w
// walks list w (sorted by (score desc, id asc), -1 ids as a tail); every round the largest head wins and advances.
// `ids` are GLOBAL row ids (< 2^32), so keys of different shards never collide.
__global__ __launch_bounds__(64) void topk_merge_shards_kernel(const char* __restrict__ scores, const char* __restrict__ ids,
                                                               int n_lists, int nq, int k_in, int k_out,
                                                               long long score_stride, long long id_stride,   // bytes between lists
                                                               float* __restrict__ out_scores, long long* __restrict__ out_ids) {
  const int q = blockIdx.x, lane = threadIdx.x;
  // lists beyond 64 are folded round-robin: lane w owns lists w, w+64, ... and merges them as it goes
  int head[4] = {0, 0, 0, 0};
  auto key_at = [&](int list, int pos) -> u64 {
    if (list >= n_lists || pos >= k_in) return 0ull;
    const size_t at = (size_t)q * k_in + pos;
    const long long id = reinterpret_cast<const long long*>(ids + (size_t)list * id_stride)[at];
    return id < 0 ? 0ull : make_key(reinterpret_cast<const float*>(scores + (size_t)list * score_stride)[at], (unsigned)id);
  };
  for (int i = 0; i < k_out; ++i) {
    u64 best = 0ull;
    int slot = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u64 v = key_at(lane + 64 * s, head[s]);
      if (v > best) {
        best = v;
        slot = s;
      }
    }
    u64 win = best;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const u64 other = __shfl_xor(win, o, 64);
      win = other > win ? other : win;
    }
    if (win != 0ull && best == win) {   // global ids are unique: exactly one lane owns the winning key
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s == slot) ++head[s];
    }
    if (lane == 0) {
      out_scores[(size_t)q * k_out + i] = win ? unorderable((unsigned)(win >> 32)) : -INFINITY;
      out_ids[(size_t)q * k_out + i] = win ? (long long)(0xFFFFFFFFu - (unsigned)(win & 0xFFFFFFFFu)) : -1ll;
    }
  }
}


// Device-resident result lists (the lists a rank contributes to the all-gather): the merged keys of one search
// decoded in place of the host's decode_keys -- ids = row_map[row] (the store's local row -> global row table) or
// row + id_base; missing hits -1 / -inf.  A row outside the map (appended after the caller captured it) is reported
// as id_base + row when no map is given and as -1 otherwise.
__global__ void topk_export_keys_kernel(const u64* __restrict__ keys, long long n, const long long* __restrict__ row_map,
                                        long long n_map, long long id_base, float* __restrict__ out_scores,
                                        long long* __restrict__ out_ids) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 key = keys[i];
  if (key == 0ull) {
    out_scores[i] = -INFINITY;
    out_ids[i] = -1ll;
    return;
  }
  const long long row = (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
  long long id = id_base + row;
  if (row_map) id = row < n_map ? row_map[row] : -1ll;
  out_scores[i] = id < 0 ? -INFINITY : unorderable((unsigned)(key >> 32));
  out_ids[i] = id;
}

__global__ void topk_fill_empty_kernel(float* __restrict__ scores, long long* __restrict__ ids, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    scores[i] = -INFINITY;
    ids[i] = -1ll;
  }
}

}  // namespace vrag

using namespace vrag;

#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return VRAG_ERR_HIP;                                                            \
    }                                                                                 \
  } while (0)
#define ARG_CHECK(cond, ...)   \
  do {                         \
    if (!(cond)) {             \
      set_error(__VA_ARGS__);  \
      return VRAG_ERR_INVALID; \
    }                          \
  } while (0)

struct vrag_dense_index {
  int dim = 0, dtype = 0, device = 0;
  int64_t capacity = 0, size = 0;
  void* rows = nullptr;
  float* stage = nullptr;  // device fp32 staging for add()
  size_t stage_rows = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // scratch (grown on demand)
  float* d_q = nullptr;
  size_t d_q_elems = 0;
  u64 *d_cand = nullptr, *d_out = nullptr, *d_bound = nullptr;   // d_bound: per-query page bound (k > KMAX)
  size_t d_cand_elems = 0, d_out_elems = 0, d_bound_elems = 0;
  // tiled batched search (batches over bf16 rows: kTiledMin*): W operand of the score GEMM, candidate buffers, thresholds, flags
  bf16_t* d_tw = nullptr;
  u64 *d_tbuf = nullptr, *d_tthr = nullptr, *d_tdir = nullptr;   // d_tdir: [nq][rows of the first stage] one key per row
  u64* d_tres = nullptr;   // [TRESCUE_SLICES][nq][k] per-slice lists of the rescue pass
  size_t d_tres_elems = 0;
  float* d_tthrs = nullptr;
  unsigned* d_tcnt = nullptr;   // [nq] counters followed by [nq] overflow flags
  size_t d_tw_elems = 0, d_tbuf_elems = 0, d_tthr_elems = 0, d_tthrs_elems = 0, d_tcnt_elems = 0, d_tdir_elems = 0;
  u64* d_pfb = nullptr;            // [nq][PFCAP] candidate keys of the batch collect route (round 6: prefilter_batch_enqueue, <= 64 queries)
  size_t d_pfb_elems = 0;
  // fp32 rows with a bf16 prefilter copy (dtype 2 at creation; `dtype` stays 1: the contract is the fp32 rows')
  void* rows16 = nullptr;          // bf16 image of `rows`
  float* d_norm2 = nullptr;        // device [2]: max squared row norm, max squared image error ||x~ - x||^2 (bits ordered as unsigned: both >= 0)
  float pf_stats[2] = {0.f, 0.f};  // their host copies, refreshed by add()
  float* d_pf_eps = nullptr;       // [nq] per-query error bound of the approximate scores
  u64* d_pf_out = nullptr;         // [nq][k] exact keys of the rescored candidates
  unsigned* d_pf_flag = nullptr;   // [nq] 1 = the candidates do not provably contain the exact top-k
  size_t d_pf_eps_elems = 0, d_pf_out_elems = 0, d_pf_flag_elems = 0;
  unsigned* d_pf_cand = nullptr;   // [PFQ][PFCAP] candidate rows of the one-pass route (1 .. PFQ queries)
  u64* d_pf_keys = nullptr;        // [PFQ][PFCAP] their exact keys
  unsigned* d_pf_cnt = nullptr;    // [PFQ] candidate counters; [PFQ, 3 PFQ): scratch counters / flags of the prefix selection
  u64* d_pf_thr = nullptr;         // [0, 2 PFQ): final selection's threshold outputs (unused); [2 PFQ, 3 PFQ): entry threshold keys; [3 PFQ, 4 PFQ): their scores
  long long pf_searches = 0, pf_fallbacks = 0;
  char* h_pin = nullptr;           // pinned host staging of the one-pass route: [PFQ][dim + 1] floats up, [PFQ k + PFQ / 2] keys + flags down
  int resident_split = 0;   // the resident queries are not all bf16-exact: batched passes carry (hi, remainder) column pairs
  hipEvent_t upload_done = nullptr;   // recorded behind the query upload: the host buffer is free once it has passed
  hipEvent_t lists_done = nullptr;    // recorded behind a device-resident search: the next search (any stream) waits for it before reusing the scratch
  bool warmed = false;                // dense_warm_query_path has run (first add that brought the shard to >= 4096 rows)
};

// May a search of `nq` queries take the prefilter-image route of an fp32 index?  The route's fallback -- the full fp32 scan
// behind the per-query flags -- is gated only in the exact kernels (dense_topk_exact*_kernel): where those do not run
// (dim > 768, dim % 32 != 0, VRAG_TOPK_NO_EXACT) the scan would answer EVERY query again, image pass on top.  Batches rank the
// image with the tiled search, which wants dim % 64 == 0; other dims would rank it on the scalar 4-queries-per-pass kernel.
// Both cases keep the plain fp32 search (same bits, the faster route there).
// Largest batch the one-pass route takes: PFQ queries in one streaming pass where the multi-query collect kernel exists
// (dim % 256 == 0), else the single-query kernel once per query for one or two.
static int pf_onepass_max(int dim) { return dim % 256 == 0 ? PFQ : 2; }
static bool prefilter_route_ok(const vrag_dense_index* ix, int nq, int k) {
  if (!ix->rows16 || k > 16 || ix->size < 4096) return false;
  if (!dense_use_exact(1, ix->dim, k)) return false;                                     // the gated fallback scan exists in the exact kernels only
  if (nq > pf_onepass_max(ix->dim) && !dense_use_tiled(0, ix->dim, nq, PFK, (long long)ix->size, kTiledMinImage)) return false;   // batches rank the image with the tiled search
  return true;
}

struct vrag_sparse_index {
  int vocab = 0, device = 0;
  int64_t n_docs = 0, nnz = 0, padded = 0;
  int n_slices = 0;
  unsigned short* cols = nullptr;
  float* vals = nullptr;
  long long* slice_off = nullptr;
  int* slice_len = nullptr;
  unsigned* d_docid = nullptr;  // [n_docs] sorted position -> caller's document index (the row field of a key)
  hipStream_t stream = nullptr;
  std::mutex mu;
  float* d_q = nullptr;
  size_t d_q_elems = 0;
  u64 *d_cand = nullptr, *d_out = nullptr, *d_bound = nullptr;
  size_t d_cand_elems = 0, d_out_elems = 0, d_bound_elems = 0;
  char* d_qcsr = nullptr;             // single-query kernels: the queries' CSR (indptr | terms | weights) as uploaded, scattered into d_q on the device
  size_t d_qcsr_bytes = 0;
  // host sources of the query uploads: they live in the handle so that a call need not wait for its own uploads before it launches
  // (the NEXT call waits for upload_done before it rewrites them; by then the event has long passed)
  std::vector<char> h_blob;
  std::vector<unsigned short> h_maps;
  std::vector<float> h_wts;
  bool upload_pending = false;
  unsigned short* d_qmap = nullptr;   // batched kernel: [passes][vpad] term -> union id
  float* d_qw = nullptr;              // [passes][SQB][SUW] union id -> weight per query
  size_t d_qmap_elems = 0, d_qw_elems = 0;
  std::vector<int> pass_union;        // union size of every pass of the resident queries
  int pass_qb = 8;                    // queries per pass the resident tables were built for (8 or 16)
  bool last_multi = false;   // which kernel family the resident queries were prepared for
  hipEvent_t upload_done = nullptr;   // recorded behind the query-table uploads
  hipEvent_t lists_done = nullptr;    // as in vrag_dense_index
};

namespace {

__global__ void cvt_f32_bf16_flat(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (bf16_t)src[i];
}

template <typename T>
int grow(T** p, size_t* have, size_t need) {
  if (need <= *have) return VRAG_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  void* q = nullptr;
  HIP_TRY(hipMalloc(&q, need * sizeof(T)));
  *p = reinterpret_cast<T*>(q);
  *have = need;
  return VRAG_OK;
}

void decode_keys(const std::vector<u64>& keys, int nq, int k, int64_t base, const int64_t* perm, float* scores,
                 int64_t* ids) {
  for (size_t i = 0; i < (size_t)nq * k; ++i) {
    const u64 key = keys[i];
    if (key == 0ull) {
      scores[i] = -INFINITY;
      ids[i] = -1;
    } else {
      scores[i] = unorderable((unsigned)(key >> 32));
      const int64_t row = (int64_t)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
      ids[i] = perm ? perm[row] : base + row;
    }
  }
}

// k > KMAX: ceil(k / KMAX) passes of KMAX; `run_page(bound)` leaves the merged page keys [nq][KMAX] in d_out.
// After each page the last key of a full page becomes that query's exclusive bound (0 = exhausted: nothing passes).
template <typename RunPage>
int paged_search(int nq, int k, u64** d_bound, size_t* d_bound_elems, const u64* d_out, hipStream_t st, RunPage run_page,
                 float* scores, int64_t* ids) {
  int rc;
  if ((rc = grow(d_bound, d_bound_elems, (size_t)nq))) return rc;
  std::vector<u64> bound((size_t)nq, ~0ull), page((size_t)nq * KMAX), all((size_t)nq * k, 0ull);
  for (int k0 = 0; k0 < k; k0 += KMAX) {
    HIP_TRY(hipMemcpyAsync(*d_bound, bound.data(), (size_t)nq * sizeof(u64), hipMemcpyHostToDevice, st));
    if ((rc = run_page(*d_bound))) return rc;
    HIP_TRY(hipMemcpyAsync(page.data(), d_out, page.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    bool more = false;
    for (int q = 0; q < nq; ++q) {
      const int take = std::min(KMAX, k - k0);
      for (int j = 0; j < take; ++j) all[(size_t)q * k + k0 + j] = page[(size_t)q * KMAX + j];
      bound[q] = page[(size_t)q * KMAX + KMAX - 1];   // 0 when the page was not full
      more = more || bound[q] != 0ull;
    }
    if (!more) break;
  }
  decode_keys(all, nq, k, 0, nullptr, scores, ids);
  return VRAG_OK;
}

__global__ void tiled_init_kernel(int nq, u64* __restrict__ thr_key, float* __restrict__ thr_score, unsigned* __restrict__ cnt_ovf) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  thr_key[q] = 0ull;
  thr_score[q] = -INFINITY;
  cnt_ovf[q] = 0u;
  cnt_ovf[nq + q] = 0u;
  cnt_ovf[2 * nq + q] = 0u;   // the rescue pass's slice counter
}

// Collect form of the tiled search (round 6; the prefilter image of an fp32 index, <= 64 queries): instead of ranking the image for
// 64 candidates per query -- lists of 64 grow 4x per stage: seven stages, their appends and selections -- the staged search runs
// over a PREFIX of the shard only, with lists of the caller's k (three sub-round stages), its k-th approximate score t1 becomes the
// entry threshold of ONE pass over the whole shard that appends every row with a'_r >= t1 - 2 eps to the query's candidate list,
// and the caller re-scores those exactly.  Why that is enough: k prefix rows have a' >= t1, hence exact scores >= t1 - eps, so
// the exact k-th score of the shard is >= t1 - eps and every row of the exact top-k has a'_r >= e_r - eps >= t1 - 2 eps (eps: the
// image + rounded-query bound of prefilter_eps).  An overflowing list (cnt > PFCAP) flags its query for the gated full scan.
struct TiledCollect {
  const float* eps;   // [nq] device
  u64* keys;          // [nq][PFCAP] candidate keys out (approximate scores)
  unsigned* flag;     // [nq] cleared here; raised by the caller's selection on overflow
};
constexpr long long TCOLLECT_PREFIX = 65536;   // one tile round of the persistent grid

__global__ void tiled_tau_kernel(int nq, u64* __restrict__ thr_key, float* __restrict__ thr_score, const float* __restrict__ eps,
                                 unsigned* __restrict__ cnt, unsigned* __restrict__ flag) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  thr_key[q] = 0ull;                              // the key test of the append passes every key
  thr_score[q] = thr_score[q] - 2.f * eps[q];     // -inf (fewer than k prefix rows) stays -inf: everything is a candidate
  cnt[q] = 0u;
  flag[q] = 0u;
}

// The tiled batched search on the resident queries (ix->d_q, fp32): leaves the [nq, k] keys in ix->d_out.  Kernels only.
int dense_tiled_search(vrag_dense_index* ix, int nq, int k, hipStream_t st, const void* rows_bf16 = nullptr, const TiledCollect* col = nullptr,
                       bool device_rescue = true) {
  const int dim = ix->dim, pairs = ix->resident_split;
  if (!rows_bf16) rows_bf16 = ix->rows;
  const int n_cols = pairs ? 2 * nq : nq;
  // up to 128 query columns: 256 x 128 tiles on a three-stage ring (64 KB of the row stream in flight per CU instead of 32: 0.63
  // instead of 0.81 ms for 64-128 queries over 1.25 M x 768 rows; from 256 columns on the 256 x 256 tiles win: 0.83 vs 0.91 ms,
  // 8.6 vs 10.4 ms at 4 096 -- tools/probes/tiled_topk_cfg_probe.py)
  // up to 64 columns (round 6): 256 x 64 tiles on a four-stage ring, 96 KB of rows in flight: 64 queries 0.566 -> 0.546 ms, 32 generic
  // fp32 queries (column pairs) 0.512 -> 0.492 ms (profiles/r06_tiled_tile64_and_stage_plan_ab.txt)
  const int tile = n_cols <= 64 ? 2 : (n_cols <= 128 ? 1 : 0);
  const int n_pad = tile == 2 ? 64 : (tile == 1 ? 128 : (n_cols + 255) / 256 * 256);
  int rc;
  if ((rc = grow(&ix->d_tw, &ix->d_tw_elems, (size_t)n_pad * dim))) return rc;
  if ((rc = grow(&ix->d_tbuf, &ix->d_tbuf_elems, (size_t)nq * TCAP))) return rc;
  if ((rc = grow(&ix->d_tthr, &ix->d_tthr_elems, (size_t)nq))) return rc;
  if ((rc = grow(&ix->d_tthrs, &ix->d_tthrs_elems, (size_t)nq))) return rc;
  if ((rc = grow(&ix->d_tcnt, &ix->d_tcnt_elems, (size_t)3 * nq))) return rc;   // counters, overflow flags, the rescue's slice counters
  const int res_slices = std::max(8, std::min(TRESCUE_SLICES, 8192 / nq));
  if (!col && (rc = grow(&ix->d_tres, &ix->d_tres_elems, (size_t)res_slices * nq * k))) return rc;
  unsigned* ovf = ix->d_tcnt + nq;
  hipLaunchKernelGGL(tiled_queries_kernel, dim3(n_pad), dim3(256), 0, st, ix->d_q, nq, dim, pairs, n_pad, ix->d_tw);
  hipLaunchKernelGGL(tiled_init_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, nq, ix->d_tthr, ix->d_tthrs, ix->d_tcnt);
  HIP_TRY(hipGetLastError());
  const long long n_all = (long long)ix->size;
  const long long n = col ? std::min<long long>(n_all, TCOLLECT_PREFIX) : n_all;   // collect form: the staged search covers a prefix only
  static bool attr = false;
  if (!attr) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&tiled_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                PFCAP * (int)sizeof(u64)));   // the largest buffer a selection sorts (PFCAP >= TCAP)
    attr = true;
  }
  // Rows seen grow by `ratio` per stage, and a stage admits ~k (ratio - 1) candidates per query, each one an atomic append on its
  // query's counter: at k = 64 (the prefilter's candidate lists) a ratio of 16 made the appends, not the row stream, the cost of
  // every stage (960 per query and stage)
  const int ratio = k > 16 ? TRATIO_WIDE : TRATIO;   // (the sweeps behind the two constants: profiles/r05_tiled_stage_ratio_probe.txt; small batches too: r06_dense_midbatch_probe.txt)
  // Stage plan.  A launch costs a tile ROUND -- 256 persistent workgroups x one 256-row tile each -- however few tiles its last
  // round holds (a lone tile is a chain of memory round trips: ~30 us against ~19 us for a full round inside a long launch), and an
  // appending stage pays for its ~k (ratio - 1) appends per query as same-address atomics (~0.2 us each, serialised per query:
  // 30-40 us per stage unless they hide under a long stream; profiles/r06_search_timeline.txt).  So (late round 6):
  //   * the FIRST stage is as large as it may be for free -- it is a partial round anyway and has no appends (one key per row into
  //     d_tdir, picked over by tiled_select_direct_kernel): it takes the rows the later stages leave over from whole rounds,
  //     n mod round, when that is at least TSTAGE0 and fits the key budget (else TSTAGE0 rows, and the last stage ends on a
  //     partial round); a shard or prefix of at most one round is ONE direct stage;
  //   * every later stage is a whole number of rounds, boundaries growing by `ratio`; the last stage absorbs up to twice that (a
  //     launch and a selection fewer: it admits ~2 k ratio keys per query, far below the buffer) for batches up to 256 queries --
  //     at 1 024 its extra appends cost what the launch saved (2.10 vs 2.125 ms).
  //   * when the key budget holds a whole round (every power-of-two batch up to 4 096 queries) the first stage IS a whole round: up
  //     to 256 queries the one appending launch left is the last, whose appends hide under its stream.
  // 1.25 M rows, <= 256 queries: 65 536 | 1 184 464 (two launches; rounds 5-6: 256 | 4 560 | 65 536 | 1 179 648 in four, the second
  // and third paying ~30 us of atomics each); where the keys of a round do not fit (or the list is longer than 16): 4 816 | 65 536 | 1 179 648.
  const int col_tiles = tile == 0 ? n_pad / 256 : 1;
  const long long R = 256 % col_tiles == 0 ? (long long)(256 / col_tiles) * 256 : 0;   // rows of one round of the persistent grid (0: the column tiles do not divide it)
  long long c0max = std::min<long long>(R ? R : 65536, (TDIRECT_KEYS / std::max(nq, 1)) / 256 * 256);
  if (k > 16) c0max = std::min<long long>(c0max, 8192);   // longer lists: the selection's windows grow 4x, keep it short
  const long long s0min = nq > 1024 ? 512 : TSTAGE0;   // thousands of queries: one key per row and query is 8 B x nq per row
  c0max = std::max<long long>(c0max, s0min);
  long long s0 = std::min<long long>(n, s0min);
  if (n <= c0max) s0 = n;
  else if (R && c0max >= R) s0 = R;   // the key budget holds a whole round: up to 256 queries no appending stage below the last one at all
  else if (R && n % R >= s0min && n % R <= c0max) s0 = n % R;
  // A whole-round first stage takes its 256 tiles as a SAMPLE of the shard -- every (tiles / 256)-th 256-row tile -- instead of the
  // first 65 536 rows: an entry threshold from a prefix is as good as the prefix is representative, and a corpus in topic order
  // (a query's few thousand relevant rows in one run somewhere beyond the prefix) overflowed the candidate buffer of the last stage
  // and fell back to the rescue / the gated full scan (1.5-3 ms instead of 0.4-0.6: profiles/r06_search_timeline.txt).  A run
  // longer than one sampling period (~4 900 rows at 1.25 M) has a tile in the sample; shorter runs fit the buffer.  The collect
  // form only needs the threshold; the ranking form keeps the sample's best k (their keys carry the corpus rows) and its appending
  // stages walk the tiles the sample did not take (GemmParams::topk_tile_skip): every row is scored once, as with a prefix.
  const long long tiles_full = n_all / 256;
  const int sample_stride = (R == 65536 && s0 == R && tiles_full >= 4 * 256) ? (int)(tiles_full / 256) : 1;
  const bool sampled = sample_stride > 1;
  // stage i >= 1 covers rows [b[i], b[i + 1]) -- behind a sampled first stage (ranking form) rows of the sequence of tiles the sample did
  // NOT take: the appending launches walk that sequence through GemmParams::topk_tile_skip, so no row is scored twice
  const bool skip = sampled && !col;
  const long long off = skip ? s0 : 0;   // rows the first stage took out of the sequence
  std::vector<long long> b = {0, s0 - off};
  while (b.back() < n - off) {
    long long hi = (b.back() + off) * ratio;          // rows seen so far x ratio
    if (nq <= 256 && hi * 2 >= n) hi = n;
    hi -= off;
    if (R && hi < n - off) hi = b.back() + std::max(R, (hi - b.back()) / R * R);   // whole rounds
    b.push_back(std::min<long long>(n - off, hi));
  }
  if ((rc = grow(&ix->d_tdir, &ix->d_tdir_elems, (size_t)nq * (size_t)std::max(c0max, s0)))) return rc;   // the batch size's maximum, not this shard size's: no re-allocation on a later search of a grown shard
  for (size_t stage = 0; stage + 1 < b.size(); ++stage) {
    const bool first = stage == 0;
    const long long lo = first ? 0 : b[stage], hi = first ? s0 : b[stage + 1];
    GemmParams g{};
    g.op_dtype = kOpBf16;
    g.A = reinterpret_cast<const bf16_t*>(rows_bf16) + (size_t)(skip ? 0 : lo) * dim;
    g.W = ix->d_tw;
    g.M = (int)(hi - lo);
    g.N = n_pad;
    g.K = dim;
    g.topk_thr_score = ix->d_tthrs;
    g.topk_thr_key = ix->d_tthr;
    g.topk_cnt = ix->d_tcnt;
    g.topk_buf = first ? ix->d_tdir : ix->d_tbuf;
    g.topk_cap = first ? (int)s0 : TCAP;
    g.topk_nq = nq;
    g.topk_pairs = pairs;
    g.topk_direct = first;
    g.topk_row_base = skip ? 0u : (unsigned)lo;
    g.topk_tile = tile;
    g.topk_tile_stride = first ? sample_stride : 1;
    g.topk_tile_skip = (skip && !first) ? sample_stride : 0;
    g.topk_tile0 = (skip && !first) ? (int)(lo / 256) : 0;
    HIP_TRY(launch_gemm(EPI_TOPK, g, st));
    const bool last = first ? s0 >= n : hi >= n - off;
    u64* const sel_out = (last && !col) ? ix->d_out : (u64*)nullptr;
    if (first)
      hipLaunchKernelGGL(tiled_select_direct_kernel, dim3(nq), dim3(TSEL_NT), (size_t)TCAP * sizeof(u64), st, ix->d_tdir, (int)s0, (int)s0, ix->d_tbuf,
                         ix->d_tcnt, TCAP, k, ix->d_tthr, ix->d_tthrs, sel_out, ovf);
    else
      hipLaunchKernelGGL(tiled_select_kernel, dim3(nq), dim3(256), (size_t)TCAP * sizeof(u64), st, ix->d_tbuf, ix->d_tcnt, TCAP, k,
                         ix->d_tthr, ix->d_tthrs, sel_out, ovf, 0);
    HIP_TRY(hipGetLastError());
  }
  if (col) {
    hipLaunchKernelGGL(tiled_tau_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, nq, ix->d_tthr, ix->d_tthrs, col->eps, ix->d_tcnt, col->flag);
    GemmParams g{};
    g.op_dtype = kOpBf16;
    g.A = reinterpret_cast<const bf16_t*>(rows_bf16);
    g.W = ix->d_tw;
    g.M = (int)n_all;
    g.N = n_pad;
    g.K = dim;
    g.topk_thr_score = ix->d_tthrs;
    g.topk_thr_key = ix->d_tthr;
    g.topk_cnt = ix->d_tcnt;
    g.topk_buf = col->keys;
    g.topk_cap = PFCAP;
    g.topk_nq = nq;
    g.topk_pairs = pairs;
    g.topk_direct = 0;
    g.topk_row_base = 0u;
    g.topk_tile = tile;
    HIP_TRY(launch_gemm(EPI_TOPK, g, st));
    return VRAG_OK;   // the caller re-scores and selects; overflow = its flag
  }
  if (!device_rescue) return VRAG_OK;   // the host call reads the overflow flags back with the lists and re-answers flagged queries through the pass kernels
  const size_t lds = (size_t)dim * sizeof(float) + (size_t)16 * k * sizeof(u64);
  hipLaunchKernelGGL(dense_tiled_rescue_kernel, dim3(nq, res_slices), dim3(256), lds, st, reinterpret_cast<const bf16_t*>(rows_bf16), n_all, dim,
                     ix->d_q, nq, k, ovf, ix->d_tres, ix->d_tcnt + 2 * (size_t)nq, ix->d_out);
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

// One device pass (k <= KMAX) of a dense search: uploads the queries, runs phase 1 + the per-query merge and leaves the
// [nq, k] keys in ix->d_out.  Returns once the query upload has been consumed (the caller's buffer may be reused); the
// kernels are only enqueued.  Caller holds ix->mu and has set the device.
int dense_search_enqueue(vrag_dense_index* ix, const float* queries, int nq, int k, hipStream_t st, int image = 0, const TiledCollect* col = nullptr,
                         bool* host_rescue = nullptr) {
  // host_rescue (non-null: the caller reads results back anyway): set when the tiled search ran WITHOUT its device rescue pass -- the
  // caller copies the overflow flags (ix->d_tcnt + nq) with the lists and re-answers flagged queries itself
  // image: rank the bf16 prefilter image of an fp32 index instead of its rows (the approximate pass of the prefilter route);
  // 2 = with the queries rounded to bf16 instead of riding as (value, remainder) column pairs -- half the GEMM columns, the
  // rounding is part of the caller's error bound
  const int dtype = image ? 0 : ix->dtype;
  const void* rows = image ? ix->rows16 : ix->rows;
  const int n_wg = dense_n_wg(dtype, ix->dim, nq, k, ix->size);
  int rc;
  if (ix->lists_done) HIP_TRY(hipStreamWaitEvent(st, ix->lists_done, 0));   // a device-resident search may still be reading the scratch
  if ((rc = grow(&ix->d_q, &ix->d_q_elems, (size_t)nq * ix->dim))) return rc;
  if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * k))) return rc;
  if ((rc = grow(&ix->d_out, &ix->d_out_elems, (size_t)nq * k + nq))) return rc;   // + per-query entry thresholds
  const size_t lds = (size_t)DQT * ix->dim * sizeof(float) + (size_t)16 * DQT * k * sizeof(u64);
  ARG_CHECK(lds <= 160 * 1024, "dim/k too large for the LDS budget");
  HIP_TRY(hipMemcpyAsync(ix->d_q, queries, (size_t)nq * ix->dim * sizeof(float), hipMemcpyHostToDevice, st));
  if (!ix->upload_done) HIP_TRY(hipEventCreateWithFlags(&ix->upload_done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ix->upload_done, st));
  // Batched search over bf16 rows runs on the matrix cores with bf16 query operands: exact when every query element
  // is a bf16 number; otherwise the queries ride as (bf16 part, bf16 remainder) column pairs, 16 queries per pass.
  // Recorded for whatever kernel family serves these queries now or in a later run_resident with another (nq, k).
  ix->resident_split = 0;
  if (dtype == 0 && image != 2) {
    const uint32_t* bits = reinterpret_cast<const uint32_t*>(queries);
    const size_t n_el = (size_t)nq * ix->dim;
    for (size_t i = 0; i < n_el; ++i)
      if (bits[i] & 0xFFFFu) {
        ix->resident_split = 1;
        break;
      }
  }
  HIP_TRY(hipEventSynchronize(ix->upload_done));
  if (ix->size == 0) {
    HIP_TRY(hipMemsetAsync(ix->d_out, 0, (size_t)nq * k * sizeof(u64), st));
  } else if (dense_use_tiled(dtype, ix->dim, nq, k, (long long)ix->size, image ? kTiledMinImage : (ix->resident_split ? kTiledMinBf16Pairs : kTiledMinBf16))) {
    if ((rc = dense_tiled_search(ix, nq, k, st, image ? rows : nullptr, col, host_rescue == nullptr))) return rc;
    if (host_rescue) *host_rescue = true;
  } else {
    HIP_TRY(dense_launch_all(dtype, rows, (long long)ix->size, ix->dim, ix->d_q, nq, k, ix->d_cand, n_wg, st,
                             ix->d_out + (size_t)nq * k, ix->d_out, nullptr, ix->resident_split));
    HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));
    HIP_TRY(hipGetLastError());
  }
  return VRAG_OK;
}

}  // namespace

extern "C" {

int vrag_dense_index_create(int32_t dim, int64_t capacity, int32_t dtype, int32_t device, vrag_dense_index** out) {
  ARG_CHECK(out, "null argument");
  *out = nullptr;
  ARG_CHECK(dim > 0 && dim % 8 == 0 && dim <= 4096, "dim must be a multiple of 8 and <= 4096 (got %d)", dim);
  ARG_CHECK(capacity > 0 && capacity < 0xFFFFFFFFll, "capacity out of range");
  ARG_CHECK(dtype >= 0 && dtype <= 2, "dtype: 0 = bf16 rows, 1 = fp32 rows, 2 = fp32 rows + bf16 prefilter image");
  const bool prefilter = dtype == 2;
  if (prefilter) dtype = 1;   // the rows, the scores and the order are the fp32 index's; the image only narrows the scan
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible (no CPU fallback)", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  auto* ix = new vrag_dense_index();
  ix->dim = dim;
  ix->dtype = dtype;
  ix->device = device;
  ix->capacity = capacity;
  const size_t esz = dtype == 0 ? 2 : 4;
  // + two 256-row tiles: the tiled batched search reads whole GEMM tiles behind the last row (results of rows >= size are dropped)
  hipError_t e = hipMalloc(&ix->rows, ((size_t)capacity + 512) * dim * esz);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
  if (e == hipSuccess && prefilter && dim % 4 == 0) {
    e = hipMalloc(&ix->rows16, ((size_t)capacity + 512) * dim * 2);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->d_norm2, 2 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(ix->d_norm2, 0, 2 * sizeof(float));
    if (e == hipSuccess) e = hipHostMalloc((void**)&ix->h_pin, (size_t)PFQ * (dim + 1) * sizeof(float) + (PFQ * KMAX + PFQ / 2) * sizeof(u64), 0);
  }
  ix->stage_rows = std::max<size_t>(1, ((size_t)64 << 20) / ((size_t)dim * 4));
  if (e == hipSuccess) {
    void* p = nullptr;
    e = hipMalloc(&p, ix->stage_rows * dim * sizeof(float));
    ix->stage = reinterpret_cast<float*>(p);
  }
  if (e != hipSuccess) {
    set_error("dense index allocation failed: %s", hipGetErrorString(e));
    vrag_dense_index_destroy(ix);
    return VRAG_ERR_HIP;
  }
  *out = ix;
  return VRAG_OK;
}

void vrag_dense_index_destroy(vrag_dense_index* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();
  if (ix->rows) (void)hipFree(ix->rows);
  if (ix->stage) (void)hipFree(ix->stage);
  if (ix->d_q) (void)hipFree(ix->d_q);
  if (ix->d_cand) (void)hipFree(ix->d_cand);
  if (ix->d_out) (void)hipFree(ix->d_out);
  if (ix->d_bound) (void)hipFree(ix->d_bound);
  for (void* p : {(void*)ix->d_tw, (void*)ix->d_tbuf, (void*)ix->d_tdir, (void*)ix->d_tres, (void*)ix->d_tthr, (void*)ix->d_tthrs, (void*)ix->d_tcnt, (void*)ix->d_pfb, ix->rows16, (void*)ix->d_norm2,
                  (void*)ix->d_pf_eps, (void*)ix->d_pf_out, (void*)ix->d_pf_flag, (void*)ix->d_pf_cand, (void*)ix->d_pf_keys, (void*)ix->d_pf_cnt,
                  (void*)ix->d_pf_thr})
    if (p) (void)hipFree(p);
  if (ix->h_pin) (void)hipHostFree(ix->h_pin);
  if (ix->upload_done) (void)hipEventDestroy(ix->upload_done);
  if (ix->lists_done) (void)hipEventDestroy(ix->lists_done);
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  delete ix;
}

int64_t vrag_dense_index_size(vrag_dense_index* ix) { return ix ? ix->size : -1; }

// No first-call cost on the query path (the first 256-query batch of an index took 58 ms -- scratch hipMallocs,
// hipFuncSetAttribute and the first load of every kernel of the route inside the search): the ingest call that brings a shard to
// >= 4096 rows sizes the query-side scratch for batches of 256 x k = 16 and runs each route once (1, 2, 32 and 256 synthetic
// queries: one-pass prefilter / single-query kernels, the 32-query passes, the tiled search).  Ingest pays for it once per index
// (a few shard scans; kernel loading once per process); larger batches still grow the scratch on their first call.
static void dense_warm_query_path(vrag_dense_index* ix) {
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->warmed || ix->size < 4096) return;
    ix->warmed = true;
  }
  const int dim = ix->dim, k = 16, nq_max = 256;
  std::vector<float> q((size_t)nq_max * dim);
  unsigned h = 2463534242u;
  for (float& v : q) {   // xorshift: distinct, non-bf16-exact query values (the (value, remainder) route is warmed too)
    h ^= h << 13; h ^= h >> 17; h ^= h << 5;
    v = ((float)(h >> 8) / 16777216.0f - 0.5f) * 0.125f;
  }
  std::vector<float> sc((size_t)nq_max * k);
  std::vector<int64_t> id((size_t)nq_max * k);
  const long long s0 = ix->pf_searches, f0 = ix->pf_fallbacks;
  for (int nq : {1, 2, 4, 32, nq_max}) (void)vrag_dense_index_search(ix, q.data(), nq, k, sc.data(), id.data(), nullptr);
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->pf_searches = s0;      // synthetic queries say nothing about how this index's data behaves under the prefilter
  ix->pf_fallbacks = f0;
}

int vrag_dense_index_add(vrag_dense_index* ix, const float* rows, int64_t n) {
  ARG_CHECK(ix && rows && n > 0, "bad arguments");
  {
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->size + n > ix->capacity) {
    set_error("dense index full: %lld + %lld > capacity %lld", (long long)ix->size, (long long)n,
              (long long)ix->capacity);
    return VRAG_ERR_CAPACITY;
  }
  HIP_TRY(hipSetDevice(ix->device));
  const size_t dim = ix->dim;
  for (int64_t r0 = 0; r0 < n; r0 += (int64_t)ix->stage_rows) {
    const size_t nr = (size_t)std::min<int64_t>((int64_t)ix->stage_rows, n - r0);
    if (ix->dtype == 1) {
      float* dst = reinterpret_cast<float*>(ix->rows) + (size_t)(ix->size + r0) * dim;
      HIP_TRY(hipMemcpy(dst, rows + (size_t)r0 * dim, nr * dim * sizeof(float), hipMemcpyHostToDevice));
      if (ix->rows16) {
        hipLaunchKernelGGL(prefilter_image_kernel, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, 0, dst,
                           reinterpret_cast<bf16_t*>(ix->rows16) + (size_t)(ix->size + r0) * dim, (long long)nr, (int)dim, ix->d_norm2);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
      }
    } else {
      HIP_TRY(hipMemcpy(ix->stage, rows + (size_t)r0 * dim, nr * dim * sizeof(float), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(cvt_f32_bf16_flat, dim3(1024), dim3(256), 0, 0, ix->stage,
                         reinterpret_cast<bf16_t*>(ix->rows) + (size_t)(ix->size + r0) * dim, nr * dim);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
    }
  }
  ix->size += n;
  if (ix->rows16) HIP_TRY(hipMemcpy(ix->pf_stats, ix->d_norm2, 2 * sizeof(float), hipMemcpyDeviceToHost));
  }
  dense_warm_query_path(ix);
  return VRAG_OK;
}

int vrag_dense_index_add_device(vrag_dense_index* ix, const float* rows, int64_t n, void* stream) {
  ARG_CHECK(ix && rows && n > 0, "bad arguments");
  {
  std::lock_guard<std::mutex> lk(ix->mu);
  if (ix->size + n > ix->capacity) {
    set_error("dense index full: %lld + %lld > capacity %lld", (long long)ix->size, (long long)n, (long long)ix->capacity);
    return VRAG_ERR_CAPACITY;
  }
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t dim = ix->dim, off = (size_t)ix->size * dim, cnt = (size_t)n * dim;
  if (ix->dtype == 1) {
    HIP_TRY(hipMemcpyAsync(reinterpret_cast<float*>(ix->rows) + off, rows, cnt * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (ix->rows16) {
      hipLaunchKernelGGL(prefilter_image_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, rows,
                         reinterpret_cast<bf16_t*>(ix->rows16) + off, (long long)n, (int)dim, ix->d_norm2);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(ix->pf_stats, ix->d_norm2, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    }
  } else {
    hipLaunchKernelGGL(cvt_f32_bf16_flat, dim3(2048), dim3(256), 0, st, rows, reinterpret_cast<bf16_t*>(ix->rows) + off, cnt);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(st));   // ingest is not the hot path: searches on any stream may follow at once
  ix->size += n;
  }
  dense_warm_query_path(ix);
  return VRAG_OK;
}

// Batch route of the prefilter (nq >= 3), with `rescan` without a host decision: tiled search of the image with bf16-rounded queries -> 64
// candidates per query -> sufficiency test + exact re-score; then the full scan behind the per-query flags (groups of 32 queries
// without a flag leave at once) and a per-query pick.  Leaves the [nq, k] keys in ix->d_pf_out and the flags in ix->d_pf_flag.
// Worst case (every query flagged: bunched scores) = the full scan plus the tiled pass.  The bound with rounded queries:
// (prefilter_eps with the rounded-query term).
// Per-query error bound of the image scores (see "fp32 rows, bf16 prefilter"): E ||q|| + Dq max||x~|| + 4 dim 2^-24 max||x|| ||q||,
// Dq = ||bf16(q) - q|| when the pass runs on rounded queries, else 0.  Host arithmetic in double, rounded up.
static float prefilter_eps(const vrag_dense_index* ix, const float* q, bool rounded_query) {
  double q2 = 0.0, dq2 = 0.0;
  for (int i = 0; i < ix->dim; ++i) {
    q2 += (double)q[i] * q[i];
    if (rounded_query) {
      uint32_t b;
      std::memcpy(&b, &q[i], 4);
      b = (b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u;   // round to nearest even on 16 bits, as the device conversion does
      float r;
      std::memcpy(&r, &b, 4);
      const double d = (double)r - (double)q[i];
      dq2 += d * d;
    }
  }
  const double xmax = std::sqrt((double)ix->pf_stats[0]), emax = std::sqrt((double)ix->pf_stats[1]);
  const double e = emax * std::sqrt(q2) + std::sqrt(dq2) * (xmax + emax) + 4.0 * ix->dim / 16777216.0 * xmax * std::sqrt(q2);
  return (float)(e * 1.001 + 1e-30);
}

static int prefilter_rescan_enqueue(vrag_dense_index* ix, int nq, int k, hipStream_t st);
constexpr int kCollectMaxQueries = 256;   // 128 / 256 queries 0.99 / 1.33 -> 0.87 / 1.20 ms; 512 and 1 024 equal to the 64-candidate route (the exact chains grow with the lists): profiles/r06_collect_batch_probe.txt
static int prefilter_batch_enqueue(vrag_dense_index* ix, const float* queries, int nq, int k, hipStream_t st, bool rescan = true) {
  int rc;
  if ((rc = grow(&ix->d_pf_eps, &ix->d_pf_eps_elems, (size_t)nq))) return rc;
  if ((rc = grow(&ix->d_pf_out, &ix->d_pf_out_elems, (size_t)nq * k + 1))) return rc;
  if ((rc = grow(&ix->d_pf_flag, &ix->d_pf_flag_elems, (size_t)nq))) return rc;
  std::vector<float> eps((size_t)nq);
  for (int q = 0; q < nq; ++q) eps[q] = prefilter_eps(ix, queries + (size_t)q * ix->dim, /*rounded_query=*/true);
  if (nq <= kCollectMaxQueries) {
    // up to kCollectMaxQueries queries: the collect form of the tiled search (TiledCollect above) -- a staged search with lists
    // of k over a 65 536-row prefix, ONE pass over the shard that appends every row within 2 eps of the prefix's k-th score,
    // exact re-score of those lists, one selection (32 queries 0.79 -> 0.65 ms, 64: 0.94 -> 0.74; profiles/r06_collect_batch_probe.txt)
    if ((rc = grow(&ix->d_pfb, &ix->d_pfb_elems, (size_t)nq * PFCAP))) return rc;
    if (!ix->upload_done) HIP_TRY(hipEventCreateWithFlags(&ix->upload_done, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(ix->d_pf_eps, eps.data(), (size_t)nq * sizeof(float), hipMemcpyHostToDevice, st));
    const TiledCollect col{ix->d_pf_eps, ix->d_pfb, ix->d_pf_flag};
    if ((rc = dense_search_enqueue(ix, queries, nq, k, st, /*image=*/2, &col))) return rc;
    HIP_TRY(hipEventRecord(ix->upload_done, st));
    hipLaunchKernelGGL(prefilter_rescore_list_kernel<true>, dim3(PFCAP / 16, nq), dim3(256), 0, st, (const unsigned*)nullptr, ix->d_tcnt,
                       reinterpret_cast<const float*>(ix->rows), ix->dim, ix->d_q, ix->d_pfb);
    hipLaunchKernelGGL(tiled_select_kernel, dim3(nq), dim3(256), (size_t)PFCAP * sizeof(u64), st, ix->d_pfb, ix->d_tcnt, PFCAP, k, ix->d_tthr,
                       ix->d_tthrs, ix->d_pf_out, ix->d_pf_flag, 0);
    HIP_TRY(hipGetLastError());
  } else {
    if ((rc = dense_search_enqueue(ix, queries, nq, PFK, st, /*image=*/2))) return rc;
    HIP_TRY(hipMemcpyAsync(ix->d_pf_eps, eps.data(), (size_t)nq * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(ix->upload_done, st));
    hipLaunchKernelGGL(prefilter_rescore_kernel, dim3(nq), dim3(64), 0, st, ix->d_out, reinterpret_cast<const float*>(ix->rows), ix->dim,
                       ix->d_q, ix->d_pf_eps, k, PFK, ix->d_pf_out, ix->d_pf_flag);
    HIP_TRY(hipGetLastError());
  }
  if (rescan && (rc = prefilter_rescan_enqueue(ix, nq, k, st))) return rc;
  HIP_TRY(hipEventSynchronize(ix->upload_done));   // the eps upload has left the host vector
  return VRAG_OK;
}

// The full scan behind the flags of prefilter_batch_enqueue + the per-query pick (in place in ix->d_pf_out).  A caller that reads the
// flags on the host anyway (vrag_dense_index_search) enqueues it only when a flag is up: eighteen launches saved per batch of 256.
static int prefilter_rescan_enqueue(vrag_dense_index* ix, int nq, int k, hipStream_t st) {
  int rc;
  const int n_wg = dense_n_wg(ix->dtype, ix->dim, nq, k, ix->size);
  if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * k))) return rc;
  HIP_TRY(dense_launch_all(ix->dtype, ix->rows, (long long)ix->size, ix->dim, ix->d_q, nq, k, ix->d_cand, n_wg, st,
                           ix->d_out + (size_t)nq * k, ix->d_out, nullptr, 0, ix->d_pf_flag));
  HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));   // lists of unflagged queries: whatever the scratch held -- never picked
  const long long n = (long long)nq * k;
  hipLaunchKernelGGL(prefilter_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ix->d_pf_out, ix->d_out, ix->d_pf_flag, nq, k);
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

// One-pass route of the prefilter for one or two queries (see prefilter_collect_kernel): the fp32 queries and their error bounds
// are resident at dq / deps; per query  entry threshold (prefix kernel + selection) -> candidates (one pass over the image) ->
// exact keys (LDS-staged chains);  one selection for all queries writes the k best keys to out_keys[q][k] and raises
// out_flags[q] where the candidate list overflowed.  Six launches for one query, nothing else on the stream.
static_assert((PFPREFIX / PFROWS) * PFBEST <= PFCAP, "the entry-threshold keys share the candidate key buffer");
static int prefilter_single_enqueue(vrag_dense_index* ix, const float* dq, const float* deps, int nq, int k, u64* out_keys,
                                    unsigned* out_flags, hipStream_t st) {
  const int dim = ix->dim;
  int rc;
  if (!ix->d_pf_cand) {
    size_t unused = 0;
    if ((rc = grow(&ix->d_pf_cand, &unused, (size_t)PFQ * PFCAP))) return rc;
    unused = 0;
    if ((rc = grow(&ix->d_pf_keys, &unused, (size_t)PFQ * PFCAP))) return rc;
    unused = 0;
    if ((rc = grow(&ix->d_pf_cnt, &unused, (size_t)4 * PFQ))) return rc;
    unused = 0;
    if ((rc = grow(&ix->d_pf_thr, &unused, (size_t)4 * PFQ))) return rc;
  }
  const long long prefix = std::min<long long>(PFPREFIX, (long long)ix->size);
  const int n_wg0 = (int)((prefix + PFROWS - 1) / PFROWS);
  // shards of at least four prefixes: the threshold rows are a sample -- one 128-row block every size / 256 rows -- not the first 32 768
  const bool pf_sampled = (long long)ix->size >= 4 * PFPREFIX;
  const long long pf_block_stride = pf_sampled ? (long long)ix->size / n_wg0 / PFROWS * PFROWS : PFROWS;
  const long long pf_rows = pf_sampled ? (long long)ix->size : prefix;
  const int per = dense_rows_per_wg((long long)ix->size);
  const int wgs = (int)(((long long)ix->size + per - 1) / per);
  const int dimc = dim % 128 == 0 ? dim / 128 : 0;
  const bf16_t* img = reinterpret_cast<const bf16_t*>(ix->rows16);
  const size_t lds_q = (size_t)dim * sizeof(float);
  // per query q: candidate counter d_pf_cnt[q], the prefix selection's counter / overflow flag behind them, the entry threshold
  // key d_pf_thr[2 PFQ + q] and its score (float) at d_pf_thr + 3 PFQ
  unsigned* sel_cnt = ix->d_pf_cnt + PFQ;
  unsigned* sel_ovf = ix->d_pf_cnt + 2 * PFQ;
  u64* kth0 = ix->d_pf_thr + 2 * PFQ;
  float* kth_score0 = reinterpret_cast<float*>(ix->d_pf_thr + 3 * PFQ);
#define VRAG_PF_PREFIX(DC_, Q0_, NQ_) hipLaunchKernelGGL((prefilter_prefix_kernel<DC_>), dim3(n_wg0, NQ_), dim3(256), lds_q + PFROWS * sizeof(u64), st, img, \
                                                         pf_rows, dim, dq + (size_t)(Q0_) * dim, ix->d_pf_keys + (size_t)(Q0_) * PFCAP, ix->d_pf_cnt + (Q0_),  \
                                                         out_flags + (Q0_), PFCAP, pf_block_stride)
  auto prefix_launch = [&](int q0, int n) {
    if (dimc == 6) VRAG_PF_PREFIX(6, q0, n);
    else if (dimc == 3) VRAG_PF_PREFIX(3, q0, n);
    else if (dimc == 8) VRAG_PF_PREFIX(8, q0, n);
    else VRAG_PF_PREFIX(0, q0, n);
  };
#undef VRAG_PF_PREFIX
  if (nq >= 2 && nq <= PFQ && dim % 256 == 0) {
    // ONE streaming pass for the batch: five launches whatever nq (entry thresholds of all queries, their selection, the collect
    // pass, the exact re-score of every list, the final selection)
    prefix_launch(0, nq);
    hipLaunchKernelGGL(tiled_select_kernel, dim3(nq), dim3(256), (size_t)PFCAP * sizeof(u64), st, ix->d_pf_keys, sel_cnt, PFCAP, k, kth0, kth_score0,
                       (u64*)nullptr, sel_ovf, n_wg0 * PFBEST);
    const size_t lds_m = (size_t)(nq <= 2 ? 2 : 4) * dim * sizeof(float);
#define VRAG_PF_COLLECT_M(DC_, QN_) hipLaunchKernelGGL((prefilter_collect_multi_kernel<DC_, QN_>), dim3(wgs), dim3(256), lds_m, st, img, (long long)ix->size, dim, dq, \
                                                       nq, kth0, deps, ix->d_pf_cnt, ix->d_pf_cand, per)
    const int d32 = dim / 256;
    if (nq <= 2) {
      if (d32 == 3) VRAG_PF_COLLECT_M(3, 2);
      else if (d32 == 2) VRAG_PF_COLLECT_M(2, 2);
      else if (d32 == 1) VRAG_PF_COLLECT_M(1, 2);
      else return VRAG_ERR_INVALID;   // prefilter_route_ok admits dim <= 768 only
    } else {
      if (d32 == 3) VRAG_PF_COLLECT_M(3, 4);
      else if (d32 == 2) VRAG_PF_COLLECT_M(2, 4);
      else if (d32 == 1) VRAG_PF_COLLECT_M(1, 4);
      else return VRAG_ERR_INVALID;
    }
#undef VRAG_PF_COLLECT_M
    hipLaunchKernelGGL(prefilter_rescore_list_kernel<false>, dim3(PFCAP / 16, nq), dim3(256), 0, st, ix->d_pf_cand, ix->d_pf_cnt,
                       reinterpret_cast<const float*>(ix->rows), dim, dq, ix->d_pf_keys);
    HIP_TRY(hipGetLastError());
  } else {
    for (int q = 0; q < nq; ++q) {
      const float* q_dev = dq + (size_t)q * dim;
      u64* qkeys = ix->d_pf_keys + (size_t)q * PFCAP;
      unsigned* cand = ix->d_pf_cand + (size_t)q * PFCAP;
      prefix_launch(q, 1);
      hipLaunchKernelGGL(tiled_select_kernel, dim3(1), dim3(256), (size_t)PFCAP * sizeof(u64), st, qkeys, sel_cnt + q, PFCAP, k, kth0 + q,
                         kth_score0 + q, (u64*)nullptr, sel_ovf + q, n_wg0 * PFBEST);
#define VRAG_PF_COLLECT(DC_) hipLaunchKernelGGL((prefilter_collect_kernel<DC_>), dim3(wgs), dim3(256), lds_q, st, img, (long long)ix->size, dim, q_dev, kth0 + q, \
                                                 deps + q, ix->d_pf_cnt + q, cand, per)
      if (dimc == 6) VRAG_PF_COLLECT(6);
      else if (dimc == 3) VRAG_PF_COLLECT(3);
      else if (dimc == 8) VRAG_PF_COLLECT(8);
      else VRAG_PF_COLLECT(0);
#undef VRAG_PF_COLLECT
      hipLaunchKernelGGL(prefilter_rescore_list_kernel<false>, dim3(PFCAP / 16, 1), dim3(256), 0, st, cand, ix->d_pf_cnt + q,
                         reinterpret_cast<const float*>(ix->rows), dim, q_dev, qkeys);
      HIP_TRY(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(tiled_select_kernel, dim3(nq), dim3(256), (size_t)PFCAP * sizeof(u64), st, ix->d_pf_keys, ix->d_pf_cnt, PFCAP, k,
                     ix->d_pf_thr, reinterpret_cast<float*>(ix->d_pf_thr + PFQ), out_keys, out_flags, 0);
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

int vrag_dense_index_search(vrag_dense_index* ix, const float* queries, int32_t nq, int32_t k, float* scores,
                            int64_t* ids, void* stream) {
  ARG_CHECK(ix && queries && scores && ids && nq > 0, "bad arguments");
  ARG_CHECK(k > 0 && k <= KPAGED_MAX, "k must be in [1, %d] (got %d)", KPAGED_MAX, k);
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : ix->stream;
  if (ix->lists_done) HIP_TRY(hipStreamWaitEvent(st, ix->lists_done, 0));
  if (k > KMAX) {   // pages of KMAX on the scalar kernels (fp32 queries; the matrix-core paths stop at k = 16)
    const int n_wg = dense_n_wg(ix->dtype, ix->dim, nq, KMAX, ix->size);
    int rc;
    if ((rc = grow(&ix->d_q, &ix->d_q_elems, (size_t)nq * ix->dim))) return rc;
    if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * KMAX))) return rc;
    if ((rc = grow(&ix->d_out, &ix->d_out_elems, (size_t)nq * KMAX + nq))) return rc;
    ARG_CHECK((size_t)DQT * ix->dim * sizeof(float) + (size_t)16 * DQT * KMAX * sizeof(u64) <= 160 * 1024,
              "dim too large for the LDS budget");
    HIP_TRY(hipMemcpyAsync(ix->d_q, queries, (size_t)nq * ix->dim * sizeof(float), hipMemcpyHostToDevice, st));
    if (ix->size == 0) {
      for (size_t i = 0; i < (size_t)nq * k; ++i) {
        scores[i] = -INFINITY;
        ids[i] = -1;
      }
      return VRAG_OK;
    }
    return paged_search(nq, k, &ix->d_bound, &ix->d_bound_elems, ix->d_out, st, [&](const u64* bound) -> int {
      HIP_TRY(dense_launch_all(ix->dtype, ix->rows, (long long)ix->size, ix->dim, ix->d_q, nq, KMAX, ix->d_cand, n_wg, st,
                               ix->d_out + (size_t)nq * KMAX, ix->d_out, bound));
      HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, KMAX, ix->d_out, st));
      return VRAG_OK;
    }, scores, ids);
  }
  int rc;
  std::vector<u64> keys((size_t)nq * k);
  // fp32 rows with a prefilter image: rank the image for 64 candidates per query, re-score them exactly (see the kernels).
  // Where it pays: one or two queries (half the bytes of the fp32 scan) and batches the tiled search takes (the shard read once
  // instead of once per 32 queries); in between the 32-queries-per-pass exact kernel is already the faster route.  An index
  // whose data keeps failing the sufficiency test (near-duplicate rows) stops trying.
  const bool pf_live = prefilter_route_ok(ix, nq, k) && !(ix->pf_searches >= 32 && ix->pf_fallbacks * 4 > ix->pf_searches);
  if (pf_live) {
    std::vector<unsigned> flags((size_t)nq);
    if (nq <= pf_onepass_max(ix->dim)) {
      // one streaming pass over the image (per query, or -- two to PFQ queries, dim % 256 == 0 -- for all of them together:
      // prefilter_single_enqueue); queries + bounds go up in one pinned copy, keys + flags come back in one
      const int dim = ix->dim;
      float eps[PFQ];
      for (int q = 0; q < nq; ++q) eps[q] = prefilter_eps(ix, queries + (size_t)q * dim, /*rounded_query=*/false);   // fp32 query against the image
      if ((rc = grow(&ix->d_q, &ix->d_q_elems, (size_t)nq * dim + nq))) return rc;
      if ((rc = grow(&ix->d_pf_out, &ix->d_pf_out_elems, (size_t)nq * k + PFQ / 2))) return rc;
      {   // the full scan's scratch too: vrag_dense_index_run_resident may follow on these resident queries
        const int n_wg = dense_n_wg(ix->dtype, dim, nq, k, ix->size);
        if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * k))) return rc;
        if ((rc = grow(&ix->d_out, &ix->d_out_elems, (size_t)nq * k + nq))) return rc;
      }
      float* up = reinterpret_cast<float*>(ix->h_pin);
      u64* down = reinterpret_cast<u64*>(ix->h_pin + (size_t)PFQ * (dim + 1) * sizeof(float));
      std::memcpy(up, queries, (size_t)nq * dim * sizeof(float));
      for (int q = 0; q < nq; ++q) up[(size_t)nq * dim + q] = eps[q];
      HIP_TRY(hipMemcpyAsync(ix->d_q, up, ((size_t)nq * dim + nq) * sizeof(float), hipMemcpyHostToDevice, st));
      ix->resident_split = 0;
      u64* flag_word = ix->d_pf_out + (size_t)nq * k;   // [PFQ] unsigned flags behind the keys (PFQ / 2 words)
      if ((rc = prefilter_single_enqueue(ix, ix->d_q, ix->d_q + (size_t)nq * dim, nq, k, ix->d_pf_out, reinterpret_cast<unsigned*>(flag_word), st)))
        return rc;
      HIP_TRY(hipMemcpyAsync(down, ix->d_pf_out, ((size_t)nq * k + PFQ / 2) * sizeof(u64), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      std::memcpy(keys.data(), down, keys.size() * sizeof(u64));
      std::memcpy(flags.data(), down + (size_t)nq * k, (size_t)nq * sizeof(unsigned));
    } else {
      // batches: flagged queries are re-answered by the full scan behind their flags (groups of 32 without a flag leave at once),
      // so one bunched query costs one pass, not the batch's
      if ((rc = prefilter_batch_enqueue(ix, queries, nq, k, st, /*rescan=*/false))) return rc;
      HIP_TRY(hipMemcpyAsync(keys.data(), ix->d_pf_out, keys.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(flags.data(), ix->d_pf_flag, flags.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      ++ix->pf_searches;
      size_t n_bad = 0;
      for (unsigned f : flags) n_bad += f != 0u ? 1 : 0;
      if (n_bad * 4 > flags.size()) ++ix->pf_fallbacks;   // a quarter of the batch re-scanned: counts against the route
      if (n_bad) {   // the flags are on the host anyway: the scan is enqueued only when one is up
        if ((rc = prefilter_rescan_enqueue(ix, nq, k, st))) return rc;
        HIP_TRY(hipMemcpyAsync(keys.data(), ix->d_pf_out, keys.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
      }
      decode_keys(keys, nq, k, 0, nullptr, scores, ids);
      return VRAG_OK;
    }
    HIP_TRY(hipStreamSynchronize(st));   // also retires the query / eps uploads before the host buffers go out of scope
    ++ix->pf_searches;
    bool bad = false;
    for (unsigned f : flags) bad = bad || f != 0u;
    if (!bad) {
      decode_keys(keys, nq, k, 0, nullptr, scores, ids);
      return VRAG_OK;
    }
    ++ix->pf_fallbacks;   // scores bunched within the image's error bound: the full fp32 scan answers
  }
  bool host_rescue = false;
  if ((rc = dense_search_enqueue(ix, queries, nq, k, st, 0, nullptr, &host_rescue))) return rc;
  HIP_TRY(hipMemcpyAsync(keys.data(), ix->d_out, keys.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
  std::vector<unsigned> ovf;
  if (host_rescue) {
    ovf.resize((size_t)nq);
    HIP_TRY(hipMemcpyAsync(ovf.data(), ix->d_tcnt + nq, (size_t)nq * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  bool flagged = false;
  for (unsigned f : ovf) flagged = flagged || f != 0u;
  if (flagged) {
    // A candidate buffer of the tiled search overflowed (rows in topic order: a query's few thousand relevant rows in one run
    // beyond the first stage).  The flags are on the host anyway: the batch goes through the pass kernels -- per-workgroup lists,
    // nothing to overflow, 0.45 ms per 32 queries -- and the flagged queries take their lists from there (the device rescue pass
    // of the resident / sharded search walks the shard per flagged query: 2 ms for one, 8 ms for 64 of 64).
    const int n_wg = dense_n_wg(ix->dtype, ix->dim, nq, k, ix->size);
    HIP_TRY(dense_launch_all(ix->dtype, ix->rows, (long long)ix->size, ix->dim, ix->d_q, nq, k, ix->d_cand, n_wg, st,
                             ix->d_out + (size_t)nq * k, ix->d_out, nullptr, ix->resident_split));
    HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));
    HIP_TRY(hipGetLastError());
    std::vector<u64> again((size_t)nq * k);
    HIP_TRY(hipMemcpyAsync(again.data(), ix->d_out, again.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int q = 0; q < nq; ++q)
      if (ovf[q]) std::copy(again.begin() + (size_t)q * k, again.begin() + (size_t)(q + 1) * k, keys.begin() + (size_t)q * k);
  }
  decode_keys(keys, nq, k, 0, nullptr, scores, ids);
  return VRAG_OK;
}

int vrag_dense_index_search_device(vrag_dense_index* ix, const float* queries, int32_t nq, int32_t k, const int64_t* row_map,
                                   int64_t n_map, int64_t id_base, float* out_scores, int64_t* out_ids, void* stream) {
  ARG_CHECK(ix && queries && out_scores && out_ids && nq > 0, "bad arguments");
  ARG_CHECK(k > 0 && k <= KMAX, "k must be in [1, %d] for a device-resident search (got %d)", KMAX, k);
  ARG_CHECK(!row_map || n_map >= 0, "negative row map length");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  // NULL = the legacy default stream, NOT the handle's own stream: the caller's next operation (the all-gather) is
  // ordered against the stream it named, and torch's default stream IS the null stream
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc;
  const long long n = (long long)nq * k;
  const u64* result = nullptr;
  if (nq > pf_onepass_max(ix->dim) && prefilter_route_ok(ix, nq, k)) {
    // fp32 rows with a prefilter image, batch route (prefilter_batch_enqueue): nothing returns to the host
    if ((rc = prefilter_batch_enqueue(ix, queries, nq, k, st))) return rc;
    result = ix->d_pf_out;
  } else if (nq <= pf_onepass_max(ix->dim) && prefilter_route_ok(ix, nq, k)) {
    // one to PFQ queries: the one-pass route, then the full scan behind the overflow flags (its workgroups leave at once when no
    // flag is up) and the per-query pick -- as above, nothing returns to the host
    const int dim = ix->dim;
    if (ix->lists_done) HIP_TRY(hipStreamWaitEvent(st, ix->lists_done, 0));
    const int n_wg = dense_n_wg(ix->dtype, dim, nq, k, ix->size);
    if ((rc = grow(&ix->d_q, &ix->d_q_elems, (size_t)nq * dim + nq))) return rc;
    if ((rc = grow(&ix->d_pf_out, &ix->d_pf_out_elems, (size_t)nq * k + PFQ / 2))) return rc;
    if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * k))) return rc;
    if ((rc = grow(&ix->d_out, &ix->d_out_elems, (size_t)nq * k + nq))) return rc;
    float* up = reinterpret_cast<float*>(ix->h_pin);
    std::memcpy(up, queries, (size_t)nq * dim * sizeof(float));
    for (int q = 0; q < nq; ++q) up[(size_t)nq * dim + q] = prefilter_eps(ix, queries + (size_t)q * dim, /*rounded_query=*/false);
    HIP_TRY(hipMemcpyAsync(ix->d_q, up, ((size_t)nq * dim + nq) * sizeof(float), hipMemcpyHostToDevice, st));
    if (!ix->upload_done) HIP_TRY(hipEventCreateWithFlags(&ix->upload_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ix->upload_done, st));
    ix->resident_split = 0;
    unsigned* flags = reinterpret_cast<unsigned*>(ix->d_pf_out + (size_t)nq * k);
    if ((rc = prefilter_single_enqueue(ix, ix->d_q, ix->d_q + (size_t)nq * dim, nq, k, ix->d_pf_out, flags, st))) return rc;
    HIP_TRY(dense_launch_all(ix->dtype, ix->rows, (long long)ix->size, dim, ix->d_q, nq, k, ix->d_cand, n_wg, st,
                             ix->d_out + (size_t)nq * k, ix->d_out, nullptr, 0, flags));
    HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));
    hipLaunchKernelGGL(prefilter_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ix->d_pf_out, ix->d_out, flags, nq, k);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(ix->upload_done));   // the pinned staging is free for the next call
    result = ix->d_pf_out;
  } else {
    if ((rc = dense_search_enqueue(ix, queries, nq, k, st))) return rc;
    result = ix->d_out;
  }
  hipLaunchKernelGGL(topk_export_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, result, n,
                     reinterpret_cast<const long long*>(row_map), (long long)n_map, (long long)id_base, out_scores,
                     reinterpret_cast<long long*>(out_ids));
  HIP_TRY(hipGetLastError());
  if (!ix->lists_done) HIP_TRY(hipEventCreateWithFlags(&ix->lists_done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ix->lists_done, st));
  return VRAG_OK;
}

// Device-resident variant for benchmarking: queries already uploaded by a previous search call;
// runs the two kernels only (no copies, no sync).
int vrag_dense_index_run_resident(vrag_dense_index* ix, int32_t nq, int32_t k, void* stream) {
  ARG_CHECK(ix && nq > 0 && k > 0 && k <= KMAX, "bad arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  ARG_CHECK(ix->d_q && ix->d_q_elems >= (size_t)nq * ix->dim && ix->size > 0, "call vrag_dense_index_search once first");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : ix->stream;
  if (dense_use_tiled(ix->dtype, ix->dim, nq, k, (long long)ix->size, ix->resident_split ? kTiledMinBf16Pairs : kTiledMinBf16)) {
    ARG_CHECK(ix->d_out_elems >= (size_t)nq * k, "scratch too small");
    return dense_tiled_search(ix, nq, k, st);
  }
  const int n_wg = dense_n_wg(ix->dtype, ix->dim, nq, k, ix->size);
  ARG_CHECK(ix->d_cand_elems >= (size_t)n_wg * nq * k && ix->d_out_elems >= (size_t)nq * k + nq, "scratch too small");
  HIP_TRY(dense_launch_all(ix->dtype, ix->rows, (long long)ix->size, ix->dim, ix->d_q, nq, k, ix->d_cand, n_wg, st,
                             ix->d_out + (size_t)nq * k, ix->d_out, nullptr, ix->resident_split));
  HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

int vrag_sparse_index_create(int32_t vocab, int64_t n_docs, const int64_t* indptr, const int32_t* indices,
                             const float* values, int32_t device, vrag_sparse_index** out) {
  ARG_CHECK(out && indptr && n_docs > 0, "bad arguments");
  *out = nullptr;
  ARG_CHECK(vocab > 0 && vocab <= 65536, "vocab must be <= 65536 (u16 term ids), got %d", vocab);
  ARG_CHECK(n_docs < 0xFFFFFFFFll - 64, "too many documents");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible (no CPU fallback)", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  const int64_t nnz = indptr[n_docs];
  ARG_CHECK(nnz == 0 || (indices && values), "null indices/values");
  // sort documents by nnz (stable) so slices have little padding
  std::vector<int64_t> perm(n_docs);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
    return (indptr[a + 1] - indptr[a]) < (indptr[b + 1] - indptr[b]);
  });
  // SELL-64 in groups of four terms: slice s holds its 64 documents' terms as ng = ceil(longest / 4) groups; group g is 64 x 4
  // consecutive (column, value) entries, lane-major -- lane l reads its document's terms 4g .. 4g+3 with ONE 8-byte and ONE
  // 16-byte load (a quarter of the load instructions of a term-per-row layout).  Padding entries are (term 0, value 0).
  const int n_slices = (int)((n_docs + 63) / 64);
  std::vector<long long> off(n_slices + 1, 0);
  std::vector<int> len(n_slices, 0);   // groups per slice
  for (int s = 0; s < n_slices; ++s) {
    const int64_t last = std::min<int64_t>(n_docs, (int64_t)(s + 1) * 64) - 1;
    const int64_t longest = indptr[perm[last] + 1] - indptr[perm[last]];  // sorted ascending: last doc is the longest
    len[s] = (int)((longest + 3) / 4);
    off[s + 1] = off[s] + (long long)len[s] * 256;
  }
  const size_t padded = (size_t)off[n_slices];
  std::vector<unsigned short> cols(std::max<size_t>(padded, 256), 0);
  std::vector<float> vals(std::max<size_t>(padded, 256), 0.f);
  for (int s = 0; s < n_slices; ++s) {
    for (int l = 0; l < 64; ++l) {
      const int64_t p = (int64_t)s * 64 + l;
      if (p >= n_docs) break;
      const int64_t d = perm[p];
      const int64_t a = indptr[d], b = indptr[d + 1];
      for (int64_t j = a; j < b; ++j) {
        const int32_t t = indices[j];
        if (t < 0 || t >= vocab) {
          set_error("document %lld: term id %d outside the vocabulary", (long long)d, t);
          return VRAG_ERR_INVALID;
        }
        const size_t at = (size_t)off[s] + ((size_t)((j - a) >> 2) * 64 + l) * 4 + (size_t)((j - a) & 3);
        cols[at] = (unsigned short)t;
        vals[at] = values[j];
      }
    }
  }
  auto* ix = new vrag_sparse_index();
  ix->vocab = vocab;
  ix->device = device;
  ix->n_docs = n_docs;
  ix->nnz = nnz;
  ix->padded = (int64_t)padded;
  ix->n_slices = n_slices;
  std::vector<unsigned> docid(std::max<size_t>(1, (size_t)n_docs));
  for (int64_t p = 0; p < n_docs; ++p) docid[p] = (unsigned)perm[p];
  hipError_t e = hipMalloc((void**)&ix->cols, cols.size() * sizeof(unsigned short));
  if (e == hipSuccess) e = hipMalloc((void**)&ix->d_docid, docid.size() * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemcpy(ix->d_docid, docid.data(), docid.size() * sizeof(unsigned), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&ix->vals, vals.size() * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&ix->slice_off, off.size() * sizeof(long long));
  if (e == hipSuccess) e = hipMalloc((void**)&ix->slice_len, std::max<size_t>(1, len.size()) * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(ix->cols, cols.data(), cols.size() * sizeof(unsigned short), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ix->vals, vals.data(), vals.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ix->slice_off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ix->slice_len, len.data(), len.size() * sizeof(int), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    set_error("sparse index allocation failed: %s", hipGetErrorString(e));
    vrag_sparse_index_destroy(ix);
    return VRAG_ERR_HIP;
  }
  *out = ix;
  return VRAG_OK;
}

void vrag_sparse_index_destroy(vrag_sparse_index* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();
  if (ix->cols) (void)hipFree(ix->cols);
  if (ix->vals) (void)hipFree(ix->vals);
  if (ix->slice_off) (void)hipFree(ix->slice_off);
  if (ix->slice_len) (void)hipFree(ix->slice_len);
  if (ix->d_q) (void)hipFree(ix->d_q);
  if (ix->d_qcsr) (void)hipFree(ix->d_qcsr);
  if (ix->d_qmap) (void)hipFree(ix->d_qmap);
  if (ix->d_qw) (void)hipFree(ix->d_qw);
  if (ix->d_cand) (void)hipFree(ix->d_cand);
  if (ix->d_out) (void)hipFree(ix->d_out);
  if (ix->d_bound) (void)hipFree(ix->d_bound);
  if (ix->d_docid) (void)hipFree(ix->d_docid);
  if (ix->upload_done) (void)hipEventDestroy(ix->upload_done);
  if (ix->lists_done) (void)hipEventDestroy(ix->lists_done);
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  delete ix;
}

int vrag_sparse_index_stats(vrag_sparse_index* ix, int64_t* n_docs, int64_t* nnz, int64_t* padded_nnz) {
  ARG_CHECK(ix && n_docs && nnz && padded_nnz, "null argument");
  *n_docs = ix->n_docs;
  *nnz = ix->nnz;
  *padded_nnz = ix->padded;
  return VRAG_OK;
}

static int sparse_slices_per_wg(const vrag_sparse_index* ix) {
  // ONE workgroup per CU (the term map / dense query vector fills the LDS, so one is resident anyway), each a multiple of its
  // 16 waves.  Every wave keeps a top-k list per query, and a list that sees few documents spends its time filling: 512
  // workgroups measured 19 % slower than 256, 1 024 42 % (profiles/r05_sparse_probes.txt).
  const int target_wgs = 256;
  int spw = (ix->n_slices + target_wgs - 1) / target_wgs;
  spw = std::max(16, (spw + 15) / 16 * 16);
  return spw;
}

// Queries per pass of the batched sparse kernel: 16 when the term map, the weight table and the lists fit the LDS (the 30 522-term
// vocabulary at k <= 32), else 8.  (Round 2 measured 16 slower -- 3.7 vs 2.0 ms for 64 queries: its accumulators and 32
// single-term loads per step did not fit 128 registers.  Round 5: four terms per load pair, two queries per v_pk_fma_f32.)
constexpr int SQB_MAX = 16;
static size_t sparse_multi_lds(int vocab, int qb, int k, int pad) {
  const int vpad = (vocab + 7) & ~7;
  return (size_t)vpad * 2 + (size_t)(qb + pad) * SUW * 4 + (size_t)16 * qb * k * sizeof(u64);
}
static bool sparse_multi_fits(int vocab, int qb, int k, int pad = 0) { return sparse_multi_lds(vocab, qb, k, pad) <= 160 * 1024; }
static int sparse_pass_queries(int vocab, int k) {
  return sparse_multi_fits(vocab, 16, k) ? 16 : 8;   // (16 vs 8 queries per pass: profiles/r05_sparse_probes.txt)
}

static int sparse_launch(vrag_sparse_index* ix, int nq, int k, hipStream_t st, int* n_wg_out, const u64* bound = nullptr) {
  const int slices_per_wg = sparse_slices_per_wg(ix);
  const int n_wg = std::max(1, (ix->n_slices + slices_per_wg - 1) / slices_per_wg);
  *n_wg_out = n_wg;

  if (ix->last_multi) {
    const int vpad = (ix->vocab + 7) & ~7;
    const int QB = ix->pass_qb;
    ARG_CHECK(sparse_multi_fits(ix->vocab, QB, k), "k = %d does not fit the batched pass the resident queries were prepared for", k);
    const int pad = (QB == 16 && sparse_multi_fits(ix->vocab, QB, k, 4)) ? 4 : 0;   // the kernel's own LDS layout: the tables in HBM do not change
    const size_t lds = sparse_multi_lds(ix->vocab, QB, k, pad);
    static bool attr_m = false;
    if (!attr_m) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_topk_multi_kernel<8, 16, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_topk_multi_kernel<16, 16, 0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_topk_multi_kernel<16, 16, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_m = true;
    }
    for (int q0 = 0, ps = 0; q0 < nq; q0 += QB, ++ps) {
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(n_wg), dim3(1024), lds, st, ix->cols, ix->vals, ix->slice_off, ix->slice_len, ix->n_slices,
                           (long long)ix->n_docs, ix->d_qmap + (size_t)ps * vpad, ix->d_qw + (size_t)ps * QB * SUW, ix->vocab,
                           ix->pass_union[ps], nq, q0, k, slices_per_wg, ix->d_cand, ix->d_docid);
      };
      if (QB == 16 && pad == 4) go(&sparse_topk_multi_kernel<16, 16, 4>);
      else if (QB == 16) go(&sparse_topk_multi_kernel<16, 16, 0>);
      else go(&sparse_topk_multi_kernel<8, 16, 0>);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));
    HIP_TRY(hipGetLastError());
    return VRAG_OK;
  }
  const bool ldsq = (size_t)ix->vocab * sizeof(float) + (size_t)16 * k * sizeof(u64) <= 160 * 1024;
  const size_t lds = (size_t)16 * k * sizeof(u64) + (ldsq ? (size_t)ix->vocab * sizeof(float) : 0);
  if (ldsq) {
    static bool attr_set = false;
    if (!attr_set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_topk_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
  }
  for (int q = 0; q < nq; ++q) {
    if (ldsq)
      hipLaunchKernelGGL((sparse_topk_kernel<true>), dim3(n_wg), dim3(1024), lds, st, ix->cols, ix->vals, ix->slice_off,
                         ix->slice_len, ix->n_slices, (long long)ix->n_docs, ix->d_q, ix->vocab, nq, q, k,
                         slices_per_wg, ix->d_cand, ix->d_docid, bound);
    else
      hipLaunchKernelGGL((sparse_topk_kernel<false>), dim3(n_wg), dim3(1024), lds, st, ix->cols, ix->vals,
                         ix->slice_off, ix->slice_len, ix->n_slices, (long long)ix->n_docs, ix->d_q, ix->vocab, nq, q, k,
                         slices_per_wg, ix->d_cand, ix->d_docid, bound);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(launch_topk_merge(ix->d_cand, n_wg, nq, k, ix->d_out, st));
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

// One device pass (k <= KMAX) of a sparse search: query tables up, phase 1 + per-query merge enqueued, keys left in
// ix->d_out.  Returns once the uploads have been consumed; caller holds ix->mu and has set the device.
static int sparse_search_enqueue(vrag_sparse_index* ix, const int64_t* q_indptr, const int32_t* q_indices,
                                 const float* q_values, int nq, int k, hipStream_t st) {
  const int slices_per_wg = sparse_slices_per_wg(ix);
  const int n_wg = std::max(1, (ix->n_slices + slices_per_wg - 1) / slices_per_wg);
  int rc;
  if (!ix->upload_done) HIP_TRY(hipEventCreateWithFlags(&ix->upload_done, hipEventDisableTiming));
  if (ix->lists_done) HIP_TRY(hipStreamWaitEvent(st, ix->lists_done, 0));
  if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * k))) return rc;
  if ((rc = grow(&ix->d_out, &ix->d_out_elems, (size_t)nq * k))) return rc;
  // Batched path: two or more queries; per pass of SQB queries the union of their terms gets ids 1 .. SUW-1, the
  // u16 map + weight tables + top-k lists must fit the LDS.
  const int vpad = (ix->vocab + 7) & ~7;
  const int QB = nq <= 8 ? 8 : sparse_pass_queries(ix->vocab, k);   // up to eight queries: the 8-query pass reads half the weight-row bytes per term (8 queries 0.183 -> 0.173 ms, 2-4 equal: profiles/r06_sparse_probes.txt)
  bool multi = nq >= 2 && ix->vocab <= 65535 && sparse_multi_fits(ix->vocab, QB, k);
  for (int q = 0; q < nq; ++q)
    for (int64_t j = q_indptr[q]; j < q_indptr[q + 1]; ++j)
      ARG_CHECK(q_indices[j] >= 0 && q_indices[j] < ix->vocab, "query %d: term id %d outside the vocabulary", q, q_indices[j]);
  const int n_pass = (nq + QB - 1) / QB;
  if (ix->upload_pending) {   // the previous call's uploads read the handle's host buffers
    HIP_TRY(hipEventSynchronize(ix->upload_done));
    ix->upload_pending = false;
  }
  std::vector<unsigned short>& maps = ix->h_maps;
  std::vector<float>& wts = ix->h_wts;
  std::vector<int> unions;
  if (multi) {
    maps.assign((size_t)n_pass * vpad, 0);
    wts.assign((size_t)n_pass * QB * SUW, 0.f);
    unions.assign(n_pass, 0);
    for (int ps = 0; ps < n_pass && multi; ++ps) {
      unsigned short* mp = maps.data() + (size_t)ps * vpad;
      float* wt = wts.data() + (size_t)ps * QB * SUW;
      int nu = 0;
      for (int q = ps * QB; q < std::min(nq, (ps + 1) * QB) && multi; ++q)
        for (int64_t j = q_indptr[q]; j < q_indptr[q + 1]; ++j) {
          const int t = q_indices[j];
          if (!mp[t]) {
            if (nu + 1 >= SUW) {
              multi = false;   // too many distinct terms in this group of queries: single-query kernels
              break;
            }
            mp[t] = (unsigned short)++nu;
          }
          wt[(size_t)(q - ps * QB) * SUW + mp[t]] = q_values[j];   // a repeated term keeps the last value, like the dense scatter
        }
      unions[ps] = nu;
    }
  }
  ix->last_multi = multi;
  if (multi) {
    auto regrow = [&](auto** ptr, size_t* have, size_t want, size_t esz) -> int {
      if (*have >= want) return VRAG_OK;
      if (*ptr) (void)hipFree(*ptr);
      *ptr = nullptr;
      *have = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(ptr), want * esz));
      *have = want;
      return VRAG_OK;
    };
    if ((rc = regrow(&ix->d_qmap, &ix->d_qmap_elems, maps.size(), sizeof(unsigned short)))) return rc;
    if ((rc = regrow(&ix->d_qw, &ix->d_qw_elems, wts.size(), sizeof(float)))) return rc;
    ix->pass_union = unions;
    ix->pass_qb = QB;
    HIP_TRY(hipMemcpyAsync(ix->d_qmap, maps.data(), maps.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ix->d_qw, wts.data(), wts.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(ix->upload_done, st));
    ix->upload_pending = true;   // the host tables stay in the handle: no wait here
    int nwg3 = 0;
    return sparse_launch(ix, nq, k, st, &nwg3);
  }
  // Dense query vectors [nq][vocab] for the single-query kernels: the (term, weight) pairs travel -- a few hundred bytes per query
  // instead of 4 B x vocab (122 KB at V = 30 522: ~40 us of a 0.24 ms single-query call went into building and copying zeros) --
  // and are scattered on the device into a cleared vector, one thread per query in the query's own term order (a repeated term
  // keeps its last value, as the host scatter did).
  const int64_t nnz_q = q_indptr[nq] - q_indptr[0];
  const size_t off_idx = (size_t)(nq + 1) * sizeof(int64_t), off_val = off_idx + (size_t)nnz_q * sizeof(int32_t);
  std::vector<char>& blob = ix->h_blob;
  blob.resize(off_val + (size_t)nnz_q * sizeof(float));
  {
    int64_t* ip = reinterpret_cast<int64_t*>(blob.data());
    for (int q = 0; q <= nq; ++q) ip[q] = q_indptr[q] - q_indptr[0];
    if (nnz_q > 0) {
      memcpy(blob.data() + off_idx, q_indices + q_indptr[0], (size_t)nnz_q * sizeof(int32_t));
      memcpy(blob.data() + off_val, q_values + q_indptr[0], (size_t)nnz_q * sizeof(float));
    }
  }
  if ((rc = grow(&ix->d_q, &ix->d_q_elems, (size_t)nq * ix->vocab))) return rc;
  if ((rc = grow(&ix->d_qcsr, &ix->d_qcsr_bytes, blob.size()))) return rc;
  HIP_TRY(hipMemsetAsync(ix->d_q, 0, (size_t)nq * ix->vocab * sizeof(float), st));
  HIP_TRY(hipMemcpyAsync(ix->d_qcsr, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
  HIP_TRY(hipEventRecord(ix->upload_done, st));
  hipLaunchKernelGGL(sparse_scatter_queries_kernel, dim3((nq + 63) / 64), dim3(64), 0, st, reinterpret_cast<const long long*>(ix->d_qcsr),
                     reinterpret_cast<const int*>(ix->d_qcsr + off_idx), reinterpret_cast<const float*>(ix->d_qcsr + off_val), nq, ix->vocab, ix->d_q);
  HIP_TRY(hipGetLastError());
  ix->upload_pending = true;   // the host blob stays in the handle: no wait here
  int nwg2 = 0;
  return sparse_launch(ix, nq, k, st, &nwg2);
}


int vrag_sparse_index_search(vrag_sparse_index* ix, const int64_t* q_indptr, const int32_t* q_indices,
                             const float* q_values, int32_t nq, int32_t k, float* scores, int64_t* ids, void* stream) {
  ARG_CHECK(ix && q_indptr && scores && ids && nq > 0, "bad arguments");
  ARG_CHECK(k > 0 && k <= KPAGED_MAX, "k must be in [1, %d] (got %d)", KPAGED_MAX, k);
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : ix->stream;
  if (ix->lists_done) HIP_TRY(hipStreamWaitEvent(st, ix->lists_done, 0));
  const int slices_per_wg = sparse_slices_per_wg(ix);
  const int n_wg = std::max(1, (ix->n_slices + slices_per_wg - 1) / slices_per_wg);
  int rc;
  if (k > KMAX) {   // pages of KMAX on the single-query kernel
    for (int q = 0; q < nq; ++q)
      for (int64_t j = q_indptr[q]; j < q_indptr[q + 1]; ++j)
        ARG_CHECK(q_indices[j] >= 0 && q_indices[j] < ix->vocab, "query %d: term id %d outside the vocabulary", q, q_indices[j]);
    if ((rc = grow(&ix->d_cand, &ix->d_cand_elems, (size_t)n_wg * nq * KMAX))) return rc;
    if ((rc = grow(&ix->d_out, &ix->d_out_elems, (size_t)nq * KMAX))) return rc;
    std::vector<float> qd((size_t)nq * ix->vocab, 0.f);
    for (int q = 0; q < nq; ++q)
      for (int64_t j = q_indptr[q]; j < q_indptr[q + 1]; ++j) qd[(size_t)q * ix->vocab + q_indices[j]] = q_values[j];
    if ((rc = grow(&ix->d_q, &ix->d_q_elems, qd.size()))) return rc;
    HIP_TRY(hipMemcpyAsync(ix->d_q, qd.data(), qd.size() * sizeof(float), hipMemcpyHostToDevice, st));
    ix->last_multi = false;
    return paged_search(nq, k, &ix->d_bound, &ix->d_bound_elems, ix->d_out, st, [&](const u64* bound) -> int {
      int nwg = 0;
      return sparse_launch(ix, nq, KMAX, st, &nwg, bound);
    }, scores, ids);   // the first page's stream sync also keeps `qd` alive until its upload has been consumed
  }
  if ((rc = sparse_search_enqueue(ix, q_indptr, q_indices, q_values, nq, k, st))) return rc;
  std::vector<u64> keys((size_t)nq * k);
  HIP_TRY(hipMemcpyAsync(keys.data(), ix->d_out, keys.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  decode_keys(keys, nq, k, 0, nullptr, scores, ids);
  return VRAG_OK;
}

int vrag_sparse_index_search_device(vrag_sparse_index* ix, const int64_t* q_indptr, const int32_t* q_indices,
                                    const float* q_values, int32_t nq, int32_t k, const int64_t* row_map, int64_t n_map,
                                    int64_t id_base, float* out_scores, int64_t* out_ids, void* stream) {
  ARG_CHECK(ix && q_indptr && out_scores && out_ids && nq > 0, "bad arguments");
  ARG_CHECK(k > 0 && k <= KMAX, "k must be in [1, %d] for a device-resident search (got %d)", KMAX, k);
  ARG_CHECK(!row_map || n_map >= 0, "negative row map length");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);   // NULL = the legacy default stream (see the dense form)
  int rc;
  if ((rc = sparse_search_enqueue(ix, q_indptr, q_indices, q_values, nq, k, st))) return rc;
  const long long n = (long long)nq * k;
  hipLaunchKernelGGL(topk_export_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ix->d_out, n,
                     reinterpret_cast<const long long*>(row_map), (long long)n_map, (long long)id_base, out_scores,
                     reinterpret_cast<long long*>(out_ids));
  HIP_TRY(hipGetLastError());
  if (!ix->lists_done) HIP_TRY(hipEventCreateWithFlags(&ix->lists_done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ix->lists_done, st));
  return VRAG_OK;
}

int vrag_sparse_index_run_resident(vrag_sparse_index* ix, int32_t nq, int32_t k, void* stream) {
  ARG_CHECK(ix && nq > 0 && k > 0 && k <= KMAX, "bad arguments");
  std::lock_guard<std::mutex> lk(ix->mu);
  ARG_CHECK(ix->last_multi ? (ix->d_qmap && (int)ix->pass_union.size() >= (nq + ix->pass_qb - 1) / ix->pass_qb)
                           : (ix->d_q && ix->d_q_elems >= (size_t)nq * ix->vocab),
            "call vrag_sparse_index_search with at least this many queries first");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = stream ? reinterpret_cast<hipStream_t>(stream) : ix->stream;
  int n_wg = 0;
  return sparse_launch(ix, nq, k, st, &n_wg);
}

// An empty contribution to the exchange (a rank that holds none of the rows): -inf / -1 lists in device memory.
int vrag_topk_fill_empty(float* scores, int64_t* ids, int64_t n, int32_t device, void* stream) {
  ARG_CHECK(scores && ids && n > 0, "bad arguments");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible (no CPU fallback)", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  hipLaunchKernelGGL(topk_fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     scores, reinterpret_cast<long long*>(ids), (long long)n);
  HIP_TRY(hipGetLastError());
  return VRAG_OK;
}

// Device-side merge of per-shard top-k lists (the step after the all-gather).
int vrag_topk_merge(const float* scores, const int64_t* ids, int32_t n_lists, int32_t nq, int32_t k_in, int32_t k_out,
                    int64_t score_list_stride, int64_t id_list_stride, float* out_scores, int64_t* out_ids, int32_t on_device,
                    int32_t device, void* stream) {
  ARG_CHECK(scores && ids && out_scores && out_ids, "null argument");
  ARG_CHECK(n_lists >= 1 && n_lists <= 256 && nq >= 1 && k_in >= 1 && k_out >= 1, "bad list geometry (1 <= n_lists <= 256)");
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible (no CPU fallback)", device);
    return VRAG_ERR_NO_DEVICE;
  }
  HIP_TRY(hipSetDevice(device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t per = (size_t)nq * k_in, n_out = (size_t)nq * k_out;
  const long long ss = score_list_stride > 0 ? score_list_stride : (long long)per * 4;
  const long long is = id_list_stride > 0 ? id_list_stride : (long long)per * 8;
  ARG_CHECK(ss >= (long long)per * 4 && ss % 4 == 0 && is >= (long long)per * 8 && is % 8 == 0, "bad list strides");
  if (on_device) {
    hipLaunchKernelGGL(topk_merge_shards_kernel, dim3(nq), dim3(64), 0, st, reinterpret_cast<const char*>(scores),
                       reinterpret_cast<const char*>(ids), n_lists, nq, k_in, k_out, ss, is, out_scores,
                       reinterpret_cast<long long*>(out_ids));
    HIP_TRY(hipGetLastError());
    return VRAG_OK;
  }
  for (int l = 0; l < n_lists; ++l) {
    const int64_t* li = reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(ids) + (size_t)l * is);
    for (size_t i = 0; i < per; ++i)
      ARG_CHECK(li[i] < 0xFFFFFFFFll, "row id %lld does not fit the 32-bit key field", (long long)li[i]);
  }
  char* buf = nullptr;
  const size_t in_i = (size_t)(n_lists - 1) * is + per * 8, in_s = (size_t)(n_lists - 1) * ss + per * 4;
  const size_t off_oi = (in_i + 7) / 8 * 8, off_s = off_oi + n_out * 8, off_os = off_s + (in_s + 7) / 8 * 8;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), off_os + n_out * 4));
  char* d_ids = buf;
  long long* d_oids = reinterpret_cast<long long*>(buf + off_oi);
  char* d_sc = buf + off_s;
  float* d_osc = reinterpret_cast<float*>(buf + off_os);
  hipError_t e = hipMemcpyAsync(d_ids, ids, in_i, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_sc, scores, in_s, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(topk_merge_shards_kernel, dim3(nq), dim3(64), 0, st, d_sc, d_ids, n_lists, nq, k_in, k_out, ss, is, d_osc, d_oids);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out_ids, d_oids, n_out * 8, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_osc, n_out * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  HIP_TRY(e);
  return VRAG_OK;
}

}  // extern "C"
