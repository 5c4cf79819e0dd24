// Byte-level text kernels of the extractor's ingest path (gfx950): sentence boundaries of many chunks at once.
//
// Replaces, for a whole batch of chunk texts, the reference's per-chunk
//   re.split(r"(?<=[.!?])\s+", text) -> strip each part -> drop the empty ones
// (packages/core/verbatim_core/extractors.py:190-195, run once per chunk and query there; here once per chunk at ingest,
// packing.split_into_sentences_batch).  Integer / byte work, HBM-bound: one lane scans one document's UTF-8 bytes and
// writes the [start, end) byte offsets of its sentences.  `\s` and str.strip() are Python's Unicode white space
// (str.isspace): U+0009-000D, U+001C-0020, U+0085, U+00A0, U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000.
#include "../../include/vrag_amd.h"

#include <hip/hip_runtime.h>
#include <cstring>

#include <cstdint>

namespace vrag {
void set_error(const char* fmt, ...);

// Length in bytes of the white-space character that starts at p[0] (n bytes are readable), 0 if it is not white space.
__device__ __forceinline__ int space_len(const unsigned char* p, long long n) {
  const unsigned c = p[0];
  if ((c >= 0x09 && c <= 0x0D) || (c >= 0x1C && c <= 0x20)) return 1;
  if (c == 0xC2 && n >= 2) return (p[1] == 0x85 || p[1] == 0xA0) ? 2 : 0;
  if (n < 3) return 0;
  if (c == 0xE1) return (p[1] == 0x9A && p[2] == 0x80) ? 3 : 0;                                         // U+1680
  if (c == 0xE2) {
    if (p[1] == 0x80) return (p[2] <= 0x8A && p[2] >= 0x80) || p[2] == 0xA8 || p[2] == 0xA9 || p[2] == 0xAF ? 3 : 0;   // U+2000-200A, 2028, 2029, 202F
    if (p[1] == 0x81) return p[2] == 0x9F ? 3 : 0;                                                     // U+205F
    return 0;
  }
  if (c == 0xE3) return (p[1] == 0x80 && p[2] == 0x80) ? 3 : 0;                                         // U+3000
  return 0;
}

// One lane per document.  A sentence ends where a run of white space follows '.', '!' or '?'; leading / trailing white space
// of a part is not part of it, empty parts vanish.  counts[d] is always exact; offsets beyond `cap` are not stored (the
// host re-splits such a document itself).
__global__ void split_sentences_kernel(const unsigned char* __restrict__ text, const long long* __restrict__ doc_off, int n_docs,
                                       int cap, int* __restrict__ counts, int* __restrict__ starts, int* __restrict__ ends) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_docs) return;
  const long long lo = doc_off[d], hi = doc_off[d + 1];
  const unsigned char* p = text + lo;
  const long long n = hi - lo;
  int count = 0;
  long long first = -1, last = -1;     // first byte / one past the last byte of non-space text of the current part
  bool after_stop = false;             // the previous character was . ! or ?
  long long i = 0;
  while (i < n) {
    const int sl = space_len(p + i, n - i);
    if (sl > 0) {
      if (after_stop) {                // a split point: the current part ends here, the whole white-space run is the separator
        if (first >= 0) {
          if (count < cap) {
            starts[(size_t)d * cap + count] = (int)first;
            ends[(size_t)d * cap + count] = (int)last;
          }
          ++count;
        }
        first = -1;
        long long j = i + sl;
        while (j < n) {
          const int s2 = space_len(p + j, n - j);
          if (s2 == 0) break;
          j += s2;
        }
        i = j;
        after_stop = false;
        continue;
      }
      i += sl;                         // white space inside a part: belongs to it unless it turns out to be trailing
      continue;
    }
    const unsigned c = p[i];
    int cl = 1;                        // bytes of this (non-space) character
    if (c >= 0xF0) cl = 4;
    else if (c >= 0xE0) cl = 3;
    else if (c >= 0xC0) cl = 2;
    if (cl > n - i) cl = (int)(n - i);
    if (first < 0) first = i;
    last = i + cl;
    after_stop = (c == '.' || c == '!' || c == '?');
    i += cl;
  }
  if (first >= 0) {
    if (count < cap) {
      starts[(size_t)d * cap + count] = (int)first;
      ends[(size_t)d * cap + count] = (int)last;
    }
    ++count;
  }
  counts[d] = count;
}

}  // namespace vrag

using namespace vrag;

extern "C" int vrag_split_sentences(const uint8_t* text, const int64_t* doc_off, int32_t n_docs, int32_t cap, int32_t* counts,
                                    int32_t* starts, int32_t* ends, int32_t device) {
  if (!text || !doc_off || !counts || !starts || !ends || n_docs <= 0 || cap <= 0) {
    set_error("vrag_split_sentences: bad arguments");
    return VRAG_ERR_INVALID;
  }
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible (no CPU fallback)", device);
    return VRAG_ERR_NO_DEVICE;
  }
  for (int d = 0; d < n_docs; ++d)
    if (doc_off[d + 1] < doc_off[d] || doc_off[d + 1] - doc_off[d] > 0x7fffffffll) {
      set_error("vrag_split_sentences: document %d has a negative or > 2 GiB length", d);
      return VRAG_ERR_INVALID;
    }
  const size_t bytes = (size_t)(doc_off[n_docs] - doc_off[0]);
  hipError_t e = hipSetDevice(device);
  unsigned char* d_text = nullptr;
  long long* d_off = nullptr;
  int *d_counts = nullptr, *d_starts = nullptr, *d_ends = nullptr;
  hipStream_t st = nullptr;
  auto fail = [&](hipError_t err) {
    set_error("vrag_split_sentences: %s", hipGetErrorString(err));
    for (void* p : {(void*)d_text, (void*)d_off, (void*)d_counts, (void*)d_starts, (void*)d_ends})
      if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
    return VRAG_ERR_HIP;
  };
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&d_text, bytes + 16);
  if (e == hipSuccess) e = hipMalloc((void**)&d_off, (size_t)(n_docs + 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&d_counts, (size_t)n_docs * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_starts, (size_t)n_docs * cap * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&d_ends, (size_t)n_docs * cap * 4);
  if (e != hipSuccess) return fail(e);
  // offsets are rebased to the first document so the text buffer can be a slice of a larger one
  if (bytes) e = hipMemcpyAsync(d_text, text + doc_off[0], bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    if (doc_off[0] == 0) {
      e = hipMemcpyAsync(d_off, doc_off, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, st);
    } else {
      set_error("vrag_split_sentences: doc_off[0] must be 0");
      (void)fail(hipSuccess);
      return VRAG_ERR_INVALID;
    }
  }
  if (e != hipSuccess) return fail(e);
  hipLaunchKernelGGL(split_sentences_kernel, dim3((n_docs + 127) / 128), dim3(128), 0, st, d_text, d_off, n_docs, cap, d_counts,
                     d_starts, d_ends);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(counts, d_counts, (size_t)n_docs * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(starts, d_starts, (size_t)n_docs * cap * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(ends, d_ends, (size_t)n_docs * cap * 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(e);
  for (void* p : {(void*)d_text, (void*)d_off, (void*)d_counts, (void*)d_starts, (void*)d_ends}) (void)hipFree(p);
  (void)hipStreamDestroy(st);
  return VRAG_OK;
}

// Host code: the question-dependent half of the packer for all pairs of one question (see include/vrag_amd.h).
extern "C" int vrag_pack_qa_pairs(const int32_t* q_ids, int32_t q_len, int32_t n_pairs, const uint64_t* tails, const uint64_t* cums,
                                  const int32_t* n_groups, int32_t budget, int32_t sep_id, int32_t* ids_out, int64_t ids_cap,
                                  int64_t* starts_out, int64_t* ends_out, int64_t ranges_cap, int32_t* seq_lens, int32_t* kept,
                                  int64_t* totals_out) {
  if (!q_ids || !tails || !cums || !n_groups || !ids_out || !starts_out || !ends_out || !seq_lens || !kept || !totals_out ||
      q_len < 0 || n_pairs < 0 || budget <= 0) {
    set_error("vrag_pack_qa_pairs: bad arguments");
    return VRAG_ERR_INVALID;
  }
  int64_t n_ids = 0, n_rng = 0;
  const int64_t room = (int64_t)budget - q_len;   // tokens left for `[SEP] sentence` groups
  for (int32_t p = 0; p < n_pairs; ++p) {
    const int32_t* tail = reinterpret_cast<const int32_t*>(tails[p]);
    const int64_t* cum = reinterpret_cast<const int64_t*>(cums[p]);
    const int32_t ng = n_groups[p];
    int32_t m = 0;
    if (room > 0 && ng > 0 && tail && cum) {   // groups with q_len + cum[i] <= budget: a prefix (cum is increasing)
      int32_t lo = 0, hi = ng;
      while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (cum[mid] <= room) lo = mid + 1;
        else hi = mid;
      }
      m = lo;
    }
    kept[p] = m;
    seq_lens[p] = 0;
    if (m == 0) continue;
    const int64_t body = cum[m - 1], n = q_len + body;
    const int32_t len = (int32_t)(n + (n < budget ? 1 : 0));   // a closing [SEP] if it fits (dataset.py:202-205)
    if (n_ids + len > ids_cap || n_rng + m > ranges_cap) {
      set_error("vrag_pack_qa_pairs: output capacity exceeded at pair %d", p);
      return VRAG_ERR_CAPACITY;
    }
    int32_t* dst = ids_out + n_ids;
    memcpy(dst, q_ids, (size_t)q_len * sizeof(int32_t));
    memcpy(dst + q_len, tail, (size_t)body * sizeof(int32_t));
    if (n < budget) dst[n] = sep_id;
    for (int32_t i = 0; i < m; ++i) {   // inclusive ranges: group i = [SEP] at q_len + cum[i-1], sentence behind it
      starts_out[n_rng + i] = q_len + (i ? cum[i - 1] : 0) + 1;
      ends_out[n_rng + i] = q_len + cum[i] - 1;
    }
    seq_lens[p] = len;
    n_ids += len;
    n_rng += m;
  }
  totals_out[0] = n_ids;
  totals_out[1] = n_rng;
  return VRAG_OK;
}
