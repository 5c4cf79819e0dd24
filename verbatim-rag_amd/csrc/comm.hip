// The one exchange step of the sharded retrieval path behind the C ABI (north_star: "RCCL all-gather over xGMI of
// per-shard top-k for the final merge"): ncclAllGather of every rank's packed [Q, k] lists on the caller's stream, then
// vrag_topk_merge in place on the gathered buffer -- no torch in the data path; a host framework (torch.distributed, MPI, a
// TCP store) is only needed to carry the 128-byte unique id from rank 0 to the other ranks once.
//
// The reference has no distributed code: these entry points replace nothing in it; they serve
// verbatim_rag/vector_stores/milvus_base.py:239-259 on a row-sharded corpus.
//
// RCCL is bound at run time (dlopen), not at link time: a Python process that has imported torch already maps torch's bundled
// librccl.so, and a second copy of the library in one process means two sets of IPC / topology state.  Order: $VRAG_RCCL_LIB,
// the copy already mapped (RTLD_NOLOAD), then the system library (/opt/rocm/lib/librccl.so.1).
#include "../../include/vrag_amd.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"

namespace vrag {
void set_error(const char* fmt, ...);

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  char path[256] = {0};
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("VRAG_RCCL_LIB");
    struct Try {
      const char* name;
      int flags;
    } tries[] = {{env, RTLD_NOW | RTLD_GLOBAL},
                 {"librccl.so", RTLD_NOW | RTLD_NOLOAD},
                 {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
                 {"librccl.so.1", RTLD_NOW | RTLD_GLOBAL},
                 {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL},
                 {"librccl.so", RTLD_NOW | RTLD_GLOBAL}};
    for (const Try& t : tries) {
      if (!t.name || !*t.name) continue;
      r.handle = dlopen(t.name, t.flags);
      if (r.handle) {
        strncpy(r.path, t.name, sizeof(r.path) - 1);
        break;
      }
    }
    if (!r.handle) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) r.handle = nullptr;
  });
  return r.handle ? &r : nullptr;
}

}  // namespace
}  // namespace vrag

using namespace vrag;

struct vrag_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  std::mutex mu;
};

#define NCCL_TRY(R, expr)                                                                   \
  do {                                                                                      \
    ncclResult_t _r = (expr);                                                               \
    if (_r != ncclSuccess) {                                                                \
      set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (R)->GetErrorString(_r));      \
      return VRAG_ERR_HIP;                                                                  \
    }                                                                                       \
  } while (0)

extern "C" {

int vrag_comm_get_unique_id(uint8_t* id) {
  if (!id) {
    set_error("null id");
    return VRAG_ERR_INVALID;
  }
  if (vrag_device_count() <= 0) {
    set_error("no HIP device visible (no CPU fallback)");
    return VRAG_ERR_NO_DEVICE;
  }
  Rccl* R = rccl();
  if (!R) {
    set_error("RCCL not found (librccl.so; set VRAG_RCCL_LIB)");
    return VRAG_ERR_NO_DEVICE;
  }
  static_assert(sizeof(ncclUniqueId) == VRAG_COMM_ID_BYTES, "unique id size");
  ncclUniqueId u;
  NCCL_TRY(R, R->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return VRAG_OK;
}

int vrag_comm_create(const uint8_t* id, int32_t rank, int32_t world, int32_t device, vrag_comm** out) {
  if (!id || !out || world <= 0 || rank < 0 || rank >= world) {
    set_error("bad communicator arguments (rank %d of %d)", rank, world);
    return VRAG_ERR_INVALID;
  }
  *out = nullptr;
  if (vrag_device_count() <= device) {
    set_error("no HIP device %d visible (no CPU fallback)", device);
    return VRAG_ERR_NO_DEVICE;
  }
  Rccl* R = rccl();
  if (!R) {
    set_error("RCCL not found (librccl.so; set VRAG_RCCL_LIB)");
    return VRAG_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed", device);
    return VRAG_ERR_HIP;
  }
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  auto* c = new vrag_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, world, device, R->GetErrorString(r));
    delete c;
    return VRAG_ERR_HIP;
  }
  *out = c;
  return VRAG_OK;
}

void vrag_comm_destroy(vrag_comm* c) {
  if (!c) return;
  Rccl* R = rccl();
  if (R && c->comm) {
    (void)hipSetDevice(c->device);
    (void)R->CommDestroy(c->comm);
  }
  delete c;
}

int vrag_comm_info(vrag_comm* c, int32_t* rank, int32_t* world, int32_t* rccl_version) {
  if (!c) {
    set_error("null communicator");
    return VRAG_ERR_INVALID;
  }
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_version) {
    int v = 0;
    Rccl* R = rccl();
    if (R && R->GetVersion) (void)R->GetVersion(&v);
    *rccl_version = v;
  }
  return VRAG_OK;
}

int vrag_comm_allgather(vrag_comm* c, const void* send, void* recv, int64_t nbytes, void* stream) {
  if (!c || !send || !recv || nbytes < 0) {
    set_error("bad all-gather arguments");
    return VRAG_ERR_INVALID;
  }
  Rccl* R = rccl();
  std::lock_guard<std::mutex> lk(c->mu);
  if (hipSetDevice(c->device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed", c->device);
    return VRAG_ERR_HIP;
  }
  if (nbytes == 0) return VRAG_OK;
  NCCL_TRY(R, R->AllGather(send, recv, (size_t)nbytes, ncclUint8, c->comm, reinterpret_cast<hipStream_t>(stream)));
  return VRAG_OK;
}

int vrag_topk_allgather_merge(vrag_comm* c, const void* payload, void* gathered, int32_t nq, int32_t k_in, int32_t k_out,
                              float* out_scores, int64_t* out_ids, void* stream) {
  if (!c || !payload || !gathered || !out_scores || !out_ids || nq <= 0 || k_in <= 0 || k_out <= 0) {
    set_error("bad exchange arguments");
    return VRAG_ERR_INVALID;
  }
  const int64_t n = (int64_t)nq * k_in;
  const int64_t nbytes = (n * 12 + 7) / 8 * 8;   // [ids i64 x n | scores f32 x n | pad]: one rank's contribution
  int rc = vrag_comm_allgather(c, payload, gathered, nbytes, stream);
  if (rc) return rc;
  const char* base = reinterpret_cast<const char*>(gathered);
  return vrag_topk_merge(reinterpret_cast<const float*>(base + n * 8), reinterpret_cast<const int64_t*>(base), c->world, nq, k_in,
                         k_out, nbytes, nbytes, out_scores, out_ids, 1, c->device, stream);
}

}  // extern "C"
