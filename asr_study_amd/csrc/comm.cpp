// C1  Gradient all-reduce over RCCL behind the C ABI (asr_comm_*).
//
// THE collective path of the product: the data-parallel step (asr_study_amd/parallel.py,
// CapiComm) reduces gradients, broadcasts parameters (sum of a zeroed staging copy) and sums
// metrics through these entry points on the communicator's own stream; torch.distributed only
// ferries the 128-byte unique id between the ranks (and stays available as a fallback behind
// ASR_COMM=torch).  One communicator per process / GPU, sum of a float buffer in place, on the
// caller's stream.  RCCL is resolved at first use with dlopen -- the library links against the
// HIP runtime only, and inside a torch process the already loaded librccl is reused -- so
// nothing changes for callers that never touch asr_comm_*.
#include "common.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

struct Id { char bytes[ASR_COMM_ID_BYTES]; };            // ncclUniqueId: 128 opaque bytes
typedef int (*get_id_t)(Id*);
typedef int (*init_rank_t)(void**, int, Id, int);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*destroy_t)(void*);
typedef const char* (*err_str_t)(int);

struct Rccl {
  void* handle = nullptr;
  get_id_t get_id = nullptr;
  init_rank_t init_rank = nullptr;
  all_reduce_t all_reduce = nullptr;
  destroy_t destroy = nullptr;
  err_str_t err_str = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("ASR_LIBRCCL"), "librccl.so.1", "librccl.so",
                           "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.get_id = (get_id_t)dlsym(r.handle, "ncclGetUniqueId");
    r.init_rank = (init_rank_t)dlsym(r.handle, "ncclCommInitRank");
    r.all_reduce = (all_reduce_t)dlsym(r.handle, "ncclAllReduce");
    r.destroy = (destroy_t)dlsym(r.handle, "ncclCommDestroy");
    r.err_str = (err_str_t)dlsym(r.handle, "ncclGetErrorString");
  });
  if (!r.handle || !r.get_id || !r.init_rank || !r.all_reduce || !r.destroy) return nullptr;
  return &r;
}

int fail(Rccl* r, const char* what, int code) {
  asr_set_error("%s failed: %s (%d)", what, (r && r->err_str) ? r->err_str(code) : "?", code);
  return ASR_ERR_LAUNCH;
}

constexpr int kNcclFloat32 = 7, kNcclSum = 0;     // ncclDataType_t / ncclRedOp_t (rccl.h)

}  // namespace

extern "C" int asr_comm_unique_id(void* id_out) {
  ASR_CHECK_ARG(id_out, "comm: null id buffer");
  Rccl* r = rccl();
  if (!r) { asr_set_error("comm: librccl not found (set ASR_LIBRCCL)"); return ASR_ERR_LAUNCH; }
  const int rc = r->get_id(reinterpret_cast<Id*>(id_out));
  return rc == 0 ? ASR_OK : fail(r, "ncclGetUniqueId", rc);
}

extern "C" int asr_comm_init(const void* id, int rank, int world, asr_comm_t* comm_out) {
  ASR_CHECK_ARG(id && comm_out && world >= 1 && rank >= 0 && rank < world, "comm: bad arguments");
  Rccl* r = rccl();
  if (!r) { asr_set_error("comm: librccl not found (set ASR_LIBRCCL)"); return ASR_ERR_LAUNCH; }
  Id uid;
  memcpy(uid.bytes, id, sizeof(uid.bytes));
  void* comm = nullptr;
  const int rc = r->init_rank(&comm, world, uid, rank);   // binds to the current HIP device
  if (rc != 0) return fail(r, "ncclCommInitRank", rc);
  *comm_out = comm;
  return ASR_OK;
}

extern "C" int asr_comm_allreduce_sum(asr_comm_t comm, float* buf, int64_t n,
                                      asr_stream_t stream) {
  ASR_CHECK_ARG(comm && buf && n > 0, "comm: bad arguments");
  Rccl* r = rccl();
  if (!r) { asr_set_error("comm: librccl not found"); return ASR_ERR_LAUNCH; }
  const int rc = r->all_reduce(buf, buf, (size_t)n, kNcclFloat32, kNcclSum, comm,
                               (hipStream_t)stream);
  return rc == 0 ? ASR_OK : fail(r, "ncclAllReduce", rc);
}

extern "C" int asr_comm_destroy(asr_comm_t comm) {
  if (!comm) return ASR_OK;
  Rccl* r = rccl();
  if (!r) return ASR_OK;
  const int rc = r->destroy(comm);
  return rc == 0 ? ASR_OK : fail(r, "ncclCommDestroy", rc);
}
