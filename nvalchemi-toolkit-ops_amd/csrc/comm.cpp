// comm.cpp -- mi_comm_*: the one collective of the multi-GPU path (an all-gather of per-system values over RCCL / xGMI,
// one rank per GPU) behind the C ABI, so a caller that is not a torch.distributed program can shard systems across GPUs too.
//
// RCCL is bound at run time (dlopen), not linked: inside a PyTorch process the copy torch has loaded (SONAME librccl.so.1) is the one used
// -- two RCCL instances in one process would each bootstrap their own xGMI topology -- and a process that never calls mi_comm_* needs no
// RCCL at all.  Only the five functions below are resolved; their prototypes follow the public rccl.h.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "../../include/nvalchemiops_hip.h"

void mi_set_error(const char* fmt, ...);

namespace {
constexpr int kIdBytes = 128;  // NCCL_UNIQUE_ID_BYTES
struct UniqueId { char internal[kIdBytes]; };
typedef int (*GetVersionFn)(int*);
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, void*);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*ErrorStringFn)(int);
constexpr int kFloat32 = 7, kFloat64 = 8;  // ncclFloat32 / ncclFloat64

struct Rccl {
  void* handle = nullptr;
  GetVersionFn version = nullptr;
  GetUniqueIdFn unique_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  AllGatherFn all_gather = nullptr;
  CommDestroyFn destroy = nullptr;
  ErrorStringFn error_string = nullptr;
  bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

const Rccl* rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.tried) return g_rccl.handle ? &g_rccl : nullptr;
  g_rccl.tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names)  // a copy this process already holds (torch's) wins
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    mi_set_error("mi_comm: librccl.so.1 not found (%s)", dlerror());
    return nullptr;
  }
  g_rccl.version = (GetVersionFn)dlsym(h, "ncclGetVersion");
  g_rccl.unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
  g_rccl.init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  g_rccl.all_gather = (AllGatherFn)dlsym(h, "ncclAllGather");
  g_rccl.destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  g_rccl.error_string = (ErrorStringFn)dlsym(h, "ncclGetErrorString");
  if (!g_rccl.version || !g_rccl.unique_id || !g_rccl.init_rank || !g_rccl.all_gather || !g_rccl.destroy) {
    mi_set_error("mi_comm: the loaded RCCL lacks one of ncclGetVersion / GetUniqueId / CommInitRank / AllGather / CommDestroy");
    return nullptr;
  }
  g_rccl.handle = h;
  return &g_rccl;
}

int fail(const Rccl* r, const char* what, int rc) {
  mi_set_error("mi_comm: %s failed: %s (ncclResult %d)", what, (r && r->error_string) ? r->error_string(rc) : "?", rc);
  return MI_ECOMM;
}

struct Comm { void* nccl; int nranks, rank; };

int gather(void* comm, const void* send, void* recv, size_t count, int type, void* stream) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c || !c->nccl) { mi_set_error("mi_comm_allgather: no communicator"); return MI_EINVAL; }
  if (count == 0) return MI_OK;
  if (!send || !recv) { mi_set_error("mi_comm_allgather: null buffer"); return MI_EINVAL; }
  const Rccl* r = rccl();
  if (!r) return MI_ECOMM;
  const int rc = r->all_gather(send, recv, count, type, c->nccl, stream);
  return rc == 0 ? MI_OK : fail(r, "ncclAllGather", rc);
}
}  // namespace

extern "C" {
int mi_comm_library_version(int* version) {
  if (!version) { mi_set_error("mi_comm_library_version: null output"); return MI_EINVAL; }
  const Rccl* r = rccl();
  if (!r) return MI_ECOMM;
  const int rc = r->version(version);
  return rc == 0 ? MI_OK : fail(r, "ncclGetVersion", rc);
}

int mi_comm_unique_id(void* id_out, size_t bytes) {
  if (!id_out || bytes < (size_t)kIdBytes) { mi_set_error("mi_comm_unique_id: the id buffer holds MI_COMM_ID_BYTES = %d bytes", kIdBytes); return MI_EINVAL; }
  const Rccl* r = rccl();
  if (!r) return MI_ECOMM;
  UniqueId id;
  const int rc = r->unique_id(&id);
  if (rc != 0) return fail(r, "ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, kIdBytes);
  return MI_OK;
}

int mi_comm_init(const void* id, size_t bytes, int n_ranks, int rank, void** comm_out) {
  if (!comm_out) { mi_set_error("mi_comm_init: null output"); return MI_EINVAL; }
  *comm_out = nullptr;
  if (!id || bytes < (size_t)kIdBytes) { mi_set_error("mi_comm_init: the id is the MI_COMM_ID_BYTES = %d bytes rank 0 got from mi_comm_unique_id", kIdBytes); return MI_EINVAL; }
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) { mi_set_error("mi_comm_init: rank %d of %d", rank, n_ranks); return MI_EINVAL; }
  const Rccl* r = rccl();
  if (!r) return MI_ECOMM;
  UniqueId uid;
  memcpy(uid.internal, id, kIdBytes);
  void* nccl = nullptr;
  const int rc = r->init_rank(&nccl, n_ranks, uid, rank);  // collective: every rank calls it, on the device it will launch on
  if (rc != 0) return fail(r, "ncclCommInitRank", rc);
  *comm_out = new Comm{nccl, n_ranks, rank};
  return MI_OK;
}

int mi_comm_size(const void* comm, int* n_ranks, int* rank) {
  const Comm* c = static_cast<const Comm*>(comm);
  if (!c) { mi_set_error("mi_comm_size: no communicator"); return MI_EINVAL; }
  if (n_ranks) *n_ranks = c->nranks;
  if (rank) *rank = c->rank;
  return MI_OK;
}

int mi_comm_allgather_f32(void* comm, const float* send, float* recv, size_t count_per_rank, void* stream) {
  return gather(comm, send, recv, count_per_rank, kFloat32, stream);
}

int mi_comm_allgather_f64(void* comm, const double* send, double* recv, size_t count_per_rank, void* stream) {
  return gather(comm, send, recv, count_per_rank, kFloat64, stream);
}

int mi_comm_destroy(void* comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (!c) return MI_OK;
  int rc = 0;
  const Rccl* r = rccl();
  if (r && c->nccl) rc = r->destroy(c->nccl);
  delete c;
  return rc == 0 ? MI_OK : fail(r, "ncclCommDestroy", rc);
}
}  // extern "C"
