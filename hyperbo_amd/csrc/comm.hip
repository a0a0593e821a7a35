// RCCL binding of libhbo (include/hbo.h "multi-GPU"): one process per GPU, ONE sum-all-reduce of [nll, count, grad] per
// evaluation of the task-sharded multi-task objective (hyperbo/gp_utils/objectives.py:181-195 is an independent sum over
// sub-datasets).  librccl is loaded with dlopen -- libhbo does not link against it; the unique id is handed out by the
// launcher (hyperbo_amd.parallel.SocketGroup.bcast_bytes).
#include "ctx.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

// ---- RCCL (loaded lazily; libhbo itself does not link against it) ----------------------------
struct hbo_nccl_id { char internal[HBO_UNIQUE_ID_BYTES]; };
typedef int (*fn_ncclAllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_ncclCommDestroy)(void*);
typedef const char* (*fn_ncclGetErrorString)(int);

static void* rccl_open() {
  static void* lib = nullptr;
  if (lib) return lib;
  // $HBO_RCCL_LIB: another library with the same five entry points (ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
  // ncclCommDestroy, ncclCommAbort) -- how tests/fake_rccl.c lets two ranks share ONE GPU, which RCCL itself refuses
  // TEST HOOK, not a deployment knob: honoured only together with HBO_TEST_HOOKS=1, so that a stray variable in a production
  // environment cannot redirect the collectives (a dynamic linker path can redirect librccl.so itself -- as for any shared library)
  const char* hooks = getenv("HBO_TEST_HOOKS");
  if (const char* over = getenv("HBO_RCCL_LIB")) {
    if (*over && hooks && hooks[0] == '1') { lib = dlopen(over, RTLD_NOW | RTLD_GLOBAL); return lib; }
  }
  for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  return lib;
}

extern "C" int hbo_comm_unique_id(void* out128) {
  if (!out128) return HBO_ERR_ARG;
  void* lib = rccl_open();
  if (!lib) return fail(nullptr, HBO_ERR_COMM, "librccl.so not found");
  auto f = (int (*)(hbo_nccl_id*))dlsym(lib, "ncclGetUniqueId");
  if (!f) return fail(nullptr, HBO_ERR_COMM, "ncclGetUniqueId not found");
  hbo_nccl_id id; memset(&id, 0, sizeof id);
  int rc = f(&id);
  if (rc != 0) return fail(nullptr, HBO_ERR_COMM, "ncclGetUniqueId failed");
  memcpy(out128, &id, HBO_UNIQUE_ID_BYTES);
  return HBO_OK;
}
extern "C" int hbo_comm_init(hbo_ctx* c, int rank, int nranks, const void* unique_id128) {
  if (!c || !unique_id128 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(c, HBO_ERR_ARG, "hbo_comm_init: bad argument");
  HIPCHK(c, hipSetDevice(c->device));
  void* lib = rccl_open();
  if (!lib) return fail(c, HBO_ERR_COMM, "librccl.so not found");
  c->rccl_lib = lib;
  auto f = (int (*)(void**, int, hbo_nccl_id, int))dlsym(lib, "ncclCommInitRank");
  if (!f) return fail(c, HBO_ERR_COMM, "ncclCommInitRank not found");
  hbo_nccl_id id; memcpy(&id, unique_id128, HBO_UNIQUE_ID_BYTES);
  int rc = f(&c->comm, nranks, id, rank);
  if (rc != 0) { c->comm = nullptr; return fail(c, HBO_ERR_COMM, "ncclCommInitRank failed with code " + std::to_string(rc)); }
  c->comm_aborted = false;
  return HBO_OK;
}
extern "C" int hbo_comm_allreduce_sum(hbo_ctx* c, double* buf, int32_t count) {
  if (!c || !buf || count <= 0) return fail(c, HBO_ERR_ARG, "hbo_comm_allreduce_sum: bad argument");
  if (!c->comm) return fail(c, HBO_ERR_COMM, "hbo_comm_allreduce_sum: communicator not initialised");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->comm_buf_count < count) { if (c->d_comm_buf) hipFree(c->d_comm_buf); HIPCHK(c, hipMalloc((void**)&c->d_comm_buf, sizeof(double) * count)); c->comm_buf_count = count; }
  auto f = (fn_ncclAllReduce)dlsym(c->rccl_lib, "ncclAllReduce");
  if (!f) return fail(c, HBO_ERR_COMM, "ncclAllReduce not found");
  HIPCHK(c, hipMemcpyAsync(c->d_comm_buf, buf, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  const int ncclFloat64 = 8, ncclSum = 0;
  int rc = f(c->d_comm_buf, c->d_comm_buf, (size_t)count, ncclFloat64, ncclSum, c->comm, c->stream);
  if (rc != 0) return fail(c, HBO_ERR_COMM, "ncclAllReduce failed with code " + std::to_string(rc));
  HIPCHK(c, hipMemcpyAsync(buf, c->d_comm_buf, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return HBO_OK;
}
// in place on a device buffer, on the given stream of the context; nothing is synchronised here
int comm_allreduce_device(hbo_ctx* c, double* d_buf, int count, hipStream_t st) {
  if (!c->comm) return fail(c, HBO_ERR_COMM, "all-reduce: communicator not initialised");
  auto f = (fn_ncclAllReduce)dlsym(c->rccl_lib, "ncclAllReduce");
  if (!f) return fail(c, HBO_ERR_COMM, "ncclAllReduce not found");
  const int ncclFloat64 = 8, ncclSum = 0;
  int rc = f(d_buf, d_buf, (size_t)count, ncclFloat64, ncclSum, c->comm, st);
  if (rc != 0) return fail(c, HBO_ERR_COMM, "ncclAllReduce failed with code " + std::to_string(rc));
  return HBO_OK;
}
// a rank that cannot take part in the evaluation's collective tears the communicator down: the peers' all-reduce then fails
// instead of waiting for it for ever (objective.hip: objective_impl)
int comm_abort(hbo_ctx* c) {
  if (c->comm && c->rccl_lib) {
    auto f = (fn_ncclCommDestroy)dlsym(c->rccl_lib, "ncclCommAbort");
    if (f) f(c->comm);
    c->comm = nullptr;
    c->comm_aborted = true;   // a null communicator otherwise reads as "single rank, no collective" (objective_impl)
  }
  return HBO_OK;
}
extern "C" int hbo_comm_destroy(hbo_ctx* c) {
  if (!c) return HBO_OK;
  if (c->comm && c->rccl_lib) {
    auto f = (fn_ncclCommDestroy)dlsym(c->rccl_lib, "ncclCommDestroy");
    if (f) f(c->comm);
  }
  c->comm = nullptr;
  c->comm_aborted = false;
  if (c->d_comm_buf) { hipFree(c->d_comm_buf); c->d_comm_buf = nullptr; c->comm_buf_count = 0; }
  return HBO_OK;
}
