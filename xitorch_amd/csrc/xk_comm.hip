// xitorch_amd :: the global decisions of the sharded solvers as device-side collectives (RCCL over xGMI).
//
// Under batch sharding the only exchange of an iteration is what reproduces the reference's GLOBAL
// decisions: MAX of {max|resid|, flags} in davidson (xitorch/_impls/linalg/symeig.py:188-203), MAX of {max residual
// norm, "someone unconverged"} in cg / bicgstab / gmres (_impls/linalg/solve.py:157,166,301,310), SUM of the inner
// products of the flat Broyden system (_impls/optimize/root/_jacobian.py:172-182).  Issued from Python through c10d
// each of them is an interpreter round trip plus a hop through the process group's own stream; the entry points here
// enqueue the all-reduce IN PLACE on the caller's stream, right behind the kernel that produced the few doubles, so the
// host's one status read per iteration returns the already-reduced values.
//
// librccl is not a link-time dependency of libxitorch_amd.so: it is looked up at the first call — the copy the
// process already has (PyTorch brings its own, soname librccl.so.1) before the system one — and every entry point
// returns XK_ERR_UNSUPPORTED when there is none.  One communicator belongs to one stream at a time (the caller keeps a
// communicator per batch group).  No RCCL type crosses the ABI: the unique id is 128 opaque bytes, a communicator a
// void*.
#include "xk_common.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace xk {

struct XkNcclId { char internal[128]; };                       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_unique_id)(XkNcclId*);
typedef int (*fn_comm_init_rank)(void**, int, XkNcclId, int);
typedef int (*fn_comm_init_all)(void**, int, const int*);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_comm_count)(void*, int*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_init_all comm_init_all = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_comm_count comm_count = nullptr, comm_user_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
};

static void rccl_load(Rccl& r) {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (int pass = 0; pass < 2 && !r.lib; ++pass)               // pass 0: only a copy that is already mapped
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (r.lib) break;
    }
  if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) return;
  r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
  r.comm_init_all = (fn_comm_init_all)dlsym(r.lib, "ncclCommInitAll");
  r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
  r.comm_count = (fn_comm_count)dlsym(r.lib, "ncclCommCount");
  r.comm_user_rank = (fn_comm_count)dlsym(r.lib, "ncclCommUserRank");
  r.all_reduce = (fn_all_reduce)dlsym(r.lib, "ncclAllReduce");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_init_all || !r.comm_destroy || !r.all_reduce) r.lib = nullptr;
}

static Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;                                  // the look-up runs once; concurrent first callers wait for it
  std::call_once(once, rccl_load, std::ref(r));
  return r;
}

// ncclResult_t -> the ABI's convention: 0 ok, > 0 = 1000 + the library's code (disjoint from hipError_t values)
static inline int rc_of(int nccl_rc) { return nccl_rc == 0 ? XK_OK : 1000 + nccl_rc; }

// ncclDataType_t: ncclFloat32 = 7, ncclFloat64 = 8; ncclRedOp_t: ncclSum = 0, ncclMax = 2, ncclMin = 3
static int all_reduce(void* comm, void* buf, long n, int dtype, int op, void* stream) {
  Rccl& r = rccl();
  if (!r.lib) return XK_ERR_UNSUPPORTED;
  if (!comm || !buf || n < 0 || op < 0 || op > 2) return XK_ERR_ARG;
  if (n == 0) return XK_OK;
  static const int redop[3] = {0, 2, 3};
  return rc_of(r.all_reduce(buf, buf, (size_t)n, dtype, redop[op], comm, (hipStream_t)stream));
}

}  // namespace xk

extern "C" {

int xk_comm_available(void) { return xk::rccl().lib ? 1 : 0; }

int xk_comm_unique_id(void* id128) {
  xk::Rccl& r = xk::rccl();
  if (!r.lib) return XK_ERR_UNSUPPORTED;
  if (!id128) return XK_ERR_ARG;
  xk::XkNcclId id;
  const int rc = r.get_unique_id(&id);
  if (rc == 0) memcpy(id128, id.internal, 128);
  return xk::rc_of(rc);
}

int xk_comm_init_rank(const void* id128, int nranks, int rank, int device, void** comm) {
  xk::Rccl& r = xk::rccl();
  if (!r.lib) return XK_ERR_UNSUPPORTED;
  if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks || device < 0) return XK_ERR_ARG;
  int prev = -1;
  hipError_t e = hipGetDevice(&prev);
  if (e != hipSuccess) return (int)e;
  e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  xk::XkNcclId id;
  memcpy(id.internal, id128, 128);
  *comm = nullptr;
  const int rc = xk::rc_of(r.comm_init_rank(comm, nranks, id, rank));
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);   // the caller's current device is not ours to change
  return rc;
}

int xk_comm_init_all(int ndev, const int* devs, void** comms) {
  xk::Rccl& r = xk::rccl();
  if (!r.lib) return XK_ERR_UNSUPPORTED;
  if (ndev < 1 || !comms) return XK_ERR_ARG;
  return xk::rc_of(r.comm_init_all(comms, ndev, devs));
}

int xk_comm_size(void* comm, int* nranks, int* rank) {
  xk::Rccl& r = xk::rccl();
  if (!r.lib || !r.comm_count || !r.comm_user_rank) return XK_ERR_UNSUPPORTED;
  if (!comm || !nranks || !rank) return XK_ERR_ARG;
  int rc = r.comm_count(comm, nranks);
  if (rc == 0) rc = r.comm_user_rank(comm, rank);
  return xk::rc_of(rc);
}

int xk_comm_destroy(void* comm) {
  xk::Rccl& r = xk::rccl();
  if (!r.lib) return XK_ERR_UNSUPPORTED;
  if (!comm) return XK_OK;
  return xk::rc_of(r.comm_destroy(comm));
}

int xk_allreduce_f64(void* comm, double* buf, long n, int op, void* stream) {
  return xk::all_reduce(comm, buf, n, 8, op, stream);
}

int xk_allreduce_f32(void* comm, float* buf, long n, int op, void* stream) {
  return xk::all_reduce(comm, buf, n, 7, op, stream);
}

}  // extern "C"
