// comm.hip — the one data-path collective: all-reduce over RCCL, on the context stream.
//
// north_star: "RCCL over xGMI only for the rare explicit all-reduce Op".  The reference has no
// distributed layer (SURVEY.md §5 last row, §8e); the Op is pytensor_amd/collective.py, this is what
// its device lowering calls.  One process per GPU, one communicator per process: rank 0 draws an
// ncclUniqueId, the host side (pytensor_amd/comm.py) hands it to the other ranks over the control
// plane (torch.distributed/gloo, or any side channel), every rank calls pthip_comm_init.  The
// collective itself is enqueued on OUR stream — stream-ordered behind the kernel that produced its
// operand and ahead of the one that consumes the result, no device synchronisation, no torch
// tensors, no staging through the host.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): RCCL's
// ring runs per link, and the operands here are small (a log-likelihood shard, a gradient vector),
// so the cost is latency, not bandwidth.
//
// librccl.so is loaded lazily (dlopen) by pthip_comm_init: single-GPU use never touches it.
#include <dlfcn.h>

#include "common.h"

namespace {

// the slice of rccl.h this file uses (ABI of RCCL 2.x / ROCm 7: /opt/rocm/include/rccl/rccl.h:40-43,
// 187, 220, 260, 339, 448-470, 611) — declared here so that the build does not need the header
constexpr int kUniqueIdBytes = 128;
struct UniqueId { char internal[kUniqueIdBytes]; };
typedef void* Comm;
typedef int Result;  // ncclSuccess == 0
enum RedOp { kSum = 0, kProd = 1, kMax = 2, kMin = 3 };
enum DataType { kInt8 = 0, kUint8 = 1, kInt32 = 2, kUint32 = 3, kInt64 = 4, kUint64 = 5, kFloat16 = 6, kFloat32 = 7, kFloat64 = 8 };

struct Api {
  void* handle = nullptr;
  Result (*GetUniqueId)(UniqueId*) = nullptr;
  Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  Result (*CommDestroy)(Comm) = nullptr;
  Result (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(Result) = nullptr;
};

Api g_api;
Comm g_comm = nullptr;
int g_nranks = 1, g_rank = 0;

int load_api() {
  if (g_api.handle) return 0;
  // By absolute path first: a process that also imported torch already holds torch's BUNDLED
  // librccl (same soname, bound to torch's own copy of the HIP runtime) and a soname lookup would
  // hand that one back — its communicator then fails with "unhandled cuda error" on our context.
  void* h = nullptr;
  if (const char* p = getenv("PTHIP_RCCL_LIB")) h = dlopen(p, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    const char* root = getenv("ROCM_PATH");
    std::string path = std::string(root && *root ? root : "/opt/rocm") + "/lib/librccl.so";
    h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return pthip::set_error("pthip_comm: cannot load librccl.so (%s)", dlerror());
  Api a;
  a.handle = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString)
    return pthip::set_error("pthip_comm: librccl.so lacks a required symbol");
  g_api = a;
  return 0;
}

int rccl_check(Result r, const char* what) {
  if (r == 0) return 0;
  return pthip::set_error("%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "RCCL error");
}

int rccl_dtype(int dtype) {
  switch (dtype) {
    case PTHIP_I8: return kInt8;
    case PTHIP_U8: case PTHIP_BOOL: return kUint8;
    case PTHIP_I32: return kInt32;
    case PTHIP_U32: return kUint32;
    case PTHIP_I64: return kInt64;
    case PTHIP_U64: return kUint64;
    case PTHIP_F16: return kFloat16;
    case PTHIP_F32: return kFloat32;
    case PTHIP_F64: return kFloat64;
  }
  return -1;  // int16 / uint16: no RCCL type
}

}  // namespace

extern "C" int pthip_comm_unique_id(void* id128) {
  PTHIP_REQUIRE_INIT();
  if (int r = load_api()) return r;
  UniqueId id;
  if (int r = rccl_check(g_api.GetUniqueId(&id), "ncclGetUniqueId")) return r;
  memcpy(id128, id.internal, kUniqueIdBytes);
  return 0;
}

extern "C" int pthip_comm_init(int nranks, int rank, const void* id128) {
  PTHIP_REQUIRE_INIT();
  if (nranks < 1 || rank < 0 || rank >= nranks) return pthip::set_error("pthip_comm_init: rank %d of %d", rank, nranks);
  if (g_comm) return pthip::set_error("pthip_comm_init: a communicator already exists (pthip_comm_destroy first)");
  if (int r = load_api()) return r;
  UniqueId id;
  memcpy(id.internal, id128, kUniqueIdBytes);
  Comm c = nullptr;
  if (int r = rccl_check(g_api.CommInitRank(&c, nranks, id, rank), "ncclCommInitRank")) return r;
  g_comm = c;
  g_nranks = nranks;
  g_rank = rank;
  return 0;
}

extern "C" int pthip_comm_size(int* nranks, int* rank) {
  if (nranks) *nranks = g_comm ? g_nranks : 1;
  if (rank) *rank = g_comm ? g_rank : 0;
  return 0;
}

extern "C" int pthip_comm_destroy(void) {
  if (!g_comm) return 0;
  (void)hipStreamSynchronize(pthip::ctx().stream);
  Result r = g_api.CommDestroy(g_comm);
  g_comm = nullptr;
  g_nranks = 1;
  g_rank = 0;
  return rccl_check(r, "ncclCommDestroy");
}

// buf[i] <- reduce over ranks of buf[i], in place, on the context stream.  op: 0 sum, 1 prod,
// 2 max, 3 min (bool: max = OR, min = AND on the 0/1 bytes).  Without a communicator (one
// process) the reduction is the identity.
extern "C" int pthip_all_reduce(int dtype, int op, int64_t n, void* buf) {
  PTHIP_REQUIRE_INIT();
  if (op < 0 || op > 3) return pthip::set_error("pthip_all_reduce: unknown reduction %d", op);
  if (!g_comm || n == 0) return 0;
  if (pthip::ctx().capturing) return pthip::set_error("pthip_all_reduce: a collective cannot be captured into a hipGraph");
  const int dt = rccl_dtype(dtype);
  if (dt < 0) return pthip::set_error("pthip_all_reduce: dtype %d has no RCCL type", dtype);
  // bool travels as uint8: a byte-wise sum over ranks would leave values > 1 in a buffer typed bool
  // (the host transport returns `sum != 0`).  On canonical 0/1 bytes  sum == OR == max  and
  // prod == AND == min, which keep the bytes canonical: same bits on both transports.
  if (dtype == PTHIP_BOOL) op = (op == 0) ? 2 : (op == 1) ? 3 : op;
  return rccl_check(g_api.AllReduce(buf, buf, (size_t)n, dt, op, g_comm, pthip::ctx().stream), "ncclAllReduce");
}
