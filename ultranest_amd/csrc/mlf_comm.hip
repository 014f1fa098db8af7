// mlf_comm.hip -- the one exchange step of the multi-GPU region rebuild behind the C ABI: MAX all-reduce of a few
// doubles over RCCL (xGMI), for callers that have no torch.distributed.  Counterpart of the
// reference's gather / bcast / np.max in `_update_region_bootstrap` (integrator.py:395-404).
//
// librccl is opened at the first call (dlopen, preferring a copy that is already in the process, e.g. the one a
// PyTorch-ROCm wheel bundles) -- libmlfriends_hip.so itself has no link-time dependency on it, so loading the library
// never drags a second RCCL into a process.  Two modes:
//   * one process per GPU (the mode of the rebuild):  rank 0 calls mlf_comm_unique_id, hands the 128 bytes to the other
//     ranks by any means (MPI_Bcast, a file, a socket), every rank calls mlf_comm_init_rank; mlf_allreduce_max then
//     reduces `count` doubles in place across the ranks.
//   * one process driving `ndev` devices (mlf_comm_init(ndev)): mlf_allreduce_max takes ndev x count doubles (one row
//     per device) and leaves the element-wise maximum in every row.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/mlfriends_hip.h"
#include "mlf_ctx.hpp"

namespace {

using namespace mlf;

// the slice of the RCCL API used here (rccl.h: ncclUniqueId = 128 opaque bytes, ncclComm_t = opaque pointer,
// ncclDouble = 8, ncclMax = 2, ncclSuccess = 0)
struct UniqueId {
  char bytes[128];
};
typedef void *Comm;
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_init_rank)(Comm *, int, UniqueId, int);
typedef int (*fn_init_all)(Comm *, int, const int *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef int (*fn_destroy)(Comm);
typedef int (*fn_group)(void);
typedef const char *(*fn_errstr)(int);
constexpr int kNcclDouble = 8, kNcclMax = 2;

struct Rccl {
  void *h = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_init_all init_all = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_destroy destroy = nullptr;
  fn_group group_start = nullptr, group_end = nullptr;
  fn_errstr errstr = nullptr;
};
Rccl g_rccl;

struct CommState {
  std::vector<Comm> comms;        // one per device driven by this process
  std::vector<int> devices;
  std::vector<hipStream_t> streams;
  std::vector<double *> bufs;
  size_t buf_count = 0;
};
CommState g_comm;

int load_rccl() {
  if (g_rccl.h) return 0;
  const char *names[] = {"librccl.so", "librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // a copy that is already loaded (torch's) wins
  if (!h)
    for (const char *n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return ctx_fail_arg(MLF_E_STATE, "librccl.so not found (needed only for mlf_comm_* / mlf_allreduce_max)");
  Rccl r;
  r.h = h;
  r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
  r.init_all = (fn_init_all)dlsym(h, "ncclCommInitAll");
  r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
  r.group_start = (fn_group)dlsym(h, "ncclGroupStart");
  r.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
  r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.init_rank || !r.init_all || !r.all_reduce || !r.destroy || !r.group_start || !r.group_end)
    return ctx_fail_arg(MLF_E_STATE, "librccl.so lacks an expected symbol");
  g_rccl = r;
  return 0;
}

int fail_rccl(int code, const char *what) {
  std::string msg = std::string("RCCL failure in ") + what + ": " +
                    (g_rccl.errstr ? g_rccl.errstr(code) : "error") + " (" + std::to_string(code) + ")";
  ctx_fail_arg(MLF_E_STATE, msg.c_str());
  return -1000 - code;
}

#define CKH(x)                                                        \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) return ctx_fail_hip(e_, #x, __FILE__, __LINE__); \
  } while (0)

int reserve_bufs(size_t count) {
  if (count <= g_comm.buf_count) return 0;
  for (size_t i = 0; i < g_comm.devices.size(); ++i) {
    CKH(hipSetDevice(g_comm.devices[i]));
    if (g_comm.bufs[i]) CKH(hipFree(g_comm.bufs[i]));
    g_comm.bufs[i] = nullptr;
    CKH(hipMalloc(reinterpret_cast<void **>(&g_comm.bufs[i]), count * sizeof(double)));
  }
  g_comm.buf_count = count;
  return 0;
}

}  // namespace

extern "C" {

int mlf_comm_unique_id(char *id_out, size_t len) {
  if (!id_out || len < 128) return ctx_fail_arg(MLF_E_BADARG, "id buffer must hold 128 bytes");
  if (int rc = load_rccl()) return rc;
  UniqueId id;
  if (int rc = g_rccl.get_unique_id(&id)) return fail_rccl(rc, "ncclGetUniqueId");
  memcpy(id_out, id.bytes, 128);
  return 0;
}

int mlf_comm_destroy(void) {
  for (size_t i = 0; i < g_comm.comms.size(); ++i) {
    if (g_comm.comms[i] && g_rccl.destroy) g_rccl.destroy(g_comm.comms[i]);
    if (g_comm.bufs[i]) {
      (void)hipSetDevice(g_comm.devices[i]);
      (void)hipFree(g_comm.bufs[i]);
      (void)hipStreamDestroy(g_comm.streams[i]);
    }
  }
  const bool had = !g_comm.devices.empty();
  const int first = had ? g_comm.devices[0] : 0;
  g_comm = CommState();
  if (had) (void)hipSetDevice(first);
  return 0;
}

static int adopt(const std::vector<int> &devices, const std::vector<Comm> &comms) {
  g_comm.devices = devices;
  g_comm.comms = comms;
  g_comm.streams.assign(devices.size(), nullptr);
  g_comm.bufs.assign(devices.size(), nullptr);
  g_comm.buf_count = 0;
  for (size_t i = 0; i < devices.size(); ++i) {
    CKH(hipSetDevice(devices[i]));
    CKH(hipStreamCreate(&g_comm.streams[i]));
  }
  CKH(hipSetDevice(devices[0]));
  return reserve_bufs(16);
}

int mlf_comm_init_rank(const char *id, size_t len, int nranks, int rank) {
  if (!id || len < 128 || nranks < 1 || rank < 0 || rank >= nranks) return ctx_fail_arg(MLF_E_BADARG, "bad communicator arguments");
  if (int rc = ctx_ensure()) return rc;
  if (int rc = load_rccl()) return rc;
  mlf_comm_destroy();
  int dev = 0;
  CKH(hipGetDevice(&dev));   // the device selected with mlf_set_device
  UniqueId uid;
  memcpy(uid.bytes, id, 128);
  Comm c = nullptr;
  if (int rc = g_rccl.init_rank(&c, nranks, uid, rank)) return fail_rccl(rc, "ncclCommInitRank");
  return adopt({dev}, {c});
}

int mlf_comm_init(int ndev) {
  int have = 0;
  CKH(hipGetDeviceCount(&have));
  if (ndev < 1 || ndev > have) return ctx_fail_arg(MLF_E_BADARG, "mlf_comm_init: ndev must be between 1 and the number of visible devices");
  if (int rc = load_rccl()) return rc;
  mlf_comm_destroy();
  std::vector<int> devices((size_t)ndev);
  for (int i = 0; i < ndev; ++i) devices[i] = i;
  std::vector<Comm> comms((size_t)ndev, nullptr);
  if (int rc = g_rccl.init_all(comms.data(), ndev, devices.data())) return fail_rccl(rc, "ncclCommInitAll");
  return adopt(devices, comms);
}

int mlf_allreduce_max(double *values, size_t count) {
  if (!values || count == 0) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (g_comm.comms.empty()) return ctx_fail_arg(MLF_E_STATE, "mlf_allreduce_max before mlf_comm_init / mlf_comm_init_rank");
  if (int rc = reserve_bufs(count)) return rc;
  const size_t ndev = g_comm.devices.size();
  for (size_t i = 0; i < ndev; ++i) {
    CKH(hipSetDevice(g_comm.devices[i]));
    CKH(hipMemcpyAsync(g_comm.bufs[i], values + i * count, count * sizeof(double), hipMemcpyHostToDevice, g_comm.streams[i]));
  }
  if (int rc = g_rccl.group_start()) return fail_rccl(rc, "ncclGroupStart");
  for (size_t i = 0; i < ndev; ++i)
    if (int rc = g_rccl.all_reduce(g_comm.bufs[i], g_comm.bufs[i], count, kNcclDouble, kNcclMax, g_comm.comms[i], g_comm.streams[i]))
      return fail_rccl(rc, "ncclAllReduce");
  if (int rc = g_rccl.group_end()) return fail_rccl(rc, "ncclGroupEnd");
  for (size_t i = 0; i < ndev; ++i) {
    CKH(hipSetDevice(g_comm.devices[i]));
    CKH(hipMemcpyAsync(values + i * count, g_comm.bufs[i], count * sizeof(double), hipMemcpyDeviceToHost, g_comm.streams[i]));
    CKH(hipStreamSynchronize(g_comm.streams[i]));
  }
  CKH(hipSetDevice(g_comm.devices[0]));
  return 0;
}

}  // extern "C"
