// orbx_comm.hip -- the one exchange step of the path behind the C ABI: gather of the fixed-capacity per-frame records
// {count, keypoints[cap], descriptors[cap][32]} of every rank's cameras onto one rank (the consumer in the
// reference would be the tracking thread that owns all cameras of a rig, src/Tracking.cc:1329-1608).
//
// One process per GPU, RCCL underneath.  xGMI is point-to-point (every peer has its own link to the root) and the payload is
// ~49 KB per camera, so the gather is written as direct sends to the root inside one ncclGroup -- no ring, no staging copy; the
// root's own slice is a device-to-device copy on the same stream.  RCCL is opened lazily with dlopen (the library a process
// already has loaded -- PyTorch bundles one -- or /opt/rocm's): liborbx.so itself carries no RCCL dependency, and a host that never
// creates a communicator never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/orbx.h"

extern "C" void orbx_internal_set_error(const char* s);
extern "C" int  orbx_internal_runtime_down(void);

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId)    getUniqueId    = nullptr;
  decltype(&ncclCommInitRank)   commInitRank   = nullptr;
  decltype(&ncclCommDestroy)    commDestroy    = nullptr;
  decltype(&ncclGroupStart)     groupStart     = nullptr;
  decltype(&ncclGroupEnd)       groupEnd       = nullptr;
  decltype(&ncclSend)           send           = nullptr;
  decltype(&ncclRecv)           recv           = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
};

Rccl       g_rccl;
std::mutex g_rcclMutex;

bool load_rccl() {
  std::lock_guard<std::mutex> lock(g_rcclMutex);
  if (g_rccl.lib) return true;
  // The copy a process already has mapped comes first (PyTorch bundles one under its soname librccl.so.1: a second RCCL beside it would
  // bring a second set of communicator state), then the loader path, then /opt/rocm.  ORBX_RCCL_LIBRARY names the one library to try
  // instead (deployments with a private RCCL; tests/test_abi_and_hostmodel.py uses it to reach the failure path).
  const char* forced  = getenv("ORBX_RCCL_LIBRARY");
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void*       lib     = nullptr;
  std::string why;
  if (forced && *forced) {
    lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
      const char* e = dlerror();  // one call: dlerror() clears the message it returns
      why = e ? e : "";
    }
  } else {
    for (const char* n : names) {
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (lib) break;
    }
    for (const char* n : names) {
      if (lib) break;
      lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) {
        const char* e = dlerror();
        if (e) why = e;
      }
    }
  }
  if (!lib) {
    orbx_internal_set_error((std::string("orbx_comm: cannot open librccl.so: ") + why).c_str());
    return false;
  }
  Rccl r;
  r.lib = lib;
#define ORBX_SYM(field, name)                                            \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(lib, name));       \
  if (!r.field) {                                                        \
    orbx_internal_set_error("orbx_comm: librccl.so lacks " name);        \
    dlclose(lib);                                                        \
    return false;                                                        \
  }
  ORBX_SYM(getUniqueId, "ncclGetUniqueId")
  ORBX_SYM(commInitRank, "ncclCommInitRank")
  ORBX_SYM(commDestroy, "ncclCommDestroy")
  ORBX_SYM(groupStart, "ncclGroupStart")
  ORBX_SYM(groupEnd, "ncclGroupEnd")
  ORBX_SYM(send, "ncclSend")
  ORBX_SYM(recv, "ncclRecv")
  ORBX_SYM(getErrorString, "ncclGetErrorString")
#undef ORBX_SYM
  g_rccl = r;
  return true;
}

int fail(const char* what, ncclResult_t rc) {
  orbx_internal_set_error((std::string(what) + ": " + (g_rccl.getErrorString ? g_rccl.getErrorString(rc) : "RCCL error")).c_str());
  return -3;
}

}  // namespace

struct orbx_comm {
  ncclComm_t comm   = nullptr;
  int        device = 0, rank = 0, nranks = 1;
};

static_assert(ORBX_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "orbx.h and rccl.h disagree on the size of the rendezvous token");

extern "C" {

int orbx_comm_unique_id(void* id_out) {
  if (!id_out) {
    orbx_internal_set_error("orbx_comm_unique_id: null output");
    return -2;
  }
  if (!load_rccl()) return -3;
  ncclUniqueId id;
  ncclResult_t rc = g_rccl.getUniqueId(&id);
  if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

int orbx_comm_create(orbx_comm_t** out, int device_id, int rank, int nranks, const void* unique_id) {
  if (!out || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) {
    orbx_internal_set_error("orbx_comm_create: bad arguments");
    return -2;
  }
  *out = nullptr;
  if (!load_rccl()) return -3;
  if (hipSetDevice(device_id) != hipSuccess) {
    orbx_internal_set_error("orbx_comm_create: hipSetDevice failed");
    return -3;
  }
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  orbx_comm* c = new orbx_comm;
  c->device = device_id, c->rank = rank, c->nranks = nranks;
  ncclResult_t rc = g_rccl.commInitRank(&c->comm, nranks, id, rank);
  if (rc != ncclSuccess) {
    delete c;
    return fail("ncclCommInitRank", rc);
  }
  *out = c;
  return 0;
}

void orbx_comm_destroy(orbx_comm_t* c) {
  if (!c) return;
  if (c->comm && g_rccl.commDestroy && !orbx_internal_runtime_down()) {
    hipSetDevice(c->device);
    g_rccl.commDestroy(c->comm);
  }
  delete c;
}

int orbx_comm_rank(const orbx_comm_t* c) { return c ? c->rank : -1; }
int orbx_comm_size(const orbx_comm_t* c) { return c ? c->nranks : -1; }

int orbx_gather(orbx_comm_t* c, const orbx_keypoint_t* d_kps, const uint8_t* d_desc, const int* d_counts, int nframes, int cap,
                orbx_keypoint_t* d_all_kps, uint8_t* d_all_desc, int* d_all_counts, int root, void* stream) {
  if (!c || !d_kps || !d_desc || !d_counts || nframes < 1 || cap < 1 || root < 0 || root >= c->nranks) {
    orbx_internal_set_error("orbx_gather: bad arguments");
    return -2;
  }
  const bool isRoot = c->rank == root;
  if (isRoot && (!d_all_kps || !d_all_desc || !d_all_counts)) {
    orbx_internal_set_error("orbx_gather: the root needs the three output arrays");
    return -2;
  }
  hipStream_t  s  = static_cast<hipStream_t>(stream);
  const size_t kb = (size_t)nframes * cap * sizeof(orbx_keypoint_t), db = (size_t)nframes * cap * 32, cb = (size_t)nframes * 2 * sizeof(int);
  if (hipSetDevice(c->device) != hipSuccess) {
    orbx_internal_set_error("orbx_gather: hipSetDevice failed");
    return -3;
  }
  if (isRoot) {  // the root's own slice
    uint8_t *ak = reinterpret_cast<uint8_t*>(d_all_kps), *ac = reinterpret_cast<uint8_t*>(d_all_counts);
    if (hipMemcpyAsync(ak + kb * root, d_kps, kb, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_all_desc + db * root, d_desc, db, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(ac + cb * root, d_counts, cb, hipMemcpyDeviceToDevice, s) != hipSuccess) {
      orbx_internal_set_error("orbx_gather: device copy of the root's slice failed");
      return -3;
    }
  }
  if (c->nranks == 1) return 0;
  ncclResult_t rc = g_rccl.groupStart();
  if (rc != ncclSuccess) return fail("ncclGroupStart", rc);
  if (isRoot) {
    uint8_t *ak = reinterpret_cast<uint8_t*>(d_all_kps), *ac = reinterpret_cast<uint8_t*>(d_all_counts);
    for (int p = 0; p < c->nranks && rc == ncclSuccess; p++) {
      if (p == root) continue;
      rc = g_rccl.recv(ak + kb * p, kb, ncclUint8, p, c->comm, s);
      if (rc == ncclSuccess) rc = g_rccl.recv(d_all_desc + db * p, db, ncclUint8, p, c->comm, s);
      if (rc == ncclSuccess) rc = g_rccl.recv(ac + cb * p, cb, ncclUint8, p, c->comm, s);
    }
  } else {
    rc = g_rccl.send(d_kps, kb, ncclUint8, root, c->comm, s);
    if (rc == ncclSuccess) rc = g_rccl.send(d_desc, db, ncclUint8, root, c->comm, s);
    if (rc == ncclSuccess) rc = g_rccl.send(d_counts, cb, ncclUint8, root, c->comm, s);
  }
  const ncclResult_t rc2 = g_rccl.groupEnd();
  if (rc != ncclSuccess) return fail("ncclSend / ncclRecv", rc);
  if (rc2 != ncclSuccess) return fail("ncclGroupEnd", rc2);
  return 0;
}

}  // extern "C"
