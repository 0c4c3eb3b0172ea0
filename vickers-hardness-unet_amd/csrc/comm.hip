// Data-parallel collectives behind the C ABI (SURVEY.md 8(b) "DP": vk_comm_init / all-reduce of a gradient bucket; 8(e)): thin
// entry points over RCCL (ncclAllReduce / ncclBroadcast over xGMI) for a host that is not PyTorch.  The Python host of this package
// uses torch.distributed (backend "nccl" IS RCCL) and never calls these; they exist so that the DP path of the engine — staged
// backward (vk_unet_backward), bucket table (vk_unet_bucket_range), averaging folded into vk_adamw_step — is complete below the
// C boundary as well.  librccl.so is opened on first use (dlopen): the library itself has no link-time dependency on it, so hosts
// that never touch vk_comm_* do not need RCCL installed.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "vk_common.h"

namespace {

// the handful of RCCL declarations used, restated from /opt/rocm/include/rccl/rccl.h (ncclUniqueId: 128 opaque bytes; ncclFloat32 = 7,
// ncclUint8 = 1, ncclSum = 0) so that this file compiles without the RCCL headers
struct NcclId { char internal[VK_COMM_ID_BYTES]; };
typedef void* NcclComm;
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(NcclComm*, int, NcclId, int);
typedef int (*fn_destroy)(NcclComm);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
  void* so = nullptr;
  fn_get_id get_id = nullptr;
  fn_init_rank init_rank = nullptr;
  fn_destroy destroy = nullptr;
  fn_allreduce allreduce = nullptr;
  fn_broadcast broadcast = nullptr;
  fn_errstr errstr = nullptr;
  bool ok = false;
  char why[256] = {0};            // why the library is unusable (captured once, inside call_once)
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
      if (const char* e = dlerror()) snprintf(r.why, sizeof(r.why), "%s", e);      // dlerror() clears itself: keep the text of the last attempt
    }
    if (!r.so) return;
    r.get_id = (fn_get_id)dlsym(r.so, "ncclGetUniqueId");
    r.init_rank = (fn_init_rank)dlsym(r.so, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(r.so, "ncclCommDestroy");
    r.allreduce = (fn_allreduce)dlsym(r.so, "ncclAllReduce");
    r.broadcast = (fn_broadcast)dlsym(r.so, "ncclBroadcast");
    r.errstr = (fn_errstr)dlsym(r.so, "ncclGetErrorString");
    r.ok = r.get_id && r.init_rank && r.destroy && r.allreduce && r.broadcast;
    if (!r.ok) snprintf(r.why, sizeof(r.why), "a required ncclXxx symbol is missing");
  });
  return r;
}

int need_rccl(const char* who) {
  if (rccl().ok) return VK_OK;
  vkh::set_error("%s: librccl.so is not usable (%s)", who, rccl().why[0] ? rccl().why : "unknown reason");
  return VK_ERR_STATE;
}

int nccl_fail(const char* what, int rc) {
  vkh::set_error("%s failed: %s (ncclResult %d)", what, rccl().errstr ? rccl().errstr(rc) : "?", rc);
  return VK_ERR_STATE;
}

}  // namespace

struct vk_comm {
  NcclComm comm;
  int rank, world;
};

extern "C" int vk_comm_unique_id(void* id_out) {
  VK_CHECK_ARG(id_out != nullptr, "vk_comm_unique_id: null argument");
  if (int rc = need_rccl("vk_comm_unique_id")) return rc;
  NcclId id;
  const int rc = rccl().get_id(&id);
  if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(id_out, &id, VK_COMM_ID_BYTES);
  return VK_OK;
}

extern "C" int vk_comm_init(int rank, int world, const void* id, vk_comm** out) {
  VK_CHECK_ARG(out && id && world >= 1 && rank >= 0 && rank < world, "vk_comm_init: bad argument (rank %d of %d)", rank, world);
  if (int rc = need_rccl("vk_comm_init")) return rc;
  NcclId nid;
  memcpy(&nid, id, VK_COMM_ID_BYTES);
  NcclComm c = nullptr;
  const int rc = rccl().init_rank(&c, world, nid, rank);
  if (rc != 0) return nccl_fail("ncclCommInitRank", rc);
  vk_comm* h = new vk_comm();
  h->comm = c; h->rank = rank; h->world = world;
  *out = h;
  return VK_OK;
}

extern "C" int vk_comm_destroy(vk_comm* c) {
  if (!c) return VK_OK;
  const int rc = rccl().ok ? rccl().destroy(c->comm) : 0;
  delete c;
  return rc == 0 ? VK_OK : nccl_fail("ncclCommDestroy", rc);
}

extern "C" int vk_comm_world(const vk_comm* c) { return c ? c->world : 0; }

extern "C" int vk_allreduce_bucket(vk_comm* c, float* grads, size_t count, void* stream) {
  VK_CHECK_ARG(c && grads && count > 0, "vk_allreduce_bucket: null argument");
  if (int rc0 = need_rccl("vk_allreduce_bucket")) return rc0;
  const int rc = rccl().allreduce(grads, grads, count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->comm, (hipStream_t)stream);
  return rc == 0 ? VK_OK : nccl_fail("ncclAllReduce", rc);
}

extern "C" int vk_comm_broadcast(vk_comm* c, void* buf, size_t bytes, int root, void* stream) {
  VK_CHECK_ARG(c && buf && bytes > 0 && root >= 0 && root < c->world, "vk_comm_broadcast: bad argument");
  if (int rc0 = need_rccl("vk_comm_broadcast")) return rc0;
  const int rc = rccl().broadcast(buf, buf, bytes, /*ncclUint8*/ 1, root, c->comm, (hipStream_t)stream);
  return rc == 0 ? VK_OK : nccl_fail("ncclBroadcast", rc);
}
