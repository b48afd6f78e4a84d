// Halo exchange of a z-slab sharded grid inside ONE process: a single RCCL all-gather.
//
// Marching-cubes cells span z-1..z (reference src/vacancy/marching_cubes.cc:93-101), so before
// extraction every slab needs the last two xy-slices of the slab below it.  The north star asks for
// "a single RCCL all-gather of boundary slabs": every device contributes the packs (vcy_halo_pack
// layout) of the slabs it holds, ncclAllGather hands every device every pack, and each slab installs
// the pack of the slab that ends at its z_begin.  One communicator rank per distinct DEVICE
// (ncclCommInitAll, one process driving all GPUs of the node); several slabs of one device share
// that device's rank.  The one-process-per-GPU form of the same exchange is vacancy_amd/dist.py
// (torch.distributed all_gather_into_tensor, which is RCCL as well).
//
// librccl.so is opened on first use (dlopen): the carve path needs no collective, and a host
// process that already carries an RCCL (PyTorch ships its own librccl.so) keeps using that one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "vcy_internal.h"

namespace vcy {

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string path;
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;

// Communicators and staging buffers are cached per device list (creating a communicator costs
// hundreds of milliseconds; an extraction per carved view would pay it every time).
struct HaloGroup {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;   // one collective stream per device
  std::vector<char*> send, recv;      // per device
  size_t send_bytes = 0, recv_bytes = 0;
};
std::vector<HaloGroup*> g_groups;

struct LastCollective {
  int ranks = 0;
  int64_t bytes_per_rank = 0;
  int64_t calls = 0;
  int version = 0;
} g_last;
thread_local std::string g_last_text;

bool load_rccl() {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) {
      g_rccl.path = n;
      break;
    }
  }
  if (!h) {
    set_error("vcy_halo_allgather: librccl.so not found (%s)", dlerror());
    return false;
  }
#define VCY_SYM(field, name)                                              \
  do {                                                                    \
    *(void**)(&g_rccl.field) = dlsym(h, name);                            \
    if (!g_rccl.field) {                                                  \
      set_error("vcy_halo_allgather: %s missing from librccl.so", name);  \
      dlclose(h);                                                         \
      return false;                                                       \
    }                                                                     \
  } while (0)
  VCY_SYM(GetVersion, "ncclGetVersion");
  VCY_SYM(CommInitAll, "ncclCommInitAll");
  VCY_SYM(CommDestroy, "ncclCommDestroy");
  VCY_SYM(AllGather, "ncclAllGather");
  VCY_SYM(GroupStart, "ncclGroupStart");
  VCY_SYM(GroupEnd, "ncclGroupEnd");
  VCY_SYM(GetErrorString, "ncclGetErrorString");
#undef VCY_SYM
  g_rccl.handle = h;
  return true;
}

#define VCY_NCCL_CHECK(expr)                                                                  \
  do {                                                                                        \
    ncclResult_t _r = (expr);                                                                 \
    if (_r != ncclSuccess) {                                                                  \
      set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
      return VCY_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

// Communicators, streams and staging of one device set (also the error path of get_group).
void destroy_group(HaloGroup* g) {
  for (size_t d = 0; d < g->devices.size(); ++d) {
    (void)hipSetDevice(g->devices[d]);
    if (d < g->streams.size() && g->streams[d]) {
      (void)hipStreamSynchronize(g->streams[d]);
      (void)hipStreamDestroy(g->streams[d]);
    }
    if (d < g->send.size() && g->send[d]) (void)hipFree(g->send[d]);
    if (d < g->recv.size() && g->recv[d]) (void)hipFree(g->recv[d]);
    if (d < g->comms.size() && g->comms[d]) (void)g_rccl.CommDestroy(g->comms[d]);
  }
  delete g;
}

int get_group(const std::vector<int>& devices, HaloGroup** out) {
  for (HaloGroup* g : g_groups)
    if (g->devices == devices) {
      *out = g;
      return VCY_OK;
    }
  HaloGroup* g = new HaloGroup;
  g->devices = devices;
  const int nd = (int)devices.size();
  g->comms.assign((size_t)nd, nullptr);
  g->streams.assign((size_t)nd, nullptr);
  g->send.assign((size_t)nd, nullptr);
  g->recv.assign((size_t)nd, nullptr);
  ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), nd, devices.data());
  if (r != ncclSuccess) {
    set_error("ncclCommInitAll(%d devices) failed: %s", nd, g_rccl.GetErrorString(r));
    delete g;
    return VCY_ERR_HIP;
  }
  for (int d = 0; d < nd; ++d) {
    hipError_t e = hipSetDevice(devices[(size_t)d]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->streams[(size_t)d], hipStreamNonBlocking);
    if (e != hipSuccess) {  // nothing half-built stays behind (a retry would initialise the communicators again)
      set_error("vcy_halo_allgather: stream on device %d: %s", devices[(size_t)d], hipGetErrorString(e));
      destroy_group(g);
      return VCY_ERR_HIP;
    }
  }
  g_groups.push_back(g);
  *out = g;
  return VCY_OK;
}

int ensure_buffers(HaloGroup* g, size_t send_bytes, size_t recv_bytes) {
  if (g->send_bytes >= send_bytes && g->recv_bytes >= recv_bytes) return VCY_OK;
  for (size_t d = 0; d < g->devices.size(); ++d) {
    VCY_HIP_CHECK(hipSetDevice(g->devices[d]));
    VCY_HIP_CHECK(hipStreamSynchronize(g->streams[d]));
    if (g->send[d]) VCY_HIP_CHECK(hipFree(g->send[d]));
    if (g->recv[d]) VCY_HIP_CHECK(hipFree(g->recv[d]));
    g->send[d] = g->recv[d] = nullptr;
    VCY_HIP_CHECK(hipMalloc(&g->send[d], send_bytes));
    VCY_HIP_CHECK(hipMalloc(&g->recv[d], recv_bytes));
  }
  g->send_bytes = send_bytes;
  g->recv_bytes = recv_bytes;
  return VCY_OK;
}

}  // namespace

}  // namespace vcy

using namespace vcy;

extern "C" {

int vcy_halo_allgather(vcy_ctx* const* slabs, int n_slabs) {
  if (!slabs || n_slabs <= 0) {
    set_error("vcy_halo_allgather: no slabs");
    return VCY_ERR_INVALID_ARG;
  }
  // the slabs must tile z in order: slab i ends where slab i + 1 begins, same xy grid and counter width
  for (int i = 0; i < n_slabs; ++i) {
    const vcy_ctx* c = slabs[i];
    if (!c) return VCY_ERR_INVALID_ARG;
    if (i > 0) {
      const vcy_ctx* p = slabs[i - 1];
      if (p->z1 != c->z0 || p->nx != c->nx || p->ny != c->ny || p->cnt_bytes_wire != c->cnt_bytes_wire) {
        set_error("vcy_halo_allgather: slab %d does not continue slab %d", i, i - 1);
        return VCY_ERR_INVALID_ARG;
      }
    }
    if (n_slabs > 1 && c->nz_local() < 2) {
      set_error("a slab needs at least 2 slices to exchange halos");
      return VCY_ERR_INVALID_ARG;
    }
  }
  if (slabs[0]->z0 != 0) {
    set_error("vcy_halo_allgather: the first slab must start at z = 0");
    return VCY_ERR_INVALID_ARG;
  }
  if (n_slabs == 1) {
    slabs[0]->halo_valid = true;  // a whole grid: nothing below
    return VCY_OK;
  }
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!load_rccl()) return VCY_ERR_UNSUPPORTED;

  // one communicator rank per distinct device, in order of first appearance
  std::vector<int> devices;
  std::vector<int> rank_of((size_t)n_slabs), slot_of((size_t)n_slabs);
  std::vector<int> held;  // slabs per rank
  for (int i = 0; i < n_slabs; ++i) {
    const int dev = slabs[i]->device;
    size_t r = 0;
    while (r < devices.size() && devices[r] != dev) ++r;
    if (r == devices.size()) {
      devices.push_back(dev);
      held.push_back(0);
    }
    rank_of[(size_t)i] = (int)r;
    slot_of[(size_t)i] = held[r]++;
  }
  const int nd = (int)devices.size();
  const int kmax = *std::max_element(held.begin(), held.end());  // equal send counts: pad to the most slabs a rank holds
  const size_t pack = (size_t)vcy_halo_bytes(slabs[0]);
  const size_t send_bytes = pack * (size_t)kmax, recv_bytes = send_bytes * (size_t)nd;
  HaloGroup* g = nullptr;
  int rc = get_group(devices, &g);
  if (rc != VCY_OK) return rc;
  rc = ensure_buffers(g, send_bytes, recv_bytes);
  if (rc != VCY_OK) return rc;

  // pack: every slab's last two slices into its rank's send buffer (applies queued views first)
  for (int i = 0; i < n_slabs; ++i) {
    rc = vcy_halo_pack(slabs[i], g->send[(size_t)rank_of[(size_t)i]] + pack * (size_t)slot_of[(size_t)i]);
    if (rc != VCY_OK) return rc;
  }
  for (int i = 0; i < n_slabs; ++i) {
    VCY_HIP_CHECK(hipSetDevice(slabs[i]->device));
    VCY_HIP_CHECK(hipStreamSynchronize(slabs[i]->stream));
  }
  // the single collective of the path
  VCY_NCCL_CHECK(g_rccl.GroupStart());
  for (int d = 0; d < nd; ++d) {
    ncclResult_t r = g_rccl.AllGather(g->send[(size_t)d], g->recv[(size_t)d], send_bytes, ncclUint8, g->comms[(size_t)d],
                                      g->streams[(size_t)d]);
    if (r != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      set_error("ncclAllGather failed: %s", g_rccl.GetErrorString(r));
      return VCY_ERR_HIP;
    }
  }
  VCY_NCCL_CHECK(g_rccl.GroupEnd());
  for (int d = 0; d < nd; ++d) {
    VCY_HIP_CHECK(hipSetDevice(devices[(size_t)d]));
    VCY_HIP_CHECK(hipStreamSynchronize(g->streams[(size_t)d]));
  }
  // install: slab i takes the pack of slab i - 1 out of its own device's gathered buffer
  for (int i = 0; i < n_slabs; ++i) {
    const char* src = nullptr;
    if (i > 0)
      src = g->recv[(size_t)rank_of[(size_t)i]] + send_bytes * (size_t)rank_of[(size_t)i - 1] +
            pack * (size_t)slot_of[(size_t)i - 1];
    rc = vcy_halo_install(slabs[i], src);
    if (rc != VCY_OK) return rc;
  }
  for (int i = 0; i < n_slabs; ++i) {  // the staging is reused by the next exchange
    VCY_HIP_CHECK(hipSetDevice(slabs[i]->device));
    VCY_HIP_CHECK(hipStreamSynchronize(slabs[i]->stream));
  }
  g_last.ranks = nd;
  g_last.bytes_per_rank = (int64_t)send_bytes;
  g_last.calls += 1;
  (void)g_rccl.GetVersion(&g_last.version);
  return VCY_OK;
}

void vcy_halo_shutdown(void) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  for (HaloGroup* g : g_groups) destroy_group(g);
  g_groups.clear();
}

const char* vcy_last_collective(void) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  char buf[256];
  if (g_last.calls == 0) {
    g_last_text = "none";
  } else {
    snprintf(buf, sizeof(buf), "backend=rccl op=ncclAllGather version=%d ranks=%d bytes_per_rank=%lld calls=%lld lib=%s",
             g_last.version, g_last.ranks, (long long)g_last.bytes_per_rank, (long long)g_last.calls,
             g_rccl.path.c_str());
    g_last_text = buf;
  }
  return g_last_text.c_str();
}

}  // extern "C"
