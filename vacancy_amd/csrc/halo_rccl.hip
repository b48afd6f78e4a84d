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
#include <arpa/inet.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vcy_internal.h"

namespace vcy {

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;                        // (process-per-GPU form, vcy_comm_create)
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // (optional: the error path of the sharded producer)
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
  VCY_SYM(GetUniqueId, "ncclGetUniqueId");
  VCY_SYM(CommInitRank, "ncclCommInitRank");
  VCY_SYM(CommDestroy, "ncclCommDestroy");
  VCY_SYM(AllGather, "ncclAllGather");
  VCY_SYM(GroupStart, "ncclGroupStart");
  VCY_SYM(GroupEnd, "ncclGroupEnd");
  VCY_SYM(GetErrorString, "ncclGetErrorString");
#undef VCY_SYM
  *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
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

}  // extern "C"

/* ---- sharded silhouette producer ---------------------------------------------------------------------------------
 * Carve(vector<Camera>, vector<Image1b>) (reference voxel_carver.cc:516-528 around :394-413) over the z-slabs of ONE
 * grid held by this process.  Round 4 handed every slab context the whole list (vcy_carve_batch_silhouettes per slab):
 * each GPU uploaded every silhouette and built every SDF, and at 8 GPUs the producer (1.6 ms per 32 views at 1280 x 720)
 * was longer than a rank's carve (1.0 ms).  Here the devices SHARE the producer: device r of R uploads and transforms the
 * views r, r + R, ... of every chunk of 32, ONE ncclAllGather per chunk hands every device all the images (W * H * 4
 * bytes each), and every slab carves the chunk from its device's copy -- while the next chunk is produced and gathered
 * on the producer streams.  Slabs that share a device share its images (round 4 built them once per slab).
 * One host thread per producer rank, like ShardedVoxelCarver's thread per slab; results are bit-identical to the
 * per-slab form (same images, same fused launches).                                                                  */
namespace vcy {
namespace {

struct ProducerRank {
  int device = 0;
  hipStream_t aux = nullptr;
  char* pool = nullptr;      // masks [2][per] | scratch [per] | send [per]
  size_t pool_bytes = 0;
  char* recv[2] = {nullptr, nullptr};
  size_t recv_bytes = 0;
  void* pinned = nullptr;
  size_t pinned_bytes = 0;
  hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_uploaded[2] = {nullptr, nullptr}, ev_sent = nullptr;
  std::vector<hipEvent_t> ev_consumed[2];  // per slab of this rank
};
struct ProducerGroup {
  std::vector<int> devices;  // per rank (distinct unless the test hook splits a device)
  std::vector<ProducerRank> ranks;
};
std::vector<ProducerGroup*> g_producers;

void destroy_producer(ProducerGroup* g) {
  for (ProducerRank& r : g->ranks) {
    (void)hipSetDevice(r.device);
    if (r.aux) {
      (void)hipStreamSynchronize(r.aux);
      (void)hipStreamDestroy(r.aux);
    }
    (void)hipFree(r.pool);
    (void)hipFree(r.recv[0]);
    (void)hipFree(r.recv[1]);
    if (r.pinned) (void)hipHostFree(r.pinned);
    for (int k = 0; k < 2; ++k) {
      if (r.ev_ready[k]) (void)hipEventDestroy(r.ev_ready[k]);
      if (r.ev_uploaded[k]) (void)hipEventDestroy(r.ev_uploaded[k]);
      for (hipEvent_t e : r.ev_consumed[k]) (void)hipEventDestroy(e);
    }
    if (r.ev_sent) (void)hipEventDestroy(r.ev_sent);
  }
  delete g;
}

int get_producer(const std::vector<int>& devices, ProducerGroup** out) {
  for (ProducerGroup* g : g_producers)
    if (g->devices == devices) {
      *out = g;
      return VCY_OK;
    }
  ProducerGroup* g = new ProducerGroup;
  g->devices = devices;
  g->ranks.resize(devices.size());
  for (size_t r = 0; r < devices.size(); ++r) {
    ProducerRank& pr = g->ranks[r];
    pr.device = devices[r];
    hipError_t e = hipSetDevice(pr.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pr.aux, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
      e = hipEventCreateWithFlags(&pr.ev_ready[k], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&pr.ev_uploaded[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pr.ev_sent, hipEventDisableTiming);
    if (e != hipSuccess) {
      set_error("sharded producer: device %d: %s", pr.device, hipGetErrorString(e));
      destroy_producer(g);
      return VCY_ERR_HIP;
    }
  }
  g_producers.push_back(g);
  *out = g;
  return VCY_OK;
}

// all threads of one call meet here (test hook path and error hand-over)
struct HostBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0, phase = 0;
  explicit HostBarrier(int n_) : n(n_) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const int ph = phase;
    if (++waiting == n) {
      waiting = 0;
      ++phase;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != ph; });
    }
  }
};

}  // namespace
}  // namespace vcy

using namespace vcy;

extern "C" {

int vcy_carve_batch_silhouettes_sharded(vcy_ctx* const* slabs, int n_slabs, int n_views, const vcy_view* views,
                                        const uint8_t* const* masks_host) {
  if (!slabs || n_slabs <= 0 || n_views <= 0 || !views || !masks_host) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  size_t max_px = 0;
  for (int i = 0; i < n_views; ++i) {
    if (!masks_host[i]) {
      set_error("null silhouette");
      return VCY_ERR_INVALID_ARG;
    }
    max_px = std::max(max_px, (size_t)views[i].width * views[i].height);
  }
  for (int s = 0; s < n_slabs; ++s) {
    if (!slabs[s]) {
      set_error("VoxelCarver::Carve voxel grid has not been initialized");
      return VCY_ERR_NOT_INITIALIZED;
    }
    const int rc = check_carve_views(slabs[s], n_views, views);
    if (rc != VCY_OK) return rc;
    const vcy_carver_option &a = slabs[0]->opt, &b = slabs[s]->opt;
    if (a.sdf_minmax_normalize != b.sdf_minmax_normalize || a.update_option.use_truncation != b.update_option.use_truncation ||
        a.update_option.truncation_band != b.update_option.truncation_band) {
      set_error("vcy_carve_batch_silhouettes_sharded: the slabs do not share one option set");
      return VCY_ERR_INVALID_ARG;
    }
  }
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  // producer ranks: one per distinct device, in order of first appearance.  Test hook VCY_TEST_SPLIT_PRODUCERS=1: one
  // rank per SLAB even on a shared device, the all-gather then being device copies -- the share / slot / gather
  // layout of an R-device run, exercised on one GPU.
  const char* split_env = std::getenv("VCY_TEST_SPLIT_PRODUCERS");
  const bool split = split_env && split_env[0] == '1';
  std::vector<int> devices, rank_of((size_t)n_slabs);
  for (int s = 0; s < n_slabs; ++s) {
    size_t r = 0;
    if (split) r = devices.size();
    else while (r < devices.size() && devices[r] != slabs[s]->device) ++r;
    if (r == devices.size()) devices.push_back(slabs[s]->device);
    rank_of[(size_t)s] = (int)r;
  }
  const int R = (int)devices.size();
  bool distinct = true;
  for (int a = 0; a < R; ++a)
    for (int b = a + 1; b < R; ++b) distinct = distinct && devices[(size_t)a] != devices[(size_t)b];
  HaloGroup* comm = nullptr;
  if (R > 1 && distinct) {
    if (!load_rccl()) return VCY_ERR_UNSUPPORTED;
    const int rc = get_group(devices, &comm);
    if (rc != VCY_OK) return rc;
  }
  ProducerGroup* pg = nullptr;
  {
    const int rc = get_producer(devices, &pg);
    if (rc != VCY_OK) return rc;
  }
  const int chunk = 32;
  const int n_chunks = (n_views + chunk - 1) / chunk;
  const int per = (std::min(chunk, n_views) + R - 1) / R;  // images a rank produces per chunk
  const size_t stride = (max_px * sizeof(float) + 255) / 256 * 256, sz_mask = (max_px + 255) / 256 * 256;
  const size_t sz_scr = (device_make_sdf_scratch_bytes(1, (int)max_px) + 255) / 256 * 256;
  const size_t send_bytes = (size_t)per * stride;
  const vcy_update_option& u = slabs[0]->opt.update_option;
  const bool normalize = slabs[0]->opt.sdf_minmax_normalize != 0;

  std::atomic<int> failed(VCY_OK);
  std::mutex err_mutex;
  std::string err_text;
  auto fail = [&](int code, const std::string& text) {
    std::lock_guard<std::mutex> lk(err_mutex);
    if (failed.load() == VCY_OK) {
      failed.store(code);
      err_text = text;
    }
  };
  HostBarrier barrier(R);
  const auto t_entry = std::chrono::steady_clock::now();
  // A collective that failed on one rank after others had enqueued it: ncclCommAbort on every communicator of the group
  // ends the stranded kernels; the group (and the producer with its streams) is dropped after the threads have joined
  // and rebuilt by the next call.
  std::atomic<bool> aborted(false);
  std::mutex abort_mutex;
  auto abort_group = [&]() {
    std::lock_guard<std::mutex> lk(abort_mutex);
    if (aborted.load() || !comm) return;
    aborted.store(true);
    for (ncclComm_t& cm : comm->comms) {
      if (cm && g_rccl.CommAbort) (void)g_rccl.CommAbort(cm);
      if (g_rccl.CommAbort) cm = nullptr;  // (aborted communicators are gone: destroy_group must not touch them)
    }
  };

  auto worker = [&](int r) {
    ProducerRank& pr = pg->ranks[(size_t)r];
    std::vector<vcy_ctx*> mine;
    for (int s = 0; s < n_slabs; ++s)
      if (rank_of[(size_t)s] == r) mine.push_back(slabs[s]);
    auto hip_ok = [&](hipError_t e, const char* what) {
      if (e != hipSuccess) fail(VCY_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
      return e == hipSuccess;
    };
    hip_ok(hipSetDevice(pr.device), "hipSetDevice");
    // buffers of this rank, grown on demand
    const size_t pool_need = 2 * (size_t)per * sz_mask + (size_t)per * sz_scr + (R > 1 ? send_bytes : 0) + 256;
    const size_t recv_need = send_bytes * (size_t)R;
    if (failed.load() == VCY_OK && pr.pool_bytes < pool_need) {
      (void)hipStreamSynchronize(pr.aux);
      (void)hipFree(pr.pool);
      pr.pool = nullptr, pr.pool_bytes = 0;
      if (hip_ok(hipMalloc((void**)&pr.pool, pool_need), "hipMalloc")) pr.pool_bytes = pool_need;
    }
    if (failed.load() == VCY_OK && pr.recv_bytes < recv_need) {
      for (vcy_ctx* c : mine) (void)hipStreamSynchronize(c->stream);
      for (int k = 0; k < 2; ++k) {
        (void)hipFree(pr.recv[k]);
        pr.recv[k] = nullptr;
      }
      pr.recv_bytes = 0;
      if (hip_ok(hipMalloc((void**)&pr.recv[0], recv_need), "hipMalloc") && hip_ok(hipMalloc((void**)&pr.recv[1], recv_need), "hipMalloc"))
        pr.recv_bytes = recv_need;
    }
    if (failed.load() == VCY_OK && pr.pinned_bytes < 2 * (size_t)per * sz_mask) {
      (void)hipStreamSynchronize(pr.aux);
      if (pr.pinned) (void)hipHostFree(pr.pinned);
      pr.pinned = nullptr, pr.pinned_bytes = 0;
      if (hip_ok(hipHostMalloc(&pr.pinned, 2 * (size_t)per * sz_mask, hipHostMallocDefault), "hipHostMalloc"))
        pr.pinned_bytes = 2 * (size_t)per * sz_mask;
    }
    for (int k = 0; k < 2; ++k)
      while (failed.load() == VCY_OK && pr.ev_consumed[k].size() < mine.size()) {
        hipEvent_t ev = nullptr;
        if (!hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) break;
        pr.ev_consumed[k].push_back(ev);
      }
    for (vcy_ctx* c : mine) {  // the timing record of vcy_last_stream_ms, per slab
      while (failed.load() == VCY_OK && (int)c->stream_events.size() < 4 * n_chunks) {
        hipEvent_t ev = nullptr;
        if (!hip_ok(hipEventCreate(&ev), "hipEventCreate")) break;
        c->stream_events.push_back(ev);
      }
      c->stream_timed_chunks = 0;
    }
    barrier.wait();  // every rank has its buffers (or the call has failed) before anything is enqueued
    char* mask_base = pr.pool;
    char* scratch = pr.pool + 2 * (size_t)per * sz_mask;
    char* send = scratch + (size_t)per * sz_scr;

    auto produce = [&](int ci) {
      const int set = ci & 1, first = ci * chunk, m = std::min(chunk, n_views - first);
      const bool live = failed.load() == VCY_OK;
      if (live && ci >= 2) {
        for (size_t s = 0; s < mine.size(); ++s) hip_ok(hipStreamWaitEvent(pr.aux, pr.ev_consumed[set][s], 0), "hipStreamWaitEvent");
        hip_ok(hipEventSynchronize(pr.ev_uploaded[set]), "hipEventSynchronize");
      }
      std::vector<const uint8_t*> mptr;
      std::vector<float*> optr;
      std::vector<vcy_view> myviews;
      if (live) {
        for (vcy_ctx* c : mine) hip_ok(hipEventRecord(c->stream_events[(size_t)4 * ci + 0], pr.aux), "hipEventRecord");
        for (int j = r, k = 0; j < m; j += R, ++k) {  // this rank's share of the chunk: views r, r + R, ...
          const vcy_view& v = views[first + j];
          const size_t npx = (size_t)v.width * v.height;
          char* stage = (char*)pr.pinned + ((size_t)set * per + k) * sz_mask;
          char* dmask = mask_base + ((size_t)set * per + k) * sz_mask;
          std::memcpy(stage, masks_host[first + j], npx);
          hip_ok(hipMemcpyAsync(dmask, stage, npx, hipMemcpyHostToDevice, pr.aux), "mask upload");
          mptr.push_back((const uint8_t*)dmask);
          optr.push_back((float*)((R > 1 ? send : pr.recv[set]) + (size_t)k * stride));
          myviews.push_back(v);
        }
        hip_ok(hipEventRecord(pr.ev_uploaded[set], pr.aux), "hipEventRecord");
        if (!mptr.empty() && failed.load() == VCY_OK) {
          // MakeSignedDistanceField(...) of voxel_carver.cc:405-408 for this rank's share
          const int rc = device_make_sdf_batch(pr.aux, (int)mptr.size(), mptr.data(), myviews.data(), normalize,
                                               u.use_truncation != 0, u.truncation_band, scratch, sz_scr, optr.data());
          if (rc != VCY_OK) fail(rc, vcy_last_error());
        }
      }
      if (R > 1) {
        if (comm) {  // the exchange: ONE all-gather per chunk, every device receives every rank's images
          // Whether this chunk is gathered is decided JOINTLY: a rank that skipped the collective (its producer, a HIP
          // call or its carve failed) while the others had enqueued theirs would leave them waiting for ever in their
          // closing hipStreamSynchronize, with g_rccl_mutex held.  Every rank has finished what can fail before the
          // first barrier; between the two barriers nobody writes `failed`, so every rank reads the same value.
          barrier.wait();
          const bool gather = failed.load() == VCY_OK;
          barrier.wait();
          if (gather) {
            const ncclResult_t nr = g_rccl.AllGather(send, pr.recv[set], send_bytes, ncclUint8, comm->comms[(size_t)r], pr.aux);
            if (nr != ncclSuccess) {
              // (some ranks may already have enqueued theirs: only aborting the communicators gets them out)
              fail(VCY_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(nr));
              abort_group();
            }
          }
        } else {  // (test hook: several ranks on one device -- the same data movement as device copies)
          if (failed.load() == VCY_OK) hip_ok(hipEventRecord(pr.ev_sent, pr.aux), "hipEventRecord");
          barrier.wait();
          for (int q = 0; q < R && failed.load() == VCY_OK; ++q) {
            hip_ok(hipStreamWaitEvent(pr.aux, pg->ranks[(size_t)q].ev_sent, 0), "hipStreamWaitEvent");
            const char* src = pg->ranks[(size_t)q].pool + 2 * (size_t)per * sz_mask + (size_t)per * sz_scr;
            hip_ok(hipMemcpyAsync(pr.recv[set] + (size_t)q * send_bytes, src, send_bytes, hipMemcpyDeviceToDevice, pr.aux), "gather copy");
          }
          (void)hipStreamSynchronize(pr.aux);
          barrier.wait();  // nobody overwrites its send buffer before every rank has copied it
        }
      }
      if (failed.load() == VCY_OK) {
        for (vcy_ctx* c : mine) hip_ok(hipEventRecord(c->stream_events[(size_t)4 * ci + 1], pr.aux), "hipEventRecord");
        hip_ok(hipEventRecord(pr.ev_ready[set], pr.aux), "hipEventRecord");
      }
    };

    produce(0);
    for (int ci = 0; ci < n_chunks; ++ci) {
      const int set = ci & 1, first = ci * chunk, m = std::min(chunk, n_views - first);
      if (ci + 1 < n_chunks) produce(ci + 1);  // the next chunk is produced and gathered while this one is carved
      if (failed.load() != VCY_OK) continue;   // (keep meeting the others at the barriers of produce)
      std::vector<const float*> ptrs((size_t)m);
      for (int j = 0; j < m; ++j) ptrs[(size_t)j] = (const float*)(pr.recv[set] + ((size_t)(j % R) * per + (size_t)(j / R)) * stride);
      for (size_t s = 0; s < mine.size() && failed.load() == VCY_OK; ++s) {
        vcy_ctx* c = mine[s];
        hip_ok(hipStreamWaitEvent(c->stream, pr.ev_ready[set], 0), "hipStreamWaitEvent");
        hip_ok(hipEventRecord(c->stream_events[(size_t)4 * ci + 2], c->stream), "hipEventRecord");
        const int rc = launch_carve(c, m, views + first, ptrs.data());
        if (rc != VCY_OK) fail(rc, vcy_last_error());
        hip_ok(hipEventRecord(c->stream_events[(size_t)4 * ci + 3], c->stream), "hipEventRecord");
        hip_ok(hipEventRecord(pr.ev_consumed[set][s], c->stream), "hipEventRecord");
        if (failed.load() == VCY_OK) c->stream_timed_chunks = ci + 1;
      }
    }
    for (vcy_ctx* c : mine) (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(pr.aux);
  };

  std::vector<std::thread> threads;
  for (int r = 1; r < R; ++r) threads.emplace_back(worker, r);
  worker(0);
  for (std::thread& t : threads) t.join();
  const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
  for (int s = 0; s < n_slabs; ++s) slabs[s]->stream_wall_ms = wall;
  if (aborted.load()) {
    g_groups.erase(std::remove(g_groups.begin(), g_groups.end(), comm), g_groups.end());
    destroy_group(comm);
    g_producers.erase(std::remove(g_producers.begin(), g_producers.end(), pg), g_producers.end());
    destroy_producer(pg);
  }
  (void)hipSetDevice(slabs[0]->device);
  if (failed.load() != VCY_OK) {
    set_error("%s", err_text.c_str());
    return failed.load();
  }
  return VCY_OK;
}

void vcy_halo_shutdown(void) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  for (ProducerGroup* g : g_producers) destroy_producer(g);
  g_producers.clear();
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


/* ---- one process per GPU, without torch ---------------------------------------------------------------------------
 * The north star keeps the host in C++; its multi-GPU form is "one process per GPU ... a single RCCL all-gather of
 * boundary slabs".  vacancy_amd/dist.py does that exchange through torch.distributed; a C++ host has no torch.  Here
 * is the same exchange for it: every process creates a vcy_comm (rank r of `world`, its device), the ncclUniqueId of
 * rank 0 reaches the others through a RENDEZVOUS that needs nothing but the filesystem or a TCP port of the node
 * ("file:<path>" or "tcp:<host>:<port>"), ncclCommInitRank builds the communicator, and vcy_halo_allgather_ranks is
 * the one collective: this rank's slabs (slab ids rank, rank + world, ...) pack their last two slices, ONE
 * ncclAllGather hands every rank every pack (rank-major, as vacancy_amd.dist.exchange_halo lays them out), every slab
 * installs the pack of the slab below it.  No reference counterpart (the reference is single-process OpenMP).         */
namespace vcy {
namespace {

constexpr size_t kRendezvousBytes = 128;  // sizeof(ncclUniqueId)
static_assert(sizeof(ncclUniqueId) == kRendezvousBytes, "rendezvous payload = one ncclUniqueId");

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool write_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n > 0) {
    const ssize_t w = ::write(fd, p, n);
    if (w <= 0) return false;
    p += w;
    n -= (size_t)w;
  }
  return true;
}
bool read_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n > 0) {
    const ssize_t r = ::read(fd, p, n);
    if (r <= 0) return false;
    p += r;
    n -= (size_t)r;
  }
  return true;
}

// file:<path> -- rank 0 writes <path>.tmp and renames it to <path> (atomic: a reader sees all 128 bytes or no file);
// the others poll for it.  Rank 0 removes a stale file of an earlier job before it writes; the path must be private to
// the job (bench-style: include the job's port or pid).
int rendezvous_file(const std::string& path, int rank, void* payload, int timeout_ms) {
  if (rank == 0) {
    (void)::unlink(path.c_str());
    const std::string tmp = path + ".tmp";
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) {
      set_error("rendezvous: cannot create %s", tmp.c_str());
      return VCY_ERR_INVALID_ARG;
    }
    const bool ok = write_all(fd, payload, kRendezvousBytes);
    ::close(fd);
    if (!ok || ::rename(tmp.c_str(), path.c_str()) != 0) {
      set_error("rendezvous: cannot publish %s", path.c_str());
      return VCY_ERR_INTERNAL;
    }
    return VCY_OK;
  }
  const double t_end = now_ms() + timeout_ms;
  while (now_ms() < t_end) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd >= 0) {
      struct stat st;
      const bool ok = ::fstat(fd, &st) == 0 && (size_t)st.st_size == kRendezvousBytes && read_all(fd, payload, kRendezvousBytes);
      ::close(fd);
      if (ok) return VCY_OK;
    }
    ::usleep(2000);
  }
  set_error("rendezvous: %s did not appear within %d ms", path.c_str(), timeout_ms);
  return VCY_ERR_INTERNAL;
}

// tcp:<host>:<port> -- rank 0 listens on the port and sends the payload to world - 1 connections; the others connect
// (retrying while rank 0 is not listening yet) and read it.
int rendezvous_tcp(const std::string& host, int port, int rank, int world, void* payload, int timeout_ms) {
  const double t_end = now_ms() + timeout_ms;
  if (rank == 0) {
    const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) {
      set_error("rendezvous: socket() failed");
      return VCY_ERR_INTERNAL;
    }
    int one = 1;
    (void)::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
    if (::bind(ls, (sockaddr*)&addr, sizeof(addr)) != 0 || ::listen(ls, world) != 0) {
      ::close(ls);
      set_error("rendezvous: cannot listen on port %d", port);
      return VCY_ERR_INVALID_ARG;
    }
    int rc = VCY_OK;
    for (int k = 1; k < world && rc == VCY_OK; ++k) {
      timeval tv;
      const double left = std::max(1.0, t_end - now_ms());
      tv.tv_sec = (long)(left / 1000.0);
      tv.tv_usec = (long)((left - 1000.0 * tv.tv_sec) * 1000.0);
      fd_set fds;
      FD_ZERO(&fds);
      FD_SET(ls, &fds);
      if (::select(ls + 1, &fds, nullptr, nullptr, &tv) <= 0) {
        set_error("rendezvous: %d of %d ranks connected within %d ms", k - 1, world - 1, timeout_ms);
        rc = VCY_ERR_INTERNAL;
        break;
      }
      const int cs = ::accept(ls, nullptr, nullptr);
      if (cs < 0 || !write_all(cs, payload, kRendezvousBytes)) {
        set_error("rendezvous: sending the id failed");
        rc = VCY_ERR_INTERNAL;
      }
      if (cs >= 0) ::close(cs);
    }
    ::close(ls);
    return rc;
  }
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  const std::string port_s = std::to_string(port);
  if (::getaddrinfo(host.c_str(), port_s.c_str(), &hints, &res) != 0 || !res) {
    set_error("rendezvous: cannot resolve %s", host.c_str());
    return VCY_ERR_INVALID_ARG;
  }
  int rc = VCY_ERR_INTERNAL;
  while (now_ms() < t_end) {
    const int cs = ::socket(AF_INET, SOCK_STREAM, 0);
    if (cs < 0) break;
    if (::connect(cs, res->ai_addr, res->ai_addrlen) == 0) {
      const bool ok = read_all(cs, payload, kRendezvousBytes);
      ::close(cs);
      if (ok) {
        rc = VCY_OK;
        break;
      }
    } else {
      ::close(cs);
    }
    ::usleep(5000);
  }
  ::freeaddrinfo(res);
  if (rc != VCY_OK) set_error("rendezvous: no id from %s:%d within %d ms", host.c_str(), port, timeout_ms);
  return rc;
}

int rendezvous(const char* where, int rank, int world, void* payload, int timeout_ms) {
  if (!where || rank < 0 || world < 1 || rank >= world || !payload || timeout_ms <= 0) {
    set_error("rendezvous: invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  if (world == 1) return VCY_OK;
  const std::string w(where);
  if (w.rfind("file:", 0) == 0 && w.size() > 5) return rendezvous_file(w.substr(5), rank, payload, timeout_ms);
  if (w.rfind("tcp:", 0) == 0) {
    const size_t colon = w.rfind(':');
    if (colon != std::string::npos && colon > 4) {
      const int port = std::atoi(w.c_str() + colon + 1);
      if (port > 0 && port < 65536) return rendezvous_tcp(w.substr(4, colon - 4), port, rank, world, payload, timeout_ms);
    }
  }
  set_error("rendezvous: expected \"file:<path>\" or \"tcp:<host>:<port>\", got \"%s\"", where);
  return VCY_ERR_INVALID_ARG;
}

}  // namespace
}  // namespace vcy

struct vcy_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  char* send = nullptr;
  char* recv = nullptr;
  size_t send_bytes = 0, recv_bytes = 0;
};

extern "C" {

int vcy_rendezvous_exchange(int rank, int world, const char* where, void* payload128, int timeout_ms) {
  return rendezvous(where, rank, world, payload128, timeout_ms);
}

int vcy_comm_create(int rank, int world, int device_id, const char* where, int timeout_ms, vcy_comm** out) {
  if (!out || rank < 0 || world < 1 || rank >= world) {
    set_error("vcy_comm_create: invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  *out = nullptr;
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!load_rccl()) return VCY_ERR_UNSUPPORTED;
  VCY_HIP_CHECK(hipSetDevice(device_id));
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  if (rank == 0) VCY_NCCL_CHECK(g_rccl.GetUniqueId(&id));
  {
    const int rc = rendezvous(where ? where : "", rank, world, &id, timeout_ms > 0 ? timeout_ms : 120000);
    if (rc != VCY_OK && world > 1) return rc;
  }
  vcy_comm* c = new vcy_comm();
  c->rank = rank, c->world = world, c->device = device_id;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device_id, g_rccl.GetErrorString(r));
    delete c;
    return VCY_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    (void)g_rccl.CommDestroy(c->comm);
    delete c;
    set_error("vcy_comm_create: stream");
    return VCY_ERR_HIP;
  }
  *out = c;
  return VCY_OK;
}

void vcy_comm_destroy(vcy_comm* c) {
  if (!c) return;
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  (void)hipSetDevice(c->device);
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  (void)hipFree(c->send);
  (void)hipFree(c->recv);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

int vcy_halo_allgather_ranks(vcy_comm* cm, vcy_ctx* const* my_slabs, int n_my_slabs) {
  if (!cm || !my_slabs || n_my_slabs <= 0) {
    set_error("vcy_halo_allgather_ranks: invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_my_slabs; ++i)
    if (!my_slabs[i] || my_slabs[i]->device != cm->device) {
      set_error("vcy_halo_allgather_ranks: slab %d is not a context on this rank's device", i);
      return VCY_ERR_INVALID_ARG;
    }
  const int k = n_my_slabs, world = cm->world;
  if (world * k == 1) {
    my_slabs[0]->halo_valid = true;
    return VCY_OK;
  }
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  VCY_HIP_CHECK(hipSetDevice(cm->device));
  const size_t pack = (size_t)vcy_halo_bytes(my_slabs[0]);
  const size_t send_bytes = pack * (size_t)k, recv_bytes = send_bytes * (size_t)world;
  if (cm->send_bytes < send_bytes || cm->recv_bytes < recv_bytes) {
    VCY_HIP_CHECK(hipStreamSynchronize(cm->stream));
    (void)hipFree(cm->send);
    (void)hipFree(cm->recv);
    cm->send = cm->recv = nullptr;
    cm->send_bytes = cm->recv_bytes = 0;
    VCY_HIP_CHECK(hipMalloc((void**)&cm->send, send_bytes));
    VCY_HIP_CHECK(hipMalloc((void**)&cm->recv, recv_bytes));
    cm->send_bytes = send_bytes, cm->recv_bytes = recv_bytes;
  }
  for (int i = 0; i < k; ++i) {
    const int rc = vcy_halo_pack(my_slabs[i], cm->send + pack * (size_t)i);  // (applies queued views first)
    if (rc != VCY_OK) return rc;
  }
  for (int i = 0; i < k; ++i) VCY_HIP_CHECK(hipStreamSynchronize(my_slabs[i]->stream));
  // the single collective of the path
  VCY_NCCL_CHECK(g_rccl.AllGather(cm->send, cm->recv, send_bytes, ncclUint8, cm->comm, cm->stream));
  VCY_HIP_CHECK(hipStreamSynchronize(cm->stream));
  for (int i = 0; i < k; ++i) {
    const int sid = cm->rank + i * world;  // this slab's id; the slab below it is sid - 1, held by rank (sid - 1) % world
    const char* src = nullptr;
    if (sid > 0) {
      const int below = sid - 1;
      src = cm->recv + ((size_t)(below % world) * (size_t)k + (size_t)(below / world)) * pack;
    }
    const int rc = vcy_halo_install(my_slabs[i], src);
    if (rc != VCY_OK) return rc;
  }
  for (int i = 0; i < k; ++i) VCY_HIP_CHECK(hipStreamSynchronize(my_slabs[i]->stream));
  g_last.ranks = world;
  g_last.bytes_per_rank = (int64_t)send_bytes;
  g_last.calls += 1;
  (void)g_rccl.GetVersion(&g_last.version);
  return VCY_OK;
}

}  // extern "C"
