// Measured device-memory bandwidth (vcy_measure_bandwidth): the second denominator of the roofline
// next to the 8 TB/s vendor figure.  Not on the carve path.
#include <algorithm>
#include <vector>

#include "vcy_internal.h"

namespace vcy {
namespace {

// Streaming read with the access shape of the grid sweeps in this library: one dword per lane per
// load, eight independent loads in flight, a workgroup walks a contiguous 8 KB chunk.
__global__ __launch_bounds__(256) void bw_read_kernel(const float* __restrict__ p, int64_t n, float* sink) {
  const int64_t base = (int64_t)blockIdx.x * (256 * 8);
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    v[k] = i < n ? p[i] : 0.0f;
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 123.456f) sink[0] = s;  // never true for the zero-filled buffer; keeps the loads alive
}

// Shader clock beside a running workload (vcy_clock_probe_*): ONE wave on a stream of its own samples the shader
// clock counter (s_memtime) against the constant 100 MHz reference (s_memrealtime) every few microseconds until the
// host raises a flag in page-locked memory.  It runs beside the kernels being timed -- an idle MI355X drops its clock
// within a millisecond, so a probe kernel run by itself afterwards would not see the clock the workload ran at.
struct ClockSample {
  unsigned long long shader, real;
};
// The wave also ends by itself after `max_ticks` of the 100 MHz reference (kClockProbeMaxMs): while it is resident every
// implicitly device-synchronising call of the workload (hipFree / hipMalloc of a growing pool, hipHostFree) blocks until
// it has finished, so a probe nobody stops must not hold the device for long -- and the probed region must be warmed up
// so that it allocates nothing (bench.py runs it behind the timed steps of a workload that has already run).
constexpr int kClockProbeMaxSamples = 1 << 16;  // ~1.8 s at a sample every ~28 us
constexpr double kClockProbeMaxMs = 2000.0;
__global__ __launch_bounds__(64) void clock_probe_kernel(ClockSample* __restrict__ samples, int max_samples,
                                                         const volatile int* __restrict__ stop, int* __restrict__ count,
                                                         unsigned long long max_ticks) {
  if (threadIdx.x != 0) return;
  int k = 0;
  const unsigned long long t_begin = wall_clock64();
  for (; k < max_samples; ++k) {
    ClockSample s;
    s.shader = __builtin_readcyclecounter();  // s_memtime
    s.real = wall_clock64();                  // s_memrealtime, 100 MHz
    samples[k] = s;
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || s.real - t_begin > max_ticks) {
      ++k;
      break;
    }
    for (int q = 0; q < 8; ++q) __builtin_amdgcn_s_sleep(127);  // ~8 x 8128 cycles: a sample every ~25-30 us
  }
  *count = k;
}

}  // namespace
}  // namespace vcy

struct vcy_clock_probe {
  int device = 0;
  hipStream_t stream = nullptr;
  vcy::ClockSample* d_samples = nullptr;
  int* d_count = nullptr;
  int* h_stop = nullptr;  // page-locked, read by the kernel
  int max_samples = 0;
};

extern "C" int vcy_clock_probe_start(int device_id, int max_samples, vcy_clock_probe** out) {
  using namespace vcy;
  if (!out || max_samples < 2 || max_samples > kClockProbeMaxSamples) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  *out = nullptr;
  VCY_HIP_CHECK(hipSetDevice(device_id));
  vcy_clock_probe* p = new vcy_clock_probe();
  p->device = device_id;
  p->max_samples = max_samples;
  hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_samples, sizeof(ClockSample) * (size_t)max_samples);
  if (e == hipSuccess) e = hipMalloc((void**)&p->d_count, sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc((void**)&p->h_stop, sizeof(int), hipHostMallocDefault);
  if (e == hipSuccess) {
    *p->h_stop = 0;
    e = hipMemsetAsync(p->d_count, 0, sizeof(int), p->stream);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, p->stream, p->d_samples, max_samples, p->h_stop, p->d_count,
                       (unsigned long long)(kClockProbeMaxMs * 1.0e5));
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    set_error("clock probe: %s", hipGetErrorString(e));
    if (p->h_stop) (void)hipHostFree(p->h_stop);
    if (p->d_count) (void)hipFree(p->d_count);
    if (p->d_samples) (void)hipFree(p->d_samples);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
    return VCY_ERR_HIP;
  }
  *out = p;
  return VCY_OK;
}

extern "C" int vcy_clock_probe_stop(vcy_clock_probe* p, double* mean_hz, double* settled_hz, double* min_hz, double* max_hz,
                                    int* n_samples, double* covered_ms) {
  using namespace vcy;
  if (!p) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  int rc = VCY_OK;
  (void)hipSetDevice(p->device);
  __atomic_store_n(p->h_stop, 1, __ATOMIC_RELEASE);
  hipError_t e = hipStreamSynchronize(p->stream);
  int n = 0;
  std::vector<ClockSample> s;
  if (e == hipSuccess) e = hipMemcpy(&n, p->d_count, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess && n > 0) {
    s.resize((size_t)n);
    e = hipMemcpy(s.data(), p->d_samples, sizeof(ClockSample) * (size_t)n, hipMemcpyDeviceToHost);
  }
  if (e != hipSuccess) {
    set_error("clock probe: %s", hipGetErrorString(e));
    rc = VCY_ERR_HIP;
  }
  double lo = 0.0, hi = 0.0, mean = 0.0, covered = 0.0, settled = 0.0;
  if (rc == VCY_OK && n >= 2) {
    // per interval: shader cycles / reference ticks x 100 MHz; the mean over the whole span
    lo = 1e30;
    for (int k = 1; k < n; ++k) {
      const double dr = (double)(s[(size_t)k].real - s[(size_t)k - 1].real);
      if (dr <= 0.0) continue;
      const double hz = (double)(s[(size_t)k].shader - s[(size_t)k - 1].shader) / dr * 100.0e6;
      lo = std::min(lo, hz);
      hi = std::max(hi, hz);
    }
    const double span = (double)(s[(size_t)n - 1].real - s[0].real);
    if (span > 0.0) mean = (double)(s[(size_t)n - 1].shader - s[0].shader) / span * 100.0e6;
    covered = span / 100.0e6 * 1e3;
    if (lo > 1e29) lo = 0.0;
    // the second half of the span alone: the device has idled for a millisecond while the probe was set up, and its
    // clock needs some milliseconds of load to come back (profiles/r04/clock_ramp.txt)
    const size_t h = (size_t)n / 2;
    const double span2 = (double)(s[(size_t)n - 1].real - s[h].real);
    settled = span2 > 0.0 ? (double)(s[(size_t)n - 1].shader - s[h].shader) / span2 * 100.0e6 : mean;
  }
  if (mean_hz) *mean_hz = mean;
  if (settled_hz) *settled_hz = settled;
  if (min_hz) *min_hz = lo;
  if (max_hz) *max_hz = hi;
  if (n_samples) *n_samples = n;
  if (covered_ms) *covered_ms = covered;
  (void)hipHostFree(p->h_stop);
  (void)hipFree(p->d_count);
  (void)hipFree(p->d_samples);
  (void)hipStreamDestroy(p->stream);
  delete p;
  return rc;
}

extern "C" int vcy_measure_bandwidth(int device_id, uint64_t bytes, int reps, double* read_gbs, double* copy_gbs) {
  using namespace vcy;
  if (bytes < (1u << 20) || reps < 1 || (!read_gbs && !copy_gbs)) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(device_id));
  const int64_t n = (int64_t)(bytes / 4);
  float *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = VCY_OK;
  auto fail = [&](hipError_t e, const char* what) {
    set_error("%s failed: %s", what, hipGetErrorString(e));
    rc = VCY_ERR_HIP;
  };
  hipError_t e = hipMalloc(&a, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc(&b, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemset(a, 0, (size_t)n * 4);
  if (e == hipSuccess) e = hipMemset(b, 0, (size_t)n * 4);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e != hipSuccess) fail(e, "bandwidth probe setup");
  double best_read = 0.0, best_copy = 0.0;
  for (int r = 0; rc == VCY_OK && r < reps + 1; ++r) {  // first pass is a warm-up
    float ms = 0.0f;
    (void)hipEventRecord(e0, nullptr);
    hipLaunchKernelGGL(bw_read_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, nullptr, a, n, b);
    (void)hipEventRecord(e1, nullptr);
    e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) { fail(e, "bandwidth probe (read)"); break; }
    if (r > 0 && ms > 0.0f) best_read = std::max(best_read, (double)n * 4.0 / (ms * 1e-3) / 1e9);
    (void)hipEventRecord(e0, nullptr);
    e = hipMemcpyAsync(b, a, (size_t)n * 4, hipMemcpyDeviceToDevice, nullptr);
    (void)hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) { fail(e, "bandwidth probe (copy)"); break; }
    if (r > 0 && ms > 0.0f) best_copy = std::max(best_copy, (double)n * 8.0 / (ms * 1e-3) / 1e9);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  if (rc == VCY_OK) {
    if (read_gbs) *read_gbs = best_read;
    if (copy_gbs) *copy_gbs = best_copy;
  }
  return rc;
}
