// Measured device-memory bandwidth (vcy_measure_bandwidth): the second denominator of the roofline
// next to the 8 TB/s vendor figure.  Not on the carve path.
#include <algorithm>

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

}  // namespace
}  // namespace vcy

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
