// calib_sync.hip — host <-> device hand-off costs at the two ends of an update cycle (gfx950).
//   (a) trivial kernel + hipStreamSynchronize                          (what a cycle's final wait costs today)
//   (b) trivial kernel that stores a sequence number into mapped pinned host memory, host spins on it
//   (c) hipMemcpyAsync(17 KB, pinned -> device) + kernel + sync         (the scan upload in front of a cycle)
//   (d) kernel that pulls the 17 KB from mapped pinned memory itself + sync
//   (e) N dependent trivial kernels + sync                              (per-boundary cost)
// Build: hipcc --offload-arch=gfx950 -O3 -o calib_sync tools/calib_sync.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void k_nop(double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1.0;
}
__global__ void k_flag(double* out, volatile unsigned long long* host_flag, unsigned long long seq) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] += 1.0;
    host_flag[1] = 42;  // payload
    __threadfence_system();
    host_flag[0] = seq;
  }
}
__global__ void k_pull(const double* __restrict__ host_src, double* __restrict__ dst, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = host_src[i];
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double* d;
  CK(hipMalloc(reinterpret_cast<void**>(&d), 4096 * sizeof(double)));
  CK(hipMemset(d, 0, 4096 * sizeof(double)));
  unsigned long long* h_flag;
  CK(hipHostMalloc(reinterpret_cast<void**>(&h_flag), 64, hipHostMallocMapped));
  unsigned long long* hd_flag;
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd_flag), h_flag, 0));
  double* h_pts;
  const int npts = 2160;
  CK(hipHostMalloc(reinterpret_cast<void**>(&h_pts), npts * sizeof(double), hipHostMallocMapped));
  double* hd_pts;
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd_pts), h_pts, 0));
  for (int i = 0; i < npts; ++i) h_pts[i] = i;
  const int reps = 2000;
  // warm up
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d);
  CK(hipStreamSynchronize(st));

  {
    const double t0 = now_us();
    for (int i = 0; i < reps; ++i) {
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d);
      CK(hipStreamSynchronize(st));
    }
    std::printf("(a) kernel + hipStreamSynchronize                 : %7.2f us\n", (now_us() - t0) / reps);
  }
  {
    h_flag[0] = 0;
    const double t0 = now_us();
    for (int i = 1; i <= reps; ++i) {
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, d, hd_flag, static_cast<unsigned long long>(i));
      while (*reinterpret_cast<volatile unsigned long long*>(h_flag) != static_cast<unsigned long long>(i)) {
      }
    }
    std::printf("(b) kernel + spin on a mapped host flag            : %7.2f us\n", (now_us() - t0) / reps);
    CK(hipStreamSynchronize(st));
  }
  {
    const double t0 = now_us();
    for (int i = 0; i < reps; ++i) {
      CK(hipMemcpyAsync(d + 8, h_pts, npts * sizeof(double), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d);
      CK(hipStreamSynchronize(st));
    }
    std::printf("(c) memcpyAsync 17 KB H2D + kernel + sync          : %7.2f us\n", (now_us() - t0) / reps);
  }
  {
    const double t0 = now_us();
    for (int i = 0; i < reps; ++i) {
      hipLaunchKernelGGL(k_pull, dim3(1), dim3(256), 0, st, hd_pts, d + 8, npts);
      CK(hipStreamSynchronize(st));
    }
    std::printf("(d) kernel pulling 17 KB from mapped memory + sync : %7.2f us\n", (now_us() - t0) / reps);
  }
  {
    h_flag[0] = 0;
    const double t0 = now_us();
    for (int i = 1; i <= reps; ++i) {
      hipLaunchKernelGGL(k_pull, dim3(1), dim3(256), 0, st, hd_pts, d + 8, npts);
      hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, d, hd_flag, static_cast<unsigned long long>(i));
      while (*reinterpret_cast<volatile unsigned long long*>(h_flag) != static_cast<unsigned long long>(i)) {
      }
    }
    std::printf("(d') pull kernel + flag kernel + spin              : %7.2f us\n", (now_us() - t0) / reps);
    CK(hipStreamSynchronize(st));
  }
  for (int chain : {4, 8, 16}) {
    const double t0 = now_us();
    for (int i = 0; i < reps / 4; ++i) {
      for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(k_nop, dim3(256), dim3(256), 0, st, d);
      CK(hipStreamSynchronize(st));
    }
    std::printf("(e) %2d dependent kernels + sync                    : %7.2f us\n", chain, (now_us() - t0) / (reps / 4));
  }
  {
    // launch enqueue cost alone (host side)
    const double t0 = now_us();
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, d);
    const double t1 = now_us();
    CK(hipStreamSynchronize(st));
    std::printf("(f) host enqueue cost per launch (back to back)    : %7.2f us\n", (t1 - t0) / reps);
  }
  return 0;
}
