// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns this project uses
// (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-counts wide coalesced reads by 2x; other widths must be calibrated).
//   k_stream16 : 16 B per lane coalesced read   (known bytes = N*16)
//   k_stream8  : 8 B per lane coalesced read    (known bytes = N*8)
//   k_gather8  : 8 B per lane random gather over a 1 GiB table (known useful bytes = N*8; lines touched = N*64..128)
//   k_write8   : 8 B per lane coalesced write   (known bytes = N*8)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_stream16(const double2* p, double* out, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) { double2 v = p[i]; if (v.x == 123.456) out[0] = v.y; } }
__global__ void k_stream8(const double* p, double* out, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) { double v = p[i]; if (v == 123.456) out[0] = v; } }
__global__ void k_gather8(const double* p, double* out, size_t n, size_t mask) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) { size_t j = (i * 0x9E3779B97F4A7C15ull >> 20) & mask; double v = p[j]; if (v == 123.456) out[0] = v; } }
__global__ void k_write8(double* p, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) p[i] = 1.0; }
int main() {
  const size_t n = 1ull << 27;  // 128 Mi elements: 1 GiB of doubles, 2 GiB of double2 (beyond the 256 MiB Infinity Cache)
  double* a; double* out;
  hipMalloc(&a, n * 16); hipMalloc(&out, 8);
  hipMemset(a, 0, n * 16);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_stream16, dim3(n / 256), dim3(256), 0, 0, (const double2*)a, out, n);
    hipLaunchKernelGGL(k_stream8, dim3(n / 256), dim3(256), 0, 0, a, out, n);
    hipLaunchKernelGGL(k_gather8, dim3(n / 256), dim3(256), 0, 0, a, out, n, n - 1);
    hipLaunchKernelGGL(k_write8, dim3(n / 256), dim3(256), 0, 0, a, n);
  }
  hipDeviceSynchronize();
  printf("n=%zu stream16_bytes=%zu stream8_bytes=%zu gather8_useful=%zu write8_bytes=%zu\n", n, n * 16, n * 8, n * 8, n * 8);
  return 0;
}
