// How many 512-thread workgroups a CU really holds as a function of their dynamic LDS size (the occupancy API's answer is not
// the hardware's): 256 x 12 workgroups, each spinning ~50 us; the kernel's time / 50 us = rounds = 12 / resident per CU.
// hipcc --offload-arch=gfx950 -O3 tools/calib_lds_residency.hip -o build/calib_lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void spin(long long cycles, int* out) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (out && s[threadIdx.x ^ 1] == -1) out[0] = 1;
}
int main() {
  int* out;
  (void)hipMalloc(&out, 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const long long cycles = 100000;
  for (int kb = 30; kb <= 64; kb += 1) {
    int api = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin, 512, static_cast<size_t>(kb) * 1024);
    spin<<<256 * 12, 512, static_cast<size_t>(kb) * 1024>>>(cycles, out);
    (void)hipEventRecord(e0);
    spin<<<256 * 12, 512, static_cast<size_t>(kb) * 1024>>>(cycles, out);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static float unit = 0.f;
    if (kb == 30) unit = ms / 3.f;  // 4 resident (8 waves per SIMD is the cap): 3 rounds
    std::printf("dynamic LDS %2d KB: API %d per CU; %.1f us = %.2f rounds -> %.2f resident per CU\n", kb, api, ms * 1e3, ms / unit, 12.f / (ms / unit));
  }
  return 0;
}
