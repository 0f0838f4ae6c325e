// How many 512-thread workgroups a CU holds as a function of their dynamic LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor),
// and the device's LDS figures.  hipcc --offload-arch=gfx950 -O3 tools/calib_lds_occupancy.hip -o /tmp/calib_lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void dummy(int* out) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (out) out[threadIdx.x] = s[threadIdx.x ^ 1];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  std::printf("sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu regsPerMultiprocessor %d\n", p.sharedMemPerBlock,
              p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor);
  for (int kb = 36; kb <= 64; kb += 2) {
    int blocks = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, dummy, 512, static_cast<size_t>(kb) * 1024);
    std::printf("dynamic LDS %2d KB -> %d workgroups of 512 per CU\n", kb, blocks);
  }
  return 0;
}
