// Issue rate of f64 VALU instructions on gfx950: v_fma_f64 vs v_mul_f64 / v_add_f64 (build: hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kMode>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int iters) {
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3 + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (kMode == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
      if (kMode == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[k]) : "v"(a));
      if (kMode == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[k]) : "v"(b));
      if (kMode == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(reinterpret_cast<int&>(v[k])) : "v"(i));
    }
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int kMode>
void run(const char* name, double* d, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = double(blocks) * 4 * iters * 8;
  // 256 CUs x 4 SIMDs; cycles per wave instruction per SIMD at 2.4 GHz
  printf("%-10s %.3f ms  -> %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
}
int main() {
  double* d; hipMalloc(&d, 256 * 8192 * 8);
  const int blocks = 256 * 8 * 2, iters = 4096;  // 8 waves per SIMD
  run<0>("v_fma_f64", d, blocks, iters);
  run<1>("v_mul_f64", d, blocks, iters);
  run<2>("v_add_f64", d, blocks, iters);
  run<3>("v_add_u32", d, blocks, iters);
  return 0;
}
