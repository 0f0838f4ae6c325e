// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950 and how it shares a SIMD with f64 VALU work of OTHER waves:
//   mode 0: every wave runs MFMAs (4 independent accumulators);  mode 1: every wave runs v_fma_f64 (8 independent chains);
//   mode 2: even waves MFMA, odd waves FMA (same per-wave counts as modes 0 / 1).
// One workgroup of `waves` waves per CU; cycles by s_memtime of wave 0.
// hipcc --offload-arch=gfx950 -O3 tools/calib_mfma_f64.hip -o /tmp/calib_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int kIters = 4096;
__global__ __launch_bounds__(1024) void k(int mode, double* out, long long* cycles) {
  const int wave = threadIdx.x >> 6;
  const bool mfma = mode == 0 || (mode == 2 && ((wave >> 2) & 1) == 0);
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  double r = 0;
  if (mfma) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll 1
    for (int i = 0; i < kIters / 4; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    r = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    double c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
#pragma unroll 1
    for (int i = 0; i < kIters / 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_fma(a, c[j], b);
    }
    for (int j = 0; j < 8; ++j) r += c[j];
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}
int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(double));
  hipMalloc(&cyc, 256 * 16 * sizeof(long long));
  for (int waves : {8, 16}) {
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      k<<<256, waves * 64>>>(mode, out, cyc);
      hipEventRecord(e0);
      k<<<256, waves * 64>>>(mode, out, cyc);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(256 * 16);
      hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
      std::printf("waves/CU %2d mode %d: kernel %.1f us; wave 0 %lld ticks, wave 4 %lld ticks for %d instructions -> %.2f / %.2f ticks per instruction per wave; per SIMD (%d waves): %.2f us/instr-round\n",
                  waves, mode, ms * 1e3, h[0], h[4], kIters, double(h[0]) / kIters, double(h[4]) / kIters, waves / 4, ms * 1e3 / kIters);
    }
  }
  return 0;
}
