// Where the waves of co-resident workgroups land: SIMD of every wave (HW_REG_HW_ID) for the LF patch kernel's shape - 512 threads, 52 KB of
// workgroup memory, 80 registers (three workgroups per CU) - and for 384 threads with 38 KB (four per CU).  The LF patch kernel's workgroup is
// seven waves of particles and a producer wave that issues little: which SIMDs the producers share decides how evenly the four SIMDs of a CU
// are loaded.
// hipcc --offload-arch=gfx950 -O3 tools/calib_simd_placement.hip -o /tmp/calib_simd && /tmp/calib_simd
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <vector>

template <int kThreads>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(6, 6))) void probe(long long cycles, unsigned* out) {
  extern __shared__ int s[];
  s[threadIdx.x] = threadIdx.x;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const unsigned wave = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
    out[2 * wave] = hw;
    out[2 * wave + 1] = xcc;
  }
  if (s[threadIdx.x ^ 1] == -1) out[0] = 1;
}

template <int kThreads>
void run(int per_cu, size_t lds) {
  constexpr int kWaves = kThreads / 64;
  const int blocks = 256 * per_cu;
  unsigned* d;
  (void)hipMalloc(&d, sizeof(unsigned) * 2 * blocks * kWaves);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<kThreads>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  probe<kThreads><<<blocks, kThreads, lds>>>(400000, d);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(2 * blocks * kWaves);
  (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  // key: (xcc, se, sh, cu) -> per SIMD: waves, and how many of them are a workgroup's LAST wave
  std::map<unsigned, std::vector<int>> waves, last;
  std::map<unsigned, std::map<int, std::vector<int>>> simd_of;  // cu -> block -> simd of wave w
  for (int b = 0; b < blocks; ++b)
    for (int w = 0; w < kWaves; ++w) {
      const unsigned hw = h[2 * (b * kWaves + w)], xcc = h[2 * (b * kWaves + w) + 1] & 0xF;
      const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      waves[key].resize(4);
      last[key].resize(4);
      waves[key][simd] += 1;
      if (w == kWaves - 1) last[key][simd] += 1;
      simd_of[key][b].push_back(static_cast<int>(simd));
    }
  std::printf("== %d threads, %zu KB of workgroup memory, %d workgroups per CU asked for: %zu CUs seen\n", kThreads, lds / 1024, per_cu, waves.size());
  std::map<std::string, int> patterns;
  int shown = 0;
  for (auto& [key, per_simd] : waves) {
    char buf[160];
    std::snprintf(buf, sizeof buf, "waves per SIMD %d %d %d %d | last waves per SIMD %d %d %d %d", per_simd[0], per_simd[1], per_simd[2], per_simd[3],
                  last[key][0], last[key][1], last[key][2], last[key][3]);
    patterns[buf] += 1;
    if (shown < 3) {
      std::printf("  CU %05x:", key);
      for (auto& [b, v] : simd_of[key]) {
        std::printf(" wg%d[", b);
        for (int s : v) std::printf("%d", s);
        std::printf("]");
      }
      std::printf("\n");
      ++shown;
    }
  }
  for (auto& [p, c] : patterns) std::printf("  %4d CUs: %s\n", c, p.c_str());
  (void)hipFree(d);
}

int main() {
  run<512>(3, 52 * 1024);
  run<384>(4, 38 * 1024);
  run<384>(4, 39 * 1024);
  run<320>(4, 38 * 1024);
  return 0;
}
