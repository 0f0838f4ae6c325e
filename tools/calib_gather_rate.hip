// Cost of a 64-lane 2-byte gather (buffer_load_ushort) on gfx950 as a function of how the lanes' addresses are spread
// over 128-byte lines.  All addresses stay inside a 64 KB window (L1/L2 resident): this measures the vector-memory
// front end (TA / L1 tag rate), not DRAM.   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k(const uint16_t* table, uint32_t bytes, uint32_t* out, int pattern, int iters) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(table), 0, static_cast<int>(bytes), 0x00020000);
  const uint32_t lane = threadIdx.x & 63;
  uint32_t base;
  switch (pattern) {
    case 0: base = 0; break;                                   // all lanes one address
    case 1: base = (lane >> 2) * 128; break;                   // one line per quad, same address inside a quad
    case 2: base = (lane >> 2) * 128 + (lane & 3) * 2; break;  // one line per quad, adjacent cells
    case 3: base = (lane >> 1) * 128; break;                   // two lines per quad
    case 4: base = lane * 128; break;                          // one line per lane
    case 5: base = (lane >> 4) * 128 + (lane & 15) * 2; break; // four lines per wave
    case 7: base = (lane & 3) * 128 + (lane >> 2) * 2; break;  // four lines per wave, interleaved lane by lane (64 runs)
    case 8: base = (lane & 1) * 128 + (lane >> 1) * 2; break;  // two lines per wave, interleaved lane by lane
    case 9: base = ((lane >> 2) & 3) * 128 + (lane & 3) * 2 + (lane >> 4) * 8; break;  // four lines, runs of 4 lanes (16 runs)
    case 10: base = ((lane >> 3) & 3) * 128 + (lane & 7) * 2 + (lane >> 5) * 16; break;  // four lines, runs of 8 lanes (8 runs)
    case 11: base = ((lane * 37) & 63) * 2; break;              // one line, permuted lanes
    case 12: base = (lane & 3) * 64 + (lane >> 2) * 2; break;  // two lines as four 64-byte halves, interleaved lane by lane
    default: base = lane * 2; break;                           // 6: fully coalesced
  }
  uint32_t acc = 0, off = base + (threadIdx.x >> 6) * 8192;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc += static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rsrc, (off + u * 16) & 0xFFFEu, 0, 0));
    }
    off += 2;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  uint16_t* t; uint32_t* o;
  hipMalloc(&t, 65536); hipMemset(t, 0, 65536); hipMalloc(&o, 256 * 4096 * 4);
  const int blocks = 256 * 8, iters = 2048;  // 8 waves per SIMD
  const char* names[] = {"same address", "1 line/quad (same cell)", "1 line/quad (4 cells)", "2 lines/quad", "1 line/lane", "4 lines/wave", "coalesced", "4 lines interleaved", "2 lines interleaved", "4 lines runs of 4", "4 lines runs of 8", "1 line permuted", "4 half-lines interleaved"};
  for (int p = 0; p < 13; ++p) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, t, 65536u, o, p, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, t, 65536u, o, p, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = double(blocks) * 4 * iters * 8 / 256;
    printf("%-26s %.3f ms -> %.1f cycles per 64-lane gather per CU (at 2.4 GHz)\n", names[p], ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  }
  return 0;
}
