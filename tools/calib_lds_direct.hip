// Semantics of buffer_load_dwordx4 ... lds on gfx950 (global -> LDS without passing through registers), checked on the device:
// lane i's 16 bytes land at M0 base + 16 i whatever its global offset, inactive lanes leave their slot alone, vmcnt covers
// the LDS write.  hipcc --offload-arch=gfx950 -O3 tools/calib_lds_direct.hip -o /tmp/calib_lds_direct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* src, uint32_t* out, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* s = reinterpret_cast<uint32_t*>(smem);
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 0xDEAD0000u + i;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, 1 << 20, 0x00020000);
  const uint32_t lane = threadIdx.x;
  // lane i fetches the 16 bytes at global offset 16 * (63 - i) * 3 (a permuted, strided pattern), scalar offset 64
  const uint32_t voffset = 16u * (63u - lane) * 3u;
  if (mode == 0 || lane < 8)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + 256), 16, voffset, 64, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = s[i];
}
int main() {
  std::vector<uint32_t> h(1 << 18);
  for (size_t i = 0; i < h.size(); ++i) h[i] = static_cast<uint32_t>(i);
  uint32_t *d, *o;
  hipMalloc(&d, h.size() * 4);
  hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, o, mode);
    std::vector<uint32_t> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
      const int slot = (i - 64) / 4, word = (i - 64) % 4;  // dword index 64 = byte 256
      uint32_t want = 0xDEAD0000u + i;
      if (i >= 64 && slot < 64 && (mode == 0 || slot < 8)) want = (64u + 16u * (63u - slot) * 3u) / 4u + word;
      if (r[i] != want) {
        if (bad < 5) std::printf("mode %d dword %d: got %08x want %08x\n", mode, i, r[i], want);
        ++bad;
      }
    }
    std::printf("mode %d (%s): %d mismatches\n", mode, mode ? "lanes 0..7 only" : "all lanes", bad);
  }
  return 0;
}
