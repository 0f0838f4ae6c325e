// v_mfma_f32_32x32x2_f32 on gfx950, for a likelihood-field look-up whose end-points would come out of the matrix pipe:
//  (1) its arithmetic: D = C + a0 b0 + a1 b1 against candidate single-precision evaluation orders (bit patterns compared);
//  (2) whether it overlaps with the vector work of the SAME and of OTHER waves of a SIMD (6 waves per SIMD, as the LF kernel runs):
//      mode 0 = the vector mix alone, 1 = the MFMAs alone, 2 = both interleaved in every wave.
// hipcc --offload-arch=gfx950 -O3 tools/calib_mfma_f32.hip -o /tmp/calib_mfma_f32
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
using f16v = __attribute__((ext_vector_type(16))) float;

__global__ void k_semantics(const float* a, const float* b, const float* c, float* d) {
  const int lane = threadIdx.x;
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = c[lane * 16 + r];
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[lane], b[lane], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[lane * 16 + r] = acc[r];
}

template <int kMode>
__global__ __launch_bounds__(512) void k_overlap(float* out, int iters, unsigned sel) {
  __shared__ double lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x + 512] = threadIdx.x;
  __syncthreads();
  f16v d0 = {}, d1 = {};
  float a = threadIdx.x * 1e-3f, b = 1.f - threadIdx.x * 1e-3f;
  double acc[4] = {0, 1, 2, 3};
  unsigned u[4] = {threadIdx.x, threadIdx.x * 3, threadIdx.x * 5, threadIdx.x * 7};
  unsigned guard = 0xFFFFFFFFu;
  for (int i = 0; i < iters; ++i) {
    if (kMode != 0) {  // a tile: 2 chained pairs
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
    }
    if (kMode != 1) {  // the vector work of 16 beams: guard 1.5, address 2, two LDS reads, f64 adds 1.75
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        unsigned t, adr, idx;
        asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(u[k & 3]), "v"(u[(k + 1) & 3]));
        if (k & 1) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(guard) : "v"(t), "v"(u[k & 3]));
        asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(adr) : "v"(u[k & 3]), "s"(sel), "v"(t));
        asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(adr) : "v"(u[(k + 2) & 3]), "s"(sel), "v"(adr));
        adr &= 0xFF8u;
        asm volatile("ds_read_u16 %0, %1" : "=v"(idx) : "v"(adr));
        asm volatile("s_waitcnt lgkmcnt(0)");
        idx &= 0xFF8u;
        double v;
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(idx));
        acc[k & 3] += v;
        if ((k & 3) == 3) acc[0] += acc[1];
        u[k & 3] += adr;
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += d0[r] + d1[r];
  out[blockIdx.x * 512 + threadIdx.x] = s + static_cast<float>(acc[0] + acc[1] + acc[2] + acc[3]) + guard + u[0];
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main() {
  // ---- (1) semantics
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> small(-0.1f, 0.1f), range(-600.f, 600.f), bias(128.f, 192.f);
  long n = 0, m_fma01 = 0, m_fma10 = 0, m_exact = 0, m_sep = 0;
  double worst = 0;
  float *da, *db, *dc, *dd;
  (void)hipMalloc(&da, 64 * 4); (void)hipMalloc(&db, 64 * 4); (void)hipMalloc(&dc, 1024 * 4); (void)hipMalloc(&dd, 1024 * 4);
  for (int trial = 0; trial < 400; ++trial) {
    std::vector<float> a(64), b(64), c(1024), d(1024);
    for (int l = 0; l < 64; ++l) {
      a[l] = (trial & 1) ? range(rng) : small(rng) * 6000.f;  // A[i = l % 32][k = l / 32]
      b[l] = (trial & 2) ? range(rng) / 600.f : small(rng);   // B[k = l / 32][j = l % 32]
    }
    for (auto& v : c) v = (trial & 4) ? bias(rng) : small(rng) * 640.f;
    (void)hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(dc, c.data(), 4096, hipMemcpyHostToDevice);
    k_semantics<<<1, 64>>>(da, db, dc, dd);
    (void)hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int j = l % 32, i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
        const float a0 = a[i], a1 = a[32 + i], b0 = b[j], b1 = b[32 + j], cc = c[l * 16 + r], got = d[l * 16 + r];
        const float f01 = std::fmaf(a1, b1, std::fmaf(a0, b0, cc)), f10 = std::fmaf(a0, b0, std::fmaf(a1, b1, cc));
        const double exact = double(cc) + double(a0) * double(b0) + double(a1) * double(b1);
        const float ex = static_cast<float>(exact);
        volatile float p0 = a0 * b0, p1 = a1 * b1;
        const float sep = (cc + p0) + p1;
        ++n;
        m_fma01 += bits(got) != bits(f01);
        m_fma10 += bits(got) != bits(f10);
        m_exact += bits(got) != bits(ex);
        m_sep += bits(got) != bits(sep);
        const double err = std::fabs(double(got) - exact) / std::ldexp(1.0, std::ilogb(std::fmax(std::fabs(exact), 1e-30)) - 23);
        if (err > worst) worst = err;
      }
  }
  std::printf("semantics over %ld outputs: differs from fma(a1,b1,fma(a0,b0,c)) %ld, from fma(a0,b0,fma(a1,b1,c)) %ld, from the exactly rounded sum %ld, from separately rounded products %ld; worst error %.3f ulp of the result\n",
              n, m_fma01, m_fma10, m_exact, m_sep, worst);
  // ---- (2) overlap
  float* out;
  (void)hipMalloc(&out, 768 * 512 * 4);
  const int iters = 2000;
  auto time = [&](auto kernel, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kernel<<<768, 512>>>(out, iters, 0x90u);
    (void)hipEventRecord(e0);
    kernel<<<768, 512>>>(out, iters, 0x90u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-34s %.3f ms  (%d tiles of 16 wave-beams per wave, 6 waves per SIMD): %.1f ns per tile per SIMD\n", name, ms, iters, ms * 1e6 / iters / 6);
  };
  time(k_overlap<0>, "vector mix alone");
  time(k_overlap<1>, "4 MFMA 32x32x2 f32 per tile alone");
  time(k_overlap<2>, "both, interleaved in every wave");
  return 0;
}
