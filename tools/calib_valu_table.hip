// Issue cost table of gfx950 vector instructions (the ones a likelihood-field look-up can be built from), same harness as
// calib_f64_rate.hip: 8 waves per SIMD, 8 independent chains per lane, cycles quoted at the nominal 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib_valu_table.hip -o /tmp/calib_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
#define FOR_EACH(X)                                                                                                    \
  X(0, "v_fma_f64 %0, %0, %1, %2", "+v"(v[k]) : "v"(a), "v"(b))                                                       \
  X(1, "v_add_f64 %0, %0, %1", "+v"(v[k]) : "v"(b))                                                                   \
  X(2, "v_mul_f64 %0, %0, %1", "+v"(v[k]) : "v"(a))                                                                   \
  X(3, "v_fma_f32 %0, %0, %1, %2", "+v"(f.x) : "v"(va.x), "v"(va.y))                                                  \
  X(4, "v_fma_f32 %0, %0, %1, %2", "+v"(f.x) : "s"(sa.x), "v"(va.y))                                                  \
  X(5, "v_fmac_f32 %0, %1, %2", "+v"(f.x) : "v"(va.x), "v"(va.y))                                                     \
  X(6, "v_mul_f32 %0, %0, %1", "+v"(f.x) : "v"(va.x))                                                                 \
  X(7, "v_add_f32 %0, %0, %1", "+v"(f.x) : "v"(va.y))                                                                 \
  X(8, "v_pk_fma_f32 %0, %0, %1, %2", "+v"(f) : "v"(va), "v"(va))                                                     \
  X(9, "v_pk_add_f32 %0, %0, %1", "+v"(f) : "s"(sa))                                                                  \
  X(10, "v_add_u32 %0, %0, %1", "+v"(u) : "v"(i))                                                                     \
  X(11, "v_lshlrev_b32 %0, 1, %0", "+v"(u) :)                                                                         \
  X(12, "v_lshrrev_b32 %0, 16, %0", "+v"(u) :)                                                                        \
  X(13, "v_and_b32 %0, %1, %0", "+v"(u) : "s"(sel))                                                                   \
  X(14, "v_or_b32 %0, %0, %1", "+v"(u) : "v"(i))                                                                      \
  X(15, "v_min_u32 %0, %0, %1", "+v"(u) : "v"(i))                                                                     \
  X(16, "v_mul_u32_u24 %0, %0, %1", "+v"(u) : "v"(i))                                                                 \
  X(17, "v_lshl_add_u32 %0, %0, 1, %1", "+v"(u) : "v"(i))                                                             \
  X(18, "v_add_lshl_u32 %0, %0, %1, 1", "+v"(u) : "v"(i))                                                             \
  X(19, "v_lshl_or_b32 %0, %0, 1, %1", "+v"(u) : "v"(i))                                                              \
  X(20, "v_and_or_b32 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                    \
  X(21, "v_add3_u32 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                      \
  X(22, "v_mad_u32_u24 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                   \
  X(23, "v_mad_i32_i24 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                   \
  X(24, "v_mad_u32_u16 %0, %0, %1, %2 op_sel:[1,0,0,0]", "+v"(u) : "s"(sel), "v"(i))                                  \
  X(25, "v_bfe_u32 %0, %0, 16, 6", "+v"(u) :)                                                                         \
  X(26, "v_perm_b32 %0, %0, %1, %2", "+v"(u) : "v"(i), "s"(sel))                                                      \
  X(27, "v_dot4_u32_u8 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                   \
  X(28, "v_min3_u32 %0, %0, %1, %2", "+v"(u) : "v"(i), "v"(i))                                                        \
  X(29, "v_med3_i32 %0, %0, %1, %2", "+v"(u) : "v"(i), "s"(sel))                                                      \
  X(30, "v_pk_min_u16 %0, %0, %1", "+v"(u) : "v"(i))                                                                  \
  X(31, "v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "+v"(u) : "v"(i)) \
  X(32, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2", "+v"(u) : "v"(i)) \
  X(33, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD", "+v"(u) : "v"(i)) \
  X(34, "v_mov_b32 %0, %1", "+v"(u) : "v"(i))                                                                         \
  X(35, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(u) : "v"(i) : "vcc")                                                    \
  X(36, "v_cvt_f32_f64 %0, %1", "=v"(f.x) : "v"(v[k]))                                                                \
  X(37, "v_cvt_f64_f32 %0, %1", "+v"(v[k]) : "v"(va.x))                                                               \
  X(38, "v_cvt_u32_f32 %0, %0", "+v"(u) :)                                                                            \
  X(39, "v_floor_f32 %0, %0", "+v"(f.x) :)                                                                            \
  X(40, "v_fract_f32 %0, %0", "+v"(f.x) :)                                                                            \
  X(41, "v_alignbit_b32 %0, %0, %0, 16", "+v"(u) :)                                                                   \
  X(42, "v_mul_lo_u32 %0, %0, %1", "+v"(u) : "v"(i))                                                                  \
  X(43, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "+v"(v[k]) : "v"(i), "s"(sel) : "vcc")                                  \
  X(44, "v_xor_b32 %0, %0, %1", "+v"(u) : "v"(i))                                                                     \
  X(45, "v_sub_u32 %0, %0, %1", "+v"(u) : "v"(i))                                                                     \
  X(46, "v_max_f32 %0, %0, %1", "+v"(f.x) : "v"(va.x))                                                                \
  X(47, "v_min3_f32 %0, %0, %1, %2", "+v"(f.x) : "v"(va.x), "v"(va.y))                                                \
  X(48, "v_fma_mix_f32 %0, %0, %1, %2", "+v"(f.x) : "v"(va.x), "v"(va.y))                                             \
  X(49, "v_pk_mul_f32 %0, %0, %1", "+v"(f) : "v"(va))                                                                 \
  X(50, "v_add_co_u32 %0, vcc, %0, %1", "+v"(u) : "v"(i) : "vcc")                                                     \
  X(51, "v_cmp_lt_u32 vcc, %0, %1", : "v"(u), "v"(i) : "vcc")                                                         \
  X(52, "v_xad_u32 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                       \
  X(53, "v_sad_u32 %0, %0, %1, %2", "+v"(u) : "s"(sel), "v"(i))                                                       \
  X(54, "v_max3_u32 %0, %0, %1, %2", "+v"(u) : "v"(i), "v"(i))                                                        \
  X(55, "v_bfi_b32 %0, %1, %0, %2", "+v"(u) : "s"(sel), "v"(i))                                                       \
  X(56, "v_fma_f64 %0, %1, %2, %0", "+v"(v[k]) : "s"(a), "v"(b))
#define MAX_MODE 57

template <int kMode>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, float2_t sa, unsigned sel, int iters) {
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3 + k;
  float2_t va = {static_cast<float>(a), static_cast<float>(b)};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      [[maybe_unused]] float2_t& f = reinterpret_cast<float2_t&>(v[k]);
      [[maybe_unused]] unsigned& u = reinterpret_cast<unsigned&>(v[k]);
#define X(mode, text, ...) if (kMode == mode) asm volatile(text : __VA_ARGS__);
      FOR_EACH(X)
#undef X
    }
  }
  double s = 0;
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
static const char* names[MAX_MODE];
template <int kMode>
void run(double* d, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const float2_t sa = {1.0000001f, 1e-9f};
  hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, sa, 0x05010400u, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<kMode>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, sa, 0x05010400u, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr = double(blocks) * 4 * iters * 8;
  printf("%-100s %.3f ms %6.2f cycles\n", names[kMode], ms, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
  if constexpr (kMode + 1 < MAX_MODE) run<kMode + 1>(d, blocks, iters);
}
int main() {
#define X(mode, text, ...) names[mode] = text;
  FOR_EACH(X)
#undef X
  double* d; (void)hipMalloc(&d, 256 * 8192 * 8);
  run<0>(d, 256 * 8 * 2, 2048);
  return 0;
}
