// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the MCL update cycle.
//
// No MFMA anywhere: the cycle is gather + streaming work.  What matters here is
// coalesced SoA access, keeping the scan in LDS / SGPRs, enough waves in flight to hide gather
// latency, and deterministic (fixed-order) f64 reductions so runs are reproducible.
//
// Built with -ffp-contract=off: the reference (and the oracle) evaluate `p*cos - q*sin + t` with
// separate roundings (likelihood_field_model.hpp:82-83); a contracted FMA would move a handful of
// beam end-points across a cell boundary of `floor(x / resolution)` (regular_grid.hpp:75-78).
#include "device_common.hpp"


namespace mcl {
namespace {

// ---- spatial ordering key ----------------------------------------------------------------------------
// 20 bits: x, y in 64 bins each, heading in 256 bins over the key frame's span; the two top heading bits first, the
// remaining 6 + 6 + 6 bits Morton-interleaved (heading, y, x).  Only locality depends on the key, never a result, so it
// is evaluated in single precision.
// KeyFrame::layout = 1 (sets reported as dispersed): 12 bits of (y, x) Morton-interleaved first, the 8 heading bits last.  Poses
// metres apart share no cache line whatever their headings, but the end-points of a REGION's poses stay within max range of it:
// with the order position-major and each XCD walking one contiguous eighth of it (k_reweight_lf_palette<true, true>), the part
// of the table an XCD's L2 has to hold at any time is the neighbourhood of a block of the map instead of all of it.
constexpr uint32_t kKeyBitsXY = 6, kKeyBitsTheta = 8, kKeyBits = 2 * kKeyBitsXY + kKeyBitsTheta;
constexpr uint32_t kDigitBits = 10;
static_assert((1u << kDigitBits) == kSortDigits && kKeyBits == 2 * kDigitBits, "two passes of one digit each");
__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // ...edcba -> ..e00d00c00b00a
  v &= 0x3FF;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}
__device__ __forceinline__ uint32_t spread2(uint32_t v) {  // ..fedcba -> .f0e0d0c0b0a
  v &= 0x3F;
  v = (v | (v << 4)) & 0x30F;
  v = (v | (v << 2)) & 0x333;
  v = (v | (v << 1)) & 0x555;
  return v;
}
__device__ __forceinline__ int unit_bin(float u, int bins) {  // u in [0, 1) inside the span; clamped outside (NaN -> 0)
  const int b = static_cast<int>(u * static_cast<float>(bins));
  return min(max(b, 0), bins - 1);
}
// atan2 to ~1e-4 rad, monotone in the angle (a cubic in min / max of |x|, |y| - Rajan et al.'s approximation - folded back into
// the octants): the ordering key needs the BIN of a heading, and nothing but locality depends on which side of a bin's edge a
// particle falls.  A dozen instructions instead of libm's forty-odd.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
  const float t = hi > 0.f ? lo * __builtin_amdgcn_rcpf(hi) : 0.f;
  const float t2 = t * t;
  float a = t * (0.99997726f + t2 * (-0.33262347f + t2 * (0.19354346f + t2 * (-0.11643287f + t2 * (0.05265332f + t2 * -0.01172120f)))));
  a = ay > ax ? 1.57079633f - a : a;
  a = x < 0.f ? 3.14159265f - a : a;
  return y < 0.f ? -a : a;
}
__device__ __forceinline__ uint32_t order_key(const double4& q, const KeyFrame& kf) {
  const float ux = static_cast<float>(q.z - kf.cx) * kf.inv_x + 0.5f;
  const float uy = static_cast<float>(q.w - kf.cy) * kf.inv_y + 0.5f;
  const float c = static_cast<float>(q.x), s = static_cast<float>(q.y);
  const float c0 = static_cast<float>(kf.c0), s0 = static_cast<float>(kf.s0);
  const float delta = fast_atan2f(s * c0 - c * s0, c * c0 + s * s0);  // heading relative to the frame's, in (-pi, pi]
  const float ut = (delta - kf.t_off) * kf.inv_t + 0.5f;
  if (kf.layout & 1u) {
    const uint32_t bx = static_cast<uint32_t>(unit_bin(ux, 1 << kKeyBitsXY)), by = static_cast<uint32_t>(unit_bin(uy, 1 << kKeyBitsXY));
    const uint32_t bt = static_cast<uint32_t>(unit_bin(ut, 1 << kKeyBitsTheta));
    return ((spread2(bx) | (spread2(by) << 1)) << kKeyBitsTheta) | bt;
  }
  // heading-major: the top heading bits select a slab, inside it the curve runs through cubes of bits_xy bits per axis
  const uint32_t bits = kf.bits_xy ? kf.bits_xy : kKeyBitsXY, bits_t = kKeyBits - 2 * bits;
  float vx = ux, vy = uy, vt = ut;
  if (kf.layout & 4u) {
    // Bins of equal MASS instead of equal width: the frame's span is +-4 sigma of a set that is close to normal, so u -> the
    // normal distribution function of 8 (u - 1/2).  The 1024 buckets of the ordering's first pass (the key's high digit) then
    // hold about the same number of particles - with bins of equal width the bucket at the centre of the cloud held ten times the
    // average, and its workgroup was the second pass's critical path - and the cells are small where the particles are.
    // (the logistic curve 1 / (1 + exp(-1.702 z)) is within 0.01 of the normal distribution function: a bin's mass is equal to 1 %)
    auto mass = [](float u) { return __builtin_amdgcn_rcpf(1.f + __expf(-13.616f * (u - 0.5f))); };  // z = 8 (u - 1/2)
    vx = mass(ux);
    vy = mass(uy);
    vt = mass(ut);
  }
  const uint32_t bx = static_cast<uint32_t>(unit_bin(vx, 1 << bits)), by = static_cast<uint32_t>(unit_bin(vy, 1 << bits));
  const uint32_t bt = static_cast<uint32_t>(unit_bin(vt, 1 << bits_t));
  const uint32_t slab = (bt >> bits) << (3 * bits), bt_in = bt & ((1u << bits) - 1);
  if (kf.layout & 2u) return slab | spread3(bx) | (spread3(by) << 1) | (spread3(bt_in) << 2);
  return slab | hilbert_index_3(bt_in, by, bx, bits);
}

// The cycle's scan, pulled from mapped pinned host memory by one workgroup (17 KB at 1080 beams): an asynchronous
// host-to-device copy on the stream costs a copy-engine hand-off of ~20 us between two kernels, this costs nothing.
__device__ __forceinline__ void pull_scan(const double* __restrict__ src, double* __restrict__ dst, uint32_t doubles) {
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const uint32_t pairs = doubles / 2;
  // (four loads in flight per thread - the reads cross PCIe -, in four named registers with clamped addresses: as an array with
  // conditional elements it was a private object of 64 bytes per thread, which the compiler promoted to workgroup memory - 64 KB per
  // workgroup of k_propagate, for a copy its last workgroup alone makes)
  const uint32_t stride = blockDim.x, last = pairs ? pairs - 1 : 0;
  for (uint32_t j = threadIdx.x; j < pairs; j += 4 * stride) {
    const uint32_t j1 = j + stride, j2 = j + 2 * stride, j3 = j + 3 * stride;
    const double2 a = s2[j], b = s2[j1 < last ? j1 : last], c = s2[j2 < last ? j2 : last], d = s2[j3 < last ? j3 : last];
    d2[j] = a;
    if (j1 < pairs) d2[j1] = b;
    if (j2 < pairs) d2[j2] = c;
    if (j3 < pairs) d2[j3] = d;
  }
  if ((doubles & 1u) && threadIdx.x == 0) dst[doubles - 1] = src[doubles - 1];
}
__global__ __launch_bounds__(kBlock) void k_pull_scan(const double* __restrict__ src, double* __restrict__ dst, uint32_t doubles) {
  pull_scan(src, dst, doubles);
}

// ---- K1 propagate ----------------------------------------------------------------------------------
// One workgroup per chunk of kChunk particles (coalesced 32-byte records).  kKeys: the ordering key of the NEW pose and the
// chunk's histogram of the key's low digit come out of the same pass (the poses are in registers here).
// One particle through the motion model (the body of actions::propagate for the three models).
// The propagation's own forms of the shared helpers (se2.h, rng.h: library sin / cos / hypot and two divisions per
// normalisation - 1150 vector instructions per particle, four fifths of them in eight trigonometric calls and six normalisations).
// Same expressions, evaluated another way; each within 2 ulp of the shared form (the parity tests' tolerance for a propagated
// state is 1e-11 relative):
//  * sin and cos of one argument together: Cody-Waite reduction by pi/2 in two fused steps (the product k * pi/2_hi is exact
//    inside the FMA; |theta| < 1e6, anything else - NaN included - goes to the library), fdlibm's kernel polynomials on
//    [-pi/4, pi/4] (k_sin.c, k_cos.c: 1.1e-16 / 1.4e-16 absolute against long double, checked over 2M points);
//  * z / |z| as z * rsqrt(|z|^2): v_rsq_f64 and two Newton steps instead of hypot and two divisions.
// (out of line: inlined at its four call sites per particle, the library's argument reduction brought 64 bytes of workgroup memory per
// thread - its private arrays, promoted - and its registers into a path that an angle beyond a million radians alone takes)
__device__ __attribute__((noinline)) double2 sincos_library(double theta) { return double2{sin(theta), cos(theta)}; }
__device__ __forceinline__ void sincos_fast(double theta, double& s, double& c) {
  if (!(fabs(theta) < 1.0e6)) {
    const double2 sc = sincos_library(theta);
    s = sc.x;
    c = sc.y;
    return;
  }
  const double kd = __builtin_rint(theta * 0x1.45f306dc9c883p-1);  // theta * 2 / pi
  const int k = static_cast<int>(kd);
  double r = __builtin_fma(-kd, 0x1.921fb54442d18p+0, theta);  // pi / 2 = hi + lo
  r = __builtin_fma(-kd, 0x1.1a62633145c07p-54, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10, pc = -1.13596475577881948265e-11;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  const double sr = __builtin_fma(r * z, ps, r);
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  // theta = r + k pi/2: (sin, cos) = (sr, cr), (cr, -sr), (-sr, -cr), (-cr, sr) for k mod 4 = 0 .. 3
  const double s0 = (k & 1) ? cr : sr, c0 = (k & 1) ? sr : cr;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ Rot2 rot_from_complex_fast(double re, double im) {  // se2.h rot_from_complex: z / hypot(z)
  const double n2 = __builtin_fma(re, re, im * im);
  double y = __builtin_amdgcn_rsq(n2);
#pragma unroll
  for (int it = 0; it < 2; ++it) {  // y <- y + y (1/2 - n2 y^2 / 2)
    const double e = __builtin_fma(-n2 * y, 0.5 * y, 0.5);
    y = __builtin_fma(y, e, y);
  }
  return Rot2{re * y, im * y};  // (|z| = 0: 0 * inf = NaN, as 0 / 0 there)
}
__device__ __forceinline__ Rot2 rot_exp_fast(double theta) {
  double sn, cs;
  sincos_fast(theta, sn, cs);
  return rot_from_complex_fast(cs, sn);
}
__device__ __forceinline__ Rot2 rot_mul_fast(const Rot2& a, const Rot2& b) {  // se2.h rot_mul
  double re = a.c * b.c - a.s * b.s;
  double im = a.c * b.s + a.s * b.c;
  const double n2 = re * re + im * im;
  if (n2 != 1.0) {
    const double scale = 2.0 / (1.0 + n2);
    re = re * scale;
    im = im * scale;
  }
  return rot_from_complex_fast(re, im);
}
__device__ __forceinline__ Pose2 pose_mul_fast(const Pose2& a, const Pose2& b) {  // se2.h pose_mul
  Pose2 o;
  o.r = rot_mul_fast(a.r, b.r);
  double tx, ty;
  rot_apply(a.r, b.x, b.y, tx, ty);
  o.x = a.x + tx;
  o.y = a.y + ty;
  return o;
}
__device__ __forceinline__ void box_muller_fast(double u1, double u2, double& z0, double& z1) {  // rng.h rng_box_muller
  const double r = sqrt(-2.0 * log(1.0 - u1));
  double sn, cs;
  sincos_fast(2.0 * kPi * u2, sn, cs);
  z0 = r * cs;
  z1 = r * sn;
}
// The four standard normals of particle `index` at `step`: the two Philox draws and Box-Muller pairs of the propagation (rng.h) - a function of
// (seed, step, index) alone, which is why they can be drawn AHEAD of the control action they will be scaled by (k_noise_ahead).
__device__ __forceinline__ double4 propagation_normals(uint64_t seed, uint32_t step, uint64_t index) {
  const RngWords a = rng_draw(seed, step, kRngPropagateA, index);
  const RngWords b = rng_draw(seed, step, kRngPropagateB, index);
  double z0, z1, z2, z3;
  box_muller_fast(rng_uniform53(a.w[0], a.w[1]), rng_uniform53(a.w[2], a.w[3]), z0, z1);
  box_muller_fast(rng_uniform53(b.w[0], b.w[1]), rng_uniform53(b.w[2], b.w[3]), z2, z3);
  return double4{z0, z1, z2, z3};
}
// One particle through the motion model, given its three standard normals (the body of actions::propagate for the three models).
__device__ __forceinline__ Pose2 propagate_with_normals(const Pose2& state, const DiffDriveSampler& smp, double z0, double z1, double z2) {
  if (smp.kind == 1) {
    // omnidirectional_drive_model.hpp:133-144 — draws in source order: rotation, translation, strafe
    const Rot2 first{smp.first_c, smp.first_s};
    const Rot2 second = rot_mul_fast(rot_exp_fast(z0 * smp.s1 + smp.m1), rot_inverse(first));
    const double t = z1 * smp.st + smp.mt;
    const double strafe = z2 * smp.s2 + 0.0;
    return pose_mul_fast(pose_mul_fast(state, Pose2{first, 0.0, 0.0}), Pose2{second, t, -strafe});
  }
  if (smp.kind == 2) {
    // stationary_model.hpp:55-61 — N(0, 0.02) on heading, x, y
    return pose_mul_fast(state, Pose2{rot_exp_fast(z0 * 0.02 + 0.0), z1 * 0.02 + 0.0, z2 * 0.02 + 0.0});
  }
  // differential_drive_model.hpp:156-163
  const double r1 = z0 * smp.s1 + smp.m1;
  const double t = z1 * smp.st + smp.mt;
  const double r2 = z2 * smp.s2 + smp.m2;
  return pose_mul_fast(pose_mul_fast(state, Pose2{rot_exp_fast(r1), 0.0, 0.0}), Pose2{rot_exp_fast(r2), t, 0.0});
}

// normals_ahead: the first three normals of every particle, drawn a cycle ahead (k_resample_draw / k_noise_ahead) - three arrays of `stride`
// doubles (the fourth is used by no motion model) -; the same expressions: the same bits
__device__ __forceinline__ Pose2 propagate_one(const Pose2& state, const DiffDriveSampler& smp, uint64_t seed, uint32_t step, uint64_t index,
                                               const double* __restrict__ normals_ahead = nullptr, uint64_t local = 0, uint64_t stride = 0) {
  double z0, z1, z2;
  if (normals_ahead) {  // (uniform)
    z0 = normals_ahead[local];
    z1 = normals_ahead[stride + local];
    z2 = normals_ahead[2 * stride + local];
  } else {
    const double4 z = propagation_normals(seed, step, index);
    z0 = z.x;
    z1 = z.y;
    z2 = z.z;
  }
  return propagate_with_normals(state, smp, z0, z1, z2);
}

// Small sets (no ordering keys): one particle per lane, 256 per workgroup - 2000 particles are 8 workgroups on 8 CUs, a wave
// per SIMD, instead of one workgroup of the chunked kernel working through them on one (256 lanes rather than 64 for the sake
// of the scan pull, which is bound by the reads in flight over PCIe).
__global__ __launch_bounds__(kBlock) void k_propagate_small(Particles p, uint64_t n, DiffDriveSampler smp, uint64_t seed, uint32_t step,
                                                           uint64_t index_offset, const double* __restrict__ scan_src,
                                                           double* __restrict__ scan_dst, uint32_t scan_doubles) {
  if (scan_dst && blockIdx.x == gridDim.x - 1) pull_scan(scan_src, scan_dst, scan_doubles);
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) store_pose(p, i, propagate_one(load_pose(p, i), smp, seed, step, index_offset + i));
}

// 512 threads, four particles each: at the kernel's 100 registers a CU holds 16 waves - one workgroup of 1024 threads (round 4: 489
// workgroups at 1M particles on 256 slots, two rounds of which the second is nine tenths full) or two of 512 (489 on 512 slots: one round).
constexpr int kPropBlock = 512;
template <bool kKeys>
__global__ __launch_bounds__(kPropBlock) void k_propagate(Particles p, uint64_t n, DiffDriveSampler smp, uint64_t seed, uint32_t step,
                                                      uint64_t index_offset, const double* __restrict__ scan_src,
                                                      double* __restrict__ scan_dst, uint32_t scan_doubles, KeyFrame kf,
                                                      uint32_t* __restrict__ keys, uint32_t* __restrict__ table, uint32_t nblocks,
                                                      const double* __restrict__ normals_ahead, uint64_t normals_stride) {
  __shared__ uint32_t hist[kKeys ? kSortDigits : 1];
  if (kKeys) {
    for (uint32_t d = threadIdx.x; d < kSortDigits; d += kPropBlock) hist[d] = 0;
    __syncthreads();
  }
  if (scan_dst && blockIdx.x == gridDim.x - 1) pull_scan(scan_src, scan_dst, scan_doubles);
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll 1
  for (int k = 0; k < kChunk / kPropBlock; ++k) {
    const uint64_t i = base + static_cast<uint64_t>(k) * kPropBlock + threadIdx.x;
    if (i >= n) break;
    const Pose2 out = propagate_one(load_pose(p, i), smp, seed, step, index_offset + i, normals_ahead, i, normals_stride);
    store_pose(p, i, out);
    if (kKeys) {
      const uint32_t key = order_key(double4{out.r.c, out.r.s, out.x, out.y}, kf);
      keys[i] = key;
      atomicAdd(&hist[key >> kDigitBits], 1u);  // the ordering's first pass goes by the HIGH digit
    }
  }
  if (kKeys) {
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < kSortDigits; d += kPropBlock) table[static_cast<size_t>(d) * nblocks + blockIdx.x] = hist[d];
  }
}

// The propagation's random numbers, a cycle AHEAD: they depend on (seed, step, particle index) alone, not on the control action that scales them
// (differential_drive_model.hpp:156-163 draws standard normals and scales them afterwards; so do the other two models) nor on the states.
// Launched behind the last kernel of a fixed-size cycle whose end the host takes from the completion word (cycle_spin): it runs while the host
// returns the estimate and comes back with the next control action - the 20 us the device used to idle between two cycles -, and the next
// k_propagate finds 60 % of its instructions (two Philox draws, two logarithms, square roots and sine / cosine pairs per particle) done.
// Same expressions as k_propagate's own: the same bits.
__global__ __launch_bounds__(kBlock) void k_noise_ahead(uint64_t seed, uint32_t step, uint64_t index_offset, uint64_t n, double* __restrict__ out) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const double4 z = propagation_normals(seed, step, index_offset + i);
  out[i] = z.x;
  out[n + i] = z.y;
  out[2 * n + i] = z.z;
}

// [lf-kernels-begin] (the HBM-traffic record of the LF kernel, profiles/lf_kernel_traffic.json, is keyed by the SHA-256 of
// the source from here to [lf-kernels-end])
// ---- K2 likelihood-field reweight ------------------------------------------------------------------
// One beam end-point -> pz^3.  likelihood_field_model.hpp:82-88, dense_grid.hpp:92-96,127-129,
// regular_grid.hpp:75-78, linear_grid.hpp:73-75.
__device__ __forceinline__ double lf_beam(const FieldView& f, double px, double py, double ct, double st, double xt, double yt) {
  const double x = px * ct - py * st + xt;
  const double y = px * st + py * ct + yt;
  const int xi = static_cast<int>(floor(x * f.inv_resolution));
  const int yi = static_cast<int>(floor(y * f.inv_resolution));
  // Branch-free: out-of-grid lanes read cell 0 and discard it, so the gathers of an unrolled group
  // of beams can all be in flight together instead of sitting behind one exec-mask branch each.
  const bool inside = static_cast<unsigned>(xi) < f.W && static_cast<unsigned>(yi) < f.H;
  const size_t idx = inside ? static_cast<size_t>(yi) * f.W + static_cast<size_t>(xi) : size_t{0};
  float v = f.data[idx];
  v = inside ? v : f.unknown_value;
  const double pz = static_cast<double>(v);
  return f.prob ? log(pz) : pz * pz * pz;
}

// floor() of 2*G doubles at once without v_floor_f64 + v_cvt_i32_f64: with the f64 rounding mode switched to
// round-toward-minus-infinity, v + 1.5*2^52 is exactly 1.5*2^52 + floor(v), whose low mantissa word is floor(v) as a
// two's complement int32 (|v| < 2^31; larger magnitudes are far outside any grid and undefined upstream as well).
// One VALU op per value instead of two.  The mode switch and the adds sit in ONE asm statement so that no other
// floating-point instruction can be scheduled into the round-down window.
constexpr double kFloorMagic = 6755399441055744.0;  // 1.5 * 2^52
#define MCL_RD_BEGIN "s_nop 1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"
#define MCL_RD_END "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0"
__device__ __forceinline__ void floor_rd_16(double (&v)[16]) {
  asm volatile(MCL_RD_BEGIN
               "v_add_f64 %0, %0, %16\n\tv_add_f64 %1, %1, %16\n\tv_add_f64 %2, %2, %16\n\tv_add_f64 %3, %3, %16\n\t"
               "v_add_f64 %4, %4, %16\n\tv_add_f64 %5, %5, %16\n\tv_add_f64 %6, %6, %16\n\tv_add_f64 %7, %7, %16\n\t"
               "v_add_f64 %8, %8, %16\n\tv_add_f64 %9, %9, %16\n\tv_add_f64 %10, %10, %16\n\tv_add_f64 %11, %11, %16\n\t"
               "v_add_f64 %12, %12, %16\n\tv_add_f64 %13, %13, %16\n\tv_add_f64 %14, %14, %16\n\tv_add_f64 %15, %15, %16\n\t"
               MCL_RD_END
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                 "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               : "s"(kFloorMagic));
}
__device__ __forceinline__ void floor_rd_8(double (&v)[8]) {
  asm volatile(MCL_RD_BEGIN
               "v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
               "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8\n\t"
               MCL_RD_END
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               : "s"(kFloorMagic));
}
__device__ __forceinline__ void floor_rd_2(double& a, double& b) {
  asm volatile(MCL_RD_BEGIN "v_add_f64 %0, %0, %2\n\tv_add_f64 %1, %1, %2\n\t" MCL_RD_END : "+v"(a), "+v"(b) : "s"(kFloorMagic));
}
__device__ __forceinline__ int floor_rd_result(double v) { return static_cast<int>(static_cast<uint32_t>(__builtin_bit_cast(uint64_t, v))); }

__device__ __forceinline__ double lf_cube_fetch(__amdgpu_buffer_rsrc_t rsrc, const FieldView& f, uint32_t row_bytes,
                                                uint32_t unknown_offset, int xi, int yi) {
  const bool inside = static_cast<unsigned>(xi) < f.W && static_cast<unsigned>(yi) < f.H;
  const uint32_t offset = inside ? __umul24(static_cast<unsigned>(yi), row_bytes) + (static_cast<unsigned>(xi) << 3) : unknown_offset;
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, offset, 0, 0));
}

// The fallback family (fields with too many distinct values for a palette - the cube table or the f32 field itself -, sets without
// an order): one lane per particle, every lane walks the scan in order, the scan read with scalar loads, the sum `1 + sum pz^3`
// added in the order of the reference's std::transform_reduce (sum4).  With `perm` the lanes are neighbours of the spatial order:
// for a given beam their end-points fall into a handful of 128-byte lines, the vector L1 sees ~10 tag look-ups per gather instead
// of 64 (profiles/r01: unordered lanes are bound by the L1/L2 request rate, not by HBM).  The order in which particles are
// visited does not change any result.
// `partial` == nullptr: the whole scan per lane, weights updated in place (the sum in the reference's order: sum4).
// `partial` != nullptr (medium particle counts, where one lane per particle cannot fill 256 CUs): blockIdx.y selects a
// contiguous segment of the scan; the segment's sum goes to partial[segment][t] and k_lf_combine adds the segments up in
// order — same terms, fixed association, bit-reproducible, differs from the whole-scan sum only in rounding.
// The pose of the particle at a position of the spatial order, moved into the table's frame.  A 32-byte record gather
// per lane, once per kernel (the ordering passes move 8 bytes per particle, not the poses).
__device__ __forceinline__ Pose2 ordered_pose(const Pose2& to_frame, const double4* __restrict__ pose, uint64_t i) {
  const double4 q = pose[i];
  return pose_mul(to_frame, Pose2{Rot2{q.x, q.y}, q.z, q.w});
}

template <bool kCube>
__global__ __launch_bounds__(kBlock) void k_reweight_lf_sorted(double* __restrict__ w, uint64_t n, FieldView f,
                                                               const double* __restrict__ pts, uint32_t B,
                                                               const uint32_t* __restrict__ perm, const double4* __restrict__ pose,
                                                               double* __restrict__ partial, uint32_t beams_per_segment) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  const uint64_t tt = t < n ? t : n - 1;
  const uint64_t i = perm ? perm[tt] : tt;  // (no order: small sets without a palette, sets beyond 2^32 particles)
  const Pose2 T = ordered_pose(f.world_to_field, pose, i);  // likelihood_field_model.hpp:70
  const double ct = T.r.c, st = T.r.s, xt = T.x, yt = T.y;
  const uint32_t b_begin = partial ? blockIdx.y * beams_per_segment : 0u;
  const uint32_t b_end = partial ? (b_begin + beams_per_segment < B ? b_begin + beams_per_segment : B) : B;
  double acc = (f.prob || partial) ? 0.0 : 1.0;
  if (kCube) {
    const uint32_t cells = f.W * f.H;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(f.cube), 0, static_cast<int>((cells + 1) * 8u), 0x00020000);
    const uint32_t row_bytes = f.W * 8u, unknown_offset = cells * 8u;
    uint32_t b = b_begin;
    for (; b + 8 <= b_end; b += 8) {
      double v[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double px = pts[2 * (b + k)], py = pts[2 * (b + k) + 1];
        v[2 * k] = (px * ct - py * st + xt) * f.inv_resolution;
        v[2 * k + 1] = (px * st + py * ct + yt) * f.inv_resolution;
      }
      floor_rd_16(v);
      double t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        t[k] = lf_cube_fetch(rsrc, f, row_bytes, unknown_offset, floor_rd_result(v[2 * k]), floor_rd_result(v[2 * k + 1]));
      acc += sum4(t[0], t[1], t[2], t[3]);
      acc += sum4(t[4], t[5], t[6], t[7]);
    }
    auto term = [&](uint32_t at) {
      const double px = pts[2 * at], py = pts[2 * at + 1];
      double vx = (px * ct - py * st + xt) * f.inv_resolution, vy = (px * st + py * ct + yt) * f.inv_resolution;
      floor_rd_2(vx, vy);
      return lf_cube_fetch(rsrc, f, row_bytes, unknown_offset, floor_rd_result(vx), floor_rd_result(vy));
    };
    if (b + 4 <= b_end) {
      const double t0 = term(b), t1 = term(b + 1), t2 = term(b + 2), t3 = term(b + 3);
      acc += sum4(t0, t1, t2, t3);
      b += 4;
    }
    for (; b < b_end; ++b) acc += term(b);
  } else {
    uint32_t b = b_begin;
#pragma unroll 2
    for (; b + 4 <= b_end; b += 4) {
      double t[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = lf_beam(f, pts[2 * (b + k)], pts[2 * (b + k) + 1], ct, st, xt, yt);
      acc += sum4(t[0], t[1], t[2], t[3]);
    }
    for (; b < b_end; ++b) acc += lf_beam(f, pts[2 * b], pts[2 * b + 1], ct, st, xt, yt);
  }
  if (t < n) {
    if (partial) {
      partial[static_cast<size_t>(blockIdx.y) * n + t] = acc;
    } else {
      w[i] = w[i] * (f.prob ? exp(acc) : acc);
    }
  }
}

// Variant C over the palette form of the table (FieldView::pal_*): the gather fetches a 2-byte LDS address from the
// 8x8-tiled, bordered index table and the exact f64 term comes from the palette copy in LDS.  Same arithmetic, same
// order, same bits as the reference's loop; per beam and lane: 8 f64 ops for the end-point, 2 to scale to cells, 2 for the
// floors, 2 clamps, 2 for the table offset (the row part comes from a row-offset table in LDS), 1 add.
// Workgroup memory: [0, (H+2)*4) row offsets for y = -1 .. H, [pal_base, pal_base + 8 * pal_count) the palette; the kernel
// has no other LDS, so these are absolute LDS addresses.
constexpr int kPalBlock = 512;
typedef __attribute__((address_space(3))) const double lds_f64_t;
typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) const uint8_t lds_u8_t;
__device__ __forceinline__ int clamp_cell(int v, uint32_t hi) {  // max(-1, min(v, hi)) in one instruction
  int r;
  asm("v_med3_i32 %0, %1, -1, %2" : "=v"(r) : "v"(v), "s"(hi));
  return r;
}
// The fast variant of the kernel (kFast, below) works on cell coordinates biased by kFastBias (the high word of
// v + 1.5 * 2^20); there the row-offset table in LDS is stored with the x part of that bias already subtracted and the
// exact evaluation adds it back (row_fix = kFastBiasX; 0 in the plain kernel).
constexpr uint32_t kFastBias = 0x41380000u;      // high word of the double 1.5 * 2^20
constexpr uint32_t kFastBiasX = kFastBias << 4;  // (mod 2^32) what a biased x contributes to the byte offset
constexpr double kFastMagic = 1572864.0;         // 1.5 * 2^20: v + kFastMagic has 32 fraction bits in its low word
__device__ __forceinline__ uint32_t lf_palette_offset(const FieldView& f, int xi, int yi, uint32_t row_fix) {
  const int xc = clamp_cell(xi, f.W), yc = clamp_cell(yi, f.H);
  const uint32_t row = *reinterpret_cast<lds_u32_t*>(static_cast<uintptr_t>(static_cast<uint32_t>(yc + 1) << 2));
  return (static_cast<uint32_t>(xc) << 4) + row_fix + row;
}
__device__ __forceinline__ int med3_i32(int v, int lo /* in a VGPR: one scalar operand per instruction */, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "s"(hi));
  return r;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b /* uniform */, uint32_t c) {  // (a mod 2^24) * (b mod 2^24) + c
  uint32_t r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
  return r;
}
// cx * pitch + cy * 2 + K (K uniform) in two instructions.  ONE asm statement: between two of them the compiler's hazard recogniser, which
// cannot look inside, puts an s_nop (8 per group of beams in the LF patch kernel's main loop).
__device__ __forceinline__ uint32_t patch_address(uint32_t cx, uint32_t cy, uint32_t K /* uniform */) {
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, 1, %3\n\tv_mad_u32_u24 %0, %2, %4, %0" : "=&v"(r) : "v"(cy), "v"(cx), "s"(K), "s"(144u));
  return r;
}
__device__ __forceinline__ uint32_t min3_u32(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ uint32_t lf_palette_fetch(__amdgpu_buffer_rsrc_t rsrc, const FieldView& f, int xi, int yi, uint32_t row_fix) {
  return static_cast<uint32_t>(
      static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rsrc, lf_palette_offset(f, xi, yi, row_fix), 0, 0)));
}
__device__ __forceinline__ double lf_palette_value(uint32_t lds_address) {
  return *reinterpret_cast<lds_f64_t*>(static_cast<uintptr_t>(lds_address));
}

// kFast (dense sets): the cell of an end-point is floor(v), v = the reference's separately rounded (p.cos - q.sin + t) / res.
// An FMA evaluation v~ of the same real number (pose pre-multiplied by 1/res: 2 ops per axis instead of 5) differs from v
// by less than 2^-35 cells as long as every term stays below 2^15 cells (7 roundings of relative size 2^-53 in all), so
// floor(v~) == floor(v) unless v~ lies within 2^-33 of an integer.  v~ + 1.5 * 2^20 (round to nearest) has exactly 32
// fraction bits in its low word and kFastBias + floor(v~) in its high word whenever the low word is not zero; a zero low
// word (once in 2^32 end-points) sends the whole group of 8 beams through the exact evaluation instead, and so does a wave
// holding a particle farther than 2^14 cells from the grid origin.  Clamps and offsets work on the biased high words.
// 12 VALU ops per end-point instead of 17, same cells bit for bit; it pays where a wave's end-points fall into few
// tiles (the gather costs a cycle per distinct line: profiles/r01_calib_gather_cost.txt), i.e. for dense particle sets.
//
// kFar (dispersed sets: global localisation, a kidnapped robot): neighbouring lanes are metres and radians apart, every look-up
// is a cache line of its own, and the launch is bound by what the L2 misses pull in (profiles/r02_dispersed_study.txt).  But
// more than half of a map is free space farther than max_obstacle_distance from anything, where the field holds one value:
// the bitmap of such tiles (FieldView::far_bits, 32 KB for 4000^2 cells) sits in LDS at far_base, a look-up into a far tile
// takes the common entry without touching memory (its buffer offset is pushed out of range, which returns 0 and moves
// nothing), and only the look-ups near obstacles or outside the grid reach the table.  Same entries, same sums.
__device__ __forceinline__ uint32_t far_block(uint32_t b, uint32_t blocks /* a multiple of 8 */) { return (b & 7u) * (blocks >> 3) + (b >> 3); }
template <bool kFast, bool kFar = false>
__global__ __launch_bounds__(kPalBlock) void k_reweight_lf_palette(double* __restrict__ w, uint64_t n, FieldView f,
                                                                   const double* __restrict__ pts, uint32_t B,
                                                                   const uint32_t* __restrict__ perm, const double4* __restrict__ pose,
                                                                   double* __restrict__ partial, uint32_t beams_per_segment,
                                                                   uint32_t far_base) {
  static_assert(kFast || !kFar, "far tiles ride on the biased coordinates of the FMA variant");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if constexpr (kFar) {
    if (static_cast<uint64_t>(far_block(blockIdx.x, gridDim.x)) * kPalBlock >= n) return;  // padding of the grid to 8 x
    const uint4* src = reinterpret_cast<const uint4*>(f.far_bits);
    uint4* dst = reinterpret_cast<uint4*>(smem + far_base);
    for (uint32_t j = threadIdx.x; j < f.far_bytes / 16; j += kPalBlock) dst[j] = src[j];
  }
  {
    uint32_t* s_row = reinterpret_cast<uint32_t*>(smem);
    constexpr uint32_t row_fix_stored = kFast ? kFastBiasX : 0u;
    for (uint32_t j = threadIdx.x; j < f.H + 2; j += kPalBlock)
      s_row[j] = palette_row_offset(static_cast<int32_t>(j) - 1, f.pal_pitch) - row_fix_stored;
    double* s_pal = reinterpret_cast<double*>(smem + f.pal_base);
    for (uint32_t k = threadIdx.x; k < f.pal_count; k += kPalBlock) s_pal[k] = f.pal_val[k];
  }
  __syncthreads();
  // kFar: block b runs on XCD b % 8 (observed; nothing but speed depends on it) - XCD k takes the k-th contiguous eighth of the
  // order, in order.  The grid is a multiple of 8 workgroups; the ones past the set have left above.
  const uint32_t block = kFar ? far_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const uint64_t t = static_cast<uint64_t>(block) * kPalBlock + threadIdx.x;
  const uint64_t tt = t < n ? t : n - 1;
  const uint32_t i = perm[tt];
  const Pose2 T = ordered_pose(f.world_to_field, pose, i);  // likelihood_field_model.hpp:70
  const double ct = T.r.c, st = T.r.s, xt = T.x, yt = T.y;
  const uint32_t b_begin = partial ? blockIdx.y * beams_per_segment : 0u;
  const uint32_t b_end = partial ? (b_begin + beams_per_segment < B ? b_begin + beams_per_segment : B) : B;
  double acc = (f.prob || partial) ? 0.0 : 1.0;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(f.pal_idx), 0, static_cast<int>(f.pal_bytes), 0x00020000);
  // Software-pipelined over groups of 8 beams, two groups (A, B) in flight alternately: the index gathers of one group
  // are outstanding while the end-points of the next are computed; a group's palette values are added, in beam order,
  // one step later.
  constexpr uint32_t row_fix = kFast ? kFastBiasX : 0u;
  auto issue = [&](uint32_t (&e)[8], uint32_t b0) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double px = pts[2 * (b0 + k)], py = pts[2 * (b0 + k) + 1];
      v[2 * k] = (px * ct - py * st + xt) * f.inv_resolution;
      v[2 * k + 1] = (px * st + py * ct + yt) * f.inv_resolution;
    }
    floor_rd_16(v);
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = lf_palette_fetch(rsrc, f, floor_rd_result(v[2 * k]), floor_rd_result(v[2 * k + 1]), row_fix);
  };
  auto consume = [&](uint32_t (&e)[8], bool) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = lf_palette_value(e[k]);
    acc += sum4(t[0], t[1], t[2], t[3]);
    acc += sum4(t[4], t[5], t[6], t[7]);
  };
  uint32_t b = b_begin;
  uint32_t groups = (b_end - b_begin) / 8;
  bool fast = false;
  if constexpr (kFast) {
    const double ict = ct * f.inv_resolution, ist = st * f.inv_resolution, ixt = xt * f.inv_resolution, iyt = yt * f.inv_resolution;
    const bool lane_small = fabs(ixt) < 16384.0 && fabs(iyt) < 16384.0;  // false for NaN as well
    fast = __builtin_amdgcn_ballot_w64(!lane_small) == 0;
    if (groups && fast) {
      const int c_lo = static_cast<int>(kFastBias) - 1, x_hi = static_cast<int>(kFastBias + f.W), y_hi = static_cast<int>(kFastBias + f.H);
      const uint32_t row_bias = 4u - (kFastBias << 2);  // LDS byte address of the row entry = (biased y << 2) + row_bias
      auto issue_fast = [&](uint32_t (&e)[8], uint32_t b0) {
        uint32_t near_integer = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const double px = pts[2 * (b0 + k)], py = pts[2 * (b0 + k) + 1];
          const double sx = __builtin_fma(px, ict, __builtin_fma(-py, ist, ixt)) + kFastMagic;
          const double sy = __builtin_fma(px, ist, __builtin_fma(py, ict, iyt)) + kFastMagic;
          const uint64_t bx = __builtin_bit_cast(uint64_t, sx), by = __builtin_bit_cast(uint64_t, sy);
          near_integer = min3_u32(near_integer, static_cast<uint32_t>(bx), static_cast<uint32_t>(by));
          const int xc = med3_i32(static_cast<int>(bx >> 32), c_lo, x_hi), yc = med3_i32(static_cast<int>(by >> 32), c_lo, y_hi);
          const uint32_t row = *reinterpret_cast<lds_u32_t*>(static_cast<uintptr_t>((static_cast<uint32_t>(yc) << 2) + row_bias));
          uint32_t offset = (static_cast<uint32_t>(xc) << 4) + row;
          uint32_t common = 0u;
          if constexpr (kFar) {
            // tile of the clamped cell, border included: (coordinate - kFastBias + 8) >> 3; kFastBias has 19 zero low bits
            const uint32_t tx8 = static_cast<uint32_t>(xc) + 8u, ty8 = static_cast<uint32_t>(yc) + 8u;
            const uint32_t at = mad_u24(__builtin_amdgcn_ubfe(ty8, 3, 16), f.far_row_bytes, __builtin_amdgcn_ubfe(tx8, 6, 13)) + far_base;
            const uint32_t byte = *reinterpret_cast<lds_u8_t*>(static_cast<uintptr_t>(at));
            const bool far = __builtin_amdgcn_ubfe(byte, __builtin_amdgcn_ubfe(tx8, 3, 3), 1) != 0u;
            offset = far ? 0xFFFFFFF0u : offset;
            common = far ? f.far_entry : 0u;
          }
          e[k] = static_cast<uint32_t>(static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rsrc, offset, 0, 0))) | common;
        }
        if (__builtin_amdgcn_ballot_w64(near_integer == 0) != 0) issue(e, b0);  // an end-point on a cell boundary: exact path
      };
      uint32_t ea[8], eb[8];
      issue_fast(ea, b);
      b += 8;
      --groups;
      while (groups >= 2) {
        issue_fast(eb, b);
        consume(ea, true);
        issue_fast(ea, b + 8);
        consume(eb, true);
        b += 16;
        groups -= 2;
      }
      if (groups) {
        issue_fast(eb, b);
        consume(ea, true);
        consume(eb, false);
        b += 8;
      } else {
        consume(ea, false);
      }
      groups = 0;
    }
  }
  if (groups) {
    uint32_t ea[8], eb[8];
    issue(ea, b);
    b += 8;
    --groups;
    while (groups >= 2) {
      issue(eb, b);
      consume(ea, true);
      issue(ea, b + 8);
      consume(eb, true);
      b += 16;
      groups -= 2;
    }
    if (groups) {
      issue(eb, b);
      consume(ea, true);
      consume(eb, false);
      b += 8;
    } else {
      consume(ea, false);
    }
  }
  auto term = [&](uint32_t at) {
    const double px = pts[2 * at], py = pts[2 * at + 1];
    double vx = (px * ct - py * st + xt) * f.inv_resolution, vy = (px * st + py * ct + yt) * f.inv_resolution;
    floor_rd_2(vx, vy);
    return lf_palette_value(lf_palette_fetch(rsrc, f, floor_rd_result(vx), floor_rd_result(vy), row_fix));
  };
  if (b + 4 <= b_end) {
    const double t0 = term(b), t1 = term(b + 1), t2 = term(b + 2), t3 = term(b + 3);
    acc += sum4(t0, t1, t2, t3);
    b += 4;
  }
  for (; b < b_end; ++b) acc += term(b);
  if (t < n) {
    if (partial) {
      partial[static_cast<size_t>(blockIdx.y) * n + t] = acc;
    } else {
      w[i] = w[i] * (f.prob ? exp(acc) : acc);
    }
  }
}

// Variant D — one wavefront per PARTICLE (or per few), one lane per BEAM, over the palette form of the table.
// (1) The kernel of SMALL sets (below 65 536 particles by default, option lf_small_particles - the reference's usual 500 - 2000
// among them): a wave owns a tile of per_wave = 1 .. 16 particles in index order, no ordering pass; their world->field
// transforms are computed lane-parallel and broadcast one at a time through SGPRs, and the 64 lanes take 64 consecutive beams
// at a time.  2000 particles x 1080 beams are 2000 waves of 17 gathers each instead of 32 waves walking 1080 beams one after the
// other (whole update 0.35 -> 0.06 ms, profiles/r02_small_filters.txt); it stays ahead of the ordered kernels up to ~60K
// particles, whose sparse clouds fit few LDS patches (tools/exp_mid.py).
// (2) An alternative for large DISPERSED sets (global localisation), behind option lf_dispersed = 1 / lf_variant = 3, 64
// particles per wave: the 64 look-ups of a gather belong to ONE pose - their end-points trace the walls the scan saw and
// neighbouring beams share 8x8-cell tiles, whatever the cloud looks like.  Measured on 1M particles spread over the 4000^2 map
// (profiles/r02_dispersed_study.txt): 23 lines per gather, 398 M L2 requests per launch - the same as the ordered-lanes gather
// kernel gets out of that set (400 M) - but fewer of them hit in L2 (33 % vs 48 %), and the launch is bound by what the L2
// misses pull in (33 GB per launch at 7.5 TB/s): 4.46 ms vs 3.72 ms.  Not chosen by default there.
// End-points by the reference's separately rounded arithmetic; a lane adds its beams in scan order, the 64 lane sums are
// added in a fixed tree (wave_sum_f64): the weight differs from the sum of the lane-per-particle kernels in rounding only.
// Workgroup memory as in k_reweight_lf_palette: [0, (H+2)*4) row offsets, [pal_base, ...) the palette; no other LDS.
constexpr int kBeamsBlock = 256;
__global__ __launch_bounds__(kBeamsBlock) void k_reweight_lf_beams(Particles p, uint64_t n, FieldView f, const double2* __restrict__ pts,
                                                                   uint32_t B, uint32_t per_wave /* particles of a wave: 1 .. 64 */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    uint32_t* s_row = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t j = threadIdx.x; j < f.H + 2; j += kBeamsBlock) s_row[j] = palette_row_offset(static_cast<int32_t>(j) - 1, f.pal_pitch);
    double* s_pal = reinterpret_cast<double*>(smem + f.pal_base);
    for (uint32_t k = threadIdx.x; k < f.pal_count; k += kBeamsBlock) s_pal[k] = f.pal_val[k];
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t tile = static_cast<uint64_t>(blockIdx.x) * (kBeamsBlock / kWave) + (threadIdx.x >> 6);
  const uint64_t base = tile * per_wave;
  if (base >= n) return;
  const uint32_t cnt = static_cast<uint32_t>(n - base < per_wave ? n - base : per_wave);
  const uint64_t i = base + lane;
  Pose2 state = pose_identity();
  if (lane < cnt) state = load_pose(p, i);
  const Pose2 T = pose_mul(f.world_to_field, state);  // likelihood_field_model.hpp:70
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(f.pal_idx), 0, static_cast<int>(f.pal_bytes), 0x00020000);
  const uint32_t full = B & ~255u;  // beams taken four per lane at a time (their gathers in flight together)
  double mine = 0.0;
#pragma unroll 1
  for (uint32_t q = 0; q < cnt; ++q) {
    const double ct = readlane_f64(T.r.c, q), st = readlane_f64(T.r.s, q);
    const double xt = readlane_f64(T.x, q), yt = readlane_f64(T.y, q);
    double acc = 0.0;
#pragma unroll 1
    for (uint32_t b0 = 0; b0 < full; b0 += 256) {
      double v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 pt = pts[b0 + 64 * k + lane];
        v[2 * k] = (pt.x * ct - pt.y * st + xt) * f.inv_resolution;
        v[2 * k + 1] = (pt.x * st + pt.y * ct + yt) * f.inv_resolution;
      }
      floor_rd_8(v);
      uint32_t e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = lf_palette_fetch(rsrc, f, floor_rd_result(v[2 * k]), floor_rd_result(v[2 * k + 1]), 0u);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += lf_palette_value(e[k]);
    }
    for (uint32_t b = full + lane; b < B; b += kWave) {
      const double2 pt = pts[b];
      double vx = (pt.x * ct - pt.y * st + xt) * f.inv_resolution, vy = (pt.x * st + pt.y * ct + yt) * f.inv_resolution;
      floor_rd_2(vx, vy);
      acc += lf_palette_value(lf_palette_fetch(rsrc, f, floor_rd_result(vx), floor_rd_result(vy), 0u));
    }
    const double total = wave_sum_f64(acc);
    if (lane == q) mine = total;
  }
  if (lane < cnt) p.w[i] = p.w[i] * (f.prob ? exp(mine) : 1.0 + mine);
}

// Variant D for large DISPERSED sets, round 6 (option lf_dispersed = 2): the lanes over the beams of a pose as above, but with everything
// that made the ordered-lanes gather kernel of such sets fast (k_reweight_lf_palette<true, true>) - the FMA end-points on biased cell
// coordinates with their exact fallback, the far-tile bitmap in LDS, the particles taken in the position-major order with XCD k walking the
// k-th contiguous eighth of it (98 % L2 hits) - and without what bounds that kernel: there every lane's look-up is a cache line of its
// own (the L1 takes a cycle per quad of lanes and line: 573 M of them per launch at 1M x 1080, 0.93 ms), here the four lanes of a quad hold
// four consecutive beams of one pose, whose end-points lie cells apart along the wall the scan saw - one tile or two -, and a quad whose
// four end-points fall into far tiles costs nothing.  The scan sits in LDS; a lane reads its beam of a round once for kFarBeamsPoses poses,
// which go through SGPRs (register blocking: 4 independent look-ups per point read, the next round's issued before this round's are
// consumed).  The far test starts from the byte offset the look-up has anyway (FieldView::far_linear: offset >> 7 is the tile's linear
// index).  A lane adds its beams in scan order, the 64 lane sums are added in a fixed tree: the weight differs from the lane-per-particle
// kernels' in rounding only (as k_reweight_lf_beams').
// Workgroup memory: the bitmap at 0 (the byte of a tile is at offset >> 10: no base to add), the row offsets (less kFastBiasX) behind it, the
// palette kFarBeamsPalShift bytes above its place in the other kernels (the table's entries are LDS addresses from pal_base on; the
// shift is a constant and rides in the DS instruction's offset field), then the scan (16 bytes per beam, whole rounds of 64).  The
// bitmap may take up to kFarBeamsPalShift bytes (maps of up to ~4000^2 cells); the rows then fit below the palette whatever H is.
constexpr int kFarBeamsBlock = 768;  // 12 waves; two workgroups per CU: six waves per SIMD at <= 80 registers
constexpr int kFarBeamsWaves = 6;
constexpr int kFarBeamsPoses = 2;    // poses per point read (measured: 1 -> + 5 - 35 %, 4 -> + 25 - 50 %: registers)
typedef double f64x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const f64x2_t lds_f64x2_t;
constexpr uint32_t kFarBeamsPalShift = 32768;
constexpr uint32_t kFarBeamsChunk = 16;  // poses a wave transforms at a time (<= 64, a multiple of kFarBeamsPoses)
// (kProb: the weight is exp(sum) - likelihood_field_prob_model.hpp:77-90; an instance of its own because the exponential's constants, hoisted
// out of every loop, would otherwise cost the likelihood-field model's instance twenty registers)
template <bool kProb>
__global__ __launch_bounds__(kFarBeamsBlock) __attribute__((amdgpu_waves_per_eu(kFarBeamsWaves, kFarBeamsWaves))) void k_reweight_lf_far_beams(double* __restrict__ w, uint64_t n, FieldView f,
                                                                          const double2* __restrict__ pts, uint32_t B,
                                                                          const uint32_t* __restrict__ perm, const double4* __restrict__ pose,
                                                                          uint32_t pts_at, uint32_t per_wave, uint32_t unit_weights) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int J = kFarBeamsPoses;
  const uint32_t rows_at = f.far_linear_bytes;  // (a multiple of 16)
  const uint32_t rounds = (B + 63u) >> 6;
  {
    const uint4* src = reinterpret_cast<const uint4*>(f.far_linear);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (uint32_t j = threadIdx.x; j < f.far_linear_bytes / 16; j += kFarBeamsBlock) dst[j] = src[j];
    uint32_t* s_row = reinterpret_cast<uint32_t*>(smem + rows_at);
    for (uint32_t j = threadIdx.x; j < f.H + 2; j += kFarBeamsBlock)
      s_row[j] = palette_row_offset(static_cast<int32_t>(j) - 1, f.pal_pitch) - kFastBiasX;
    double* s_pal = reinterpret_cast<double*>(smem + kFarBeamsPalShift + f.pal_base);
    for (uint32_t k = threadIdx.x; k < f.pal_count; k += kFarBeamsBlock) s_pal[k] = f.pal_val[k];
    double2* s_pts = reinterpret_cast<double2*>(smem + pts_at);
    for (uint32_t b = threadIdx.x; b < rounds * 64u; b += kFarBeamsBlock) s_pts[b] = b < B ? pts[b] : double2{0.0, 0.0};
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t first = (static_cast<uint64_t>(far_block(blockIdx.x, gridDim.x)) * (kFarBeamsBlock / kWave) + wave) * per_wave;
  if (first >= n) return;  // (no barrier below)
  const uint32_t cnt = static_cast<uint32_t>(n - first < per_wave ? n - first : per_wave);
  const bool dead_last = ((rounds - 1u) << 6) + lane >= B;  // no such beam in the last round
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(f.pal_idx), 0, static_cast<int>(f.pal_bytes), 0x00020000);
  const int c_lo = static_cast<int>(kFastBias) - 1, x_hi = static_cast<int>(kFastBias + f.W), y_hi = static_cast<int>(kFastBias + f.H);
  const uint32_t row_bias = rows_at + 4u - (kFastBias << 2);  // LDS byte address of the row entry = (biased y << 2) + row_bias
  const uint32_t my_point = pts_at + (lane << 4);
  // The poses of a wave, kFarBeamsChunk at a time: lane q of the wave transforms pose q and leaves (cos, sin, x, y) / resolution in the wave's
  // 32-byte slots of LDS; the scan loop reads a pose back as a broadcast - every lane holds it in VECTOR registers (an FMA takes one
  // scalar operand: from SGPRs the translations would have to be moved to vector registers in every round).
  const uint32_t slots_at = pts_at + (rounds << 10) + wave * (kFarBeamsChunk * 32u);
#pragma unroll 1
  for (uint32_t c0 = 0; c0 < cnt; c0 += kFarBeamsChunk) {
    const uint32_t m = cnt - c0 < kFarBeamsChunk ? cnt - c0 : kFarBeamsChunk;
    uint32_t i = 0;
    Pose2 T = pose_identity();
    if (lane < m) {
      i = perm[first + c0 + lane];
      T = ordered_pose(f.world_to_field, pose, i);  // likelihood_field_model.hpp:70
    }
    unsigned long long large;
    {
      const double l_ict = T.r.c * f.inv_resolution, l_ist = T.r.s * f.inv_resolution, l_ixt = T.x * f.inv_resolution,
                   l_iyt = T.y * f.inv_resolution;
      large = __builtin_amdgcn_ballot_w64(!(fabs(l_ixt) < 16384.0 && fabs(l_iyt) < 16384.0));  // NaN as well
      if (lane < kFarBeamsChunk) {
        f64x2_t* slot = reinterpret_cast<f64x2_t*>(smem + slots_at + (lane << 5));
        slot[0] = f64x2_t{l_ict, l_ist};
        slot[1] = f64x2_t{l_ixt, l_iyt};
      }
    }
    double mine = 0.0;
#pragma unroll 1
    for (uint32_t q0 = 0; q0 < m; q0 += J) {  // poses q0 .. q0 + J - 1 (beyond m: the identity's, computed and dropped)
      double ict[J], ist[J], ixt[J], iyt[J], acc[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const uint32_t at = slots_at + ((q0 + j) << 5);  // (q0 + j < kFarBeamsChunk: J divides it)
        const f64x2_t rot = *reinterpret_cast<lds_f64x2_t*>(static_cast<uintptr_t>(at));
        const f64x2_t tr = *reinterpret_cast<lds_f64x2_t*>(static_cast<uintptr_t>(at + 16u));
        ict[j] = rot.x;
        ist[j] = rot.y;
        ixt[j] = tr.x;
        iyt[j] = tr.y;
        acc[j] = 0.0;
      }
      bool exact = ((large >> q0) & ((1ull << J) - 1ull)) != 0;  // (uniform)
      if (!exact) {
        uint32_t near_integer = 0xFFFFFFFFu;
        auto issue = [&](uint32_t (&e)[J], uint32_t k) {
          const f64x2_t pt = *reinterpret_cast<lds_f64x2_t*>(static_cast<uintptr_t>(my_point + (k << 10)));
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const double sx = __builtin_fma(pt.x, ict[j], __builtin_fma(-pt.y, ist[j], ixt[j])) + kFastMagic;
            const double sy = __builtin_fma(pt.x, ist[j], __builtin_fma(pt.y, ict[j], iyt[j])) + kFastMagic;
            const uint64_t bx = __builtin_bit_cast(uint64_t, sx), by = __builtin_bit_cast(uint64_t, sy);
            near_integer = min3_u32(near_integer, static_cast<uint32_t>(bx), static_cast<uint32_t>(by));
            const int xc = med3_i32(static_cast<int>(bx >> 32), c_lo, x_hi), yc = med3_i32(static_cast<int>(by >> 32), c_lo, y_hi);
            const uint32_t row = *reinterpret_cast<lds_u32_t*>(static_cast<uintptr_t>((static_cast<uint32_t>(yc) << 2) + row_bias));
            const uint32_t offset = (static_cast<uint32_t>(xc) << 4) + row;
            const uint32_t byte = *reinterpret_cast<lds_u8_t*>(static_cast<uintptr_t>(offset >> 10));
            const uint32_t far = __builtin_amdgcn_ubfe(byte, __builtin_amdgcn_ubfe(offset, 7, 3), 1);
            // a far look-up's offset is pushed out of range (the load returns 0 and moves nothing); its entry is the common one
            const uint32_t loaded = static_cast<uint32_t>(
                static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rsrc, offset | (far << 31), 0, 0)));
            e[j] = mad_u24(far, f.far_entry, loaded);
          }
        };
        // (a lane without a beam in the scan's last round looked up the pose's own cell: not added)
        auto consume = [&](const uint32_t (&e)[J], bool last) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
            double v = lf_palette_value(e[j] + kFarBeamsPalShift);
            if (last) v = dead_last ? 0.0 : v;  // (uniform branch)
            acc[j] += v;
          }
        };
        // two rounds in flight alternately; the scan's last round (its dead lanes) outside the loop
        uint32_t ea[J], eb[J];
        if (rounds > 1) {
          issue(ea, 0);
          uint32_t k = 1;
          while (k + 2 < rounds) {
            issue(eb, k);
            consume(ea, false);
            issue(ea, k + 1);
            consume(eb, false);
            k += 2;
          }
          if (k + 1 < rounds) {
            issue(eb, k);
            consume(ea, false);
            issue(ea, k + 1);
            consume(eb, false);
            consume(ea, true);
          } else {
            issue(eb, k);
            consume(ea, false);
            consume(eb, true);
          }
        } else {
          issue(ea, 0);
          consume(ea, true);
        }
        exact = __builtin_amdgcn_ballot_w64(near_integer == 0) != 0;  // an end-point on a cell boundary: these poses again, exactly
      }
      if (exact) {  // the reference's separately rounded arithmetic (rare: a pose 2^14 cells from the origin, an end-point on a boundary)
#pragma unroll 1
        for (int j = 0; j < J; ++j) {
          const uint32_t iq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(i), static_cast<int>(q0) + j));
          const Pose2 Tq = ordered_pose(f.world_to_field, pose, iq);
          double sum = 0.0;
#pragma unroll 1
          for (uint32_t k = 0; k < rounds; ++k) {
            if (64u * k + lane < B) {
              const double2 pt = pts[64u * k + lane];
              double vx = (pt.x * Tq.r.c - pt.y * Tq.r.s + Tq.x) * f.inv_resolution, vy = (pt.x * Tq.r.s + pt.y * Tq.r.c + Tq.y) * f.inv_resolution;
              floor_rd_2(vx, vy);
              const int xc = clamp_cell(floor_rd_result(vx), f.W), yc = clamp_cell(floor_rd_result(vy), f.H);
              const uint32_t row = *reinterpret_cast<lds_u32_t*>(static_cast<uintptr_t>(rows_at + (static_cast<uint32_t>(yc + 1) << 2)));
              const uint32_t at = (static_cast<uint32_t>(xc) << 4) + kFastBiasX + row;
              sum += lf_palette_value(static_cast<uint32_t>(static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(rsrc, at, 0, 0))) +
                                      kFarBeamsPalShift);
            }
          }
          // (acc[j] with a runtime j would go through scratch)
#pragma unroll
          for (int jj = 0; jj < J; ++jj)
            if (j == jj) acc[jj] = sum;
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double total = wave_sum_f64(acc[j]);
        if (lane == q0 + j) mine = total;
      }
    }
    if (lane < m) w[i] = (unit_weights != 0u ? 1.0 : w[i]) * (kProb ? exp(mine) : 1.0 + mine);
  }
}

// The palette kernel with the index table read through LDS patches.
// A scattered 64-lane 2-byte gather costs the CU's texture-address pipe one cycle per quad of lanes per line (16+ per
// instruction, profiles/r02_calib_gather_cost.txt): the floor of k_reweight_lf_palette.  An LDS read of the same shape costs
// two.  A workgroup holds neighbours of the spatial order, one lane per particle.  For a group of 8 consecutive beams the
// end-point of particle p differs from that of a reference pose in the middle of the workgroup's particles by
// (t_p - t_ref) + (R_p - R_ref) q, at most
//     (Dx, Dy) + |q| Drot      Dx, Dy = max |t_p - t_ref| per axis,  Drot = max |R_p - R_ref| = max 2 |sin(dtheta / 2)|
// over the workgroup (one reduction in the prologue).  If the bounding box of the reference's 8 end-point cells, widened by
// that margin (+2 cells for every rounding involved), fits into 64 x 64 cells, the group goes through a PATCH: that part of the
// index table is copied into LDS - 512 coalesced 16-byte pieces (a tile column of the 8x8-tiled table = 8 cells in y),
// column-major, 144 bytes per column - and every look-up of the group is an LDS read at cx * 144 + cy * 2 + K, with no test: the
// bound is the proof that it lies inside.  Otherwise (the cloud's fringe, a range discontinuity inside the group) the group is
// gathered as in k_reweight_lf_palette.  Where a look-up comes from changes no result.
// The plan - origin and mode of every group - is made once, in the prologue, one thread per group.  Then one s_barrier per
// group: behind barrier g the patch of group g is visible and the buffer of g - 1 is free.
//
// Who copies the patches: a workgroup is seven waves of particles (448) and a PRODUCER wave without particles, which fetches every
// patch through its registers two groups ahead.  Round 5 built the alternative - all eight waves hold particles (512) and every wave
// copies one tile row of each patch, a 16-byte piece per lane in flight across a step: no wave slot without arithmetic, bit-identical -
// and measured it 5 % SLOWER (0.404 against 0.382 ms on a settled cloud; profiles/r05_lf_coop_v2_ab.txt, r05_lf_diet_ab.txt): what a
// wave's instruction stream carries per step - 5 vector, 10 scalar and 2 memory instructions for its piece - costs more than the
// producer's idle slot.  The first version, with 45 scalar instructions per step for the piece's offsets, was 23 % slower
// (r05_lf_coop_v1_ab.txt): every instruction of the main loop, scalar ones included, costs about 2.7 us per step at 1M particles.
//
// End-points: v = the reference's separately rounded (p.cos - q.sin + t) / res, cell = floor(v).  Evaluated here as
//     s = fma(p, c', fma(-q, s', t' + M)),   c' = cos / res, s' = sin / res, t' = t / res,   M = 1.5 * 2^20 + 2^-31
// every partial result lies in [2^20, 2^21) and is rounded to the grid u = 2^-32: three roundings of at most u / 2, plus the
// roundings of c', s', t' and of the reference's own evaluation (below 2^-36 cells while every term stays below 2^15 cells),
// so s = v + M + err with |err| < 2u.  With I = floor(v): s lies in (1.5 * 2^20 + I + frac(v), ... + 4u), on the grid, so
// either its integer part is I (high word = kFastBias + I, whatever the low word) or it carried and its low word is below 4.
// A group with a low word below 4 (4 in 2^32 end-points) is added by the exact evaluation, beam by beam, and so is every
// group of a wave holding a particle farther than 2^14 cells from the grid origin.  4 VALU operations per end-point.
// (The measurement builds of rounds 3 / 4 - ablations, barrier timers, the workgroup timeline - are built from that round's sources:
// tools/build_variant.sh.)
constexpr int kPatchW = 64, kPatchH = 64;  // cells
// Bytes per patch column: the 128 of its cells + 16, so that the bank of a cell is (4 x + y / 2) mod 32 - neighbouring
// columns on different banks (with 128, every column of a row pair would share one).
constexpr uint32_t kPatchPitch = kPatchH * 2 + 16;
static_assert(kPatchPitch == 144, "patch_address() carries the pitch as a literal");
constexpr uint32_t kPatchBytes = kPatchW * kPatchPitch;
constexpr int kPatchBlock = 512;                      // threads of k_reweight_lf_patch
constexpr uint32_t kPatchParticles = kPatchBlock - 64;  // per workgroup: seven waves of particles and a producer wave
constexpr uint32_t kPatchPlanned = 192;               // groups with a plan entry (scans of up to 1536 points); the ones beyond are gathered
// THREE patch buffers, so that nobody waits for its own look-ups at a group's barrier - the look-ups of group g (issued at the end of
// step g, used in step g + 1) have returned long before the buffer of group g is written again behind barrier g + 2, whereas with
// two buffers every wave had to sit out its outstanding LDS reads in front of each barrier.  Behind them: the two plans and the
// prologue's partial results.
constexpr uint32_t kPatchBuffers = 3;
constexpr uint32_t kPatchLdsBytes = kPatchBuffers * kPatchBytes + kPatchPlanned * 32 + 16 + 48 * 4;  // (+ 16: the zero entry behind the plan)
static_assert(kPatchPlanned * 8 * 16 <= kPatchBuffers * kPatchBytes, "the per-beam records of the plan live in the patch buffers");
constexpr double kPatchMagic = 1572864.0 + 4.656612873077392578125e-10;     // 1.5 * 2^20 + 2^-31
// 6 waves per SIMD = three workgroups per CU: at most 80 registers
// exp() as a call: inlined into the queue's loop, its dozen polynomial coefficients are hoisted out of the loop into registers and from
// there into scratch memory (likelihood_field_prob_model.hpp:89 - once per particle and launch, and not at all for the plain model).
__device__ __attribute__((noinline)) double exp_out_of_line(double x) { return exp(x); }
// The kernel's arguments as one structure: it is the kernel argument segment, byte for byte.
struct PatchArgs {
  double* w;
  uint64_t n;
  FieldView f;
  const double* pts;
  uint32_t B;
  const uint32_t* perm;
  const double4* pose;
  double* partial;
  uint32_t beams_per_segment;
  uint32_t patch_base;  // LDS byte offset, 16-aligned
  PatchStats stats;
  uint32_t nblocks;
  uint32_t ends_first;  // 1: the blocks are taken from both ends of the order inwards
  uint32_t unit_weights;  // 1: every weight of the set is 1.0 (a set fresh from a resampling or an initialisation - particle_traits.hpp:105 -:
                          // the host knows): the new weight is the sensor term itself, 1.0 x = x, and the scattered load of the old weight -
                          // the last dependent memory round trip of a block's end, 8 us of a 1M launch - is not made
};
template <bool kQueue>
__global__ __launch_bounds__(kPatchBlock) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_reweight_lf_patch(PatchArgs args) {
  // The arguments are read from the kernel argument segment at the start of every block (scalar loads), its address through an empty asm
  // statement so that the loads are not hoisted out of the queue's loop (kQueue): held in registers from the kernel's entry on they would all
  // stay alive around that loop - 90 scalar registers spilled.
  typedef const __attribute__((address_space(4))) unsigned char* kernarg_bytes_t;
  const kernarg_bytes_t kernarg = (kernarg_bytes_t)(__builtin_amdgcn_kernarg_segment_ptr());
  const PatchArgs* ka = (const PatchArgs*)(kernarg);
  const uint32_t patch_base = ka->patch_base;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {
    const FieldView& f = ka->f;  // (used once, in front of the loop)
    uint32_t* s_row = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t j = threadIdx.x; j < f.H + 2; j += kPatchBlock)
      s_row[j] = palette_row_offset(static_cast<int32_t>(j) - 1, f.pal_pitch) - kFastBiasX;
    double* s_pal = reinterpret_cast<double*>(smem + f.pal_base);
    for (uint32_t k = threadIdx.x; k < f.pal_count; k += kPatchBlock) s_pal[k] = f.pal_val[k];
    // the entry behind the plan, read by the groups beyond it (scans of more than 8 kPatchPlanned points): gathered, nothing to fetch
    if (threadIdx.x == 0) *reinterpret_cast<int4*>(smem + patch_base + kPatchBuffers * kPatchBytes + kPatchPlanned * 32) = int4{0, 0, 0, 0};
  }
  // plan entry of group g: {x0A, y0A | flags, x0B, y0B | first beam of half B} - biased origins, y0 multiples of 8;
  // flags: 1 = the group goes through a patch, 2 = split into two halves side by side (32 x 64 cells each), 4 = split into two
  // halves one above the other (64 x 32 cells each).  A group whose 8 end-points straddle a range discontinuity fits no single
  // patch however tight the cloud (3 - 5 % of the groups of an indoor scan); its beams [0, k) and [k, 8) almost always fit two
  // half patches, which share the buffer of one whole patch: the addressing of the look-ups does not change, only the constant K
  // differs between the two halves (a scalar select per beam).
  // The look-ups read a second entry per group: {KA', meta, KB', 0}: the constants of the two halves' LDS addresses less the
  // buffer's base (cell (cx, cy) sits at cx * pitch + cy * 2 + K), meta = 0: gathered, 8: one whole patch, k = 1 .. 7: two halves,
  // the second one from beam k on.
  auto buffer_of = [&](uint32_t g) -> uint32_t {  // LDS byte address of the buffer that holds the patch of group g (g uniform, < 2^16)
    const uint32_t slot = g - 3u * ((g * 0xAAABu) >> 17);
    return patch_base + slot * kPatchBytes;
  };
  int4* s_plan = reinterpret_cast<int4*>(smem + patch_base + kPatchBuffers * kPatchBytes);
  int4* s_plan_k = s_plan + kPatchPlanned;
  float* s_bound = reinterpret_cast<float*>(smem + patch_base + kPatchBuffers * kPatchBytes + kPatchPlanned * 32 + 16);  // [8][6]
  constexpr uint32_t kConsumers = kPatchBlock / 64 - 1;  // waves that hold particles
  constexpr uint32_t kParticles = kConsumers * 64;
  const uint32_t wave_id = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
  const bool producer = wave_id == (kPatchBlock / 64 - 1);  // a scalar branch: the roles run different loops
  // kQueue: the launch holds as many workgroups as the device keeps resident, and each takes blocks of the order from a counter until
  // none is left (stats.arrivals: nblocks + gridDim.x fetches per launch, the last of which wraps it to zero for the next one).  The
  // hardware deals the workgroups of a grid out to the XCDs in turn, the same number to each - at 1M particles 279 blocks for 96 slots,
  // and the XCDs are done with these equal shares 10 - 20 % apart (tools/exp_lf_workgroups.py, profiles/r04_lf_workgroup_timeline.txt);
  // with the queue an XCD that is ahead takes more blocks (256 .. 318 each): 2 - 4 % of the kernel at 1M, 1.3 % at 10M.
  uint32_t* s_next = reinterpret_cast<uint32_t*>(s_bound) + 46;
#pragma unroll 1
  for (;;) {
  uint32_t tid = threadIdx.x;  // (through an empty asm statement in the queue's loop: what derives from it is not kept across the blocks)
  if constexpr (kQueue) asm volatile("" : "+v"(tid));
  const uint32_t lane = tid & 63;
  uint32_t kernarg_offset = 0;
  if constexpr (kQueue) asm volatile("" : "+s"(kernarg_offset));
  const PatchArgs* kb = (const PatchArgs*)(kernarg + kernarg_offset);
  // References: read where they are used (again behind a barrier), which keeps what only the block's start, its end and the rare exact
  // path need out of the main loop's registers.  Copies: what the main loop reads.
  double* const& w = kb->w;
  const uint64_t& n = kb->n;
  const FieldView& f = kb->f;
  const double* const pts = kb->pts;
  const uint32_t B = kb->B;
  const uint32_t* const& perm = kb->perm;
  const double4* const& pose = kb->pose;
  double* const& partial = kb->partial;
  const uint32_t beams_per_segment = kb->beams_per_segment;
  const PatchStats& stats = kb->stats;
  const uint32_t nblocks = kb->nblocks;
  const bool ends_first = kb->ends_first != 0u;
  const uint32_t& unit_weights = kb->unit_weights;
  uint32_t block = blockIdx.x;
  if constexpr (kQueue) {
    __syncthreads();  // the block before is done with the workgroup's memory
    if (tid == 0) *s_next = atomicInc(stats.arrivals, nblocks + gridDim.x - 1u);
    __syncthreads();
    block = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(*s_next));
    if (block >= nblocks) break;
  }
  // The blocks are taken from both ends of the order inwards: 0, N - 1, 1, N - 2, ...  The ends of the heading-major order are the cloud's
  // fringe - the blocks that gather everything, twice as slow as the others, sit in its first and last tenth (tools/exp_lf_workgroups.py) -
  // and taken in order the slowest blocks were the launch's last; this way its end is made of the compact blocks of the middle.
  if (ends_first) block = (block & 1u) ? nblocks - 1u - (block >> 1) : (block >> 1);
  const uint64_t first = static_cast<uint64_t>(block) * kParticles;
  const uint64_t t = producer ? first : first + tid;  // (the producer holds no particle; it reads a valid one)
  const uint64_t tt = t < n ? t : n - 1;
  const uint32_t i = perm[tt];
  const Pose2 T = ordered_pose(f.world_to_field, pose, i);  // likelihood_field_model.hpp:70
  const double ct = T.r.c, st = T.r.s, xt = T.x, yt = T.y;
  const uint32_t b_begin = partial ? blockIdx.y * beams_per_segment : 0u;
  const uint32_t b_end = partial ? (b_begin + beams_per_segment < B ? b_begin + beams_per_segment : B) : B;
  const uint32_t groups = (b_end - b_begin) / 8;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(f.pal_idx), 0, static_cast<int>(f.pal_bytes), 0x00020000);
  const double ict = ct * f.inv_resolution, ist = st * f.inv_resolution, ixt = xt * f.inv_resolution, iyt = yt * f.inv_resolution;
  const double ixm = ixt + kPatchMagic, iym = iyt + kPatchMagic;
  const bool lane_small = fabs(ixt) < 16384.0 && fabs(iyt) < 16384.0;  // false for NaN as well

  // ---- the reference pose: the middle of the workgroup's box in x and y, its mean heading (any pose would do: the bound
  // below is taken against whatever is chosen here; a central one makes it small).  In cells, like ixt, iyt.
  float* s_part = s_bound;  // [8][6] partial results, then [8][4]
  if (!producer) {
    float lo_x = static_cast<float>(ixt), hi_x = lo_x, lo_y = static_cast<float>(iyt), hi_y = lo_y;
    float sum_c = static_cast<float>(ct), sum_s = static_cast<float>(st);
    for (int o = 32; o > 0; o >>= 1) {
      lo_x = fminf(lo_x, __shfl_xor(lo_x, o));
      hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
      lo_y = fminf(lo_y, __shfl_xor(lo_y, o));
      hi_y = fmaxf(hi_y, __shfl_xor(hi_y, o));
      sum_c += __shfl_xor(sum_c, o);
      sum_s += __shfl_xor(sum_s, o);
    }
    if (lane == 0) {
      float* mine = s_part + 6 * (tid >> 6);
      mine[0] = lo_x;
      mine[1] = hi_x;
      mine[2] = lo_y;
      mine[3] = hi_y;
      mine[4] = sum_c;
      mine[5] = sum_s;
    }
  }
  __syncthreads();
  double ref_c, ref_s, ref_x, ref_y;
  {
    float lo_x = s_part[0], hi_x = s_part[1], lo_y = s_part[2], hi_y = s_part[3], sum_c = s_part[4], sum_s = s_part[5];
    for (uint32_t k = 1; k < kConsumers; ++k) {
      lo_x = fminf(lo_x, s_part[6 * k]);
      hi_x = fmaxf(hi_x, s_part[6 * k + 1]);
      lo_y = fminf(lo_y, s_part[6 * k + 2]);
      hi_y = fmaxf(hi_y, s_part[6 * k + 3]);
      sum_c += s_part[6 * k + 4];
      sum_s += s_part[6 * k + 5];
    }
    const float len = sqrtf(sum_c * sum_c + sum_s * sum_s);
    ref_c = len > 0.f ? static_cast<double>(sum_c / len) : 1.0;  // a NaN stays one, and switches the patches off below
    ref_s = len > 0.f ? static_cast<double>(sum_s / len) : 0.0;
    ref_x = static_cast<double>(0.5f * (lo_x + hi_x));
    ref_y = static_cast<double>(0.5f * (lo_y + hi_y));
  }
  const double rc = ref_c * f.inv_resolution, rs = ref_s * f.inv_resolution;
  const double rxm = ref_x + kPatchMagic, rym = ref_y + kPatchMagic;
  __syncthreads();  // the partial results are read; their place takes the next ones
  // ---- the bound: every wave's lanes against the reference pose, then the workgroup's maxima
  // Rotation part: with M the reference's (c, s) as a matrix (a rotation up to the rounding of its float entries) and q' = M q / res
  // the reference end-point relative to the reference position - what the planner below evaluates anyway -, the particle's end-point
  // is off by (R_p M^-1 - I) q' = (A q'x - B q'y, B q'x + A q'y),  A = (c_p c + s_p s) / |M|^2 - 1,  B = (s_p c - c_p s) / |M|^2:
  // |B| = |sin d| and |A| = 1 - cos d for a heading difference d, so a beam along x moves in y only (to first order), and the
  // margin is taken per axis: max|A| max|q'x| + max|B| max|q'y| in x, max|B| max|q'x| + max|A| max|q'y| in y.
  // (stats.isotropic_margin: round 2's bound, |R_p - M| |q| on both axes, for A/B measurements.)
  if (!producer) {
    const double dc = ct - ref_c, ds = st - ref_s;
    float dx = static_cast<float>(fabs(ixt - ref_x)), dy = static_cast<float>(fabs(iyt - ref_y));
    float da, db;
    if (stats.isotropic_margin) {
      da = db = static_cast<float>(sqrt(dc * dc + ds * ds));
    } else {
      const double norm2 = ref_c * ref_c + ref_s * ref_s;
      da = static_cast<float>(fabs((ct * ref_c + st * ref_s) / norm2 - 1.0));
      db = static_cast<float>(fabs((st * ref_c - ct * ref_s) / norm2));
    }
    if (!(lane_small && dx < 1e6f && dy < 1e6f && da < 4.f && db < 4.f)) dx = dy = da = db = INFINITY;  // a far or non-finite particle: no patches
    for (int o = 32; o > 0; o >>= 1) {
      dx = fmaxf(dx, __shfl_xor(dx, o));
      dy = fmaxf(dy, __shfl_xor(dy, o));
      da = fmaxf(da, __shfl_xor(da, o));
      db = fmaxf(db, __shfl_xor(db, o));
    }
    if (lane == 0) {
      float* mine = s_bound + 4 * (tid >> 6);
      mine[0] = dx;
      mine[1] = dy;
      mine[2] = da;
      mine[3] = db;
    }
  }
  // ---- the scan through the reference pose, a beam per thread (every wave, a producer's too): end-point cell and reach per
  // axis of each beam of the planned groups, 16 bytes per beam in the patch buffers (nothing else uses them before the main
  // loop).  The plan below then works on these records: the double precision arithmetic of a workgroup's 1080 beams runs once,
  // 512 wide, instead of once per question a group's thread asks about its 8 beams, one beam after the other.
  const uint32_t planned = groups < kPatchPlanned ? groups : kPatchPlanned;
  int4* s_beam = reinterpret_cast<int4*>(smem + patch_base);
  {
    const double2* scan = reinterpret_cast<const double2*>(pts) + b_begin;
#pragma unroll
    for (uint32_t pass = 0; pass < (kPatchPlanned * 8 + kPatchBlock - 1) / kPatchBlock; ++pass) {
      const uint32_t b = pass * kPatchBlock + tid;
      if (b < planned * 8) {
        const double2 p = scan[b];
        const double sx = __builtin_fma(p.x, rc, __builtin_fma(-p.y, rs, rxm));
        const double sy = __builtin_fma(p.x, rs, __builtin_fma(p.y, rc, rym));
        const int cx = static_cast<int>(__builtin_bit_cast(uint64_t, sx) >> 32), cy = static_cast<int>(__builtin_bit_cast(uint64_t, sy) >> 32);
        const float reach_x = static_cast<float>(fabs(p.x * rc - p.y * rs)), reach_y = static_cast<float>(fabs(p.x * rs + p.y * rc));
        s_beam[b] = int4{cx, cy, __builtin_bit_cast(int, reach_x), __builtin_bit_cast(int, reach_y)};
      }
    }
  }
  __syncthreads();
  // ---- the plan: thread g looks at group g through the reference pose
  bool mine_fits = false;
  if (tid < planned) {
    float Dx = 0.f, Dy = 0.f, Da = 0.f, Db = 0.f;
    for (uint32_t k = 0; k < kConsumers; ++k) {
      Dx = fmaxf(Dx, s_bound[4 * k]);
      Dy = fmaxf(Dy, s_bound[4 * k + 1]);
      Da = fmaxf(Da, s_bound[4 * k + 2]);
      Db = fmaxf(Db, s_bound[4 * k + 3]);
    }
    // every float operation below may round down: scaled up by 1 + 2^-10 where it matters, and two cells of slack
    Dx = Dx * 1.001f + 2.f;
    Dy = Dy * 1.001f + 2.f;
    int4 rec[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rec[k] = s_beam[8 * tid + k];
    struct Range {
      int lo_x, hi_x, lo_y, hi_y;
      float reach_x, reach_y;
    };
    // Does the range fit a patch of PW x PH cells?  -> its origin
    auto fits_patch = [&](const Range& r, int PW, int PH, int& x0, int& y0) -> bool {
      float turn_x, turn_y;  // cells
      if (stats.isotropic_margin) {  // |q'| <= sqrt(max q'x^2 + max q'y^2)
        turn_x = turn_y = sqrtf(r.reach_x * r.reach_x + r.reach_y * r.reach_y) * 1.002f * (Da * 1.001f);
      } else {
        const float A = Da * 1.001f, Bv = Db * 1.001f, qx = r.reach_x * 1.001f, qy = r.reach_y * 1.001f;
        turn_x = (A * qx + Bv * qy) * 1.001f;
        turn_y = (Bv * qx + A * qy) * 1.001f;
      }
      const float mx = ceilf(Dx + turn_x), my = ceilf(Dy + turn_y);
      const bool fits = mx < 64.f && my < 64.f;  // false for NaN and infinity
      const int margin_x = fits ? static_cast<int>(mx) : 0, margin_y = fits ? static_cast<int>(my) : 0;
      x0 = r.lo_x - margin_x;
      y0 = (r.lo_y - margin_y) & ~7;
      return fits && r.hi_x + margin_x - x0 < PW && r.hi_y + margin_y - y0 < PH;
    };
    Range whole{INT_MAX, INT_MIN, INT_MAX, INT_MIN, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      whole.lo_x = min(whole.lo_x, rec[k].x);
      whole.hi_x = max(whole.hi_x, rec[k].x);
      whole.lo_y = min(whole.lo_y, rec[k].y);
      whole.hi_y = max(whole.hi_y, rec[k].y);
      whole.reach_x = fmaxf(whole.reach_x, __builtin_bit_cast(float, rec[k].z));
      whole.reach_y = fmaxf(whole.reach_y, __builtin_bit_cast(float, rec[k].w));
    }
    int x0a = 0, y0a = 0, x0b = 0, y0b = 0;
    uint32_t flags = 0u, first_b = 0u;
    if (fits_patch(whole, kPatchW, kPatchH, x0a, y0a)) {
      flags = 1u;
    } else if (stats.split_patches) {
      // split where the scan jumps: between the two consecutive beams whose end-points lie farthest apart (the first such pair)
      int widest_jump = -1, k_split = 4;
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const int jump = max(abs(rec[k].x - rec[k - 1].x), abs(rec[k].y - rec[k - 1].y));
        if (jump > widest_jump) {
          widest_jump = jump;
          k_split = k;
        }
      }
      Range ra{INT_MAX, INT_MIN, INT_MAX, INT_MIN, 0.f, 0.f}, rb = ra;
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // (k_split is a run-time value: both halves in one pass)
        Range& r = k < k_split ? ra : rb;
        r.lo_x = min(r.lo_x, rec[k].x);
        r.hi_x = max(r.hi_x, rec[k].x);
        r.lo_y = min(r.lo_y, rec[k].y);
        r.hi_y = max(r.hi_y, rec[k].y);
        r.reach_x = fmaxf(r.reach_x, __builtin_bit_cast(float, rec[k].z));
        r.reach_y = fmaxf(r.reach_y, __builtin_bit_cast(float, rec[k].w));
      }
      int xa, ya, xb, yb;
      if ((stats.split_patches & 1u) && fits_patch(ra, kPatchW / 2, kPatchH, xa, ya) && fits_patch(rb, kPatchW / 2, kPatchH, xb, yb)) flags = 1u | 2u;
      else if ((stats.split_patches & 2u) && fits_patch(ra, kPatchW, kPatchH / 2, xa, ya) && fits_patch(rb, kPatchW, kPatchH / 2, xb, yb)) flags = 1u | 4u;
      if (flags) {
        x0a = xa;
        y0a = ya;
        x0b = xb;
        y0b = yb;
        first_b = static_cast<uint32_t>(k_split);
      } else {
        x0a = y0a = 0;
      }
    } else {
      x0a = y0a = 0;
    }
    s_plan[tid] = int4{x0a, y0a | static_cast<int>(flags), x0b, y0b | static_cast<int>(first_b)};
    {
      const uint32_t ka = 0u - (static_cast<uint32_t>(x0a) & 0xFFFFFFu) * kPatchPitch - (static_cast<uint32_t>(y0a) << 1);
      // half B lives in columns 32 .. 63 (side by side) or in rows 32 .. 63 (stacked) of the same buffer
      const uint32_t kb = ((flags & 2u) ? (kPatchW / 2) * kPatchPitch : static_cast<uint32_t>(kPatchH)) -
                          (static_cast<uint32_t>(x0b) & 0xFFFFFFu) * kPatchPitch - (static_cast<uint32_t>(y0b) << 1);
      const uint32_t meta = flags == 0u ? 0u : ((flags & 6u) ? first_b : 8u);
      s_plan_k[tid] = int4{static_cast<int>(ka), static_cast<int>(meta), static_cast<int>(kb), 0};
    }
    mine_fits = flags != 0u;
  }
  // A workgroup with too few of its groups through a patch drops the machinery: no patches, no barriers, every look-up a
  // gather.  Not only dispersed sets: a gathered group INSIDE a patched workgroup costs 2.6x a patched one (the workgroup waits
  // for the gathers at its next barrier), one of an all-gathering workgroup 1.3x, so mixing pays only above ~2/3 fitting
  // (stats.loose_below, in 256ths: measured, profiles/r02_lf_series.txt).
  // (counted by hand: __syncthreads_count brings a static LDS variable with it, and this kernel addresses LDS from 0)
  uint32_t* s_count = reinterpret_cast<uint32_t*>(s_bound) + 4 * (kPatchBlock / 64);  // behind the bound's [8][4]
  {
    const uint32_t in_wave = static_cast<uint32_t>(__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine_fits)));
    if (lane == 0) s_count[tid >> 6] = in_wave;
  }
  __syncthreads();
  uint32_t fitting = 0;
  for (uint32_t k = 0; k < kPatchBlock / 64; ++k) fitting += s_count[k];
  const bool loose = __builtin_amdgcn_readfirstlane(fitting) * 256u < groups * stats.loose_below;
  struct Plan {  // scalars
    uint32_t ka;    // less the buffer's base
    uint32_t meta;  // 0: gathered, 8: one whole patch, 1 .. 7: two halves, the second one from this beam on
  };
  auto plan_of = [&](uint32_t g, Plan& plan) {  // g uniform; scalar results
    // (the groups beyond the plan read the entry behind it: zeros - gathered)
    const int2 e = *reinterpret_cast<const int2*>(s_plan_k + (g < kPatchPlanned ? g : kPatchPlanned));
    plan.ka = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(e.x));
    plan.meta = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(e.y));
  };

  // The launch's statistics (what the host picks the next launch's kernel by): groups planned and groups through a
  // patch, summed over a sample of the workgroups - every 16th of a large launch: thousands of atomic operations on one
  // address would cost more than the kernel's other work - ; the last one to report copies the running totals to the
  // host's mirror.  Called by the workgroup's last wave (the producer, if there is one).
  auto report = [&]() {
      const uint32_t stride = nblocks >= 256 ? 16u : 1u;
      if (stats.device && lane == 0 && block % stride == 0) {
        atomicAdd(stats.device + 0, static_cast<unsigned long long>(groups));
        atomicAdd(stats.device + 1, static_cast<unsigned long long>(loose ? 0u : fitting));
        __threadfence();
        const unsigned long long reporters = static_cast<unsigned long long>((nblocks + stride - 1) / stride) * gridDim.y;
        const unsigned long long ticket = atomicAdd(stats.device + 2, 1ull) + 1ull;
        if (ticket % reporters == 0 && stats.mirror) {
          const unsigned long long planned = atomicAdd(stats.device + 0, 0ull), through = atomicAdd(stats.device + 1, 0ull);
          stats.mirror[0] = planned;
          stats.mirror[1] = through;
          // the pair as ONE word (low halves) for the host's unsynchronised look between launches: never torn
          stats.mirror[2] = (planned & 0xFFFFFFFFull) | (through << 32);
        }
      }
  };
  const uint32_t last_planned = (groups < kPatchPlanned ? groups : kPatchPlanned) - 1u;
  if (producer) {
    if (loose) {
      report();
      if constexpr (kQueue) {
        if (stats.weight_sums) __syncthreads();  // (the consumers' barrier around the block's sum)
        continue;
      }
      return;
    }
    const int y_last = static_cast<int>((f.H + 7u) & ~7u);  // first row of the bottom border tiles
    // The patch of group g, clamped into the bordered table (whatever lies outside the grid reads the unknown entry, like
    // a clamped gather): this lane's column, tile row by tile row.  While the consumers work on group g the patch of g + 1
    // goes from registers to LDS and the ones of g + 2 and g + 3 are on their way.
    using Pieces = uint4[kPatchH / 8];
    // Fetches and stores are unconditional (a group without a patch, or past the last one, moves a patch nobody reads):
    // straight-line code lets the compiler count the loads in flight exactly, so that a store waits for ITS fetch only.
    // A whole patch: this lane's column x0A + lane, tile rows from y0A.  Halves side by side (flags & 2): the lanes from 32 on
    // fetch columns x0B + lane - 32, rows from y0B.  Halves one above the other (flags & 4): the pieces from 4 on come from
    // rows y0B + 8 (r - 4).
    auto fetch = [&](uint32_t g, Pieces& piece) {
      const int4 e = s_plan[g < last_planned ? g : last_planned];
      const int ya = __builtin_amdgcn_readfirstlane(e.y);
      const int x0a = __builtin_amdgcn_readfirstlane(e.x), y0a = ya & ~7;
      if ((ya & 6) == 0) {  // one whole patch (or none: then nobody reads it): this lane's column, scalar row offsets
        const int xu = x0a + static_cast<int>(lane) - static_cast<int>(kFastBias);
        // + 8: the border tile's share of palette_row_offset goes here, so that the vector offset - the one the buffer's
        // range check looks at - is never negative
        const uint32_t column = static_cast<uint32_t>(min(max(xu, -1), static_cast<int>(f.W)) + 8) << 4;
#pragma unroll
        for (int r = 0; r < kPatchH / 8; ++r) {
          const int yu = y0a + 8 * r - static_cast<int>(kFastBias);
          const int yc = min(max(yu, -8), y_last);  // scalar
          piece[r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, column, palette_row_offset(yc, f.pal_pitch) - 128u, 0));
        }
        return;
      }
      const int yb = __builtin_amdgcn_readfirstlane(e.w), x0b = __builtin_amdgcn_readfirstlane(e.z), y0b = yb & ~7;
      const bool side_by_side = (ya & 2) != 0, stacked = (ya & 4) != 0;  // scalar
      const bool half_b = side_by_side && lane >= static_cast<uint32_t>(kPatchW / 2);
      const int xu = (half_b ? x0b + static_cast<int>(lane) - kPatchW / 2 : x0a + static_cast<int>(lane)) - static_cast<int>(kFastBias);
      const uint32_t column = static_cast<uint32_t>(min(max(xu, -1), static_cast<int>(f.W)) + 8) << 4;
      // stacked halves: the lower half of the buffer (pieces 4 .. 7) holds the columns from x0B on
      const int xu_low = (stacked ? x0b : x0a) + static_cast<int>(lane) - static_cast<int>(kFastBias);
      const uint32_t column_low = stacked ? static_cast<uint32_t>(min(max(xu_low, -1), static_cast<int>(f.W)) + 8) << 4 : column;
#pragma unroll
      for (int r = 0; r < kPatchH / 8; ++r) {
        const bool low = r >= kPatchH / 16;
        const int yu_a = ((stacked && low) ? y0b + 8 * (r - kPatchH / 16) : y0a + 8 * r) - static_cast<int>(kFastBias);
        const int yu_b = y0b + 8 * r - static_cast<int>(kFastBias);
        const uint32_t row_a = palette_row_offset(min(max(yu_a, -8), y_last), f.pal_pitch) - 128u;  // scalar
        const uint32_t row_b = palette_row_offset(min(max(yu_b, -8), y_last), f.pal_pitch) - 128u;  // scalar
        piece[r] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (low ? column_low : column) + (half_b ? row_b : row_a), 0, 0));
      }
    };
    auto store = [&](uint32_t g, const Pieces& piece) {
      unsigned char* dst = smem + buffer_of(g) + lane * kPatchPitch;
#pragma unroll
      for (int r = 0; r < kPatchH / 8; ++r) *reinterpret_cast<uint4*>(dst + r * 16) = piece[r];
    };
    // Two register sets (even / odd groups): a patch is fetched two steps before it is stored (measured against one set and
    // one step: 0.478 vs 0.485 ms on a fixed cloud, tools/exp_lf_fixed.py).
    Pieces even, odd;
    fetch(0, even);
    store(0, even);
    fetch(1, odd);
    fetch(2, even);
    uint32_t g = 0;
    for (; g + 1 < groups; g += 2) {
      __syncthreads();  // the consumers are done with group g - 1: its buffer takes the patch of g + 1
      store(g + 1, odd);
      fetch(g + 3, odd);
      __syncthreads();
      store(g + 2, even);
      fetch(g + 4, even);
    }
    if (g < groups) __syncthreads();
    report();  // off the consumers' path: they are still at their last group
    if constexpr (kQueue) {
      if (stats.weight_sums) __syncthreads();  // (the consumers' barrier around the block's sum)
      continue;
    }
    return;
  }

  double acc = (f.prob || partial) ? 0.0 : 1.0;
  // A wave that holds a far particle goes through the exact evaluation group by group - by the main loop's own means: its end-point
  // constants are replaced by ones that put every end-point ON a cell boundary (the guard word comes out zero), so that every group is
  // marked for add_exact; what its look-ups read meanwhile (cell 0 of whatever patch, an address that may lie outside LDS: zero) is dropped.
  const bool fast = __builtin_amdgcn_ballot_w64(!lane_small) == 0;
  const double e_ct = fast ? ict : 0.0, e_st = fast ? ist : 0.0, e_xm = fast ? ixm : 1572864.0, e_ym = fast ? iym : 1572864.0;
  const int c_lo = static_cast<int>(kFastBias) - 1, x_hi = static_cast<int>(kFastBias + f.W), y_hi = static_cast<int>(kFastBias + f.H);
  const uint32_t row_bias = 4u - (kFastBias << 2);  // LDS byte address of the row entry = (biased y << 2) + row_bias
  // Palette addresses are kept 32 bits wide from the load on: the LDS look-ups SIGN-extend (ds_read_i16), the gathered ones zero-extend
  // (buffer_load_ushort) - the same value either way, as this kernel's palette lies below 32 KB of LDS (its workgroup memory is
  // pal_lds + 34 KB <= 64 KB).  Were both zero extensions, the compiler would merge the two sources as 16-bit values and every look-up
  // would pay a v_and_b32 to widen it again (8 per group on the LDS path); rounds 2 - 4 kept them apart by OR-ing the gathered value
  // with an opaque zero - an instruction that USES the value where it is loaded, i.e. a wait for every gather right behind its issue
  // instead of one step later: what made a gathered group cost 2.4 x a patched one.
  struct Lookups {
    uint32_t e[8];  // palette addresses (LDS byte addresses of the f64 terms)
    uint64_t redo;  // nonzero (the lanes with an end-point on a cell boundary): the group is added by add_exact instead
  };
  // The separately rounded evaluation, beam by beam with plain gathers.  It needs the pose as the reference holds it (not
  // pre-multiplied by 1 / res); this path runs for 4 in 2^32 end-points, so the pose is fetched again here rather than kept
  // alive - or parked in scratch memory: 16 bytes per lane written by every launch - across the main loop.  (The pointers go
  // through an empty asm statement so that the compiler does not fold the second fetch into the first.)
  // Likewise the lane's position in the order and its particle index: derived again where they are needed (the cold path,
  // the final store) instead of occupying three registers - or scratch - through the main loop.
  auto particle_again = [&](uint64_t& position) -> uint32_t {
    uint32_t zero = 0;
    const uint32_t* perm_again = perm;
    asm volatile("" : "+v"(zero), "+s"(perm_again));
    position = static_cast<uint64_t>(block) * kParticles + (tid + zero);
    return perm_again[position < n ? position : n - 1];
  };
  auto add_exact = [&](uint32_t b0, uint32_t count) {
    if (count == 0) return;
    const double4* pose_again = pose;
    asm volatile("" : "+s"(pose_again));
    uint64_t position;
    const Pose2 T_again = ordered_pose(f.world_to_field, pose_again, particle_again(position));
    const double ct = T_again.r.c, st = T_again.r.s, xt = T_again.x, yt = T_again.y;
    auto term = [&](uint32_t at) {
      const double px = pts[2 * at], py = pts[2 * at + 1];
      double vx = (px * ct - py * st + xt) * f.inv_resolution, vy = (px * st + py * ct + yt) * f.inv_resolution;
      floor_rd_2(vx, vy);
      return lf_palette_value(lf_palette_fetch(rsrc, f, floor_rd_result(vx), floor_rd_result(vy), kFastBiasX));
    };
    uint32_t b = b0;
    const uint32_t end = b0 + count;
#pragma unroll 1
    for (; b + 4 <= end; b += 4) {  // (groups start at multiples of 8: the blocks of four are the scan's)
      const double t0 = term(b), t1 = term(b + 1), t2 = term(b + 2), t3 = term(b + 3);
      acc += sum4(t0, t1, t2, t3);
    }
#pragma unroll 1
    for (; b < end; ++b) acc += term(b);
  };
  auto consume = [&](const Lookups& e, uint32_t b0) {
    if (e.redo != 0) {  // (a ballot: uniform)
      add_exact(b0, 8);
      return;
    }
    {
      const double t0 = lf_palette_value(e.e[0]), t1 = lf_palette_value(e.e[1]), t2 = lf_palette_value(e.e[2]), t3 = lf_palette_value(e.e[3]);
      acc += sum4(t0, t1, t2, t3);
    }
    {
      const double t4 = lf_palette_value(e.e[4]), t5 = lf_palette_value(e.e[5]), t6 = lf_palette_value(e.e[6]), t7 = lf_palette_value(e.e[7]);
      acc += sum4(t4, t5, t6, t7);
    }
  };
  // One step: the end-points of group g, then the sum of the group before it (its gathers, if any, had the end-point
  // arithmetic to arrive), then the look-ups of group g.  `redo`: 1 if the group has to be added by add_exact instead.
  // `rotor`: the buffer of group g on entry, that of group g + 1 on exit (the three buffers in turn).
  // `cursor`: the scan points of group g on entry, those of group g + 1 on exit (the constant address space: scalar loads - `pts` is no
  // kernel argument any more, of which the compiler knows that nobody writes there).
  typedef const __attribute__((address_space(4))) double* scan_ptr_t;
  // `carried`: the plan entry of group g on entry, that of group g + 1 on exit - read in the middle of the step, in front of the group's
  // look-ups: the wait for an LDS read is a wait for every LDS read issued before it, and a plan entry read at a step's start would
  // make the wave sit out the look-ups it has just issued in front of the barrier instead of behind the next end-points.
  // `in_loop`: the step is one of the main loop's, not one of the two in front of it - the gathered look-ups of the former extend with
  // zeros, those of the latter with the sign (the same value: palette addresses are below 2^15).  A lane's look-up registers are
  // loop-carried; while every value that flows into them is a ZERO extension of a 16-bit load - which is all an all-gathering workgroup
  // has - the compiler carries them as 16-bit values and widens each behind its load: an instruction that uses the gather right where it
  // is issued, i.e. a memory latency per step in the open instead of one hidden behind the next group's end-points.
  // `whole_patch`: the group is known to go through one whole patch (19 groups in 20).  The caller branches on the plan entry ONCE, in front
  // of the step, between two instances of it: inside one instance the three-way choice (whole patch / two halves / gathered) cost the
  // common group fourteen scalar instructions of predicate bookkeeping around its look-ups.
  auto step_as = [&](auto is_loose, auto add_before, auto in_loop, auto whole_patch, uint32_t g, Lookups& now, const Lookups& before, uint32_t& rotor,
                     scan_ptr_t& cursor, Plan& carried) {
    const Plan plan = carried;
    const uint32_t buffer = rotor;
    rotor = buffer + kPatchBytes == patch_base + kPatchBuffers * kPatchBytes ? patch_base : buffer + kPatchBytes;
    if constexpr (!decltype(is_loose)::value) {
      // A bare barrier: no wait for this wave's outstanding LDS reads (see kPatchBuffers).  What it orders: the producer's stores of
      // patch g (complete before ITS barrier: it keeps the fence) against the look-ups below.  The empty asm statements keep the
      // compiler from moving memory operations across it.
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const scan_ptr_t q = cursor;
    cursor = q + 16;
    int cx[8], cy[8];
    uint32_t lowest = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const double px = q[2 * k], py = q[2 * k + 1];
      const double sx = __builtin_fma(px, e_ct, __builtin_fma(-py, e_st, e_xm));
      const double sy = __builtin_fma(px, e_st, __builtin_fma(py, e_ct, e_ym));
      const uint64_t bx = __builtin_bit_cast(uint64_t, sx), by = __builtin_bit_cast(uint64_t, sy);
      if (k == 0) lowest = min(static_cast<uint32_t>(bx), static_cast<uint32_t>(by));
      else lowest = min3_u32(lowest, static_cast<uint32_t>(bx), static_cast<uint32_t>(by));
      cx[k] = static_cast<int>(bx >> 32);
      cy[k] = static_cast<int>(by >> 32);
    }
    int2 next_entry{0, 0};
    if constexpr (!decltype(is_loose)::value)
      next_entry = *reinterpret_cast<const int2*>(s_plan_k + (g + 1u < kPatchPlanned ? g + 1u : kPatchPlanned));  // (beyond the plan: the zero entry)
    if constexpr (decltype(add_before)::value) consume(before, b_begin + 8 * g - 8);
    if constexpr (!decltype(is_loose)::value) {
      carried.ka = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(next_entry.x));
      carried.meta = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(next_entry.y));
    }
    if constexpr (decltype(whole_patch)::value) {  // one constant for the eight look-ups
      const uint32_t K = buffer + plan.ka;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t at = patch_address(static_cast<uint32_t>(cx[k]), static_cast<uint32_t>(cy[k]), K);
        now.e[k] = static_cast<uint32_t>(static_cast<int32_t>(*reinterpret_cast<__attribute__((address_space(3))) const int16_t*>(static_cast<uintptr_t>(at))));
      }
    } else {
      // (one `else` for the two rare forms: the common one falls through a single scalar branch)
      if (plan.meta != 0u) {  // two halves: the beams from plan.meta on read the second one
        const uint32_t KA = buffer + plan.ka, KB = buffer + static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(s_plan_k[g].z));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t K = static_cast<uint32_t>(k) < plan.meta ? KA : KB;  // scalar: goes into the add as a scalar operand
          now.e[k] = static_cast<uint32_t>(static_cast<int32_t>(*reinterpret_cast<__attribute__((address_space(3))) const int16_t*>(
              static_cast<uintptr_t>(patch_address(static_cast<uint32_t>(cx[k]), static_cast<uint32_t>(cy[k]), K)))));
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int xc = med3_i32(cx[k], c_lo, x_hi), yc = med3_i32(cy[k], c_lo, y_hi);
          const uint32_t row = *reinterpret_cast<lds_u32_t*>(static_cast<uintptr_t>((static_cast<uint32_t>(yc) << 2) + row_bias));
          const auto raw = __builtin_amdgcn_raw_buffer_load_b16(rsrc, (static_cast<uint32_t>(xc) << 4) + row, 0, 0);
          if constexpr (decltype(in_loop)::value) now.e[k] = static_cast<uint32_t>(static_cast<uint16_t>(raw));
          else now.e[k] = static_cast<uint32_t>(static_cast<int32_t>(static_cast<int16_t>(raw)));
        }
      }
    }
    now.redo = __builtin_amdgcn_ballot_w64(lowest < 4u);
  };
  auto step = [&](auto is_loose, auto add_before, auto in_loop, uint32_t g, Lookups& now, const Lookups& before, uint32_t& rotor,
                  scan_ptr_t& cursor, Plan& carried) {
    if (!decltype(is_loose)::value && __builtin_expect(carried.meta == 8u, 1))
      step_as(is_loose, add_before, in_loop, std::true_type{}, g, now, before, rotor, cursor, carried);
    else
      step_as(is_loose, add_before, in_loop, std::false_type{}, g, now, before, rotor, cursor, carried);
  };
  auto run = [&](auto is_loose) {
    Lookups a, c;
    uint32_t rotor = patch_base;  // (group 0's buffer)
    scan_ptr_t cursor = (scan_ptr_t)(pts + 2 * b_begin);
    Plan carried{0u, 0u};
    if constexpr (!decltype(is_loose)::value) plan_of(0, carried);
    uint32_t g;
    if (groups & 1) {
      step(is_loose, std::false_type{}, std::false_type{}, 0, a, a, rotor, cursor, carried);
      g = 1;
    } else {
      step(is_loose, std::false_type{}, std::false_type{}, 0, c, c, rotor, cursor, carried);
      step(is_loose, std::true_type{}, std::false_type{}, 1, a, c, rotor, cursor, carried);
      g = 2;
    }
    for (; g < groups; g += 2) {  // `a` holds group g - 1
      step(is_loose, std::true_type{}, std::true_type{}, g, c, a, rotor, cursor, carried);
      step(is_loose, std::true_type{}, std::true_type{}, g + 1, a, c, rotor, cursor, carried);
    }
    consume(a, b_begin + 8 * groups - 8);
  };
  if (groups) {
    if (loose) run(std::true_type{});
    else run(std::false_type{});
  }
  add_exact(b_begin + 8 * groups, b_end - (b_begin + 8 * groups));
  uint64_t t_end;
  const uint32_t i_end = particle_again(t_end);
  double new_weight = 0.0;
  if (t_end < n) {
    if (partial) {
      partial[static_cast<size_t>(blockIdx.y) * n + t_end] = acc;
    } else {
      double old_weight = 1.0;
      if (unit_weights == 0u) old_weight = w[i_end];  // (uniform)
      new_weight = old_weight * (f.prob ? exp_out_of_line(acc) : acc);
      w[i_end] = new_weight;
    }
  }
  // The sum of the workgroup's new weights (fixed order: lanes, then waves), for the normalisation that follows: the weights
  // need no pass of their own to be added up (actions/normalize.hpp:70).  The producer, if any, has left: the barrier counts
  // the waves that are still there.
  if (stats.weight_sums) {
    double* s_sum = reinterpret_cast<double*>(s_bound);
    const double wave_total = wave_sum_f64(new_weight);
    if (lane == 0) s_sum[tid >> 6] = wave_total;
    __syncthreads();
    if (tid == 0) {
      double total = s_sum[0];
      for (uint32_t k = 1; k < kConsumers; ++k) total += s_sum[k];
      stats.weight_sums[block] = total;
    }
  }
  if constexpr (!kQueue) break;
  }  // the next block of the queue
}

__global__ __launch_bounds__(kBlock) void k_lf_combine(double* __restrict__ w, uint64_t n, const uint32_t* __restrict__ perm,
                                                       const double* __restrict__ partial, uint32_t segments, int mode) {
  // mode 0: w *= 1 + sum (likelihood field), 1: w *= exp(sum) (its log form), 2: w *= sum (beam model)
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= n) return;
  double acc = mode == 0 ? 1.0 : 0.0;
  for (uint32_t s = 0; s < segments; ++s) acc += partial[static_cast<size_t>(s) * n + t];
  const uint32_t i = perm[t];
  w[i] = w[i] * (mode == 1 ? exp(acc) : acc);
}

// [lf-kernels-end]
// -- spatial ordering of the particles --------------------------------------------------------------
// A full sort of (key, index) by the 20-bit ordering key, most significant digit first, two digits of 10 bits:
//   keys + block histograms of the HIGH digit (inside k_propagate, or k_order_keys)  ->  row scan (+ digit totals)  ->
//   stable scatter by the high digit into 1024 buckets (digit bases scanned per workgroup)  ->  every bucket sorted by the low
//   digit in LDS-resident counters, stable  ->  perm.     3 launches behind the keys (round 2's least-significant-digit-first
//   sort took 5: a bucket's low-digit pass needs no second table of block histograms and no second row scan).
// The order is the one by (key, particle index): identical in every run, so what depends on the ORDER of the lanes (the
// LF kernel's workgroup sums of the new weights) is reproducible bit for bit.
// Global atomics are slow on this part (~6 per ns, device scope resolves at the memory side), so there are none: block
// histograms in LDS, [digit][block] offset tables, per-wave counters.  Only 8 bytes per particle move; the kernels that
// consume the order gather the pose records through perm.
__device__ __forceinline__ double heading_delta(double c, double s, double c0, double s0) {
  return atan2(s * c0 - c * s0, c * c0 + s * s0);  // angle of (c,s) relative to (c0,s0), in (-pi, pi]
}

// bbox = {min x, max x, min y, max y, min dtheta, max dtheta}; dtheta relative to particle 0's heading.
__global__ __launch_bounds__(kBlock) void k_bbox_partials(Particles p, uint64_t n, double* __restrict__ partials, uint32_t stride) {
  __shared__ double scratch[(kBlock / 64) * 6];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + threadIdx.x * (kChunk / kBlock);
  const double c0 = p.pose[0].x, s0 = p.pose[0].y;
  double v[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) {
    const uint64_t i = base + k;
    if (i < n) {
      const double4 q = p.pose[i];
      const double x = q.z, y = q.w, d = heading_delta(q.x, q.y, c0, s0);
      v[0] = fmin(v[0], x);
      v[1] = fmax(v[1], x);
      v[2] = fmin(v[2], y);
      v[3] = fmax(v[3], y);
      v[4] = fmin(v[4], d);
      v[5] = fmax(v[5], d);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int k = 0; k < 6; k += 2) {
      v[k] = fmin(v[k], __shfl_down(v[k], o));
      v[k + 1] = fmax(v[k + 1], __shfl_down(v[k + 1], o));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int k = 0; k < 6; ++k) scratch[wave * 6 + k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < kBlock / 64; ++q)
      for (int k = 0; k < 6; k += 2) {
        v[k] = fmin(v[k], scratch[q * 6 + k]);
        v[k + 1] = fmax(v[k + 1], scratch[q * 6 + k + 1]);
      }
    for (int k = 0; k < 6; ++k) partials[static_cast<size_t>(k) * stride + blockIdx.x] = v[k];
  }
}

// Also turns the box into the key frame of the ordering keys (bins over the box instead of +-4 sigma).
__global__ __launch_bounds__(kBlock) void k_bbox_final(const double* __restrict__ partials, uint32_t count, uint32_t stride,
                                                       double* __restrict__ out, Particles p, KeyFrame* __restrict__ frame,
                                                       uint32_t layout) {
  __shared__ double scratch[(kBlock / 64) * 6];
  double v[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
  for (uint32_t b = threadIdx.x; b < count; b += kBlock)
    for (int k = 0; k < 6; k += 2) {
      v[k] = fmin(v[k], partials[static_cast<size_t>(k) * stride + b]);
      v[k + 1] = fmax(v[k + 1], partials[static_cast<size_t>(k + 1) * stride + b]);
    }
  for (int o = 32; o > 0; o >>= 1)
    for (int k = 0; k < 6; k += 2) {
      v[k] = fmin(v[k], __shfl_down(v[k], o));
      v[k + 1] = fmax(v[k + 1], __shfl_down(v[k + 1], o));
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
    for (int k = 0; k < 6; ++k) scratch[wave * 6 + k] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int q = 1; q < kBlock / 64; ++q)
      for (int k = 0; k < 6; k += 2) {
        v[k] = fmin(v[k], scratch[q * 6 + k]);
        v[k + 1] = fmax(v[k + 1], scratch[q * 6 + k + 1]);
      }
    for (int k = 0; k < 6; ++k) out[k] = v[k];
    auto inverse_span = [](double lo, double hi) { return hi > lo ? static_cast<float>(1.0 / (hi - lo)) : 0.f; };
    KeyFrame kf;
    kf.cx = 0.5 * (v[0] + v[1]);
    kf.cy = 0.5 * (v[2] + v[3]);
    kf.c0 = p.pose[0].x;
    kf.s0 = p.pose[0].y;
    kf.inv_x = inverse_span(v[0], v[1]);
    kf.inv_y = inverse_span(v[2], v[3]);
    kf.inv_t = inverse_span(v[4], v[5]);
    kf.t_off = static_cast<float>(0.5 * (v[4] + v[5]));
    kf.layout = layout;
    kf.bits_xy = 0;  // (the default split: 6 + 6 + 8)
    *frame = kf;
  }
}

// Keys + block histograms of the low digit as a pass of its own (stage-level calls, where k_propagate did not emit them).
__global__ __launch_bounds__(kWide) void k_order_keys(Particles p, uint64_t n, KeyFrame kf_value, const KeyFrame* __restrict__ kf_device,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ table, uint32_t nblocks) {
  __shared__ uint32_t hist[kSortDigits];
  for (uint32_t d = threadIdx.x; d < kSortDigits; d += kWide) hist[d] = 0;
  __syncthreads();
  const KeyFrame kf = kf_device ? *kf_device : kf_value;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kWide; ++k) {
    const uint64_t i = base + k * kWide + threadIdx.x;  // coalesced
    if (i < n) {
      const uint32_t key = order_key(p.pose[i], kf);
      keys[i] = key;
      atomicAdd(&hist[key >> kDigitBits], 1u);
    }
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < kSortDigits; d += kWide) table[static_cast<size_t>(d) * nblocks + blockIdx.x] = hist[d];
}

// The block histograms of keys that exist already (the ones the draw kernel predicted for the next cycle: DrawNormals::keys), for the ordering's
// first pass: k_order_keys without the keys.
__global__ __launch_bounds__(kWide) void k_key_hist(const uint32_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ table, uint32_t nblocks) {
  __shared__ uint32_t hist[kSortDigits];
  for (uint32_t d = threadIdx.x; d < kSortDigits; d += kWide) hist[d] = 0;
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kWide; ++k) {
    const uint64_t i = base + k * kWide + threadIdx.x;
    if (i < n) atomicAdd(&hist[keys[i] >> kDigitBits], 1u);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < kSortDigits; d += kWide) table[static_cast<size_t>(d) * nblocks + blockIdx.x] = hist[d];
}

// totals[d] = sum of row d of the [digit][block] table; one wave per digit.
// Row d of the table (block histograms of digit d) becomes its exclusive scan; totals[d] = the row's sum.  The scatter
// kernels add the sum of all smaller digits themselves (digit_bases): one launch less per pass.
__global__ __launch_bounds__(kBlock) void k_row_scan(uint32_t* __restrict__ table, uint32_t nblocks, uint32_t* __restrict__ totals) {
  const uint32_t d = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint32_t* row = table + static_cast<size_t>(d) * nblocks;
  uint32_t carry = 0;
  for (uint32_t start = 0; start < nblocks; start += 64) {
    const uint32_t b = start + lane;
    const uint32_t v = b < nblocks ? row[b] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o);
      if (lane >= static_cast<uint32_t>(o)) incl += up;
    }
    if (b < nblocks) row[b] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) totals[d] = carry;
}
// s_base[d] = sum of totals[q], q < d, for the kSortDigits digits; T threads, every one of them.
template <int T>
__device__ __forceinline__ void digit_bases(const uint32_t* __restrict__ totals, uint32_t* s_base, uint32_t* s_wave /* [T / 64] */) {
  constexpr int kPer = kSortDigits / T;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t v[kPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    v[k] = totals[threadIdx.x * kPer + k];
    sum += v[k];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o);
    if (lane >= static_cast<uint32_t>(o)) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t prefix = incl - sum;
  for (uint32_t q = 0; q < wave; ++q) prefix += s_wave[q];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    s_base[threadIdx.x * kPer + k] = prefix;
    prefix += v[k];
  }
  __syncthreads();
}

// The lanes of a wave that hold the same digit find each other with ten ballots: -> the mask of those lanes (for an invalid
// lane: the mask of the invalid ones, which nobody uses).
__device__ __forceinline__ unsigned long long same_digit_lanes(uint32_t digit, bool valid) {
  unsigned long long same = __builtin_amdgcn_ballot_w64(valid);
  same = valid ? same : ~same;
#pragma unroll
  for (uint32_t bit = 0; bit < kDigitBits; ++bit) {
    const bool set = (digit >> bit) & 1u;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(set);
    same &= set ? b : ~b;
  }
  return same;
}

// First pass: by the HIGH digit, stable, straight from the keys.  Every wave owns a contiguous eighth of the block and walks
// it 64 elements at a time: rank inside the group of equal digits = lanes below with the same digit, the wave's running count
// per digit lives in LDS; the waves' counts are then chained in wave order behind the block's offset of that digit.  No
// cross-wave step until every wave has ranked its part.  Out: (low digit << 32 | particle) at the element's place in its
// bucket - the buckets hold their particles in index order.
constexpr int kStable = 512;  // 8 waves x 256 elements: 48 KB of LDS counters per workgroup
constexpr uint32_t kSortHugeBucket = 65535;  // a bucket beyond this is not sorted by its low digit (k_sort_buckets)
__global__ __launch_bounds__(kStable) void k_sort_scatter_high(const uint32_t* __restrict__ keys, uint64_t n,
                                                              const uint32_t* __restrict__ table, uint32_t nblocks,
                                                              const uint32_t* __restrict__ totals, unsigned long long* __restrict__ out,
                                                              uint32_t* __restrict__ bases_out) {
  constexpr int kWaves = kStable / 64, kRounds = kChunk / kStable;
  __shared__ uint16_t wave_count[kWaves][kSortDigits];
  __shared__ uint32_t wave_base[kWaves][kSortDigits];
  __shared__ uint32_t base_of[kSortDigits], wave_sums[kWaves];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  digit_bases<kStable>(totals, base_of, wave_sums);
  if (blockIdx.x == 0) {  // for the second pass: every bucket's first position, and whether any bucket is beyond what one workgroup sorts
    bool huge = false;
    for (uint32_t d = threadIdx.x; d < kSortDigits; d += kStable) {
      bases_out[d] = base_of[d];
      huge = huge || totals[d] > kSortHugeBucket;
    }
    if (threadIdx.x == 0) bases_out[kSortDigits] = 0u;
    __syncthreads();
    if (huge) bases_out[kSortDigits] = 1u;
  }
  for (uint32_t d = threadIdx.x; d < kWaves * kSortDigits; d += kStable) (&wave_count[0][0])[d] = 0;
  __syncthreads();
  volatile uint16_t* mine = wave_count[wave];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + static_cast<uint64_t>(wave) * (kChunk / kWaves);
  uint32_t key[kRounds], rank[kRounds];
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const uint64_t e = base + static_cast<uint64_t>(k) * 64 + lane;
    const bool valid = e < n;
    key[k] = valid ? keys[e] : 0u;
    const uint32_t digit = key[k] >> kDigitBits;
    const unsigned long long same = same_digit_lanes(digit, valid);
    const uint32_t below = static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
    const uint32_t count = static_cast<uint32_t>(__popcll(same));
    const uint32_t before = valid ? mine[digit] : 0u;
    rank[k] = before + below;
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) mine[digit] = static_cast<uint16_t>(before + count);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < kSortDigits; d += kStable) {
    uint32_t run = base_of[d] + table[static_cast<size_t>(d) * nblocks + blockIdx.x];
#pragma unroll
    for (int q = 0; q < kWaves; ++q) {
      wave_base[q][d] = run;
      run += wave_count[q][d];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const uint64_t e = base + static_cast<uint64_t>(k) * 64 + lane;
    if (e < n)
      out[wave_base[wave][key[k] >> kDigitBits] + rank[k]] =
          (static_cast<unsigned long long>(key[k] & (kSortDigits - 1)) << 32) | static_cast<uint32_t>(e);
  }
}

// Second pass: one workgroup per bucket (= high digit) sorts the bucket's particles by the low digit, stable: the order is the
// sort by (key, particle index), the same in every run.  Two walks over the bucket, each wave over a contiguous eighth of it:
// the first counts the digits per wave (ballots, as above), a scan turns the counts into every wave's first destination per
// digit, the second walk ranks again and writes.  A wave's share of an ordinary bucket (an average one holds a thousand particles:
// 128 per wave) stays in REGISTERS between the two walks - up to four rounds of 64; round 4 read it from global memory twice and
// had every workgroup scan the 1024 digit totals for its bucket's first position (now left by the first pass: `bases`), 18.9 us of
// dependent latencies for 12 MB.  Buckets of any size (a degenerate set is one bucket): longer shares are walked from memory.
__global__ __launch_bounds__(kStable) void k_sort_buckets(const unsigned long long* __restrict__ in, const uint32_t* __restrict__ totals,
                                                         const uint32_t* __restrict__ bases, uint32_t* __restrict__ perm) {
  constexpr int kWaves = kStable / 64, kHeld = 4;
  __shared__ uint16_t run16[kWaves][kSortDigits];  // per wave and digit: counts, then (as offsets from the bucket's start) destinations
  __shared__ uint32_t wave_sums[kWaves];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // A bucket far beyond the average (a degenerate set: identical poses, a zero-covariance initialisation - all of a million
  // particles in ONE bucket, which one workgroup would walk twice on its own: milliseconds) is not sorted by its low digit at
  // all: it keeps the first pass's order (by particle index), and every workgroup of the launch copies a slice of it.  Only
  // locality depends on the order, never a result; the order stays a pure function of the keys (deterministic).
  if (bases[kSortDigits] != 0u) {  // (uniform: the first pass's flag)
#pragma unroll 1
    for (uint32_t d = 0; d < kSortDigits; ++d) {
      const uint32_t count = totals[d];
      if (count <= kSortHugeBucket) continue;  // (uniform)
      const uint32_t from = bases[d];
      for (uint32_t e = blockIdx.x * kStable + threadIdx.x; e < count; e += gridDim.x * kStable) perm[from + e] = static_cast<uint32_t>(in[from + e]);
    }
  }
  const uint32_t begin = bases[blockIdx.x], size = totals[blockIdx.x];
  if (size == 0 || size > kSortHugeBucket) return;  // (uniform)
  const uint32_t per_wave = ((size + kWaves - 1) / kWaves + 63u) & ~63u;
  const uint32_t first = min(wave * per_wave, size), last = min(first + per_wave, size);
  const bool held = per_wave <= 64u * kHeld;  // (uniform over the workgroup)
  // the wave's share, as far as it is held (loads in flight together: a round by itself would wait out a memory latency for 64 elements)
  unsigned long long mine_v[kHeld];
#pragma unroll
  for (int r = 0; r < kHeld; ++r) {
    const uint32_t at = first + 64u * r + lane;
    mine_v[r] = (held && at < last) ? in[begin + at] : 0ull;
  }
  {
    uint32_t* zero = reinterpret_cast<uint32_t*>(&run16[0][0]);
    for (uint32_t d = threadIdx.x; d < kWaves * kSortDigits / 2; d += kStable) zero[d] = 0u;
  }
  __syncthreads();
  volatile uint16_t* mine = run16[wave];
  auto count_round = [&](uint32_t digit, bool valid) {
    const unsigned long long same = same_digit_lanes(digit, valid);
    if (valid && (same & ((1ull << lane) - 1ull)) == 0) mine[digit] = static_cast<uint16_t>(mine[digit] + static_cast<uint32_t>(__popcll(same)));
    __builtin_amdgcn_wave_barrier();
  };
  if (held) {
#pragma unroll
    for (int r = 0; r < kHeld; ++r) {
      const uint32_t at = first + 64u * r;
      if (at >= last) break;  // (uniform)
      count_round(static_cast<uint32_t>(mine_v[r] >> 32), at + lane < last);
    }
  } else {
    for (uint32_t at0 = first; at0 < last; at0 += 256) {
      uint32_t digit4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t at = at0 + 64u * r;
        digit4[r] = at + lane < last ? static_cast<uint32_t>(in[begin + at + lane] >> 32) : 0u;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t at = at0 + 64u * r;
        if (at >= last) break;  // (uniform)
        count_round(digit4[r], at + lane < last);
      }
    }
  }
  __syncthreads();
  // counts -> first destinations (offsets from the bucket's start; a bucket holds at most 65535): digits in order, inside a digit the
  // waves in order
  {
    constexpr int kPer = kSortDigits / kStable;
    uint32_t total[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      total[k] = 0;
      for (int q = 0; q < kWaves; ++q) total[k] += run16[q][threadIdx.x * kPer + k];
      sum += total[k];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o);
      if (lane >= static_cast<uint32_t>(o)) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t prefix = incl - sum;
    for (uint32_t q = 0; q < wave; ++q) prefix += wave_sums[q];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      uint32_t at = prefix;
      for (int q = 0; q < kWaves; ++q) {
        const uint32_t c = run16[q][threadIdx.x * kPer + k];
        run16[q][threadIdx.x * kPer + k] = static_cast<uint16_t>(at);
        at += c;
      }
      prefix += total[k];
    }
  }
  __syncthreads();
  auto place_round = [&](unsigned long long v, bool valid) {
    const uint32_t digit = static_cast<uint32_t>(v >> 32);
    const unsigned long long same = same_digit_lanes(digit, valid);
    const uint32_t below = static_cast<uint32_t>(__popcll(same & ((1ull << lane) - 1ull)));
    const uint32_t to = valid ? mine[digit] : 0u;
    if (valid) perm[begin + to + below] = static_cast<uint32_t>(v);
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) mine[digit] = static_cast<uint16_t>(to + static_cast<uint32_t>(__popcll(same)));
    __builtin_amdgcn_wave_barrier();
  };
  if (held) {
#pragma unroll
    for (int r = 0; r < kHeld; ++r) {
      const uint32_t at = first + 64u * r;
      if (at >= last) break;  // (uniform)
      place_round(mine_v[r], at + lane < last);
    }
  } else {
    for (uint32_t at0 = first; at0 < last; at0 += 256) {
      unsigned long long v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t at = at0 + 64u * r;
        v4[r] = at + lane < last ? in[begin + at + lane] : 0ull;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t at = at0 + 64u * r;
        if (at >= last) break;  // (uniform)
        place_round(v4[r], at + lane < last);
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_u32_chunk_sum(const uint32_t* __restrict__ v, uint32_t n, uint32_t* __restrict__ chunk_sum) {
  __shared__ uint32_t s_wave[kBlock / 64];
  const uint32_t base = blockIdx.x * kChunk + threadIdx.x * (kChunk / kBlock);
  uint32_t local = 0;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) local += (base + k < n) ? v[base + k] : 0u;
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// In-place exclusive scan of v within each chunk, offset by chunk_offset[chunk].
__global__ __launch_bounds__(kBlock) void k_u32_exclusive_apply(uint32_t* __restrict__ v, uint32_t n,
                                                                const uint32_t* __restrict__ chunk_offset) {
  __shared__ uint32_t s_wave[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t base = blockIdx.x * kChunk + threadIdx.x * (kChunk / kBlock);
  uint32_t loc[kChunk / kBlock];
  uint32_t run = 0;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) {
    loc[k] = run;
    run += (base + k < n) ? v[base + k] : 0u;
  }
  uint32_t incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t prefix = chunk_offset[blockIdx.x];
  for (int q = 0; q < wave; ++q) prefix += s_wave[q];
  prefix += incl - run;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k)
    if (base + k < n) v[base + k] = prefix + loc[k];
}

// ---- K3 weight sums / normalize ----------------------------------------------------------------------
// Each workgroup owns chunk b = [b*kChunk, (b+1)*kChunk): thread t holds elements t*8 .. t*8+7.
constexpr int kItems = kChunk / kBlock;  // 8

// A chunk between global memory and the threads' items, through LDS.  A thread reading its own eight consecutive doubles asks
// the vector-memory pipe for 64 different lines per instruction (one per lane, eight instructions per line: measured 1.1 TB/s at
// 10M particles); here the global side moves 16 bytes per lane with consecutive lanes on consecutive addresses, and the items
// are read back from an LDS copy padded by one double per eight (stride 9 doubles = 18 banks: conflict free for the 32 lanes
// of a b64 pass).  Which thread adds which elements in which order - and with it every bit of the sums and of the CDF - is
// unchanged.  Elements beyond n read as 0.
constexpr int kChunkPadded = kChunk + kChunk / kItems;
__device__ __forceinline__ int chunk_slot(int e) { return e + (e >> 3); }
__device__ __forceinline__ void chunk_items_load(const double* __restrict__ a, uint64_t n, double* lds, double (&x)[kItems],
                                                 uint32_t chunk = blockIdx.x) {
  const uint64_t base = static_cast<uint64_t>(chunk) * kChunk;
  if ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) {
#pragma unroll
    for (int k = 0; k < kChunk / 2 / kBlock; ++k) {
      const int e = 2 * (k * kBlock + static_cast<int>(threadIdx.x));
      const uint64_t i = base + e;
      double2 v{0.0, 0.0};
      if (i + 1 < n) v = *reinterpret_cast<const double2*>(a + i);
      else if (i < n) v.x = a[i];
      lds[chunk_slot(e)] = v.x;  // e is even: e + 1 lies in the same group of eight
      lds[chunk_slot(e) + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int e = k * kBlock + static_cast<int>(threadIdx.x);
      lds[chunk_slot(e)] = base + e < n ? a[base + e] : 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kItems; ++k) x[k] = lds[threadIdx.x * (kItems + 1) + k];
}
// The way back: the threads' items to a[chunk] (elements below n only).  The LDS copy may be the one chunk_items_load filled: a
// thread overwrites the slots it read itself.
__device__ __forceinline__ void chunk_items_store(double* __restrict__ a, uint64_t n, double* lds, const double (&x)[kItems],
                                                  uint32_t chunk = blockIdx.x) {
  const uint64_t base = static_cast<uint64_t>(chunk) * kChunk;
#pragma unroll
  for (int k = 0; k < kItems; ++k) lds[threadIdx.x * (kItems + 1) + k] = x[k];
  __syncthreads();
  if ((reinterpret_cast<uintptr_t>(a) & 15u) == 0) {
#pragma unroll
    for (int k = 0; k < kChunk / 2 / kBlock; ++k) {
      const int e = 2 * (k * kBlock + static_cast<int>(threadIdx.x));
      const uint64_t i = base + e;
      const double2 v{lds[chunk_slot(e)], lds[chunk_slot(e) + 1]};
      if (i + 1 < n) *reinterpret_cast<double2*>(a + i) = v;
      else if (i < n) a[i] = v.x;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int e = k * kBlock + static_cast<int>(threadIdx.x);
      if (base + e < n) a[base + e] = lds[chunk_slot(e)];
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_chunk_sum(const double* __restrict__ w, uint64_t n, double* __restrict__ partials) {
  __shared__ double scratch[kBlock / 64];
  __shared__ double s_items[kChunkPadded];
  double x[kItems];
  chunk_items_load(w, n, s_items, x);
  double v[1] = {0.0};
#pragma unroll
  for (int k = 0; k < kItems; ++k) v[0] += x[k];  // (an element beyond n adds + 0.0: the sum keeps its bits)
  block_reduce<1>(v, scratch);
  if (threadIdx.x == 0) partials[blockIdx.x] = v[0];
}

// out[k] = sum_b partials[k][b] (partials laid out [rows][stride]); one workgroup per row k = blockIdx.x, fixed order.
// host_mirror (optional): a mapped pinned-host copy of the result, written by the kernel itself — the host reads it after
// the stream synchronisation, no separate device-to-host copy (a blit kernel of its own) is enqueued.
__device__ __forceinline__ double row_total(const double* __restrict__ row, uint32_t count, double* scratch /* [kBlock/64] */) {
  double v[1] = {0.0};
  for (uint32_t b = threadIdx.x; b < count; b += kBlock) v[0] += row[b];
  block_reduce<1>(v, scratch);
  return v[0];  // valid in thread 0
}
// done (optional, Completion): the launch's last workgroup to finish stores done.seq to a word of mapped host memory, behind
// everything the launch mirrored - the host can wait for that word instead of the stream's completion signal.
__global__ __launch_bounds__(kBlock) void k_final_rows(const double* __restrict__ partials, uint32_t count, uint32_t stride,
                                                       double* __restrict__ out, double* __restrict__ host_mirror, Completion done) {
  __shared__ double scratch[kBlock / 64];
  const uint32_t k = blockIdx.x;
  const double total = row_total(partials + static_cast<size_t>(k) * stride, count, scratch);
  if (threadIdx.x == 0) {
    out[k] = total;
    if (host_mirror) host_mirror[k] = total;
    if (done.host_flag) {
      __threadfence_system();  // this workgroup's mirrored value is visible to the host before its ticket counts
      const unsigned long long ticket = atomicAdd(done.d_ticket, 1ull);
      if (ticket % gridDim.x == gridDim.x - 1) {
        __threadfence_system();
        __hip_atomic_store(done.host_flag, done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// sum_partials != nullptr: the factor is the total of these chunk sums, added up by every workgroup exactly as
// k_final_rows does (same strides, same reduction tree, same bits) instead of being read from *d_factor — one launch
// less on the cycle's critical path; workgroup 0 stores the total to d_sum_out (and its host mirror).
__global__ __launch_bounds__(kBlock) void k_normalize(double* __restrict__ w, uint64_t n, const double* __restrict__ d_factor,
                                                      double* __restrict__ chunk_sum, double* __restrict__ chunk_sumsq,
                                                      const double* __restrict__ sum_partials, uint32_t sum_count,
                                                      double* __restrict__ d_sum_out, double* __restrict__ sum_mirror, int store_weights) {
  __shared__ double scratch[(kBlock / 64) * 2];
  __shared__ double s_factor;
  if (sum_partials) {
    double t[1] = {0.0};
    for (uint32_t b = threadIdx.x; b < sum_count; b += kBlock) t[0] += sum_partials[b];
    block_reduce<1>(t, scratch);
    if (threadIdx.x == 0) {
      s_factor = t[0];
      if (blockIdx.x == 0) {
        d_sum_out[0] = t[0];
        if (sum_mirror) sum_mirror[0] = t[0];
      }
    }
    __syncthreads();
  }
  const double factor = sum_partials ? s_factor : *d_factor;
  const bool skip = fabs(factor - 1.0) < DBL_EPSILON;  // normalize.hpp:73
  __shared__ double s_items[kChunkPadded];
  double x[kItems];
  chunk_items_load(w, n, s_items, x);
  double v[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < kItems; ++k) {  // (an element beyond n is 0: 0 / factor = 0 adds nothing)
    if (!skip) x[k] = x[k] / factor;
    v[0] += x[k];
    v[1] += x[k] * x[k];
  }
  // store_weights == 0 (a cycle that resamples at once): the normalised weights themselves are not written - the CDF kernel that follows
  // divides again (k_cdf, d_factor: the same division, the same bits), nothing else reads them
  if (!skip && store_weights) chunk_items_store(w, n, s_items, x);
  block_reduce<2>(v, scratch);
  if (threadIdx.x == 0) {
    chunk_sum[blockIdx.x] = v[0];
    chunk_sumsq[blockIdx.x] = v[1];
  }
}

// ---- K5 CDF -------------------------------------------------------------------------------------------
// Exclusive scan of `count` chunk sums by one workgroup (sequential carry across 256-wide tiles);
// total[0] = grand total.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scan_chunks(const T* __restrict__ chunk_sum, uint32_t count, T* __restrict__ chunk_offset,
                                                        T* __restrict__ total, const T* __restrict__ base_value) {
  __shared__ T s_wave[kBlock / 64];
  __shared__ T s_carry;
  if (threadIdx.x == 0) s_carry = base_value ? *base_value : T(0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t start = 0; start < count; start += kBlock) {
    const uint32_t i = start + threadIdx.x;
    const T v = i < count ? chunk_sum[i] : T(0);
    T incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const T up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    T wave_prefix = T(0);
    for (int q = 0; q < wave; ++q) wave_prefix += s_wave[q];
    const T carry = s_carry;
    if (i < count) chunk_offset[i] = carry + wave_prefix + (incl - v);
    __syncthreads();
    if (threadIdx.x == kBlock - 1) s_carry = carry + wave_prefix + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = s_carry;
}

// Also writes the sampled levels of the search tree (CdfTree): element e is entry (e + 1) / 16^l - 1 of level l whenever
// 16^l divides e + 1, and the last element closes the last (partial) group of every level.
// The offset of chunk `me` exactly as k_scan_chunks<double> computes it (same tiles, same order, same bits), replayed by
// a whole workgroup for itself: for a few hundred chunks this is cheaper than a single-workgroup kernel in between.
__device__ __forceinline__ double chunk_offset_replay(const double* __restrict__ chunk_sum, uint32_t count, uint32_t me) {
  __shared__ double r_wave[kBlock / 64];
  __shared__ double r_carry, r_result;
  if (threadIdx.x == 0) r_carry = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t start = 0; start <= me; start += kBlock) {
    const uint32_t i = start + threadIdx.x;
    const double v = i < count ? chunk_sum[i] : 0.0;
    double incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) r_wave[wave] = incl;
    __syncthreads();
    double wave_prefix = 0.0;
    for (int q = 0; q < wave; ++q) wave_prefix += r_wave[q];
    const double carry = r_carry;
    if (i == me) r_result = carry + wave_prefix + (incl - v);
    __syncthreads();
    if (threadIdx.x == kBlock - 1) r_carry = carry + wave_prefix + incl;
    __syncthreads();
  }
  return r_result;
}

// What the last normalisation kernel of a cycle leaves for the policies: the totals of the normalised weights and of their
// squares (same rows, same reduction tree as k_final_rows) and, when the cycle takes no host-side decision, one step of
// ThrunRecoveryProbabilityEstimator (thrun_recovery_probability_estimator.hpp:69-89, exponential_filter.hpp:32-44):
// policy = {slow, fast, p}; both filters advance on average = norm_sum / n; p = clamp(1 - fast / slow, 0, 1) (0 while
// |slow| < eps); if this cycle resamples and p > 0 the filters are reset (amcl_core.hpp:184-186).
struct NormFinalize {
  const double* chunk_sum;    // [chunks] sums of the normalised weights per chunk (k_normalize)
  const double* chunk_sumsq;  // [chunks]
  uint32_t chunks;
  double* d_sums;             // d_sums[0] = norm_sum, d_sums[1] = norm_sumsq
  double* sums_mirror;        // optional host mirror of the two
  int policy;                 // run the recovery estimator
  uint64_t n;
  double alpha_slow, alpha_fast;
  int resampling;
  double* d_policy;           // {slow, fast, p}
  double* policy_mirror;      // optional: receives p at [2]
};
__device__ __forceinline__ void recovery_policy_step(double norm_sum, const NormFinalize& f) {
  const double average = norm_sum / static_cast<double>(f.n);
  double slow = f.d_policy[0], fast = f.d_policy[1];
  fast += (fast == 0.) ? average : f.alpha_fast * (average - fast);
  slow += (slow == 0.) ? average : f.alpha_slow * (average - slow);
  double p = 0.0;
  if (fabs(slow) >= 2.220446049250313e-16) p = fmin(fmax(1.0 - fast / slow, 0.0), 1.0);
  if (f.resampling && p > 0.0) slow = fast = 0.0;
  f.d_policy[0] = slow;
  f.d_policy[1] = fast;
  f.d_policy[2] = p;
  if (f.policy_mirror) f.policy_mirror[2] = p;
}
// Whole workgroup; scratch [kBlock/64].
__device__ __forceinline__ void norm_finalize(const NormFinalize& f, double* scratch) {
  const double sum = row_total(f.chunk_sum, f.chunks, scratch);
  const double sumsq = row_total(f.chunk_sumsq, f.chunks, scratch);
  if (threadIdx.x == 0) {
    f.d_sums[0] = sum;
    f.d_sums[1] = sumsq;
    if (f.sums_mirror) {
      f.sums_mirror[0] = sum;
      f.sums_mirror[1] = sumsq;
    }
    if (f.policy) recovery_policy_step(sum, f);
  }
}
__global__ __launch_bounds__(kBlock) void k_norm_finalize(NormFinalize f) {
  __shared__ double scratch[kBlock / 64];
  norm_finalize(f, scratch);
}

__global__ __launch_bounds__(kBlock) void k_cdf(const double* __restrict__ w, uint64_t n, const double* __restrict__ chunk_offset,
                                                double* __restrict__ cdf, double* __restrict__ total, CdfTree tree,
                                                double* __restrict__ levels, const double* __restrict__ chunk_sum_to_scan,
                                                uint32_t chunk_count, NormFinalize fin, const double* __restrict__ d_factor) {
  __shared__ double s_wave[kBlock / 64];
  if (fin.d_sums && blockIdx.x == 0) {  // the totals (and the recovery estimator) ride on the first workgroup
    norm_finalize(fin, s_wave);
    __syncthreads();
  }
  const double my_offset = chunk_sum_to_scan ? chunk_offset_replay(chunk_sum_to_scan, chunk_count, blockIdx.x) : chunk_offset[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + threadIdx.x * kItems;
  __shared__ double s_items[kChunkPadded];
  double loc[kItems];
  chunk_items_load(w, n, s_items, loc);
  if (d_factor) {  // w holds the weights as the reweight left them: actions::normalize's division here (k_normalize did not store it)
    const double factor = *d_factor;
    if (!(fabs(factor - 1.0) < DBL_EPSILON)) {  // normalize.hpp:73
#pragma unroll
      for (int k = 0; k < kItems; ++k) loc[k] = loc[k] / factor;
    }
  }
  double run = 0.0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    run += loc[k];
    loc[k] = run;
  }
  double incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  double prefix = my_offset;
  for (int q = 0; q < wave; ++q) prefix += s_wave[q];
  prefix += incl - run;
#pragma unroll
  for (int k = 0; k < kItems; ++k) loc[k] = prefix + loc[k];
  chunk_items_store(cdf, n, s_items, loc);
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t i = base + k;
    if (i < n) {
      const double v = loc[k];
      if (levels) {
        uint64_t q = i + 1;
        for (int l = 0; l < tree.depth && (q & 15) == 0; ++l) {
          q >>= 4;
          levels[tree.offset[l] + q - 1] = v;
        }
      }
      if (i == n - 1) {
        *total = v;
        if (levels)
          for (int l = 0; l < tree.depth; ++l) levels[tree.offset[l] + tree.size[l] - 1] = v;
      }
    }
  }
}

// ---- K3 + K5 in one pass: normalise, totals, recovery estimator and CDF (sets of up to kScanFusedMaxChunks chunks) ----------------
// k_normalize and k_cdf were two launches of a few microseconds of bandwidth each (16 MB apiece at 1M particles) and two ramps: one
// kernel reads the weights ONCE and leaves w / total, the chunk sums, the totals of the normalised weights (+ one step of the recovery
// estimator) and the CDF with its search tree.  What a workgroup needs from the others - the sums of the chunks before its own - travels
// INSIDE the launch (cdna_hip_programming.md section 6, guideline 16, form R2: the data is the flag): a chunk's two sums are published as
// four 8-byte granules {epoch : 32 | half of the double : 32}, each ONE relaxed agent-scope store (sc1: written through, no release
// fence, no L2 write-back); a reader re-reads a granule (relaxed agent-scope loads) until its tag is this launch's epoch.  A workgroup
// publishes before it waits for anything, and the chunk it works on is the TICKET it draws (atomicInc, wrapping to zero behind the
// launch's last workgroup), not blockIdx: every chunk it waits for belongs to a workgroup that is already running - forward progress
// holds for any dispatch order, resident or not.  Epochs never repeat within 2^32 launches of a context and the granules start at zero:
// nothing is reset between launches.
// Same threads, same elements, same order of additions as k_normalize + k_cdf (the replayed chunk offsets, norm_finalize's rows):
// every bit of the weights, the sums, the recovery probability and the CDF is the two-kernel path's (tests compare them for equality).
constexpr uint32_t kScanFusedMaxChunks = 4 * kBlock;  // = k_cdf's replay bound; beyond it a workgroup's look-back would be most of its time
struct ScanState {
  unsigned int* ticket;           // one word, zero between launches
  unsigned long long* granules;   // [chunks][4]: sum lo, sum hi, sum of squares lo, hi
  uint32_t epoch;                 // != 0
};
__device__ __forceinline__ void publish_f64(unsigned long long* g, uint32_t epoch, double v) {
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v), tag = static_cast<unsigned long long>(epoch) << 32;
  __hip_atomic_store(g + 0, tag | (bits & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double await_f64(unsigned long long* g, uint32_t epoch) {
  for (;;) {
    const unsigned long long lo = __hip_atomic_load(g + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (static_cast<uint32_t>(lo >> 32) == epoch && static_cast<uint32_t>(hi >> 32) == epoch)
      return __builtin_bit_cast(double, (hi << 32) | (lo & 0xFFFFFFFFull));
    __builtin_amdgcn_s_sleep(2);
  }
}
// chunk_offset_replay on values the threads hold: v[t] = the sum of chunk t * kBlock + threadIdx.x (0 from chunk `me` on).
__device__ __forceinline__ double chunk_offset_replay_held(const double (&held)[kScanFusedMaxChunks / kBlock], uint32_t me) {
  __shared__ double h_wave[kBlock / 64];
  __shared__ double h_carry, h_result;
  if (threadIdx.x == 0) h_carry = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (uint32_t t = 0; t < kScanFusedMaxChunks / kBlock; ++t) {
    const uint32_t start = t * kBlock;
    if (start > me) break;  // (uniform)
    const uint32_t i = start + threadIdx.x;
    const double v = held[t];
    double incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) h_wave[wave] = incl;
    __syncthreads();
    double wave_prefix = 0.0;
    for (int q = 0; q < wave; ++q) wave_prefix += h_wave[q];
    const double carry = h_carry;
    if (i == me) h_result = carry + wave_prefix + (incl - v);
    __syncthreads();
    if (threadIdx.x == kBlock - 1) h_carry = carry + wave_prefix + incl;
    __syncthreads();
  }
  return h_result;
}
__global__ __launch_bounds__(kBlock) void k_normalize_cdf(double* __restrict__ w, uint64_t n, const double* __restrict__ d_factor,
                                                          const double* __restrict__ sum_partials, uint32_t sum_count,
                                                          double* __restrict__ d_sum_out, double* __restrict__ sum_mirror,
                                                          double* __restrict__ chunk_sum, double* __restrict__ chunk_sumsq,
                                                          uint32_t chunk_count, int write_weights, double* __restrict__ cdf,
                                                          double* __restrict__ total, CdfTree tree, double* __restrict__ levels,
                                                          NormFinalize fin, ScanState state) {
  __shared__ double scratch[(kBlock / 64) * 2];
  __shared__ double s_factor, s_own[2];
  __shared__ uint32_t s_chunk;
  __shared__ double s_items[kChunkPadded];
  if (threadIdx.x == 0) s_chunk = atomicInc(state.ticket, gridDim.x - 1u);
  if (sum_partials) {  // the factor: the total of these sums, added up by every workgroup as k_final_rows does (k_normalize)
    double t[1] = {0.0};
    for (uint32_t b = threadIdx.x; b < sum_count; b += kBlock) t[0] += sum_partials[b];
    block_reduce<1>(t, scratch);
    if (threadIdx.x == 0) s_factor = t[0];
  }
  __syncthreads();
  const uint32_t me = s_chunk;
  const double factor = sum_partials ? s_factor : *d_factor;
  if (sum_partials && me == 0 && threadIdx.x == 0) {
    d_sum_out[0] = factor;
    if (sum_mirror) sum_mirror[0] = factor;
  }
  const bool skip = fabs(factor - 1.0) < DBL_EPSILON;  // normalize.hpp:73
  double x[kItems];
  chunk_items_load(w, n, s_items, x, me);
  double v[2] = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    if (!skip) x[k] = x[k] / factor;
    v[0] += x[k];
    v[1] += x[k] * x[k];
  }
  block_reduce<2>(v, scratch);
  unsigned long long* mine = state.granules + 4ull * me;
  if (threadIdx.x == 0) {  // published before this workgroup waits for anybody
    publish_f64(mine + 0, state.epoch, v[0]);
    publish_f64(mine + 2, state.epoch, v[1]);
    chunk_sum[me] = v[0];
    chunk_sumsq[me] = v[1];
    s_own[0] = v[0];
    s_own[1] = v[1];
  }
  if (!skip && write_weights) chunk_items_store(w, n, s_items, x, me);
  // the chunk's own scan (k_cdf)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ double s_wave[kBlock / 64];
  double run = 0.0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    run += x[k];
    x[k] = run;
  }
  double incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  // the sums of the chunks before this one
  double held[kScanFusedMaxChunks / kBlock];
#pragma unroll
  for (uint32_t t = 0; t < kScanFusedMaxChunks / kBlock; ++t) {
    const uint32_t i = t * kBlock + threadIdx.x;
    held[t] = i < me ? await_f64(state.granules + 4ull * i, state.epoch) : 0.0;
  }
  __syncthreads();  // (s_wave, s_own)
#pragma unroll
  for (uint32_t t = 0; t < kScanFusedMaxChunks / kBlock; ++t)  // (the replay's scan runs over this chunk's own sum as well: same bits as k_cdf's)
    if (t * kBlock + threadIdx.x == me) held[t] = s_own[0];
  const double my_offset = chunk_offset_replay_held(held, me);
  double prefix = my_offset;
  for (int q = 0; q < wave; ++q) prefix += s_wave[q];
  prefix += incl - run;
#pragma unroll
  for (int k = 0; k < kItems; ++k) x[k] = prefix + x[k];
  __syncthreads();  // (s_items: chunk_items_store of the weights may still be read)
  chunk_items_store(cdf, n, s_items, x, me);
  const uint64_t base = static_cast<uint64_t>(me) * kChunk + threadIdx.x * kItems;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t i = base + k;
    if (i < n) {
      const double c = x[k];
      if (levels) {
        uint64_t q = i + 1;
        for (int l = 0; l < tree.depth && (q & 15) == 0; ++l) {
          q >>= 4;
          levels[tree.offset[l] + q - 1] = c;
        }
      }
      if (i == n - 1) {
        *total = c;
        if (levels)
          for (int l = 0; l < tree.depth; ++l) levels[tree.offset[l] + tree.size[l] - 1] = c;
      }
    }
  }
  // The last chunk's workgroup has every chunk's sum at hand: the totals of the normalised weights and the recovery estimator
  // (norm_finalize: thread t adds the chunks t, t + kBlock, ... in order, then the block reduction).
  if (fin.d_sums && me + 1 == chunk_count) {  // (uniform)
    double sq[kScanFusedMaxChunks / kBlock];
#pragma unroll
    for (uint32_t t = 0; t < kScanFusedMaxChunks / kBlock; ++t) {
      const uint32_t i = t * kBlock + threadIdx.x;
      sq[t] = i < me ? await_f64(state.granules + 4ull * i + 2, state.epoch) : 0.0;
      if (i == me) sq[t] = s_own[1];
    }
    double a[1] = {0.0}, b[1] = {0.0};
#pragma unroll
    for (uint32_t t = 0; t < kScanFusedMaxChunks / kBlock; ++t) {
      if (t * kBlock + threadIdx.x < chunk_count) {
        a[0] += held[t];
        b[0] += sq[t];
      }
    }
    __syncthreads();
    block_reduce<1>(a, scratch);
    block_reduce<1>(b, scratch);
    if (threadIdx.x == 0) {
      fin.d_sums[0] = a[0];
      fin.d_sums[1] = b[0];
      if (fin.sums_mirror) {
        fin.sums_mirror[0] = a[0];
        fin.sums_mirror[1] = b[0];
      }
      if (fin.policy) recovery_policy_step(a[0], fin);
    }
  }
}

// ---- K6 resample draw -----------------------------------------------------------------------------------
// spatial_hash.hpp:45-75,87-94,190-193
__device__ __forceinline__ unsigned long long floor_and_fibo_hash(double value, unsigned shift) {
  const long long sv = static_cast<long long>(floor(value));
  const unsigned long long h = 11400714819323198485ull * static_cast<unsigned long long>(sv);
  return shift ? ((h << shift) | (h >> (64 - shift))) : h;
}
__device__ __forceinline__ unsigned long long spatial_hash(const Pose2& s, const HashParams& hp) {
  return floor_and_fibo_hash(s.x / hp.res_x, 0) ^ floor_and_fibo_hash(s.y / hp.res_y, 21) ^
         floor_and_fibo_hash(rot_log(s.r) / hp.res_theta, 42);
}

// std::lower_bound over the cdf: first index with cdf[i] >= target, clamped to n-1 (discrete_distribution forces the last
// cumulative probability to one), through the 16-ary tree (CdfTree): one group of <= 16 entries (one cache line) per level.
__device__ __forceinline__ uint64_t group_lower_bound(const double* __restrict__ a, uint64_t begin, uint64_t end, double target) {
  uint64_t lo = begin, len = end - begin;
  while (len > 0) {
    const uint64_t half = len >> 1;
    if (a[lo + half] < target) {
      lo += half + 1;
      len -= half + 1;
    } else {
      len = half;
    }
  }
  return lo < end ? lo : end - 1;
}
__device__ __forceinline__ uint64_t cdf_tree_lower_bound(const CdfTree& t, double target) {
  uint64_t pos = 0;
  for (int l = t.depth - 1; l >= 0; --l) {
    const uint64_t begin = pos * 16, size = t.size[l];
    pos = group_lower_bound(t.levels + t.offset[l], begin, begin + 16 < size ? begin + 16 : size, target);
  }
  const uint64_t begin = pos * 16;
  return group_lower_bound(t.cdf, begin, begin + 16 < t.n ? begin + 16 : t.n, target);
}
// The same search with the sampled levels >= first_staged read from a workgroup copy in LDS (they are contiguous in
// t.levels from t.offset[first_staged] on): a look-up of 64 unrelated lines costs the vector-memory pipe a cycle per lane,
// an LDS read two per wave.  Same comparisons, same result.
__device__ __forceinline__ uint64_t lds_group_lower_bound(const double* a, uint32_t begin, uint32_t end, double target) {
  uint32_t lo = begin, len = end - begin;
  while (len > 0) {
    const uint32_t half = len >> 1;
    if (a[lo + half] < target) {
      lo += half + 1;
      len -= half + 1;
    } else {
      len = half;
    }
  }
  return lo < end ? lo : end - 1;
}
// One group of the search, for all 64 lanes of a wave at once (every lane has to be here).  A lane's group is one 128-byte
// line somewhere in `a`; four dependent 8-byte loads per lane - the binary search - make four instructions of 64 unrelated
// lines each.  Instead the 8 lanes of an octet read ONE lane's line together, 16 bytes each (a fully coalesced access), eight
// times; every lane compares its two entries with the served lane's target, and that lane counts the entries below the
// target among the ballots' bits of its octet: in a sorted group that count is std::lower_bound's offset.
__device__ __forceinline__ uint64_t wave_group_lower_bound(const double* __restrict__ a, uint32_t pos, uint64_t size, double target) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u, octet = lane & ~7u;
  uint32_t below = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; ++j) {
    const uint32_t served = octet + j;
    const uint64_t first = static_cast<uint64_t>(__shfl(pos, served)) * 16u + 2u * sub;  // this lane's two entries of that group
    const double their_target = __shfl(target, served);
    double2 v;
    if (first + 1 < size) {
      v = *reinterpret_cast<const double2*>(a + first);
    } else {  // the array's last, partial group
      v.x = first < size ? a[first] : INFINITY;
      v.y = INFINITY;
    }
    const uint64_t low = __builtin_amdgcn_ballot_w64(v.x < their_target), high = __builtin_amdgcn_ballot_w64(v.y < their_target);
    if (sub == j)
      below = static_cast<uint32_t>(__builtin_popcountll((low >> octet) & 0xFFull) + __builtin_popcountll((high >> octet) & 0xFFull));
  }
  const uint64_t begin = static_cast<uint64_t>(pos) * 16u, end = begin + 16 < size ? begin + 16 : size;
  const uint64_t at = begin + below;
  return at < end ? at : end - 1;
}
// The search with the sampled levels >= first_staged read from a workgroup copy in LDS (they are contiguous in t.levels
// from t.offset[first_staged] on) and the groups in global memory read a wave at a time.  Same comparisons as
// cdf_tree_lower_bound, same result.  Every lane of the wave has to call it (lanes with nothing to draw: any target).
__device__ __forceinline__ uint64_t cdf_tree_lower_bound_staged(const CdfTree& t, const double* staged, int first_staged, double target) {
  uint64_t pos = 0;
  for (int l = t.depth - 1; l >= first_staged; --l) {
    const uint32_t begin = static_cast<uint32_t>(pos) * 16u, size = t.size[l];
    pos = lds_group_lower_bound(staged + (t.offset[l] - t.offset[first_staged]), begin, begin + 16 < size ? begin + 16 : size, target);
  }
  for (int l = (first_staged < t.depth ? first_staged : t.depth) - 1; l >= 0; --l)
    pos = wave_group_lower_bound(t.levels + t.offset[l], static_cast<uint32_t>(pos), t.size[l], target);
  return wave_group_lower_bound(t.cdf, static_cast<uint32_t>(pos), t.n, target);
}
// multivariate_uniform_distribution.hpp:145-147 over occupancy_grid.hpp:140-146,164-171: uniform heading,
// centre of a uniformly chosen free cell in the world frame.  Addressed by the candidate's global index.
__device__ __forceinline__ Pose2 random_free_state(uint64_t seed, uint32_t step, uint64_t j, const GridView& g, const FreeCells& fc) {
  const RngWords q = rng_draw(seed, step, kRngRandomState, j);
  uint64_t cell = static_cast<uint64_t>(rng_uniform53(q.w[0], q.w[1]) * static_cast<double>(fc.count));
  if (cell >= fc.count) cell = fc.count - 1;
  const double theta = -kPi + 2.0 * kPi * rng_uniform53(q.w[2], q.w[3]);
  const uint32_t idx = fc.index[cell];
  const double lx = (static_cast<double>(static_cast<int>(idx % g.W)) + 0.5) * g.resolution;
  const double ly = (static_cast<double>(static_cast<int>(idx / g.W)) + 0.5) * g.resolution;
  double gx, gy;
  rot_apply(g.origin.r, lx, ly, gx, gy);
  Pose2 s;
  s.r = rot_exp(theta);
  s.x = gx + g.origin.x;
  s.y = gy + g.origin.y;
  return s;
}

// random_intersperse.hpp:90-100: never before the first element; Bernoulli(p) afterwards.
__device__ __forceinline__ bool intersperse_here(const RngWords& r, uint64_t j, double p, uint64_t n_free) {
  return j > 0 && p > 0.0 && rng_uniform32(r.w[2]) < p && n_free > 0;
}

// kEstimate: the nine sums of beluga::estimate (estimation.hpp:436-475) over the new set (weights 1) are accumulated on the
// way out — est_partials[k][workgroup] — so that the estimate needs no pass of its own over the particles it just wrote.
// Workgroups of 1024 outputs share an LDS copy of the upper levels of the search tree (staged doubles from level
// first_staged on; dynamic shared memory).
constexpr int kDrawBlock = 1024;
constexpr uint32_t kDrawStageMax = 4608;  // doubles (36 KB): two workgroups per CU (which also takes 8 waves per SIMD: 64 registers)
// fold (kEstimate, optional): no k_final_rows launch behind this one - a workgroup stores its nine partial sums WRITTEN THROUGH (one
// relaxed agent-scope store each: sc1), waits for them to land (s_waitcnt vmcnt(0)) and draws a ticket (atomicInc, wrapping to zero
// behind the launch's last workgroup); the workgroup that draws the last ticket reads all partials back (relaxed agent-scope loads) and
// adds every row up as k_final_rows does - 256 threads per row, thread t the workgroups t, t + 256, ... in order, wave sums, the four
// waves in order: the same bits - four rows at a time on its sixteen waves (cdna_hip_programming.md section 6, guideline 16: written-through
// payload + drained counter, no release fence and no L2 write-back).
struct DrawFold {
  unsigned int* ticket{nullptr};  // zero between launches; nullptr = k_final_rows follows
  double* out{nullptr};           // [9] the sums
  double* host_mirror{nullptr};   // optional
  Completion done{};              // optional completion word (cycle_spin)
};
// The NEXT cycle's propagation normals of the particle an output slot becomes (k_noise_ahead's work, inside this kernel): the draw waits for
// the fabric - random 128-byte lines of CDF groups and pose records - with its vector units a quarter busy, and the normals are pure arithmetic on
// (seed, step + 1, global particle index).  normals == nullptr: not drawn here.
struct DrawNormals {
  double* normals{nullptr};  // three arrays of `stride` doubles
  uint64_t stride{0};
  uint64_t index_offset{0};  // global index of output slot 0's particle less a.out_offset (the shard's offset)
  uint32_t step{0};          // the step they are for
  // Option order_ahead: the ordering key of where the slot's particle will be after the NEXT propagation - its normals are at hand, the
  // control action is the predicted one, the frame the host's prediction of that set's - in single precision and to first order (only
  // locality depends on a key).  keys == nullptr: none.
  uint32_t* keys{nullptr};
  DiffDriveSampler predicted{};
  KeyFrame frame{};
};
// Where a pose goes under the motion model with the given normals, to first order and in single precision: (c, s, x, y).
__device__ __forceinline__ double4 predicted_pose_f32(const Pose2& p, const DiffDriveSampler& m, float z0, float z1, float z2) {
  const float c = static_cast<float>(p.r.c), s = static_cast<float>(p.r.s);
  float turn, ahead, left, dir_c, dir_s;  // heading change; translation along / across the direction (dir_c, dir_s) in the robot's frame
  if (m.kind == 1) {  // omnidirectional: rotation z0, translation z1 along `first`, strafe z2
    turn = z0 * static_cast<float>(m.s1) + static_cast<float>(m.m1);
    ahead = z1 * static_cast<float>(m.st) + static_cast<float>(m.mt);
    left = -(z2 * static_cast<float>(m.s2));
    dir_c = static_cast<float>(m.first_c);
    dir_s = static_cast<float>(m.first_s);
  } else if (m.kind == 2) {  // stationary
    turn = z0 * 0.02f;
    ahead = z1 * 0.02f;
    left = z2 * 0.02f;
    dir_c = 1.f;
    dir_s = 0.f;
  } else {  // differential: first rotation, translation along the new heading, second rotation
    const float r1 = z0 * static_cast<float>(m.s1) + static_cast<float>(m.m1);
    turn = r1 + z2 * static_cast<float>(m.s2) + static_cast<float>(m.m2);
    ahead = z1 * static_cast<float>(m.st) + static_cast<float>(m.mt);
    left = 0.f;
    dir_c = __cosf(r1);
    dir_s = __sinf(r1);
  }
  const float hc = c * dir_c - s * dir_s, hs = s * dir_c + c * dir_s;  // the direction of the translation in the world
  const float tc = __cosf(turn), ts = __sinf(turn);
  return double4{static_cast<double>(c * tc - s * ts), static_cast<double>(s * tc + c * ts),
                 p.x + static_cast<double>(ahead * hc - left * hs), p.y + static_cast<double>(ahead * hs + left * hc)};
}
template <bool kEstimate>
__global__ __launch_bounds__(kDrawBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_resample_draw(Particles src, CdfTree cdf, const double* __restrict__ d_total,
                                                              Particles dst, ResampleArgs a, GridView g, FreeCells fc, HashParams hp,
                                                              unsigned long long* __restrict__ hashes, double pivot_x, double pivot_y,
                                                              double* __restrict__ est_partials, uint32_t est_stride, int first_staged,
                                                              uint32_t staged_doubles, DrawFold fold, DrawNormals ahead) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* staged = reinterpret_cast<double*>(smem);
  __shared__ double scratch[kEstimate ? (kDrawBlock / 64) * 9 : 1];
  if (first_staged < cdf.depth) {
    const double* from = cdf.levels + cdf.offset[first_staged];
    for (uint32_t k = threadIdx.x; k < staged_doubles; k += kDrawBlock) staged[k] = from[k];
  }
  __syncthreads();
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * kDrawBlock + threadIdx.x;
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // The search reads global memory a wave at a time: every lane goes through it, the ones with nothing to draw (beyond
  // the count, or taking an injected random state) with a target of zero.
  const uint64_t j = a.first_candidate + t;
  const RngWords r = rng_draw(a.seed, a.step, kRngResample, j);
  const double p_random = a.d_random_state_probability ? *a.d_random_state_probability : a.random_state_probability;
  const bool intersperse = t < a.count && intersperse_here(r, j, p_random, fc.count);
  uint64_t idx = 0;
  if (a.n_in >= 2) {
    const double target = (t < a.count && !intersperse) ? rng_uniform53(r.w[0], r.w[1]) * (*d_total) : 0.0;
    idx = cdf_tree_lower_bound_staged(cdf, staged, first_staged, target);
  }
  if (t < a.count) {
    Pose2 s;
    if (intersperse) {
      s = random_free_state(a.seed, a.step, j, g, fc);
    } else {
      s = load_pose(src, idx);
    }
    const uint64_t o = a.out_offset + t;
    store_pose(dst, o, s);
    dst.w[o] = 1.0;  // particle_traits.hpp:105
    if (hashes) hashes[o] = spatial_hash(s, hp);
    if (ahead.normals) {  // (uniform) behind the slot's own stores: nothing of the search is alive any more
      const double4 z = propagation_normals(a.seed, ahead.step, ahead.index_offset + o);
      ahead.normals[o] = z.x;
      ahead.normals[ahead.stride + o] = z.y;
      ahead.normals[2 * ahead.stride + o] = z.z;
      if (ahead.keys)  // (uniform)
        ahead.keys[o] = order_key(predicted_pose_f32(s, ahead.predicted, static_cast<float>(z.x), static_cast<float>(z.y), static_cast<float>(z.z)),
                                  ahead.frame);
    }
    if (kEstimate) {
      const double dx = s.x - pivot_x, dy = s.y - pivot_y;
      v[0] = 1.0;
      v[1] = 1.0;
      v[2] = s.r.c;
      v[3] = s.r.s;
      v[4] = dx;
      v[5] = dy;
      v[6] = dx * dx;
      v[7] = dx * dy;
      v[8] = dy * dy;
    }
  }
  if (kEstimate) {
    block_reduce<9, kDrawBlock>(v, scratch);
    if (fold.ticket == nullptr) {
      if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) est_partials[static_cast<size_t>(k) * est_stride + blockIdx.x] = v[k];
      }
      return;
    }
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k)
        __hip_atomic_store(est_partials + static_cast<size_t>(k) * est_stride + blockIdx.x, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the written-through partials have landed before the ticket counts
      s_last = atomicInc(fold.ticket, gridDim.x - 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;  // (uniform)
    // the launch's last workgroup: rows k = group, group + 4, group + 8 on the 256 threads of wave group `group`
    const uint32_t group = threadIdx.x >> 8, t = threadIdx.x & 255u, wave_in_group = t >> 6;
    double* group_scratch = scratch + group * 4;  // (9 * 16 doubles: room for 4 x 4)
#pragma unroll 1
    for (uint32_t round = 0; round < 3; ++round) {
      const uint32_t k = round * 4 + group;
      double acc = 0.0;
      if (k < 9)
        for (uint32_t b = t; b < gridDim.x; b += 256u)
          acc += __hip_atomic_load(est_partials + static_cast<size_t>(k) * est_stride + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc = wave_sum_f64(acc);
      __syncthreads();
      if ((t & 63u) == 0) group_scratch[wave_in_group] = acc;
      __syncthreads();
      if (t == 0 && k < 9) {
        double total = group_scratch[0];
        for (int w = 1; w < 4; ++w) total += group_scratch[w];
        fold.out[k] = total;
        if (fold.host_mirror) fold.host_mirror[k] = total;
      }
    }
    if (fold.done.host_flag) {
      __threadfence_system();  // every mirrored value is visible to the host before the completion word
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(fold.done.host_flag, fold.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// -- sharded resampling (one context per GPU; the exchange between them is done by the caller) -----------
// targets[t] = u_j * total for output slot j = first_slot + t, NaN where the slot takes an injected random state.
// d_plan (optional): {total of the global CDF, random state probability} in device memory (k_shard_plan) instead of p / total
__global__ __launch_bounds__(kBlock) void k_resample_targets(uint64_t seed, uint32_t step, double p, double total, uint64_t first_slot,
                                                             uint64_t count, uint64_t n_free, double* __restrict__ targets,
                                                             const double* __restrict__ d_plan) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= count) return;
  if (d_plan) {
    total = d_plan[0];
    p = d_plan[1];
  }
  const uint64_t j = first_slot + t;
  const RngWords r = rng_draw(seed, step, kRngResample, j);
  targets[t] = intersperse_here(r, j, p, n_free) ? __builtin_nan("") : rng_uniform53(r.w[0], r.w[1]) * total;
}

// Routing of the resample targets to the shards that own them (counting sort by destination rank, <= 64 ranks).
constexpr uint32_t kMaxRanks = 64;
// skip_injected (the fixed-capacity exchange): an injected slot (NaN) gets no destination at all (dest 0xFF) - it would take an entry of
// the self segment, whose capacity budgets the requests alone (with a recovery probability p the self segment would need m / world + p m
// entries and overflow in every cycle of the recovery phase); k_commit_injected writes those slots straight from the targets.
__global__ __launch_bounds__(kBlock) void k_route_hist(const double* __restrict__ targets, uint64_t count, const double* __restrict__ ends,
                                                       uint32_t world, uint32_t self_rank, uint8_t* __restrict__ dest,
                                                       uint32_t* __restrict__ block_hist, uint32_t nblocks, int skip_injected) {
  __shared__ uint32_t hist[kMaxRanks];
  __shared__ double s_ends[kMaxRanks];
  if (threadIdx.x < kMaxRanks) {
    hist[threadIdx.x] = 0;
    s_ends[threadIdx.x] = threadIdx.x < world ? ends[threadIdx.x] : INFINITY;
  }
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) {
    const uint64_t i = base + k * kBlock + threadIdx.x;
    if (i < count) {
      const double t = targets[i];
      uint32_t d = self_rank;  // injected slots (NaN) are served locally and ignored at commit time
      if (t == t) {
        d = 0;
        while (d + 1 < world && s_ends[d] < t) ++d;  // std::lower_bound over the shard interval ends
      } else if (skip_injected) {
        dest[i] = 0xFFu;
        continue;
      }
      dest[i] = static_cast<uint8_t>(d);
      atomicAdd(&hist[d], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < world) block_hist[static_cast<size_t>(threadIdx.x) * nblocks + blockIdx.x] = hist[threadIdx.x];
}

// pad_capacity == 0: the requests of destination d follow those of d - 1 (compact; the counts go to the host, which sizes the exchange).
// pad_capacity > 0 (the fixed-capacity exchange): destination d owns the entries [d * pad_capacity, (d + 1) * pad_capacity) of lists the
// caller has filled with 0xFF bytes - a NaN target, slot 0xFFFFFFFF: "no request" -; a destination with more requests than that sets
// *overflow and loses the ones beyond (the cycle's resampling is then run again with exact counts: sharded_update).
__global__ __launch_bounds__(kBlock) void k_route_scatter(const double* __restrict__ targets, uint64_t count,
                                                          const double* __restrict__ shard_offsets, uint32_t world,
                                                          const uint8_t* __restrict__ dest, const uint32_t* __restrict__ block_offsets,
                                                          uint32_t nblocks, double* __restrict__ send_targets,
                                                          uint32_t* __restrict__ order, uint32_t pad_capacity, double* __restrict__ overflow) {
  __shared__ uint32_t cursor[kMaxRanks];
  __shared__ double s_off[kMaxRanks];
  if (threadIdx.x < world) {
    const uint32_t compact = block_offsets[static_cast<size_t>(threadIdx.x) * nblocks + blockIdx.x];
    cursor[threadIdx.x] = pad_capacity ? threadIdx.x * pad_capacity + (compact - block_offsets[static_cast<size_t>(threadIdx.x) * nblocks]) : compact;
    s_off[threadIdx.x] = shard_offsets[threadIdx.x];
  }
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) {
    const uint64_t i = base + k * kBlock + threadIdx.x;
    if (i < count) {
      const uint32_t d = dest[i];
      if (d == 0xFFu) continue;  // an injected slot of the fixed-capacity exchange: no request (k_route_hist)
      const double t = targets[i];
      const uint32_t slot = atomicAdd(&cursor[d], 1u);
      if (pad_capacity && slot >= (d + 1u) * pad_capacity) {
        *overflow = 1.0;  // (every writer stores the same value)
        continue;
      }
      send_targets[slot] = t == t ? t - s_off[d] : 0.0;
      order[slot] = static_cast<uint32_t>(i);
    }
  }
}

__global__ void k_route_counts(const uint32_t* __restrict__ block_offsets, uint32_t nblocks, uint32_t world, uint64_t count,
                               long long* __restrict__ counts) {
  const uint32_t d = threadIdx.x;
  if (d >= world) return;
  const uint64_t begin = block_offsets[static_cast<size_t>(d) * nblocks];
  const uint64_t end = d + 1 < world ? block_offsets[static_cast<size_t>(d + 1) * nblocks] : count;
  counts[d] = static_cast<long long>(end - begin);
}

// AoS variants for the exchange buffers: reply[t] = (x, y, c, s) of the served ancestor.
__global__ __launch_bounds__(kBlock) void k_gather_by_cdf_aos(Particles src, CdfTree cdf,
                                                              const double* __restrict__ targets, uint64_t m, double4* __restrict__ out) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= m) return;
  const double target = targets[t];
  if (!(target == target)) {  // "no request" (an entry of the fixed-capacity exchange that nobody filled)
    out[t] = double4{0.0, 0.0, 0.0, 0.0};
    return;
  }
  const uint64_t idx = cdf_tree_lower_bound(cdf, target);
  const double4 v = src.pose[idx];
  out[t] = double4{v.z, v.w, v.x, v.y};
}

__global__ __launch_bounds__(kBlock) void k_commit_routed(Particles dst, uint64_t seed, uint32_t step, uint64_t first_slot,
                                                          uint64_t count, const double4* __restrict__ replies,
                                                          const uint32_t* __restrict__ order, const double* __restrict__ targets,
                                                          GridView g, FreeCells fc) {
  const uint64_t k = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= count) return;
  const uint32_t t = order[k];  // reply k answers output slot t
  if (t == 0xFFFFFFFFu) return;  // (an unused entry of the fixed-capacity exchange)
  Pose2 v;
  if (targets[t] != targets[t]) {
    v = random_free_state(seed, step, first_slot + t, g, fc);
  } else {
    const double4 r = replies[k];
    v = Pose2{Rot2{r.z, r.w}, r.x, r.y};
  }
  store_pose(dst, t, v);
  dst.w[t] = 1.0;
}

// The injected slots of the fixed-capacity exchange (their targets are NaN and they are in no request list: k_route_hist): the random
// state of output slot first_slot + t, exactly as k_commit_routed writes it in the exact exchange.
__global__ __launch_bounds__(kBlock) void k_commit_injected(Particles dst, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count,
                                                            const double* __restrict__ targets, GridView g, FreeCells fc) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= count || targets[t] == targets[t]) return;
  store_pose(dst, t, random_free_state(seed, step, first_slot + t, g, fc));
  dst.w[t] = 1.0;
}

// KLD form of the routed exchange: the candidates of slots [first_slot, first_slot + count) are only *staged* (the cut
// of take_while_kld is not known yet): states go to a caller buffer in API order (cos, sin, x, y) with their spatial hashes.
__global__ __launch_bounds__(kBlock) void k_finish_candidates(uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count,
                                                              const double4* __restrict__ replies, const uint32_t* __restrict__ order,
                                                              const double* __restrict__ targets, GridView g, FreeCells fc,
                                                              HashParams hp, double4* __restrict__ states,
                                                              unsigned long long* __restrict__ hashes) {
  const uint64_t k = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= count) return;
  const uint32_t t = order[k];
  Pose2 v;
  if (targets[t] != targets[t]) {
    v = random_free_state(seed, step, first_slot + t, g, fc);
  } else {
    const double4 r = replies[k];
    v = Pose2{Rot2{r.z, r.w}, r.x, r.y};
  }
  states[t] = double4{v.r.c, v.r.s, v.x, v.y};
  hashes[t] = spatial_hash(v, hp);
}

// ---- K7 KLD ---------------------------------------------------------------------------------------------
constexpr unsigned long long kEmptyKey = ~0ull;
__device__ __forceinline__ unsigned long long kld_key(unsigned long long h) { return h == kEmptyKey ? h - 1 : h; }
__device__ __forceinline__ uint64_t kld_slot(unsigned long long key, uint64_t mask) { return (key ^ (key >> 29) ^ (key >> 47)) & mask; }

// Insert (hash -> smallest candidate index).  A tight particle cloud has a few hundred distinct bins, so a naive
// per-candidate global atomic would serialise tens of thousands of updates on the same few addresses; each workgroup
// first de-duplicates its 2048 candidates in an LDS table and only the per-workgroup winners touch global memory.
constexpr uint32_t kLocalSlots = 4096;
__global__ __launch_bounds__(kBlock) void k_kld_insert(const unsigned long long* __restrict__ hashes, uint64_t first, uint64_t count,
                                                       KldTable t) {
  __shared__ unsigned long long lkeys[kLocalSlots];
  __shared__ unsigned int lfirst[kLocalSlots];
  for (uint32_t s = threadIdx.x; s < kLocalSlots; s += kBlock) {
    lkeys[s] = kEmptyKey;
    lfirst[s] = 0xFFFFFFFFu;
  }
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) {
    const uint64_t q = base + k * kBlock + threadIdx.x;
    if (q < count) {
      const uint64_t j = first + q;
      const unsigned long long key = kld_key(hashes[j]);
      uint32_t slot = static_cast<uint32_t>(kld_slot(key, kLocalSlots - 1));
      while (true) {
        const unsigned long long prev = atomicCAS(&lkeys[slot], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) {
          atomicMin(&lfirst[slot], static_cast<unsigned int>(j));
          break;
        }
        slot = (slot + 1) & (kLocalSlots - 1);
      }
    }
  }
  __syncthreads();
  const uint64_t mask = t.capacity - 1;
  for (uint32_t s = threadIdx.x; s < kLocalSlots; s += kBlock) {
    const unsigned long long key = lkeys[s];
    if (key == kEmptyKey) continue;
    uint64_t slot = kld_slot(key, mask);
    while (true) {
      const unsigned long long prev = atomicCAS(&t.keys[slot], kEmptyKey, key);
      if (prev == kEmptyKey || prev == key) {
        atomicMin(&t.first[slot], lfirst[s]);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

// take_while_kld.hpp:73-81
__device__ __forceinline__ unsigned long long kld_target_size(unsigned long long k, double two_epsilon, double z) {
  if (k <= 2ull) return ULLONG_MAX;
  const double common = 2. / static_cast<double>(9 * (k - 1));
  const double base = 1. - common + sqrt(common) * z;
  const double result = (static_cast<double>(k - 1) / two_epsilon) * base * base * base;
  return static_cast<unsigned long long>(ceil(result));
}

__global__ __launch_bounds__(kBlock) void k_kld_flags(const unsigned long long* __restrict__ hashes, uint64_t first, uint64_t count,
                                                      KldTable t, uint32_t* __restrict__ flags, uint32_t* __restrict__ chunk_sum) {
  __shared__ uint32_t s_wave[kBlock / 64];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + threadIdx.x * kItems;
  const uint64_t mask = t.capacity - 1;
  uint32_t local = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t q = base + k;
    if (q < count) {
      const uint64_t j = first + q;
      const unsigned long long key = kld_key(hashes[j]);
      uint64_t slot = kld_slot(key, mask);
      while (t.keys[slot] != key) slot = (slot + 1) & mask;
      const uint32_t f = t.first[slot] == static_cast<unsigned int>(j) ? 1u : 0u;
      flags[q] = f;
      local += f;
    }
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

__global__ __launch_bounds__(kBlock) void k_kld_check(uint64_t first, uint64_t count, const uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ chunk_offset, uint64_t min_particles,
                                                      double two_epsilon, double z, unsigned long long* __restrict__ first_fail) {
  __shared__ uint32_t s_wave[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + threadIdx.x * kItems;
  uint32_t loc[kItems];
  uint32_t run = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t q = base + k;
    run += q < count ? flags[q] : 0u;
    loc[k] = run;
  }
  uint32_t incl = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  uint32_t prefix = chunk_offset[blockIdx.x];  // includes k_base
  for (int q = 0; q < wave; ++q) prefix += s_wave[q];
  prefix += incl - run;
  unsigned long long fail = ULLONG_MAX;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t q = base + k;
    if (q < count) {
      const unsigned long long cnt = first + q + 1;  // kld_condition's count after this element
      const unsigned long long buckets = prefix + loc[k];
      const bool keep = cnt <= min_particles || cnt <= kld_target_size(buckets, two_epsilon, z);
      if (!keep && fail == ULLONG_MAX) fail = first + q;
    }
  }
  if (fail != ULLONG_MAX) atomicMin(first_fail, fail);
}

// ---- K8 estimate -------------------------------------------------------------------------------------------
constexpr int kEstK = 9;
__global__ __launch_bounds__(kBlock) void k_estimate_partials(Particles p, uint64_t n, double pivot_x, double pivot_y,
                                                              double* __restrict__ partials, uint32_t stride) {
  __shared__ double scratch[(kBlock / 64) * kEstK];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + threadIdx.x * kItems;
  double v[kEstK];
#pragma unroll
  for (int k = 0; k < kEstK; ++k) v[k] = 0.0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t i = base + k;
    if (i < n) {
      const double w = p.w[i];
      const double4 q = p.pose[i];
      const double dx = q.z - pivot_x, dy = q.w - pivot_y;
      v[0] += w;
      v[1] += w * w;
      v[2] += w * q.x;
      v[3] += w * q.y;
      v[4] += w * dx;
      v[5] += w * dy;
      v[6] += w * dx * dx;
      v[7] += w * dx * dy;
      v[8] += w * dy * dy;
    }
  }
  block_reduce<kEstK>(v, scratch);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kEstK; ++k) partials[static_cast<size_t>(k) * stride + blockIdx.x] = v[k];
  }
}

// ---- the whole tail of the cycle in ONE workgroup (small sets: the reference's own sizes, amcl_core.hpp:44-46: 500 .. 2000) --------------
// Behind the reweight, a cycle of a few thousand particles was seven launches when its size is fixed and a dozen with three host
// synchronisations when it is KLD-adaptive - the reference's default configuration - every one of them a few microseconds of launch and
// dependency latency around nanoseconds of work (profiles/r04_small_filters.txt: 0.05 / 0.11 ms per update).  Here one workgroup of 1024
// threads does, for a set and a candidate stream of up to kSmallMax particles:
//   actions::normalize (normalize.hpp:54-85), the totals of the normalised weights, ThrunRecoveryProbabilityEstimator on them
//   (thrun_recovery_probability_estimator.hpp:69-89, exponential_filter.hpp:32-44; reset rule amcl_core.hpp:184-186), the resampling policy
//   every_n [&& on_effective_size_drop] (every_n.hpp:47-50, on_effective_size_drop.hpp:45-49, effective_sample_size.hpp:46-59), and, where it
//   fires, views::sample | random_intersperse | take_while_kld | take(max) | assign (sample.hpp:128-159, random_intersperse.hpp:90-115,
//   take_while_kld.hpp:72-88 over spatial_hash.hpp:45-94,190-193) - the CDF in workgroup memory, std::lower_bound per candidate, the KLD cut
//   order-exact: k(j) = distinct hashes among candidates 0 .. j through a table that keeps the smallest candidate index per hash - and the sums
//   of beluga::estimate (estimation.hpp:436-475) over the set it leaves.
// Everything the host needs comes back through the mirror in mapped host memory behind ONE synchronisation.  Same Philox stream, same
// expressions as the kernels of the large path; its sums are added in this kernel's own fixed order (results within the rounding of a sum:
// the parity tests' tolerances, the KLD cut and the particle counts exact).
constexpr uint32_t kSmallMax = 4096, kSmallBlock = 1024, kSmallItems = kSmallMax / kSmallBlock, kSmallSlots = 2 * kSmallMax;
constexpr size_t kSmallLdsBytes = kSmallMax * 8 /* cdf */ + kSmallMax * 8 /* hashes */ + kSmallSlots * 4 /* table */ + 16 * 9 * 8 + 128;
struct SmallTailArgs {
  Particles src, dst;
  uint32_t n;               // live particles
  uint32_t min_particles, max_particles;
  uint64_t seed;
  uint32_t step;
  int fires;                // every_n says so
  int selective;            // && on_effective_size_drop
  int adaptive;             // min < max: take_while_kld
  double alpha_slow, alpha_fast, slow, fast;  // the recovery estimator's filters (the host keeps their state)
  double two_epsilon, z;
  HashParams hp;
  GridView g;
  FreeCells fc;
  double pivot_x, pivot_y;
  double* out;              // [32] mirror in mapped host memory (the context's h_scalars): see the stores below
  double* d_out;            // the same values in device memory (d_scalars)
  unsigned long long* done_flag;  // optional: a word of mapped host memory that takes done_seq behind everything mirrored (cycle_spin)
  unsigned long long done_seq;
};
__device__ __forceinline__ double small_block_sum(double v, double* s_wave /* [16] */) {  // every thread gets the total; fixed order
  v = wave_sum_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
  __syncthreads();
  double total = s_wave[0];
  for (uint32_t q = 1; q < kSmallBlock / 64; ++q) total += s_wave[q];
  return total;
}
__global__ __launch_bounds__(kSmallBlock) void k_small_tail(SmallTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* s_cdf = reinterpret_cast<double*>(smem);
  unsigned long long* s_hash = reinterpret_cast<unsigned long long*>(smem + kSmallMax * 8);
  uint32_t* s_table = reinterpret_cast<uint32_t*>(smem + 2 * kSmallMax * 8);
  double* s_wave = reinterpret_cast<double*>(smem + 2 * kSmallMax * 8 + kSmallSlots * 4);  // [16][9]
  uint32_t* s_word = reinterpret_cast<uint32_t*>(s_wave + 16 * 9);                          // [16]
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = a.n;
  // ---- normalise (thread t holds the elements 4 t .. 4 t + 3)
  double x[kSmallItems];
  double local = 0.0;
#pragma unroll
  for (uint32_t k = 0; k < kSmallItems; ++k) {
    const uint32_t i = tid * kSmallItems + k;
    x[k] = i < n ? a.src.w[i] : 0.0;
    local += x[k];
  }
  const double total = small_block_sum(local, s_wave);
  const bool skip = fabs(total - 1.0) < DBL_EPSILON;  // normalize.hpp:73
  double l1 = 0.0, l2 = 0.0;
#pragma unroll
  for (uint32_t k = 0; k < kSmallItems; ++k) {
    if (!skip) x[k] = x[k] / total;
    l1 += x[k];
    l2 += x[k] * x[k];
  }
  const double norm_sum = small_block_sum(l1, s_wave), norm_sumsq = small_block_sum(l2, s_wave);
  // ---- recovery estimator, resampling policy (uniform: every thread computes the same)
  double slow = a.slow, fast = a.fast, p = 0.0;
  {
    const double average = norm_sum / static_cast<double>(n);
    fast += (fast == 0.) ? average : a.alpha_fast * (average - fast);
    slow += (slow == 0.) ? average : a.alpha_slow * (average - slow);
    if (fabs(slow) >= 2.220446049250313e-16) p = fmin(fmax(1.0 - fast / slow, 0.0), 1.0);
  }
  bool resample = a.fires != 0;
  double ess = -1.0;
  if (resample && a.selective) {
    ess = norm_sum == 0.0 ? 0.0 : (norm_sum * norm_sum) / norm_sumsq;
    resample = ess < static_cast<double>(n) * 0.5;
  }
  if (resample && p > 0.0) slow = fast = 0.0;  // amcl_core.hpp:184-186
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t n_out = n;
  if (!resample) {
    // the set stays: its normalised weights, and the estimate over them
#pragma unroll
    for (uint32_t k = 0; k < kSmallItems; ++k) {
      const uint32_t i = tid * kSmallItems + k;
      if (i < n) {
        if (!skip) a.src.w[i] = x[k];
        const double w = x[k];
        const double4 q = a.src.pose[i];
        const double dx = q.z - a.pivot_x, dy = q.w - a.pivot_y;
        v[0] += w;
        v[1] += w * w;
        v[2] += w * q.x;
        v[3] += w * q.y;
        v[4] += w * dx;
        v[5] += w * dy;
        v[6] += w * dx * dx;
        v[7] += w * dx * dy;
        v[8] += w * dy * dy;
      }
    }
  } else {
    // ---- CDF: inclusive scan of the normalised weights, in workgroup memory
    double run = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < kSmallItems; ++k) {
      run += x[k];
      x[k] = run;
    }
    double incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o);
      if (lane >= static_cast<uint32_t>(o)) incl += up;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    double prefix = incl - run;
    for (uint32_t q = 0; q < wave; ++q) prefix += s_wave[q];
#pragma unroll
    for (uint32_t k = 0; k < kSmallItems; ++k) {
      const uint32_t i = tid * kSmallItems + k;
      if (i < n) s_cdf[i] = prefix + x[k];
    }
    for (uint32_t s = tid; s < kSmallSlots; s += kSmallBlock) s_table[s] = 0xFFFFFFFFu;
    if (tid == 0) s_word[0] = 0xFFFFFFFFu;  // first candidate that fails kld_condition
    __syncthreads();
    const double cdf_total = s_cdf[n - 1];
    // ---- the candidates (candidate j = output slot j), thread t takes j = t, t + 1024, ...
    const uint32_t candidates = a.max_particles;
    Pose2 state[kSmallItems];
#pragma unroll
    for (uint32_t k = 0; k < kSmallItems; ++k) {
      const uint32_t j = k * kSmallBlock + tid;
      if (j < candidates) {
        const RngWords r = rng_draw(a.seed, a.step, kRngResample, j);
        if (intersperse_here(r, j, p, a.fc.count)) {
          state[k] = random_free_state(a.seed, a.step, j, a.g, a.fc);
        } else {
          const double target = rng_uniform53(r.w[0], r.w[1]) * cdf_total;
          uint32_t lo = 0, len = n;  // std::lower_bound, clamped to n - 1 (discrete_distribution forces the last cumulative probability to one)
          while (len > 0) {
            const uint32_t half = len >> 1;
            if (s_cdf[lo + half] < target) {
              lo += half + 1;
              len -= half + 1;
            } else {
              len = half;
            }
          }
          state[k] = load_pose(a.src, lo < n ? lo : n - 1);
        }
        store_pose(a.dst, j, state[k]);
        a.dst.w[j] = 1.0;  // particle_traits.hpp:105
        if (a.adaptive) s_hash[j] = kld_key(spatial_hash(state[k], a.hp));
      }
    }
    n_out = candidates;
    if (a.adaptive) {
      __syncthreads();
      // the smallest candidate index per hash
#pragma unroll
      for (uint32_t k = 0; k < kSmallItems; ++k) {
        const uint32_t j = k * kSmallBlock + tid;
        if (j < candidates) {
          const unsigned long long key = s_hash[j];
          uint32_t slot = static_cast<uint32_t>(kld_slot(key, kSmallSlots - 1));
          for (;;) {
            const uint32_t holder = atomicCAS(&s_table[slot], 0xFFFFFFFFu, j);
            if (holder == 0xFFFFFFFFu) break;  // this candidate represents its hash (until a smaller index of the same hash comes)
            if (s_hash[holder] == key) {       // (a slot's holders all carry one hash)
              atomicMin(&s_table[slot], j);
              break;
            }
            slot = (slot + 1) & (kSmallSlots - 1);
          }
        }
      }
      __syncthreads();
      // first occurrences in candidate order: thread t looks at the candidates 4 t .. 4 t + 3 now (a blocked scan), k(j) = their inclusive count
      uint32_t f[kSmallItems], count_run = 0;
#pragma unroll
      for (uint32_t k = 0; k < kSmallItems; ++k) {
        const uint32_t j = tid * kSmallItems + k;
        f[k] = 0;
        if (j < candidates) {
          const unsigned long long key = s_hash[j];
          uint32_t slot = static_cast<uint32_t>(kld_slot(key, kSmallSlots - 1));
          while (s_hash[s_table[slot]] != key) slot = (slot + 1) & (kSmallSlots - 1);
          f[k] = s_table[slot] == j ? 1u : 0u;
        }
        count_run += f[k];
        f[k] = count_run;
      }
      uint32_t incl_u = count_run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl_u, o);
        if (lane >= static_cast<uint32_t>(o)) incl_u += up;
      }
      if (lane == 63) s_word[1 + wave] = incl_u;  // (s_word[0] is the failing candidate's index)
      __syncthreads();
      uint32_t before = incl_u - count_run;
      for (uint32_t q = 0; q < wave; ++q) before += s_word[1 + q];
      uint32_t fail = 0xFFFFFFFFu;
#pragma unroll
      for (uint32_t k = 0; k < kSmallItems; ++k) {
        const uint32_t j = tid * kSmallItems + k;
        if (j < candidates) {
          const unsigned long long cnt = static_cast<unsigned long long>(j) + 1ull;  // kld_condition's count after this element
          const bool keep = cnt <= a.min_particles || cnt <= kld_target_size(before + f[k], a.two_epsilon, a.z);
          if (!keep && fail == 0xFFFFFFFFu) fail = j;
        }
      }
      if (fail != 0xFFFFFFFFu) atomicMin(&s_word[0], fail);
      __syncthreads();
      const uint32_t first_fail = s_word[0];
      n_out = first_fail < candidates ? first_fail : candidates;  // the first element failing the predicate is dropped (take_while) | take(max)
    }
    // ---- the estimate's sums over the new set (weights 1)
#pragma unroll
    for (uint32_t k = 0; k < kSmallItems; ++k) {
      const uint32_t j = k * kSmallBlock + tid;
      if (j < n_out) {
        const double dx = state[k].x - a.pivot_x, dy = state[k].y - a.pivot_y;
        v[0] += 1.0;
        v[1] += 1.0;
        v[2] += state[k].r.c;
        v[3] += state[k].r.s;
        v[4] += dx;
        v[5] += dy;
        v[6] += dx * dx;
        v[7] += dx * dy;
        v[8] += dy * dy;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = wave_sum_f64(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) s_wave[wave * 9 + k] = v[k];
  }
  __syncthreads();
  if (tid < 9) {
    double acc = s_wave[tid];
    for (uint32_t q = 1; q < kSmallBlock / 64; ++q) acc += s_wave[q * 9 + tid];
    a.out[8 + tid] = acc;
    a.d_out[8 + tid] = acc;
  }
  if (tid == 0) {
    const double scalars[10] = {total, norm_sum, norm_sumsq, 0.0, 0.0, resample ? 1.0 : 0.0, static_cast<double>(n_out), ess, slow, fast};
    for (int k = 0; k < 3; ++k) a.out[k] = a.d_out[k] = scalars[k];
    a.out[5] = a.d_out[5] = scalars[5];
    a.out[6] = a.d_out[6] = scalars[6];
    a.out[7] = a.d_out[7] = scalars[7];
    a.out[18] = a.d_out[18] = slow;
    a.out[19] = a.d_out[19] = fast;
    a.out[20] = a.d_out[20] = slow;  // (the device-side policy slot of the large path: {slow, fast, p})
    a.out[21] = a.d_out[21] = fast;
    a.out[22] = a.d_out[22] = p;
  }
  if (a.done_flag) {  // (uniform) the host may be watching this word instead of the stream
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- cluster_based_estimate (algorithm/cluster_based_estimation.hpp) ---------------------------------------------
// Device side: spatial hash of every particle, per-cell aggregation (weight sum, count, first particle), compaction of
// the occupied cells for the host, and the masked estimate of the winning cluster.  The cluster assignment itself is
// a priority-queue flood fill over a few hundred cells and stays on the host (context.hip), with the reference's own
// standard containers so that ties resolve the same way.
__global__ __launch_bounds__(kBlock) void k_cluster_hash(Particles p, uint64_t n, HashParams hp, unsigned long long* __restrict__ hashes) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) hashes[i] = spatial_hash(load_pose(p, i), hp);
}

struct CellTable {
  unsigned long long* keys;
  unsigned int* first;
  double* wsum;
  unsigned int* count;
  unsigned int* cluster;  // written by the host pass
  uint64_t capacity;      // power of two
};

__global__ __launch_bounds__(kBlock) void k_cell_aggregate(const unsigned long long* __restrict__ hashes, const double* __restrict__ w,
                                                           uint64_t n, CellTable t) {
  __shared__ unsigned long long lkeys[kLocalSlots];
  __shared__ unsigned int lfirst[kLocalSlots];
  __shared__ unsigned int lcount[kLocalSlots];
  __shared__ double lsum[kLocalSlots];
  for (uint32_t s = threadIdx.x; s < kLocalSlots; s += kBlock) {
    lkeys[s] = kEmptyKey;
    lfirst[s] = 0xFFFFFFFFu;
    lcount[s] = 0;
    lsum[s] = 0.0;
  }
  __syncthreads();
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kBlock; ++k) {
    const uint64_t i = base + k * kBlock + threadIdx.x;
    if (i < n) {
      const unsigned long long key = kld_key(hashes[i]);
      uint32_t slot = static_cast<uint32_t>(kld_slot(key, kLocalSlots - 1));
      while (true) {
        const unsigned long long prev = atomicCAS(&lkeys[slot], kEmptyKey, key);
        if (prev == kEmptyKey || prev == key) break;
        slot = (slot + 1) & (kLocalSlots - 1);
      }
      atomicMin(&lfirst[slot], static_cast<unsigned int>(i));
      atomicAdd(&lcount[slot], 1u);
      atomicAdd(&lsum[slot], w[i]);
    }
  }
  __syncthreads();
  const uint64_t mask = t.capacity - 1;
  for (uint32_t s = threadIdx.x; s < kLocalSlots; s += kBlock) {
    const unsigned long long key = lkeys[s];
    if (key == kEmptyKey) continue;
    uint64_t slot = kld_slot(key, mask);
    while (true) {
      const unsigned long long prev = atomicCAS(&t.keys[slot], kEmptyKey, key);
      if (prev == kEmptyKey || prev == key) break;
      slot = (slot + 1) & mask;
    }
    atomicMin(&t.first[slot], lfirst[s]);
    atomicAdd(&t.count[slot], lcount[s]);
    atomicAdd(&t.wsum[slot], lsum[s]);
  }
}

struct CellList {  // compacted occupied cells, arbitrary order (the host sorts by `first`)
  unsigned long long* key;
  unsigned int* first;
  unsigned int* count;
  unsigned int* slot;
  double* wsum;
  double4* state;  // representative state = state of particle `first` as (c, s, x, y)
  unsigned int* size;
};

// One pass instead of six fills: the hash table's slots back to empty, the cell counter to zero.
__global__ __launch_bounds__(kBlock) void k_cell_table_clear(CellTable t, unsigned int* __restrict__ list_size) {
  const uint64_t s = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (s == 0) *list_size = 0u;
  if (s >= t.capacity) return;
  t.keys[s] = kEmptyKey;
  t.first[s] = 0xFFFFFFFFu;
  t.wsum[s] = 0.0;
  t.count[s] = 0u;
  t.cluster[s] = 0xFFFFFFFFu;
}
__global__ __launch_bounds__(kBlock) void k_cell_compact(CellTable t, Particles p, CellList out, unsigned int list_capacity) {
  const uint64_t s = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (s >= t.capacity || t.keys[s] == kEmptyKey) return;
  const unsigned int k = atomicAdd(out.size, 1u);
  if (k >= list_capacity) return;  // counted, not stored: the caller sees size > capacity and compacts again into a larger list
  const unsigned int f = t.first[s];
  out.key[k] = t.keys[s];
  out.first[k] = f;
  out.count[k] = t.count[s];
  out.slot[k] = static_cast<unsigned int>(s);
  out.wsum[k] = t.wsum[s];
  out.state[k] = p.pose[f];
}

__global__ __launch_bounds__(kBlock) void k_cell_set_cluster(const unsigned int* __restrict__ slot, const unsigned int* __restrict__ cluster,
                                                             uint32_t m, unsigned int* __restrict__ table_cluster) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k < m) table_cluster[slot[k]] = cluster[k];
}

// estimation.hpp:436-475 restricted to the particles whose cell belongs to cluster `wanted`.
__global__ __launch_bounds__(kBlock) void k_estimate_partials_cluster(Particles p, uint64_t n, const unsigned long long* __restrict__ hashes,
                                                                      CellTable t, unsigned int wanted, double pivot_x, double pivot_y,
                                                                      double* __restrict__ partials, uint32_t stride) {
  __shared__ double scratch[(kBlock / 64) * kEstK];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kChunk + threadIdx.x * kItems;
  const uint64_t mask = t.capacity - 1;
  double v[kEstK];
#pragma unroll
  for (int k = 0; k < kEstK; ++k) v[k] = 0.0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint64_t i = base + k;
    if (i < n) {
      const unsigned long long key = kld_key(hashes[i]);
      uint64_t slot = kld_slot(key, mask);
      while (t.keys[slot] != key) slot = (slot + 1) & mask;
      if (t.cluster[slot] == wanted) {
        const double w = p.w[i];
        const double4 q = p.pose[i];
        const double dx = q.z - pivot_x, dy = q.w - pivot_y;
        v[0] += w;
        v[1] += w * w;
        v[2] += w * q.x;
        v[3] += w * q.y;
        v[4] += w * dx;
        v[5] += w * dy;
        v[6] += w * dx * dx;
        v[7] += w * dx * dy;
        v[8] += w * dy * dy;
      }
    }
  }
  block_reduce<kEstK>(v, scratch);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < kEstK; ++k) partials[static_cast<size_t>(k) * stride + blockIdx.x] = v[k];
  }
}

// ---- cluster_based_estimate of a small set (up to kSmallMax particles): two launches of one workgroup each around the host's pass --------
// The large path is five launches and two synchronisations (clear, hash, aggregate, compact | set clusters, masked sums, final rows): a
// tenth of a millisecond on a set of 2000 particles, twice the rest of its cycle - and cluster_based_estimate is what beluga_ros::Amcl
// returns on every update (beluga_ros/src/amcl.cpp:125).  k_small_cluster_cells: hash of every particle (spatial_hash.hpp:190-193 at the
// clustering's resolutions), a table in workgroup memory that keeps the smallest particle index per hash, counts and weight sums per cell,
// the occupied cells straight into the mapped host list (make_cluster_map, cluster_based_estimation.hpp:137-157: key, weight, count, first
// particle and its state).  k_small_cluster_sums: the host's cluster ids per cell back into a table, the estimate's sums over the particles
// of the winning cluster (estimate_clusters :345-411 over estimation.hpp:436-475).
constexpr size_t kSmallClusterLdsBytes = kSmallMax * 8 /* hashes */ + kSmallSlots * 4 /* table */ + kSmallMax * 4 /* counts */ + kSmallMax * 8 /* sums */ + 64;
__global__ __launch_bounds__(kSmallBlock) void k_small_cluster_cells(Particles p, uint32_t n, HashParams hp, CellList out, unsigned int* size_mirror) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* s_hash = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* s_table = reinterpret_cast<uint32_t*>(smem + kSmallMax * 8);
  uint32_t* s_count = s_table + kSmallSlots;
  double* s_sum = reinterpret_cast<double*>(s_count + kSmallMax);
  uint32_t* s_size = reinterpret_cast<uint32_t*>(s_sum + kSmallMax);
  const uint32_t tid = threadIdx.x;
  for (uint32_t s = tid; s < kSmallSlots; s += kSmallBlock) s_table[s] = 0xFFFFFFFFu;
  for (uint32_t i = tid; i < kSmallMax; i += kSmallBlock) {
    s_count[i] = 0u;
    s_sum[i] = 0.0;
    if (i < n) s_hash[i] = kld_key(spatial_hash(load_pose(p, i), hp));
  }
  if (tid == 0) *s_size = 0u;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kSmallBlock) {  // the smallest particle index per hash
    const unsigned long long key = s_hash[i];
    uint32_t slot = static_cast<uint32_t>(kld_slot(key, kSmallSlots - 1));
    for (;;) {
      const uint32_t holder = atomicCAS(&s_table[slot], 0xFFFFFFFFu, i);
      if (holder == 0xFFFFFFFFu) break;
      if (s_hash[holder] == key) {
        atomicMin(&s_table[slot], i);
        break;
      }
      slot = (slot + 1) & (kSmallSlots - 1);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kSmallBlock) {  // counts and weight sums, kept with the cell's first particle
    const unsigned long long key = s_hash[i];
    uint32_t slot = static_cast<uint32_t>(kld_slot(key, kSmallSlots - 1));
    while (s_hash[s_table[slot]] != key) slot = (slot + 1) & (kSmallSlots - 1);
    const uint32_t first = s_table[slot];
    atomicAdd(&s_count[first], 1u);
    atomicAdd(&s_sum[first], p.w[i]);
  }
  __syncthreads();
  for (uint32_t i = tid; i < n; i += kSmallBlock) {
    if (s_count[i] == 0u) continue;  // (not a cell's first particle)
    const uint32_t k = atomicAdd(s_size, 1u);
    out.key[k] = s_hash[i];
    out.first[k] = i;
    out.count[k] = s_count[i];
    out.slot[k] = 0u;
    out.wsum[k] = s_sum[i];
    out.state[k] = p.pose[i];
  }
  __syncthreads();
  if (tid == 0) {
    *out.size = *s_size;
    *size_mirror = *s_size;
  }
}
__global__ __launch_bounds__(kSmallBlock) void k_small_cluster_sums(Particles p, uint32_t n, HashParams hp, const unsigned long long* __restrict__ keys,
                                                                    const unsigned int* __restrict__ cluster, uint32_t cells, unsigned int wanted,
                                                                    double pivot_x, double pivot_y, double* __restrict__ d_out,
                                                                    double* __restrict__ mirror) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(smem);  // [kSmallMax] the cells' keys
  uint32_t* s_table = reinterpret_cast<uint32_t*>(smem + kSmallMax * 8);   // [kSmallSlots] -> cell
  uint32_t* s_cluster = s_table + kSmallSlots;                             // [kSmallMax]
  double* s_wave = reinterpret_cast<double*>(s_cluster + kSmallMax);       // [16][9] (behind: kSmallMax * 8 of sums' room)
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t s = tid; s < kSmallSlots; s += kSmallBlock) s_table[s] = 0xFFFFFFFFu;
  for (uint32_t j = tid; j < cells; j += kSmallBlock) {
    s_key[j] = keys[j];
    s_cluster[j] = cluster[j];
  }
  __syncthreads();
  for (uint32_t j = tid; j < cells; j += kSmallBlock) {  // (the keys are distinct)
    uint32_t slot = static_cast<uint32_t>(kld_slot(s_key[j], kSmallSlots - 1));
    while (atomicCAS(&s_table[slot], 0xFFFFFFFFu, j) != 0xFFFFFFFFu) slot = (slot + 1) & (kSmallSlots - 1);
  }
  __syncthreads();
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = tid; i < n; i += kSmallBlock) {
    const double4 q = p.pose[i];
    const unsigned long long key = kld_key(spatial_hash(Pose2{Rot2{q.x, q.y}, q.z, q.w}, hp));
    uint32_t slot = static_cast<uint32_t>(kld_slot(key, kSmallSlots - 1));
    uint32_t cell = s_table[slot];
    while (cell != 0xFFFFFFFFu && s_key[cell] != key) {
      slot = (slot + 1) & (kSmallSlots - 1);
      cell = s_table[slot];
    }
    if (cell == 0xFFFFFFFFu || s_cluster[cell] != wanted) continue;
    const double w = p.w[i];
    const double dx = q.z - pivot_x, dy = q.w - pivot_y;
    v[0] += w;
    v[1] += w * w;
    v[2] += w * q.x;
    v[3] += w * q.y;
    v[4] += w * dx;
    v[5] += w * dy;
    v[6] += w * dx * dx;
    v[7] += w * dx * dy;
    v[8] += w * dy * dy;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) v[k] = wave_sum_f64(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) s_wave[wave * 9 + k] = v[k];
  }
  __syncthreads();
  if (tid < 9) {
    double acc = s_wave[tid];
    for (uint32_t q = 1; q < kSmallBlock / 64; ++q) acc += s_wave[q * 9 + tid];
    d_out[tid] = acc;
    mirror[tid] = acc;
  }
}


// ---- misc -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_init_normal(Particles p, uint64_t n, double m0, double m1, double m2, double t00,
                                                        double t01, double t02, double t10, double t11, double t12, double t20,
                                                        double t21, double t22, uint64_t seed, uint64_t index_offset) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const RngWords a = rng_draw(seed, 0, kRngInitA, index_offset + i);
  const RngWords b = rng_draw(seed, 0, kRngInitB, index_offset + i);
  double z0, z1, z2, z3;
  rng_box_muller(rng_uniform53(a.w[0], a.w[1]), rng_uniform53(a.w[2], a.w[3]), z0, z1);
  rng_box_muller(rng_uniform53(b.w[0], b.w[1]), rng_uniform53(b.w[2], b.w[3]), z2, z3);
  const double vx = m0 + (t00 * z0 + t01 * z1 + t02 * z2);
  const double vy = m1 + (t10 * z0 + t11 * z1 + t12 * z2);
  const double vt = m2 + (t20 * z0 + t21 * z1 + t22 * z2);
  store_pose(p, i, Pose2{rot_exp(vt), vx, vy});
  p.w[i] = 1.0;
}

// beluga_ros::Amcl::initialize_from_map (beluga_ros/include/beluga_ros/amcl.hpp:191-198,209): particle i takes the i-th draw of
// MultivariateUniformDistribution<SE2d, OccupancyGrid> (random/multivariate_uniform_distribution.hpp:145-147), weight 1.
// Same generator as random_intersperse's injected states, addressed by (step 0, global particle index).
__global__ __launch_bounds__(kBlock) void k_init_from_map(Particles p, uint64_t n, uint64_t seed, uint64_t index_offset, GridView g,
                                                          FreeCells fc) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  store_pose(p, i, random_free_state(seed, 0u, index_offset + i, g, fc));
  p.w[i] = 1.0;  // particle_traits.hpp:105
}

// ---- likelihood field on the device -----------------------------------------------------------
// LikelihoodFieldModelBase::make_likelihood_field (likelihood_field_model_base.hpp:130-185) with an EXACT Euclidean distance
// transform in place of nearest_obstacle_distance_map (distance_map.hpp:55-98), whose priority-queue wavefront hands every
// cell the obstacle of whichever neighbour reaches it first — not always the nearest one.  The device field is therefore
// <= the reference's distance (>= its likelihood) and equal at all but a few cells; mcl_set_map's default stays the
// bit-identical host wavefront (map_build.cpp), this build is selected with option field_build = 1.
//   pass 1 (k_edt_columns): per column, distance in cells to the nearest seed of the column, capped (two sweeps);
//   pass 2 (k_edt_field): per cell, min over the capped row neighbourhood of dx^2 + g^2 (row segment staged in LDS), the
//   squared metric distance to that seed as the reference computes it (double arithmetic on cell centres, float result),
//   the cap, the unknown-space overlay and the Gaussian map.
constexpr int kEdtInf = 0xFFFF;
struct EdtParams {
  uint32_t W, H;
  double resolution;
  int8_t free_value, unknown_value, occupied_value;
  int only_obstacle_boundaries, model_unknown_space;
  int cap_cells;               // seeds farther than this (per axis) cannot be within max_obstacle_distance
  float max_sq, overlay_sq;    // squared max_obstacle_distance; squared distance written over unknown space
  double amplitude, two_squared_sigma, offset;
};
__device__ __forceinline__ bool edt_obstacle_edge(const int8_t* __restrict__ cells, const EdtParams& p, uint32_t x, uint32_t y) {
  const size_t i = static_cast<size_t>(y) * p.W + x;  // occupancy_grid.hpp:191-206: occupied with a free 4-neighbour
  if (cells[i] != p.occupied_value) return false;
  return (x + 1 < p.W && cells[i + 1] == p.free_value) || (y + 1 < p.H && cells[i + p.W] == p.free_value) ||
         (x > 0 && cells[i - 1] == p.free_value) || (y > 0 && cells[i - p.W] == p.free_value);
}
__device__ __forceinline__ bool edt_seed(const int8_t* __restrict__ cells, const EdtParams& p, uint32_t x, uint32_t y) {
  return p.only_obstacle_boundaries ? edt_obstacle_edge(cells, p, x, y) : cells[static_cast<size_t>(y) * p.W + x] == p.occupied_value;
}
// column_distance[y][x] = |y - y'| of the nearest seed (x, y') of column x, kEdtInf beyond the cap; column_offset = y' - y.
__global__ __launch_bounds__(kBlock) void k_edt_columns(const int8_t* __restrict__ cells, EdtParams p, uint16_t* __restrict__ column_distance,
                                                        int16_t* __restrict__ column_offset) {
  const uint32_t x = blockIdx.x * kBlock + threadIdx.x;
  if (x >= p.W) return;
  int since = kEdtInf;  // rows since the last seed, going down the column
  for (uint32_t y = 0; y < p.H; ++y) {
    since = edt_seed(cells, p, x, y) ? 0 : (since >= p.cap_cells ? kEdtInf : since + 1);
    const size_t i = static_cast<size_t>(y) * p.W + x;
    column_distance[i] = static_cast<uint16_t>(since);
    column_offset[i] = static_cast<int16_t>(-since);
  }
  since = kEdtInf;
  for (uint32_t yy = p.H; yy > 0; --yy) {
    const uint32_t y = yy - 1;
    const size_t i = static_cast<size_t>(y) * p.W + x;
    since = column_distance[i] == 0 ? 0 : (since >= p.cap_cells ? kEdtInf : since + 1);
    if (since < column_distance[i]) {
      column_distance[i] = static_cast<uint16_t>(since);
      column_offset[i] = static_cast<int16_t>(since);
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_edt_field(const int8_t* __restrict__ cells, EdtParams p, const uint16_t* __restrict__ column_distance,
                                                      const int16_t* __restrict__ column_offset, float* __restrict__ field) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* s_g = reinterpret_cast<uint16_t*>(smem);
  const uint32_t y = blockIdx.y;
  const int x0 = static_cast<int>(blockIdx.x * kBlock), R = p.cap_cells;
  const int span = kBlock + 2 * R;
  for (int k = threadIdx.x; k < span; k += kBlock) {
    const int xs = x0 - R + k;
    s_g[k] = (xs >= 0 && xs < static_cast<int>(p.W)) ? column_distance[static_cast<size_t>(y) * p.W + xs] : static_cast<uint16_t>(kEdtInf);
  }
  __syncthreads();
  const int x = x0 + static_cast<int>(threadIdx.x);
  if (x >= static_cast<int>(p.W)) return;
  int best = 0x7FFFFFFF, best_dx = 0;
  for (int dx = 0; dx <= R && dx * dx < best; ++dx) {  // outwards: a candidate column farther than the best distance cannot win
    const int gl = s_g[threadIdx.x + R - dx], gr = s_g[threadIdx.x + R + dx];
    if (gl != kEdtInf && dx * dx + gl * gl < best) {
      best = dx * dx + gl * gl;
      best_dx = -dx;
    }
    if (gr != kEdtInf && dx * dx + gr * gr < best) {
      best = dx * dx + gr * gr;
      best_dx = dx;
    }
  }
  const size_t i = static_cast<size_t>(y) * p.W + static_cast<size_t>(x);
  float squared = p.max_sq;
  if (best != 0x7FFFFFFF) {
    const int ox = x + best_dx, oy = static_cast<int>(y) + column_offset[static_cast<size_t>(y) * p.W + ox];
    // likelihood_field_model_base.hpp:131-133 over regular_grid.hpp:87-89: cell centres in double, float result
    const double ax = (static_cast<double>(x) + 0.5) * p.resolution, ay = (static_cast<double>(static_cast<int>(y)) + 0.5) * p.resolution;
    const double bx = (static_cast<double>(ox) + 0.5) * p.resolution, by = (static_cast<double>(oy) + 0.5) * p.resolution;
    const double ddx = ax - bx, ddy = ay - by;
    const float d = static_cast<float>(ddx * ddx + ddy * ddy);
    if (d < p.max_sq) squared = d;  // distance_map.hpp:87-90
  }
  if (p.model_unknown_space) {  // :160-179
    const int8_t c = cells[i];
    const bool masked = p.only_obstacle_boundaries ? (c == p.unknown_value || (c == p.occupied_value && !edt_obstacle_edge(cells, p, x, y)))
                                                   : c == p.unknown_value;
    if (masked) squared = p.overlay_sq;
  }
  field[i] = static_cast<float>(p.amplitude * exp(-static_cast<double>(squared) / p.two_squared_sigma) + p.offset);  // :146,181-182
}

// out[k] = sum over ranks r (in order) of gathered[r][k]: the shards' scalars after an all-gather.
__global__ void k_sum_rows(const double* __restrict__ gathered, uint32_t rows, uint32_t columns, double* __restrict__ out,
                           double* __restrict__ host_mirror) {
  const uint32_t k = threadIdx.x;
  if (k >= columns) return;
  double acc = 0.0;
  for (uint32_t r = 0; r < rows; ++r) acc += gathered[r * columns + k];
  out[k] = acc;
  if (host_mirror) host_mirror[k] = acc;
}

__global__ __launch_bounds__(kBlock) void k_cube_table(const float* __restrict__ field, uint64_t cells, float unknown_value,
                                                       double* __restrict__ cube, int prob) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i > cells) return;
  const double pz = static_cast<double>(i < cells ? field[i] : unknown_value);
  cube[i] = prob ? log(pz) : pz * pz * pz;
}
__device__ __forceinline__ uint32_t palette_find(const uint32_t* __restrict__ keys, uint32_t count, uint32_t bits) {
  uint32_t lo = 0, hi = count;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < bits) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(kBlock) void k_palette_values(const uint32_t* __restrict__ keys, uint32_t count, int prob,
                                                           double* __restrict__ val) {
  const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= count) return;
  const double pz = static_cast<double>(__builtin_bit_cast(float, keys[k]));
  val[k] = prob ? log(pz) : pz * pz * pz;
}
// One thread per slot of the tiled table (border tiles and padding slots take the unknown entry).
__global__ __launch_bounds__(kBlock) void k_palette_indices(const float* __restrict__ field, uint32_t W, uint32_t H, uint32_t tiles_x,
                                                            uint32_t tiles_y, float unknown_value, const uint32_t* __restrict__ keys,
                                                            uint32_t count, uint32_t pal_base, uint16_t* __restrict__ idx) {
  const uint64_t slot = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (slot >= static_cast<uint64_t>(tiles_x) * tiles_y * 64) return;
  const uint64_t tile = slot >> 6;
  const uint32_t in = static_cast<uint32_t>(slot & 63);
  const int64_t x = (static_cast<int64_t>(tile % tiles_x) - 1) * 8 + (in >> 3), y = (static_cast<int64_t>(tile / tiles_x) - 1) * 8 + (in & 7);
  float v = unknown_value;
  if (x >= 0 && x < W && y >= 0 && y < H) v = field[static_cast<size_t>(y) * W + static_cast<size_t>(x)];
  idx[slot] = static_cast<uint16_t>(pal_base + palette_find(keys, count, __builtin_bit_cast(uint32_t, v)) * 8u);
}
// Far tiles (FieldView::far_bits).  A tile is 64 uint16 = 8 x uint4; votes[k] counts the tiles uniformly equal to entry k.
__device__ __forceinline__ bool tile_is_uniform(const uint16_t* __restrict__ idx, uint64_t tile, uint32_t* entry) {
  const uint4* t = reinterpret_cast<const uint4*>(idx) + tile * 8;
  const uint4 first = t[0];
  const uint32_t pair = first.x;
  bool same = (pair >> 16) == (pair & 0xFFFFu) && first.y == pair && first.z == pair && first.w == pair;
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const uint4 v = t[k];
    same = same && v.x == pair && v.y == pair && v.z == pair && v.w == pair;
  }
  *entry = pair & 0xFFFFu;
  return same;
}
__global__ __launch_bounds__(kBlock) void k_far_tile_votes(const uint16_t* __restrict__ idx, uint32_t tiles, uint32_t pal_base,
                                                           uint32_t count, uint32_t* __restrict__ votes) {
  const uint32_t tile = blockIdx.x * kBlock + threadIdx.x;
  uint32_t entry = 0;
  const bool uniform = tile < tiles && tile_is_uniform(idx, tile, &entry);
  const uint32_t k = (entry - pal_base) >> 3;
  // most tiles of a wave vote for the same entry: one atomic per distinct entry and wave (132 K atomics on one word took 1.5 ms)
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t pending = __builtin_amdgcn_ballot_w64(uniform && k < count);
  while (pending) {
    const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(pending));
    const uint32_t kk = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(k), static_cast<int>(leader)));
    const uint64_t same = __builtin_amdgcn_ballot_w64(uniform && k == kk) & pending;
    if (lane == leader) atomicAdd(votes + kk, static_cast<uint32_t>(__builtin_popcountll(same)));
    pending &= ~same;
  }
}
// One thread per byte of the bitmap (8 tiles of one row of tiles).
__global__ __launch_bounds__(kBlock) void k_far_tile_bits(const uint16_t* __restrict__ idx, uint32_t tiles_x, uint32_t tiles_y,
                                                          uint32_t entry, uint32_t row_bytes, uint32_t far_bytes,
                                                          uint8_t* __restrict__ bits) {
  const uint32_t at = blockIdx.x * kBlock + threadIdx.x;
  if (at >= far_bytes) return;
  const uint32_t ty = at / row_bytes, bx = at % row_bytes;
  uint32_t byte = 0;
  if (ty < tiles_y) {
    for (uint32_t k = 0; k < 8; ++k) {
      const uint32_t tx = bx * 8 + k;
      uint32_t found;
      if (tx < tiles_x && tile_is_uniform(idx, static_cast<uint64_t>(ty) * tiles_x + tx, &found) && found == entry) byte |= 1u << k;
    }
  }
  bits[at] = static_cast<uint8_t>(byte);
}
__global__ __launch_bounds__(kBlock) void k_far_tile_bits_linear(const uint16_t* __restrict__ idx, uint32_t tiles, uint32_t entry, uint32_t bytes,
                                                                 uint8_t* __restrict__ bits) {
  const uint32_t at = blockIdx.x * kBlock + threadIdx.x;
  if (at >= bytes) return;
  uint32_t byte = 0;
  for (uint32_t k = 0; k < 8; ++k) {
    const uint64_t tile = static_cast<uint64_t>(at) * 8 + k;
    uint32_t found;
    if (tile < tiles && tile_is_uniform(idx, tile, &found) && found == entry) byte |= 1u << k;
  }
  bits[at] = static_cast<uint8_t>(byte);
}
__global__ __launch_bounds__(kBlock) void k_fill(double* p, uint64_t n, double v) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

// =====================================================================================================
void launch_propagate(hipStream_t st, Particles p, uint64_t n, DiffDriveSampler smp, uint64_t seed, uint32_t step,
                      uint64_t index_offset, const double* scan_src, double* scan_dst, uint32_t scan_doubles, const SortScratch* sort,
                      const KeyFrame* frame, const double* normals_ahead, uint64_t normals_stride) {
  if (n == 0) {
    if (scan_dst && scan_doubles) launch_pull_scan(st, scan_src, scan_dst, scan_doubles);
    return;
  }
  const uint32_t nblocks = num_chunks(n);
  if (!(sort && frame) && n <= 65536) {
    hipLaunchKernelGGL(k_propagate_small, dim3(blocks_for(n)), dim3(kBlock), 0, st, p, n, smp, seed, step,
                       index_offset, scan_src, scan_dst, scan_doubles);
    return;
  }
  if (sort && frame && n < (1ull << 32))
    hipLaunchKernelGGL(k_propagate<true>, dim3(nblocks), dim3(kPropBlock), 0, st, p, n, smp, seed, step, index_offset, scan_src, scan_dst,
                       scan_doubles, *frame, sort->keys, sort->table, nblocks, normals_ahead, normals_stride);
  else
    hipLaunchKernelGGL(k_propagate<false>, dim3(nblocks), dim3(kPropBlock), 0, st, p, n, smp, seed, step, index_offset, scan_src, scan_dst,
                       scan_doubles, KeyFrame{}, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), nblocks, normals_ahead,
                       normals_stride);
}

void launch_noise_ahead(hipStream_t st, uint64_t seed, uint32_t step, uint64_t index_offset, uint64_t n, double* d_normals) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_noise_ahead, dim3(blocks_for(n)), dim3(kBlock), 0, st, seed, step, index_offset, n, d_normals);
}

void launch_pull_scan(hipStream_t st, const double* scan_src, double* scan_dst, uint32_t scan_doubles) {
  if (scan_doubles == 0) return;
  hipLaunchKernelGGL(k_pull_scan, dim3(1), dim3(kBlock), 0, st, scan_src, scan_dst, scan_doubles);
}

void launch_order_particles(hipStream_t st, Particles p, uint64_t n, const SortScratch* sort, const KeyFrame* frame, bool keys_ready,
                            uint32_t layout) {
  if (n == 0 || !sort || n >= (1ull << 32)) return;
  const uint32_t nblocks = num_chunks(n);
  if (!keys_ready) {
    const KeyFrame* device_frame = nullptr;
    if (!frame) {  // no estimate of the set on the host: bins over its bounding box
      double* partials = sort->bbox + 8;
      hipLaunchKernelGGL(k_bbox_partials, dim3(nblocks), dim3(kBlock), 0, st, p, n, partials, nblocks);
      hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(kBlock), 0, st, partials, nblocks, nblocks, sort->bbox, p, sort->frame, layout);
      device_frame = sort->frame;
    }
    hipLaunchKernelGGL(k_order_keys, dim3(nblocks), dim3(kWide), 0, st, p, n, frame ? *frame : KeyFrame{}, device_frame, sort->keys,
                       sort->table, nblocks);
  }
  const dim3 rows(kSortDigits / (kBlock / 64));
  hipLaunchKernelGGL(k_row_scan, rows, dim3(kBlock), 0, st, sort->table, nblocks, sort->totals);
  hipLaunchKernelGGL(k_sort_scatter_high, dim3(nblocks), dim3(kStable), 0, st, sort->keys, n, sort->table, nblocks, sort->totals,
                     sort->keyidx, sort->totals + kSortDigits);
  hipLaunchKernelGGL(k_sort_buckets, dim3(kSortDigits), dim3(kStable), 0, st, sort->keyidx, sort->totals, sort->totals + kSortDigits,
                     sort->perm);
}

void launch_order_ahead(hipStream_t st, uint64_t n, const SortScratch* sort) {
  if (n == 0 || !sort || n >= (1ull << 32)) return;
  const uint32_t nblocks = num_chunks(n);
  hipLaunchKernelGGL(k_key_hist, dim3(nblocks), dim3(kWide), 0, st, sort->keys, n, sort->table, nblocks);
  const dim3 rows(kSortDigits / (kBlock / 64));
  hipLaunchKernelGGL(k_row_scan, rows, dim3(kBlock), 0, st, sort->table, nblocks, sort->totals);
  hipLaunchKernelGGL(k_sort_scatter_high, dim3(nblocks), dim3(kStable), 0, st, sort->keys, n, sort->table, nblocks, sort->totals,
                     sort->keyidx, sort->totals + kSortDigits);
  hipLaunchKernelGGL(k_sort_buckets, dim3(kSortDigits), dim3(kStable), 0, st, sort->keyidx, sort->totals, sort->totals + kSortDigits,
                     sort->perm);
}

namespace {
constexpr uint32_t kFarBeamsMaxLds = 80 * 1024;  // two workgroups per CU
uint32_t far_beams_points_at(const FieldView& f) {  // the scan's place in the workgroup memory of k_reweight_lf_far_beams
  return (kFarBeamsPalShift + f.pal_base + f.pal_count * 8u + 15u) & ~15u;
}
uint32_t far_beams_lds(const FieldView& f, uint32_t B) {
  return far_beams_points_at(f) + ((B + 63u) & ~63u) * 16u + (kFarBeamsBlock / kWave) * kFarBeamsChunk * 32u;
}
// (hipFuncSetAttribute is per device)
bool far_beams_configured() {
  static bool done[64] = {};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 64) return false;
  if (!done[device]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_reweight_lf_far_beams<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kFarBeamsMaxLds)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_reweight_lf_far_beams<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kFarBeamsMaxLds)) != hipSuccess)
      return false;
    done[device] = true;
  }
  return true;
}
}  // namespace

void launch_reweight_lf(hipStream_t st, Particles p, uint64_t n, FieldView f, const double* d_points, uint32_t B, int variant,
                        const SortScratch* sort, bool scan_is_short, const Tuning& tuning, bool use_patches, PatchStats patch_stats,
                        bool dispersed, bool* far_tiles_used, uint32_t* weight_sums_written, bool* queue_used, bool unit_weights,
                        bool* far_beams_used) {
  if (far_beams_used) *far_beams_used = false;
  if (weight_sums_written) *weight_sums_written = 0;
  if (far_tiles_used) *far_tiles_used = false;
  if (queue_used) *queue_used = false;
  if (n == 0) return;
  if (variant == kLfSortedLanes && sort && n < (1ull << 32)) {
    const uint64_t cells = static_cast<uint64_t>(f.W) * f.H;
    const bool cube_ok = f.cube != nullptr && f.W < (1u << 21) && (cells + 1) * 8 < (1ull << 31);
    // One lane per particle fills the chip from ~260K particles (4096 waves).  Below that, split the scan into segments
    // (second grid dimension) as long as a segment keeps >= 64 beams, and add the segment sums in a second pass.
    uint32_t segments = 1;
    const uint64_t waves = (n + kWave - 1) / kWave;
    if (sort->partial && waves < 4096) {
      segments = static_cast<uint32_t>(std::min<uint64_t>((4096 + waves - 1) / waves, kLfMaxSegments));
      segments = std::max(1u, std::min(segments, B / 64));
    }
    const uint32_t per_segment = (B + segments - 1) / segments;
    double* partial = segments > 1 ? sort->partial : nullptr;
    const dim3 grid(blocks_for(n), segments);
    const size_t pal_lds = static_cast<size_t>(f.pal_base) + static_cast<size_t>(f.pal_count) * sizeof(double);
    const bool palette_ok = tuning.lf_table == 0 && f.pal_idx != nullptr && f.pal_count > 0 && pal_lds <= 65536;
    if (palette_ok) {
      const dim3 pgrid(static_cast<unsigned>((n + kPalBlock - 1) / kPalBlock), segments);
      // The FMA variant needs a scan within 8192 cells of the sensor and a grid below 2^14 cells per side (its exact
      // fallback handles everything else inside the kernel); tuning.lf_fast = 0 forces the separately rounded arithmetic.
      const bool fast = tuning.lf_fast != 0 && scan_is_short && f.W < 16384 && f.H < 16384;
      const uint32_t patch_base = (static_cast<uint32_t>(pal_lds) + 15u) & ~15u;
      const size_t patch_lds = patch_base + kPatchLdsBytes;
      if (fast && use_patches && patch_lds <= 65536) {
        const uint32_t per_group = kPatchParticles;
        const unsigned groups_x = static_cast<unsigned>((n + per_group - 1) / per_group);
        if (segments > 1) patch_stats.weight_sums = nullptr;  // the segments' sums are combined by k_lf_combine
        // A queue of blocks and as many workgroups as stay resident (three per CU) instead of a workgroup per block, where the launch
        // has more blocks than that: see k_reweight_lf_patch.
        const uint32_t cus = tuning.device_cus > 0 ? static_cast<uint32_t>(tuning.device_cus) : 256u;
        const uint32_t resident = tuning.lf_queue_grid > 0 ? static_cast<uint32_t>(tuning.lf_queue_grid) : 3u * cus;
        const PatchArgs args{p.w, n, f, d_points, B, sort->perm, p.pose, partial, per_segment, patch_base, patch_stats, groups_x,
                             tuning.lf_ends_first != 0 ? 1u : 0u, (unit_weights && segments == 1) ? 1u : 0u};
        if (tuning.lf_queue != 0 && segments == 1 && patch_stats.arrivals != nullptr && groups_x > resident) {
          if (queue_used) *queue_used = true;
          hipLaunchKernelGGL(k_reweight_lf_patch<true>, dim3(resident), dim3(kPatchBlock), patch_lds, st, args);
        } else {
          hipLaunchKernelGGL(k_reweight_lf_patch<false>, dim3(groups_x, segments), dim3(kPatchBlock), patch_lds, st, args);
        }
        if (weight_sums_written && patch_stats.weight_sums) *weight_sums_written = groups_x;
      }
      else if (fast && dispersed && tuning.lf_far_tiles != 0 && tuning.lf_dispersed == 2 && f.far_linear != nullptr && B > 0 &&
               f.far_linear_bytes <= kFarBeamsPalShift && far_beams_lds(f, B) <= kFarBeamsMaxLds && far_beams_configured()) {
        // lanes over the beams of one pose, the poses in the position-major order (k_reweight_lf_far_beams); a small set needs no
        // segments of the scan to fill the chip: fewer poses per wave
        segments = 1;
        const uint64_t waves_wanted = 256ull * 4 * kFarBeamsWaves;
        const uint32_t per_wave_auto = static_cast<uint32_t>(std::min<uint64_t>(32, std::max<uint64_t>(kFarBeamsPoses, n / waves_wanted)));
        const uint32_t per_wave = tuning.lf_far_beams_per_wave > 0 ? static_cast<uint32_t>(tuning.lf_far_beams_per_wave) : per_wave_auto;
        const uint64_t per_block = static_cast<uint64_t>(per_wave) * (kFarBeamsBlock / kWave);
        const unsigned blocks = (static_cast<unsigned>((n + per_block - 1) / per_block) + 7u) & ~7u;
        if (f.prob)
          hipLaunchKernelGGL(k_reweight_lf_far_beams<true>, dim3(blocks), dim3(kFarBeamsBlock), far_beams_lds(f, B), st, p.w, n, f,
                             reinterpret_cast<const double2*>(d_points), B, sort->perm, p.pose, far_beams_points_at(f), per_wave,
                             unit_weights ? 1u : 0u);
        else
          hipLaunchKernelGGL(k_reweight_lf_far_beams<false>, dim3(blocks), dim3(kFarBeamsBlock), far_beams_lds(f, B), st, p.w, n, f,
                             reinterpret_cast<const double2*>(d_points), B, sort->perm, p.pose, far_beams_points_at(f), per_wave,
                             unit_weights ? 1u : 0u);
        if (far_tiles_used) *far_tiles_used = true;
        if (far_beams_used) *far_beams_used = true;
      }
      else if (fast && dispersed && tuning.lf_far_tiles != 0 && f.far_bits != nullptr && patch_base + f.far_bytes <= 65536) {
        const dim3 fgrid((pgrid.x + 7u) & ~7u, segments);
        hipLaunchKernelGGL((k_reweight_lf_palette<true, true>), fgrid, dim3(kPalBlock), patch_base + f.far_bytes, st, p.w, n, f, d_points, B,
                           sort->perm, p.pose, partial, per_segment, patch_base);
        if (far_tiles_used) *far_tiles_used = true;
      } else if (fast)
        hipLaunchKernelGGL(k_reweight_lf_palette<true>, pgrid, dim3(kPalBlock), pal_lds, st, p.w, n, f, d_points, B, sort->perm, p.pose,
                           partial, per_segment, 0u);
      else
        hipLaunchKernelGGL(k_reweight_lf_palette<false>, pgrid, dim3(kPalBlock), pal_lds, st, p.w, n, f, d_points, B, sort->perm, p.pose,
                           partial, per_segment, 0u);
    } else if (cube_ok) {
      hipLaunchKernelGGL(k_reweight_lf_sorted<true>, grid, dim3(kBlock), 0, st, p.w, n, f, d_points, B, sort->perm, p.pose, partial,
                         per_segment);
    } else {
      hipLaunchKernelGGL(k_reweight_lf_sorted<false>, grid, dim3(kBlock), 0, st, p.w, n, f, d_points, B, sort->perm, p.pose, partial,
                         per_segment);
    }
    if (segments > 1)
      hipLaunchKernelGGL(k_lf_combine, dim3(blocks_for(n)), dim3(kBlock), 0, st, p.w, n, sort->perm, partial, segments, f.prob);
  } else if (variant == kLfBeamLanes && tuning.lf_table == 0 && f.pal_idx != nullptr && f.pal_count > 0 &&
             static_cast<size_t>(f.pal_base) + static_cast<size_t>(f.pal_count) * sizeof(double) <= 65536) {
    // particles per wave: enough waves to fill the chip (4096) before a wave takes a second particle
    const uint32_t per_wave = static_cast<uint32_t>(std::min<uint64_t>(kWave, std::max<uint64_t>(1, (n + 4095) / 4096)));
    const uint64_t tiles = (n + per_wave - 1) / per_wave;
    const dim3 grid(static_cast<unsigned>((tiles + (kBeamsBlock / kWave) - 1) / (kBeamsBlock / kWave)));
    const size_t pal_lds = static_cast<size_t>(f.pal_base) + static_cast<size_t>(f.pal_count) * sizeof(double);
    hipLaunchKernelGGL(k_reweight_lf_beams, grid, dim3(kBeamsBlock), pal_lds, st, p, n, f, reinterpret_cast<const double2*>(d_points), B,
                       per_wave);
  } else {
    // no order (option lf_variant 0 / 1, small sets whose field has too many distinct values for a palette, sets beyond 2^32 particles):
    // a lane per particle in index order over the f32 field
    hipLaunchKernelGGL(k_reweight_lf_sorted<false>, dim3(blocks_for(n)), dim3(kBlock), 0, st, p.w, n, f, d_points, B,
                       static_cast<const uint32_t*>(nullptr), p.pose, static_cast<double*>(nullptr), 0u);
  }
}

}  // namespace mcl
// Nonzero for a measurement build of the kernels (tools/build_variant.sh: ablations compute nonsense by design, timing builds
// distort): beluga_amd/capi.py refuses to load one as the product library unless told so.
extern "C" int mcl_measurement_build(void) {
#ifdef MCL_MEASUREMENT_BUILD
  return 1;
#else
  return 0;
#endif
}
namespace mcl {
void launch_lf_combine(hipStream_t st, double* w, uint64_t n, const uint32_t* perm, const double* partial, uint32_t segments, int mode) {
  hipLaunchKernelGGL(k_lf_combine, dim3(blocks_for(n)), dim3(kBlock), 0, st, w, n, perm, partial, segments, mode);
}

void launch_weight_sum(hipStream_t st, const double* w, uint64_t n, double* d_partials, double* d_out, double* host_mirror) {
  const uint32_t chunks = num_chunks(n);
  if (chunks) hipLaunchKernelGGL(k_chunk_sum, dim3(chunks), dim3(kBlock), 0, st, w, n, d_partials);
  hipLaunchKernelGGL(k_final_rows, dim3(1), dim3(kBlock), 0, st, d_partials, chunks, chunks, d_out, host_mirror, Completion{});
}

void launch_normalize(hipStream_t st, double* w, uint64_t n, const double* d_factor, double* d_chunk_sum, double* d_chunk_sumsq,
                      double* d_out, double* host_mirror) {
  constexpr bool store_weights = true;
  const uint32_t chunks = num_chunks(n);
  if (chunks)
    hipLaunchKernelGGL(k_normalize, dim3(chunks), dim3(kBlock), 0, st, w, n, d_factor, d_chunk_sum, d_chunk_sumsq,
                       static_cast<const double*>(nullptr), 0u, static_cast<double*>(nullptr), static_cast<double*>(nullptr), store_weights ? 1 : 0);
  // d_chunk_sum and d_chunk_sumsq are adjacent rows of one [2][stride] buffer (see context.hip)
  hipLaunchKernelGGL(k_final_rows, dim3(2), dim3(kBlock), 0, st, d_chunk_sum, chunks, static_cast<uint32_t>(d_chunk_sumsq - d_chunk_sum),
                     d_out, host_mirror, Completion{});
}

// actions::normalize by the set's own total (normalize.hpp:70): weight chunk sums, then the division with the total added
// up inside k_normalize.  d_sums[0] = total before, d_sums[1], d_sums[2] = sum and sum of squares after (mirrored likewise).
// finalize == false: the totals after (and the recovery estimator) are left to the kernel that follows — launch_cdf with its
// finalize arguments, or launch_norm_finalize.
void launch_sum_and_normalize(hipStream_t st, double* w, uint64_t n, double* d_partials, double* d_chunk_sum, double* d_chunk_sumsq,
                              double* d_sums, double* host_mirror, bool finalize, const double* known_partials, uint32_t known_count,
                              bool store_weights) {
  const uint32_t chunks = num_chunks(n);
  if (chunks) {
    // known_partials: sums whose total is the factor already exist (the LF kernel's workgroup sums): no pass to add the weights up
    if (!known_partials) hipLaunchKernelGGL(k_chunk_sum, dim3(chunks), dim3(kBlock), 0, st, w, n, d_partials);
    const double* partials = known_partials ? known_partials : d_partials;
    const uint32_t count = known_partials ? known_count : chunks;
    // Every workgroup adding the partial sums up for itself saves a launch while they are few (2232 at 1M particles); at 10M particles (22k sums x
    // 4.9k workgroups) it was most of the kernel's time: then one workgroup adds them up first (same order, same bits).
    if (count > 4096u) {
      hipLaunchKernelGGL(k_final_rows, dim3(1), dim3(kBlock), 0, st, partials, count, count, d_sums, host_mirror, Completion{});
      hipLaunchKernelGGL(k_normalize, dim3(chunks), dim3(kBlock), 0, st, w, n, static_cast<const double*>(d_sums), d_chunk_sum, d_chunk_sumsq,
                         static_cast<const double*>(nullptr), 0u, static_cast<double*>(nullptr), static_cast<double*>(nullptr), store_weights ? 1 : 0);
    } else {
      hipLaunchKernelGGL(k_normalize, dim3(chunks), dim3(kBlock), 0, st, w, n, static_cast<const double*>(nullptr), d_chunk_sum,
                         d_chunk_sumsq, partials, count, d_sums, host_mirror, store_weights ? 1 : 0);
    }
  } else {
    hipLaunchKernelGGL(k_final_rows, dim3(1), dim3(kBlock), 0, st, d_partials, chunks, chunks, d_sums, host_mirror, Completion{});
  }
  if (finalize)
    hipLaunchKernelGGL(k_final_rows, dim3(2), dim3(kBlock), 0, st, d_chunk_sum, chunks, static_cast<uint32_t>(d_chunk_sumsq - d_chunk_sum),
                       d_sums + 1, host_mirror ? host_mirror + 1 : nullptr, Completion{});
}

namespace {
NormFinalize make_norm_finalize(const double* d_chunk_sum, const double* d_chunk_sumsq, uint64_t n, double* d_sums, double* sums_mirror,
                                const RecoveryPolicy* policy) {
  NormFinalize f{};
  f.chunk_sum = d_chunk_sum;
  f.chunk_sumsq = d_chunk_sumsq;
  f.chunks = num_chunks(n);
  f.d_sums = d_sums;
  f.sums_mirror = sums_mirror;
  f.n = n;
  if (policy) {
    f.policy = 1;
    f.alpha_slow = policy->alpha_slow;
    f.alpha_fast = policy->alpha_fast;
    f.resampling = policy->resampling;
    f.d_policy = policy->d_policy;
    f.policy_mirror = policy->host_mirror;
  }
  return f;
}
}  // namespace

void launch_norm_finalize(hipStream_t st, const double* d_chunk_sum, const double* d_chunk_sumsq, uint64_t n, double* d_sums,
                          double* sums_mirror, const RecoveryPolicy* policy) {
  hipLaunchKernelGGL(k_norm_finalize, dim3(1), dim3(kBlock), 0, st,
                     make_norm_finalize(d_chunk_sum, d_chunk_sumsq, n, d_sums, sums_mirror, policy));
}

void launch_cdf(hipStream_t st, const double* w, uint64_t n, double* d_chunk_sum, double* d_chunk_offset, double* cdf,
                double* d_total, double* tree_levels, const double* known_chunk_sum, const double* finalize_sumsq, double* finalize_sums,
                double* finalize_mirror, const RecoveryPolicy* policy, const double* d_factor) {
  const uint32_t chunks = num_chunks(n);
  if (!chunks) return;
  const double* sums = known_chunk_sum;
  if (!sums) {
    hipLaunchKernelGGL(k_chunk_sum, dim3(chunks), dim3(kBlock), 0, st, w, n, d_chunk_sum);
    sums = d_chunk_sum;
  }
  const bool replay = chunks <= 4 * kBlock;  // every workgroup re-derives its own offset: no single-workgroup scan in between
  if (!replay)
    hipLaunchKernelGGL(k_scan_chunks<double>, dim3(1), dim3(kBlock), 0, st, sums, chunks, d_chunk_offset,
                       static_cast<double*>(nullptr), static_cast<const double*>(nullptr));
  NormFinalize fin{};
  if (finalize_sums) fin = make_norm_finalize(sums, finalize_sumsq, n, finalize_sums, finalize_mirror, policy);
  hipLaunchKernelGGL(k_cdf, dim3(chunks), dim3(kBlock), 0, st, w, n, d_chunk_offset, cdf, d_total,
                     make_cdf_tree(cdf, tree_levels, n), tree_levels, replay ? sums : static_cast<const double*>(nullptr), chunks, fin, d_factor);
}

// launch_sum_and_normalize (known partial sums, the totals left to this kernel) + launch_cdf (with its finalize arguments) in ONE launch:
// k_normalize_cdf.  scan_state: 8 + 4 * kScanFusedMaxChunks words of 8 bytes, zero when the context allocated them; epoch != 0 and different
// from the previous launch's.  Returns false (nothing launched) where the set is beyond what the kernel takes.
bool launch_normalize_cdf(hipStream_t st, double* w, uint64_t n, double* d_partials, const double* known_partials, uint32_t known_count,
                          double* d_sums, double* sums_mirror, double* d_chunk_sum, double* d_chunk_sumsq, bool write_weights, double* cdf,
                          double* d_total, double* tree_levels, const RecoveryPolicy* policy, unsigned long long* scan_state, uint32_t epoch) {
  static_assert(kScanStateWords >= 8 + 4 * kScanFusedMaxChunks, "a granule quadruple per chunk behind the ticket's line");
  const uint32_t chunks = num_chunks(n);
  if (chunks == 0 || chunks > kScanFusedMaxChunks || (known_partials && known_count > 4096u)) return false;
  if (!known_partials) {
    hipLaunchKernelGGL(k_chunk_sum, dim3(chunks), dim3(kBlock), 0, st, w, n, d_partials);
    known_partials = d_partials;
    known_count = chunks;
  }
  const NormFinalize fin = make_norm_finalize(d_chunk_sum, d_chunk_sumsq, n, d_sums + 1, sums_mirror ? sums_mirror + 1 : nullptr, policy);
  const ScanState state{reinterpret_cast<unsigned int*>(scan_state), scan_state + 8, epoch};
  hipLaunchKernelGGL(k_normalize_cdf, dim3(chunks), dim3(kBlock), 0, st, w, n, static_cast<const double*>(nullptr), known_partials, known_count,
                     d_sums, sums_mirror, d_chunk_sum, d_chunk_sumsq, chunks, write_weights ? 1 : 0, cdf, d_total,
                     make_cdf_tree(cdf, tree_levels, n), tree_levels, fin, state);
  return true;
}

namespace {
// Which sampled levels of the search tree the draw kernel stages in LDS: all from `first` on, as many as fit.
void draw_staging(const CdfTree& t, int& first, uint32_t& doubles) {
  first = t.depth;
  doubles = 0;
  while (first > 0 && t.offset[t.depth - 1] + t.size[t.depth - 1] - t.offset[first - 1] <= kDrawStageMax) {
    --first;
    doubles = t.offset[t.depth - 1] + t.size[t.depth - 1] - t.offset[first];  // the levels' padding included
  }
}
}  // namespace

void launch_resample_draw(hipStream_t st, Particles src, CdfTree cdf, const double* d_total, Particles dst,
                          ResampleArgs a, GridView g, FreeCells fc, HashParams hp, unsigned long long* d_hashes) {
  if (a.count == 0) return;
  int first;
  uint32_t doubles;
  draw_staging(cdf, first, doubles);
  const unsigned blocks = static_cast<unsigned>((a.count + kDrawBlock - 1) / kDrawBlock);
  hipLaunchKernelGGL(k_resample_draw<false>, dim3(blocks), dim3(kDrawBlock), doubles * sizeof(double), st, src, cdf, d_total, dst, a, g,
                     fc, hp, d_hashes, 0.0, 0.0, static_cast<double*>(nullptr), 0u, first, doubles, DrawFold{}, DrawNormals{});
}

// The draw plus the estimate sums of the set it produces: d_partials needs 9 * ceil(count / 1024) doubles.
void launch_resample_draw_and_estimate(hipStream_t st, Particles src, CdfTree cdf, const double* d_total, Particles dst, ResampleArgs a,
                                       GridView g, FreeCells fc, HashParams hp, double pivot_x, double pivot_y, double* d_partials,
                                       double* d_sums, double* host_mirror, const Completion* done, unsigned int* fold_ticket,
                                       double* normals_ahead, uint64_t normals_stride, uint64_t normals_index_offset, uint32_t normals_step,
                                       uint32_t* keys_ahead, const DiffDriveSampler* predicted, const KeyFrame* frame_ahead) {
  int first;
  uint32_t doubles;
  draw_staging(cdf, first, doubles);
  const unsigned blocks = static_cast<unsigned>((a.count + kDrawBlock - 1) / kDrawBlock);
  // fold_ticket: the sums are added up by the draw's own last workgroup (k_resample_draw, DrawFold) while one workgroup reads them back in
  // a few loads per thread (4096 workgroups = 4M particles: 36 loads); beyond that, and without a ticket word, k_final_rows follows.
  const bool fold = fold_ticket != nullptr && blocks > 0 && blocks <= 4096u;
  if (blocks) {
    DrawFold f{};
    if (fold) f = DrawFold{fold_ticket, d_sums, host_mirror, done ? *done : Completion{}};
    hipLaunchKernelGGL(k_resample_draw<true>, dim3(blocks), dim3(kDrawBlock), doubles * sizeof(double), st, src, cdf, d_total, dst, a, g,
                       fc, hp, static_cast<unsigned long long*>(nullptr), pivot_x, pivot_y, d_partials, blocks, first, doubles, f,
                       DrawNormals{normals_ahead, normals_stride, normals_index_offset, normals_step,
                                   (normals_ahead && predicted && frame_ahead) ? keys_ahead : nullptr,
                                   predicted ? *predicted : DiffDriveSampler{}, frame_ahead ? *frame_ahead : KeyFrame{}});
  }
  if (!fold)
    hipLaunchKernelGGL(k_final_rows, dim3(9), dim3(kBlock), 0, st, d_partials, blocks, blocks, d_sums, host_mirror, done ? *done : Completion{});
}

void launch_resample_targets(hipStream_t st, uint64_t seed, uint32_t step, double p, double total, uint64_t first_slot,
                             uint64_t count, uint64_t n_free, double* d_targets, const double* d_plan) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_resample_targets, dim3(blocks_for(count)), dim3(kBlock), 0, st, seed, step, p, total, first_slot, count, n_free,
                     d_targets, d_plan);
}

// What every rank derives from the gathered shard statistics [world][3] = {shard CDF total, sum and sum of squares of the
// shard's normalised weights}, in rank order, without a host round trip (fixed-size cycles): the intervals of the global CDF
// (ends[r], offsets[r]), d_plan = {global total, random state probability}, the totals of the normalised weights, and one step
// of the recovery estimator on their average (thrun_recovery_probability_estimator.hpp:69-89; reset when the cycle resamples
// with p > 0: amcl_core.hpp:184-186).  Same additions in the same order on every rank.
__global__ void k_shard_plan(const double* __restrict__ stats, uint32_t world, NormFinalize fin, double* __restrict__ intervals,
                             double* __restrict__ d_plan) {
  if (threadIdx.x != 0) return;
  double run = 0.0, norm_sum = 0.0, norm_sumsq = 0.0;
  for (uint32_t r = 0; r < world; ++r) {
    intervals[world + r] = run;
    run += stats[3 * r];
    intervals[r] = run;
    norm_sum += stats[3 * r + 1];
    norm_sumsq += stats[3 * r + 2];
  }
  fin.d_sums[0] = norm_sum;
  fin.d_sums[1] = norm_sumsq;
  if (fin.sums_mirror) {
    fin.sums_mirror[0] = norm_sum;
    fin.sums_mirror[1] = norm_sumsq;
  }
  recovery_policy_step(norm_sum, fin);
  d_plan[0] = run;
  d_plan[1] = fin.d_policy[2];
}
void launch_shard_plan(hipStream_t st, const double* d_stats, uint32_t world, uint64_t n_total, double* d_sums, double* sums_mirror,
                       const RecoveryPolicy& policy, double* d_intervals, double* d_plan) {
  NormFinalize fin{};
  fin.d_sums = d_sums;
  fin.sums_mirror = sums_mirror;
  fin.n = n_total;
  fin.policy = 1;
  fin.alpha_slow = policy.alpha_slow;
  fin.alpha_fast = policy.alpha_fast;
  fin.resampling = policy.resampling;
  fin.d_policy = policy.d_policy;
  fin.policy_mirror = policy.host_mirror;
  hipLaunchKernelGGL(k_shard_plan, dim3(1), dim3(64), 0, st, d_stats, world, fin, d_intervals, d_plan);
}

void launch_route_targets(hipStream_t st, const double* d_targets, uint64_t count, const double* d_ends, const double* d_offsets,
                          uint32_t world, uint32_t self_rank, uint8_t* d_dest, uint32_t* d_block_hist, uint32_t* d_chunk_sum,
                          uint32_t* d_chunk_off, double* d_send_targets, uint32_t* d_order, long long* d_counts, uint32_t pad_capacity,
                          double* d_overflow) {
  const uint32_t nblocks = num_chunks(count);
  if (nblocks == 0) {
    (void)hipMemsetAsync(d_counts, 0, sizeof(long long) * world, st);
    return;
  }
  hipLaunchKernelGGL(k_route_hist, dim3(nblocks), dim3(kBlock), 0, st, d_targets, count, d_ends, world, self_rank, d_dest, d_block_hist,
                     nblocks, pad_capacity != 0 ? 1 : 0);
  const uint32_t m = world * nblocks;
  const uint32_t mchunks = num_chunks(m);
  hipLaunchKernelGGL(k_u32_chunk_sum, dim3(mchunks), dim3(kBlock), 0, st, d_block_hist, m, d_chunk_sum);
  hipLaunchKernelGGL(k_scan_chunks<uint32_t>, dim3(1), dim3(kBlock), 0, st, d_chunk_sum, mchunks, d_chunk_off,
                     static_cast<uint32_t*>(nullptr), static_cast<const uint32_t*>(nullptr));
  hipLaunchKernelGGL(k_u32_exclusive_apply, dim3(mchunks), dim3(kBlock), 0, st, d_block_hist, m, d_chunk_off);
  hipLaunchKernelGGL(k_route_counts, dim3(1), dim3(kMaxRanks), 0, st, d_block_hist, nblocks, world, count, d_counts);
  hipLaunchKernelGGL(k_route_scatter, dim3(nblocks), dim3(kBlock), 0, st, d_targets, count, d_offsets, world, d_dest, d_block_hist,
                     nblocks, d_send_targets, d_order, pad_capacity, d_overflow);
}
void launch_gather_by_cdf_aos(hipStream_t st, Particles src, CdfTree cdf, const double* d_targets, uint64_t m,
                              double* d_out) {
  if (m == 0) return;
  hipLaunchKernelGGL(k_gather_by_cdf_aos, dim3(blocks_for(m)), dim3(kBlock), 0, st, src, cdf, d_targets, m,
                     reinterpret_cast<double4*>(d_out));
}
void launch_commit_routed(hipStream_t st, Particles dst, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count,
                          const double* d_replies, const uint32_t* d_order, const double* d_targets, GridView g, FreeCells fc) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_commit_routed, dim3(blocks_for(count)), dim3(kBlock), 0, st, dst, seed, step, first_slot, count,
                     reinterpret_cast<const double4*>(d_replies), d_order, d_targets, g, fc);
}

void launch_commit_injected(hipStream_t st, Particles dst, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count,
                            const double* d_targets, GridView g, FreeCells fc) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_commit_injected, dim3(blocks_for(count)), dim3(kBlock), 0, st, dst, seed, step, first_slot, count, d_targets, g, fc);
}

void launch_finish_candidates(hipStream_t st, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count, const double* d_replies,
                              const uint32_t* d_order, const double* d_targets, GridView g, FreeCells fc, HashParams hp,
                              double* d_states, unsigned long long* d_hashes) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_finish_candidates, dim3(blocks_for(count)), dim3(kBlock), 0, st, seed, step, first_slot, count,
                     reinterpret_cast<const double4*>(d_replies), d_order, d_targets, g, fc, hp, reinterpret_cast<double4*>(d_states),
                     d_hashes);
}

void launch_kld_insert(hipStream_t st, const unsigned long long* d_hashes, uint64_t first, uint64_t count, KldTable t) {
  if (count == 0) return;
  hipLaunchKernelGGL(k_kld_insert, dim3(num_chunks(count)), dim3(kBlock), 0, st, d_hashes, first, count, t);
}

void launch_kld_scan(hipStream_t st, const unsigned long long* d_hashes, uint64_t first, uint64_t count, KldTable t,
                     uint32_t* d_flags_scan, uint32_t* d_chunk_sum, uint32_t* d_chunk_offset, const uint32_t* d_k_base,
                     uint32_t* d_k_total, uint64_t min_particles, double epsilon, double z, unsigned long long* d_first_fail) {
  if (count == 0) return;
  const uint32_t chunks = num_chunks(count);
  hipLaunchKernelGGL(k_kld_flags, dim3(chunks), dim3(kBlock), 0, st, d_hashes, first, count, t, d_flags_scan, d_chunk_sum);
  hipLaunchKernelGGL(k_scan_chunks<uint32_t>, dim3(1), dim3(kBlock), 0, st, d_chunk_sum, chunks, d_chunk_offset, d_k_total,
                     d_k_base);
  hipLaunchKernelGGL(k_kld_check, dim3(chunks), dim3(kBlock), 0, st, first, count, d_flags_scan, d_chunk_offset, min_particles,
                     2 * epsilon, z, d_first_fail);
}

bool launch_small_tail(hipStream_t st, const SmallTail& t) {
  if (t.n == 0 || t.n > kSmallMax || t.max_particles == 0 || t.max_particles > kSmallMax) return false;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_tail), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kSmallLdsBytes)) != hipSuccess)
      return false;
    configured = true;
  }
  SmallTailArgs a{};
  a.src = t.src;
  a.dst = t.dst;
  a.n = t.n;
  a.min_particles = t.min_particles;
  a.max_particles = t.max_particles;
  a.seed = t.seed;
  a.step = t.step;
  a.fires = t.fires ? 1 : 0;
  a.selective = t.selective ? 1 : 0;
  a.adaptive = t.min_particles < t.max_particles ? 1 : 0;
  a.alpha_slow = t.alpha_slow;
  a.alpha_fast = t.alpha_fast;
  a.slow = t.slow;
  a.fast = t.fast;
  a.two_epsilon = 2.0 * t.kld_epsilon;
  a.z = t.kld_z;
  a.hp = t.hp;
  a.g = t.g;
  a.fc = t.fc;
  a.pivot_x = t.pivot_x;
  a.pivot_y = t.pivot_y;
  a.out = t.mirror;
  a.d_out = t.d_scalars;
  a.done_flag = t.done_flag;
  a.done_seq = t.done_seq;
  hipLaunchKernelGGL(k_small_tail, dim3(1), dim3(kSmallBlock), kSmallLdsBytes, st, a);
  return true;
}

void launch_estimate_sums(hipStream_t st, Particles p, uint64_t n, double pivot_x, double pivot_y, double* d_partials,
                          double* d_out, double* host_mirror) {
  const uint32_t chunks = num_chunks(n);
  if (chunks) hipLaunchKernelGGL(k_estimate_partials, dim3(chunks), dim3(kBlock), 0, st, p, n, pivot_x, pivot_y, d_partials, chunks);
  hipLaunchKernelGGL(k_final_rows, dim3(kEstK), dim3(kBlock), 0, st, d_partials, chunks, chunks, d_out, host_mirror, Completion{});
}

void launch_cluster_cells(hipStream_t st, Particles p, uint64_t n, HashParams hp, unsigned long long* d_hashes,
                          unsigned long long* t_keys, unsigned int* t_first, double* t_wsum, unsigned int* t_count,
                          unsigned int* t_cluster, uint64_t capacity, unsigned long long* c_key, unsigned int* c_first,
                          unsigned int* c_count, unsigned int* c_slot, double* c_wsum, double* c_state, unsigned int* c_size,
                          unsigned int list_capacity, bool table_ready) {
  if (n == 0) return;
  const CellTable t{t_keys, t_first, t_wsum, t_count, t_cluster, capacity};
  if (!table_ready) {  // clear, hash, aggregate; table_ready: only the compaction again (into a larger list)
    hipLaunchKernelGGL(k_cell_table_clear, dim3(blocks_for(capacity)), dim3(kBlock), 0, st, t, c_size);
    hipLaunchKernelGGL(k_cluster_hash, dim3(blocks_for(n)), dim3(kBlock), 0, st, p, n, hp, d_hashes);
    hipLaunchKernelGGL(k_cell_aggregate, dim3(num_chunks(n)), dim3(kBlock), 0, st, d_hashes, p.w, n, t);
  }
  const CellList out{c_key, c_first, c_count, c_slot, c_wsum, reinterpret_cast<double4*>(c_state), c_size};
  hipLaunchKernelGGL(k_cell_compact, dim3(blocks_for(capacity)), dim3(kBlock), 0, st, t, p, out, list_capacity);
}
bool launch_small_cluster_cells(hipStream_t st, Particles p, uint64_t n, HashParams hp, unsigned long long* c_key, unsigned int* c_first,
                                unsigned int* c_count, unsigned int* c_slot, double* c_wsum, double* c_state, unsigned int* c_size,
                                unsigned int* size_mirror) {
  if (n == 0 || n > kSmallMax) return false;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_cluster_cells), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kSmallClusterLdsBytes)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_cluster_sums), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(kSmallClusterLdsBytes)) != hipSuccess)
      return false;
    configured = true;
  }
  const CellList out{c_key, c_first, c_count, c_slot, c_wsum, reinterpret_cast<double4*>(c_state), c_size};
  hipLaunchKernelGGL(k_small_cluster_cells, dim3(1), dim3(kSmallBlock), kSmallClusterLdsBytes, st, p, static_cast<uint32_t>(n), hp, out, size_mirror);
  return true;
}
void launch_small_cluster_sums(hipStream_t st, Particles p, uint64_t n, HashParams hp, const unsigned long long* d_keys,
                               const unsigned int* d_cluster, uint32_t cells, unsigned int wanted, double pivot_x, double pivot_y, double* d_out,
                               double* host_mirror) {
  hipLaunchKernelGGL(k_small_cluster_sums, dim3(1), dim3(kSmallBlock), kSmallClusterLdsBytes, st, p, static_cast<uint32_t>(n), hp, d_keys, d_cluster,
                     cells, wanted, pivot_x, pivot_y, d_out, host_mirror);
}
void launch_cell_set_cluster(hipStream_t st, const unsigned int* d_slot, const unsigned int* d_cluster, uint32_t m,
                             unsigned int* t_cluster) {
  if (m == 0) return;
  hipLaunchKernelGGL(k_cell_set_cluster, dim3(blocks_for(m)), dim3(kBlock), 0, st, d_slot, d_cluster, m, t_cluster);
}
void launch_estimate_sums_cluster(hipStream_t st, Particles p, uint64_t n, const unsigned long long* d_hashes,
                                  unsigned long long* t_keys, unsigned int* t_cluster, uint64_t capacity, unsigned int wanted,
                                  double pivot_x, double pivot_y, double* d_partials, double* d_out, double* host_mirror) {
  const uint32_t chunks = num_chunks(n);
  const CellTable t{t_keys, nullptr, nullptr, nullptr, t_cluster, capacity};
  if (chunks)
    hipLaunchKernelGGL(k_estimate_partials_cluster, dim3(chunks), dim3(kBlock), 0, st, p, n, d_hashes, t, wanted, pivot_x, pivot_y,
                       d_partials, chunks);
  hipLaunchKernelGGL(k_final_rows, dim3(kEstK), dim3(kBlock), 0, st, d_partials, chunks, chunks, d_out, host_mirror, Completion{});
}

void launch_init_normal(hipStream_t st, Particles p, uint64_t n, const double mean[3], const double T[9], uint64_t seed,
                        uint64_t index_offset) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_init_normal, dim3(blocks_for(n)), dim3(kBlock), 0, st, p, n, mean[0], mean[1], mean[2], T[0], T[1], T[2],
                     T[3], T[4], T[5], T[6], T[7], T[8], seed, index_offset);
}

void launch_init_from_map(hipStream_t st, Particles p, uint64_t n, uint64_t seed, uint64_t index_offset, GridView g, FreeCells fc) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_init_from_map, dim3(blocks_for(n)), dim3(kBlock), 0, st, p, n, seed, index_offset, g, fc);
}

bool launch_build_field(hipStream_t st, const int8_t* d_cells, uint32_t W, uint32_t H, double resolution, int8_t free_value,
                        int8_t unknown_value, int8_t occupied_value, const FieldBuildParams& fp, uint16_t* d_column_distance,
                        int16_t* d_column_offset, float* d_field) {
  constexpr double kPiD = 3.14159265358979323846264338327950288;
  EdtParams p{};
  p.W = W;
  p.H = H;
  p.resolution = resolution;
  p.free_value = free_value;
  p.unknown_value = unknown_value;
  p.occupied_value = occupied_value;
  p.only_obstacle_boundaries = fp.only_obstacle_boundaries;
  p.model_unknown_space = fp.model_unknown_space;
  const double cells_in_reach = std::ceil(fp.max_obstacle_distance / resolution) + 1.0;
  if (!(cells_in_reach <= kFieldBuildMaxReach) || H >= 32768) return false;  // wide caps: the host build
  p.cap_cells = static_cast<int>(cells_in_reach);
  p.max_sq = static_cast<float>(fp.max_obstacle_distance * fp.max_obstacle_distance);
  p.two_squared_sigma = 2 * fp.sigma_hit * fp.sigma_hit;
  p.amplitude = fp.z_hit / (fp.sigma_hit * std::sqrt(2 * kPiD));
  p.offset = fp.z_random / fp.max_laser_distance;
  const double squared_background_distance = -p.two_squared_sigma * std::log((1 / fp.max_laser_distance - p.offset) / p.amplitude);
  p.overlay_sq = std::min(p.max_sq, static_cast<float>(squared_background_distance));
  hipLaunchKernelGGL(k_edt_columns, dim3(blocks_for(W)), dim3(kBlock), 0, st, d_cells, p, d_column_distance, d_column_offset);
  const size_t lds = static_cast<size_t>(kBlock + 2 * p.cap_cells) * sizeof(uint16_t);
  hipLaunchKernelGGL(k_edt_field, dim3(blocks_for(W), H), dim3(kBlock), lds, st, d_cells, p, d_column_distance, d_column_offset, d_field);
  return true;
}

void launch_sum_rows(hipStream_t st, const double* d_gathered, uint32_t rows, uint32_t columns, double* d_out, double* host_mirror) {
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(64), 0, st, d_gathered, rows, columns, d_out, host_mirror);
}

void launch_cube_table(hipStream_t st, const float* field, uint64_t cells, float unknown_value, double* cube, int prob) {
  hipLaunchKernelGGL(k_cube_table, dim3(blocks_for(cells + 1)), dim3(kBlock), 0, st, field, cells, unknown_value, cube, prob);
}
void launch_palette_table(hipStream_t st, const float* field, uint32_t W, uint32_t H, float unknown_value, const uint32_t* keys,
                          uint32_t count, int prob, uint16_t* idx, double* val, uint32_t pal_base) {
  const uint32_t tiles_x = (W + 7) / 8 + 2, tiles_y = (H + 7) / 8 + 2;
  hipLaunchKernelGGL(k_palette_values, dim3(blocks_for(count)), dim3(kBlock), 0, st, keys, count, prob, val);
  const uint64_t slots = static_cast<uint64_t>(tiles_x) * tiles_y * 64;
  hipLaunchKernelGGL(k_palette_indices, dim3(blocks_for(slots)), dim3(kBlock), 0, st, field, W, H, tiles_x, tiles_y, unknown_value,
                     keys, count, pal_base, idx);
}
void launch_far_tile_votes(hipStream_t st, const uint16_t* idx, uint32_t tiles, uint32_t pal_base, uint32_t count, uint32_t* votes) {
  (void)hipMemsetAsync(votes, 0, count * sizeof(uint32_t), st);
  hipLaunchKernelGGL(k_far_tile_votes, dim3(blocks_for(tiles)), dim3(kBlock), 0, st, idx, tiles, pal_base, count, votes);
}
void launch_far_tile_bits_linear(hipStream_t st, const uint16_t* idx, uint32_t tiles, uint32_t entry, uint32_t bytes, uint8_t* bits) {
  hipLaunchKernelGGL(k_far_tile_bits_linear, dim3(blocks_for(bytes)), dim3(kBlock), 0, st, idx, tiles, entry, bytes, bits);
}
void launch_far_tile_bits(hipStream_t st, const uint16_t* idx, uint32_t tiles_x, uint32_t tiles_y, uint32_t entry, uint32_t row_bytes,
                          uint32_t far_bytes, uint8_t* bits) {
  hipLaunchKernelGGL(k_far_tile_bits, dim3(blocks_for(far_bytes)), dim3(kBlock), 0, st, idx, tiles_x, tiles_y, entry, row_bytes, far_bytes,
                     bits);
}
void launch_fill(hipStream_t st, double* p, uint64_t n, double v) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(n)), dim3(kBlock), 0, st, p, n, v);
}

}  // namespace mcl
