// device_common.hpp - what every translation unit of the kernels shares: block sizes, pose load / store, cross-lane helpers and the
// deterministic reductions (fixed association: results do not depend on the launch).  Everything lives in an anonymous namespace of
// mcl: each translation unit gets its own copy, all of it inlined.
#pragma once
#include "kernels.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <climits>

#include "rng.h"

namespace mcl {
namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;
// The chunk-granular kernels of the ordering (one workgroup per kChunk = 2048 elements, LDS histogram / cursors of 1024 digits)
// run 1024 threads per workgroup: with 256 they put two waves on a SIMD and could not hide their own latencies.
constexpr int kWide = 1024;

__device__ __forceinline__ Pose2 load_pose(const Particles& p, uint64_t i) {
  const double4 v = p.pose[i];
  return Pose2{Rot2{v.x, v.y}, v.z, v.w};
}
__device__ __forceinline__ void store_pose(const Particles& p, uint64_t i, const Pose2& v) {
  p.pose[i] = double4{v.r.c, v.r.s, v.x, v.y};
}

// ---- cross-lane helpers ---------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane must be wave-uniform -> v_readlane_b32 x2 (SGPRs)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int kCtrl>
__device__ __forceinline__ double dpp_f64(double v) {  // row-local DPP permutation of a 64-bit value
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// Sum over the 64 lanes, fixed association: quads, octets, rows of 16 (DPP), then the four rows in order.
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
  return (r0 + r1) + (r2 + r3);
}

// The order in which a particle's terms are added over the scan.  The reference calls std::transform_reduce
// (likelihood_field_model.hpp:76, beam_model.hpp:108, likelihood_field_prob_model.hpp:77), whose association the standard leaves
// open; libstdc++ (<numeric>, random-access overload: the reference's toolchain on Linux) adds blocks of four as
// (f0 + f1) + (f2 + f3) to the running sum and the last B mod 4 terms one by one.  The kernels with a lane per particle do the
// same, so their weights carry the same roundings as that build's (and the chain of dependent additions is a third as long).
__device__ __forceinline__ double sum4(double a, double b, double c, double d) { return (a + b) + (c + d); }

// Deterministic block reduction of K doubles per thread.  Result valid in thread 0.
template <int K, int kThreads = kBlock>
__device__ __forceinline__ void block_reduce(double (&v)[K], double* s_scratch /* [kThreads/64][K] */) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum_f64(v[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) s_scratch[wave * K + k] = v[k];
  }
  __syncthreads();
  // The waves' partial sums of value k are added by thread k, waves in order (one thread adding all K columns held K x waves values in
  // registers at once: 60 bytes of scratch per lane in the draw kernel).  Thread k touches column k only.
  if (threadIdx.x < K) {
    double acc = s_scratch[threadIdx.x];
    for (int w = 1; w < kThreads / 64; ++w) acc += s_scratch[w * K + threadIdx.x];
    s_scratch[threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = s_scratch[k];
  }
  __syncthreads();
}

inline unsigned blocks_for(uint64_t n) { return static_cast<unsigned>((n + kBlock - 1) / kBlock); }

}  // namespace
}  // namespace mcl
