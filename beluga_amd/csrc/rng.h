// rng.h — the filter's random stream: counter-based Philox4x32-10.
//
// The reference draws from an unseeded thread-local mt19937 (actions/propagate.hpp:64-66,
// views/sample.hpp:58), which cannot be reproduced across 10^6 GPU threads.  Here every random
// number is a pure function of (seed, step, purpose, GLOBAL particle/candidate index), so the result
// does not depend on the thread mapping, the launch geometry or the number of GPUs.
//   counter = (index_lo, index_hi, step, purpose)    key = (seed_lo, seed_hi)
#pragma once
#include <cstdint>

#include "se2.h"

namespace mcl {

enum RngPurpose : uint32_t {
  kRngPropagateA = 0,  // words 0..3 -> (u1,u2) of the first Box-Muller pair
  kRngPropagateB = 1,  // second pair
  kRngResample = 2,    // words 0,1 -> multinomial uniform; word 2 -> Bernoulli(intersperse)
  kRngRandomState = 3, // words 0,1 -> free cell; words 2,3 -> heading
  kRngInitA = 4,
  kRngInitB = 5,
};

struct RngWords {
  uint32_t w[4];
};

MCL_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  const uint64_t p = static_cast<uint64_t>(a) * static_cast<uint64_t>(b);
  hi = static_cast<uint32_t>(p >> 32);
  lo = static_cast<uint32_t>(p);
}

MCL_HD RngWords philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    uint32_t hi0, lo0, hi1, lo1;
    philox_mulhilo(0xD2511F53u, c0, hi0, lo0);
    philox_mulhilo(0xCD9E8D57u, c2, hi1, lo1);
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return RngWords{{c0, c1, c2, c3}};
}

MCL_HD RngWords rng_draw(uint64_t seed, uint32_t step, uint32_t purpose, uint64_t index) {
  return philox4x32_10(static_cast<uint32_t>(index), static_cast<uint32_t>(index >> 32), step, purpose,
                       static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}

// 53-bit uniform in [0,1).
MCL_HD double rng_uniform53(uint32_t hi, uint32_t lo) {
  const uint64_t v = (static_cast<uint64_t>(hi) << 32) | lo;
  return static_cast<double>(v >> 11) * 0x1.0p-53;
}
// 32-bit uniform in [0,1).
MCL_HD double rng_uniform32(uint32_t w) { return static_cast<double>(w) * 0x1.0p-32; }

// Box-Muller; u1 reflected to (0,1].
MCL_HD void rng_box_muller(double u1, double u2, double& z0, double& z1) {
  const double r = sqrt(-2.0 * log(1.0 - u1));
  const double a = 2.0 * kPi * u2;
  z0 = r * cos(a);
  z1 = r * sin(a);
}

}  // namespace mcl
