// se2.h — SO(2)/SE(2) arithmetic shared by host and device code.
//
// The reference stores particle states as Sophus::SE2d (unit complex + translation); the filter's
// results depend on Sophus 1.22.10's exact operation order (products renormalise, constructors
// normalise with hypot).  These functions reproduce that order so that host-side policy code and
// the device kernels agree with the reference to rounding.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define MCL_HD __host__ __device__ __forceinline__
#else
#define MCL_HD inline
#endif

namespace mcl {

constexpr double kPi = 3.14159265358979323846264338327950288;

struct Rot2 {
  double c, s;
};
struct Pose2 {
  Rot2 r;
  double x, y;
};

// SO2(real, imag): store then normalize() (hypot).
MCL_HD Rot2 rot_from_complex(double re, double im) {
  const double len = hypot(re, im);
  return Rot2{re / len, im / len};
}
MCL_HD Rot2 rot_exp(double theta) { return rot_from_complex(cos(theta), sin(theta)); }
MCL_HD double rot_log(const Rot2& r) { return atan2(r.s, r.c); }
MCL_HD Rot2 rot_inverse(const Rot2& r) { return rot_from_complex(r.c, -r.s); }
// SO2 * SO2: complex product, first-order renormalisation if |z|^2 != 1, then the ctor's normalize().
MCL_HD Rot2 rot_mul(const Rot2& a, const Rot2& b) {
  double re = a.c * b.c - a.s * b.s;
  double im = a.c * b.s + a.s * b.c;
  const double n2 = re * re + im * im;
  if (n2 != 1.0) {
    const double scale = 2.0 / (1.0 + n2);
    re = re * scale;
    im = im * scale;
  }
  return rot_from_complex(re, im);
}
MCL_HD void rot_apply(const Rot2& r, double px, double py, double& ox, double& oy) {
  ox = r.c * px - r.s * py;
  oy = r.s * px + r.c * py;
}
MCL_HD Pose2 pose_mul(const Pose2& a, const Pose2& b) {
  Pose2 o;
  o.r = rot_mul(a.r, b.r);
  double tx, ty;
  rot_apply(a.r, b.x, b.y, tx, ty);
  o.x = a.x + tx;
  o.y = a.y + ty;
  return o;
}
MCL_HD Pose2 pose_inverse(const Pose2& a) {
  Pose2 o;
  o.r = rot_inverse(a.r);
  rot_apply(o.r, a.x * -1.0, a.y * -1.0, o.x, o.y);
  return o;
}
MCL_HD Pose2 pose_identity() { return Pose2{Rot2{1.0, 0.0}, 0.0, 0.0}; }

}  // namespace mcl
