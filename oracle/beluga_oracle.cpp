// beluga_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A dependency-free C++17 restatement of the Monte-Carlo-Localization update path of
// Ekumen-OS/beluga (`beluga::Amcl::update`, beluga/include/beluga/algorithm/amcl_core.hpp:165-201)
// and of the models / views / policies it composes.  Every function cites the reference
// file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
// leg of `bench.py` may load this library; the product (beluga_amd/) never does.
//
// Pinning: the reference itself cannot be built here (needs Eigen 3.4.0, Sophus 1.22.10,
// range-v3 0.12.0 — MODULE.bazel:21,25,27 — none installed, no network).  The restatement is
// pinned by the golden vectors of the reference's own unit tests (tests/test_oracle_golden.py
// lists them one by one with their source lines).  The END-TO-END estimate of `Amcl::update`
// has no numeric pin in the reference (test_amcl_core.cpp:73-186 are smoke tests) => for that
// level this oracle IS the pin ("parity unpinned" upstream).
//
// Third-party arithmetic restated from the published sources of the pinned versions:
//   Sophus 1.22.10 so2.hpp / se2.hpp : SO2(real,imag) ctor normalises with hypot; SO2*SO2 does the
//     complex product, a first-order renormalisation when |z|^2 != 1, then the ctor normalise;
//     SO2::log = atan2(imag, real); SE2*SE2 = (R1 R2, t1 + R1 t2); SE2::inverse = (R^-1, R^-1 (-t)).
//   libstdc++ <random> (GCC 11): discrete_distribution = normalise, partial_sum, last := 1.0,
//     draw = lower_bound(cp, u); bernoulli = (u < p); normal = z*stddev + mean.
//   range-v3 0.12.0: take_while / take semantics (the first failing element is dropped).
//
// Randomness: the reference draws from an UNSEEDED thread-local engine
// (actions/propagate.hpp:64-66, views/sample.hpp:58) so its stochastic stages are only
// reproducible in distribution.  The oracle (and the device path it checks) use a
// counter-based Philox4x32-10 stream addressed by (seed, step, purpose, global index); the
// decision structure around the draws is the reference's.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off so no FMA contraction changes a floor()).

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <queue>
#include <optional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

namespace {

constexpr double kPi = 3.14159265358979323846264338327950288;

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants).  Shared stream
// definition with the device path (DESIGN.md "RNG stream") but written independently here.
// ----------------------------------------------------------------------------------------------
struct Philox {
  static void round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = uint64_t{0xD2511F53u} * c[0];
    const uint64_t p1 = uint64_t{0xCD9E8D57u} * c[2];
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c[1] ^ k[0];
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c[3] ^ k[1];
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c[0] = n0;
    c[1] = n1;
    c[2] = n2;
    c[3] = n3;
  }
  static void run(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int i = 0; i < 10; ++i) {
      round(c, k);
      k[0] += 0x9E3779B9u;
      k[1] += 0xBB67AE85u;
    }
    std::memcpy(out, c, sizeof(c));
  }
};

// Purposes (4th counter word).  Same numbering as beluga_amd/csrc/mcl_rng.h.
enum : uint32_t {
  kPurposePropagateA = 0,
  kPurposePropagateB = 1,
  kPurposeResample = 2,
  kPurposeRandomState = 3,
  kPurposeInitA = 4,
  kPurposeInitB = 5,
};

struct Draw4 {
  uint32_t r[4];
};

inline Draw4 draw(uint64_t seed, uint32_t step, uint32_t purpose, uint64_t index) {
  const uint32_t ctr[4] = {static_cast<uint32_t>(index), static_cast<uint32_t>(index >> 32), step, purpose};
  const uint32_t key[2] = {static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32)};
  Draw4 d;
  Philox::run(ctr, key, d.r);
  return d;
}

// 53-bit uniform in [0,1) from two words.
inline double u53(uint32_t hi, uint32_t lo) {
  const uint64_t v = (uint64_t{hi} << 32) | lo;
  return static_cast<double>(v >> 11) * 0x1.0p-53;
}
// 32-bit uniform in [0,1).
inline double u32(uint32_t w) { return static_cast<double>(w) * 0x1.0p-32; }

// Box–Muller on (u1,u2) in [0,1)^2; u1 is reflected to (0,1] so the log is finite.
inline void box_muller(double u1, double u2, double* z0, double* z1) {
  const double r = std::sqrt(-2.0 * std::log(1.0 - u1));
  const double a = 2.0 * kPi * u2;
  *z0 = r * std::cos(a);
  *z1 = r * std::sin(a);
}

// ----------------------------------------------------------------------------------------------
// Sophus 1.22.10 restatement (SO2/SE2, double).  Memory layout (c, s, x, y) = SE2d::data().
// ----------------------------------------------------------------------------------------------
struct SO2 {
  double c{1.0}, s{0.0};
};
struct SE2 {
  SO2 r;
  double x{0.0}, y{0.0};
};

inline SO2 so2_normalized(double re, double im) {  // SO2(real, imag) ctor -> normalize()
  const double len = std::hypot(re, im);
  return SO2{re / len, im / len};
}
inline SO2 so2_exp(double theta) { return so2_normalized(std::cos(theta), std::sin(theta)); }
inline double so2_log(const SO2& r) { return std::atan2(r.s, r.c); }
inline SO2 so2_inverse(const SO2& r) { return so2_normalized(r.c, -r.s); }
inline SO2 so2_mul(const SO2& a, const SO2& b) {
  const double re = a.c * b.c - a.s * b.s;
  const double im = a.c * b.s + a.s * b.c;
  const double n2 = re * re + im * im;
  if (n2 != 1.0) {
    const double scale = 2.0 / (1.0 + n2);
    return so2_normalized(re * scale, im * scale);
  }
  return so2_normalized(re, im);
}
inline void so2_act(const SO2& r, double px, double py, double* ox, double* oy) {
  *ox = r.c * px - r.s * py;
  *oy = r.s * px + r.c * py;
}
inline SE2 se2_mul(const SE2& a, const SE2& b) {
  SE2 o;
  o.r = so2_mul(a.r, b.r);
  double tx, ty;
  so2_act(a.r, b.x, b.y, &tx, &ty);
  o.x = a.x + tx;
  o.y = a.y + ty;
  return o;
}
inline SE2 se2_inverse(const SE2& a) {
  SE2 o;
  o.r = so2_inverse(a.r);
  so2_act(o.r, a.x * -1.0, a.y * -1.0, &o.x, &o.y);
  return o;
}
inline SE2 se2_load(const double* p) { return SE2{SO2{p[0], p[1]}, p[2], p[3]}; }
inline void se2_store(const SE2& a, double* p) {
  p[0] = a.r.c;
  p[1] = a.r.s;
  p[2] = a.x;
  p[3] = a.y;
}

// ----------------------------------------------------------------------------------------------
// Grids.  sensor/data/regular_grid.hpp:75-89, dense_grid.hpp:92-129, linear_grid.hpp:73-130,
// occupancy_grid.hpp:101-213.  Cells are int8 with (free, unknown, occupied) value traits, like
// beluga_ros::OccupancyGrid::ValueTraits (0 / -1 / 100) and the test fixture
// beluga/test/beluga/include/beluga/test/static_occupancy_grid.hpp:38-48.
// ----------------------------------------------------------------------------------------------
struct Traits {
  int8_t free_value, unknown_value, occupied_value;
  // beluga_ros trait semantics (occupancy_grid.hpp:48-64 in beluga_ros): exact-equality classes.
  bool is_free(int8_t v) const { return v == free_value; }
  bool is_unknown(int8_t v) const { return v == unknown_value; }
  bool is_occupied(int8_t v) const { return v == occupied_value; }
};

struct Grid {
  const int8_t* cells;
  int W, H;
  double res;
  SE2 origin;
  Traits traits;

  size_t size() const { return static_cast<size_t>(W) * static_cast<size_t>(H); }
  // regular_grid.hpp:75-78  (multiply by 1/res, floor, cast<int>)
  void cell_near(double px, double py, int* xi, int* yi) const {
    const double inv = 1. / res;
    *xi = static_cast<int>(std::floor(px * inv));
    *yi = static_cast<int>(std::floor(py * inv));
  }
  // regular_grid.hpp:87-89
  void coordinates_at(int xi, int yi, double* x, double* y) const {
    *x = (static_cast<double>(xi) + 0.5) * res;
    *y = (static_cast<double>(yi) + 0.5) * res;
  }
  void coordinates_at(size_t index, double* x, double* y) const {  // linear_grid.hpp:92-95
    coordinates_at(static_cast<int>(index % W), static_cast<int>(index / W), x, y);
  }
  bool contains(int xi, int yi) const { return xi >= 0 && yi >= 0 && xi < W && yi < H; }  // dense_grid.hpp:92-96
  size_t index_at(int xi, int yi) const { return static_cast<size_t>(yi) * W + static_cast<size_t>(xi); }
  // occupancy_grid.hpp:101-117 ; out-of-range index is non-free (linear_grid.hpp:102-104)
  bool free_at(int xi, int yi) const {
    const size_t idx = index_at(xi, yi);
    if (!(idx < size())) return false;
    return traits.is_free(cells[idx]);
  }
  // linear_grid.hpp:113-130 — order: +x, +y, -x, -y
  template <class F>
  void neighborhood4(size_t index, F&& f) const {
    const size_t xi = index % W;
    const size_t yi = index / W;
    if (xi < static_cast<size_t>(W - 1)) f(index + 1);
    if (yi < static_cast<size_t>(H - 1)) f(index + W);
    if (xi > 0) f(index - 1);
    if (yi > 0) f(index - W);
  }
  bool obstacle_edge(size_t index) const {  // occupancy_grid.hpp:191-206
    if (!traits.is_occupied(cells[index])) return false;
    bool any_free = false;
    neighborhood4(index, [&](size_t n) { any_free = any_free || traits.is_free(cells[n]); });
    return any_free;
  }
};

// algorithm/distance_map.hpp:55-98.  Same container (std::priority_queue over a vector with the
// same comparator and the same push order) so heap tie-breaking matches libstdc++'s.
template <class Mask, class DistFn, class NeighFn>
std::vector<float> nearest_obstacle_distance_map(size_t n, Mask&& mask, DistFn&& dist, NeighFn&& neigh, float max_value) {
  struct IndexPair {
    size_t nearest_obstacle_index;
    size_t index;
  };
  std::vector<float> distance_map(n, max_value);
  std::vector<bool> visited(n, false);
  auto compare = [&distance_map](const IndexPair& a, const IndexPair& b) {
    return distance_map[a.index] > distance_map[b.index];
  };
  std::priority_queue<IndexPair, std::vector<IndexPair>, decltype(compare)> queue{compare};
  for (size_t i = 0; i < n; ++i) {
    if (mask(i)) {
      visited[i] = true;
      distance_map[i] = 0;
      queue.push(IndexPair{i, i});
    }
  }
  while (!queue.empty()) {
    const auto parent = queue.top();
    queue.pop();
    neigh(parent.index, [&](size_t index) {
      if (!visited[index]) {
        visited[index] = true;
        const float d = dist(parent.nearest_obstacle_index, index);
        if (d < max_value) {
          distance_map[index] = d;
          queue.push(IndexPair{parent.nearest_obstacle_index, index});
        }
      }
    });
  }
  return distance_map;
}

struct LfParams {  // sensor/likelihood_field_model_base.hpp:42-64
  double max_obstacle_distance, max_laser_distance, z_hit, z_random, sigma_hit;
  int model_unknown_space, only_obstacle_boundaries;
};

// sensor/likelihood_field_model_base.hpp:130-185
void make_likelihood_field(const Grid& g, const LfParams& p, float* out) {
  const auto squared_distance = [&g](size_t a, size_t b) {
    double ax, ay, bx, by;
    g.coordinates_at(a, &ax, &ay);
    g.coordinates_at(b, &bx, &by);
    const double dx = ax - bx, dy = ay - by;
    return static_cast<float>(dx * dx + dy * dy);
  };
  const double two_squared_sigma = 2 * p.sigma_hit * p.sigma_hit;
  const double amplitude = p.z_hit / (p.sigma_hit * std::sqrt(2 * kPi));
  const double offset = p.z_random / p.max_laser_distance;
  const auto to_likelihood = [=](double sq) { return amplitude * std::exp(-sq / two_squared_sigma) + offset; };
  const auto neighborhood = [&g](size_t i, auto&& f) { g.neighborhood4(i, f); };
  const float squared_max_distance = static_cast<float>(p.max_obstacle_distance * p.max_obstacle_distance);

  std::vector<float> dm =
      p.only_obstacle_boundaries
          ? nearest_obstacle_distance_map(
                g.size(), [&g](size_t i) { return g.obstacle_edge(i); }, squared_distance, neighborhood,
                squared_max_distance)
          : nearest_obstacle_distance_map(
                g.size(), [&g](size_t i) { return g.traits.is_occupied(g.cells[i]); }, squared_distance, neighborhood,
                squared_max_distance);

  if (p.model_unknown_space) {
    const double inverse_max_distance = 1 / p.max_laser_distance;
    const double squared_background_distance = -two_squared_sigma * std::log((inverse_max_distance - offset) / amplitude);
    const float overlay_value = std::min(squared_max_distance, static_cast<float>(squared_background_distance));
    for (size_t i = 0; i < g.size(); ++i) {
      const bool is_obstacle = g.traits.is_occupied(g.cells[i]);
      const bool is_unknown = g.traits.is_unknown(g.cells[i]);
      const bool effective = p.only_obstacle_boundaries ? (is_unknown || (is_obstacle && !g.obstacle_edge(i))) : is_unknown;
      if (effective) dm[i] = overlay_value;  // actions/overlay.hpp:47-60
    }
  }
  for (size_t i = 0; i < g.size(); ++i) {
    out[i] = static_cast<float>(to_likelihood(static_cast<double>(dm[i])));  // ranges::actions::transform in place on float
  }
}

struct ScanPoint {  // std::pair<double, double> of the reference's measurement_type, over the caller's packed (x, y) doubles
  double first, second;
};
static_assert(sizeof(ScanPoint) == 2 * sizeof(double), "ScanPoint must be two packed doubles");

// sensor/likelihood_field_model.hpp:68-91 — ONE particle.
inline double lf_weight(
    const float* field, int W, int H, double res, const SE2& world_to_field, double max_laser_distance, const SE2& state,
    const double* pts, size_t B) {
  const SE2 transform = se2_mul(world_to_field, state);
  const double x_offset = transform.x, y_offset = transform.y;
  const double cos_theta = transform.r.c, sin_theta = transform.r.s;
  const float unknown_space_occupancy_prob = static_cast<float>(1. / max_laser_distance);
  const double inv_resolution = 1. / res;  // regular_grid.hpp:76
  // The reference calls std::transform_reduce over the points (:76), whose order of additions the standard leaves open; called
  // here as well, over the same kind of range (random access), it associates as the reference's own standard library does -
  // libstdc++: blocks of four, (f0 + f1) + (f2 + f3), added to the running sum, the rest one by one.
  const ScanPoint* points = reinterpret_cast<const ScanPoint*>(pts);
  return std::transform_reduce(points, points + B, 1.0, std::plus<>{}, [&](const ScanPoint& point) {
    const double x = point.first * cos_theta - point.second * sin_theta + x_offset;
    const double y = point.first * sin_theta + point.second * cos_theta + y_offset;
    const int xi = static_cast<int>(std::floor(x * inv_resolution));
    const int yi = static_cast<int>(std::floor(y * inv_resolution));
    float v = unknown_space_occupancy_prob;
    if (xi >= 0 && yi >= 0 && xi < W && yi < H) v = field[static_cast<size_t>(yi) * W + static_cast<size_t>(xi)];
    const double pz = static_cast<double>(v);
    return pz * pz * pz;
  });
}

// sensor/likelihood_field_prob_model.hpp:68-90 — ONE particle: exp(sum log pz).
inline double lf_prob_weight(
    const float* field, int W, int H, double res, const SE2& world_to_field, double max_laser_distance, const SE2& state,
    const double* pts, size_t B) {
  const SE2 transform = se2_mul(world_to_field, state);
  const double x_offset = transform.x, y_offset = transform.y;
  const double cos_theta = transform.r.c, sin_theta = transform.r.s;
  const float unknown_space_occupancy_prob = static_cast<float>(1. / max_laser_distance);
  const double inv_resolution = 1. / res;
  const ScanPoint* points = reinterpret_cast<const ScanPoint*>(pts);  // std::transform_reduce as in lf_weight (:77)
  return std::exp(std::transform_reduce(points, points + B, 0.0, std::plus<>{}, [&](const ScanPoint& point) {
    const double x = point.first * cos_theta - point.second * sin_theta + x_offset;
    const double y = point.first * sin_theta + point.second * cos_theta + y_offset;
    const int xi = static_cast<int>(std::floor(x * inv_resolution));
    const int yi = static_cast<int>(std::floor(y * inv_resolution));
    float v = unknown_space_occupancy_prob;
    if (xi >= 0 && yi >= 0 && xi < W && yi < H) v = field[static_cast<size_t>(yi) * W + static_cast<size_t>(xi)];
    return std::log(static_cast<double>(v));
  }));
}

// algorithm/raycasting/bresenham.hpp:84-192, iterator restated as a small state machine.
struct Bresenham {
  int cx, cy;  // current_point_
  int x_, y_, xspan_, yspan_, dxspan_, dyspan_, xstep_, ystep_, step_{0};
  int prev_error_, error_;
  size_t checks_{0};
  bool modified_, reversed_{false};

  Bresenham(int x0, int y0, int x1, int y1, bool modified) : cx(x0), cy(y0), x_(x0), y_(y0), modified_(modified) {
    xspan_ = x1 - x0;
    xstep_ = 1;
    if (xspan_ < 0) {
      xspan_ = -xspan_;
      xstep_ = -xstep_;
    }
    yspan_ = y1 - y0;
    ystep_ = 1;
    if (yspan_ < 0) {
      yspan_ = -yspan_;
      ystep_ = -ystep_;
    }
    if (xspan_ < yspan_) {
      std::swap(x_, y_);
      std::swap(xspan_, yspan_);
      std::swap(xstep_, ystep_);
      reversed_ = true;
    }
    dxspan_ = 2 * xspan_;
    dyspan_ = 2 * yspan_;
    error_ = prev_error_ = xspan_;
  }
  bool done() const { return step_ > xspan_; }
  void step_to(int x, int y) {
    if (reversed_) std::swap(x, y);
    cx = x;
    cy = y;
  }
  void next() {
    if (checks_ == 0) {
      if (++step_ > xspan_) return;
      x_ += xstep_;
      error_ += dyspan_;
      ++checks_;
      if (error_ > dxspan_) {
        y_ += ystep_;
        error_ -= dxspan_;
        if (modified_) {
          ++checks_;
          ++checks_;
        }
      }
    }
    if (checks_ > 1) {
      if (checks_ > 2) {
        --checks_;
        if (error_ + prev_error_ <= dxspan_) {
          step_to(x_, y_ - ystep_);
          return;
        }
      }
      --checks_;
      if (error_ + prev_error_ >= dxspan_) {
        step_to(x_ - xstep_, y_);
        return;
      }
    }
    --checks_;
    step_to(x_, y_);
    prev_error_ = error_;
  }
};

// algorithm/raycasting.hpp:62-107.  Returns true and *out if a non-free cell is hit.
struct Ray2d {
  const Grid& g;
  SE2 source_local;
  int sx, sy;
  double max_range;
  Ray2d(const Grid& grid, const SE2& source_pose, double mr)
      : g(grid), source_local(se2_mul(se2_inverse(grid.origin), source_pose)), max_range(mr) {
    g.cell_near(source_local.x, source_local.y, &sx, &sy);
  }
  bool cast(const SO2& bearing, double* out, long* steps = nullptr) const {
    const double t2x = bearing.c * max_range, t2y = bearing.s * max_range;
    double ex, ey;
    so2_act(source_local.r, t2x, t2y, &ex, &ey);
    ex += source_local.x;
    ey += source_local.y;
    int fx, fy;
    g.cell_near(ex, ey, &fx, &fy);
    Bresenham it(sx, sy, fx, fy, false);
    for (; !it.done(); it.next()) {
      if (!g.contains(it.cx, it.cy)) break;  // take_while(cell_is_valid)
      if (steps) ++*steps;
      if (!g.free_at(it.cx, it.cy)) {
        double ax, ay, bx, by;
        g.coordinates_at(sx, sy, &ax, &ay);
        g.coordinates_at(it.cx, it.cy, &bx, &by);
        const double dx = bx - ax, dy = by - ay;
        *out = std::min(std::sqrt(dx * dx + dy * dy), max_range);
        return true;
      }
    }
    return false;
  }
};

struct BeamParams {  // sensor/beam_model.hpp:43-58
  double z_hit, z_short, z_max, z_rand, sigma_hit, lambda_short, beam_max_range;
};

// sensor/beam_model.hpp:104-150 — ONE particle.
inline double beam_weight(const Grid& g, const BeamParams& p, const SE2& state, const double* pts, size_t B, long* steps) {
  const Ray2d beam{g, state, p.beam_max_range};
  const double n = 1. / (std::sqrt(2. * M_PI) * p.sigma_hit);
  const ScanPoint* points = reinterpret_cast<const ScanPoint*>(pts);  // std::transform_reduce as in lf_weight (:108)
  return std::transform_reduce(points, points + B, 0.0, std::plus<>{}, [&](const ScanPoint& point) {
    const double px = point.first, py = point.second;
    const double z = std::sqrt(px * px + py * py);
    SO2 bearing;
    bearing.c = px / z;
    bearing.s = py / z;
    double z_mean = p.beam_max_range;
    double hit;
    if (beam.cast(bearing, &hit, steps)) z_mean = hit;
    const double eta_hit = 2. / (std::erf((p.beam_max_range - z_mean) / (std::sqrt(2.) * p.sigma_hit)) -
                                 std::erf(-z_mean / (std::sqrt(2.) * p.sigma_hit)));
    const double d = (z - z_mean) / p.sigma_hit;
    double pz = p.z_hit * eta_hit * n * std::exp(-(d * d) / 2.);
    if (z < z_mean) {
      const double eta_short = 1. / (1. - std::exp(-p.lambda_short * z_mean));
      pz += p.z_short * p.lambda_short * eta_short * std::exp(-p.lambda_short * z);
    }
    if (z < p.beam_max_range) {
      pz += p.z_rand / p.beam_max_range;
    } else {
      pz += p.z_max;
    }
    return pz * pz * pz;
  });
}

// motion/differential_drive_model.hpp:129-173
double rotation_variance(const SO2& r) {
  const SO2 flipping = so2_exp(kPi);
  const SO2 flipped = so2_mul(r, flipping);
  const double delta = std::min(std::abs(so2_log(r)), std::abs(so2_log(flipped)));
  return delta * delta;
}
struct DiffDriveSampler {  // three (mean, stddev) pairs
  double m1, s1, mt, st, m2, s2;
};
DiffDriveSampler diffdrive_sampler(const SE2& pose, const SE2& prev, const double a[4], double distance_threshold) {
  const double tx = pose.x - prev.x, ty = pose.y - prev.y;
  const double distance = std::sqrt(tx * tx + ty * ty);
  const double distance_variance = distance * distance;
  const SO2 heading = so2_exp(std::atan2(ty, tx));
  const SO2 first = distance > distance_threshold ? so2_mul(heading, so2_inverse(prev.r)) : SO2{};
  const SO2 second = so2_mul(so2_mul(pose.r, so2_inverse(prev.r)), so2_inverse(first));
  DiffDriveSampler s;
  s.m1 = so2_log(first);
  s.s1 = std::sqrt(a[0] * rotation_variance(first) + a[1] * distance_variance);
  s.mt = distance;
  s.st = std::sqrt(a[2] * distance_variance + a[3] * (rotation_variance(first) + rotation_variance(second)));
  s.m2 = so2_log(second);
  s.s2 = std::sqrt(a[0] * rotation_variance(second) + a[1] * distance_variance);
  return s;
}
// differential_drive_model.hpp:156-163 with the Philox/Box–Muller stream.
inline SE2 diffdrive_apply(const SE2& state, const DiffDriveSampler& s, uint64_t seed, uint32_t step, uint64_t index) {
  const Draw4 a = draw(seed, step, kPurposePropagateA, index);
  const Draw4 b = draw(seed, step, kPurposePropagateB, index);
  double z0, z1, z2, z3;
  box_muller(u53(a.r[0], a.r[1]), u53(a.r[2], a.r[3]), &z0, &z1);
  box_muller(u53(b.r[0], b.r[1]), u53(b.r[2], b.r[3]), &z2, &z3);
  const double r1 = z0 * s.s1 + s.m1;  // libstdc++ normal_distribution: ret * stddev + mean
  const double t = z1 * s.st + s.mt;
  const double r2 = z2 * s.s2 + s.m2;
  const SE2 first{so2_exp(r1), 0.0, 0.0};
  const SE2 second{so2_exp(r2), t, 0.0};
  return se2_mul(se2_mul(state, first), second);
}

// motion/omnidirectional_drive_model.hpp:102-146 (+ rotation_variance :150-155)
struct OmniSampler {
  double m_rot, s_rot, m_trans, s_trans, s_strafe;
  SO2 first;
};
OmniSampler omni_sampler(const SE2& pose, const SE2& prev, const double a[5], double distance_threshold) {
  const double tx = pose.x - prev.x, ty = pose.y - prev.y;
  const double distance = std::sqrt(tx * tx + ty * ty);
  const double distance_variance = distance * distance;
  const SO2 rotation = so2_mul(pose.r, so2_inverse(prev.r));
  const SO2 heading = so2_exp(std::atan2(ty, tx));
  OmniSampler s;
  s.first = distance > distance_threshold ? so2_mul(heading, so2_inverse(prev.r)) : SO2{};
  s.m_rot = so2_log(rotation);
  s.s_rot = std::sqrt(a[0] * rotation_variance(rotation) + a[1] * distance_variance);
  s.m_trans = distance;
  s.s_trans = std::sqrt(a[2] * distance_variance + a[3] * rotation_variance(rotation));
  s.s_strafe = std::sqrt(a[4] * distance_variance + a[3] * rotation_variance(rotation));
  return s;
}
inline void three_normals(uint64_t seed, uint32_t step, uint64_t index, double z[3]) {
  const Draw4 a = draw(seed, step, kPurposePropagateA, index);
  const Draw4 b = draw(seed, step, kPurposePropagateB, index);
  double z3;
  box_muller(u53(a.r[0], a.r[1]), u53(a.r[2], a.r[3]), &z[0], &z[1]);
  box_muller(u53(b.r[0], b.r[1]), u53(b.r[2], b.r[3]), &z[2], &z3);
}
// omnidirectional_drive_model.hpp:133-144 — draws in source order: rotation, translation, strafe.
inline SE2 omni_apply(const SE2& state, const OmniSampler& s, uint64_t seed, uint32_t step, uint64_t index) {
  double z[3];
  three_normals(seed, step, index, z);
  const SO2 second = so2_mul(so2_exp(z[0] * s.s_rot + s.m_rot), so2_inverse(s.first));
  const double t = z[1] * s.s_trans + s.m_trans;
  const double strafe = z[2] * s.s_strafe + 0.0;
  const SE2 a{s.first, 0.0, 0.0};
  const SE2 b{second, t, -strafe};
  return se2_mul(se2_mul(state, a), b);
}
// motion/stationary_model.hpp:55-61 — N(0, 0.02) on rotation, x, y (in that order).
inline SE2 stationary_apply(const SE2& state, uint64_t seed, uint32_t step, uint64_t index) {
  double z[3];
  three_normals(seed, step, index, z);
  const SE2 d{so2_exp(z[0] * 0.02 + 0.0), z[1] * 0.02 + 0.0, z[2] * 0.02 + 0.0};
  return se2_mul(state, d);
}

// algorithm/spatial_hash.hpp:45-75,87-94,190-193
inline uint64_t floor_and_fibo_hash(double value, unsigned shift) {
  const int64_t sv = static_cast<int64_t>(std::floor(value));
  const uint64_t uv = static_cast<uint64_t>(sv);
  const uint64_t h = 11400714819323198485ull * uv;
  if (shift != 0) return (h << shift) | (h >> (64 - shift));
  return h;
}
inline uint64_t spatial_hash(const SE2& s, const double res[3]) {
  constexpr unsigned kBits = 64 / 3;
  return floor_and_fibo_hash(s.x / res[0], 0) ^ floor_and_fibo_hash(s.y / res[1], kBits) ^
         floor_and_fibo_hash(so2_log(s.r) / res[2], 2 * kBits);
}

// views/take_while_kld.hpp:73-81
inline size_t kld_target_size(size_t k, double epsilon, double z) {
  const double two_epsilon = 2 * epsilon;
  if (k <= 2U) return std::numeric_limits<size_t>::max();
  const double common = 2. / static_cast<double>(9 * (k - 1));
  const double base = 1. - common + std::sqrt(common) * z;
  const double result = (static_cast<double>(k - 1) / two_epsilon) * base * base * base;
  return static_cast<size_t>(std::ceil(result));
}

// actions/normalize.hpp:54-85 (sequential accumulate, skip if |sum-1|<eps)
inline double normalize(double* w, size_t n, int threads) {
  double sum = 0.0;
  for (size_t i = 0; i < n; ++i) sum += w[i];
  if (std::abs(sum - 1.0) < std::numeric_limits<double>::epsilon()) return sum;
  (void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
  for (long i = 0; i < static_cast<long>(n); ++i) w[i] = w[i] / sum;
  return sum;
}

// algorithm/effective_sample_size.hpp:46-59
inline double effective_sample_size(const double* w, size_t n) {
  double total = 0.0;
  for (size_t i = 0; i < n; ++i) total += w[i];
  if (total == 0.0) return 0.0;
  double acc = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double nw = w[i] / total;
    acc += nw * nw;
  }
  return 1.0 / acc;
}

// algorithm/exponential_filter.hpp:32-44 + thrun_recovery_probability_estimator.hpp:47-89
struct ExponentialFilter {
  double output{0.}, alpha{0.};
  void reset() { output = 0.; }
  double operator()(double input) {
    output += (output == 0.) ? input : alpha * (input - output);
    return output;
  }
};
struct Thrun {
  ExponentialFilter slow, fast;
  void reset() {
    slow.reset();
    fast.reset();
  }
  double operator()(const double* w, size_t n) {
    if (n == 0) {
      reset();
      return 0.0;
    }
    double total = 0.0;
    for (size_t i = 0; i < n; ++i) total += w[i];
    const double average = total / static_cast<double>(n);
    const double fast_average = fast(average);
    const double slow_average = slow(average);
    if (std::abs(slow_average) < std::numeric_limits<double>::epsilon()) return 0.0;
    return std::clamp(1.0 - fast_average / slow_average, 0.0, 1.0);
  }
};

// algorithm/estimation.hpp:436-475 (+ mean_fn :49-73, covariance_fn :237-272)
void estimate(const double* states, const double* w, size_t n, double mean_out[4], double cov_out[9]) {
  double sum = 0.0;
  for (size_t i = 0; i < n; ++i) sum += w[i];
  double m[4] = {0, 0, 0, 0};
  for (size_t i = 0; i < n; ++i) {
    const double nw = w[i] / sum;
    for (int k = 0; k < 4; ++k) m[k] += nw * states[4 * i + k];
  }
  double acc[4] = {0, 0, 0, 0};
  double sq = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double nw = w[i] / sum;
    const double dx = states[4 * i + 2] - m[2];
    const double dy = states[4 * i + 3] - m[3];
    acc[0] += nw * dx * dx;
    acc[1] += nw * dx * dy;
    acc[2] += nw * dy * dx;
    acc[3] += nw * dy * dy;
    sq += nw * nw;
  }
  for (int k = 0; k < 9; ++k) cov_out[k] = 0.0;
  const double corr = 1.0 - sq;
  cov_out[0] = acc[0] / corr;
  cov_out[1] = acc[1] / corr;
  cov_out[3] = acc[2] / corr;
  cov_out[4] = acc[3] / corr;
  const double norm = std::sqrt(m[0] * m[0] + m[1] * m[1]);  // Eigen norm()
  if (norm < std::numeric_limits<double>::epsilon()) {
    cov_out[8] = std::numeric_limits<double>::infinity();
    const SO2 zero = so2_exp(0.0);
    m[0] = zero.c;
    m[1] = zero.s;
  } else {
    cov_out[8] = -2.0 * std::log(norm);
    const SO2 nrm = so2_normalized(m[0], m[1]);  // so2().normalize()
    m[0] = nrm.c;
    m[1] = nrm.s;
  }
  for (int k = 0; k < 4; ++k) mean_out[k] = m[k];
}

// ----------------------------------------------------------------------------------------------
// algorithm/cluster_based_estimation.hpp — the estimate beluga_ros::Amcl returns (beluga_ros/src/amcl.cpp:125).
// Same standard containers as the reference (std::unordered_map with reserve(n/5), std::priority_queue built from
// the map's iteration order, std::nth_element) so that ties resolve the way libstdc++ resolves them upstream.
// ----------------------------------------------------------------------------------------------
struct ClusterCell {  // :113-119
  SE2 representative_state;
  double weight{0.0};
  size_t num_particles{0};
  std::optional<size_t> cluster_id;
};
using ClusterMap = std::unordered_map<size_t, ClusterCell>;

struct ClusterParams {  // ParticleClusterizerParam :243-259
  double linear_hash_resolution = 0.20, angular_hash_resolution = 0.524, weight_cap_percentile = 0.90;
};

std::vector<size_t> cluster_ids(const double* states, const double* w, size_t n, const ClusterParams& p) {
  const double res[3] = {p.linear_hash_resolution, p.linear_hash_resolution, p.angular_hash_resolution};
  std::vector<size_t> hashes(n);
  for (size_t i = 0; i < n; ++i) hashes[i] = spatial_hash(se2_load(states + 4 * i), res);  // :291
  // make_cluster_map :137-157
  ClusterMap map;
  map.reserve(n / 5);
  for (size_t i = 0; i < n; ++i) {
    auto [it, inserted] = map.try_emplace(hashes[i], ClusterCell{});
    ClusterCell& entry = it->second;
    entry.weight += w[i];
    entry.num_particles++;
    if (inserted) entry.representative_state = se2_load(states + 4 * i);
  }
  // normalize_and_cap_weights :173-189 ; calculate_percentile_threshold :103-109
  for (auto& kv : map) kv.second.weight /= static_cast<double>(kv.second.num_particles);
  {
    std::vector<double> values;
    values.reserve(map.size());
    for (auto& kv : map) values.push_back(kv.second.weight);
    const auto nth = static_cast<std::ptrdiff_t>(static_cast<double>(values.size()) * p.weight_cap_percentile);
    std::nth_element(values.begin(), values.begin() + nth, values.end());
    const double max_weight = values[static_cast<size_t>(nth)];
    for (auto& kv : map) kv.second.weight = std::min(kv.second.weight, max_weight);
  }
  // assign_clusters :203-238 ; make_priority_queue :73-92
  struct KeyWithPriority {
    double priority;
    size_t key;
    bool operator<(const KeyWithPriority& other) const { return priority < other.priority; }
  };
  std::vector<KeyWithPriority> init;
  init.reserve(map.size());
  for (auto& kv : map) init.push_back(KeyWithPriority{kv.second.weight, kv.first});
  std::priority_queue<KeyWithPriority> queue(init.begin(), init.end());
  const double max_priority = queue.top().priority;
  const SE2 adjacent[6] = {  // :323-331
      SE2{so2_exp(0.0), +p.linear_hash_resolution, 0.0}, SE2{so2_exp(0.0), -p.linear_hash_resolution, 0.0},
      SE2{so2_exp(0.0), 0.0, +p.linear_hash_resolution}, SE2{so2_exp(0.0), 0.0, -p.linear_hash_resolution},
      SE2{so2_exp(+p.angular_hash_resolution), 0.0, 0.0}, SE2{so2_exp(-p.angular_hash_resolution), 0.0, 0.0}};
  size_t next_cluster_id = 0;
  while (!queue.empty()) {
    const size_t hash = queue.top().key;
    queue.pop();
    ClusterCell& cell = map[hash];
    if (!cell.cluster_id.has_value()) cell.cluster_id = next_cluster_id++;
    for (const SE2& adj : adjacent) {
      const size_t neighbor_hash = spatial_hash(se2_mul(cell.representative_state, adj), res);  // neighbors() :271-275
      auto it = map.find(neighbor_hash);
      const bool valid = it != map.end() && !it->second.cluster_id.has_value() && it->second.weight <= cell.weight;
      if (!valid) continue;
      ClusterCell& neighbor = map[neighbor_hash];
      neighbor.cluster_id = cell.cluster_id;
      queue.push(KeyWithPriority{max_priority + neighbor.weight, neighbor_hash});
    }
  }
  std::vector<size_t> out(n);
  for (size_t i = 0; i < n; ++i) out[i] = map[hashes[i]].cluster_id.value();
  return out;
}

// estimate_clusters :345-411 + cluster_based_estimate :415-433
void cluster_based_estimate(const double* states, const double* w, size_t n, const ClusterParams& p, double mean_out[4], double cov_out[9]) {
  const std::vector<size_t> clusters = cluster_ids(states, w, n, p);
  std::vector<size_t> order(n);
  std::iota(order.begin(), order.end(), size_t{0});
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return clusters[a] < clusters[b]; });
  bool have_best = false;
  double best_weight = 0.0;
  std::vector<double> cs, cw;
  for (size_t begin = 0; begin < n;) {
    size_t end = begin;
    while (end < n && clusters[order[end]] == clusters[order[begin]]) ++end;
    if (end - begin > 1) {  // a single sample has no covariance
      cs.clear();
      cw.clear();
      double total = 0.0;
      for (size_t k = begin; k < end; ++k) {
        const size_t i = order[k];
        cs.insert(cs.end(), states + 4 * i, states + 4 * i + 4);
        cw.push_back(w[i]);
        total += w[i];
      }
      if (!have_best || best_weight < total) {  // ranges::max_element keeps the first maximum
        double m[4], c[9];
        estimate(cs.data(), cw.data(), cw.size(), m, c);
        std::memcpy(mean_out, m, sizeof(m));
        std::memcpy(cov_out, c, sizeof(c));
        best_weight = total;
        have_best = true;
      }
    }
    begin = end;
  }
  if (!have_best) estimate(states, w, n, mean_out, cov_out);
}

struct ResampleParams {
  uint64_t min_particles, max_particles;
  double kld_epsilon, kld_z;
  double hash_res[3];
  double random_state_probability;
  uint64_t seed;
  uint32_t step;
};

// The random state generator of beluga_ros::Amcl (beluga_ros/src/amcl.cpp:107):
// MultivariateUniformDistribution over free cells, random/multivariate_uniform_distribution.hpp:126-161
// — a uniformly chosen free cell centre (global frame) and a uniform angle in [-pi, pi).
inline SE2 random_state(const double* free_xy, uint64_t n_free, uint64_t seed, uint32_t step, uint64_t index) {
  const Draw4 d = draw(seed, step, kPurposeRandomState, index);
  uint64_t cell = static_cast<uint64_t>(u53(d.r[0], d.r[1]) * static_cast<double>(n_free));
  if (cell >= n_free) cell = n_free - 1;
  const double theta = -kPi + 2.0 * kPi * u53(d.r[2], d.r[3]);
  SE2 s;
  s.r = so2_exp(theta);
  s.x = free_xy[2 * cell];
  s.y = free_xy[2 * cell + 1];
  return s;
}

// amcl_core.hpp:188-196:  views::sample | random_intersperse | take_while_kld | assign
//   views/sample.hpp:128-136,74-102 ; views/random_intersperse.hpp:90-115 ;
//   views/take_while_kld.hpp:83-87,134-136 ; type_traits/particle_traits.hpp:92-107 (weight := 1).
// Candidate j draws Philox(seed, step, kPurposeResample, j): words 0,1 -> multinomial uniform,
// word 2 -> Bernoulli uniform.  Candidate 0 is never interspersed.
size_t resample(
    const double* states, const double* w, size_t n, const ResampleParams& p, const double* free_xy, uint64_t n_free,
    double* out_states, int64_t* out_ancestor) {
  // libstdc++ discrete_distribution::param_type::_M_initialize
  std::vector<double> cp(n);
  if (n >= 2) {
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum += w[i];
    double run = 0.0;
    for (size_t i = 0; i < n; ++i) {
      run += w[i] / sum;
      cp[i] = run;
    }
    cp[n - 1] = 1.0;
  }
  std::unordered_set<uint64_t> buckets;
  size_t count = 0;
  size_t out_n = 0;
  for (uint64_t j = 0; j < p.max_particles; ++j) {  // take(max)
    const Draw4 d = draw(p.seed, p.step, kPurposeResample, j);
    SE2 s;
    int64_t ancestor;
    const bool intersperse = (j > 0) && (p.random_state_probability > 0.0) && (u32(d.r[2]) < p.random_state_probability) &&
                             n_free > 0;
    if (intersperse) {
      s = random_state(free_xy, n_free, p.seed, p.step, j);
      ancestor = -1;
    } else {
      size_t idx = 0;
      if (n >= 2) {
        const double u = u53(d.r[0], d.r[1]);
        idx = static_cast<size_t>(std::lower_bound(cp.begin(), cp.end(), u) - cp.begin());
      }
      s = se2_load(states + 4 * idx);
      ancestor = static_cast<int64_t>(idx);
    }
    // kld_condition
    count++;
    buckets.insert(spatial_hash(s, p.hash_res));
    const bool keep = count <= p.min_particles || count <= kld_target_size(buckets.size(), p.kld_epsilon, p.kld_z);
    if (!keep) break;  // take_while drops the failing element
    se2_store(s, out_states + 4 * out_n);
    if (out_ancestor) out_ancestor[out_n] = ancestor;
    ++out_n;
  }
  return out_n;
}

// policies/on_motion.hpp:63-67,121-133
struct OnMotion {
  double min_d, min_a;
  bool has_latest{false};
  SE2 latest;
  bool operator()(const SE2& pose) {
    if (!has_latest) {
      latest = pose;
      has_latest = true;
      return true;
    }
    const SE2 delta = se2_mul(se2_inverse(latest), pose);
    const bool moved = std::sqrt(delta.x * delta.x + delta.y * delta.y) > min_d || std::abs(so2_log(delta.r)) > min_a;
    if (moved) latest = pose;
    return moved;
  }
};

// ----------------------------------------------------------------------------------------------
// The filter: beluga::Amcl (amcl_core.hpp:81-233) with DifferentialDriveModel +
// LikelihoodFieldModel | BeamSensorModel and the beluga_ros free-space random state generator.
// ----------------------------------------------------------------------------------------------
struct AmclConfig {
  // AmclParams amcl_core.hpp:34-55
  double update_min_d, update_min_a;
  uint64_t resample_interval;
  int selective_resampling;
  uint64_t min_particles, max_particles;
  double alpha_slow, alpha_fast, kld_epsilon, kld_z;
  double hash_res[3];
  // DifferentialDriveModelParam
  double alphas[5], distance_threshold;  // alpha5 = strafe noise (omni only)
  int motion_kind;  // 0 = differential, 1 = omnidirectional, 2 = stationary
  // sensor
  int sensor_kind;  // 0 = likelihood field, 1 = beam, 2 = likelihood field prob
  LfParams lf;
  BeamParams beam;
  uint64_t seed;
  int threads;  // 1 = std::execution::seq, >1 = par (OpenMP on the three transforms the reference parallelises)
};

struct Amcl {
  AmclConfig cfg;
  std::vector<int8_t> cells;
  Grid grid;
  std::vector<float> field;
  SE2 world_to_field;
  std::vector<double> free_xy;
  std::vector<double> states, weights;  // N x 4, N
  Thrun thrun;
  OnMotion on_motion;
  uint64_t every_n_current{0};
  bool force_update{true};
  // RollingWindow<SE2,2> (containers/circular_array.hpp:461-465): newest first, extrapolate on read
  bool have_prev{false};
  SE2 window0, window1;
  uint32_t step{0};
  // stage timings of the last update (seconds): propagate, reweight, normalize+policies, resample, estimate
  double t_stage[5]{0, 0, 0, 0, 0};
  long beam_steps{0};

  void set_map(const int8_t* c, int W, int H, double res, const double origin[4], const Traits& t) {
    cells.assign(c, c + static_cast<size_t>(W) * H);
    grid = Grid{cells.data(), W, H, res, se2_load(origin), t};
    world_to_field = se2_inverse(grid.origin);  // likelihood_field_model_base.hpp:99
    if (cfg.sensor_kind != 1) {
      field.resize(grid.size());
      make_likelihood_field(grid, cfg.lf, field.data());
    }
    free_xy.clear();  // occupancy_grid.hpp:164-171 + coordinates_at(index, kGlobal) :140-146
    for (size_t i = 0; i < grid.size(); ++i) {
      if (t.is_free(cells[i])) {
        double lx, ly, gx, gy;
        grid.coordinates_at(i, &lx, &ly);
        so2_act(grid.origin.r, lx, ly, &gx, &gy);
        free_xy.push_back(gx + grid.origin.x);
        free_xy.push_back(gy + grid.origin.y);
      }
    }
  }
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// 3x3 symmetric eigen-decomposition by cyclic Jacobi (stand-in for Eigen::SelfAdjointEigenSolver,
// random/multivariate_normal_distribution.hpp:117); returns transform = V * sqrt(diag(lambda)).
bool covariance_transform(const double cov[9], double T[9]) {
  double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = cov[3 * i + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double d = std::abs(a[i][j] - a[j][i]);
      if (d > 1e-12 * std::min(std::abs(a[i][j]), std::abs(a[j][i])) && d > 1e-300) return false;  // isApprox
    }
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int j = 0; j < 3; ++j) {
    if (a[j][j] < 0.0) {
      if (a[j][j] > -1e-14) a[j][j] = 0.0;
      else return false;  // negative eigenvalues
    }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = v[i][j] * std::sqrt(a[j][j]);
  return true;
}

}  // namespace

// ================================================================================================
// C ABI for ctypes (tests/, bench.py cpu_baseline, __graft_entry__.smoke only).
// ================================================================================================
extern "C" {

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { Philox::run(ctr, key, out); }

void orc_draw(uint64_t seed, uint32_t step, uint32_t purpose, uint64_t index, uint32_t out[4]) {
  const Draw4 d = draw(seed, step, purpose, index);
  std::memcpy(out, d.r, sizeof(d.r));
}

void orc_se2_mul(const double a[4], const double b[4], double out[4]) { se2_store(se2_mul(se2_load(a), se2_load(b)), out); }
void orc_se2_inverse(const double a[4], double out[4]) { se2_store(se2_inverse(se2_load(a)), out); }
void orc_se2_from_xytheta(double x, double y, double theta, double out[4]) { se2_store(SE2{so2_exp(theta), x, y}, out); }
double orc_so2_log(const double a[4]) { return so2_log(SO2{a[0], a[1]}); }

static Traits make_traits(const int8_t t[3]) { return Traits{t[0], t[1], t[2]}; }

void orc_make_likelihood_field(
    const int8_t* cells, int W, int H, double res, const int8_t traits[3], const double lf_params[5], int model_unknown_space,
    int only_obstacle_boundaries, float* out) {
  const double origin[4] = {1, 0, 0, 0};
  Grid g{cells, W, H, res, se2_load(origin), make_traits(traits)};
  LfParams p{lf_params[0], lf_params[1], lf_params[2], lf_params[3], lf_params[4], model_unknown_space, only_obstacle_boundaries};
  make_likelihood_field(g, p, out);
}

// algorithm/test_distance_map.cpp fixture: 1-D array, distance = |i-j|, neighbours = i±1.
void orc_distance_map_1d(const uint8_t* mask, int n, int max_value, int* out) {
  struct IndexPair {
    size_t nearest, index;
  };
  // same algorithm instantiated with the integer distance the reference test uses
  std::vector<int> dm(n, max_value);
  std::vector<bool> visited(n, false);
  auto compare = [&dm](const IndexPair& a, const IndexPair& b) { return dm[a.index] > dm[b.index]; };
  std::priority_queue<IndexPair, std::vector<IndexPair>, decltype(compare)> queue{compare};
  for (int i = 0; i < n; ++i)
    if (mask[i]) {
      visited[i] = true;
      dm[i] = 0;
      queue.push({static_cast<size_t>(i), static_cast<size_t>(i)});
    }
  while (!queue.empty()) {
    auto parent = queue.top();
    queue.pop();
    std::vector<size_t> nb;
    if (parent.index > 0) nb.push_back(parent.index - 1);  // fixture order: test_distance_map.cpp:32-42
    if (parent.index + 1 < static_cast<size_t>(n)) nb.push_back(parent.index + 1);
    for (size_t index : nb) {
      if (!visited[index]) {
        visited[index] = true;
        const int d = std::abs(static_cast<int>(parent.nearest) - static_cast<int>(index));
        if (d < max_value) {
          dm[index] = d;
          queue.push({parent.nearest, index});
        }
      }
    }
  }
  for (int i = 0; i < n; ++i) out[i] = dm[i];
}

void orc_lf_weights(
    const float* field, int W, int H, double res, const double origin[4], double max_laser_distance, const double* states,
    uint64_t n, const double* pts, uint64_t B, int threads, double* out) {
  const SE2 w2f = se2_inverse(se2_load(origin));
  (void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
  for (long i = 0; i < static_cast<long>(n); ++i) {
    out[i] = lf_weight(field, W, H, res, w2f, max_laser_distance, se2_load(states + 4 * i), pts, B);
  }
}

void orc_lf_prob_weights(
    const float* field, int W, int H, double res, const double origin[4], double max_laser_distance, const double* states,
    uint64_t n, const double* pts, uint64_t B, double* out) {
  const SE2 w2f = se2_inverse(se2_load(origin));
  for (uint64_t i = 0; i < n; ++i) out[i] = lf_prob_weight(field, W, H, res, w2f, max_laser_distance, se2_load(states + 4 * i), pts, B);
}

// kind 1 = omnidirectional (alphas[5]), kind 2 = stationary (control ignored)
void orc_propagate_kind(double* states, uint64_t n, int kind, const double pose[4], const double prev[4], const double alphas[5],
                        double distance_threshold, uint64_t seed, uint32_t step, uint64_t index_offset) {
  if (kind == 1) {
    const OmniSampler s = omni_sampler(se2_load(pose), se2_load(prev), alphas, distance_threshold);
    for (uint64_t i = 0; i < n; ++i) se2_store(omni_apply(se2_load(states + 4 * i), s, seed, step, index_offset + i), states + 4 * i);
  } else if (kind == 2) {
    for (uint64_t i = 0; i < n; ++i) se2_store(stationary_apply(se2_load(states + 4 * i), seed, step, index_offset + i), states + 4 * i);
  } else {
    const DiffDriveSampler s = diffdrive_sampler(se2_load(pose), se2_load(prev), alphas, distance_threshold);
    for (uint64_t i = 0; i < n; ++i) se2_store(diffdrive_apply(se2_load(states + 4 * i), s, seed, step, index_offset + i), states + 4 * i);
  }
}

void orc_beam_weights(
    const int8_t* cells, int W, int H, double res, const double origin[4], const int8_t traits[3], const double beam_params[7],
    const double* states, uint64_t n, const double* pts, uint64_t B, int threads, double* out, int64_t* total_steps) {
  Grid g{cells, W, H, res, se2_load(origin), make_traits(traits)};
  BeamParams p{beam_params[0], beam_params[1], beam_params[2], beam_params[3], beam_params[4], beam_params[5], beam_params[6]};
  long steps = 0;
  (void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static) reduction(+ : steps)
  for (long i = 0; i < static_cast<long>(n); ++i) {
    long s = 0;
    out[i] = beam_weight(g, p, se2_load(states + 4 * i), pts, B, &s);
    steps += s;
  }
  if (total_steps) *total_steps = steps;
}

int orc_ray_cast(
    const int8_t* cells, int W, int H, double res, const double origin[4], const int8_t traits[3], const double pose[4],
    double max_range, double bearing_theta, double* out) {
  Grid g{cells, W, H, res, se2_load(origin), make_traits(traits)};
  Ray2d ray{g, se2_load(pose), max_range};
  return ray.cast(so2_exp(bearing_theta), out) ? 1 : 0;
}

int orc_bresenham(int x0, int y0, int x1, int y1, int modified, int* out_xy, int max_points) {
  Bresenham it(x0, y0, x1, y1, modified != 0);
  int n = 0;
  for (; !it.done() && n < max_points; it.next()) {
    out_xy[2 * n] = it.cx;
    out_xy[2 * n + 1] = it.cy;
    ++n;
  }
  return n;
}

void orc_diffdrive_sampler(const double pose[4], const double prev[4], const double alphas[4], double distance_threshold, double out[6]) {
  const DiffDriveSampler s = diffdrive_sampler(se2_load(pose), se2_load(prev), alphas, distance_threshold);
  out[0] = s.m1;
  out[1] = s.s1;
  out[2] = s.mt;
  out[3] = s.st;
  out[4] = s.m2;
  out[5] = s.s2;
}

void orc_propagate(double* states, uint64_t n, const double sampler[6], uint64_t seed, uint32_t step, uint64_t index_offset, int threads) {
  const DiffDriveSampler s{sampler[0], sampler[1], sampler[2], sampler[3], sampler[4], sampler[5]};
  (void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
  for (long i = 0; i < static_cast<long>(n); ++i) {
    se2_store(diffdrive_apply(se2_load(states + 4 * i), s, seed, step, index_offset + i), states + 4 * i);
  }
}

double orc_normalize(double* w, uint64_t n) { return normalize(w, n, 1); }
double orc_effective_sample_size(const double* w, uint64_t n) { return effective_sample_size(w, n); }

// Thrun estimator as an opaque two-filter state: state[0]=slow.output, state[1]=fast.output
double orc_thrun(double state[2], double alpha_slow, double alpha_fast, const double* w, uint64_t n) {
  Thrun t;
  t.slow = ExponentialFilter{state[0], alpha_slow};
  t.fast = ExponentialFilter{state[1], alpha_fast};
  const double p = t(w, n);
  state[0] = t.slow.output;
  state[1] = t.fast.output;
  return p;
}

uint64_t orc_kld_target_size(uint64_t k, double epsilon, double z) { return kld_target_size(k, epsilon, z); }

// views/test_take_while_kld.cpp: count how many leading hashes satisfy kld_condition(min, eps, z).
uint64_t orc_kld_take_while(const uint64_t* hashes, uint64_t n, uint64_t min, double epsilon, double z) {
  std::unordered_set<uint64_t> buckets;
  uint64_t count = 0, kept = 0;
  for (uint64_t i = 0; i < n; ++i) {
    count++;
    buckets.insert(hashes[i]);
    if (!(count <= min || count <= kld_target_size(buckets.size(), epsilon, z))) break;
    ++kept;
  }
  return kept;
}

uint64_t orc_spatial_hash(const double state[4], const double res[3]) { return spatial_hash(se2_load(state), res); }
uint64_t orc_spatial_hash_xyt(double x, double y, double t, const double res[3]) {
  constexpr unsigned kBits = 64 / 3;
  return floor_and_fibo_hash(x / res[0], 0) ^ floor_and_fibo_hash(y / res[1], kBits) ^ floor_and_fibo_hash(t / res[2], 2 * kBits);
}

uint64_t orc_resample(
    const double* states, const double* w, uint64_t n, uint64_t min_particles, uint64_t max_particles, double kld_epsilon,
    double kld_z, const double hash_res[3], double random_state_probability, uint64_t seed, uint32_t step, const double* free_xy,
    uint64_t n_free, double* out_states, int64_t* out_ancestor) {
  ResampleParams p{min_particles, max_particles, kld_epsilon, kld_z, {hash_res[0], hash_res[1], hash_res[2]},
                   random_state_probability, seed, step};
  return resample(states, w, n, p, free_xy, n_free, out_states, out_ancestor);
}

void orc_estimate(const double* states, const double* w, uint64_t n, double mean[4], double cov[9]) { estimate(states, w, n, mean, cov); }

void orc_cluster_ids(const double* states, const double* w, uint64_t n, double linear_res, double angular_res, double percentile,
                     uint64_t* out) {
  const std::vector<size_t> ids = cluster_ids(states, w, n, ClusterParams{linear_res, angular_res, percentile});
  for (uint64_t i = 0; i < n; ++i) out[i] = ids[i];
}
void orc_cluster_based_estimate(const double* states, const double* w, uint64_t n, double linear_res, double angular_res,
                                double percentile, double mean[4], double cov[9]) {
  cluster_based_estimate(states, w, n, ClusterParams{linear_res, angular_res, percentile}, mean, cov);
}

int orc_covariance_transform(const double cov[9], double T[9]) { return covariance_transform(cov, T) ? 1 : 0; }

// amcl_core.hpp:131-147 + multivariate_normal_distribution.hpp:96-126 (mean + T * delta, delta ~ N(0,I)^3)
int orc_init_normal(double* states, double* w, uint64_t n, const double mean_xytheta[3], const double cov[9], uint64_t seed, uint64_t index_offset) {
  double T[9];
  if (!covariance_transform(cov, T)) return 0;
  for (uint64_t i = 0; i < n; ++i) {
    const Draw4 a = draw(seed, 0, kPurposeInitA, index_offset + i);
    const Draw4 b = draw(seed, 0, kPurposeInitB, index_offset + i);
    double z[4];
    box_muller(u53(a.r[0], a.r[1]), u53(a.r[2], a.r[3]), &z[0], &z[1]);
    box_muller(u53(b.r[0], b.r[1]), u53(b.r[2], b.r[3]), &z[2], &z[3]);
    double v[3];
    for (int r = 0; r < 3; ++r) v[r] = mean_xytheta[r] + (T[3 * r] * z[0] + T[3 * r + 1] * z[1] + T[3 * r + 2] * z[2]);
    se2_store(SE2{so2_exp(v[2]), v[0], v[1]}, states + 4 * i);
    w[i] = 1.0;
  }
  return 1;
}

// ---- filter object ----------------------------------------------------------------------------
struct orc_amcl_config {
  double update_min_d, update_min_a;
  uint64_t resample_interval;
  int32_t selective_resampling;
  int32_t sensor_kind;
  uint64_t min_particles, max_particles;
  double alpha_slow, alpha_fast, kld_epsilon, kld_z;
  double hash_res[3];
  double alphas[4];
  double distance_threshold;
  double lf[5];
  int32_t lf_model_unknown_space, lf_only_obstacle_boundaries;
  double beam[7];
  uint64_t seed;
  int32_t threads;
  int32_t motion_kind;
  double alpha5;
};

void* orc_amcl_create(const orc_amcl_config* c) {
  auto* a = new Amcl();
  AmclConfig& k = a->cfg;
  k.update_min_d = c->update_min_d;
  k.update_min_a = c->update_min_a;
  k.resample_interval = c->resample_interval;
  k.selective_resampling = c->selective_resampling;
  k.min_particles = c->min_particles;
  k.max_particles = c->max_particles;
  k.alpha_slow = c->alpha_slow;
  k.alpha_fast = c->alpha_fast;
  k.kld_epsilon = c->kld_epsilon;
  k.kld_z = c->kld_z;
  for (int i = 0; i < 3; ++i) k.hash_res[i] = c->hash_res[i];
  for (int i = 0; i < 4; ++i) k.alphas[i] = c->alphas[i];
  k.alphas[4] = c->alpha5;
  k.motion_kind = c->motion_kind;
  k.distance_threshold = c->distance_threshold;
  k.sensor_kind = c->sensor_kind;
  k.lf = LfParams{c->lf[0], c->lf[1], c->lf[2], c->lf[3], c->lf[4], c->lf_model_unknown_space, c->lf_only_obstacle_boundaries};
  k.beam = BeamParams{c->beam[0], c->beam[1], c->beam[2], c->beam[3], c->beam[4], c->beam[5], c->beam[6]};
  k.seed = c->seed;
  k.threads = c->threads < 1 ? 1 : c->threads;
  a->thrun.slow.alpha = k.alpha_slow;
  a->thrun.fast.alpha = k.alpha_fast;
  a->on_motion.min_d = k.update_min_d;
  a->on_motion.min_a = k.update_min_a;
  return a;
}
void orc_amcl_destroy(void* h) { delete static_cast<Amcl*>(h); }

void orc_amcl_set_map(void* h, const int8_t* cells, int W, int H, double res, const double origin[4], const int8_t traits[3]) {
  static_cast<Amcl*>(h)->set_map(cells, W, H, res, origin, make_traits(traits));
}
// Install a pre-built likelihood field (e.g. produced by the product's builder) instead of rebuilding.
void orc_amcl_set_field(void* h, const float* field) {
  auto* a = static_cast<Amcl*>(h);
  a->field.assign(field, field + a->grid.size());
}
void orc_amcl_get_field(void* h, float* out) {
  auto* a = static_cast<Amcl*>(h);
  std::memcpy(out, a->field.data(), a->field.size() * sizeof(float));
}
uint64_t orc_amcl_num_free(void* h) { return static_cast<Amcl*>(h)->free_xy.size() / 2; }

void orc_amcl_set_particles(void* h, const double* states, const double* w, uint64_t n) {
  auto* a = static_cast<Amcl*>(h);
  a->states.assign(states, states + 4 * n);
  a->weights.assign(w, w + n);
  a->force_update = true;  // amcl_core.hpp:136
}
uint64_t orc_amcl_num_particles(void* h) { return static_cast<Amcl*>(h)->weights.size(); }
void orc_amcl_get_particles(void* h, double* states, double* w) {
  auto* a = static_cast<Amcl*>(h);
  std::memcpy(states, a->states.data(), a->states.size() * sizeof(double));
  std::memcpy(w, a->weights.data(), a->weights.size() * sizeof(double));
}
int orc_amcl_init_normal(void* h, const double mean_xytheta[3], const double cov[9]) {
  auto* a = static_cast<Amcl*>(h);
  const uint64_t n = a->cfg.max_particles;  // take_exactly(max_particles) amcl_core.hpp:134
  a->states.resize(4 * n);
  a->weights.resize(n);
  if (!orc_init_normal(a->states.data(), a->weights.data(), n, mean_xytheta, cov, a->cfg.seed, 0)) return 0;
  a->force_update = true;
  return 1;
}
// beluga_ros::Amcl::initialize_from_map (beluga_ros/include/beluga_ros/amcl.hpp:191-198,209): take_exactly(max_particles)
// draws of MultivariateUniformDistribution over the free cells (random/multivariate_uniform_distribution.hpp:126-161),
// weight 1 (particle_traits.hpp:92-107).  Particle i takes the stream's (step 0, index i) random state.
int orc_amcl_init_from_map(void* h) {
  auto* a = static_cast<Amcl*>(h);
  const uint64_t n_free = a->free_xy.size() / 2;
  if (n_free == 0) return 0;  // the reference asserts !free_states_.empty()
  const uint64_t n = a->cfg.max_particles;
  a->states.resize(4 * n);
  a->weights.assign(n, 1.0);
  for (uint64_t i = 0; i < n; ++i) se2_store(random_state(a->free_xy.data(), n_free, a->cfg.seed, 0, i), a->states.data() + 4 * i);
  a->force_update = true;
  return 1;
}
void orc_amcl_force_update(void* h) { static_cast<Amcl*>(h)->force_update = true; }
void orc_amcl_stage_times(void* h, double out[5]) { std::memcpy(out, static_cast<Amcl*>(h)->t_stage, sizeof(double) * 5); }
int64_t orc_amcl_beam_steps(void* h) { return static_cast<Amcl*>(h)->beam_steps; }

// amcl_core.hpp:165-201.  Returns 1 and fills mean/cov if an update ran, 0 for std::nullopt.
// info[0] = resampled (0/1), info[1] = random_state_probability, info[2] = ESS (if evaluated, else -1),
// info[3] = sum of weights before normalisation.
int orc_amcl_update(void* h, const double control[4], const double* pts, uint64_t B, double mean[4], double cov[9], double info[4]) {
  auto* a = static_cast<Amcl*>(h);
  AmclConfig& k = a->cfg;
  const uint64_t n = a->weights.size();
  if (n == 0) return 0;  // :166-168
  const SE2 pose = se2_load(control);
  if (!a->on_motion(pose) && !a->force_update) return 0;  // :170-172

  // control_action_window_ << control  (RollingWindow<SE2,2>, newest first, extrapolated when short)
  if (!a->have_prev) {
    a->window0 = pose;
    a->window1 = pose;
    a->have_prev = true;
  } else {
    a->window1 = a->window0;
    a->window0 = pose;
  }
  a->step += 1;
  const int threads = k.threads;

  double t0 = now_s();
  // propagate (:174-175 ; actions/propagate.hpp:57-79)
  double* S = a->states.data();
  double* Wt = a->weights.data();
  if (k.motion_kind == 0) {
    const DiffDriveSampler sampler = diffdrive_sampler(a->window0, a->window1, k.alphas, k.distance_threshold);
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
    for (long i = 0; i < static_cast<long>(n); ++i) {
      se2_store(diffdrive_apply(se2_load(S + 4 * i), sampler, k.seed, a->step, static_cast<uint64_t>(i)), S + 4 * i);
    }
  } else if (k.motion_kind == 1) {
    const OmniSampler sampler = omni_sampler(a->window0, a->window1, k.alphas, k.distance_threshold);
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
    for (long i = 0; i < static_cast<long>(n); ++i) {
      se2_store(omni_apply(se2_load(S + 4 * i), sampler, k.seed, a->step, static_cast<uint64_t>(i)), S + 4 * i);
    }
  } else {
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
    for (long i = 0; i < static_cast<long>(n); ++i) {
      se2_store(stationary_apply(se2_load(S + 4 * i), k.seed, a->step, static_cast<uint64_t>(i)), S + 4 * i);
    }
  }
  double t1 = now_s();
  // reweight (:176 ; actions/reweight.hpp:53-60)
  long steps = 0;
  if (k.sensor_kind == 0) {
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
    for (long i = 0; i < static_cast<long>(n); ++i) {
      Wt[i] = Wt[i] * lf_weight(a->field.data(), a->grid.W, a->grid.H, a->grid.res, a->world_to_field, k.lf.max_laser_distance,
                                se2_load(S + 4 * i), pts, B);
    }
  } else if (k.sensor_kind == 2) {
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
    for (long i = 0; i < static_cast<long>(n); ++i) {
      Wt[i] = Wt[i] * lf_prob_weight(a->field.data(), a->grid.W, a->grid.H, a->grid.res, a->world_to_field, k.lf.max_laser_distance,
                                     se2_load(S + 4 * i), pts, B);
    }
  } else {
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static) reduction(+ : steps)
    for (long i = 0; i < static_cast<long>(n); ++i) {
      long s = 0;
      Wt[i] = Wt[i] * beam_weight(a->grid, k.beam, se2_load(S + 4 * i), pts, B, &s);
      steps += s;
    }
  }
  a->beam_steps = steps;
  double t2 = now_s();
  // normalize (:177)
  const double sum = normalize(Wt, n, threads);
  // :179
  const double random_state_probability = a->thrun(Wt, n);
  // :181 resample_policy_: every_n [&& on_effective_size_drop]
  a->every_n_current = (a->every_n_current + 1) % k.resample_interval;
  bool do_resample = a->every_n_current == 0;
  double ess = -1.0;
  if (do_resample && k.selective_resampling) {
    ess = effective_sample_size(Wt, n);
    do_resample = ess < static_cast<double>(n) * 0.5;
  }
  double t3 = now_s();
  if (do_resample) {
    if (random_state_probability > 0.0) a->thrun.reset();  // :184-186
    ResampleParams rp{k.min_particles, k.max_particles, k.kld_epsilon, k.kld_z, {k.hash_res[0], k.hash_res[1], k.hash_res[2]},
                      random_state_probability, k.seed, a->step};
    std::vector<double> out(4 * k.max_particles);
    const size_t m = resample(S, Wt, n, rp, a->free_xy.data(), a->free_xy.size() / 2, out.data(), nullptr);
    out.resize(4 * m);
    a->states.swap(out);
    a->weights.assign(m, 1.0);
  }
  double t4 = now_s();
  a->force_update = false;  // :199
  estimate(a->states.data(), a->weights.data(), a->weights.size(), mean, cov);  // :200
  double t5 = now_s();
  a->t_stage[0] = t1 - t0;
  a->t_stage[1] = t2 - t1;
  a->t_stage[2] = t3 - t2;
  a->t_stage[3] = t4 - t3;
  a->t_stage[4] = t5 - t4;
  if (info) {
    info[0] = do_resample ? 1.0 : 0.0;
    info[1] = random_state_probability;
    info[2] = ess;
    info[3] = sum;
  }
  return 1;
}

// Caller-side scan preparation: beluga_ros::Amcl::update(pose, LaserScan) (beluga_ros/src/amcl.cpp:54-63) over
// beluga_ros::LaserScan (beluga_ros/include/beluga_ros/laser_scan.hpp:46-100), BaseLaserScan
// (beluga/sensor/data/laser_scan.hpp:64-90) and views::take_evenly (beluga/views/take_evenly.hpp:126-148).
// origin_se3 = Sophus::SE3d::data() = (qx, qy, qz, qw, tx, ty, tz).  Returns the number of points written.
uint64_t orc_take_evenly_index(uint64_t pos, uint64_t size, uint64_t count) {  // index of the pos-th element taken
  if (count > size) return pos;
  if (pos == 0) return 0;
  if (count == 1) return size;
  const int64_t a = static_cast<int64_t>(pos) * (static_cast<int64_t>(size) - 1);
  const int64_t b = static_cast<int64_t>(count) - 1;
  return static_cast<uint64_t>(a / b + ((a % b == 0) ? 0 : 1));
}

uint64_t orc_prepare_laser_scan(const float* ranges, uint64_t n, float angle_min, float angle_increment, float range_min,
                                float range_max, const double origin_se3[7], uint64_t max_beams, double min_range,
                                double max_range, double* out_xy) {
  const double lo = std::max(static_cast<double>(range_min), min_range);
  const double hi = std::min(static_cast<double>(range_max), max_range);
  const uint64_t taken = n == 0 ? 0 : (max_beams > n ? n : max_beams);
  const double qx = origin_se3[0], qy = origin_se3[1], qz = origin_se3[2], qw = origin_se3[3];
  uint64_t m = 0;
  for (uint64_t k = 0; k < taken; ++k) {
    const uint64_t i = orc_take_evenly_index(k, n, max_beams);
    if (i >= n) break;
    const double range = static_cast<double>(ranges[i]);
    const double theta = static_cast<double>(angle_min + static_cast<float>(static_cast<int>(i)) * angle_increment);
    if (std::isnan(range) || !(range >= lo) || !(range <= hi)) continue;
    const double px = range * std::cos(theta), py = range * std::sin(theta), pz = 0.0;
    // Sophus SO3 * point: uv = 2 * (q.vec x p); p + q.w * uv + q.vec x uv   (so3.hpp operator*)
    double ux = qy * pz - qz * py, uy = qz * px - qx * pz, uz = qx * py - qy * px;
    ux += ux;
    uy += uy;
    uz += uz;
    const double rx = px + qw * ux + (qy * uz - qz * uy);
    const double ry = py + qw * uy + (qz * ux - qx * uz);
    out_xy[2 * m] = rx + origin_se3[4];
    out_xy[2 * m + 1] = ry + origin_se3[5];
    ++m;
  }
  return m;
}

// beluga_ros::Amcl::update(pose, SparsePointCloud3f) (beluga_ros/src/amcl.cpp:67-81): origin * p.cast<double>(), x and y kept.
void orc_project_point_cloud(const float* xyz, uint64_t n, const double origin_se3[7], double* out_xy) {
  const double qx = origin_se3[0], qy = origin_se3[1], qz = origin_se3[2], qw = origin_se3[3];
  for (uint64_t i = 0; i < n; ++i) {
    const double px = static_cast<double>(xyz[3 * i]), py = static_cast<double>(xyz[3 * i + 1]), pz = static_cast<double>(xyz[3 * i + 2]);
    // Sophus SO3 * point: uv = 2 * (q.vec x p); p + q.w * uv + q.vec x uv   (so3.hpp operator*)
    double ux = qy * pz - qz * py, uy = qz * px - qx * pz, uz = qx * py - qy * px;
    ux += ux;
    uy += uy;
    uz += uz;
    out_xy[2 * i] = (px + qw * ux + (qy * uz - qz * uy)) + origin_se3[4];
    out_xy[2 * i + 1] = (py + qw * uy + (qz * ux - qx * uz)) + origin_se3[5];
  }
}

int orc_max_threads() {
#if defined(_OPENMP)
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
