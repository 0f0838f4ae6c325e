// Exercises include/beluga_amd/amcl.hpp the way beluga's own tests exercise beluga::Amcl
// (beluga/test/beluga/algorithm/test_amcl_core.cpp:73-186): construct, initialize, update, read particles.
// Prints "key value" lines that tests/test_cpp_facade.py compares with the Python facade run on the same inputs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "beluga_amd/amcl.hpp"

int main(int argc, char** argv) {
  using namespace beluga_amd;
  // 64 x 64 grid, 0.1 m, a wall along x at row 40 and one along y at column 50.
  const std::uint32_t W = 64, H = 64;
  std::vector<std::int8_t> cells(W * H, 0);
  for (std::uint32_t x = 0; x < W; ++x) cells[40 * W + x] = 100;
  for (std::uint32_t y = 0; y < H; ++y) cells[y * W + 50] = 100;
  OccupancyGridView map;
  map.cells = cells.data();
  map.width = W;
  map.height = H;
  map.resolution = 0.1;
  map.origin = SE2d{0.0, -1.0, -1.0};

  AmclParams params;
  params.min_particles = 300;
  params.max_particles = 1500;
  const DifferentialDriveModelParam motion{0.1, 0.05, 0.1, 0.05};
  LikelihoodFieldModelParam lf;
  lf.max_obstacle_distance = 2.0;
  lf.max_laser_distance = 100.0;

  try {
    Amcl filter{map, motion, lf, params, /*seed=*/123};
    if (argc > 1 && std::strcmp(argv[1], "bad-covariance") == 0) {
      try {
        filter.initialize(SE2d{0.0, 1.0, 1.0}, Matrix3d{1, 2, 0, 0, 1, 0, 0, 0, 1});
        std::printf("bad_covariance accepted\n");
        return 2;
      } catch (const std::runtime_error& e) {
        std::printf("bad_covariance %s\n", e.what());
        return 0;
      }
    }
    // empty filter: update returns nullopt (amcl_core.hpp:166-168)
    std::printf("empty_update %d\n", filter.update(SE2d{}, Amcl::measurement_type{{1.0, 0.0}}).has_value() ? 1 : 0);
    filter.initialize(SE2d{0.3, 1.0, 1.0}, Matrix3d{0.04, 0, 0, 0, 0.04, 0, 0, 0, 0.01});
    std::printf("initial_particles %zu\n", filter.particles().size());
    std::vector<std::pair<double, double>> scan;
    for (int b = 0; b < 90; ++b) {
      const double a = -1.5 + b * (3.0 / 90);
      scan.emplace_back(1.8 * std::cos(a), 1.8 * std::sin(a));
    }
    double ox = 0, oy = 0, ot = 0;
    for (int c = 0; c < 4; ++c) {
      ox += 0.3 * std::cos(ot);
      oy += 0.3 * std::sin(ot);
      ot += 0.05;
      const auto est = filter.update(SE2d{ot, ox, oy}, scan);
      if (!est) {
        std::printf("update %d nullopt\n", c);
        continue;
      }
      std::printf("update %d %.17g %.17g %.17g %.17g %.17g %.17g %zu\n", c, est->first.c, est->first.s, est->first.x, est->first.y,
                  est->second[0], est->second[8], filter.particles().size());
    }
    std::printf("below_threshold %d\n", filter.update(SE2d{ot, ox + 0.001, oy}, scan).has_value() ? 1 : 0);
    filter.force_update();
    std::printf("forced %d\n", filter.update(SE2d{ot, ox + 0.001, oy}, scan).has_value() ? 1 : 0);
    std::printf("cloud %zu\n", filter.sample_particle_cloud(64, 1).size());
    std::printf("field_center %.9g\n", static_cast<double>(filter.likelihood_field()[40 * W + 10]));
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error %s\n", e.what());
    return 3;
  }
  return 0;
}
