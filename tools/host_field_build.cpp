// Times mcl::build_likelihood_field (the product's host build of the reference's wavefront field) on a synthetic rooms map and prints an
// FNV-1a hash of the field's bits: the same hash before and after a change to map_build.cpp = the same field.
//   g++ -O3 -std=c++17 -pthread -Iinclude -Ibeluga_amd/csrc tools/host_field_build.cpp beluga_amd/csrc/map_build.cpp -o /tmp/host_field_build
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "map_build.h"

int main(int argc, char** argv) {
  const uint32_t W = argc > 1 ? std::atoi(argv[1]) : 4000, H = argc > 2 ? std::atoi(argv[2]) : W;
  const int boundaries = argc > 3 ? std::atoi(argv[3]) : 0, unknown_space = argc > 4 ? std::atoi(argv[4]) : 0;
  std::vector<int8_t> cells(static_cast<size_t>(W) * H, 0);
  uint64_t state = 0x9E3779B97F4A7C15ull;
  auto next = [&state]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
  for (uint32_t x = 0; x < W; ++x) cells[x] = cells[static_cast<size_t>(H - 1) * W + x] = 100;
  for (uint32_t y = 0; y < H; ++y) cells[static_cast<size_t>(y) * W] = cells[static_cast<size_t>(y) * W + W - 1] = 100;
  for (int k = 0; k < static_cast<int>(W / 20); ++k) {  // walls with doors, blobs, patches of unknown space
    const uint32_t x0 = next() % W, y0 = next() % H, len = 20 + next() % (W / 4);
    const bool horizontal = next() & 1;
    for (uint32_t t = 0; t < len; ++t) {
      const uint32_t x = horizontal ? x0 + t : x0, y = horizontal ? y0 : y0 + t;
      if (x < W && y < H && (t % 97) > 8) cells[static_cast<size_t>(y) * W + x] = 100;
    }
    const uint32_t ux = next() % W, uy = next() % H;
    for (uint32_t dy = 0; dy < 12 && uy + dy < H; ++dy)
      for (uint32_t dx = 0; dx < 12 && ux + dx < W; ++dx) cells[static_cast<size_t>(uy + dy) * W + ux + dx] = (k % 3 == 0) ? -1 : 100;
  }
  mcl_lf_params p{};
  p.max_obstacle_distance = 2.0;
  p.max_laser_distance = 100.0;
  p.z_hit = 0.5;
  p.z_random = 0.5;
  p.sigma_hit = 0.2;
  p.only_obstacle_boundaries = boundaries;
  p.model_unknown_space = unknown_space;
  mcl::OccupancyTraits traits{0, -1, 100};  // free, unknown, occupied
  std::vector<float> field;
  const auto t0 = std::chrono::steady_clock::now();
  mcl::build_likelihood_field(cells.data(), W, H, 0.05, traits, p, field);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  uint64_t h = 1469598103934665603ull;
  for (float v : field) {
    uint32_t b;
    std::memcpy(&b, &v, 4);
    for (int k = 0; k < 4; ++k) h = (h ^ ((b >> (8 * k)) & 0xFF)) * 1099511628211ull;
  }
  std::printf("%u x %u boundaries %d unknown %d: %.0f ms, field hash %016llx\n", W, H, boundaries, unknown_space, ms, static_cast<unsigned long long>(h));
  return 0;
}
