// map_build.cpp — see map_build.h.
#include "map_build.h"

#include <algorithm>
#include <cmath>
#include <queue>
#include <thread>

namespace mcl {
namespace {

constexpr double kPiD = 3.14159265358979323846264338327950288;

struct Cells {
  const int8_t* data;
  uint32_t W, H;
  OccupancyTraits t;
  bool occupied(size_t i) const { return data[i] == t.occupied_value; }
  bool free_cell(size_t i) const { return data[i] == t.free_value; }
  bool unknown(size_t i) const { return data[i] == t.unknown_value; }
  // occupancy_grid.hpp:191-206: occupied with at least one free 4-neighbour
  bool obstacle_edge(size_t i) const {
    if (!occupied(i)) return false;
    const uint32_t xi = static_cast<uint32_t>(i % W), yi = static_cast<uint32_t>(i / W);
    return (xi + 1 < W && free_cell(i + 1)) || (yi + 1 < H && free_cell(i + W)) || (xi > 0 && free_cell(i - 1)) ||
           (yi > 0 && free_cell(i - W));
  }
};

struct Entry {
  uint32_t nearest_obstacle, index;
};

// Squared distance between cell centres, evaluated like the reference's lambda
// (likelihood_field_model_base.hpp:131-133 over regular_grid.hpp:87-89): double arithmetic, float result.
inline float squared_distance(uint32_t W, double res, uint32_t a, uint32_t b) {
  const double ax = (static_cast<double>(static_cast<int>(a % W)) + 0.5) * res;
  const double ay = (static_cast<double>(static_cast<int>(a / W)) + 0.5) * res;
  const double bx = (static_cast<double>(static_cast<int>(b % W)) + 0.5) * res;
  const double by = (static_cast<double>(static_cast<int>(b / W)) + 0.5) * res;
  const double dx = ax - bx, dy = ay - by;
  return static_cast<float>(dx * dx + dy * dy);
}

// Rows [first, last) of a pass whose cells do not depend on one another, on all cores (any split gives the same bits).
template <class Fn>
void parallel_rows(uint32_t rows, Fn&& fn) {
  unsigned workers = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
  if (rows < 512) workers = 1;
  if (workers <= 1) {
    fn(0u, rows);
    return;
  }
  std::vector<std::thread> pool;
  const uint32_t per = (rows + workers - 1) / workers;
  for (unsigned k = 0; k < workers; ++k) {
    const uint32_t a = std::min(rows, k * per), b = std::min(rows, a + per);
    if (a < b) pool.emplace_back([&fn, a, b] { fn(a, b); });
  }
  for (auto& t : pool) t.join();
}

}  // namespace

void build_likelihood_field(const int8_t* cells, uint32_t W, uint32_t H, double resolution, const OccupancyTraits& traits,
                            const mcl_lf_params& p, std::vector<float>& field) {
  const Cells g{cells, W, H, traits};
  const size_t n = static_cast<size_t>(W) * H;
  const float max_sq = static_cast<float>(p.max_obstacle_distance * p.max_obstacle_distance);

  // distance_map.hpp:55-98.  The wavefront pops the closest frontier cell and labels its unvisited
  // 4-neighbours (+x, +y, -x, -y: linear_grid.hpp:113-130) with the distance to the PARENT's obstacle.
  // Only the wavefront itself depends on an order (the pop order of the reference's std::priority_queue decides which neighbour
  // labels a cell: it is kept as it is, 85 % of the build's time at 16 M cells); the seed mask before it and the passes behind
  // it go row by row on all cores, with the same bits.
  std::vector<float>& dist = field;
  dist.resize(n);
  std::vector<uint8_t> seed_mask(n);
  parallel_rows(H, [&](uint32_t y0, uint32_t y1) {
    for (size_t i = static_cast<size_t>(y0) * W; i < static_cast<size_t>(y1) * W; ++i) {
      const bool seed = p.only_obstacle_boundaries ? g.obstacle_edge(i) : g.occupied(i);
      seed_mask[i] = seed ? 1 : 0;
      dist[i] = seed ? 0.f : max_sq;
    }
  });
  std::vector<bool> visited(n, false);
  auto farther = [&dist](const Entry& a, const Entry& b) { return dist[a.index] > dist[b.index]; };
  std::priority_queue<Entry, std::vector<Entry>, decltype(farther)> frontier(farther);
  for (size_t i = 0; i < n; ++i) {  // in index order, like the reference's enumerate (distance_map.hpp:74-80)
    if (seed_mask[i]) {
      visited[i] = true;
      frontier.push(Entry{static_cast<uint32_t>(i), static_cast<uint32_t>(i)});
    }
  }
  std::vector<uint8_t>().swap(seed_mask);
  auto relax = [&](const Entry& parent, size_t index) {
    if (visited[index]) return;
    visited[index] = true;
    const float d = squared_distance(W, resolution, parent.nearest_obstacle, static_cast<uint32_t>(index));
    if (d < max_sq) {
      dist[index] = d;
      frontier.push(Entry{parent.nearest_obstacle, static_cast<uint32_t>(index)});
    }
  };
  while (!frontier.empty()) {
    const Entry parent = frontier.top();
    frontier.pop();
    const uint32_t xi = parent.index % W, yi = parent.index / W;
    if (xi + 1 < W) relax(parent, static_cast<size_t>(parent.index) + 1);
    if (yi + 1 < H) relax(parent, static_cast<size_t>(parent.index) + W);
    if (xi > 0) relax(parent, static_cast<size_t>(parent.index) - 1);
    if (yi > 0) relax(parent, static_cast<size_t>(parent.index) - W);
  }

  // likelihood_field_model_base.hpp:136-146
  const double two_squared_sigma = 2 * p.sigma_hit * p.sigma_hit;
  const double amplitude = p.z_hit / (p.sigma_hit * std::sqrt(2 * kPiD));
  const double offset = p.z_random / p.max_laser_distance;

  float overlay = 0.f;
  if (p.model_unknown_space) {  // :160-179
    const double inverse_max_distance = 1 / p.max_laser_distance;
    const double squared_background_distance = -two_squared_sigma * std::log((inverse_max_distance - offset) / amplitude);
    overlay = std::min(max_sq, static_cast<float>(squared_background_distance));
  }
  parallel_rows(H, [&](uint32_t y0, uint32_t y1) {
    for (size_t i = static_cast<size_t>(y0) * W; i < static_cast<size_t>(y1) * W; ++i) {
      if (p.model_unknown_space) {
        const bool masked = p.only_obstacle_boundaries ? (g.unknown(i) || (g.occupied(i) && !g.obstacle_edge(i))) : g.unknown(i);
        if (masked) dist[i] = overlay;
      }
      // :181-182, in place on the float map
      field[i] = static_cast<float>(amplitude * std::exp(-static_cast<double>(dist[i]) / two_squared_sigma) + offset);
    }
  });
}

void collect_free_cells(const int8_t* cells, uint32_t W, uint32_t H, const OccupancyTraits& traits, std::vector<uint32_t>& out) {
  out.clear();
  const size_t n = static_cast<size_t>(W) * H;
  for (size_t i = 0; i < n; ++i)
    if (cells[i] == traits.free_value) out.push_back(static_cast<uint32_t>(i));
}

}  // namespace mcl
