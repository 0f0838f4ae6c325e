// exp_field_tie_order.cpp — the experiment VERDICT r04 item 5 asks for (f1): how much of the reference's
// nearest_obstacle_distance_map (distance_map.hpp:55-98) depends on the ORDER in which std::priority_queue pops equal keys?
//
//   variant 0  the reference's own std::priority_queue (what map_build.cpp runs: the product's field)
//   variant 1  FIFO bucket queue keyed on the float key (equal keys pop in push order)
//   variant 2  LIFO bucket queue (equal keys pop in reverse push order)
//   variant 3  equal keys pop in order of the CELL INDEX (a rule a parallel wavefront could implement: no history)
//   variant 4  equal keys pop in a seeded random order
//
// Output: the squared-distance map (float, the state before overlay / exp), so that the caller counts differing cells.
// Test infrastructure only (built by tools/exp_field_tie_order.py); nothing in the product links it.
#include <algorithm>
#include <cstdint>
#include <deque>
#include <map>
#include <queue>
#include <random>
#include <vector>

namespace {

struct Entry {
  uint32_t nearest_obstacle, index;
};

inline float squared_distance(uint32_t W, double res, uint32_t a, uint32_t b) {
  const double ax = (static_cast<double>(static_cast<int>(a % W)) + 0.5) * res;
  const double ay = (static_cast<double>(static_cast<int>(a / W)) + 0.5) * res;
  const double bx = (static_cast<double>(static_cast<int>(b % W)) + 0.5) * res;
  const double by = (static_cast<double>(static_cast<int>(b / W)) + 0.5) * res;
  const double dx = ax - bx, dy = ay - by;
  return static_cast<float>(dx * dx + dy * dy);
}

template <class Queue>
void wavefront(Queue& q, const uint8_t* seeds, uint32_t W, uint32_t H, double res, float max_sq, float* dist,
               uint64_t* non_monotone_pops) {
  const size_t n = static_cast<size_t>(W) * H;
  std::vector<bool> visited(n, false);
  for (size_t i = 0; i < n; ++i) {
    dist[i] = seeds[i] ? 0.f : max_sq;
    if (seeds[i]) {
      visited[i] = true;
      q.push(Entry{static_cast<uint32_t>(i), static_cast<uint32_t>(i)}, 0.f);
    }
  }
  float last = 0.f;
  uint64_t back = 0;
  while (!q.empty()) {
    const Entry parent = q.pop();
    const float key = dist[parent.index];
    if (key < last) ++back;
    last = key;
    const uint32_t xi = parent.index % W, yi = parent.index / W;
    auto relax = [&](size_t index) {
      if (visited[index]) return;
      visited[index] = true;
      const float d = squared_distance(W, res, parent.nearest_obstacle, static_cast<uint32_t>(index));
      if (d < max_sq) {
        dist[index] = d;
        q.push(Entry{parent.nearest_obstacle, static_cast<uint32_t>(index)}, d);
      }
    };
    if (xi + 1 < W) relax(static_cast<size_t>(parent.index) + 1);
    if (yi + 1 < H) relax(static_cast<size_t>(parent.index) + W);
    if (xi > 0) relax(static_cast<size_t>(parent.index) - 1);
    if (yi > 0) relax(static_cast<size_t>(parent.index) - W);
  }
  if (non_monotone_pops) *non_monotone_pops = back;
}

struct StdHeap {  // variant 0: the reference's queue, comparison through the distance map as in distance_map.hpp:70-73
  const float* dist;
  struct Cmp {
    const float* d;
    bool operator()(const Entry& a, const Entry& b) const { return d[a.index] > d[b.index]; }
  };
  std::priority_queue<Entry, std::vector<Entry>, Cmp> q;
  explicit StdHeap(const float* d) : dist(d), q(Cmp{d}) {}
  void push(const Entry& e, float) { q.push(e); }
  Entry pop() {
    Entry e = q.top();
    q.pop();
    return e;
  }
  bool empty() const { return q.empty(); }
};

template <bool Lifo>
struct Buckets {  // variants 1 / 2
  std::map<float, std::deque<Entry>> b;
  void push(const Entry& e, float key) { b[key].push_back(e); }
  Entry pop() {
    auto it = b.begin();
    Entry e;
    if (Lifo) {
      e = it->second.back();
      it->second.pop_back();
    } else {
      e = it->second.front();
      it->second.pop_front();
    }
    if (it->second.empty()) b.erase(it);
    return e;
  }
  bool empty() const { return b.empty(); }
};

struct Keyed {  // variants 3 / 4: (key, tie) lexicographic, tie = cell index or a random word
  struct Item {
    float key;
    uint64_t tie;
    Entry e;
  };
  struct Cmp {
    bool operator()(const Item& a, const Item& b) const { return a.key > b.key || (a.key == b.key && a.tie > b.tie); }
  };
  std::priority_queue<Item, std::vector<Item>, Cmp> q;
  std::mt19937_64 rng;
  bool random;
  Keyed(bool random_ties, uint64_t seed) : rng(seed), random(random_ties) {}
  void push(const Entry& e, float key) { q.push(Item{key, random ? rng() : e.index, e}); }
  Entry pop() {
    Entry e = q.top().e;
    q.pop();
    return e;
  }
  bool empty() const { return q.empty(); }
};

}  // namespace

extern "C" int field_tie_variant(const uint8_t* seeds, uint32_t W, uint32_t H, double res, double max_obstacle_distance,
                                 int variant, uint64_t seed, float* dist, uint64_t* non_monotone_pops) {
  const float max_sq = static_cast<float>(max_obstacle_distance * max_obstacle_distance);
  switch (variant) {
    case 0: {
      StdHeap q(dist);
      wavefront(q, seeds, W, H, res, max_sq, dist, non_monotone_pops);
      return 0;
    }
    case 1: {
      Buckets<false> q;
      wavefront(q, seeds, W, H, res, max_sq, dist, non_monotone_pops);
      return 0;
    }
    case 2: {
      Buckets<true> q;
      wavefront(q, seeds, W, H, res, max_sq, dist, non_monotone_pops);
      return 0;
    }
    case 3: {
      Keyed q(false, 0);
      wavefront(q, seeds, W, H, res, max_sq, dist, non_monotone_pops);
      return 0;
    }
    case 4: {
      Keyed q(true, seed);
      wavefront(q, seeds, W, H, res, max_sq, dist, non_monotone_pops);
      return 0;
    }
  }
  return -1;
}
