// beam_kernels.hip - the BeamSensorModel (sensor/beam_model.hpp, algorithm/raycasting/bresenham.hpp) for gfx950: the non-free bit maps of the
// occupancy grid, the per-beam tables and the two reweight kernels (a wave per particle for small sets; ordered lanes over an LDS bit
// window with closed-form skips of empty space for large ones), and their launchers.  Built with -ffp-contract=off like kernels.hip.
#include "device_common.hpp"

namespace mcl {
namespace {

// The pose of the particle at a position of the spatial order, moved into the grid's frame (kernels.hip has the same helper for the
// likelihood-field kernels).
__device__ __forceinline__ Pose2 ordered_pose(const Pose2& to_frame, const double4* __restrict__ pose, uint64_t i) {
  const double4 q = pose[i];
  return pose_mul(to_frame, Pose2{Rot2{q.x, q.y}, q.z, q.w});
}

// ---- K2' beam model ---------------------------------------------------------------------------------
// One wavefront per particle, one lane per beam; each lane walks its own Bresenham line on the int8 grid.
__device__ __forceinline__ void cell_near(const GridView& g, double px, double py, int& xi, int& yi) {
  const double inv = 1. / g.resolution;
  xi = static_cast<int>(floor(px * inv));
  yi = static_cast<int>(floor(py * inv));
}

// Ray2d::cast over Bresenham2i's standard variant (raycasting.hpp:78-107, bresenham.hpp:84-160): walk the integer line
// from the source cell towards the far-end cell, stop at the first cell that is outside the grid (no hit), non-free
// (hit) or past the end of the line (no hit).  The walk is evaluated kSpec cells at a time: the Bresenham state of the
// next cells does not depend on the grid, so their loads are issued together and examined in order — same cells, same
// order, same result, but the load latency of a step is no longer serialised behind the previous step's compare.
constexpr int kSpec = 8;
// The loop below is the same walk in a form that costs ~8 integer ops per cell instead of ~30:
//  * the line is expressed as a major step (every cell) and a minor step (when the error term trips), applied to
//    the linear cell index, so there is no per-cell axis swap and no multiply;
//  * "the trace ends at the first cell outside the grid" (take_while(contains)) is turned into a step count up front:
//    along the major axis the k-th cell is x0 + k*xstep; along the minor axis it is y0 + ystep*m_k with
//    m_k = ceil((xspan + k*dyspan) / dxspan) - 1 (the error term stays in (0, dxspan]), so the first k that leaves
//    the grid has a closed form.  Cells 0..last are then all inside and only their occupancy has to be read.
struct RayWalk {
  int sx, sy;
  bool steep;
  int major_span, major_step, minor_step, dmajor, dminor;
  int last;   // last cell index k of the trace that is inside the grid (-1: empty trace)
  int k, error, trips;
  int x, y;   // cell k
};

// floor(num / den) for num >= 0, den > 0: a 32-bit division whenever the operands allow it (they do for every grid below
// 2^16 cells per side), the 64-bit one (~4x the instructions) otherwise.
__device__ __forceinline__ long long floor_div(long long num, int den) {
  if (num < (1ll << 22)) {
    // the numerator is an exact float and the quotient stays below 2^22: with v_rcp_f32's one ulp and the two roundings
    // around it the product is off by less than one (~8 instructions instead of ~30 for the integer division)
    const int n = static_cast<int>(num);
    int q = static_cast<int>(static_cast<float>(n) * __builtin_amdgcn_rcpf(static_cast<float>(den)));
    const int rem = n - q * den;
    q += (rem >= den ? 1 : 0) - (rem < 0 ? 1 : 0);
    return q;
  }
  if (num < (1ll << 32)) return static_cast<long long>(static_cast<unsigned>(num) / static_cast<unsigned>(den));
  return num / den;
}

// Number of cells (minus one) of the walk that stay inside the box [lo_major, hi_major] x [lo_minor, hi_minor].
__device__ __forceinline__ int walk_room(const RayWalk& r, int major_pos, int minor_pos, int lo_major, int hi_major, int lo_minor,
                                         int hi_minor) {
  if (major_pos < lo_major || major_pos > hi_major || minor_pos < lo_minor || minor_pos > hi_minor) return -1;
  int last = r.major_span;
  last = min(last, r.major_step > 0 ? hi_major - major_pos : major_pos - lo_major);
  if (r.dminor > 0) {
    const long long room_minor = r.minor_step > 0 ? hi_minor - minor_pos : minor_pos - lo_minor;  // trips that stay inside
    // first k with m_k >= room_minor + 1  <=>  major_span + k*dminor > (room_minor + 1) * dmajor
    const long long k_exit = floor_div((room_minor + 1) * r.dmajor - r.major_span, r.dminor) + 1;
    if (k_exit - 1 < last) last = static_cast<int>(k_exit - 1);
  }
  return last;
}
__device__ __forceinline__ int walk_room_in_grid(const GridView& g, const RayWalk& r) {
  return walk_room(r, r.steep ? r.sy : r.sx, r.steep ? r.sx : r.sy, 0, static_cast<int>(r.steep ? g.H : g.W) - 1, 0,
                   static_cast<int>(r.steep ? g.W : g.H) - 1);
}

// kGridRoom == false leaves r.last unset (the caller bounds the walk itself and asks walk_room_in_grid only if it needs to).
template <bool kGridRoom = true>
__device__ __forceinline__ RayWalk walk_begin(const GridView& g, int sx, int sy, int fx, int fy) {
  RayWalk r;
  r.sx = sx;
  r.sy = sy;
  int xspan = fx - sx, xstep = 1;
  if (xspan < 0) {
    xspan = -xspan;
    xstep = -1;
  }
  int yspan = fy - sy, ystep = 1;
  if (yspan < 0) {
    yspan = -yspan;
    ystep = -1;
  }
  r.steep = xspan < yspan;  // bresenham.hpp:98-106 swaps the axes of a steep line
  r.major_span = r.steep ? yspan : xspan;
  const int minor_span = r.steep ? xspan : yspan;
  r.major_step = r.steep ? ystep : xstep;
  r.minor_step = r.steep ? xstep : ystep;
  r.dmajor = 2 * r.major_span;
  r.dminor = 2 * minor_span;
  r.last = kGridRoom ? walk_room_in_grid(g, r) : -1;
  r.k = 0;
  r.error = r.major_span;
  r.trips = 0;
  r.x = sx;
  r.y = sy;
  return r;
}

// Examines cells k .. upto (inclusive) kSpec at a time; `occupied(x, y)` says whether a cell is non-free.
// Returns true and leaves (x, y, k) at the hit; otherwise k = upto + 1 and the state is ready to continue.
template <class Fetch>
__device__ __forceinline__ bool walk_until(RayWalk& r, int upto, Fetch&& occupied) {
  const int mx = r.steep ? 0 : r.major_step, my = r.steep ? r.major_step : 0;
  const int nx = r.steep ? r.minor_step : 0, ny = r.steep ? 0 : r.minor_step;
  while (r.k <= upto) {
    int xs[kSpec], ys[kSpec];
    int err[kSpec], trp[kSpec];
#pragma unroll
    for (int u = 0; u < kSpec; ++u) {
      xs[u] = r.x;
      ys[u] = r.y;
      err[u] = r.error;
      trp[u] = r.trips;
      r.error += r.dminor;
      const bool trip = r.error > r.dmajor;
      r.x += mx + (trip ? nx : 0);
      r.y += my + (trip ? ny : 0);
      r.error -= trip ? r.dmajor : 0;
      r.trips += trip ? 1 : 0;
    }
    bool occ[kSpec];
#pragma unroll
    for (int u = 0; u < kSpec; ++u) occ[u] = (r.k + u <= upto) ? occupied(xs[u], ys[u]) : false;
#pragma unroll
    for (int u = 0; u < kSpec; ++u) {
      if (occ[u]) {
        r.x = xs[u];
        r.y = ys[u];
        r.k += u;
        return true;
      }
    }
    if (r.k + kSpec > upto + 1) {  // rewind the speculative overshoot so that a later phase continues at upto + 1
      const int keep = upto + 1 - r.k;  // 1 .. kSpec-1 cells of this group were real
      r.x = xs[keep];
      r.y = ys[keep];
      r.error = err[keep];
      r.trips = trp[keep];
      r.k = upto + 1;
      return false;
    }
    r.k += kSpec;
  }
  return false;
}

// kCell: instead of the range, the hit cell (x << 32 | y, both below 2^31) or kNoHitCell - the ordered beam kernel looks the
// range's terms up in a table over the squared cell distance (BeamTable).
constexpr long long kNoHitCell = -1;
__device__ __forceinline__ double walk_range(const GridView& g, const RayWalk& r, bool hit, double max_range, unsigned long long& steps);
template <bool kCell = false>
__device__ __forceinline__ auto walk_result(const GridView& g, const RayWalk& r, bool hit, double max_range, unsigned long long& steps) {
  if constexpr (kCell) {
    steps += static_cast<unsigned long long>((hit ? r.k : r.last) + 1);
    return hit ? ((static_cast<long long>(r.x) << 32) | static_cast<long long>(static_cast<uint32_t>(r.y))) : kNoHitCell;
  } else {
    return walk_range(g, r, hit, max_range, steps);
  }
}
__device__ __forceinline__ double walk_range(const GridView& g, const RayWalk& r, bool hit, double max_range, unsigned long long& steps) {
  if (!hit) {
    steps += static_cast<unsigned long long>(r.last + 1);
    return max_range;  // std::nullopt -> value_or(max_range)
  }
  steps += static_cast<unsigned long long>(r.k + 1);
  // cast(): distance between cell centres (raycasting.hpp:97-107)
  const double ax = (static_cast<double>(r.sx) + 0.5) * g.resolution, ay = (static_cast<double>(r.sy) + 0.5) * g.resolution;
  const double bx = (static_cast<double>(r.x) + 0.5) * g.resolution, by = (static_cast<double>(r.y) + 0.5) * g.resolution;
  const double dx = bx - ax, dy = by - ay;
  return fmin(sqrt(dx * dx + dy * dy), max_range);
}

__device__ __forceinline__ double cast_ray(const GridView& g, int sx, int sy, int fx, int fy, double max_range,
                                           unsigned long long& steps) {
  RayWalk r = walk_begin(g, sx, sy, fx, fy);
  const bool hit = walk_until(r, r.last, [&g](int x, int y) {
    return g.cells[static_cast<size_t>(y) * g.W + static_cast<size_t>(x)] != g.free_value;
  });
  return walk_result(g, r, hit, max_range, steps);
}

// The same cast with the occupancy of a kWin x kWin cell window around the workgroup's particles staged in LDS as one
// bit per cell (row stride kWinStride words: one word of padding keeps vertically adjacent cells on different banks).
// Cells of the trace beyond the window (long rays near its edge) are read from the global bit mask.
constexpr int kWin = 1024, kWinWords = kWin / 32, kWinStride = kWinWords + 1;
// Error trips after k steps of the walk: the error term stays in (0, dmajor], so m_k = ceil((major_span + k*dminor) / dmajor) - 1.
__device__ __forceinline__ int walk_trips_at(const RayWalk& r, int k) {
  if (r.dmajor == 0) return 0;
  return static_cast<int>(floor_div(r.major_span + static_cast<long long>(k) * r.dminor + r.dmajor - 1, r.dmajor)) - 1;
}
__device__ __forceinline__ void walk_seek(RayWalk& r, int k, int error) {
  const int trips = walk_trips_at(r, k);
  r.k = k;
  r.error = error;
  r.trips = trips;
  r.x = r.steep ? r.sx + r.minor_step * trips : r.sx + r.major_step * k;
  r.y = r.steep ? r.sy + r.major_step * k : r.sy + r.minor_step * trips;
}

// The window walk of cast_ray_window.  The map is mostly free space, so the walk asks a coarse bitmap — one bit per
// 8 x 8-cell block of the window, "any cell not free" — about a whole block column of the line at a time: 8 steps along the
// major axis move the minor coordinate by at most 8 cells, so the 8 cells lie in the blocks of the first cell and of the
// cell after the last one (a superset).  When both are empty the 8 cells are skipped; the Bresenham state after 8 steps
// is exact and division free: error' = error + (8 * dminor mod dmajor), one more trip if that exceeds dmajor.  Otherwise
// the 8 cells are examined one by one, exactly as a plain walk would.  Skipped cells are free by construction, so the
// first non-free cell, and with it Ray2d::cast's result (raycasting.hpp:97-107, bresenham.hpp:122-160), is unchanged bit
// for bit.  The coarse bitmap is stored twice, row-major and column-major, so that the block the line moves through
// along its minor axis is always a bit position inside one row of 128 bits.
// Step directions are template parameters (the lanes of a wave follow one beam from neighbouring poses and almost
// always share the line's octant), or kRuntimeStep to take them from the arguments.
constexpr int kRuntimeStep = 0x7FFFFFFF;
constexpr int kCoarse = kWin / 8, kCoarseWords = kCoarse / 32;  // 128 x 128 blocks, 4 words per row of blocks
// LDS of the ordered beam kernel: the bit window, its two coarse bitmaps, the block distance map
constexpr uint32_t kBeamCertified = 2048;  // beams with a "free ahead" entry (2 bytes each) behind the maps; 64 floats of scratch behind them
constexpr size_t kBeamLds = (static_cast<size_t>(kWin) * kWinStride + 2 * kCoarse * kCoarseWords) * sizeof(uint32_t) + kCoarse * kCoarse +
                            kBeamCertified * sizeof(uint16_t) + 64 * sizeof(float);

// Cells u = 0 .. count-1 (count <= 8) from (lx, ly, error): their words are fetched together (where they lie does not depend
// on what they hold) and examined in order.  Returns the index of the first non-free one or -1; `advance` also moves
// (lx, ly, error) count cells on.
// A bit-per-cell occupancy view the block walk reads: the LDS window (coordinates relative to its corner) or the whole grid in
// global memory.  rows / columns: the coarse bitmap (one bit per 8 x 8 block), row-major and column-major.
struct BlockMaps {
  const uint32_t* fine;
  int fine_stride;           // words per row of cells
  int x_max, y_max;          // last valid cell coordinates
  const uint32_t* rows;      // [ceil(height / 8)][row_words]
  const uint32_t* columns;   // [ceil(width / 8)][column_words]
  int row_words, column_words;
  const uint8_t* dist;       // [block row][dist_stride]: Chebyshev distance, in blocks, to the nearest block with a non-free cell
  int dist_stride;           // (0 = this one), capped at kDistCap
};
struct BitWindow {
  const uint32_t* lds;      // kWin rows x kWinStride words, then the two coarse bitmaps of the window
  int x0, y0;               // grid cell of window bit (0, 0); x0 is a multiple of 32, y0 of 8
  BlockMaps grid_maps;      // the whole grid (cells beyond the window)
};
// Chebyshev distance, in blocks, from block (bx, by) to the nearest block with a bit in the row-major coarse bitmap (rows of
// row_words words, block_rows of them; blocks beyond the bitmap count as empty), capped at kDistCap: per row within reach, the
// bits around the block's column.
constexpr int kDistCap = 17;  // block distances 0 .. 17 (17 = nothing within 16 blocks)
__device__ __forceinline__ int block_distance(const uint32_t* coarse_rows, int row_words, int block_rows, int bx, int by) {
  int best = kDistCap;
#pragma unroll 1
  for (int dy = -(kDistCap - 1); dy <= kDistCap - 1; ++dy) {
    const int row = by + dy;
    if (row < 0 || row >= block_rows) continue;
    const int ady = dy < 0 ? -dy : dy;
    if (ady >= best) continue;
    const uint32_t* rw = coarse_rows + static_cast<size_t>(row) * row_words;
    const int first = bx - (kDistCap - 1);  // may be negative
    const int word = first >> 5;            // floor
    const uint64_t lo = (word >= 0 && word < row_words) ? rw[word] : 0u;
    const uint64_t hi = (word + 1 >= 0 && word + 1 < row_words) ? rw[word + 1] : 0u;
    // the 2 kDistCap - 1 = 33 bits around the column (the shifted pair holds at least 33: 64 - 31)
    const uint64_t around = (((hi << 32) | lo) >> (first & 31)) & ((1ull << (2 * kDistCap - 1)) - 1);  // bit kDistCap - 1 = this column
    if (around == 0u) continue;
    const uint32_t right = static_cast<uint32_t>(around >> (kDistCap - 1));              // bit j = j columns to the right (0 = this one)
    const uint32_t left = static_cast<uint32_t>(around) & ((1u << (kDistCap - 1)) - 1);  // top bit = one column to the left
    int across = kDistCap;
    if (right) across = __builtin_ctz(right);
    if (left) across = min(across, (kDistCap - 1) - (31 - __builtin_clz(left)));
    best = min(best, max(ady, across));
  }
  return best;
}

template <bool kAdvance, bool kClamp = true>
__device__ __forceinline__ int examine_cells(const BlockMaps& maps, int& lx, int& ly, int& error, int count, int dminor, int dmajor,
                                             bool steep, int major_step, int minor_step) {
  int fx = lx, fy = ly, fe = error;
  uint32_t words[8];
  int bits[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    // cells behind the last one are not examined; they may lie outside (kClamp == false: the caller knows all 8 are inside)
    const int cx = kClamp ? min(max(fx, 0), maps.x_max) : fx, cy = kClamp ? min(max(fy, 0), maps.y_max) : fy;
    words[u] = maps.fine[cy * maps.fine_stride + (cx >> 5)];
    bits[u] = cx & 31;
    if (kAdvance && u == count) {
      lx = fx;
      ly = fy;
      error = fe;
    }
    fe += dminor;
    const bool trip = fe > dmajor;
    fe -= trip ? dmajor : 0;
    if (steep) {
      fy += major_step;
      fx += trip ? minor_step : 0;
    } else {
      fx += major_step;
      fy += trip ? minor_step : 0;
    }
  }
  if (kAdvance && count == 8) {
    lx = fx;
    ly = fy;
    error = fe;
  }
  int first = -1;
#pragma unroll
  for (int u = 7; u >= 0; --u) first = (u < count && ((words[u] >> bits[u]) & 1u)) ? u : first;  // the smallest u with a set bit wins
  return first;
}

// The 8 cells of one whole block column of the line (all of them inside the map), major axis known at compile time: the
// same cells in the same order as examine_cells, for about half the instructions.  Along x (kSteep == false) the cells share
// one word column of the bit map and the row moves when the error term trips; along y every cell is a row further and the
// bit moves on a trip.  v_bfe_u32 takes the bit position modulo 32 by itself.  Returns the first non-free cell or -1.
template <bool kSteep>
__device__ __forceinline__ int examine_column(const BlockMaps& maps, int major, int minor, int error, int dminor, int dmajor, int major_step,
                                              int minor_step) {
  uint32_t hits = 0;  // bit u: cell u is not free
  int fe = error;
  if (!kSteep) {
    const uint32_t* p = maps.fine + minor * maps.fine_stride + (major >> 5);
    const int row_step = minor_step * maps.fine_stride;
    int x = major;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      hits |= __builtin_amdgcn_ubfe(*p, static_cast<uint32_t>(x), 1u) << u;
      x += major_step;
      fe += dminor;
      const bool trip = fe > dmajor;
      fe -= trip ? dmajor : 0;
      p += trip ? row_step : 0;
    }
  } else {
    const uint32_t* row = maps.fine + major * maps.fine_stride;
    const int row_step = major_step * maps.fine_stride;
    int x = minor;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      hits |= __builtin_amdgcn_ubfe(row[x >> 5], static_cast<uint32_t>(x), 1u) << u;
      row += row_step;
      fe += dminor;
      const bool trip = fe > dmajor;
      fe -= trip ? dmajor : 0;
      x += trip ? minor_step : 0;
    }
  }
  return hits ? __builtin_ctz(hits) : -1;
}

// j cells along a stretch known to be free, in closed form: error + j dminor brought back into (0, dmajor], one trip per dmajor taken off (a
// float quotient and a +-1 correction: the operands are far below 2^24).
__device__ __forceinline__ void walk_advance_free(int j, int dminor, int dmajor, float inv_dmajor, bool steep, int major_step, int minor_step,
                                                  int& lx, int& ly, int& error) {
  // (values selected, not `if (steep) lx += ...; else ly += ...`: with a divergent `steep` the compiler turned those into ONE store through
  // a selected address, and the walk's position lived in scratch memory - a store, a dependent load and their waits per beam)
  int along_minor = 0;
  if (dmajor > 0) {
    const int total = error + j * dminor;
    int trips = static_cast<int>(static_cast<float>(total - 1) * inv_dmajor);
    int rem = total - trips * dmajor;
    const int up = rem > dmajor ? 1 : 0, down = rem <= 0 ? 1 : 0;
    trips += up - down;
    rem -= (up - down) * dmajor;
    error = rem;
    along_minor = trips * minor_step;
  }
  const int along_major = j * major_step;
  const int nx = lx + (steep ? along_minor : along_major), ny = ly + (steep ? along_major : along_minor);
  lx = nx;
  ly = ny;
}
// k_start: that many cells from the current one on are known to be free (the ordered beam kernel's per-beam certificate, see
// k_reweight_beam_sorted): they are passed in one closed-form step.
template <int STEEP, int MAJ, int MIN>
__device__ __forceinline__ void walk_blocks(const BlockMaps& maps, int& lx, int& ly, int& error, int& k, int& hit_k, int upto, int dminor,
                                            int dmajor, bool r_steep, int r_major_step, int r_minor_step, int k_start = 0) {
  const bool steep = STEEP == kRuntimeStep ? r_steep : (STEEP != 0);
  const int major_step = MAJ == kRuntimeStep ? r_major_step : MAJ, minor_step = MIN == kRuntimeStep ? r_minor_step : MIN;
  if (k > upto) return;
  // Up to 8 cells that stay inside one block column: with nothing in that block or around it (block distance >= 2) they are
  // free, and the state behind them has a closed form - error + j dminor brought back into (0, dmajor], one trip per dmajor
  // taken off (a float quotient and a +-1 correction: the operands are far below 2^24).
  const float inv_dmajor = dmajor > 0 ? __builtin_amdgcn_rcpf(static_cast<float>(dmajor)) : 0.f;  // (quotients below 2^8: one ulp is plenty)
  const bool closed_forms = dmajor < (1 << 16);  // (lines of 32K cells and more walk block column by block column)
  // (plain functions, not lambdas that capture by reference: the closure object of such a lambda - seven pointers to the walk's position,
  // error term and constants - outlives the inlining as a dead store, the variables it points to count as escaped and live in scratch
  // memory through the loops below: 18 stack slots and their loads and stores in every iteration)
  auto clear_ahead = [closed_forms, &maps](int cx, int cy) __attribute__((always_inline)) {
    return closed_forms && maps.dist[(cy >> 3) * maps.dist_stride + (cx >> 3)] >= 2;
  };
#define advance_free(j) walk_advance_free((j), dminor, dmajor, inv_dmajor, steep, major_step, minor_step, lx, ly, error)
  if (k_start > 0 && closed_forms) {
    const int j = min(k_start, upto - k + 1);
    advance_free(j);
    k += j;
    if (k > upto) return;
  }
  // 1. up to the end of the first block column
  {
    const int major = steep ? ly : lx;
    int j = major_step > 0 ? 8 - (major & 7) : (major & 7) + 1;
    j = min(j, upto - k + 1);
    if (clear_ahead(lx, ly)) {
      advance_free(j);
    } else {
      const int first = examine_cells<true>(maps, lx, ly, error, j, dminor, dmajor, steep, major_step, minor_step);
      if (first >= 0) {
        hit_k = k + first;
        return;
      }
    }
    k += j;
  }
  // 2. whole block columns.  8 * dminor = trips8 * dmajor + rest8 (dminor <= dmajor: trips8 <= 8).
  int trips8 = 0;
  if (dmajor > 0) {
    const int x = 8 * dminor;  // < 2^24: one float multiply and a +-1 correction
    int q = static_cast<int>(static_cast<float>(x) * inv_dmajor);
    const int rem = x - q * dmajor;
    q += (rem >= dmajor ? 1 : 0) - (rem < 0 ? 1 : 0);
    trips8 = q;
  }
  const int rest8 = 8 * dminor - trips8 * dmajor;
  const int minor8 = trips8 * minor_step;
  // rows of bits indexed by the block along the major axis, bit = block along the minor axis
  const uint32_t* bitmap = steep ? maps.rows : maps.columns;
  const int bitmap_words = steep ? maps.row_words : maps.column_words;
  int major = steep ? ly : lx, minor = steep ? lx : ly;
  while (k + 8 <= upto) {  // cells k .. k+7 and the cell behind them are inside
    {
      // Empty space in closed form: d = block distance from the block of cell k to the nearest block holding anything.  The next
      // 8 s cells stay within s blocks of it along either axis (8 s - 1 steps along the major axis, at most as many along the
      // minor one), so with d >= s + 1 they are all free: s = d - 1, or what is left of the walk, and the state after 8 s steps
      // is error + 8 s dminor brought back into (0, dmajor] (advance_free's rule; dmajor > 0 here: the line has more than 8
      // cells).  Same cells skipped as a cell-by-cell walk would have found free.
      const uint32_t d = maps.dist[steep ? (major >> 3) * maps.dist_stride + (minor >> 3) : (minor >> 3) * maps.dist_stride + (major >> 3)];
      if (d >= 2u && closed_forms) {
        const uint32_t room = static_cast<uint32_t>(upto - k) >> 3;  // >= 1
        const int s_columns = static_cast<int>(min(d - 1u, room));
        const int cells = 8 * s_columns;
        const int total = error + cells * dminor;  // <= 129 dmajor, far below 2^24
        int trips = static_cast<int>(static_cast<float>(total - 1) * inv_dmajor);
        int rem = total - trips * dmajor;
        const int up = rem > dmajor ? 1 : 0, down = rem <= 0 ? 1 : 0;
        trips += up - down;
        rem -= (up - down) * dmajor;
        k += cells;
        major += cells * major_step;
        minor += trips * minor_step;
        error = rem;
        continue;
      }
    }
    int raised = error + rest8;
    const bool extra = raised > dmajor;
    raised -= extra ? dmajor : 0;
    const int minor_next = minor + minor8 + (extra ? minor_step : 0);
    const uint32_t* row = bitmap + (major >> 3) * bitmap_words;
    const int b0 = minor >> 3, b1 = minor_next >> 3;
    const uint32_t occupied = ((row[b0 >> 5] >> (b0 & 31)) | (row[b1 >> 5] >> (b1 & 31))) & 1u;
    if (occupied) {
      int first;
      if (STEEP == kRuntimeStep) {
        int fx = steep ? minor : major, fy = steep ? major : minor, fe = error;
        first = examine_cells<false, false>(maps, fx, fy, fe, 8, dminor, dmajor, steep, major_step, minor_step);
      } else {
        first = examine_column<(STEEP != 0 && STEEP != kRuntimeStep)>(maps, major, minor, error, dminor, dmajor, major_step, minor_step);
      }
      if (first >= 0) {
        hit_k = k + first;
        return;
      }
    }
    k += 8;
    major += 8 * major_step;
    minor = minor_next;
    error = raised;
  }
  lx = steep ? minor : major;
  ly = steep ? major : minor;
  // 3. the last cells (fewer than a block column)
  while (k <= upto) {
    const int j = min(8, upto - k + 1);
    if (clear_ahead(lx, ly)) {
      advance_free(j);
    } else {
      const int first = examine_cells<true>(maps, lx, ly, error, j, dminor, dmajor, steep, major_step, minor_step);
      if (first >= 0) {
        hit_k = k + first;
        return;
      }
    }
    k += j;
  }
}
#undef advance_free
// Dispatch on the line's orientation: a wave whose lanes agree on it (they follow one beam from neighbouring poses) runs the
// instance with a compile-time major axis; the step directions stay run-time values (one instance per octant made the kernel
// outgrow the instruction cache).
__device__ __forceinline__ void walk_blocks_any(const BlockMaps& maps, const RayWalk& r, int& lx, int& ly, int& error, int& k, int& hit_k,
                                                int upto, int k_start = 0) {
  const int wave_steep = __builtin_amdgcn_readfirstlane(r.steep ? 1 : 0);
  if (__builtin_amdgcn_ballot_w64((r.steep ? 1 : 0) != wave_steep) == 0) {
    if (wave_steep) walk_blocks<1, kRuntimeStep, kRuntimeStep>(maps, lx, ly, error, k, hit_k, upto, r.dminor, r.dmajor, true, r.major_step, r.minor_step, k_start);
    else walk_blocks<0, kRuntimeStep, kRuntimeStep>(maps, lx, ly, error, k, hit_k, upto, r.dminor, r.dmajor, false, r.major_step, r.minor_step, k_start);
  } else {
    walk_blocks<kRuntimeStep, kRuntimeStep, kRuntimeStep>(maps, lx, ly, error, k, hit_k, upto, r.dminor, r.dmajor, r.steep, r.major_step,
                                                          r.minor_step, k_start);
  }
}

// free_ahead: cells of Euclidean distance from the source up to which every cell of this lane's trace is known to be free (0: nothing known)
template <bool kCell = false>
__device__ __forceinline__ auto cast_ray_window(const GridView& g, const BitWindow& w, int sx, int sy, int fx, int fy,
                                                double max_range, unsigned long long& steps, float free_ahead = 0.f) {
  RayWalk r = walk_begin<false>(g, sx, sy, fx, fy);
  // Cells inside the grid AND the window (a box): one closed-form bound.  The grid's own bound is only needed by a ray that
  // leaves the window without a hit.
  const int gw = static_cast<int>(g.W) - 1, gh = static_cast<int>(g.H) - 1;
  const int x_lo = max(w.x0, 0), x_hi = min(w.x0 + kWin - 1, gw), y_lo = max(w.y0, 0), y_hi = min(w.y0 + kWin - 1, gh);
  const int upto = walk_room(r, r.steep ? sy : sx, r.steep ? sx : sy, r.steep ? y_lo : x_lo, r.steep ? y_hi : x_hi, r.steep ? x_lo : y_lo,
                             r.steep ? x_hi : y_hi);
  int k = 0, hit_k = -1, error = r.error;
  if (upto >= 0) {
    // Cells 0 .. upto are inside the grid and the window: walked in LDS, block column by block column.
    int lx = sx - w.x0, ly = sy - w.y0;
    const uint32_t* rows = w.lds + kWin * kWinStride;
    const BlockMaps lds_maps{w.lds, kWinStride, kWin - 1, kWin - 1, rows, rows + kCoarse * kCoarseWords, kCoarseWords, kCoarseWords,
                             reinterpret_cast<const uint8_t*>(rows + 2 * kCoarse * kCoarseWords), kCoarse};
    // cell k of the trace lies k * |line| / major_span from the source (within a cell): the cells below free_ahead, less two for the roundings
    int k_start = 0;
    if (free_ahead > 0.f) {
      const float dxs = static_cast<float>(fx - sx), dys = static_cast<float>(fy - sy);
      const float length = sqrtf(dxs * dxs + dys * dys);
      k_start = length > 0.f ? static_cast<int>(free_ahead * static_cast<float>(r.major_span) * __builtin_amdgcn_rcpf(length) * 0.999f) - 2 : 0;
    }
    walk_blocks_any(lds_maps, r, lx, ly, error, k, hit_k, upto, k_start);
    if (hit_k >= 0) {
      walk_seek(r, hit_k, 0);
      return walk_result<kCell>(g, r, true, max_range, steps);
    }
  }
  // Cells beyond the window (long rays near its edge): the same walk over the whole-grid maps in global memory.
  r.last = walk_room_in_grid(g, r);
  if (k <= r.last) {
    walk_seek(r, k, error);
    int gx = r.x, gy = r.y;
    walk_blocks_any(w.grid_maps, r, gx, gy, error, k, hit_k, r.last);
    if (hit_k >= 0) {
      walk_seek(r, hit_k, 0);
      return walk_result<kCell>(g, r, true, max_range, steps);
    }
  }
  return walk_result<kCell>(g, r, false, max_range, steps);
}

// The cast over the whole-grid maps alone (no LDS window): small sets, one wave per particle.
template <bool kCell = false>
__device__ __forceinline__ auto cast_ray_grid(const GridView& g, const BlockMaps& grid_maps, int sx, int sy, int fx, int fy, double max_range,
                                              unsigned long long& steps) {
  RayWalk r = walk_begin(g, sx, sy, fx, fy);
  int k = 0, hit_k = -1, error = r.error;
  if (r.last >= 0) {
    int gx = sx, gy = sy;
    walk_blocks_any(grid_maps, r, gx, gy, error, k, hit_k, r.last);
    if (hit_k >= 0) {
      walk_seek(r, hit_k, 0);
      return walk_result<kCell>(g, r, true, max_range, steps);
    }
  }
  return walk_result<kCell>(g, r, false, max_range, steps);
}

// What beam_model.hpp:110-147 computes from the scan point alone (the same for every particle): the measured range, the
// far end of the trace in the sensor frame (raycasting.hpp:78-88: bearing * max_range), and the terms of the mixture that do
// not depend on the expected range.  The ordered kernel reads them from a table (k_beam_points), one entry per beam.
struct BeamPoint {
  double z;         // |p|
  double ux, uy;    // p / |p| * beam_max_range
  double short_e;   // exp(-lambda_short * z)
  double tail;      // z < beam_max_range ? z_rand / beam_max_range : z_max
};
__device__ __forceinline__ BeamPoint beam_point(const BeamModel& m, double px, double py) {
  BeamPoint q;
  q.z = sqrt(px * px + py * py);
  const double bc = px / q.z, bs = py / q.z;
  q.ux = bc * m.beam_max_range;
  q.uy = bs * m.beam_max_range;
  q.short_e = exp(-m.lambda_short * q.z);
  q.tail = q.z < m.beam_max_range ? m.z_rand / m.beam_max_range : m.z_max;
  return q;
}
__global__ __launch_bounds__(kBlock) void k_beam_points(const double* __restrict__ pts, uint32_t B, BeamModel m, BeamPoint* __restrict__ out) {
  const uint32_t b = blockIdx.x * kBlock + threadIdx.x;
  if (b < B) out[b] = beam_point(m, pts[2 * b], pts[2 * b + 1]);
}

// One beam of beam_model.hpp:110-147 for a source pose already in the grid frame (Ray2d ctor: raycasting.hpp:62-70).
// erf saturates: for |x| >= 6.5 it is +-1 to the last bit (erfc(6.5) < 4e-20), so when the expected range is more than
// 6.5 * sqrt(2) * sigma_hit away from both 0 and max_range — nearly every beam — the normaliser eta_hit is exactly 2 / 2
// and neither erf is evaluated (a wave takes the general path only if one of its lanes needs it).
// What depends on the expected range alone, tabulated over the squared cell distance r2 between the hit and the source (an
// integer the walk produces anyway): {z = min(sqrt(r2) * resolution, max_range), z_hit * eta_hit(z) * norm_hit,
// z_short * lambda_short * eta_short(z), -}, built on the device at mcl_set_map by the expressions of beam_term below (same
// functions, same order of operations); the last entry is the ray that hits nothing (z = max_range).  The table's z is the
// distance of the cell centres up to the rounding of (x + 0.5) * resolution - a relative 1e-16 - which the mixture's terms do not
// feel (the parity bar of the beam model's weights is 1e-10).  Per beam it replaces a square root, the short-return normaliser's
// exp and division and, near the ends of the range, two erf by one 32-byte look-up.
struct BeamTable {
  const double4* entries;  // nullptr: no table (max_range spans more than kBeamTableMaxCells cells)
  uint32_t no_hit;         // index of the last entry
};
__device__ __forceinline__ double4 beam_table_entry(const BeamModel& m, double resolution, double norm_hit, uint32_t r2, bool no_hit) {
  const double z_mean = no_hit ? m.beam_max_range : fmin(sqrt(static_cast<double>(r2)) * resolution, m.beam_max_range);
  const double scale = sqrt(2.) * m.sigma_hit;
  const double hi = (m.beam_max_range - z_mean) / scale, lo = -z_mean / scale;
  const bool saturated = m.beam_max_range - z_mean >= 6.6 * scale && z_mean >= 6.6 * scale;  // erf is +-1 to the last bit there
  const double eta_hit = saturated ? 1.0 : 2. / (erf(hi) - erf(lo));
  const double eta_short = 1. / (1. - exp(-m.lambda_short * z_mean));
  return double4{z_mean, m.z_hit * eta_hit * norm_hit, m.z_short * m.lambda_short * eta_short, 0.0};
}
__global__ __launch_bounds__(kBlock) void k_beam_table(BeamModel m, double resolution, uint32_t entries, double4* __restrict__ out) {
  const uint32_t r2 = blockIdx.x * kBlock + threadIdx.x;
  if (r2 >= entries) return;
  const double norm_hit = 1. / (sqrt(2. * kPi) * m.sigma_hit);
  out[r2] = beam_table_entry(m, resolution, norm_hit, r2, r2 == entries - 1);
}

// kTable: `cast` returns the hit cell (walk_result<true>) instead of the range; (sx, sy) is the source cell.
template <bool kTable, class Cast>
__device__ __forceinline__ double beam_term(const GridView& g, const BeamModel& m, double norm_hit, const Pose2& src, const BeamPoint& q,
                                            const BeamTable& table, int sx, int sy, Cast&& cast) {
  double ex, ey;  // trace(): raycasting.hpp:78-88
  rot_apply(src.r, q.ux, q.uy, ex, ey);
  ex += src.x;
  ey += src.y;
  int fx, fy;
  cell_near(g, ex, ey, fx, fy);
  if constexpr (kTable) {
    const long long hit = cast(fx, fy);
    const int hx = static_cast<int>(hit >> 32), hy = static_cast<int>(static_cast<uint32_t>(hit));
    uint32_t at = table.no_hit;
    if (hit != kNoHitCell) {
      const int cx = hx - sx, cy = hy - sy;
      const uint32_t r2 = static_cast<uint32_t>(cx * cx + cy * cy);
      at = r2 < table.no_hit ? r2 : table.no_hit;  // (beyond the table: at least max_range away - the same entry)
    }
    const double4 e = table.entries[at];
    double z_mean = e.x;
    // The one place where the last bits of the expected range decide something: `measured < expected` below.  A measured range
    // within 1e-9 m of the table's value takes the reference's own expression - the distance of the two cell centres
    // (raycasting.hpp:97-107) - so that a tie falls as it does there.
    const bool tie = hit != kNoHitCell && fabs(q.z - z_mean) < 1e-9;
    if (__builtin_amdgcn_ballot_w64(tie) != 0 && tie) {
      const double ax = (static_cast<double>(sx) + 0.5) * g.resolution, ay = (static_cast<double>(sy) + 0.5) * g.resolution;
      const double bx = (static_cast<double>(hx) + 0.5) * g.resolution, by = (static_cast<double>(hy) + 0.5) * g.resolution;
      const double dx = bx - ax, dy = by - ay;
      z_mean = fmin(sqrt(dx * dx + dy * dy), m.beam_max_range);
    }
    const double d = (q.z - z_mean) / m.sigma_hit;
    double pz = e.y * exp(-(d * d) / 2.);
    if (q.z < z_mean) pz += e.z * q.short_e;
    pz += q.tail;
    return pz * pz * pz;
  }
  const double z_mean = cast(fx, fy);
  const double scale = sqrt(2.) * m.sigma_hit;
  double eta_hit = 1.0;
  // (a margin of 6.6 scales, compared without the divisions, implies both arguments beyond 6.5)
  const bool saturated = m.beam_max_range - z_mean >= 6.6 * scale && z_mean >= 6.6 * scale;
  if (__builtin_amdgcn_ballot_w64(!saturated) != 0) {
    const double hi = (m.beam_max_range - z_mean) / scale, lo = -z_mean / scale;
    eta_hit = 2. / (erf(hi) - erf(lo));
  }
  const double d = (q.z - z_mean) / m.sigma_hit;
  double pz = m.z_hit * eta_hit * norm_hit * exp(-(d * d) / 2.);
  if (q.z < z_mean) {
    const double eta_short = 1. / (1. - exp(-m.lambda_short * z_mean));
    pz += m.z_short * m.lambda_short * eta_short * q.short_e;
  }
  pz += q.tail;
  return pz * pz * pz;
}

// Variant A: one wavefront per particle, one lane per beam (small particle sets).
__global__ __launch_bounds__(kBlock) void k_reweight_beam(Particles p, uint64_t n, GridView g, BeamModel m, NonFreeBits bits,
                                                          const double2* __restrict__ pts, uint32_t B, unsigned long long* d_steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double2* s_pts = reinterpret_cast<double2*>(smem);
  for (uint32_t i = threadIdx.x; i < B; i += kBlock) s_pts[i] = pts[i];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * (kBlock / kWave) + (threadIdx.x >> 6);
  if (i >= n) return;
  const Pose2 src = pose_mul(g.origin_inverse, load_pose(p, i));
  int sx, sy;
  cell_near(g, src.x, src.y, sx, sy);
  const double norm_hit = 1. / (sqrt(2. * kPi) * m.sigma_hit);
  const BlockMaps grid_maps{bits.fine, static_cast<int>(bits.words_per_row), static_cast<int>(g.W) - 1, static_cast<int>(g.H) - 1, bits.rows,
                            bits.columns, static_cast<int>(bits.row_words), static_cast<int>(bits.column_words), bits.dist,
                            static_cast<int>(bits.dist_stride)};
  double acc = 0.0;
  unsigned long long steps = 0;
  for (uint32_t b = lane; b < B; b += kWave) {
    const double2 pt = s_pts[b];
    acc += beam_term<false>(g, m, norm_hit, src, beam_point(m, pt.x, pt.y), BeamTable{nullptr, 0u}, sx, sy, [&](int fx, int fy) {
      return bits.fine ? cast_ray_grid(g, grid_maps, sx, sy, fx, fy, m.beam_max_range, steps) : cast_ray(g, sx, sy, fx, fy, m.beam_max_range, steps);
    });
  }
  const double total = wave_sum_f64(acc);
  if (d_steps) {
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_down(steps, o);
    if (lane == 0) atomicAdd(d_steps, steps);
  }
  if (lane == 0) p.w[i] = p.w[i] * total;
}

// Variant B (default above 16K particles): one lane per spatially ordered particle, every lane walks the same beam at
// the same time.  Neighbouring lanes trace nearly the same line and finish together; the sum is taken in scan order (the
// reference's std::transform_reduce leaves the order open).  Per-lane byte gathers top out at ~2 lanes/clk/CU on this chip (profiles/r01: the L1 handles a
// gather lane by lane even when the lanes share a line), so the occupancy the walks read is staged ONCE per workgroup
// into LDS as a 1024 x 1024-cell bit window (132 KB) centred on the workgroup's particles; LDS serves 32 lanes/clk.
constexpr int kBeamBlock = 1024;
template <bool kTable>
__global__ __launch_bounds__(kBeamBlock) void k_reweight_beam_sorted(double* __restrict__ w, uint64_t n, GridView g, BeamModel m,
                                                                     NonFreeBits bits, const BeamPoint* __restrict__ pts,
                                                                     uint32_t B, const uint32_t* __restrict__ perm,
                                                                     const double4* __restrict__ pose, unsigned long long* d_steps,
                                                                     double* __restrict__ partial, uint32_t beams_per_segment,
                                                                     BeamTable table, uint32_t free_ahead_on, uint32_t sectors_on) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* win = reinterpret_cast<uint32_t*>(smem);
  // (the workgroups take the blocks of the order from both ends inwards - 0, N - 1, 1, N - 2, ... -: the ends are the cloud's fringe, whose rays
  // run less alongside one another; see k_reweight_lf_patch)
  const uint32_t block = (blockIdx.x & 1u) ? gridDim.x - 1u - (blockIdx.x >> 1) : (blockIdx.x >> 1);
  const uint64_t t0 = static_cast<uint64_t>(block) * kBeamBlock;
  const uint64_t t = t0 + threadIdx.x;
  const uint64_t tt = t < n ? t : n - 1;
  // window around the middle particle of the workgroup (they are spatial neighbours after the ordering pass)
  const uint64_t tm = t0 + kBeamBlock / 2 < n ? t0 + kBeamBlock / 2 : n - 1;
  const Pose2 middle = ordered_pose(g.origin_inverse, pose, perm[tm]);
  int cx, cy;
  cell_near(g, middle.x, middle.y, cx, cy);
  BitWindow bw;
  bw.lds = win;
  bw.grid_maps = BlockMaps{bits.fine, static_cast<int>(bits.words_per_row), static_cast<int>(g.W) - 1, static_cast<int>(g.H) - 1, bits.rows,
                           bits.columns, static_cast<int>(bits.row_words), static_cast<int>(bits.column_words), bits.dist,
                           static_cast<int>(bits.dist_stride)};
  const uint32_t* nonfree_bits = bits.fine;
  const uint32_t words_per_row = bits.words_per_row;
  // The window at (bw.x0, bw.y0) - x0 a multiple of 32 cells, y0 of 8 - with its coarse bitmaps and block distances behind it.
  auto stage_window = [&]() {
    for (int i = threadIdx.x; i < kWin * kWinWords; i += kBeamBlock) {
      const int row = i >> 5, col = i & 31;
      const int gy = bw.y0 + row, gw = (bw.x0 >> 5) + col;
      uint32_t v = 0;
      if (gy >= 0 && gy < static_cast<int>(g.H) && gw >= 0 && gw < static_cast<int>(words_per_row))
        v = nonfree_bits[static_cast<size_t>(gy) * words_per_row + gw];
      win[row * kWinStride + col] = v;
    }
    __syncthreads();
    // coarse bitmap behind the window: bit (bx, by) = any cell of block (bx, by) not free
    uint32_t* coarse = win + kWin * kWinStride;
    for (int cw = threadIdx.x; cw < kCoarse * kCoarseWords; cw += kBeamBlock) {
      const int by = cw / kCoarseWords, quarter = cw % kCoarseWords;
      uint32_t bits = 0;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        uint32_t any = 0;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) any |= win[(by * 8 + rr) * kWinStride + quarter * 8 + f];
#pragma unroll
        for (int q = 0; q < 4; ++q) bits |= ((any >> (8 * q)) & 0xFFu) ? (1u << (f * 4 + q)) : 0u;
      }
      coarse[cw] = bits;
    }
    __syncthreads();
    // column-major copy: word (bx, q) holds blocks (bx, 32 q .. 32 q + 31)
    uint32_t* columns = coarse + kCoarse * kCoarseWords;
    for (int cw = threadIdx.x; cw < kCoarse * kCoarseWords; cw += kBeamBlock) {
      const int bx = cw / kCoarseWords, quarter = cw % kCoarseWords;
      uint32_t bits = 0;
#pragma unroll 8
      for (int q = 0; q < 32; ++q) bits |= ((coarse[(quarter * 32 + q) * kCoarseWords + (bx >> 5)] >> (bx & 31)) & 1u) << q;
      columns[cw] = bits;
    }
    // block distance map of the window: a copy of the whole grid's (the window's blocks are the grid's: x0 is a multiple of 32
    // cells, y0 of 8); blocks outside the grid are never entered
    uint8_t* dist = reinterpret_cast<uint8_t*>(columns + kCoarse * kCoarseWords);
    const int grid_block_columns = static_cast<int>((g.W + 7u) >> 3), grid_block_rows = static_cast<int>((g.H + 7u) >> 3);
    for (int blk = threadIdx.x; blk < kCoarse * kCoarse; blk += kBeamBlock) {
      const int gy = (bw.y0 >> 3) + blk / kCoarse, gx = (bw.x0 >> 3) + blk % kCoarse;
      const bool inside = gy >= 0 && gy < grid_block_rows && gx >= 0 && gx < grid_block_columns;
      dist[blk] = inside ? bits.dist[static_cast<size_t>(gy) * bits.dist_stride + gx] : static_cast<uint8_t>(kDistCap);
    }
    __syncthreads();
  };

  const uint32_t i = perm[tt];
  const Pose2 src = ordered_pose(g.origin_inverse, pose, i);  // Ray2d ctor: raycasting.hpp:69
  int sx, sy;
  cell_near(g, src.x, src.y, sx, sy);
  const double norm_hit = 1. / (sqrt(2. * kPi) * m.sigma_hit);
  double acc = 0.0;
  unsigned long long steps = 0;
  // partial != nullptr (sets of fewer workgroups than CUs): blockIdx.y takes a contiguous segment of the scan, the segment's sum
  // goes to partial[segment][t] and k_lf_combine adds the segments in order
  const uint32_t b_begin = partial ? blockIdx.y * beams_per_segment : 0u;
  const uint32_t b_end = partial ? (b_begin + beams_per_segment < B ? b_begin + beams_per_segment : B) : B;
  // "Free ahead", per beam and workgroup: the workgroup's 1024 poses are neighbours of the spatial order, so their traces of one
  // beam run side by side - a lane's point at distance t along its ray is within  D + t R  of the middle particle's point at the same
  // distance (D: the largest distance of a pose from the middle one, R: the largest |R_p - R_middle| = chord of the heading
  // difference), plus a cell for each of the roundings involved (cell centres for poses, the integer line for the ray, Bresenham's
  // half cell).  Where the middle particle's ray sits in a block whose Chebyshev distance c to the nearest block holding a non-free
  // cell (the block distance map) leaves that much room - every cell within 8 (c - 1) cells of any cell of the block is free -,
  // all the lanes' cells up to there are free: one thread per beam walks the middle ray block by block and leaves the distance, and
  // every lane passes its share of it in ONE closed-form step instead of five or six block-distance skips (which then only
  // serve the rest of the trace).  Free cells stay free: the first non-free cell, hence Ray2d::cast's result and the count of
  // cells visited (raycasting.hpp:97-107, bresenham.hpp:122-160), do not change.
  uint16_t* s_free_ahead = reinterpret_cast<uint16_t*>(smem + (static_cast<size_t>(kWin) * kWinStride + 2 * kCoarse * kCoarseWords) * sizeof(uint32_t) +
                                                       kCoarse * kCoarse);
  float* s_spread = reinterpret_cast<float*>(s_free_ahead + kBeamCertified);
  const float inv_res = static_cast<float>(1.0 / g.resolution);
  const float reach = static_cast<float>(m.beam_max_range) * inv_res;
  const bool per_beam_entries = b_end - b_begin <= kBeamCertified && reach < 4096.f;
  const bool certified = free_ahead_on != 0u && per_beam_entries;
  float D = 0.f, R = 0.f;
  if (certified) {
    float d_pos = sqrtf(static_cast<float>((src.x - middle.x) * (src.x - middle.x) + (src.y - middle.y) * (src.y - middle.y))) * inv_res;
    float d_rot = sqrtf(static_cast<float>((src.r.c - middle.r.c) * (src.r.c - middle.r.c) + (src.r.s - middle.r.s) * (src.r.s - middle.r.s)));
    if (!(d_pos < 1e6f && d_rot < 4.f)) d_pos = d_rot = INFINITY;  // (a non-finite pose: no certificate)
    for (int o = 32; o > 0; o >>= 1) {
      d_pos = fmaxf(d_pos, __shfl_xor(d_pos, o));
      d_rot = fmaxf(d_rot, __shfl_xor(d_rot, o));
    }
    if ((threadIdx.x & 63) == 0) {
      s_spread[2 * (threadIdx.x >> 6)] = d_pos;
      s_spread[2 * (threadIdx.x >> 6) + 1] = d_rot;
    }
    __syncthreads();
    for (int q = 0; q < kBeamBlock / 64; ++q) {
      D = fmaxf(D, s_spread[2 * q]);
      R = fmaxf(R, s_spread[2 * q + 1]);
    }
    D = D * 1.001f + 3.f;  // + the cells of the roundings
    R = R * 1.001f;
  }
  // One window centred on the workgroup's particles holds every ray of up to ~kWin / 2 cells.  Longer ones (a 30 m scanner on a 5 cm
  // grid reaches 600 cells) left it and went on over the whole-grid maps in global memory: a fifth of this kernel's time for three
  // wave-beams in ten.  They stay in workgroup memory if the scan is taken in FOUR SECTORS - the quadrant the middle particle's ray of a
  // beam points into -, each with a window of its own that has the particles near the corner the rays leave from (kWin - reach cells
  // of room, shared between the two sides): four stagings of the window per workgroup instead of one (each ~0.1 % of the workgroup's
  // time).  A lane's ray that still leaves its window (a pose far from the middle one) goes on over the whole-grid maps as before.
  // The sum over the beams is then taken sector by sector (the reference's std::transform_reduce leaves the order open).
  const uint32_t sectors = (sectors_on != 0u && per_beam_entries && reach > static_cast<float>(kWin / 2 - 64) && reach < static_cast<float>(kWin - 128)) ? 4u : 1u;
  const int room = sectors > 1 ? (kWin - static_cast<int>(reach)) / 2 : kWin / 2;
#pragma unroll 1
  for (uint32_t sector = 0; sector < sectors; ++sector) {
    // sector 0: rays towards +x +y, 1: -x +y, 2: -x -y, 3: +x -y
    const bool to_left = sector == 1 || sector == 2, down = sector >= 2;
    bw.x0 = ((sectors > 1 ? (to_left ? cx + room - (kWin - 1) : cx - room) : cx - kWin / 2) >> 5) << 5;
    bw.y0 = ((sectors > 1 ? (down ? cy + room - (kWin - 1) : cy - room) : cy - kWin / 2) >> 3) << 3;  // block rows start on multiples of 8 cells
    if (sector > 0) __syncthreads();  // every lane is done with the window before
    stage_window();
    if (certified || sectors > 1) {
      // s_free_ahead[beam]: bit 15 = the beam belongs to this sector, bits 0 .. 14 = its certificate in cells
      const uint8_t* dist = reinterpret_cast<const uint8_t*>(win + kWin * kWinStride + 2 * kCoarse * kCoarseWords);
      const float wx = static_cast<float>(middle.x) * inv_res - static_cast<float>(bw.x0), wy = static_cast<float>(middle.y) * inv_res - static_cast<float>(bw.y0);
      for (uint32_t b = b_begin + threadIdx.x; b < b_end; b += kBeamBlock) {
        const BeamPoint q = pts[b];
        double ex, ey;
        rot_apply(middle.r, q.ux, q.uy, ex, ey);
        const uint32_t its_sector = ex >= 0.0 ? (ey >= 0.0 ? 0u : 3u) : (ey >= 0.0 ? 1u : 2u);
        if (sectors > 1 && its_sector != sector) {
          s_free_ahead[b - b_begin] = 0;
          continue;
        }
        float t = 0.f;
        if (certified) {
          const float dx = static_cast<float>(ex / m.beam_max_range), dy = static_cast<float>(ey / m.beam_max_range);  // the middle ray's direction
#pragma unroll 1
          for (; t < reach; t += 8.f) {
            const float px = wx + t * dx, py = wy + t * dy;
            if (!(px >= 0.f && py >= 0.f && px < static_cast<float>(kWin) && py < static_cast<float>(kWin))) break;  // (NaN: no certificate)
            // (the window's distance map says nothing about blocks beyond the grid)
            if (!(px + static_cast<float>(bw.x0) >= 0.f && py + static_cast<float>(bw.y0) >= 0.f && px + static_cast<float>(bw.x0) < static_cast<float>(g.W) &&
                  py + static_cast<float>(bw.y0) < static_cast<float>(g.H)))
              break;
            const int c = dist[(static_cast<int>(py) >> 3) * kCoarse + (static_cast<int>(px) >> 3)];
            // the points of [t, t + 8) lie within 8 cells of this one
            if (!(D + (t + 8.f) * R + 8.f <= static_cast<float>(8 * (c - 1)))) break;
          }
        }
        s_free_ahead[b - b_begin] = static_cast<uint16_t>(0x8000u | static_cast<uint32_t>(t));
      }
      __syncthreads();
    }
    for (uint32_t b = b_begin; b < b_end; ++b) {
      float free_ahead = 0.f;
      if (certified || sectors > 1) {
        const uint32_t entry = s_free_ahead[b - b_begin];  // (the same for every lane: a scalar branch)
        if (!(entry & 0x8000u)) continue;
        free_ahead = static_cast<float>(entry & 0x7FFFu);
      }
      acc += beam_term<kTable>(g, m, norm_hit, src, pts[b], table, sx, sy, [&](int fx, int fy) {
        return cast_ray_window<kTable>(g, bw, sx, sy, fx, fy, m.beam_max_range, steps, free_ahead);
      });
    }
  }
  if (d_steps) {
    if (t >= n) steps = 0;
    for (int o = 32; o > 0; o >>= 1) steps += __shfl_down(steps, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(d_steps, steps);
  }
  if (t < n) {
    if (partial) partial[static_cast<size_t>(blockIdx.y) * n + t] = acc;
    else w[i] = w[i] * acc;
  }
}

// nonfree_bits: one bit per cell, row-major, words_per_row = ceil(W / 32) words per row.
__global__ __launch_bounds__(kBlock) void k_pack_nonfree(const int8_t* __restrict__ cells, uint32_t W, uint32_t H, int8_t free_value,
                                                         uint32_t words_per_row, uint32_t* __restrict__ bits) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= static_cast<uint64_t>(words_per_row) * H) return;
  const uint32_t y = static_cast<uint32_t>(i / words_per_row), wx = static_cast<uint32_t>(i % words_per_row);
  uint32_t v = 0;
  for (uint32_t b = 0; b < 32; ++b) {
    const uint32_t x = wx * 32 + b;
    if (x < W && cells[static_cast<size_t>(y) * W + x] != free_value) v |= 1u << b;
  }
  bits[i] = v;
}

// Coarse bitmaps of the whole grid: bit (bx, by) = any cell of the 8 x 8 block not free (cells beyond the grid count as free:
// the walks never go there).  rows[by][bx bits], columns[bx][by bits].
__global__ __launch_bounds__(kBlock) void k_pack_coarse_rows(const uint32_t* __restrict__ fine, uint32_t words_per_row, uint32_t H,
                                                             uint32_t block_rows, uint32_t row_words, uint32_t* __restrict__ rows) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= block_rows * row_words) return;
  const uint32_t by = i / row_words, q = i % row_words;
  uint32_t out = 0;
  for (uint32_t f = 0; f < 8; ++f) {
    const uint32_t word = q * 8 + f;
    uint32_t any = 0;
    if (word < words_per_row)
      for (uint32_t r = 0; r < 8; ++r)
        if (by * 8 + r < H) any |= fine[static_cast<size_t>(by * 8 + r) * words_per_row + word];
    for (uint32_t b = 0; b < 4; ++b) out |= ((any >> (8 * b)) & 0xFFu) ? (1u << (f * 4 + b)) : 0u;
  }
  rows[i] = out;
}
__global__ __launch_bounds__(kBlock) void k_pack_coarse_columns(const uint32_t* __restrict__ rows, uint32_t block_rows, uint32_t row_words,
                                                                uint32_t block_columns, uint32_t column_words,
                                                                uint32_t* __restrict__ columns) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= block_columns * column_words) return;
  const uint32_t bx = i / column_words, q = i % column_words;
  uint32_t out = 0;
  for (uint32_t b = 0; b < 32; ++b) {
    const uint32_t by = q * 32 + b;
    if (by < block_rows) out |= ((rows[static_cast<size_t>(by) * row_words + (bx >> 5)] >> (bx & 31)) & 1u) << b;
  }
  columns[i] = out;
}

// Block distance map of the whole grid (block_distance above), one byte per block, dist_stride bytes per row of blocks.
__global__ __launch_bounds__(kBlock) void k_block_distances(const uint32_t* __restrict__ rows, uint32_t block_rows, uint32_t row_words,
                                                            uint32_t block_columns, uint32_t dist_stride, uint8_t* __restrict__ dist) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= static_cast<uint64_t>(block_rows) * dist_stride) return;
  const uint32_t by = static_cast<uint32_t>(i / dist_stride), bx = static_cast<uint32_t>(i % dist_stride);
  dist[i] = bx < block_columns ? static_cast<uint8_t>(block_distance(rows, static_cast<int>(row_words), static_cast<int>(block_rows),
                                                                     static_cast<int>(bx), static_cast<int>(by)))
                               : static_cast<uint8_t>(kDistCap);
}

}  // namespace

NonFreeBits nonfree_layout(uint32_t W, uint32_t H, uint32_t* base) {
  NonFreeBits b{};
  b.words_per_row = (W + 31) / 32;
  const uint32_t block_columns = (W + 7) / 8, block_rows = (H + 7) / 8;
  b.row_words = (block_columns + 31) / 32;
  b.column_words = (block_rows + 31) / 32;
  b.fine = base;
  b.rows = base + static_cast<size_t>(b.words_per_row) * H;
  b.columns = b.rows + static_cast<size_t>(block_rows) * b.row_words;
  b.dist_stride = (block_columns + 3u) & ~3u;
  b.dist = reinterpret_cast<const uint8_t*>(b.columns + static_cast<size_t>(block_columns) * b.column_words);
  return b;
}
size_t nonfree_words(uint32_t W, uint32_t H) {
  const NonFreeBits b = nonfree_layout(W, H, nullptr);
  return static_cast<size_t>(b.columns - b.fine) + static_cast<size_t>((W + 7) / 8) * b.column_words +
         static_cast<size_t>((H + 7) / 8) * (b.dist_stride / 4);
}
void launch_pack_nonfree(hipStream_t st, const int8_t* cells, uint32_t W, uint32_t H, int8_t free_value, uint32_t* bits) {
  const NonFreeBits b = nonfree_layout(W, H, bits);
  const uint64_t words = static_cast<uint64_t>(b.words_per_row) * H;
  hipLaunchKernelGGL(k_pack_nonfree, dim3(blocks_for(words)), dim3(kBlock), 0, st, cells, W, H, free_value, b.words_per_row, bits);
  const uint32_t block_columns = (W + 7) / 8, block_rows = (H + 7) / 8;
  hipLaunchKernelGGL(k_pack_coarse_rows, dim3(blocks_for(static_cast<uint64_t>(block_rows) * b.row_words)), dim3(kBlock), 0, st, b.fine,
                     b.words_per_row, H, block_rows, b.row_words, const_cast<uint32_t*>(b.rows));
  hipLaunchKernelGGL(k_pack_coarse_columns, dim3(blocks_for(static_cast<uint64_t>(block_columns) * b.column_words)), dim3(kBlock), 0, st,
                     b.rows, block_rows, b.row_words, block_columns, b.column_words, const_cast<uint32_t*>(b.columns));
  hipLaunchKernelGGL(k_block_distances, dim3(blocks_for(static_cast<uint64_t>(block_rows) * b.dist_stride)), dim3(kBlock), 0, st, b.rows,
                     block_rows, b.row_words, block_columns, b.dist_stride, const_cast<uint8_t*>(b.dist));
}

// hipFuncSetAttribute is per device: contexts on several GPUs of one process each opt in (mcl_create calls this).
void configure_device_kernels() {
  const size_t lds = kBeamLds;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_reweight_beam_sorted<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_reweight_beam_sorted<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds));
}

uint32_t beam_table_entries(double beam_max_range, double resolution) {
  const double reach = std::ceil(beam_max_range / resolution) + 3.0;  // cells between the source and the far end of a trace, with slack
  if (!(reach > 0.0) || reach > kBeamTableMaxCells) return 0;
  const uint32_t r = static_cast<uint32_t>(reach);
  return r * r + 1u;  // squared distances below reach^2, + the entry of the ray that hits nothing
}
void launch_beam_table(hipStream_t st, BeamModel m, double resolution, uint32_t entries, double* table) {
  if (entries) hipLaunchKernelGGL(k_beam_table, dim3(blocks_for(entries)), dim3(kBlock), 0, st, m, resolution, entries, reinterpret_cast<double4*>(table));
}

void launch_reweight_beam(hipStream_t st, Particles p, uint64_t n, GridView g, BeamModel m, const double* d_points, uint32_t B,
                          unsigned long long* d_steps, const SortScratch* sorted, const uint32_t* nonfree_bits, double* d_beam_points,
                          const double* d_beam_table, uint32_t beam_table_count, bool free_ahead, bool sectors) {
  if (n == 0 || B == 0) return;
  if (sorted && nonfree_bits) {
    const size_t lds = kBeamLds;
    // One workgroup of 1024 particles per CU: below 256 workgroups the scan is split into segments (second grid dimension)
    // until the chip is covered twice, and the segment sums are added in a second pass - same terms, fixed association.
    const uint32_t groups = static_cast<uint32_t>((n + kBeamBlock - 1) / kBeamBlock);
    uint32_t segments = 1;
    if (sorted->partial && n < kLfSegmentedBelow && groups < 256) {
      segments = std::min<uint32_t>((512 + groups - 1) / groups, kLfMaxSegments);
      segments = std::max(1u, std::min(segments, B / 8));
    }
    const uint32_t per_segment = (B + segments - 1) / segments;
    segments = (B + per_segment - 1) / per_segment;
    double* partial = segments > 1 ? sorted->partial : nullptr;
    BeamPoint* table = reinterpret_cast<BeamPoint*>(d_beam_points);
    hipLaunchKernelGGL(k_beam_points, dim3(blocks_for(B)), dim3(kBlock), 0, st, d_points, B, m, table);
    if (d_beam_table && beam_table_count)
      hipLaunchKernelGGL(k_reweight_beam_sorted<true>, dim3(groups, segments), dim3(kBeamBlock), lds, st, p.w, n, g, m,
                         nonfree_layout(g.W, g.H, const_cast<uint32_t*>(nonfree_bits)), table, B, sorted->perm, p.pose, d_steps, partial,
                         per_segment, BeamTable{reinterpret_cast<const double4*>(d_beam_table), beam_table_count - 1u}, free_ahead ? 1u : 0u, sectors ? 1u : 0u);
    else
      hipLaunchKernelGGL(k_reweight_beam_sorted<false>, dim3(groups, segments), dim3(kBeamBlock), lds, st, p.w, n, g, m,
                         nonfree_layout(g.W, g.H, const_cast<uint32_t*>(nonfree_bits)), table, B, sorted->perm, p.pose, d_steps, partial,
                         per_segment, BeamTable{nullptr, 0u}, free_ahead ? 1u : 0u, sectors ? 1u : 0u);
    if (segments > 1) launch_lf_combine(st, p.w, n, sorted->perm, partial, segments, 2);
    return;
  }
  const dim3 grid(static_cast<unsigned>((n + (kBlock / kWave) - 1) / (kBlock / kWave)));
  hipLaunchKernelGGL(k_reweight_beam, grid, dim3(kBlock), static_cast<size_t>(B) * sizeof(double2), st, p, n, g, m,
                     nonfree_bits ? nonfree_layout(g.W, g.H, const_cast<uint32_t*>(nonfree_bits)) : NonFreeBits{},
                     reinterpret_cast<const double2*>(d_points), B, d_steps);
}

}  // namespace mcl
