// kernels.h — launchers of the gfx950 kernels behind the C ABI (include/beluga_mcl.h).
// Data layout in HBM (all owned by mcl_ctx):
//   particles : pose records of 4 f64 (cos, sin, x, y) + w[cap]   (two sets: live + resample target)
//   field     : f32 row-major H x W likelihood field (likelihood_field_model_base.hpp:120), 64 MB at 4000^2
//   cells     : int8 row-major H x W occupancy grid (beam model + free-space sampling), 16 MB at 4000^2
//   points    : f64 (x,y) pairs of the current scan, 17 KB at 1080 beams
//   cdf       : f64 inclusive scan of the normalised weights
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "se2.h"

namespace mcl {

// One particle set: poses as records of 4 doubles (cos, sin, x, y) — Sophus::SE2d::data() order, the order of the C ABI —
// and the weights as a separate array.  Records, not one array per component: the multinomial draw and the spatial
// ordering gather poses of random particles, and a record is one 32-byte access instead of four cache lines.
struct Particles {
  double4* pose;
  double* w;
};

struct FieldView {
  const float* data;
  uint32_t W, H;
  double inv_resolution;  // 1. / resolution (regular_grid.hpp:76)
  Pose2 world_to_field;   // grid.origin().inverse() (likelihood_field_model_base.hpp:99)
  float unknown_value;    // float(1 / max_laser_distance) (likelihood_field_model.hpp:75)
  // Derived table for the hot kernel: cube[i] = pz*pz*pz with pz = double(data[i]) — exactly the per-beam
  // term of likelihood_field_model.hpp:84-88 — and cube[W*H] = the same for unknown_value.  8 B per cell.
  const double* cube;
  int prob;  // LikelihoodFieldProbModel: the table holds log(pz) and the weight is exp(sum) (likelihood_field_prob_model.hpp:76-88)
  // Palette form of the same table (used when the field has few distinct values, which a quantised distance map always
  // has): pal_val[k] = the cube / log term of the k-th distinct field value; pal_idx = one uint16 per cell, stored in
  // 8x8-cell tiles of 128 bytes (see palette_offset) with a border of one tile of "unknown" cells all around, so that
  // clamping a cell coordinate to [-1, W] x [-1, H] replaces the in-grid test.  The uint16 is the LDS byte address of the
  // cell's palette entry inside k_reweight_lf_palette's workgroup memory: pal_base + 8 * k.
  // 2 B per cell instead of 8 and square tiles: a wave's gather touches ~4x fewer cache lines.
  const uint16_t* pal_idx;
  const double* pal_val;
  uint32_t pal_count;  // 0 = no palette
  uint32_t pal_pitch;  // bytes per row of tiles (border included)
  uint32_t pal_base;   // LDS byte offset of the palette (after the row-offset table of H + 2 words)
  uint32_t pal_bytes;  // size of pal_idx in bytes
  // Far tiles: one bit per 8x8 tile of pal_idx (border tiles included), set where all 64 cells hold the table's most common
  // entry far_entry (free space beyond max_obstacle_distance of anything: more than half of a typical map).  Rows of
  // far_row_bytes bytes, bit (tx & 7) of byte tx >> 3.  A look-up into such a tile needs no memory access at all; the gather
  // kernel of DISPERSED sets keeps the bitmap in LDS (k_reweight_lf_palette<true, true>).  nullptr = none.
  const uint8_t* far_bits;
  uint32_t far_row_bytes;
  uint32_t far_bytes;  // size of far_bits, a multiple of 16
  uint32_t far_entry;  // LDS byte address of the common entry (as stored in pal_idx)
  // The same bits by the tile's LINEAR index (ty * tiles_x + tx = a cell's byte offset in pal_idx >> 7): bit (index & 7) of byte
  // index >> 3 - the test then starts from the offset a look-up has computed anyway (k_reweight_lf_far_beams).  nullptr = none.
  const uint8_t* far_linear;
  uint32_t far_linear_bytes;  // a multiple of 16
};

constexpr uint32_t kMaxPalette = 2048;
// Byte offset of cell (x, y), -8 <= x < W + 8, -8 <= y < H + 8, in the tiled uint16 table: tiles are 8x8 cells,
// column-major inside a tile, so x contributes a plain shift and y = (row of tiles) * pitch + (y & 7) * 2.
__host__ __device__ inline uint32_t palette_row_offset(int32_t y, uint32_t pitch) {
  const uint32_t py = static_cast<uint32_t>(y + 8);
  return (py >> 3) * pitch + ((py & 7u) << 1) + 128u;  // + 128: the x border tile
}
__host__ __device__ inline uint32_t palette_offset(int32_t x, int32_t y, uint32_t pitch) {
  return palette_row_offset(y, pitch) + (static_cast<uint32_t>(x) << 4);
}

struct GridView {
  const int8_t* cells;
  uint32_t W, H;
  double resolution;
  Pose2 origin;          // grid frame in the world
  Pose2 origin_inverse;  // world -> grid
  int8_t free_value;
};

struct BeamModel {
  double z_hit, z_short, z_max, z_rand, sigma_hit, lambda_short, beam_max_range;
};

// Per-cycle constants of the motion model's sampling function (computed on the host from the control action).
//   differential   : three (mean, stddev) pairs: first rotation, translation, second rotation
//   omnidirectional: (mean, stddev) of the rotation and of the translation, stddev of the strafe, first rotation
//   stationary     : nothing (N(0, 0.02) on heading, x, y)
struct DiffDriveSampler {
  int kind;  // MCL_MOTION_*
  double m1, s1, mt, st, m2, s2;
  double first_c, first_s;
};

struct FreeCells {
  const uint32_t* index;  // linear indices of free cells
  uint64_t count;
};

struct HashParams {
  double res_x, res_y, res_theta;
};

struct ResampleArgs {
  uint64_t seed;
  uint32_t step;
  double random_state_probability;
  const double* d_random_state_probability;  // if set, read the probability from device memory instead (the recovery estimator's output, see RecoveryPolicy)
  uint64_t n_in;            // live particles of the source set
  uint64_t first_candidate; // global index of candidate 0 of this launch
  uint64_t count;           // candidates in this launch
  uint64_t out_offset;      // where candidate `first_candidate` lands in the output set
};

enum LfVariant : int {
  kLfWavePerParticle = 0,  // (rounds 1 - 4: a wave per particle over the f32 field; now the same kernel as 1)
  kLfLanePerParticle = 1,  // a lane per particle in index order over the f32 field (no ordering pass, no palette)
  kLfSortedLanes = 2,      // default: lanes = spatial neighbours (ordering pass), palette table, LDS patches
  kLfBeamLanes = 3         // wave per particle, lanes = beams, palette table: dispersed sets (chosen by the cycle, or forced)
};

// Per-context switches for A/B measurements and tests (mcl_set_option); no switch changes a result beyond the rounding of a
// particle's sum over the scan (libstdc++'s transform_reduce order with a lane per particle, a fixed tree with a wave per particle).
struct Tuning {
  int lf_variant = kLfSortedLanes;  // kernel family of the likelihood-field reweight
  int lf_fast = -1;                 // FMA variant with exact fallback: -1 / 1 = whenever its preconditions hold, 0 = never
  int lf_table = 0;                 // 0 = palette table when the field allows it, 1 = force the 8-byte cube table
  int lf_patch = 1;                 // index table through per-workgroup LDS patches: 1 = where the last launch found them useful,
                                    // 0 = never (per-lane gathers only), 2 = always
  int lf_loose_below = 224;         // LF patch kernel: a workgroup with fewer than this many 256ths of its beam groups fitting a patch
                                    // drops the patches (no producer, no barriers) and gathers every look-up
  int lf_dispersed = 2;             // a set the patch kernel reports as dispersed (lf_patch = 1): 2 = lanes over the beams of a pose, the poses
                                    // in the position-major order, far-tile bitmap (k_reweight_lf_far_beams; where its tables fit LDS, else
                                    // as 0), 0 = the ordered-lanes gather kernel (a lane per particle; rounds 2 - 5), 1 = wave per particle /
                                    // lane per beam without any order (k_reweight_lf_beams; 20 % slower than 0: profiles/r02_dispersed_study.txt)
  int lf_far_beams_per_wave = 0;    // particles a wave of k_reweight_lf_far_beams takes (0 = 32)
  int device_policy = 1;            // recovery estimator on the device when the cycle has no host-side decision
  int sort_min_particles = 16384;   // below this the ordering passes cost more than they save (likelihood-field models)
  int beam_sort_min_particles = 16384;  // beam model: the ordered kernel (LDS bit window, scan segments) from here on; below, a wave per
                                       // particle over the whole-grid maps (measured crossover: 12K particles at 180 beams, 28K at 1080)
  int lf_far_tiles = 1;             // the gather kernel skips look-ups into far tiles (FieldView::far_bits): 1 = for sets reported as
                                    // dispersed (lf_patch = 1), 0 = never, 2 = whenever it gathers
  int key_layout = -1;              // ordering key: -1 = position-major for dispersed likelihood-field sets, heading-major otherwise; 0 / 1 force
  int lf_small_particles = 65536;   // likelihood-field sets below this: a wave per particle with the lanes over the beams, no ordering
                                    // (measured: 25 % faster than the ordered kernels at 20K particles, 10 % at 50K, 12 % slower at 100K)
  int field_build = 0;              // mcl_set_map: 0 = host wavefront (bit-identical to the reference), 1 = exact EDT on the device
  int key_curve = 1;                // heading-major ordering key: 1 = Hilbert curve through (heading, y, x), 0 = Morton order
  int key_warp = 1;                 // heading-major key: 1 = bins of equal mass (the frame's +-4 sigma mapped through the normal distribution
                                    // function) when the frame comes from an estimate of the set, 0 = bins of equal width
  int key_bits_xy = 0;              // bits of the x / y bins of that key: 0 = chosen per cycle from the cloud's spread and the scan's
                                    // reach (4 .. 6), otherwise forced; round 2: 6 (8 heading bits)
  int cycle_spin = -1;              // fixed-size cycles: 1 = the host waits for the cycle's own completion word (written to mapped host memory
                                    // by the last kernel, Completion) instead of the stream's completion signal; 0 = hipStreamSynchronize;
                                    // -1 = the word for sets of 256K particles and more.  Measured (round 6, three alternating runs of 65
                                    // cycles each): 1749 against 1731 cycles/s in the driver's window at 1M particles, 1933 against 1901 once
                                    // the cloud has settled; round 3 at 2000 particles: 4 us per cycle SLOWER (the launches behind an
                                    // unsynchronised stream cost the host more) - hence the threshold.  The waiting thread spins.
  int beam_table = 1;               // beam model, ordered kernel: the terms that depend on the expected range alone from a table over the hit's
                                    // squared cell distance (built at mcl_set_map); 0 = evaluated per beam
  int lf_weight_sums = 1;           // fixed-size cycle: the normalisation factor is added up from the LF patch kernel's workgroup sums of the
                                    // new weights (no k_chunk_sum pass); 0 = from chunk sums of the weights
  int lf_split = 3;                 // LDS-patch planner: a group of 8 beams that fits no whole 64 x 64 patch (a range discontinuity inside
                                    // it) may go through two half patches (beams [0, k) and [k, 8): 32 x 64 or 64 x 32 cells each); 0 = never
  int lf_margin = 1;                // LDS-patch planner, rotation part of the bound: 1 = per axis (|sin d| |q'y| + (1 - cos d) |q'x|),
                                    // 0 = round 2's |R_p - R_ref| |q| on both axes
  int beam_free_ahead = 1;          // beam model, ordered kernel: a workgroup's lanes pass the cells its middle ray's clearance proves free in one
                                    // closed-form step (per beam and workgroup); 0 = block-distance skips only
  int lf_queue_grid = 0;            // workgroups of the queue form (lf_queue): 0 = three per CU, otherwise this many (tests: few workgroups, many
                                    // blocks each)
  int device_cus = 0;               // compute units of the context's device (filled in by mcl_create; 0 = assume 256)
  int shard_pad_permille = 1063;    // sharded fixed-size cycle: the ancestor exchange moves a FIXED number of entries per pair of ranks - this many
                                    // thousandths of a shard's share of another shard's draws, plus eight standard deviations - so that no count
                                    // is read by the host before the cycle's end; 0 = exact counts (one more host synchronisation per cycle)
  int lf_ends_first = 1;            // LDS-patch kernel: the blocks are taken from both ends of the order inwards (the fringe's slow blocks first)
  int beam_sectors = 1;             // beam model, ordered kernel, scanners that reach beyond half the LDS window: the scan in four sectors, each with
                                    // a window of its own that holds its rays (0 = one centred window; the rays that leave it go on in global memory)
  int lf_queue = 1;                 // LDS-patch kernel: 1 = as many workgroups as stay resident (lf_queue_grid) take the blocks from a queue where
                                    // there are more blocks than that (k_reweight_lf_patch<true>), 0 = one workgroup per block.  Bit-identical.
  int scan_fused = 1;               // fixed-size cycle that resamples: normalisation, totals, recovery estimator and CDF in ONE launch
                                    // (k_normalize_cdf): 1 = for sets of up to 64K particles (where the cycle is bound by the host's launches:
                                    // one less), 2 = wherever the kernel takes the set (up to 2M particles; measured at 1M: 18.6 us against
                                    // 10.3 + 7.2 - a hand-off inside a launch costs what the kernel boundary did), 0 = k_normalize + k_cdf.
                                    // Bit-identical.
  int draw_fold = 1;                // the draw kernel's last workgroup to finish adds up the estimate sums (no k_final_rows launch behind it):
                                    // 1 = for sets of up to 64K particles, 2 = up to 4M (measured at 1M: 56.2 us against 49.2 + 4.4 - every
                                    // workgroup ends on the ticket's round trip), 0 = k_final_rows.  Bit-identical.
  int noise_ahead = 1;              // fixed-size cycles, sets of more than 64K particles: the next cycle's propagation normals are drawn a cycle AHEAD:
                                    // 1 = by the draw kernel, whose vector units wait for the fabric (draw + 7.4 us, k_propagate - 12 at 1M); 2 = by a
                                    // kernel of its own behind the cycle's last one, while the host is away (k_noise_ahead, 17 us: cycles that end
                                    // on the completion word); both up to 2M particles (at 10M the draw is at the HBM's limit: measured a loss);
                                    // 0 = by k_propagate itself.  Bit-identical.
  int order_ahead = 1;              // with noise_ahead = 1: the NEXT cycle's spatial order is computed behind a cycle's last kernel, while the host is
                                    // away, from the predicted control action (the one of the cycle that ends); the next cycle uses it if the action it
                                    // gets is close to the prediction, else it orders by the real poses as before.  Only locality depends on the order.
  int norm_store = 0;               // fixed-size cycle that resamples at once: 0 = k_normalize leaves the chunk sums of the normalised weights
                                    // but does not store them - the CDF kernel divides again (same division, same bits), nothing else reads them;
                                    // 1 = stored
  int small_fused = 1;              // sets of up to 4096 particles (plain estimate, one context): everything behind the reweight - normalise,
                                    // policies, fixed-size or KLD resampling, estimate sums - in one launch of one workgroup and one host
                                    // synchronisation (k_small_tail); 0 = the kernels of the large path
  int lf_unit_weights = 1;          // LF patch kernel on a set whose weights are all 1.0 (fresh from a resampling or an initialisation):
                                    // the old weight is not loaded (1.0 x = x: bit-identical); 0 = always loaded
};

// Spatial ordering of the particles (kLfSortedLanes, ordered beam kernel): the 64 lanes of a wave should hold neighbouring
// poses, so that their look-ups for a given beam fall into the same few table lines.  Order = full sort by a 20-bit key:
// bins of x, y (6 bits each) and heading (8 bits) over +-4 sigma around the cloud's centre, the two extra heading bits on
// top, the rest Morton-interleaved.  The frame of the bins comes from the previous cycle's estimate moved by the control
// action (host, no pass over the particles), or from a bounding-box pass when the host has no estimate of the set.
struct KeyFrame {
  double cx, cy;          // centre of the x / y bins
  double c0, s0;          // heading of the centre of the heading bins
  float inv_x, inv_y;     // 1 / span of the x / y bins (span = 8 sigma)
  float inv_t, t_off;     // heading bins: u = (delta - t_off) * inv_t + 0.5
  uint32_t layout;        // 0: heading-major key (dense sets: a workgroup's poses fit an LDS patch), 1: position-major key
                          // (dispersed sets: neighbours in the order share a region of the map, whatever their heading);
                          // | 2: the heading-major key follows the Z (Morton) curve instead of the Hilbert curve (option key_curve)
                          // | 4: heading-major key over bins of equal mass of a normal set instead of equal width (option key_warp)
  uint32_t bits_xy;       // heading-major key: bits of the x and of the y bins (4 .. 6; 0 = 6); the heading takes the other 20 - 2 bits_xy
};
constexpr uint32_t kSortDigits = 1024;  // two least-significant-digit-first passes of 10 bits each
// Position of the cell (a0, a1, a2), `bits` bits each (bits <= 6), along the 3-D Hilbert curve through the (2^bits)^3 cells
// (Skilling's transpose form, "Programming the Hilbert curve", 2004: undo the excess work, Gray-encode, interleave with a0 most
// significant).  Consecutive positions are face neighbours, so ANY run of the order is a connected, compact set of cells - a run
// of the Morton order that crosses a high-level boundary of the Z curve is two pieces far apart, and the workgroup that holds it
// (448 consecutive particles of the order, k_reweight_lf_patch) fits no LDS patch.  The curve enters at (0, 0, 0) and leaves at
// (2^bits - 1, 0, 0): with the heading on axis 0, the slabs of the key's top heading bits chain into one continuous curve.
template <uint32_t bits>
__host__ __device__ inline uint32_t hilbert_index_3_fixed(uint32_t a0, uint32_t a1, uint32_t a2) {
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t x0 = a0 & mask, x1 = a1 & mask, x2 = a2 & mask;
#pragma unroll
  for (uint32_t q = 1u << (bits - 1); q > 1; q >>= 1) {
    const uint32_t p = q - 1;
    x0 ^= (x0 & q) ? p : 0u;  // axis 0 against itself: invert or nothing
    {
      const uint32_t t = (x0 ^ x1) & p;
      const bool inv = (x1 & q) != 0;
      x0 ^= inv ? p : t;
      x1 ^= inv ? 0u : t;
    }
    {
      const uint32_t t = (x0 ^ x2) & p;
      const bool inv = (x2 & q) != 0;
      x0 ^= inv ? p : t;
      x2 ^= inv ? 0u : t;
    }
  }
  x1 ^= x0;
  x2 ^= x1;
  uint32_t t = 0;
#pragma unroll
  for (uint32_t q = 1u << (bits - 1); q > 1; q >>= 1) t ^= (x2 & q) ? q - 1 : 0u;
  x0 ^= t;
  x1 ^= t;
  x2 ^= t;
  auto spread = [](uint32_t v) {  // ..fedcba -> f00e00d00c00b00a
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
  };
  return (spread(x0) << 2) | (spread(x1) << 1) | spread(x2);
}
__host__ __device__ inline uint32_t hilbert_index_3(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t bits) {  // straight-line code per width
  switch (bits) {
    case 1: return hilbert_index_3_fixed<1>(a0, a1, a2);
    case 2: return hilbert_index_3_fixed<2>(a0, a1, a2);
    case 3: return hilbert_index_3_fixed<3>(a0, a1, a2);
    case 4: return hilbert_index_3_fixed<4>(a0, a1, a2);
    case 5: return hilbert_index_3_fixed<5>(a0, a1, a2);
    default: return hilbert_index_3_fixed<6>(a0, a1, a2);
  }
}
struct SortScratch {
  uint32_t* keys;                // [n] key of particle i
  uint32_t* perm;                // [n] sorted position -> particle index
  uint32_t* table;               // [kSortDigits][nblocks] block histograms -> exclusive offsets (reused by both passes)
  uint32_t* totals;              // [kSortDigits] digit totals, then [kSortDigits] their exclusive scan (the buckets' first positions) and
                                 // [16] flags ([0]: some bucket is beyond kSortHugeBucket) - written by the first pass's first workgroup
  unsigned long long* keyidx;    // [n] (high digit << 32 | index) after the first pass
  double* bbox;                  // [8] min/max of x, y, relative heading (+ [6 * nblocks] partials behind it)
  KeyFrame* frame;               // key frame derived from the bounding box (device-resident fallback)
  double* partial;               // [kLfMaxSegments][min(n, 262144)] scan-segment sums (medium particle counts); may be null
};
constexpr uint32_t kLfMaxSegments = 16;
constexpr uint64_t kLfSegmentedBelow = 262144;  // particles

// K1  actions/propagate.hpp:57-79 + differential_drive_model.hpp:156-163
// scan_src / scan_dst (optional): the cycle's scan, copied by the kernel from mapped pinned host memory into HBM (no
// copy-engine hand-off on the stream).  sort + frame (optional): the kernel also emits the ordering keys and the first
// pass's block histograms (launch_order_particles then skips its own key pass).
void launch_propagate(hipStream_t st, Particles p, uint64_t n, DiffDriveSampler smp, uint64_t seed, uint32_t step,
                      uint64_t index_offset, const double* scan_src = nullptr, double* scan_dst = nullptr, uint32_t scan_doubles = 0,
                      const SortScratch* sort = nullptr, const KeyFrame* frame = nullptr, const double* normals_ahead = nullptr,
                      uint64_t normals_stride = 0);
// The propagation's standard normals per particle for `step`, drawn ahead of the cycle that uses them (k_noise_ahead: the three the motion
// models use, as three arrays of n doubles): launch_propagate(..., normals_ahead, stride = that n) then reads them instead of drawing (not the
// small-set kernel).  Same bits.
void launch_noise_ahead(hipStream_t st, uint64_t seed, uint32_t step, uint64_t index_offset, uint64_t n, double* d_normals);
void launch_pull_scan(hipStream_t st, const double* scan_src, double* scan_dst, uint32_t scan_doubles);
// Full sort of the particles by the ordering key -> sort->perm.  frame == nullptr: bounding-box pass + device-resident frame.
// keys_ready: launch_propagate already wrote sort->keys and the first pass's block histograms.
// layout: KeyFrame::layout of the device-resident frame (a host frame carries its own).
void launch_order_particles(hipStream_t st, Particles p, uint64_t n, const SortScratch* sort, const KeyFrame* frame, bool keys_ready,
                            uint32_t layout = 0);
// The same order a cycle AHEAD: sort->keys hold the keys the draw kernel predicted for the next cycle (launch_resample_draw_and_estimate,
// keys_ahead) - their block histograms and the three ordering kernels -> sort->perm.
void launch_order_ahead(hipStream_t st, uint64_t n, const SortScratch* sort);
// K2  actions/reweight.hpp:53-60 + likelihood_field_model.hpp:68-91 (kLfSortedLanes needs launch_order_particles first)
// scan_is_short: every scan point lies within 8192 cells of the sensor (precondition of the kernel's FMA variant)
// use_patches: the LDS-patch kernel where its preconditions hold (dense sets); patch_stats: running totals it reports
struct PatchStats {
  unsigned long long* device;  // [3]: groups planned, groups through a patch, workgroups reported (never reset)
  unsigned long long* mirror;  // [3]: mapped host copy of the first two, written by the last workgroup of a launch, and their
                               // low halves packed into one word (planned | through << 32: one store, read without synchronisation)
  uint32_t loose_below;        // a workgroup with fewer than loose_below / 256 of its groups fitting a patch gathers them all
  uint32_t isotropic_margin;   // 1: the rotation part of the bound as |R_p - R_ref| |q| on both axes (Tuning::lf_margin = 0)
  uint32_t split_patches;      // 1: a group that fits no whole patch may go through two half patches (Tuning::lf_split)
  double* weight_sums;         // optional: [workgroups] sums of the new weights, one per workgroup of the patch kernel (the
                               // normalisation's input: launch_sum_and_normalize); only written by single-segment launches
  unsigned int* arrivals;      // the queue of blocks (k_reweight_lf_patch<true>): the next block to take; wraps to 0 behind a launch's last fetch
};
// *weight_sums_written (optional): how many workgroup sums of the new weights the launch left in patch_stats.weight_sums (0: none -
// another kernel ran, or the launch was segmented)
void launch_reweight_lf(hipStream_t st, Particles p, uint64_t n, FieldView f, const double* d_points, uint32_t B, int variant,
                        const SortScratch* sort, bool scan_is_short, const Tuning& tuning, bool use_patches, PatchStats patch_stats,
                        bool dispersed = false, bool* far_tiles_used = nullptr, uint32_t* weight_sums_written = nullptr,
                        bool* queue_used = nullptr, bool unit_weights = false, bool* far_beams_used = nullptr);
// K2' beam_model.hpp:104-150 + raycasting.hpp:62-107 + bresenham.hpp:84-160
// `sorted` != nullptr: lane-per-ordered-particle variant (needs launch_order_particles first).
// d_beam_points: scratch of kBeamPointDoubles * B doubles (per-beam terms shared by all particles; ordered variant only).
constexpr uint32_t kBeamPointDoubles = 5;
// d_beam_table (optional, ordered variant): launch_beam_table's output, beam_table_count entries of 4 doubles
// w[perm[t]] *= the sum of a scan's segment sums partial[s][t] (mode 0: 1 + sum, 1: exp(sum), 2: sum - the beam model), segments in order
void launch_lf_combine(hipStream_t st, double* w, uint64_t n, const uint32_t* perm, const double* partial, uint32_t segments, int mode);
void launch_reweight_beam(hipStream_t st, Particles p, uint64_t n, GridView g, BeamModel m, const double* d_points, uint32_t B,
                          unsigned long long* d_steps, const SortScratch* sorted, const uint32_t* nonfree_bits, double* d_beam_points,
                          const double* d_beam_table = nullptr, uint32_t beam_table_count = 0, bool free_ahead = true, bool sectors = true);
// The beam model's terms that depend on the expected range alone, tabulated over the squared cell distance of the hit (beam_kernels.hip
// BeamTable): entries = beam_table_entries(...) (0: the range spans too many cells for a table), 4 doubles each.
constexpr double kBeamTableMaxCells = 2046.0;
uint32_t beam_table_entries(double beam_max_range, double resolution);
void launch_beam_table(hipStream_t st, BeamModel m, double resolution, uint32_t entries, double* table);
// The occupancy the ray walks read, in one buffer of nonfree_words(W, H) words: one bit per cell (1 = not free), ceil(W/32)
// words per row, followed by two coarse bitmaps — one bit per 8 x 8-cell block, "any cell not free" — row-major and column-major —
// and the block distance map (one byte per block).
struct NonFreeBits {
  const uint32_t* fine;
  const uint32_t* rows;     // [ceil(H/8)][row_words]
  const uint32_t* columns;  // [ceil(W/8)][column_words]
  const uint8_t* dist;      // [ceil(H/8)][dist_stride]: Chebyshev distance, in blocks, to the nearest block with a bit in `rows`
                            // (0 = the block itself), capped at 17
  uint32_t words_per_row, row_words, column_words, dist_stride;
};
NonFreeBits nonfree_layout(uint32_t W, uint32_t H, uint32_t* base);
size_t nonfree_words(uint32_t W, uint32_t H);
void launch_pack_nonfree(hipStream_t st, const int8_t* cells, uint32_t W, uint32_t H, int8_t free_value, uint32_t* bits);
// Per-device kernel attributes (dynamic LDS opt-in of the ordered beam kernel); call once per context after hipSetDevice.
void configure_device_kernels();

// Deterministic chunked reductions / scans.  Chunk = 2048 consecutive elements per workgroup.
constexpr uint32_t kChunk = 2048;
inline uint32_t num_chunks(uint64_t n) { return static_cast<uint32_t>((n + kChunk - 1) / kChunk); }

// K3a: partial[b] = sum of w over chunk b ; then d_out[0] = sum of partials (fixed order).
// host_mirror (here and below, optional): mapped pinned-host memory that also receives the result
void launch_weight_sum(hipStream_t st, const double* w, uint64_t n, double* d_partials, double* d_out, double* host_mirror = nullptr);
// K3b: w[i] /= *d_factor unless |factor-1| < eps (normalize.hpp:73-82); chunk sums of the new w and w^2;
//      d_out[0] = total of new w, d_out[1] = total of squares.
void launch_normalize(hipStream_t st, double* w, uint64_t n, const double* d_factor, double* d_chunk_sum, double* d_chunk_sumsq,
                      double* d_out, double* host_mirror = nullptr);
// known_partials (optional): `known_count` sums whose total is the normalisation factor (PatchStats::weight_sums) - k_chunk_sum is skipped
void launch_sum_and_normalize(hipStream_t st, double* w, uint64_t n, double* d_partials, double* d_chunk_sum, double* d_chunk_sumsq,
                              double* d_sums, double* host_mirror, bool finalize = true, const double* known_partials = nullptr,
                              uint32_t known_count = 0, bool store_weights = true);
// ThrunRecoveryProbabilityEstimator (thrun_recovery_probability_estimator.hpp:69-89, exponential_filter.hpp:32-44) evaluated
// on the device so that a cycle without host-side decisions needs no mid-cycle read-back: d_policy = {slow, fast, p}.
// It rides on the workgroup that adds up the totals of the normalised weights (launch_cdf / launch_norm_finalize).
struct RecoveryPolicy {
  double alpha_slow, alpha_fast;
  int resampling;       // this cycle resamples: reset the filters when p > 0 (amcl_core.hpp:184-186)
  double* d_policy;     // {slow, fast, p}
  double* host_mirror;  // optional: p at [2]
};
// d_sums[0] = sum, d_sums[1] = sum of squares of the normalised weights (from k_normalize's chunk rows) + optional policy step.
void launch_norm_finalize(hipStream_t st, const double* d_chunk_sum, const double* d_chunk_sumsq, uint64_t n, double* d_sums,
                          double* sums_mirror, const RecoveryPolicy* policy);
// K5: cdf[i] = inclusive scan of w; d_chunk_sum is recomputed; d_total[0] = cdf[n-1].
// 16-ary search tree over the cdf: level l (l = 1 .. depth) keeps every 16^l-th cumulative sum (the last of each group
// of 16 entries of the level below, one 128-byte line per group), so that std::lower_bound touches one line per level.
// Pure comparisons: the result is exactly cdf_lower_bound's.
// A launch's own completion word (k_final_rows): d_ticket counts its workgroups (never reset), host_flag is a word of mapped host
// memory that receives seq when the last of them is done, behind everything the launch mirrored to the host.
struct Completion {
  unsigned long long* d_ticket{nullptr};
  unsigned long long* host_flag{nullptr};
  unsigned long long seq{0};
};
constexpr int kCdfTreeMaxDepth = 8;
struct CdfTree {
  const double* cdf;
  const double* levels;                   // level 1 first
  uint32_t offset[kCdfTreeMaxDepth];      // of level l + 1 inside `levels`
  uint32_t size[kCdfTreeMaxDepth];
  int depth;                              // number of sampled levels (0 for n <= 16)
  uint64_t n;
};
inline uint64_t cdf_tree_doubles(uint64_t n) { return n / 15 + 32 * kCdfTreeMaxDepth; }
inline CdfTree make_cdf_tree(const double* cdf, const double* levels, uint64_t n) {
  CdfTree t{};
  t.cdf = cdf;
  t.levels = levels;
  t.n = n;
  uint64_t size = n, off = 0;
  while (size > 16 && t.depth < kCdfTreeMaxDepth) {
    size = (size + 15) / 16;
    t.offset[t.depth] = static_cast<uint32_t>(off);
    t.size[t.depth] = static_cast<uint32_t>(size);
    off += (size + 15) & ~15ull;  // every group of 16 entries in a 128-byte line of its own
    ++t.depth;
  }
  return t;
}
// launch_cdf also fills the tree levels when `tree_levels` is given (cdf_tree_doubles(n) doubles).
// finalize_*: the first workgroup also leaves the totals of known_chunk_sum / finalize_sumsq in finalize_sums[0..1] and runs
// the recovery estimator (see launch_norm_finalize) — for the cycle that goes straight from k_normalize into a resample.
void launch_cdf(hipStream_t st, const double* w, uint64_t n, double* d_chunk_sum, double* d_chunk_offset, double* cdf,
                double* d_total, double* tree_levels, const double* known_chunk_sum = nullptr, const double* finalize_sumsq = nullptr,
                double* finalize_sums = nullptr, double* finalize_mirror = nullptr, const RecoveryPolicy* policy = nullptr,
                const double* d_factor = nullptr);
// Normalisation by the set's own total + totals of the normalised weights (+ recovery estimator) + CDF and its search tree in one launch
// (k_normalize_cdf): what launch_sum_and_normalize(finalize = false) followed by launch_cdf(known chunk sums, finalize arguments) leave,
// bit for bit.  known_partials as in launch_sum_and_normalize (nullptr: k_chunk_sum into d_partials first).  scan_state: kScanStateWords
// words of 8 bytes, zero when allocated, owned by the context; epoch: nonzero, another one than the previous launch's on that state.
// write_weights = false: the normalised weights themselves are not stored (a cycle that resamples at once never reads them).
// Returns false, nothing launched, where the set is beyond what the kernel takes (more than 2M particles).
constexpr size_t kScanStateWords = 8 + 4 * 1024;
bool launch_normalize_cdf(hipStream_t st, double* w, uint64_t n, double* d_partials, const double* known_partials, uint32_t known_count,
                          double* d_sums, double* sums_mirror, double* d_chunk_sum, double* d_chunk_sumsq, bool write_weights, double* cdf,
                          double* d_total, double* tree_levels, const RecoveryPolicy* policy, unsigned long long* scan_state, uint32_t epoch);
// The whole tail of a small set's cycle - normalise, policies, resample (fixed size or KLD-adaptive), estimate sums - in one launch of one
// workgroup (k_small_tail; sets and candidate streams of up to 4096 particles).  Results through `mirror` (32 doubles of mapped host memory)
// and d_scalars: [0] weight sum, [1] [2] sum and sum of squares of the normalised weights, [5] resampled (0 / 1), [6] particles after the
// cycle, [7] effective sample size (-1: not evaluated), [8..17) the estimate's sums, [18] [19] the recovery filters' new outputs, [22] the
// random state probability.  Returns false, nothing launched, where the set does not fit.
struct SmallTail {
  Particles src, dst;
  uint32_t n, min_particles, max_particles;
  uint64_t seed;
  uint32_t step;
  bool fires, selective;
  double alpha_slow, alpha_fast, slow, fast, kld_epsilon, kld_z;
  HashParams hp;
  GridView g;
  FreeCells fc;
  double pivot_x, pivot_y;
  double* mirror;
  double* d_scalars;
  unsigned long long* done_flag{nullptr};  // completion word in mapped host memory (optional) and the value it takes
  unsigned long long done_seq{0};
};
bool launch_small_tail(hipStream_t st, const SmallTail& t);
// K6: one thread per candidate (views/sample.hpp:102,133-135; random_intersperse.hpp:90-115; particle_traits.hpp:105).
void launch_resample_draw(hipStream_t st, Particles src, CdfTree cdf, const double* d_total, Particles dst,
                          ResampleArgs a, GridView g, FreeCells fc, HashParams hp, unsigned long long* d_hashes);
void launch_resample_draw_and_estimate(hipStream_t st, Particles src, CdfTree cdf, const double* d_total, Particles dst, ResampleArgs a,
                                       GridView g, FreeCells fc, HashParams hp, double pivot_x, double pivot_y, double* d_partials,
                                       double* d_sums, double* host_mirror, const Completion* done = nullptr,
                                       unsigned int* fold_ticket = nullptr, double* normals_ahead = nullptr, uint64_t normals_stride = 0,
                                       uint64_t normals_index_offset = 0, uint32_t normals_step = 0, uint32_t* keys_ahead = nullptr,
                                       const DiffDriveSampler* predicted = nullptr, const KeyFrame* frame_ahead = nullptr);
// Sharded variant: targets given, no RNG (mcl_gather_by_cdf).
// Sharded resampling helpers (mcl_resample_targets / mcl_commit_resampled).
void launch_resample_targets(hipStream_t st, uint64_t seed, uint32_t step, double p, double total, uint64_t first_slot,
                             uint64_t count, uint64_t n_free, double* d_targets, const double* d_plan = nullptr);
// Sharded fixed-size cycle: CDF intervals, global total, totals of the normalised weights and the recovery estimator from the
// gathered shard statistics, on the device (d_plan = {total, random state probability}; d_intervals = ends[world], offsets[world]).
struct RecoveryPolicy;
void launch_shard_plan(hipStream_t st, const double* d_stats, uint32_t world, uint64_t n_total, double* d_sums, double* sums_mirror,
                       const RecoveryPolicy& policy, double* d_intervals, double* d_plan);
// Counting sort of resample targets by owning shard; d_block_hist needs world * num_chunks(count) words.
void launch_route_targets(hipStream_t st, const double* d_targets, uint64_t count, const double* d_ends, const double* d_offsets,
                          uint32_t world, uint32_t self_rank, uint8_t* d_dest, uint32_t* d_block_hist, uint32_t* d_chunk_sum,
                          uint32_t* d_chunk_off, double* d_send_targets, uint32_t* d_order, long long* d_counts, uint32_t pad_capacity = 0,
                          double* d_overflow = nullptr);
void launch_gather_by_cdf_aos(hipStream_t st, Particles src, CdfTree cdf, const double* d_targets, uint64_t m,
                              double* d_out);
void launch_commit_routed(hipStream_t st, Particles dst, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count,
                          const double* d_replies, const uint32_t* d_order, const double* d_targets, GridView g, FreeCells fc);
// The fixed-capacity exchange keeps injected slots (NaN targets) out of its request lists: their random states, straight from d_targets.
void launch_commit_injected(hipStream_t st, Particles dst, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count,
                            const double* d_targets, GridView g, FreeCells fc);
// K7: exact parallel take_while_kld (take_while_kld.hpp:72-88).
void launch_finish_candidates(hipStream_t st, uint64_t seed, uint32_t step, uint64_t first_slot, uint64_t count, const double* d_replies,
                              const uint32_t* d_order, const double* d_targets, GridView g, FreeCells fc, HashParams hp,
                              double* d_states, unsigned long long* d_hashes);
struct KldTable {
  unsigned long long* keys;  // 0 = empty (hash 0 is remapped)
  unsigned int* first;       // smallest candidate index that produced the key
  uint64_t capacity;         // power of two
};
void launch_kld_insert(hipStream_t st, const unsigned long long* d_hashes, uint64_t first, uint64_t count, KldTable t);
// flags[j] = 1 if candidate j is the first with its hash; inclusive scan -> d_k[j]; then the first j
// violating (j+1 <= min || j+1 <= target(k_base + k[j])) is atomically minimised into *d_first_fail.
void launch_kld_scan(hipStream_t st, const unsigned long long* d_hashes, uint64_t first, uint64_t count, KldTable t,
                     uint32_t* d_flags_scan, uint32_t* d_chunk_sum, uint32_t* d_chunk_offset, const uint32_t* d_k_base,
                     uint32_t* d_k_total, uint64_t min_particles, double epsilon, double z, unsigned long long* d_first_fail);
// K8: estimation.hpp:436-475 sufficient statistics; d_out[9].
void launch_estimate_sums(hipStream_t st, Particles p, uint64_t n, double pivot_x, double pivot_y, double* d_partials,
                          double* d_out, double* host_mirror = nullptr);
// cluster_based_estimate (algorithm/cluster_based_estimation.hpp): hash + per-cell aggregation + compaction of the occupied
// cells (for the host's cluster assignment), the write-back of the cells' cluster ids and the masked estimate sums.
void launch_cluster_cells(hipStream_t st, Particles p, uint64_t n, HashParams hp, unsigned long long* d_hashes,
                          unsigned long long* t_keys, unsigned int* t_first, double* t_wsum, unsigned int* t_count,
                          unsigned int* t_cluster, uint64_t capacity, unsigned long long* c_key, unsigned int* c_first,
                          unsigned int* c_count, unsigned int* c_slot, double* c_wsum, double* c_state, unsigned int* c_size,
                          unsigned int list_capacity, bool table_ready = false);
// The same for a set of up to 4096 particles: one workgroup each (k_small_cluster_cells: straight into the caller's list - the mapped host
// list - and its size into size_mirror as well; false = the set does not fit; k_small_cluster_sums: the cells' keys and cluster ids back
// in, the sums of the particles of cluster `wanted` out).
bool launch_small_cluster_cells(hipStream_t st, Particles p, uint64_t n, HashParams hp, unsigned long long* c_key, unsigned int* c_first,
                                unsigned int* c_count, unsigned int* c_slot, double* c_wsum, double* c_state, unsigned int* c_size,
                                unsigned int* size_mirror);
void launch_small_cluster_sums(hipStream_t st, Particles p, uint64_t n, HashParams hp, const unsigned long long* d_keys,
                               const unsigned int* d_cluster, uint32_t cells, unsigned int wanted, double pivot_x, double pivot_y, double* d_out,
                               double* host_mirror);
void launch_cell_set_cluster(hipStream_t st, const unsigned int* d_slot, const unsigned int* d_cluster, uint32_t m,
                             unsigned int* t_cluster);
void launch_estimate_sums_cluster(hipStream_t st, Particles p, uint64_t n, const unsigned long long* d_hashes,
                                  unsigned long long* t_keys, unsigned int* t_cluster, uint64_t capacity, unsigned int wanted,
                                  double pivot_x, double pivot_y, double* d_partials, double* d_out, double* host_mirror = nullptr);
// init: multivariate_normal_distribution.hpp:96-126 with T = V sqrt(L)
void launch_init_normal(hipStream_t st, Particles p, uint64_t n, const double mean[3], const double T[9], uint64_t seed,
                        uint64_t index_offset);
// initialize_from_map: multivariate_uniform_distribution.hpp:126-161 over the free cells, weight 1
void launch_init_from_map(hipStream_t st, Particles p, uint64_t n, uint64_t seed, uint64_t index_offset, GridView g, FreeCells fc);
void launch_fill(hipStream_t st, double* p, uint64_t n, double v);
// d_out[k] = sum over r < rows (in order) of d_gathered[r * columns + k], columns <= 64 (gathered per-shard scalars)
void launch_sum_rows(hipStream_t st, const double* d_gathered, uint32_t rows, uint32_t columns, double* d_out, double* host_mirror);
// Likelihood field built on the device: exact Euclidean distance transform + the reference's Gaussian map, unknown-space
// overlay and edge mask (likelihood_field_model_base.hpp:130-185).  Scratch: W * H uint16 and int16.  Returns false (nothing
// launched) when max_obstacle_distance spans more than kFieldBuildMaxReach cells: the caller then builds on the host.
constexpr double kFieldBuildMaxReach = 1024.0;
struct FieldBuildParams {
  double max_obstacle_distance, max_laser_distance, z_hit, z_random, sigma_hit;
  int model_unknown_space, only_obstacle_boundaries;
};
bool launch_build_field(hipStream_t st, const int8_t* d_cells, uint32_t W, uint32_t H, double resolution, int8_t free_value,
                        int8_t unknown_value, int8_t occupied_value, const FieldBuildParams& fp, uint16_t* d_column_distance,
                        int16_t* d_column_offset, float* d_field);
// cube[i] = double(field[i])^3 (or log(double(field[i])) for the prob model) for i < cells, cube[cells] = same for `unknown`
void launch_cube_table(hipStream_t st, const float* field, uint64_t cells, float unknown_value, double* cube, int prob);
// keys: the sorted bit patterns of the distinct field values (count entries, unknown_value among them)
void launch_palette_table(hipStream_t st, const float* field, uint32_t W, uint32_t H, float unknown_value, const uint32_t* keys,
                          uint32_t count, int prob, uint16_t* idx, double* val, uint32_t pal_base);
// Far tiles of a palette table (FieldView::far_bits).  votes[k] (count entries, zeroed here) = number of tiles uniformly equal
// to entry k; the caller picks the entry and has launch_far_tile_bits write the bitmap (far_bytes bytes).
void launch_far_tile_votes(hipStream_t st, const uint16_t* idx, uint32_t tiles, uint32_t pal_base, uint32_t count, uint32_t* votes);
void launch_far_tile_bits_linear(hipStream_t st, const uint16_t* idx, uint32_t tiles, uint32_t entry, uint32_t bytes, uint8_t* bits);
void launch_far_tile_bits(hipStream_t st, const uint16_t* idx, uint32_t tiles_x, uint32_t tiles_y, uint32_t entry, uint32_t row_bytes,
                          uint32_t far_bytes, uint8_t* bits);
// AoS (c,s,x,y) host layout <-> SoA device layout

}  // namespace mcl
