// context.hip — mcl_ctx and the C ABI of include/beluga_mcl.h.
//
// Host-side control flow of beluga::Amcl::update (amcl_core.hpp:165-201): the policies, the 2-deep
// control window and the recovery estimator run on the host exactly as in the reference and consume
// device-computed sums; everything that touches N particles is a kernel launch on the context's
// stream (kernels.hip).  There is no CPU fallback for any per-particle stage.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cctype>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <numeric>
#include <optional>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "beluga_mcl.h"
#include "kernels.h"
#include "map_build.h"

namespace {

using namespace mcl;

thread_local std::string g_create_error;

// algorithm/exponential_filter.hpp:32-44
struct ExponentialFilter {
  double alpha{0.}, output{0.};
  void reset() { output = 0.; }
  double operator()(double input) {
    output += (output == 0.) ? input : alpha * (input - output);
    return output;
  }
};

template <class T>
struct DeviceBuffer {
  T* ptr{nullptr};
  size_t count{0};
  hipError_t ensure(size_t n) {
    if (n <= count) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
    if (e == hipSuccess) count = n;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
};

struct ParticleSet {
  DeviceBuffer<double4> pose;
  DeviceBuffer<double> w;
  hipError_t ensure(size_t n) {
    if (const hipError_t e = pose.ensure(n); e != hipSuccess) return e;
    return w.ensure(n);
  }
  void release() {
    pose.release();
    w.release();
  }
  Particles view() const { return Particles{pose.ptr, w.ptr}; }
};

Pose2 pose_from(const double p[4]) { return Pose2{Rot2{p[0], p[1]}, p[2], p[3]}; }

// motion/differential_drive_model.hpp:129-154,167-173
double rotation_variance(const Rot2& r) {
  const Rot2 flipped = rot_mul(r, rot_exp(kPi));
  const double delta = std::min(std::abs(rot_log(r)), std::abs(rot_log(flipped)));
  return delta * delta;
}
DiffDriveSampler make_sampler(const Pose2& pose, const Pose2& prev, const mcl_diffdrive_params& a, int kind, double alpha5) {
  const double tx = pose.x - prev.x, ty = pose.y - prev.y;
  const double distance = std::sqrt(tx * tx + ty * ty);
  const double distance_variance = distance * distance;
  const Rot2 heading = rot_exp(std::atan2(ty, tx));
  const Rot2 first = distance > a.distance_threshold ? rot_mul(heading, rot_inverse(prev.r)) : Rot2{1.0, 0.0};
  DiffDriveSampler s{};
  s.kind = kind;
  s.first_c = first.c;
  s.first_s = first.s;
  if (kind == MCL_MOTION_STATIONARY) return s;  // stationary_model.hpp:53-61 ignores the control action
  if (kind == MCL_MOTION_OMNIDIRECTIONAL) {     // omnidirectional_drive_model.hpp:102-131
    const Rot2 rotation = rot_mul(pose.r, rot_inverse(prev.r));
    s.m1 = rot_log(rotation);
    s.s1 = std::sqrt(a.rotation_noise_from_rotation * rotation_variance(rotation) + a.rotation_noise_from_translation * distance_variance);
    s.mt = distance;
    s.st = std::sqrt(a.translation_noise_from_translation * distance_variance + a.translation_noise_from_rotation * rotation_variance(rotation));
    s.m2 = 0.0;
    s.s2 = std::sqrt(alpha5 * distance_variance + a.translation_noise_from_rotation * rotation_variance(rotation));
    return s;
  }
  const Rot2 second = rot_mul(rot_mul(pose.r, rot_inverse(prev.r)), rot_inverse(first));
  s.m1 = rot_log(first);
  s.s1 = std::sqrt(a.rotation_noise_from_rotation * rotation_variance(first) + a.rotation_noise_from_translation * distance_variance);
  s.mt = distance;
  s.st = std::sqrt(a.translation_noise_from_translation * distance_variance +
                   a.translation_noise_from_rotation * (rotation_variance(first) + rotation_variance(second)));
  s.m2 = rot_log(second);
  s.s2 = std::sqrt(a.rotation_noise_from_rotation * rotation_variance(second) + a.rotation_noise_from_translation * distance_variance);
  return s;
}

// Symmetric 3x3 eigen-decomposition (cyclic Jacobi): T = V sqrt(L)  (multivariate_normal_distribution.hpp:109-126).
bool covariance_to_transform(const double cov[9], double T[9]) {
  double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a[i][j] = cov[3 * i + j];
  for (int i = 0; i < 3; ++i)
    for (int j = i + 1; j < 3; ++j) {
      const double scale = std::max(std::abs(a[i][j]), std::abs(a[j][i]));
      if (std::abs(a[i][j] - a[j][i]) > 1e-12 * scale) return false;  // "not symmetric"
      if (!std::isfinite(a[i][j])) return false;
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
    if (!std::isfinite(a[j][j])) return false;
    if (a[j][j] < 0.0) {
      if (a[j][j] > -1e-14) a[j][j] = 0.0;
      else return false;  // "negative eigenvalues"
    }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[3 * i + j] = v[i][j] * std::sqrt(a[j][j]);
  return true;
}

}  // namespace

struct mcl_ctx {
  mcl_config cfg;
  std::string error;
  int device{0};
  hipStream_t stream{nullptr};
  bool own_stream{false};

  uint64_t capacity{0};
  uint64_t n{0};
  ParticleSet sets[2];
  int live{0};

  // map
  bool have_map{false};
  uint32_t W{0}, H{0};
  double resolution{0};
  Pose2 origin{}, origin_inverse{};
  OccupancyTraits traits{0, -1, 100};
  DeviceBuffer<float> d_field;
  DeviceBuffer<double> d_cube;  // pz^3 table of the field (+1 slot for out-of-grid beams)
  // palette form of the same table (FieldView::pal_*), built when the field has <= kMaxPalette distinct values
  DeviceBuffer<uint16_t> d_pal_idx;
  DeviceBuffer<double> d_pal_val;
  DeviceBuffer<uint32_t> d_pal_keys;
  uint32_t pal_count{0}, pal_pitch{0}, pal_base{0}, pal_bytes{0};
  DeviceBuffer<uint8_t> d_far_bits;   // FieldView::far_bits: tiles of d_pal_idx uniformly equal to the table's most common entry
  DeviceBuffer<uint32_t> d_far_votes;
  uint32_t far_row_bytes{0}, far_bytes{0}, far_entry{0};
  DeviceBuffer<uint8_t> d_far_linear;  // FieldView::far_linear: the same bits by the tiles' linear index
  uint32_t far_linear_bytes{0};
  uint64_t far_tiles{0};  // number of set bits' worth of tiles voted for far_entry (0 = no bitmap)
  DeviceBuffer<int8_t> d_cells;
  DeviceBuffer<uint32_t> d_nonfree_bits;  // beam model: 1 bit per cell
  DeviceBuffer<uint32_t> d_free;
  uint64_t n_free{0};
  std::vector<float> h_field;
  DeviceBuffer<uint32_t> d_field_scratch;  // device field build: uint16 column distances + int16 offsets per cell
  bool field_built_on_device{false};
  uint64_t comm_bytes_out{0};   // bytes this rank has handed to the transport (all-gather contributions + all-to-all sends), cumulative
  uint64_t comm_collectives{0}; // collectives called, cumulative
  uint64_t comm_host_syncs{0};  // host synchronisations inside sharded update cycles, cumulative
  uint64_t comm_overflows{0};   // cycles whose fixed-capacity exchange overflowed and ran again with exact counts
  uint64_t comm_ranks_seen{0};  // ranks the communicator reports (ncclCommCount), or the attached world size
  int comm_backend{0};          // 0 none, 1 caller's transport, 2 RCCL inside the library
  uint64_t cluster_cells{0};  // occupied cells of the last cluster_based_estimate on this context (before the merge over shards)
  double field_build_ms{0.0};

  // scan: staged in mapped pinned host memory and pulled into d_points by a kernel of the cycle (no copy-engine hand-off)
  DeviceBuffer<double> d_points;
  DeviceBuffer<double> d_beam_points;  // beam model: per-beam terms (kBeamPointDoubles per beam)
  DeviceBuffer<double> d_beam_table;   // beam model: 4 doubles per squared cell distance of a hit (launch_beam_table), built by mcl_set_map
  uint32_t beam_table_count{0};
  bool beam_table_ready{false};        // d_beam_table holds the table of the current map (built lazily: do_reweight)
  double* h_points{nullptr};   // pinned, mapped
  double* hd_points{nullptr};  // the same memory as the device sees it
  double scan_extent{0.0};     // max |x| + |y| of the uploaded scan points (NaN if any is NaN)
  size_t h_points_cap{0};
  hipEvent_t points_event{nullptr};  // recorded behind the kernel that pulls h_points
  bool points_in_flight{false}, points_event_valid{false};

  // reductions / scans
  DeviceBuffer<double> d_chunk;      // [12][stride]
  uint32_t chunk_stride{0};
  DeviceBuffer<double> d_scalars;    // 32 doubles
  double* h_scalars{nullptr};        // pinned, 32 doubles
  double* hd_scalars{nullptr};       // the same memory as the device sees it: kernels mirror their scalar results into it
  DeviceBuffer<double> d_cdf;
  DeviceBuffer<double4> d_cloud;    // mcl_sample_particle_cloud staging
  DeviceBuffer<double> d_est_partials;  // [9][ceil(n / 256)] estimate sums left by the draw kernel
  // The fixed-size cycle's completion word (Completion): ticket in d_scalars[27], word in h_scalars[31]; done_seq counts the
  // cycles that armed it, done_armed: this cycle's last kernel carries it.
  uint64_t done_seq{0};
  bool done_armed{false};
  // Host time of mcl_update on the one-synchronisation path, running totals in ns (mcl_get_counter host_ns_*): entry -> first
  // kernel enqueued, -> last kernel enqueued, the wait for the cycle, the rest until the return; host_cycles counts them.
  uint64_t host_ns[4]{0, 0, 0, 0};
  uint64_t host_cycles{0};
  uint64_t noise_ahead_used{0};  // launches of k_propagate that found their normals drawn ahead (mcl_get_counter)
  DeviceBuffer<double> d_cloud_w;
  DeviceBuffer<double> d_cdf_tree;  // sampled levels of the 16-ary search tree over d_cdf (CdfTree)
  DeviceBuffer<unsigned long long> d_scan_state;  // k_normalize_cdf: ticket word + the chunk sums' granules (kScanStateWords, zeroed once)
  uint32_t scan_epoch{0};           // of the last k_normalize_cdf launch on that state
  DeviceBuffer<double> d_lf_wsum;   // sums of the new weights per workgroup of the LF patch kernel (PatchStats::weight_sums)
  uint32_t lf_wsum_count{0};        // how many the last reweight left (0: none; consumed by the normalisation right behind it)
  // k_noise_ahead: the propagation normals of step `noise_step` for the particles [noise_offset, noise_offset + noise_n) of the global index
  // space, drawn behind the previous cycle (they depend on nothing else: any set of that size at that step may use them)
  DeviceBuffer<double> d_noise;
  uint32_t noise_step{0};
  uint64_t noise_n{0}, noise_offset{0}, noise_seed{0};
  // launch_order_ahead: the spatial order (sort_scratch().perm) of the particles as the propagation of step `order_step` will leave them, if the
  // control action of that step is the predicted one (order_sampler); the frame it was computed in
  bool order_valid{false};
  uint32_t order_step{0};
  uint64_t order_n{0};
  DiffDriveSampler order_sampler{};
  uint32_t order_layout{0};
  bool order_ready{false};      // this cycle's propagation found it usable: do_reweight skips the ordering passes
  uint64_t order_ahead_used{0}, order_ahead_missed{0};
  DiffDriveSampler last_sampler{};  // of the last propagation (the prediction for the next one)
  bool cdf_divides{false};          // the last normalisation left the weights undivided: the CDF kernel right behind it divides (do_normalize)
  bool weights_unit{false};         // every weight of the live set is exactly 1.0: set by what writes them all (initialisation, resampling,
                                    // particle_traits.hpp:105), cleared by whatever else touches a weight
  CdfTree cdf_tree() const { return make_cdf_tree(d_cdf.ptr, d_cdf_tree.ptr, n); }
  // mcl_set_map_async: the next map, its likelihood field being built on a worker thread (state 1) or built and waiting for its swap (2)
  struct PendingMap {
    std::vector<int8_t> cells;
    uint32_t W{0}, H{0};
    double resolution{0.0}, origin[4]{1.0, 0.0, 0.0, 0.0};
    int8_t traits[3]{0, -1, 100};
    std::vector<float> field;  // (likelihood-field models with the host build; empty otherwise: nothing to build ahead)
    std::thread worker;
    std::atomic<int> state{0};
  };
  PendingMap* pending_map{nullptr};

  // KLD
  DeviceBuffer<unsigned long long> d_hashes;
  DeviceBuffer<unsigned long long> d_table_keys;
  DeviceBuffer<unsigned int> d_table_first;
  uint64_t table_capacity{0};
  DeviceBuffer<uint32_t> d_flags, d_uchunk;  // flags[kld_capacity]; uchunk[2][kld_chunks]
  uint64_t kld_capacity{0};                  // candidates one KLD pass can hold (a shard sees the GLOBAL candidate stream)
  uint32_t kld_chunks{0};
  // state of the running KLD pass (do_resample, or mcl_kld_begin / mcl_kld_feed for the sharded driver)
  uint64_t kld_pos{0}, kld_table_slots{0};
  int kld_flip{0};
  DeviceBuffer<unsigned long long> d_kld_scalars;  // [0]=first_fail, [1]=beam steps; as u32 view: k words at [4..]
  unsigned long long* h_kld_scalars{nullptr};      // pinned, 8 words

  // cluster_based_estimate scratch
  DeviceBuffer<double> d_cell_f64;             // table wsum[cap_t] | list wsum[m_cap] | list state[4*m_cap]
  DeviceBuffer<unsigned int> d_cell_u32;       // table count[cap_t] cluster[cap_t] | list first,count,slot,cluster[m_cap] | size
  DeviceBuffer<unsigned long long> d_cell_u64; // list key[m_cap]
  DeviceBuffer<double> d_cell_exchange;        // shards: this rank's cell records | the gathered records of all ranks
  int estimate_kind{0};
  mcl_cluster_params cluster_params{0.20, 0.524, 0.90};

  // host-side filter state (amcl_core.hpp:206-232)
  ExponentialFilter slow, fast;
  bool have_latest{false};
  Pose2 latest{};
  uint64_t every_n_current{0};
  bool force_update{true};
  bool have_window{false};
  Pose2 window0{}, window1{};
  uint32_t step{0};
  bool have_pivot{false};
  double pivot[2]{0, 0};
  Tuning tuning{};  // mcl_set_option / BELUGA_MCL_* at mcl_create (A/B measurements, tests)
  // Key frame of the spatial ordering (kernels.h KeyFrame): the last estimate of the set, when the host has one.
  bool have_cloud_estimate{false};
  double cloud_mean[3]{0, 0, 0};   // x, y, theta
  double cloud_sigma[3]{0, 0, 0};  // standard deviations of x, y, theta
  uint64_t lf_fast_launches{0};    // launches of the FMA variant of the LF kernel (mcl_get_counter)
  // The LDS-patch kernel reports how many beam groups it planned and how many went through a patch (running totals in
  // d_scalars[24..27), mirrored to h_scalars[28..30)); a launch that found few sends the next ones to the gather kernel,
  // with a probe every 16th launch (option lf_patch = 1).
  uint64_t lf_patch_launches{0};
  uint64_t lf_queue_launches{0};  // of which by resident workgroups that take their blocks from a queue (k_reweight_lf_patch<true>)
  uint64_t patch_seen_planned{0}, patch_seen_through{0};
  bool patch_useful{true};
  int patch_probe_in{0};
  // What this cycle's LF launch uses, decided once per cycle (the ordering pass in front of it depends on it):
  // patches = the LDS-patch kernel; beams = a dispersed set goes to k_reweight_lf_beams (wave per particle, no ordering).
  struct LfMode { bool decided, patches, beams; } lf_mode{false, false, false};
  uint64_t lf_beams_launches{0};   // launches of k_reweight_lf_beams (mcl_get_counter)
  uint64_t lf_far_launches{0};     // launches of the gather kernel with the far-tile bitmap (dispersed sets)
  uint64_t lf_far_beams_launches{0};  // those of them that were k_reweight_lf_far_beams (lf_dispersed = 2)
  // scratch of the spatial ordering
  DeviceBuffer<uint32_t> d_route_u32;           // scratch of mcl_route_targets
  DeviceBuffer<uint32_t> d_sort_u32;            // keys[cap] perm[cap] table[1024 * nblocks] totals[1024] bases[1024] flags[16]
  DeviceBuffer<unsigned long long> d_sort_u64;  // keyidx[cap]
  DeviceBuffer<double> d_sort_f64;              // frame[8] bbox[8 + 6*nblocks] partial[...]

  // particle shards: communicator + scratch of the exchange (sharded_update)
  bool have_comm{false};
  uint32_t comm_rank{0}, comm_world{1};
  mcl_transport transport{};
  void* rccl_comm{nullptr};                 // ncclComm_t when the transport is the built-in RCCL one
  struct RcclUserStorage { void* comm; uint32_t rank, world; } rccl_user{nullptr, 0, 1};
  DeviceBuffer<double> d_comm_f64;          // [0..8) locals | gathered scalars | ends, offsets | estimate gather
  DeviceBuffer<long long> d_comm_i64;       // counts[world] | gathered counts[world * world]
  DeviceBuffer<double> d_targets, d_send_targets, d_requests_in, d_replies_out, d_replies_in;
  DeviceBuffer<uint32_t> d_route_order;
  // KLD-adaptive resampling over shards: this rank's slices of the candidate blocks, hash staging, the re-balanced shard
  DeviceBuffer<double> d_cand_states, d_new_shard;
  DeviceBuffer<unsigned long long> d_cand_hashes, d_block_hashes;
  uint64_t global_n{0};                     // particles of the logical filter over all shards (0: max_particles)
  bool global_n_unknown{false};             // the caller loaded this shard itself (mcl_set_particles): counts are gathered first
  double* h_comm{nullptr};                  // pinned staging for the exchange's host reads / uploads
  unsigned char* h_cells{nullptr};          // cluster_based_estimate: the list of occupied cells in mapped pinned memory
  unsigned char* hd_cells{nullptr};         // ... its device address

  // profiling: 0 = off, 1 = the sensor kernel only (two events per cycle), 2 = every stage
  int profile{0};
  uint32_t profile_tick{0};  // reweight launches since profiling was switched on
  hipEvent_t ev[MCL_NUM_STAGES][2]{};
  bool ev_pending[MCL_NUM_STAGES]{};
  double prof_ms[MCL_NUM_STAGES]{};
  uint64_t prof_count[MCL_NUM_STAGES]{};

  Particles cur() const { return sets[live].view(); }
  Particles other() const { return sets[live ^ 1].view(); }
  double* chunk_row(int k) { return d_chunk.ptr + static_cast<size_t>(k) * chunk_stride; }
  FieldView field_view() const {
    return FieldView{d_field.ptr, W, H, 1. / resolution, origin_inverse, static_cast<float>(1. / cfg.lf.max_laser_distance),
                     d_cube.ptr, cfg.sensor_kind == MCL_SENSOR_LIKELIHOOD_FIELD_PROB ? 1 : 0,
                     pal_count ? d_pal_idx.ptr : nullptr, d_pal_val.ptr, pal_count, pal_pitch, pal_base, pal_bytes,
                     pal_count && far_tiles ? d_far_bits.ptr : nullptr, far_row_bytes, far_bytes, far_entry,
                     pal_count && far_tiles && far_linear_bytes ? d_far_linear.ptr : nullptr, far_linear_bytes};
  }
  SortScratch sort_scratch() {
    SortScratch s{};
    const size_t nblocks = num_chunks(capacity);
    s.keys = d_sort_u32.ptr;
    s.perm = s.keys + capacity;
    s.table = s.perm + capacity;
    s.totals = s.table + kSortDigits * nblocks;
    s.keyidx = d_sort_u64.ptr;
    s.frame = reinterpret_cast<KeyFrame*>(d_sort_f64.ptr);
    s.bbox = d_sort_f64.ptr + 8;
    s.partial = s.bbox + 8 + 6 * nblocks;
    return s;
  }
  GridView grid_view() const { return GridView{d_cells.ptr, W, H, resolution, origin, origin_inverse, traits.free_value}; }
};

namespace {

mcl_status fail(mcl_ctx* ctx, mcl_status code, const std::string& msg) {
  if (ctx) ctx->error = msg;
  else g_create_error = msg;
  return code;
}

#define MCL_HIP(ctx, expr)                                                                               \
  do {                                                                                                   \
    const hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                              \
      return fail(ctx, e_ == hipErrorOutOfMemory ? MCL_ERR_OUT_OF_MEMORY : MCL_ERR_HIP,                  \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                    \
    }                                                                                                    \
  } while (0)

#define MCL_REQUIRE(ctx, cond, msg) \
  do {                              \
    if (!(cond)) return fail(ctx, MCL_ERR_INVALID_ARGUMENT, msg); \
  } while (0)

mcl_status bind_device(mcl_ctx* ctx) {
  MCL_HIP(ctx, hipSetDevice(ctx->device));
  return MCL_OK;
}

// Level 1 times the sensor kernel of every kProfileSampleEvery-th cycle only: an event record costs ~5 us of stream time,
// and this is the level a timed run uses.
constexpr uint32_t kProfileSampleEvery = 4;
bool stage_timed(const mcl_ctx* ctx, int stage) {
  return ctx->profile >= 2 || (ctx->profile == 1 && stage == MCL_STAGE_SENSOR_KERNEL && ctx->profile_tick % kProfileSampleEvery == 0);
}
void stage_begin(mcl_ctx* ctx, int stage) {
  if (!stage_timed(ctx, stage)) return;
  (void)hipEventRecord(ctx->ev[stage][0], ctx->stream);
}
void stage_end(mcl_ctx* ctx, int stage) {
  if (!stage_timed(ctx, stage)) return;
  (void)hipEventRecord(ctx->ev[stage][1], ctx->stream);
  ctx->ev_pending[stage] = true;
}
// Call after the stream has been synchronised.
void stage_collect(mcl_ctx* ctx) {
  ctx->points_in_flight = false;  // the stream has been synchronised: the kernel that pulled the scan is done
  if (!ctx->profile) return;
  for (int s = 0; s < MCL_NUM_STAGES; ++s) {
    if (!ctx->ev_pending[s]) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev[s][0], ctx->ev[s][1]) == hipSuccess) {
      ctx->prof_ms[s] += ms;
      ctx->prof_count[s] += 1;
    }
    ctx->ev_pending[s] = false;
  }
}

// The end of a cycle whose last kernel is the draw's k_final_rows.  Armed (done_armed): that kernel stores the cycle's number
// to a word of mapped host memory behind everything the cycle mirrored there; the host watches the word - the kernel's own
// store arrives a few microseconds before the stream's completion signal has been raised and noticed.  The stream is in order:
// everything before that kernel is complete too, and later calls that need the stream itself idle still synchronise it.  Not
// armed (profiling on, another path), or the word does not show up within 20 ms: hipStreamSynchronize.
mcl_status wait_for_cycle(mcl_ctx* ctx) {
  if (ctx->done_armed) {
    ctx->done_armed = false;
    const volatile uint64_t* word = reinterpret_cast<const volatile uint64_t*>(ctx->h_scalars + 31);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; ++spins) {
      if (*word == ctx->done_seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        // (now and then the runtime gets to see an idle stream all the same: it retires its bookkeeping of finished commands there)
        if ((ctx->done_seq & 0xFFu) == 0) break;
        return MCL_OK;
      }
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#elif defined(__aarch64__)
      asm volatile("yield" ::: "memory");
#else
      std::this_thread::yield();
#endif
      if ((spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
  }
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MCL_OK;
}

mcl_status ensure_capacity(mcl_ctx* ctx, uint64_t cap) {
  for (auto& set : ctx->sets) MCL_HIP(ctx, set.ensure(cap));
  const uint32_t chunks = num_chunks(cap) + 1;
  ctx->chunk_stride = chunks;
  MCL_HIP(ctx, ctx->d_chunk.ensure(static_cast<size_t>(12) * chunks));
  MCL_HIP(ctx, ctx->d_cdf.ensure(cap));
  MCL_HIP(ctx, ctx->d_cdf_tree.ensure(cdf_tree_doubles(cap)));
  MCL_HIP(ctx, ctx->d_lf_wsum.ensure(cap / 448 + 2));
  if (!ctx->d_scan_state.ptr) {
    MCL_HIP(ctx, ctx->d_scan_state.ensure(kScanStateWords));
    MCL_HIP(ctx, hipMemset(ctx->d_scan_state.ptr, 0, kScanStateWords * sizeof(unsigned long long)));
  }
  ctx->capacity = cap;
  {
    static_assert(sizeof(KeyFrame) <= 8 * sizeof(double), "the key frame sits in the first 8 doubles of d_sort_f64");
    const size_t table = static_cast<size_t>(kSortDigits) * num_chunks(cap);
    MCL_HIP(ctx, ctx->d_sort_u32.ensure(2 * cap + table + 2 * kSortDigits + 16));
    MCL_HIP(ctx, ctx->d_sort_u64.ensure(cap));
    MCL_HIP(ctx, ctx->d_sort_f64.ensure(8 + 8 + 6 * static_cast<size_t>(chunks) + kLfMaxSegments * std::min<uint64_t>(cap, kLfSegmentedBelow)));
  }
  return MCL_OK;
}

mcl_status ensure_kld(mcl_ctx* ctx) {
  // A shard context (shard_capacity set) checks the KLD bound over the global candidate stream, not only its slice.
  const uint64_t cap = std::max<uint64_t>(ctx->capacity, ctx->cfg.shard_capacity ? ctx->cfg.amcl.max_particles : 0);
  MCL_HIP(ctx, ctx->d_hashes.ensure(cap));
  MCL_HIP(ctx, ctx->d_flags.ensure(cap));
  ctx->kld_chunks = num_chunks(cap) + 1;
  MCL_HIP(ctx, ctx->d_uchunk.ensure(static_cast<size_t>(2) * ctx->kld_chunks));
  uint64_t tc = 1024;
  while (tc < 2 * cap) tc <<= 1;
  MCL_HIP(ctx, ctx->d_table_keys.ensure(tc));
  MCL_HIP(ctx, ctx->d_table_first.ensure(tc));
  ctx->table_capacity = tc;
  ctx->kld_capacity = cap;
  return MCL_OK;
}

// take_while_kld (views/take_while_kld.hpp:72-88,112-137) as a running pass over the candidate stream: kld_begin, then
// kld_process(cnt) for every block of candidates whose hashes were appended to d_hashes[kld_pos ...).
mcl_status kld_begin(mcl_ctx* ctx) {
  if (ctx->table_capacity == 0) {
    if (const mcl_status s = ensure_kld(ctx)) return s;
  }
  MCL_HIP(ctx, hipMemsetAsync(ctx->d_kld_scalars.ptr, 0xFF, sizeof(unsigned long long), ctx->stream));  // first_fail = ~0
  uint32_t* kwords = reinterpret_cast<uint32_t*>(ctx->d_kld_scalars.ptr + 4);                          // [0]=k_base,[1]=k_total
  MCL_HIP(ctx, hipMemsetAsync(kwords, 0, 2 * sizeof(uint32_t), ctx->stream));
  ctx->kld_pos = 0;
  ctx->kld_table_slots = 0;
  ctx->kld_flip = 0;
  return MCL_OK;
}

// The open-addressing table is sized for the candidates seen so far (load <= 1/2), not for max_particles: a tight
// cloud stops after ~min_particles candidates and must not pay for clearing a table of 2 * max_particles slots.
// When the next block outgrows it, it is cleared at the larger size and the earlier hashes are re-inserted.
mcl_status kld_grow_table(mcl_ctx* ctx, uint64_t candidates) {
  uint64_t want = 1024;
  while (want < 2 * candidates) want <<= 1;
  want = std::min<uint64_t>(want, ctx->table_capacity);
  if (want <= ctx->kld_table_slots) return MCL_OK;
  ctx->kld_table_slots = want;
  MCL_HIP(ctx, hipMemsetAsync(ctx->d_table_keys.ptr, 0xFF, want * sizeof(unsigned long long), ctx->stream));
  MCL_HIP(ctx, hipMemsetAsync(ctx->d_table_first.ptr, 0xFF, want * sizeof(unsigned int), ctx->stream));
  launch_kld_insert(ctx->stream, ctx->d_hashes.ptr, 0, ctx->kld_pos, KldTable{ctx->d_table_keys.ptr, ctx->d_table_first.ptr, want});
  return MCL_OK;
}

// -> *first_fail = index of the first candidate failing the predicate (it is dropped), ~0 if all cnt candidates pass.
mcl_status kld_process(mcl_ctx* ctx, uint64_t cnt, uint64_t* first_fail) {
  const mcl_amcl_params& a = ctx->cfg.amcl;
  const KldTable table{ctx->d_table_keys.ptr, ctx->d_table_first.ptr, ctx->kld_table_slots};
  uint32_t* kwords = reinterpret_cast<uint32_t*>(ctx->d_kld_scalars.ptr + 4);
  launch_kld_insert(ctx->stream, ctx->d_hashes.ptr, ctx->kld_pos, cnt, table);
  launch_kld_scan(ctx->stream, ctx->d_hashes.ptr, ctx->kld_pos, cnt, table, ctx->d_flags.ptr, ctx->d_uchunk.ptr,
                  ctx->d_uchunk.ptr + ctx->kld_chunks, kwords + ctx->kld_flip, kwords + (ctx->kld_flip ^ 1), a.min_particles,
                  a.kld_epsilon, a.kld_z, ctx->d_kld_scalars.ptr);
  MCL_HIP(ctx, hipGetLastError());
  MCL_HIP(ctx, hipMemcpyAsync(ctx->h_kld_scalars, ctx->d_kld_scalars.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *first_fail = ctx->h_kld_scalars[0];
  ctx->kld_pos += cnt;
  ctx->kld_flip ^= 1;
  return MCL_OK;
}

mcl_status rebuild_cube(mcl_ctx* ctx, const float* h_field) {
  const uint64_t cells = static_cast<uint64_t>(ctx->W) * ctx->H;
  const float unknown_value = static_cast<float>(1. / ctx->cfg.lf.max_laser_distance);
  const int prob = ctx->cfg.sensor_kind == MCL_SENSOR_LIKELIHOOD_FIELD_PROB ? 1 : 0;
  MCL_HIP(ctx, ctx->d_cube.ensure(cells + 1));
  launch_cube_table(ctx->stream, ctx->d_field.ptr, cells, unknown_value, ctx->d_cube.ptr, prob);
  MCL_HIP(ctx, hipGetLastError());
  // Palette: the distinct values of the field (a distance map quantised to cell offsets has a few hundred).
  ctx->pal_count = 0;
  ctx->far_tiles = 0;
  const uint64_t tiles_x = (ctx->W + 7) / 8 + 2, tiles_y = (ctx->H + 7) / 8 + 2;  // one border tile on every side
  const uint32_t pal_base = ((ctx->H + 2) * 4u + 7u) & ~7u;                       // the kernel's row-offset table comes first in LDS
  if (h_field && tiles_x * tiles_y * 128 < (1ull << 31) && ctx->W < (1u << 26) && pal_base + 8 <= 65536) {
    const size_t max_entries = std::min<size_t>(kMaxPalette, (65536 - pal_base) / 8);
    std::vector<uint32_t> keys;
    {
      std::unordered_set<uint32_t> seen;
      uint32_t bits;
      std::memcpy(&bits, &unknown_value, sizeof bits);
      seen.insert(bits);
      uint32_t last = bits;
      for (uint64_t i = 0; i < cells && seen.size() <= max_entries; ++i) {
        std::memcpy(&bits, h_field + i, sizeof bits);
        if (bits != last) {
          seen.insert(bits);
          last = bits;
        }
      }
      if (seen.size() <= max_entries) keys.assign(seen.begin(), seen.end());
    }
    if (!keys.empty()) {
      std::sort(keys.begin(), keys.end());
      MCL_HIP(ctx, ctx->d_pal_keys.ensure(keys.size()));
      MCL_HIP(ctx, ctx->d_pal_val.ensure(keys.size()));
      MCL_HIP(ctx, ctx->d_pal_idx.ensure(tiles_x * tiles_y * 64));
      MCL_HIP(ctx, hipMemcpyAsync(ctx->d_pal_keys.ptr, keys.data(), keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
      launch_palette_table(ctx->stream, ctx->d_field.ptr, ctx->W, ctx->H, unknown_value, ctx->d_pal_keys.ptr,
                           static_cast<uint32_t>(keys.size()), prob, ctx->d_pal_idx.ptr, ctx->d_pal_val.ptr, pal_base);
      MCL_HIP(ctx, hipGetLastError());
      MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // keys is a local
      ctx->pal_count = static_cast<uint32_t>(keys.size());
      ctx->pal_pitch = static_cast<uint32_t>(tiles_x * 128);
      ctx->pal_base = pal_base;
      ctx->pal_bytes = static_cast<uint32_t>(tiles_x * tiles_y * 128);
      // Far tiles: the entry most tiles are uniformly equal to, and the bitmap of those tiles (FieldView::far_bits).
      ctx->far_tiles = 0;
      const uint32_t row_bytes = static_cast<uint32_t>((tiles_x + 7) / 8);
      const uint32_t far_bytes = static_cast<uint32_t>((static_cast<uint64_t>(row_bytes) * tiles_y + 15) & ~15ull);
      if (tiles_x * tiles_y < (1ull << 31) && far_bytes <= 48 * 1024 && row_bytes < (1u << 13)) {
        MCL_HIP(ctx, ctx->d_far_votes.ensure(keys.size()));
        MCL_HIP(ctx, ctx->d_far_bits.ensure(far_bytes));
        launch_far_tile_votes(ctx->stream, ctx->d_pal_idx.ptr, static_cast<uint32_t>(tiles_x * tiles_y), pal_base,
                              static_cast<uint32_t>(keys.size()), ctx->d_far_votes.ptr);
        std::vector<uint32_t> votes(keys.size());
        MCL_HIP(ctx, hipMemcpyAsync(votes.data(), ctx->d_far_votes.ptr, votes.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const size_t best = static_cast<size_t>(std::max_element(votes.begin(), votes.end()) - votes.begin());
        if (votes[best] * 8ull >= tiles_x * tiles_y) {  // worth a test per look-up from one tile in eight
          ctx->far_entry = pal_base + static_cast<uint32_t>(best) * 8u;
          ctx->far_row_bytes = row_bytes;
          ctx->far_bytes = far_bytes;
          launch_far_tile_bits(ctx->stream, ctx->d_pal_idx.ptr, static_cast<uint32_t>(tiles_x), static_cast<uint32_t>(tiles_y), ctx->far_entry,
                               row_bytes, far_bytes, ctx->d_far_bits.ptr);
          ctx->far_linear_bytes = static_cast<uint32_t>(((tiles_x * tiles_y + 7) / 8 + 15) & ~15ull);
          MCL_HIP(ctx, ctx->d_far_linear.ensure(ctx->far_linear_bytes));
          launch_far_tile_bits_linear(ctx->stream, ctx->d_pal_idx.ptr, static_cast<uint32_t>(tiles_x * tiles_y), ctx->far_entry,
                                      ctx->far_linear_bytes, ctx->d_far_linear.ptr);
          MCL_HIP(ctx, hipGetLastError());
          ctx->far_tiles = votes[best];
        }
      }
    }
  }
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MCL_OK;
}

// Stages the scan in mapped pinned memory; a kernel of the cycle pulls it into d_points (pull_scan_args / launch_pull_scan).
mcl_status stage_points(mcl_ctx* ctx, const double* pts, uint64_t B) {
  if (B == 0) return MCL_OK;
  MCL_HIP(ctx, ctx->d_points.ensure(2 * B));
  if (ctx->cfg.sensor_kind == MCL_SENSOR_BEAM) MCL_HIP(ctx, ctx->d_beam_points.ensure(kBeamPointDoubles * B));
  if (ctx->points_in_flight) {  // an earlier call may still be reading the staging buffer
    if (ctx->points_event_valid) MCL_HIP(ctx, hipEventSynchronize(ctx->points_event));
    else MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->points_in_flight = false;
  }
  if (ctx->h_points_cap < 2 * B) {
    if (ctx->h_points) (void)hipHostFree(ctx->h_points);
    ctx->h_points = nullptr;
    ctx->hd_points = nullptr;
    ctx->h_points_cap = 0;
    MCL_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_points), 2 * B * sizeof(double), hipHostMallocMapped));
    MCL_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->hd_points), ctx->h_points, 0));
    ctx->h_points_cap = 2 * B;
  }
  double extent = 0.0;  // maximum of |x| + |y| over the scan; a NaN point makes it NaN (and every comparison with it false)
  bool poisoned = false;
  for (uint64_t i = 0; i < 2 * B; i += 2) {
    const double x = pts[i], y = pts[i + 1];
    ctx->h_points[i] = x;
    ctx->h_points[i + 1] = y;
    const double e = std::abs(x) + std::abs(y);
    if (e != e) poisoned = true;
    extent = e > extent ? e : extent;
  }
  ctx->scan_extent = poisoned ? std::numeric_limits<double>::quiet_NaN() : extent;
  return MCL_OK;
}
// Call right behind the launch that pulls h_points.  with_event: a stage-level call that may return before the stream is
// synchronised (mcl_update always ends with a synchronisation, which clears the flag: no event record on its stream).
void points_pulled(mcl_ctx* ctx, bool with_event) {
  if (with_event) (void)hipEventRecord(ctx->points_event, ctx->stream);
  ctx->points_event_valid = with_event;
  ctx->points_in_flight = true;
}

// The frame of the ordering keys for the set as it will be AFTER this propagation: the last estimate moved by the mean
// motion, spans widened by the motion noise.  Only the balance of the key's bins depends on it.
// KeyFrame::layout of the next ordering: position-major for likelihood-field sets reported as dispersed (their gather kernel
// walks the order region by region, kernels.hip: k_reweight_lf_palette<true, true>), heading-major otherwise.
uint32_t key_layout(const mcl_ctx* ctx) {
  const uint32_t curve = ctx->tuning.key_curve ? 0u : 2u;  // heading-major keys: Hilbert curve (default) / Morton order
  if (ctx->tuning.key_layout >= 0) return (ctx->tuning.key_layout ? 1u : 0u) | curve;
  return (ctx->cfg.sensor_kind != MCL_SENSOR_BEAM && ctx->tuning.lf_patch == 1 && !ctx->patch_useful && ctx->tuning.lf_far_tiles != 0 &&
                  ctx->far_tiles != 0
              ? 1u
              : 0u) |
         curve;
}
// moves: how often the set's centre is moved by `motion` (2: the frame of the cycle AFTER the one that is running - launch_order_ahead -, whose
// set is the remembered one moved twice; its spread grows once: the resampling in between takes it back to where it was).
bool predict_key_frame(const mcl_ctx* ctx, const DiffDriveSampler* motion, KeyFrame* out, int moves = 1) {
  out->layout = key_layout(ctx);
  if (!ctx->have_cloud_estimate) return false;
  double x = ctx->cloud_mean[0], y = ctx->cloud_mean[1], t = ctx->cloud_mean[2];
  double sx = ctx->cloud_sigma[0], sy = ctx->cloud_sigma[1], st = ctx->cloud_sigma[2];
  if (motion && motion->kind != MCL_MOTION_STATIONARY) {
    for (int move = 1; move < moves; ++move) {  // (the centre alone)
      const double heading = t + (motion->kind == MCL_MOTION_DIFFERENTIAL ? motion->m1 : std::atan2(motion->first_s, motion->first_c));
      x += motion->mt * std::exp(-0.5 * st * st) * std::cos(heading);
      y += motion->mt * std::exp(-0.5 * st * st) * std::sin(heading);
      t += motion->kind == MCL_MOTION_DIFFERENTIAL ? motion->m1 + motion->m2 : motion->m1;
    }
    const double heading = t + (motion->kind == MCL_MOTION_DIFFERENTIAL ? motion->m1 : std::atan2(motion->first_s, motion->first_c));
    // every pose moves along ITS heading: the set's mean moves by the translation times the mean resultant length of the headings
    // (next to nothing for a set that points everywhere), and a heading error turns into a lateral one over the translation -
    // mt * sigma_theta for a narrow set, at most mt / sqrt(2) per axis for headings all around
    const double resultant = std::exp(-0.5 * st * st);
    x += motion->mt * resultant * std::cos(heading);
    y += motion->mt * resultant * std::sin(heading);
    t += motion->kind == MCL_MOTION_DIFFERENTIAL ? motion->m1 + motion->m2 : motion->m1;
    const double lateral = motion->mt * std::min(st, std::sqrt(0.5));
    const double noise2 = motion->st * motion->st + lateral * lateral + (motion->kind == MCL_MOTION_OMNIDIRECTIONAL ? motion->s2 * motion->s2 : 0.0);
    sx = std::sqrt(sx * sx + noise2);
    sy = std::sqrt(sy * sy + noise2);
    st = std::sqrt(st * st + motion->s1 * motion->s1 + (motion->kind == MCL_MOTION_DIFFERENTIAL ? motion->s2 * motion->s2 : 0.0));
  } else if (motion) {
    sx = std::sqrt(sx * sx + 0.02 * 0.02);
    sy = std::sqrt(sy * sy + 0.02 * 0.02);
    st = std::sqrt(st * st + 0.02 * 0.02);
  }
  if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(t) && std::isfinite(sx) && std::isfinite(sy) && std::isfinite(st))) return false;
  // +- 4 sigma; a set reported as dispersed is closer to uniform than to normal: +- 2 sigma hold all of a uniform one
  const double spans = ctx->patch_useful ? 8.0 : 4.0;
  auto inverse_span = [spans](double sigma) { return sigma > 0.0 ? static_cast<float>(1.0 / (spans * sigma)) : 0.f; };
  auto sigma_span_t = [](double sigma) { return sigma > 0.0 ? static_cast<float>(1.0 / (8.0 * sigma)) : 0.f; };
  out->cx = x;
  out->cy = y;
  out->c0 = std::cos(t);
  out->s0 = std::sin(t);
  out->inv_x = inverse_span(sx);
  out->inv_y = inverse_span(sy);
  out->inv_t = sigma_span_t(std::min(st, kPi / 4.0));  // the heading bins never span more than the circle
  out->t_off = 0.f;
  if (ctx->tuning.key_warp && !(out->layout & 1u) && spans == 8.0) out->layout |= 4u;  // bins of equal mass over the +-4 sigma
  // How the 20 bits are split: a run of the curve is roughly a cube of bins, and what a workgroup's LDS patch has to absorb is
  // its extent in x (or y) PLUS its extent in heading times the scan's reach - so the split that minimises the sum of the two
  // bin sizes, in cells: 8 sigma_xy / res / 2^b  +  8 sigma_theta reach / res / 2^(20 - 2 b), b = 4 .. 6.
  out->bits_xy = 6;
  if (ctx->tuning.key_bits_xy >= 4 && ctx->tuning.key_bits_xy <= 6) {
    out->bits_xy = static_cast<uint32_t>(ctx->tuning.key_bits_xy);
  } else if (ctx->tuning.key_bits_xy == 0 && ctx->resolution > 0.0 && std::isfinite(ctx->scan_extent)) {
    const double reach = 0.5 * ctx->scan_extent / ctx->resolution;  // cells; scan_extent = max |x| + |y| of the scan, ~ sqrt 2 the longest beam
    const double span_xy = spans * std::max(sx, sy) / ctx->resolution, span_t = 8.0 * std::min(st, kPi / 4.0) * reach;
    double best = std::numeric_limits<double>::infinity();
    for (uint32_t b = 4; b <= 6; ++b) {
      const double cost = std::ldexp(span_xy, -static_cast<int>(b)) + std::ldexp(span_t, -static_cast<int>(20 - 2 * b));
      if (cost < best) {
        best = cost;
        out->bits_xy = b;
      }
    }
  }
  return true;
}
void remember_cloud_estimate(mcl_ctx* ctx, const mcl_estimate& est) {
  const double vx = est.covariance[0], vy = est.covariance[4], vt = est.covariance[8];
  ctx->cloud_mean[0] = est.pose[2];
  ctx->cloud_mean[1] = est.pose[3];
  ctx->cloud_mean[2] = std::atan2(est.pose[1], est.pose[0]);
  ctx->cloud_sigma[0] = vx > 0.0 ? std::sqrt(vx) : 0.0;
  ctx->cloud_sigma[1] = vy > 0.0 ? std::sqrt(vy) : 0.0;
  ctx->cloud_sigma[2] = std::isfinite(vt) ? (vt > 0.0 ? std::sqrt(vt) : 0.0) : kPi;  // infinite circular variance: all headings
  ctx->have_cloud_estimate = std::isfinite(ctx->cloud_mean[0]) && std::isfinite(ctx->cloud_mean[1]) && std::isfinite(ctx->cloud_mean[2]) &&
                             std::isfinite(ctx->cloud_sigma[0]) && std::isfinite(ctx->cloud_sigma[1]);
}

// Whether the next LF launch goes to the LDS-patch kernel (where its other preconditions hold): by the verdict of the last
// launch that has reported.  A dispersed set (global localisation) has no group that fits a patch, and the patch kernel's
// workgroups carry a wave that would then do nothing.
// synchronised: the stream is idle (the two 64-bit totals are consistent); otherwise the pair comes from the packed word the
// kernel stores last (low 32 bits of each total in one 8-byte store: never torn; differences are taken modulo 2^32).
void patch_totals(const mcl_ctx* ctx, uint64_t* planned, uint64_t* through, bool synchronised = false) {
  const volatile uint64_t* mirror = reinterpret_cast<const volatile uint64_t*>(ctx->h_scalars + 28);
  if (synchronised) {
    *planned = mirror[0];
    *through = mirror[1];
    return;
  }
  const uint64_t packed = mirror[2];
  *planned = packed & 0xFFFFFFFFull;
  *through = packed >> 32;
}
// A workgroup of the patch kernel needs its 448 poses within a patch (64 x 64 cells less the margins) and within a few
// hundredths of a radian.  From the last estimate's spread, taken as uniform (12 sigma_x sigma_y of area, sqrt(12) sigma_theta
// of heading, at most the circle): the number of poses in such a volume.  Below an eighth of a workgroup no probe is worth it.
bool hopelessly_sparse(const mcl_ctx* ctx) {
  if (!ctx->have_cloud_estimate) return false;
  const double side = 40.0 * ctx->resolution;
  const double area = 12.0 * ctx->cloud_sigma[0] * ctx->cloud_sigma[1];
  const double arc = std::min(2.0 * kPi, std::sqrt(12.0) * ctx->cloud_sigma[2]);
  const double volume = std::max(area, side * side) * std::max(arc, 0.05);
  const double poses = static_cast<double>(ctx->n) * (side * side * 0.05) / volume;
  return poses < 448.0 / 8.0;
}
bool wants_patches(mcl_ctx* ctx) {
  if (ctx->tuning.lf_patch == 0) return false;
  if (ctx->tuning.lf_patch != 1) return true;
  uint64_t planned, through;
  patch_totals(ctx, &planned, &through);
  if (planned != ctx->patch_seen_planned) {  // a launch has reported since the last look
    const uint64_t dp = (planned - ctx->patch_seen_planned) & 0xFFFFFFFFull, dt = (through - ctx->patch_seen_through) & 0xFFFFFFFFull;
    ctx->patch_seen_planned = planned;
    ctx->patch_seen_through = through;
    ctx->patch_useful = 4 * dt >= dp;
    if (!ctx->patch_useful) ctx->patch_probe_in = 16;
  }
  if (ctx->patch_useful) return true;
  if (--ctx->patch_probe_in <= 0) {
    ctx->patch_probe_in = 16;
    return !hopelessly_sparse(ctx);  // a probe (3.7 ms instead of 1.1 on 1M dispersed particles: not where it cannot succeed)
  }
  return false;
}

// Likelihood-field sets below the threshold of the ordered kernels (the larger of the two options: the ordering itself and
// the LF kernels' own crossover).
bool lf_set_is_small(const mcl_ctx* ctx) {
  return ctx->n < static_cast<uint64_t>(std::max(ctx->tuning.sort_min_particles, ctx->tuning.lf_small_particles));
}
// The LF launch of this cycle: the patch kernel, the gather kernel, or - for a set the patch kernel has reported as
// dispersed (no probe due) - the wave-per-particle kernel, which needs no ordering pass.  Decided once per cycle, before the
// propagation kernel (which emits the ordering keys); cleared by do_reweight.
void decide_lf_mode(mcl_ctx* ctx) {
  if (ctx->lf_mode.decided) return;
  ctx->lf_mode.decided = true;
  ctx->lf_mode.patches = false;
  ctx->lf_mode.beams = false;
  if (ctx->cfg.sensor_kind == MCL_SENSOR_BEAM) return;
  const bool palette = ctx->pal_count != 0 && ctx->tuning.lf_table == 0;
  if (ctx->tuning.lf_variant == kLfBeamLanes) {
    ctx->lf_mode.beams = palette;
    return;
  }
  if (ctx->tuning.lf_variant != kLfSortedLanes) return;
  ctx->lf_mode.patches = wants_patches(ctx);
  ctx->lf_mode.beams = !ctx->lf_mode.patches && ctx->tuning.lf_patch == 1 && ctx->tuning.lf_dispersed == 1 && !ctx->patch_useful && palette &&
                       !lf_set_is_small(ctx);
}

bool wants_ordering(const mcl_ctx* ctx) {
  if (ctx->n >= (1ull << 32)) return false;
  if (ctx->cfg.sensor_kind == MCL_SENSOR_BEAM) return ctx->n >= static_cast<uint64_t>(ctx->tuning.beam_sort_min_particles);
  if (ctx->n < static_cast<uint64_t>(ctx->tuning.sort_min_particles)) return false;
  if (ctx->lf_mode.decided && ctx->lf_mode.beams) return false;
  return ctx->tuning.lf_variant == kLfSortedLanes && !(lf_set_is_small(ctx) && ctx->pal_count != 0 && ctx->tuning.lf_table == 0);
}

// Is the control action that came close enough to the one the order was predicted with (launch_order_ahead)?  What matters is that the
// particles end up in the same ARRANGEMENT: a common shift does not change it, different noise scales or a translation along another
// heading do.  Generous bounds: a miss costs look-ups that fit no LDS patch, a fallback costs the ordering passes on the critical path.
bool samplers_close(const DiffDriveSampler& now, const DiffDriveSampler& predicted) {
  if (now.kind != predicted.kind) return false;
  auto ratio_ok = [](double a, double b) { return a <= 1.5 * b + 1e-3 && b <= 1.5 * a + 1e-3; };
  if (!ratio_ok(now.s1, predicted.s1) || !ratio_ok(now.st, predicted.st) || !ratio_ok(now.s2, predicted.s2)) return false;
  if (std::abs(now.mt - predicted.mt) > 0.3 * std::max(std::abs(predicted.mt), 0.02)) return false;
  const double turn_now = now.kind == MCL_MOTION_DIFFERENTIAL ? now.m1 + now.m2 : now.m1;
  const double turn_predicted = predicted.kind == MCL_MOTION_DIFFERENTIAL ? predicted.m1 + predicted.m2 : predicted.m1;
  if (std::abs(turn_now - turn_predicted) > 0.15) return false;
  // the direction of the translation in the robot's frame (differential: the first rotation; omnidirectional: `first`)
  const double heading_now = now.kind == MCL_MOTION_DIFFERENTIAL ? now.m1 : std::atan2(now.first_s, now.first_c);
  const double heading_predicted = predicted.kind == MCL_MOTION_DIFFERENTIAL ? predicted.m1 : std::atan2(predicted.first_s, predicted.first_c);
  const double apart = std::abs(std::remainder(heading_now - heading_predicted, 2.0 * kPi));
  return apart * std::max(std::abs(now.mt), std::abs(predicted.mt)) <= 0.05;  // (metres of lateral disagreement)
}

// fused (mcl_update): the scan staged by stage_points is pulled by the same kernel, and the ordering keys of the new poses
// come out of it when the host knows where the set is (*keys_emitted).
mcl_status do_propagate(mcl_ctx* ctx, const Pose2& pose, const Pose2& prev, uint32_t step, uint64_t scan_points = 0,
                        bool* keys_emitted = nullptr) {
  stage_begin(ctx, MCL_STAGE_PROPAGATE);
  const DiffDriveSampler sampler = make_sampler(pose, prev, ctx->cfg.motion, ctx->cfg.motion_kind, ctx->cfg.strafe_noise_from_translation);
  KeyFrame frame{};
  const SortScratch sort = ctx->sort_scratch();
  if (keys_emitted) decide_lf_mode(ctx);  // the fused cycle: the reweight follows, and the keys depend on its kernel
  // The order the previous cycle computed AHEAD for this step (launch_order_ahead) serves if the control action it predicted is close to the
  // one that came: then no keys, no ordering passes - the reweight follows the propagation at once.  Only locality depends on it.
  ctx->order_ready = false;
  bool use_ahead = false;
  if (keys_emitted && ctx->order_valid) {
    ctx->order_valid = false;
    use_ahead = ctx->tuning.order_ahead != 0 && ctx->order_step == step && ctx->order_n == ctx->n && wants_ordering(ctx) &&
                ctx->order_layout == key_layout(ctx) && samplers_close(sampler, ctx->order_sampler);
    if (use_ahead) ctx->order_ahead_used += 1;
    else ctx->order_ahead_missed += 1;
  }
  ctx->last_sampler = sampler;
  const bool keys = !use_ahead && keys_emitted && wants_ordering(ctx) && predict_key_frame(ctx, &sampler, &frame);
  // (the normals of this step, if the previous cycle left them: k_noise_ahead)
  const bool ahead = ctx->d_noise.ptr && ctx->noise_n >= ctx->n && ctx->n > 65536 && ctx->noise_step == step && ctx->noise_seed == ctx->cfg.seed &&
                     ctx->noise_offset == ctx->cfg.shard_offset;
  launch_propagate(ctx->stream, ctx->cur(), ctx->n, sampler, ctx->cfg.seed, step, ctx->cfg.shard_offset,
                   scan_points ? ctx->hd_points : nullptr, scan_points ? ctx->d_points.ptr : nullptr, static_cast<uint32_t>(2 * scan_points),
                   keys ? &sort : nullptr, keys ? &frame : nullptr, ahead ? ctx->d_noise.ptr : nullptr, ctx->noise_n);
  if (ahead) ctx->noise_ahead_used += 1;
  if (scan_points) points_pulled(ctx, false);
  if (keys_emitted) *keys_emitted = keys;
  ctx->order_ready = use_ahead;
  stage_end(ctx, MCL_STAGE_PROPAGATE);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status reweight_preconditions(mcl_ctx* ctx, uint64_t B) {
  if (!ctx->have_map) return fail(ctx, MCL_ERR_NOT_READY, "mcl_reweight: no map set");
  MCL_REQUIRE(ctx, B <= 0x7FFFFFFFull, "too many points");
  // (no likelihood-field kernel stages the whole scan in LDS any more: lf_variant 0 launches the lane-per-particle kernel of variant 1)
  if (ctx->cfg.sensor_kind == MCL_SENSOR_BEAM && !(ctx->n < (1ull << 32) && ctx->n >= static_cast<uint64_t>(ctx->tuning.beam_sort_min_particles)))
    // the beam model's small-set kernel (a wave per particle) stages the scan in 64 KB of dynamic LDS
    MCL_REQUIRE(ctx, B * sizeof(double2) <= 64 * 1024, "scan too large for the beam model's small-set kernel (4096 points)");
  return MCL_OK;
}

// points_staged: stage_points + the pull already happened (mcl_update); keys_ready: k_propagate emitted the ordering keys.
// want_weight_sums: the normalisation follows at once (mcl_update): the LF patch kernel leaves the sums of its workgroups' new
// weights in d_lf_wsum (ctx->lf_wsum_count of them; 0 if another kernel ran).
mcl_status do_reweight(mcl_ctx* ctx, const double* pts, uint64_t B, bool points_staged = false, bool keys_ready = false,
                       bool want_weight_sums = false) {
  ctx->lf_wsum_count = 0;
  if (const mcl_status s = reweight_preconditions(ctx, B)) return s;
  const bool unit_weights = ctx->weights_unit && ctx->tuning.lf_unit_weights != 0;
  ctx->weights_unit = false;
  if (!points_staged) {
    if (const mcl_status s = stage_points(ctx, pts, B)) return s;
    if (B) {
      launch_pull_scan(ctx->stream, ctx->hd_points, ctx->d_points.ptr, static_cast<uint32_t>(2 * B));
      points_pulled(ctx, true);
    }
  }
  stage_begin(ctx, MCL_STAGE_REWEIGHT);
  decide_lf_mode(ctx);
  const mcl_ctx::LfMode mode = ctx->lf_mode;
  ctx->lf_mode.decided = false;  // the next cycle decides again
  const SortScratch sort = ctx->sort_scratch();
  const bool ordered = wants_ordering(ctx) && !mode.beams;
  const bool order_ready = ctx->order_ready;  // (launch_order_ahead's, accepted by this cycle's propagation)
  ctx->order_ready = false;
  if (ordered && !order_ready) {
    KeyFrame frame{};
    // The ordering also serves the beam model: both kernels gather the pose records through sort.perm.
    const bool have_frame = !keys_ready && predict_key_frame(ctx, nullptr, &frame);
    launch_order_particles(ctx->stream, ctx->cur(), ctx->n, &sort, have_frame ? &frame : nullptr, keys_ready, frame.layout);
  }
  if (ctx->cfg.sensor_kind != MCL_SENSOR_BEAM) {
    // Below a few thousand particles the ordering passes cost more than they save.
    // Sets below the ordering threshold (the reference's usual sizes): one or a few particles per wave, lanes over the beams
    // (launch_reweight_lf falls back to the lane-per-particle kernel where the field has no palette form).
    const int variant = mode.beams ? kLfBeamLanes
                                   : ((ctx->tuning.lf_variant == kLfSortedLanes || ctx->tuning.lf_variant == kLfBeamLanes) && !ordered)
                                         ? (lf_set_is_small(ctx) ? kLfBeamLanes : kLfLanePerParticle)
                                         : ctx->tuning.lf_variant;
    if (variant == kLfBeamLanes && !mode.beams) ctx->lf_beams_launches += 1;
    const bool scan_is_short = ctx->scan_extent / ctx->resolution < 8192.0;
    stage_begin(ctx, MCL_STAGE_SENSOR_KERNEL);
    const bool use_patches = mode.patches;
    bool far_tiles_used = false, queue_used = false, far_beams_used = false;
    if (mode.beams) ctx->lf_beams_launches += 1;
    launch_reweight_lf(ctx->stream, ctx->cur(), ctx->n, ctx->field_view(), ctx->d_points.ptr, static_cast<uint32_t>(B), variant, &sort,
                       scan_is_short, ctx->tuning, use_patches,
                       PatchStats{reinterpret_cast<unsigned long long*>(ctx->d_scalars.ptr + 24),
                                  reinterpret_cast<unsigned long long*>(ctx->hd_scalars + 28),
                                  static_cast<uint32_t>(ctx->tuning.lf_loose_below), ctx->tuning.lf_margin ? 0u : 1u,
                                  static_cast<uint32_t>(ctx->tuning.lf_split), want_weight_sums ? ctx->d_lf_wsum.ptr : nullptr,
                                  reinterpret_cast<unsigned int*>(ctx->d_scalars.ptr + 30)},
                       /*dispersed=*/!use_patches && (ctx->tuning.lf_far_tiles == 2 || (ctx->tuning.lf_patch == 1 && !ctx->patch_useful)),
                       &far_tiles_used, &ctx->lf_wsum_count, &queue_used, unit_weights, &far_beams_used);
    if (far_tiles_used) ctx->lf_far_launches += 1;
    if (far_beams_used) ctx->lf_far_beams_launches += 1;
    if (queue_used) ctx->lf_queue_launches += 1;
    stage_end(ctx, MCL_STAGE_SENSOR_KERNEL);
    if (variant == kLfSortedLanes && ctx->tuning.lf_fast != 0 && scan_is_short && ctx->W < 16384 && ctx->H < 16384 && ctx->pal_count &&
        ctx->tuning.lf_table == 0)
    {
      ctx->lf_fast_launches += 1;
      if (use_patches) ctx->lf_patch_launches += 1;  // (unless the tables leave no room in LDS for the patches: launch_reweight_lf)
    }
  } else {
    const mcl_beam_params& b = ctx->cfg.beam;
    const BeamModel model{b.z_hit, b.z_short, b.z_max, b.z_rand, b.sigma_hit, b.lambda_short, b.beam_max_range};
    const bool use_table = ordered && ctx->tuning.beam_table != 0 && ctx->beam_table_count != 0;  // (only the ordered kernel reads it)
    if (use_table && !ctx->beam_table_ready) {
      hipError_t e = ctx->d_beam_table.ensure(4 * static_cast<size_t>(ctx->beam_table_count));
      if (e == hipSuccess) {
        launch_beam_table(ctx->stream, model, ctx->resolution, ctx->beam_table_count, ctx->d_beam_table.ptr);
        e = hipGetLastError();
      }
      if (e != hipSuccess) {
        stage_end(ctx, MCL_STAGE_REWEIGHT);  // (the stage opened above is closed on this way out as well)
        return fail(ctx, MCL_ERR_HIP, std::string("reweight (beam table): ") + hipGetErrorString(e));
      }
      ctx->beam_table_ready = true;
    }
    stage_begin(ctx, MCL_STAGE_SENSOR_KERNEL);
    launch_reweight_beam(ctx->stream, ctx->cur(), ctx->n, ctx->grid_view(), model,
                         ctx->d_points.ptr, static_cast<uint32_t>(B), ctx->d_kld_scalars.ptr + 1, ordered ? &sort : nullptr,
                         ctx->d_nonfree_bits.ptr, ctx->d_beam_points.ptr, use_table ? ctx->d_beam_table.ptr : nullptr,
                         use_table ? ctx->beam_table_count : 0u, ctx->tuning.beam_free_ahead != 0, ctx->tuning.beam_sectors != 0);
    stage_end(ctx, MCL_STAGE_SENSOR_KERNEL);
  }
  stage_end(ctx, MCL_STAGE_REWEIGHT);
  ctx->profile_tick += 1;
  if (const hipError_t e = hipGetLastError(); e != hipSuccess) {
    ctx->lf_wsum_count = 0;  // (sums that no longer describe the weights must not reach a later normalisation)
    return fail(ctx, MCL_ERR_HIP, std::string("reweight: ") + hipGetErrorString(e));
  }
  return MCL_OK;
}

// d_scalars layout: [0] weight sum, [1] norm_sum, [2] norm_sumsq, [3] factor override, [4] cdf total, [8..16] estimate sums
// finalize == false (only with factor = NaN and read_back == false): the totals of the normalised weights in d_scalars[1..3)
// are left to the next kernel (do_build_cdf with a policy, or launch_norm_finalize).
// store_weights == false (mcl_update, a resampling follows at once): the normalised weights are not stored; do_build_cdf has to divide
// (ctx->cdf_divides tells it).
mcl_status do_normalize(mcl_ctx* ctx, double factor, mcl_weight_stats* stats, bool read_back = true, bool finalize = true,
                        bool store_weights = true) {
  ctx->weights_unit = false;
  ctx->cdf_divides = false;
  stage_begin(ctx, MCL_STAGE_NORMALIZE);
  if (std::isnan(factor)) {  // by the set's own total
    launch_sum_and_normalize(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(0), ctx->chunk_row(1), ctx->chunk_row(2),
                             ctx->d_scalars.ptr + 0, ctx->hd_scalars + 0, finalize, ctx->lf_wsum_count ? ctx->d_lf_wsum.ptr : nullptr,
                             ctx->lf_wsum_count, store_weights);
    ctx->cdf_divides = !store_weights;
    ctx->lf_wsum_count = 0;  // (they described the weights as the reweight left them)
  } else {
    launch_weight_sum(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(0), ctx->d_scalars.ptr + 0, ctx->hd_scalars + 0);
    ctx->h_scalars[3] = factor;
    MCL_HIP(ctx, hipMemcpyAsync(ctx->d_scalars.ptr + 3, ctx->h_scalars + 3, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch_normalize(ctx->stream, ctx->cur().w, ctx->n, ctx->d_scalars.ptr + 3, ctx->chunk_row(1), ctx->chunk_row(2),
                     ctx->d_scalars.ptr + 1, ctx->hd_scalars + 1);
  }
  if (!read_back) {  // the caller reads d_scalars[0..3) back later, with its own synchronisation
    stage_end(ctx, MCL_STAGE_NORMALIZE);
    MCL_HIP(ctx, hipGetLastError());
    return MCL_OK;
  }
  stage_end(ctx, MCL_STAGE_NORMALIZE);
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  stage_collect(ctx);
  if (stats) {
    stats->sum = ctx->h_scalars[0];
    stats->norm_sum = ctx->h_scalars[1];
    stats->norm_sumsq = ctx->h_scalars[2];
  }
  return MCL_OK;
}

// do_normalize(NaN, nullptr, false, false) + do_build_cdf(true, policy, true) in one launch where the set allows it (*done says whether it
// did): the fixed-size cycle that resamples at once.  d_scalars[0..3) and [4] as the two leave them, the recovery estimator included.
mcl_status do_normalize_cdf(mcl_ctx* ctx, const RecoveryPolicy& policy, bool* done) {
  *done = false;
  if (ctx->tuning.scan_fused == 0 || !ctx->d_scan_state.ptr || (ctx->tuning.scan_fused == 1 && ctx->n > 65536)) return MCL_OK;
  ctx->weights_unit = false;
  stage_begin(ctx, MCL_STAGE_NORMALIZE);
  if (++ctx->scan_epoch == 0) ctx->scan_epoch = 1;
  *done = launch_normalize_cdf(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(0), ctx->lf_wsum_count ? ctx->d_lf_wsum.ptr : nullptr,
                               ctx->lf_wsum_count, ctx->d_scalars.ptr + 0, ctx->hd_scalars + 0, ctx->chunk_row(1), ctx->chunk_row(2),
                               /*write_weights=*/false, ctx->d_cdf.ptr, ctx->d_scalars.ptr + 4, ctx->d_cdf_tree.ptr, &policy,
                               ctx->d_scan_state.ptr, ctx->scan_epoch);
  if (*done) ctx->lf_wsum_count = 0;
  stage_end(ctx, MCL_STAGE_NORMALIZE);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

// normalized_just_now: the chunk sums k_normalize left in chunk_row(1) are those of the current weights (same summation,
// same bits as k_chunk_sum would produce) and are reused.  policy (needs normalized_just_now): the CDF kernel's first
// workgroup also finishes the normalisation's totals (d_scalars[1..3)) and runs the recovery estimator.
mcl_status do_build_cdf(mcl_ctx* ctx, bool normalized_just_now = false, const RecoveryPolicy* policy = nullptr,
                        bool finalize_norm = false) {
  launch_cdf(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(3), ctx->chunk_row(4), ctx->d_cdf.ptr, ctx->d_scalars.ptr + 4,
             ctx->d_cdf_tree.ptr, normalized_just_now ? ctx->chunk_row(1) : nullptr, finalize_norm ? ctx->chunk_row(2) : nullptr,
             finalize_norm ? ctx->d_scalars.ptr + 1 : nullptr, finalize_norm ? ctx->hd_scalars + 1 : nullptr, policy,
             (normalized_just_now && ctx->cdf_divides) ? ctx->d_scalars.ptr + 0 : nullptr);
  ctx->cdf_divides = false;
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

// with_estimate (fixed-N path only): the draw kernel also leaves the estimate sums of the new set in d_scalars[8..17) and
// their host mirror; *estimate_enqueued says whether it did.
mcl_status do_resample(mcl_ctx* ctx, double random_state_probability, uint32_t step, uint64_t* n_out,
                       const double* d_random_state_probability = nullptr, bool normalized_just_now = false,
                       bool with_estimate = false, bool* estimate_enqueued = nullptr, const RecoveryPolicy* policy = nullptr,
                       bool finalize_norm = false, bool cdf_ready = false) {
  const mcl_amcl_params& a = ctx->cfg.amcl;
  MCL_REQUIRE(ctx, ctx->n > 0, "mcl_resample: empty particle set");
  ctx->lf_wsum_count = 0;  // (the set changes: workgroup sums of an earlier reweight describe another one)
  const uint64_t max_p = std::min<uint64_t>(a.max_particles, ctx->capacity);
  stage_begin(ctx, MCL_STAGE_RESAMPLE);
  if (!cdf_ready)  // (cdf_ready: do_normalize_cdf left the CDF, the totals and the recovery probability)
    if (const mcl_status s = do_build_cdf(ctx, normalized_just_now, policy, finalize_norm)) return s;
  ResampleArgs ra{};
  ra.seed = ctx->cfg.seed;
  ra.step = step;
  ra.random_state_probability = random_state_probability;
  ra.d_random_state_probability = d_random_state_probability;
  ra.n_in = ctx->n;
  const FreeCells fc{ctx->d_free.ptr, ctx->have_map ? ctx->n_free : 0};
  const HashParams hp{a.spatial_resolution_x, a.spatial_resolution_y, a.spatial_resolution_theta};
  const GridView gv = ctx->grid_view();
  uint64_t result = max_p;
  if (a.min_particles >= max_p) {
    // count <= min holds for every candidate (take_while_kld.hpp:86): plain take(max).
    ra.first_candidate = 0;
    ra.count = max_p;
    ra.out_offset = 0;
    if (with_estimate) {
      MCL_HIP(ctx, ctx->d_est_partials.ensure(static_cast<size_t>(9) * ((max_p + 1023) / 1024)));
      Completion done{};
      // (cycle_spin -1: where the cycle is long enough for the stream's completion signal to be what the host waits for last - measured
      // + 1 % at 1M particles in alternating runs, profiles/r06_ab_cycle_end.txt; small sets are bound by the host, which the spin costs)
      const bool spin = ctx->tuning.cycle_spin > 0 || (ctx->tuning.cycle_spin < 0 && max_p >= 262144);
      // (profile 1 times the sensor kernel alone: its events are complete long before the cycle's last kernel stores the word; the per-stage
      // events of profile 2 include stages behind it, which need the stream's own completion)
      ctx->done_armed = spin && ctx->profile <= 1 && estimate_enqueued;
      if (ctx->done_armed) {
        done.d_ticket = reinterpret_cast<unsigned long long*>(ctx->d_scalars.ptr + 27);
        done.host_flag = reinterpret_cast<unsigned long long*>(ctx->hd_scalars + 31);
        done.seq = ++ctx->done_seq;
      }
      // (behind the draw and its sums - the completion word is theirs -: the next cycle's propagation normals, while the host is away)
      // noise_ahead 1: inside the draw kernel (its vector units wait for the fabric); 2: a kernel of its own behind the cycle's last one
      // (both up to 2M particles: at 10M the draw sits at 0.85 of the HBM peak and the normals' stores cost it more than k_propagate saves -
      // 214.4 against 215.8 cycles/s measured)
      const bool noise_in_draw = ctx->tuning.noise_ahead == 1 && max_p > 65536 && max_p <= 2000000 &&
                                 ctx->d_noise.ensure(3 * static_cast<size_t>(max_p)) == hipSuccess;
      const bool noise_ahead = ctx->done_armed && ctx->tuning.noise_ahead == 2 && max_p > 65536 && max_p <= 2000000 &&
                               ctx->d_noise.ensure(3 * static_cast<size_t>(max_p)) == hipSuccess;
      // Option order_ahead (cycles that end on the completion word): the draw kernel also leaves the ordering keys of where its particles will be
      // after the NEXT propagation - through the predicted control action (this cycle's) and the frame of that set as the host predicts it
      // from the last estimate it has, moved twice -, and the ordering passes run behind the cycle's last kernel, while the host is away.
      KeyFrame ahead_frame{};
      const bool order_keys = noise_in_draw && ctx->done_armed && ctx->tuning.order_ahead != 0 && max_p < (1ull << 32) && wants_ordering(ctx) &&
                              predict_key_frame(ctx, &ctx->last_sampler, &ahead_frame, 2);
      launch_resample_draw_and_estimate(ctx->stream, ctx->cur(), ctx->cdf_tree(), ctx->d_scalars.ptr + 4, ctx->other(), ra, gv, fc, hp,
                                        ctx->pivot[0], ctx->pivot[1], ctx->d_est_partials.ptr, ctx->d_scalars.ptr + 8,
                                        ctx->hd_scalars + 8, ctx->done_armed ? &done : nullptr,
                                        ((ctx->tuning.draw_fold == 2 || (ctx->tuning.draw_fold == 1 && max_p <= 65536)) && ctx->d_scan_state.ptr)
                                            ? reinterpret_cast<unsigned int*>(ctx->d_scan_state.ptr + 4)
                                            : nullptr,
                                        noise_in_draw ? ctx->d_noise.ptr : nullptr, max_p, ctx->cfg.shard_offset, step + 1,
                                        order_keys ? ctx->sort_scratch().keys : nullptr, &ctx->last_sampler, &ahead_frame);
      if (noise_ahead) launch_noise_ahead(ctx->stream, ctx->cfg.seed, step + 1, ctx->cfg.shard_offset, max_p, ctx->d_noise.ptr);
      if (noise_ahead || noise_in_draw) {
        ctx->noise_step = step + 1;
        ctx->noise_n = max_p;
        ctx->noise_offset = ctx->cfg.shard_offset;
        ctx->noise_seed = ctx->cfg.seed;
      }
      if (order_keys) {
        const SortScratch sort = ctx->sort_scratch();
        launch_order_ahead(ctx->stream, max_p, &sort);
        ctx->order_valid = true;
        ctx->order_step = step + 1;
        ctx->order_n = max_p;
        ctx->order_sampler = ctx->last_sampler;
        ctx->order_layout = key_layout(ctx);
      }
      if (estimate_enqueued) *estimate_enqueued = true;
    } else {
      launch_resample_draw(ctx->stream, ctx->cur(), ctx->cdf_tree(), ctx->d_scalars.ptr + 4, ctx->other(), ra, gv, fc, hp, nullptr);
    }
    MCL_HIP(ctx, hipGetLastError());
  } else {
    MCL_REQUIRE(ctx, max_p < 0xFFFFFFFFull, "max_particles too large for KLD resampling");
    if (const mcl_status s = kld_begin(ctx)) return s;
    uint64_t chunk = std::max<uint64_t>(a.min_particles + 1, 1ull << 16);
    while (ctx->kld_pos < max_p) {
      const uint64_t pos = ctx->kld_pos;
      const uint64_t cnt = std::min(chunk, max_p - pos);
      if (const mcl_status s = kld_grow_table(ctx, pos + cnt)) return s;
      ra.first_candidate = pos;
      ra.count = cnt;
      ra.out_offset = pos;
      launch_resample_draw(ctx->stream, ctx->cur(), ctx->cdf_tree(), ctx->d_scalars.ptr + 4, ctx->other(), ra, gv, fc, hp,
                           ctx->d_hashes.ptr);
      uint64_t first_fail = ~0ull;
      if (const mcl_status s = kld_process(ctx, cnt, &first_fail)) return s;
      if (first_fail != ~0ull) {
        result = first_fail;  // the first element failing the predicate is dropped (take_while)
        break;
      }
      chunk *= 2;
    }
    result = std::min(result, max_p);  // | take(max)
  }
  stage_end(ctx, MCL_STAGE_RESAMPLE);
  ctx->live ^= 1;
  ctx->n = result;
  ctx->weights_unit = true;  // particle_traits.hpp:105: the draw kernel wrote 1.0 to every weight of the new set
  if (n_out) *n_out = result;
  return MCL_OK;
}

mcl_status do_estimate_sums(mcl_ctx* ctx, const double pivot[2], double sums[12]) {
  stage_begin(ctx, MCL_STAGE_ESTIMATE);
  launch_estimate_sums(ctx->stream, ctx->cur(), ctx->n, pivot[0], pivot[1], ctx->chunk_row(0), ctx->d_scalars.ptr + 8,
                       ctx->hd_scalars + 8);
  stage_end(ctx, MCL_STAGE_ESTIMATE);
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  stage_collect(ctx);
  for (int k = 0; k < 9; ++k) sums[k] = ctx->h_scalars[8 + k];
  sums[9] = pivot[0];
  sums[10] = pivot[1];
  sums[11] = 0.0;
  return MCL_OK;
}


// ---- particle shards: the cycle over a communicator --------------------------------------------------------------------
constexpr size_t kCommScalars = 20;  // d_comm_f64[0] local sum | [1] cdf total | [2] norm sum | [3] norm sumsq | [4] global sum | [5..14) estimate sums |
                                     // [14] this rank's overflow flag of the fixed-capacity exchange (gathered with the sums) | [16..18) the shard plan
constexpr size_t kEstRecord = 10;    // doubles a rank contributes to the estimate's all-gather: nine sums + the overflow flag

mcl_status comm_scratch(mcl_ctx* ctx) {
  const size_t world = ctx->comm_world;
  MCL_HIP(ctx, ctx->d_comm_f64.ensure(kCommScalars + world * (1 + 3 + 2 + kEstRecord)));
  MCL_HIP(ctx, ctx->d_comm_i64.ensure(world + world * world));
  if (!ctx->h_comm) MCL_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_comm), (kCommScalars + 64 * (1 + 3 + 2 + kEstRecord) + 64 * 64 + 64) * sizeof(double)));
  return MCL_OK;
}
mcl_status comm_gather(mcl_ctx* ctx, const void* d_send, void* d_recv, uint64_t bytes) {
  ctx->comm_bytes_out += bytes;
  ctx->comm_collectives += 1;
  if (ctx->transport.all_gather(ctx->transport.user, d_send, d_recv, bytes, ctx->stream) != 0)
    return fail(ctx, MCL_ERR_HIP, "transport all_gather failed");
  return MCL_OK;
}
mcl_status comm_exchange(mcl_ctx* ctx, const void* d_send, const uint64_t* send_bytes, void* d_recv, const uint64_t* recv_bytes) {
  for (uint32_t q = 0; q < ctx->comm_world; ++q)
    if (q != ctx->comm_rank) ctx->comm_bytes_out += send_bytes[q];
  ctx->comm_collectives += 1;
  if (ctx->transport.all_to_all(ctx->transport.user, d_send, send_bytes, d_recv, recv_bytes, ctx->stream) != 0)
    return fail(ctx, MCL_ERR_HIP, "transport all_to_all failed");
  return MCL_OK;
}

// What selects the SEQUENCE of collectives a sharded cycle runs (fixed-size or KLD-adaptive, selective resampling, the recovery
// estimator on the device or on the host, the plain or the cluster-based estimate, the resampling interval) has to be the same on
// every rank, or the ranks block in different collectives for ever.  Part of it comes from each process's own environment
// (BELUGA_MCL_DEVICE_POLICY) or from calls made after the communicator exists: the word below is all-gathered and compared when
// the communicator is attached and whenever one of those calls changes it (they are then collective calls: every rank makes them).
uint64_t comm_path_word(const mcl_ctx* ctx) {
  const mcl_amcl_params& ap = ctx->cfg.amcl;
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](uint64_t v) {
    for (int k = 0; k < 8; ++k) {
      h ^= (v >> (8 * k)) & 0xFFu;
      h *= 1099511628211ull;
    }
  };
  mix(ap.min_particles);
  mix(ap.max_particles);
  mix(static_cast<uint64_t>(ap.resample_interval));
  mix(ap.selective_resampling ? 1u : 0u);
  mix(static_cast<uint64_t>(ctx->cfg.sensor_kind));
  mix(static_cast<uint64_t>(ctx->cfg.motion_kind));
  mix(ctx->cfg.seed);
  mix(static_cast<uint64_t>(ctx->tuning.device_policy != 0));
  mix(static_cast<uint64_t>(ctx->estimate_kind));
  mix(static_cast<uint64_t>(ctx->tuning.shard_pad_permille));  // (the byte counts of the fixed-capacity exchange's two all-to-alls)
  // (the thresholds decide update / no update, the recovery alphas whether a cycle injects, the KLD parameters where a cut falls:
  // each of them changes which collectives a cycle reaches)
  for (const double v : {ap.update_min_d, ap.update_min_a, ap.alpha_slow, ap.alpha_fast, ap.kld_epsilon, ap.kld_z, ap.spatial_resolution_x,
                         ap.spatial_resolution_y, ap.spatial_resolution_theta, ctx->cluster_params.linear_hash_resolution,
                         ctx->cluster_params.angular_hash_resolution, ctx->cluster_params.weight_cap_percentile}) {
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    mix(bits);
  }
  mix(static_cast<uint64_t>(ctx->comm_world));
  return h;
}
mcl_status comm_agree(mcl_ctx* ctx, const char* where) {
  if (!ctx->have_comm || ctx->comm_world <= 1) return MCL_OK;
  if (const mcl_status s = bind_device(ctx)) return s;
  if (const mcl_status s = comm_scratch(ctx)) return s;
  const uint32_t world = ctx->comm_world;
  long long* d_words = ctx->d_comm_i64.ptr;
  long long* h_words = reinterpret_cast<long long*>(ctx->h_comm + kCommScalars + 64 * (1 + 3 + 2 + kEstRecord));
  const uint64_t mine = comm_path_word(ctx);
  std::memcpy(h_words, &mine, sizeof(mine));
  MCL_HIP(ctx, hipMemcpyAsync(d_words, h_words, sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
  if (const mcl_status s = comm_gather(ctx, d_words, d_words + world, sizeof(long long))) return s;
  MCL_HIP(ctx, hipMemcpyAsync(h_words, d_words + world, world * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t r = 0; r < world; ++r) {
    uint64_t theirs;
    std::memcpy(&theirs, h_words + r, sizeof(theirs));
    if (theirs != mine)
      return fail(ctx, MCL_ERR_INVALID_ARGUMENT,
                  std::string(where) + ": rank " + std::to_string(r) + " runs another configuration (particle bounds, resampling policy, models, seed, "
                  "device_policy, estimate kind, shard_pad_permille): every shard of a filter must be created and switched alike");
  }
  return MCL_OK;
}

// The nine estimate sums (estimation.hpp:436-475) over all shards: local sums - of the particles whose cell carries cluster id
// wanted_plus_1 - 1 in t_cluster when t_cluster is given -, gathered, added in rank order by every rank.
mcl_status sharded_estimate_sums(mcl_ctx* ctx, unsigned int* t_cluster, unsigned int wanted_plus_1, double sums[12], uint64_t slots = 0) {
  if (const mcl_status s = comm_scratch(ctx)) return s;
  const uint32_t world = ctx->comm_world;
  double* d = ctx->d_comm_f64.ptr;
  double* d_gather_est = d + kCommScalars + world * (1 + 3 + 2);  // [world][kEstRecord]
  if (ctx->n == 0) {
    MCL_HIP(ctx, hipMemsetAsync(d + 5, 0, 9 * sizeof(double), ctx->stream));
  } else if (t_cluster) {
    launch_estimate_sums_cluster(ctx->stream, ctx->cur(), ctx->n, ctx->d_hashes.ptr, ctx->d_table_keys.ptr, t_cluster, slots, wanted_plus_1 - 1u,
                                 ctx->pivot[0], ctx->pivot[1], ctx->chunk_row(0), d + 5);
  } else {
    launch_estimate_sums(ctx->stream, ctx->cur(), ctx->n, ctx->pivot[0], ctx->pivot[1], ctx->chunk_row(0), d + 5);
  }
  MCL_HIP(ctx, hipGetLastError());
  // (with the nine sums travels d[14]: this rank's overflow flag of the cycle's fixed-capacity exchange - their sum lands in h_scalars[17])
  if (const mcl_status s = comm_gather(ctx, d + 5, d_gather_est, kEstRecord * sizeof(double))) return s;
  launch_sum_rows(ctx->stream, d_gather_est, world, kEstRecord, ctx->d_scalars.ptr + 8, ctx->hd_scalars + 8);
  MCL_HIP(ctx, hipGetLastError());
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->comm_host_syncs += 1;
  for (int k = 0; k < 9; ++k) sums[k] = ctx->h_scalars[8 + k];
  sums[9] = ctx->pivot[0];
  sums[10] = ctx->pivot[1];
  sums[11] = 0.0;
  return MCL_OK;
}

// algorithm/spatial_hash.hpp:45-75,87-94,190-193 on the host (neighbour cells of the cluster flood fill).
uint64_t host_floor_and_fibo_hash(double value, unsigned shift) {
  const int64_t sv = static_cast<int64_t>(std::floor(value));
  const uint64_t h = 11400714819323198485ull * static_cast<uint64_t>(sv);
  return shift ? ((h << shift) | (h >> (64 - shift))) : h;
}
uint64_t host_spatial_hash(const Pose2& s, double res_xy, double res_theta) {
  return host_floor_and_fibo_hash(s.x / res_xy, 0) ^ host_floor_and_fibo_hash(s.y / res_xy, 21) ^
         host_floor_and_fibo_hash(rot_log(s.r) / res_theta, 42);
}

// cluster_based_estimation.hpp:415-433.  Device: hashing, per-cell aggregation, masked sums.  Host: the cluster
// assignment over the (few) occupied cells with std::unordered_map / std::priority_queue / std::nth_element, fed in
// first-occurrence order so the containers evolve as they do in the reference.
mcl_status do_cluster_estimate(mcl_ctx* ctx, const mcl_cluster_params& cp, mcl_estimate* out) {
  const uint64_t n = ctx->n;
  const bool sharded = ctx->have_comm && ctx->comm_world > 1;
  if (n == 0 && !sharded) return fail(ctx, MCL_ERR_NOT_READY, "no particles");  // (an empty shard still takes part in the exchange)
  MCL_REQUIRE(ctx, cp.linear_hash_resolution > 0 && cp.angular_hash_resolution > 0 && cp.weight_cap_percentile >= 0 &&
                       cp.weight_cap_percentile < 1.0, "bad cluster parameters");
  MCL_REQUIRE(ctx, n < 0xFFFFFFFFull, "too many particles");
  if (ctx->table_capacity == 0) {
    if (const mcl_status s = ensure_kld(ctx)) return s;
  }
  uint64_t slots = 1024;
  while (slots < 2 * n) slots <<= 1;
  slots = std::min<uint64_t>(slots, ctx->table_capacity);
  const uint64_t m_cap = std::min<uint64_t>(n, slots);
  const uint64_t tcap = ctx->table_capacity;
  MCL_HIP(ctx, ctx->d_cell_f64.ensure(tcap + 5 * ctx->capacity));
  MCL_HIP(ctx, ctx->d_cell_u32.ensure(2 * tcap + 4 * ctx->capacity + 4));
  MCL_HIP(ctx, ctx->d_cell_u64.ensure(ctx->capacity));
  double* t_wsum = ctx->d_cell_f64.ptr;
  double* c_wsum = t_wsum + tcap;
  double* c_state = c_wsum + ctx->capacity;
  unsigned int* t_count = ctx->d_cell_u32.ptr;
  unsigned int* t_cluster = t_count + tcap;
  unsigned int* c_first = t_cluster + tcap;
  unsigned int* c_count = c_first + ctx->capacity;
  unsigned int* c_slot = c_count + ctx->capacity;
  unsigned int* c_cluster = c_slot + ctx->capacity;
  unsigned int* c_size = c_cluster + ctx->capacity;
  unsigned long long* c_key = ctx->d_cell_u64.ptr;

  // The occupied cells come to the host through a list in MAPPED pinned memory that the compaction kernel writes itself (a
  // converged cloud has a few hundred cells): one synchronisation instead of seven.  A cloud with more cells than the list
  // holds (global localisation) is compacted again into the device arrays and copied.
  constexpr unsigned int kHostCells = 16384;
  if (!ctx->h_cells) {
    const size_t bytes = kHostCells * (sizeof(unsigned long long) + 5 * sizeof(double) + 4 * sizeof(unsigned int)) + 64;
    MCL_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_cells), bytes, hipHostMallocMapped));
    MCL_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->hd_cells), ctx->h_cells, 0));
  }
  auto cell_views = [&](unsigned char* base, unsigned long long*& key, double*& wsum, double*& state, unsigned int*& first,
                        unsigned int*& count, unsigned int*& slot, unsigned int*& cluster, unsigned int*& size) {
    key = reinterpret_cast<unsigned long long*>(base);
    wsum = reinterpret_cast<double*>(key + kHostCells);
    state = wsum + kHostCells;
    first = reinterpret_cast<unsigned int*>(state + 4 * kHostCells);
    count = first + kHostCells;
    slot = count + kHostCells;
    cluster = slot + kHostCells;
    size = cluster + kHostCells;
  };
  unsigned long long *hk, *dk;
  double *hw, *hs, *dw, *ds;
  unsigned int *hf, *hc, *hsl, *hcl, *hsize, *df, *dc, *dsl, *dcl, *dsize;
  cell_views(ctx->h_cells, hk, hw, hs, hf, hc, hsl, hcl, hsize);
  cell_views(ctx->hd_cells, dk, dw, ds, df, dc, dsl, dcl, dsize);
  const HashParams hp{cp.linear_hash_resolution, cp.linear_hash_resolution, cp.angular_hash_resolution};
  unsigned int m = 0;
  bool local_failure = false;  // (sharded: reported to the peers with the count, so that every rank leaves together)
  // a small set on one context: one workgroup writes the cells straight into the mapped list (k_small_cluster_cells), another one adds the
  // winning cluster's particles up (k_small_cluster_sums) - two launches instead of eight
  bool small = false;
  if (n && !sharded && ctx->tuning.small_fused != 0 && n <= 4096)
    small = launch_small_cluster_cells(ctx->stream, ctx->cur(), n, hp, dk, df, dc, dsl, dw, ds, c_size, dsize);
  if (small) {
    MCL_HIP(ctx, hipGetLastError());
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m = *hsize;
    if (!(m >= 1 && m <= m_cap)) return fail(ctx, MCL_ERR_HIP, "cell compaction failed");
  } else if (n) {
    launch_cluster_cells(ctx->stream, ctx->cur(), n, hp, ctx->d_hashes.ptr, ctx->d_table_keys.ptr, ctx->d_table_first.ptr, t_wsum,
                         t_count, t_cluster, slots, dk, df, dc, dsl, dw, ds, c_size, kHostCells);
    MCL_HIP(ctx, hipGetLastError());
    MCL_HIP(ctx, hipMemcpyAsync(hsize, c_size, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m = *hsize;
    if (!(m >= 1 && m <= m_cap)) {
      if (!sharded) return fail(ctx, MCL_ERR_HIP, "cell compaction failed");
      local_failure = true;
      m = 0;
    }
  }
  ctx->cluster_cells = m;
  std::vector<unsigned long long> key_big;
  std::vector<unsigned int> first_big, count_big;
  std::vector<double> wsum_big, state_big;
  const unsigned long long* key = hk;
  const unsigned int *first = hf, *count = hc, *slot_list = dsl;
  const double *wsum = hw, *state = hs;
  const bool on_host_list = m <= kHostCells;
  if (!on_host_list) {
    MCL_HIP(ctx, hipMemsetAsync(c_size, 0, sizeof(unsigned int), ctx->stream));
    launch_cluster_cells(ctx->stream, ctx->cur(), n, hp, ctx->d_hashes.ptr, ctx->d_table_keys.ptr, ctx->d_table_first.ptr, t_wsum,
                         t_count, t_cluster, slots, c_key, c_first, c_count, c_slot, c_wsum, c_state, c_size,
                         static_cast<unsigned int>(std::min<uint64_t>(ctx->capacity, 0xFFFFFFFFull)), /*table_ready=*/true);
    MCL_HIP(ctx, hipGetLastError());
    key_big.resize(m);
    first_big.resize(m);
    count_big.resize(m);
    wsum_big.resize(m);
    state_big.resize(4 * static_cast<size_t>(m));
    // on the context's stream, behind the compaction (the stream does not synchronise with the null stream)
    MCL_HIP(ctx, hipMemcpyAsync(key_big.data(), c_key, m * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipMemcpyAsync(first_big.data(), c_first, m * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipMemcpyAsync(count_big.data(), c_count, m * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipMemcpyAsync(wsum_big.data(), c_wsum, m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipMemcpyAsync(state_big.data(), c_state, 4 * static_cast<size_t>(m) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    key = key_big.data();
    first = first_big.data();
    count = count_big.data();
    wsum = wsum_big.data();
    state = state_big.data();
    slot_list = c_slot;
  }

  // The occupied cells in the order their first particle appears in the set (make_cluster_map :137-157 inserts them in that
  // order).  Over shards: every rank's list in that order, gathered, and merged rank by rank - shards are contiguous pieces of
  // the global index space, so rank order followed by local order IS the global first-occurrence order; a cell seen by
  // several ranks keeps the state of its first particle and adds up weights and counts.
  std::vector<uint32_t> order(m);
  std::iota(order.begin(), order.end(), 0u);
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
  std::vector<unsigned long long> g_key;
  std::vector<double> g_wsum, g_state;
  std::vector<uint64_t> g_count;
  if (!sharded) {
    g_key.resize(m);
    g_wsum.resize(m);
    g_count.resize(m);
    g_state.resize(4 * static_cast<size_t>(m));
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t k = order[j];
      g_key[j] = key[k];
      g_wsum[j] = wsum[k];
      g_count[j] = count[k];
      std::memcpy(&g_state[4 * static_cast<size_t>(j)], state + 4 * static_cast<size_t>(k), 4 * sizeof(double));
    }
  } else {
    if (const mcl_status s = comm_scratch(ctx)) return s;
    const uint32_t world = ctx->comm_world;
    constexpr size_t kRecord = 7;  // doubles per cell: key (bit pattern), weight sum, count (bit pattern), state[4]
    long long* d_counts = ctx->d_comm_i64.ptr;
    long long* h_counts = reinterpret_cast<long long*>(ctx->h_comm + kCommScalars + 64 * (1 + 3 + 2 + kEstRecord));
    h_counts[0] = local_failure ? -1 : static_cast<long long>(m);  // -1: this rank's compaction failed
    MCL_HIP(ctx, hipMemcpyAsync(d_counts, h_counts, sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
    if (const mcl_status s = comm_gather(ctx, d_counts, d_counts + world, sizeof(long long))) return s;
    MCL_HIP(ctx, hipMemcpyAsync(h_counts, d_counts + world, world * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> m_of(world);
    uint64_t widest = 0;
    for (uint32_t r = 0; r < world; ++r) {
      if (h_counts[r] < 0)  // every rank sees it behind the same collective and returns here: nobody is left in the next one
        return fail(ctx, MCL_ERR_HIP, "cluster_based_estimate: cell compaction failed on rank " + std::to_string(r));
      m_of[r] = static_cast<uint64_t>(h_counts[r]);
      widest = std::max(widest, m_of[r]);
    }
    std::vector<double> packed(kRecord * widest, 0.0);
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t k = order[j];
      double* rec = &packed[kRecord * static_cast<size_t>(j)];
      const unsigned long long cnt = count[k];
      std::memcpy(rec + 0, &key[k], sizeof(double));
      rec[1] = wsum[k];
      std::memcpy(rec + 2, &cnt, sizeof(double));
      std::memcpy(rec + 3, state + 4 * static_cast<size_t>(k), 4 * sizeof(double));
    }
    MCL_HIP(ctx, ctx->d_cell_exchange.ensure(kRecord * widest * (world + 1)));
    double* d_send = ctx->d_cell_exchange.ptr;
    double* d_recv = d_send + kRecord * widest;
    MCL_HIP(ctx, hipMemcpyAsync(d_send, packed.data(), kRecord * widest * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (const mcl_status s = comm_gather(ctx, d_send, d_recv, kRecord * widest * sizeof(double))) return s;
    std::vector<double> all(kRecord * widest * world);
    MCL_HIP(ctx, hipMemcpyAsync(all.data(), d_recv, all.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::unordered_map<unsigned long long, size_t> seen;
    for (uint32_t r = 0; r < world; ++r) {
      for (uint64_t j = 0; j < m_of[r]; ++j) {
        const double* rec = &all[kRecord * (static_cast<size_t>(r) * widest + j)];
        unsigned long long k64, cnt;
        std::memcpy(&k64, rec + 0, sizeof k64);
        std::memcpy(&cnt, rec + 2, sizeof cnt);
        const auto [it, fresh] = seen.try_emplace(k64, g_key.size());
        if (fresh) {
          g_key.push_back(k64);
          g_wsum.push_back(rec[1]);
          g_count.push_back(cnt);
          g_state.insert(g_state.end(), rec + 3, rec + 7);
        } else {
          g_wsum[it->second] += rec[1];
          g_count[it->second] += cnt;
        }
      }
    }
  }
  const size_t cells = g_key.size();
  if (cells == 0) return fail(ctx, MCL_ERR_NOT_READY, "no particles");
  uint64_t n_global = 0;
  for (const uint64_t c : g_count) n_global += c;

  // make_cluster_map :137-157
  struct Cell {
    Pose2 representative_state;
    double weight;
    size_t num_particles;
    std::optional<size_t> cluster_id;
    size_t k;
  };
  std::unordered_map<size_t, Cell> map;
  map.reserve(n_global / 5);
  for (size_t k = 0; k < cells; ++k) {
    map.try_emplace(static_cast<size_t>(g_key[k]),
                    Cell{Pose2{Rot2{g_state[4 * k], g_state[4 * k + 1]}, g_state[4 * k + 2], g_state[4 * k + 3]}, g_wsum[k],
                         static_cast<size_t>(g_count[k]), std::nullopt, k});
  }
  // normalize_and_cap_weights :173-189 (+ calculate_percentile_threshold :103-109)
  for (auto& kv : map) kv.second.weight /= static_cast<double>(kv.second.num_particles);
  {
    std::vector<double> values;
    values.reserve(map.size());
    for (auto& kv : map) values.push_back(kv.second.weight);
    const auto nth = static_cast<std::ptrdiff_t>(static_cast<double>(values.size()) * cp.weight_cap_percentile);
    std::nth_element(values.begin(), values.begin() + nth, values.end());
    const double max_weight = values[static_cast<size_t>(nth)];
    for (auto& kv : map) kv.second.weight = std::min(kv.second.weight, max_weight);
  }
  // assign_clusters :203-238
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
  const double lin = cp.linear_hash_resolution, ang = cp.angular_hash_resolution;
  const Pose2 adjacent[6] = {Pose2{rot_exp(0.0), +lin, 0.0}, Pose2{rot_exp(0.0), -lin, 0.0}, Pose2{rot_exp(0.0), 0.0, +lin},
                             Pose2{rot_exp(0.0), 0.0, -lin}, Pose2{rot_exp(+ang), 0.0, 0.0}, Pose2{rot_exp(-ang), 0.0, 0.0}};
  size_t next_cluster_id = 0;
  while (!queue.empty()) {
    const size_t hash = queue.top().key;
    queue.pop();
    Cell& cell = map[hash];
    if (!cell.cluster_id.has_value()) cell.cluster_id = next_cluster_id++;
    for (const Pose2& adj : adjacent) {
      uint64_t neighbor_hash = host_spatial_hash(pose_mul(cell.representative_state, adj), lin, ang);
      if (neighbor_hash == ~0ull) neighbor_hash -= 1;  // the device table's reserved key
      auto it = map.find(static_cast<size_t>(neighbor_hash));
      if (it == map.end() || it->second.cluster_id.has_value() || !(it->second.weight <= cell.weight)) continue;
      it->second.cluster_id = cell.cluster_id;
      queue.push(KeyWithPriority{max_priority + it->second.weight, static_cast<size_t>(neighbor_hash)});
    }
  }
  // estimate_clusters :345-411: clusters with more than one particle, the first one of maximum total weight
  std::vector<double> total_w(next_cluster_id, 0.0);
  std::vector<uint64_t> total_n(next_cluster_id, 0);
  std::vector<unsigned int> cluster_of_cell(cells);
  for (auto& kv : map) cluster_of_cell[kv.second.k] = static_cast<unsigned int>(kv.second.cluster_id.value());
  for (size_t k = 0; k < cells; ++k) {  // particle-order accumulation is not reproducible from cell sums; cell order is fixed
    total_w[cluster_of_cell[k]] += g_wsum[k];
    total_n[cluster_of_cell[k]] += g_count[k];
  }
  long best = -1;
  for (size_t c = 0; c < next_cluster_id; ++c)
    if (total_n[c] > 1 && (best < 0 || total_w[static_cast<size_t>(best)] < total_w[c])) best = static_cast<long>(c);
  double sums[12];
  if (best < 0) {  // :424-427 no cluster: overall mean and covariance
    if (sharded) {
      if (const mcl_status s = sharded_estimate_sums(ctx, nullptr, 0, sums)) return s;
    } else if (const mcl_status s = do_estimate_sums(ctx, ctx->pivot, sums)) {
      return s;
    }
    return mcl_estimate_from_sums(sums, out);
  }
  // this rank's cells -> their clusters (list order, the order of slot_list)
  std::vector<unsigned int> cluster_of(m);
  for (uint32_t j = 0; j < m; ++j) {
    const uint32_t k = order[j];
    cluster_of[k] = sharded ? static_cast<unsigned int>(map[static_cast<size_t>(key[k])].cluster_id.value()) : cluster_of_cell[j];
  }
  const unsigned int* cluster_list = c_cluster;
  if (on_host_list) {  // the kernel reads the cluster ids from the mapped list
    std::memcpy(hcl, cluster_of.data(), m * sizeof(unsigned int));
    cluster_list = dcl;
  } else {
    MCL_HIP(ctx, hipMemcpyAsync(c_cluster, cluster_of.data(), m * sizeof(unsigned int), hipMemcpyHostToDevice, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (cluster_of is pageable host memory of this call)
  }
  if (small) {  // (the cells' keys and their cluster ids are in the mapped list: hk / hcl)
    launch_small_cluster_sums(ctx->stream, ctx->cur(), n, hp, dk, dcl, m, static_cast<unsigned int>(best), ctx->pivot[0], ctx->pivot[1],
                              ctx->d_scalars.ptr + 8, ctx->hd_scalars + 8);
    MCL_HIP(ctx, hipGetLastError());
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 9; ++k) sums[k] = ctx->h_scalars[8 + k];
    sums[9] = ctx->pivot[0];
    sums[10] = ctx->pivot[1];
    sums[11] = 0.0;
    return mcl_estimate_from_sums(sums, out);
  }
  if (m) launch_cell_set_cluster(ctx->stream, slot_list, cluster_list, m, t_cluster);
  if (sharded) return [&] {
    if (const mcl_status s = sharded_estimate_sums(ctx, t_cluster, static_cast<unsigned int>(best) + 1u, sums, slots)) return s;
    return mcl_estimate_from_sums(sums, out);
  }();
  launch_estimate_sums_cluster(ctx->stream, ctx->cur(), n, ctx->d_hashes.ptr, ctx->d_table_keys.ptr, t_cluster, slots,
                               static_cast<unsigned int>(best), ctx->pivot[0], ctx->pivot[1], ctx->chunk_row(0), ctx->d_scalars.ptr + 8,
                               ctx->hd_scalars + 8);
  MCL_HIP(ctx, hipGetLastError());
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 9; ++k) sums[k] = ctx->h_scalars[8 + k];
  sums[9] = ctx->pivot[0];
  sums[10] = ctx->pivot[1];
  sums[11] = 0.0;
  return mcl_estimate_from_sums(sums, out);
}


// Contiguous, balanced split of [0, n_total) over the ranks.
void shard_bounds(uint64_t n_total, uint32_t world, uint32_t rank, uint64_t* first, uint64_t* count) {
  const uint64_t base = n_total / world, rem = n_total % world;
  *first = rank * base + std::min<uint64_t>(rank, rem);
  *count = base + (rank < rem ? 1 : 0);
}

// The ancestor exchange for the output slots [first_slot, first_slot + m) of views::sample | random_intersperse: every
// slot's point of the global CDF goes to the shard that owns it, which answers with the state.  Leaves the targets in
// d_targets, the replies (request order) in d_replies_in and the slot of every request in d_route_order.
// d_plan (optional): {total, random state probability} on the device (launch_shard_plan) instead of the two values.
// Entries per pair of ranks in the fixed-capacity exchange: what a shard's m output slots ask of one other shard - m / world on average,
// the shards' weight sums being those of equal random samples of one set - plus `permille` / 1000 - 1 of it, eight standard deviations
// of the binomial count and 64.  Every rank derives the same number from the same arguments.
uint64_t padded_capacity(uint64_t n_total, uint32_t world, uint32_t permille) {
  const uint64_t m_max = (n_total + world - 1) / world;
  const double mean = static_cast<double>(m_max) / world;
  const double cap = mean * (permille / 1000.0) + 8.0 * std::sqrt(mean) + 64.0;
  return (static_cast<uint64_t>(cap) + 63u) & ~63ull;
}

// The same exchange without a host read in the middle of the cycle: every pair of ranks moves `cap` entries whatever the counts are
// (requests: cap doubles, NaN = none; replies: cap states), so that the sizes of both all-to-alls are known before anything is
// computed.  6 % more bytes than the exact form; a rank whose requests to one shard do not fit sets its
// overflow flag (d_comm_f64[14]), which travels with the estimate sums: sharded_update then runs the resampling again, exactly.
// Leaves targets, replies and slots as sharded_draw does, in lists of world * cap entries.
mcl_status sharded_draw_padded(mcl_ctx* ctx, const double* d_intervals, uint64_t first_slot, uint64_t m, const double* d_plan, uint64_t cap,
                               uint64_t* entries_out) {
  const uint32_t world = ctx->comm_world, rank = ctx->comm_rank;
  const uint64_t entries = cap * world;
  MCL_REQUIRE(ctx, entries < (1ull << 32), "too many exchange entries");
  long long* d_counts = ctx->d_comm_i64.ptr;  // [world] (not read by the host here)
  MCL_HIP(ctx, ctx->d_targets.ensure(std::max<uint64_t>(m, 1)));
  MCL_HIP(ctx, ctx->d_send_targets.ensure(entries));
  MCL_HIP(ctx, ctx->d_route_order.ensure(entries));
  MCL_HIP(ctx, ctx->d_replies_in.ensure(4 * entries));
  MCL_HIP(ctx, ctx->d_requests_in.ensure(entries));
  MCL_HIP(ctx, ctx->d_replies_out.ensure(4 * entries));
  MCL_HIP(ctx, hipMemsetAsync(ctx->d_send_targets.ptr, 0xFF, entries * sizeof(double), ctx->stream));    // NaN: no request
  MCL_HIP(ctx, hipMemsetAsync(ctx->d_route_order.ptr, 0xFF, entries * sizeof(uint32_t), ctx->stream));    // 0xFFFFFFFF: no slot
  launch_resample_targets(ctx->stream, ctx->cfg.seed, ctx->step, 0.0, 0.0, first_slot, m, ctx->have_map ? ctx->n_free : 0, ctx->d_targets.ptr, d_plan);
  MCL_HIP(ctx, hipGetLastError());
  {
    const size_t nblocks = num_chunks(m);
    const size_t hist = static_cast<size_t>(world) * nblocks;
    MCL_HIP(ctx, ctx->d_route_u32.ensure(hist + 2 * (hist / kChunk + 1) + (m + 3) / 4 + 4));
    uint32_t* block_hist = ctx->d_route_u32.ptr;
    uint32_t* chunk_sum = block_hist + hist;
    uint32_t* chunk_off = chunk_sum + (hist / kChunk + 1);
    uint8_t* dest = reinterpret_cast<uint8_t*>(chunk_off + (hist / kChunk + 1));
    launch_route_targets(ctx->stream, ctx->d_targets.ptr, m, d_intervals, d_intervals + world, world, rank, dest, block_hist, chunk_sum, chunk_off,
                         ctx->d_send_targets.ptr, ctx->d_route_order.ptr, d_counts, static_cast<uint32_t>(cap), ctx->d_comm_f64.ptr + 14);
    MCL_HIP(ctx, hipGetLastError());
  }
  std::vector<uint64_t> request_bytes(world, cap * sizeof(double)), reply_bytes(world, cap * 4 * sizeof(double));
  if (const mcl_status s = comm_exchange(ctx, ctx->d_send_targets.ptr, request_bytes.data(), ctx->d_requests_in.ptr, request_bytes.data())) return s;
  if (const mcl_status s = mcl_serve_requests(ctx, ctx->d_requests_in.ptr, entries, ctx->d_replies_out.ptr)) return s;
  *entries_out = entries;
  return comm_exchange(ctx, ctx->d_replies_out.ptr, reply_bytes.data(), ctx->d_replies_in.ptr, reply_bytes.data());
}

mcl_status sharded_draw(mcl_ctx* ctx, double random_state_probability, double total, const double* d_intervals, uint64_t first_slot,
                        uint64_t m, const double* d_plan = nullptr) {
  const uint32_t world = ctx->comm_world, rank = ctx->comm_rank;
  long long* d_counts = ctx->d_comm_i64.ptr;   // [world]
  long long* d_all_counts = d_counts + world;  // [world][world]
  MCL_HIP(ctx, ctx->d_targets.ensure(std::max<uint64_t>(m, 1)));
  MCL_HIP(ctx, ctx->d_send_targets.ensure(std::max<uint64_t>(m, 1)));
  MCL_HIP(ctx, ctx->d_route_order.ensure(std::max<uint64_t>(m, 1)));
  MCL_HIP(ctx, ctx->d_replies_in.ensure(std::max<uint64_t>(4 * m, 4)));
  launch_resample_targets(ctx->stream, ctx->cfg.seed, ctx->step, random_state_probability, total, first_slot, m, ctx->have_map ? ctx->n_free : 0,
                          ctx->d_targets.ptr, d_plan);
  MCL_HIP(ctx, hipGetLastError());
  if (const mcl_status s = mcl_route_targets(ctx, ctx->d_targets.ptr, m, d_intervals, d_intervals + world, world, rank, ctx->d_send_targets.ptr,
                                             ctx->d_route_order.ptr, reinterpret_cast<int64_t*>(d_counts))) return s;
  if (const mcl_status s = comm_gather(ctx, d_counts, d_all_counts, world * sizeof(long long))) return s;  // counts[r][q]: r asks q
  long long* h_counts = reinterpret_cast<long long*>(ctx->h_comm + kCommScalars + 64 * (1 + 3 + 2 + kEstRecord));
  MCL_HIP(ctx, hipMemcpyAsync(h_counts, d_all_counts, world * world * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->comm_host_syncs += 1;
  std::vector<uint64_t> send_requests(world), recv_requests(world), send_replies(world), recv_replies(world);
  uint64_t incoming = 0;
  for (uint32_t q = 0; q < world; ++q) {
    const uint64_t out = static_cast<uint64_t>(h_counts[rank * world + q]), in = static_cast<uint64_t>(h_counts[q * world + rank]);
    send_requests[q] = out * sizeof(double);
    recv_requests[q] = in * sizeof(double);
    send_replies[q] = in * 4 * sizeof(double);
    recv_replies[q] = out * 4 * sizeof(double);
    incoming += in;
  }
  MCL_HIP(ctx, ctx->d_requests_in.ensure(std::max<uint64_t>(incoming, 1)));
  MCL_HIP(ctx, ctx->d_replies_out.ensure(std::max<uint64_t>(4 * incoming, 4)));
  if (const mcl_status s = comm_exchange(ctx, ctx->d_send_targets.ptr, send_requests.data(), ctx->d_requests_in.ptr, recv_requests.data())) return s;
  if (incoming) {
    if (const mcl_status s = mcl_serve_requests(ctx, ctx->d_requests_in.ptr, incoming, ctx->d_replies_out.ptr)) return s;
  }
  return comm_exchange(ctx, ctx->d_replies_out.ptr, send_replies.data(), ctx->d_replies_in.ptr, recv_replies.data());
}

// views::sample | random_intersperse | take_while_kld | take(max) | actions::assign (amcl_core.hpp:188-196) over the shards:
// candidates are drawn block by block (doubling), every rank drawing its slice of a block through the ancestor exchange;
// the spatial hashes of the whole block are all-gathered so that every rank checks kld_condition over the GLOBAL candidate
// stream (take_while_kld.hpp:72-88,112-137) and takes the same cut; the kept candidates [0, n_out) are then re-balanced into
// contiguous shards of the new set.  Same candidate stream, same cut as the single-context filter.
mcl_status sharded_resample_kld(mcl_ctx* ctx, double random_state_probability, double total, const double* d_intervals, uint64_t* n_out_total) {
  const mcl_amcl_params& ap = ctx->cfg.amcl;
  const uint32_t world = ctx->comm_world, rank = ctx->comm_rank;
  const uint64_t max_p = ap.max_particles, min_p = ap.min_particles;
  if (const mcl_status s = mcl_kld_begin(ctx)) return s;
  struct Block { uint64_t pos, cnt, offset, mine; };  // offset: of this rank's slice in d_cand_states (particles)
  std::vector<Block> blocks;
  MCL_HIP(ctx, ctx->d_cand_states.ensure(4 * (max_p / world + 66)));
  uint64_t pos = 0, block = std::max<uint64_t>(min_p + 1, 8192ull * world), held = 0, n_out = max_p;
  while (pos < max_p) {
    const uint64_t cnt = std::min(block, max_p - pos);
    uint64_t lo, m;
    shard_bounds(cnt, world, rank, &lo, &m);
    if (const mcl_status s = sharded_draw(ctx, random_state_probability, total, d_intervals, pos + lo, m)) return s;
    const uint64_t width = (cnt + world - 1) / world, rem = cnt % world;
    MCL_HIP(ctx, ctx->d_cand_hashes.ensure(width));
    MCL_HIP(ctx, ctx->d_block_hashes.ensure(2 * world * width));
    if (m < width) MCL_HIP(ctx, hipMemsetAsync(ctx->d_cand_hashes.ptr + m, 0, (width - m) * sizeof(unsigned long long), ctx->stream));
    if (const mcl_status s = mcl_finish_candidates(ctx, ctx->step, pos + lo, m, ctx->d_replies_in.ptr, ctx->d_route_order.ptr, ctx->d_targets.ptr,
                                                   ctx->d_cand_states.ptr + 4 * held, reinterpret_cast<uint64_t*>(ctx->d_cand_hashes.ptr))) return s;
    unsigned long long* gathered = ctx->d_block_hashes.ptr;
    if (const mcl_status s = comm_gather(ctx, ctx->d_cand_hashes.ptr, gathered, width * sizeof(unsigned long long))) return s;
    const unsigned long long* in_order = gathered;
    if (rem) {  // ranks >= rem hold one candidate less: their rows lose the padding
      unsigned long long* compact = gathered + static_cast<size_t>(world) * width;
      uint64_t at = 0;
      for (uint32_t r = 0; r < world; ++r) {
        const uint64_t have = r < rem ? width : width - 1;
        if (have) MCL_HIP(ctx, hipMemcpyAsync(compact + at, gathered + static_cast<size_t>(r) * width, have * sizeof(unsigned long long),
                                              hipMemcpyDeviceToDevice, ctx->stream));
        at += have;
      }
      in_order = compact;
    }
    blocks.push_back(Block{pos, cnt, held, m});
    held += m;
    uint64_t first_fail = ~0ull;
    if (const mcl_status s = mcl_kld_feed(ctx, reinterpret_cast<const uint64_t*>(in_order), cnt, &first_fail)) return s;
    if (first_fail != ~0ull) {
      n_out = first_fail;  // the first candidate failing the predicate is dropped (take_while)
      break;
    }
    pos += cnt;
    block *= 2;
  }
  n_out = std::min(n_out, max_p);  // | take(max)
  // re-balance: [0, n_out) becomes contiguous shards; what a rank receives from a block is one contiguous run of its new shard
  uint64_t new_first, new_n;
  shard_bounds(n_out, world, rank, &new_first, &new_n);
  MCL_HIP(ctx, ctx->d_new_shard.ensure(std::max<uint64_t>(4 * new_n, 4)));
  auto overlap = [](uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1) {
    const uint64_t lo = std::max(a0, b0), hi = std::min(a1, b1);
    return hi > lo ? hi - lo : 0;
  };
  std::vector<uint64_t> send(world), recv(world);
  for (const Block& b : blocks) {
    if (b.pos >= n_out) break;
    uint64_t received = 0;
    for (uint32_t q = 0; q < world; ++q) {
      uint64_t q_lo, q_m, span_first, span_n;
      shard_bounds(b.cnt, world, q, &q_lo, &q_m);
      shard_bounds(n_out, world, q, &span_first, &span_n);
      uint64_t my_lo, my_m;
      shard_bounds(b.cnt, world, rank, &my_lo, &my_m);
      send[q] = overlap(b.pos + my_lo, std::min(b.pos + my_lo + my_m, n_out), span_first, span_first + span_n) * 4 * sizeof(double);
      recv[q] = overlap(b.pos + q_lo, std::min(b.pos + q_lo + q_m, n_out), new_first, new_first + new_n) * 4 * sizeof(double);
      received += recv[q];
    }
    const uint64_t out0 = std::max(b.pos, new_first) - new_first;
    if (const mcl_status s = comm_exchange(ctx, ctx->d_cand_states.ptr + 4 * b.offset, send.data(),
                                           ctx->d_new_shard.ptr + 4 * std::min(out0, new_n), recv.data())) return s;
    (void)received;
  }
  if (const mcl_status s = mcl_load_shard(ctx, ctx->d_new_shard.ptr, new_n, new_first)) return s;
  *n_out_total = n_out;
  return MCL_OK;
}

// What can be refused without touching the filter's state (mcl_update checks it before the motion is consumed).
mcl_status sharded_preconditions(mcl_ctx* ctx) {
  const mcl_amcl_params& ap = ctx->cfg.amcl;
  if (ap.min_particles < ap.max_particles && ap.max_particles >= 0xFFFFFFFFull)
    return fail(ctx, MCL_ERR_UNSUPPORTED, "max_particles too large for KLD resampling");
  return MCL_OK;
}

// beluga::Amcl::update (amcl_core.hpp:165-201) over the sharded set; same statements as mcl_update, with the exchanges of
// include/beluga_mcl.h ("Particle shards") between them.  Every rank takes the same decisions: they depend on the control
// action (identical inputs) and on gathered sums (identical values, added in rank order everywhere).
mcl_status sharded_update(mcl_ctx* ctx, const Pose2& pose, const double* points_xy, uint64_t num_points, mcl_estimate* estimate,
                          mcl_update_info* info) {
  const mcl_amcl_params& ap = ctx->cfg.amcl;
  const uint32_t world = ctx->comm_world, rank = ctx->comm_rank;
  const bool adaptive = ap.min_particles < ap.max_particles;
  if (const mcl_status s = comm_scratch(ctx)) return s;
  if (ctx->global_n_unknown) {  // shards loaded by the caller: total = sum of the counts, this shard starts behind the ranks before it
    long long* d_counts = ctx->d_comm_i64.ptr;
    long long mine = static_cast<long long>(ctx->n);
    MCL_HIP(ctx, hipMemcpyAsync(d_counts, &mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (mine is a local)
    if (const mcl_status s = comm_gather(ctx, d_counts, d_counts + world, sizeof(long long))) return s;
    long long* h_counts = reinterpret_cast<long long*>(ctx->h_comm + kCommScalars + 64 * (1 + 3 + 2 + kEstRecord));
    MCL_HIP(ctx, hipMemcpyAsync(h_counts, d_counts + world, world * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t sum = 0, before = 0;
    for (uint32_t r = 0; r < world; ++r) {
      if (r < rank) before += static_cast<uint64_t>(h_counts[r]);
      sum += static_cast<uint64_t>(h_counts[r]);
    }
    ctx->global_n = sum;
    ctx->cfg.shard_offset = before;
    ctx->global_n_unknown = false;
  }
  const uint64_t n_total = ctx->global_n ? ctx->global_n : ap.max_particles;  // particles over all shards before this cycle's resampling
  if (n_total == 0) return MCL_OK;
  if (const mcl_status s = stage_points(ctx, points_xy, num_points)) return s;
  if (!ctx->have_window) {
    ctx->window0 = ctx->window1 = pose;
    ctx->have_window = true;
  } else {
    ctx->window1 = ctx->window0;
    ctx->window0 = pose;
  }
  ctx->step += 1;
  double* d = ctx->d_comm_f64.ptr;
  double* d_gather_sums = d + kCommScalars;            // [world]
  double* d_gather_stats = d_gather_sums + world;      // [world][3]
  double* d_intervals = d_gather_stats + 3 * world;    // ends[world], offsets[world]; behind them [world][9] estimate sums
  double* h = ctx->h_comm;

  MCL_HIP(ctx, hipMemsetAsync(d + 14, 0, sizeof(double), ctx->stream));  // this cycle's overflow flag (the fixed-capacity exchange)
  bool keys_ready = false;
  if (const mcl_status s = do_propagate(ctx, ctx->window0, ctx->window1, ctx->step, num_points, &keys_ready)) return s;  // :174-175
  if (const mcl_status s = do_reweight(ctx, points_xy, num_points, true, keys_ready)) return s;                         // :176
  ctx->every_n_current = (ctx->every_n_current + 1) % ap.resample_interval;  // every_n does not depend on data
  const bool fires = ctx->every_n_current == 0;
  // :177 normalise by the GLOBAL sum: shard sums gathered, added in rank order by every rank
  stage_begin(ctx, MCL_STAGE_NORMALIZE);
  launch_weight_sum(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(0), d + 0);
  if (const mcl_status s = comm_gather(ctx, d + 0, d_gather_sums, sizeof(double))) return s;
  launch_sum_rows(ctx->stream, d_gather_sums, world, 1, d + 4, nullptr);
  launch_normalize(ctx->stream, ctx->cur().w, ctx->n, d + 4, ctx->chunk_row(1), ctx->chunk_row(2), d + 2);
  stage_end(ctx, MCL_STAGE_NORMALIZE);
  if (fires) {
    launch_cdf(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(3), ctx->chunk_row(4), ctx->d_cdf.ptr, d + 1, ctx->d_cdf_tree.ptr,
               ctx->chunk_row(1));
  } else {
    MCL_HIP(ctx, hipMemsetAsync(d + 1, 0, sizeof(double), ctx->stream));
  }
  MCL_HIP(ctx, hipGetLastError());
  if (const mcl_status s = comm_gather(ctx, d + 1, d_gather_stats, 3 * sizeof(double))) return s;
  // A fixed-size cycle without selective resampling takes no decision that depends on the gathered numbers: the CDF intervals,
  // the totals and the recovery estimator are derived on the device by every rank (launch_shard_plan), the draw reads them
  // there, and the host reads them back with the estimate.  The cycle then has ONE host round trip before its end: the request
  // counts of the ancestor exchange (a collective's send / receive counts are host values).
  if (!adaptive && !ap.selective_resampling && ctx->tuning.device_policy != 0) {
    constexpr int kPolicySlot = 20;  // d_scalars[20..23) = {slow, fast, p}, as in the single-context cycle
    const RecoveryPolicy policy{ap.alpha_slow, ap.alpha_fast, fires ? 1 : 0, ctx->d_scalars.ptr + kPolicySlot, ctx->hd_scalars + kPolicySlot};
    double* d_plan = d + 16;  // {total, p}
    launch_shard_plan(ctx->stream, d_gather_stats, world, n_total, ctx->d_scalars.ptr + 1, ctx->hd_scalars + 1, policy, d_intervals, d_plan);
    launch_sum_rows(ctx->stream, d + 4, 1, 1, ctx->d_scalars.ptr + 0, ctx->hd_scalars + 0);  // the global weight sum, for the info
    MCL_HIP(ctx, hipGetLastError());
    // The ancestor exchange: fixed capacity per pair of ranks (no host read before the cycle's end) where the plain estimate follows -
    // its all-gather carries the overflow flags -, exact counts (one host read) otherwise.
    const bool padded = fires && ctx->estimate_kind == 0 && ctx->tuning.shard_pad_permille > 0 && world > 1;
    const uint64_t m = ctx->n, first_slot = ctx->cfg.shard_offset;
    auto commit = [&](uint64_t entries, bool injected_apart) -> mcl_status {
      const FreeCells fc{ctx->d_free.ptr, ctx->have_map ? ctx->n_free : 0};
      launch_commit_routed(ctx->stream, ctx->other(), ctx->cfg.seed, ctx->step, first_slot, entries, ctx->d_replies_in.ptr, ctx->d_route_order.ptr,
                           ctx->d_targets.ptr, ctx->grid_view(), fc);
      // (the fixed-capacity exchange routes no injected slot: sharded_draw_padded / k_route_hist)
      if (injected_apart) launch_commit_injected(ctx->stream, ctx->other(), ctx->cfg.seed, ctx->step, first_slot, m, ctx->d_targets.ptr, ctx->grid_view(), fc);
      MCL_HIP(ctx, hipGetLastError());
      ctx->live ^= 1;
      ctx->n = m;
      ctx->weights_unit = true;  // (every output slot took a weight of 1.0: particle_traits.hpp:105)
      return MCL_OK;
    };
    if (fires) {
      stage_begin(ctx, MCL_STAGE_RESAMPLE);
      if (padded) {
        uint64_t entries = 0;
        if (const mcl_status s = sharded_draw_padded(ctx, d_intervals, first_slot, m, d_plan,
                                                    padded_capacity(n_total, world, static_cast<uint32_t>(ctx->tuning.shard_pad_permille)), &entries)) return s;
        if (const mcl_status s = commit(entries, true)) return s;
      } else {
        if (const mcl_status s = sharded_draw(ctx, 0.0, 0.0, d_intervals, first_slot, m, d_plan)) return s;
        if (const mcl_status s = commit(m, false)) return s;
      }
      stage_end(ctx, MCL_STAGE_RESAMPLE);
    }
    ctx->force_update = false;  // :199
    stage_begin(ctx, MCL_STAGE_ESTIMATE);
    mcl_estimate est{};
    if (ctx->estimate_kind == 1) {
      if (const mcl_status s = do_cluster_estimate(ctx, ctx->cluster_params, &est)) return s;
      stage_end(ctx, MCL_STAGE_ESTIMATE);
      MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
      ctx->comm_host_syncs += 1;
    } else {
      double sums[12];
      if (const mcl_status s = sharded_estimate_sums(ctx, nullptr, 0, sums)) return s;
      if (padded && ctx->h_scalars[17] != 0.0) {
        // Some rank's requests to one shard did not fit the fixed capacity (every rank reads the same sum of flags and gets here
        // together): the new set is incomplete.  The old one and its CDF are untouched - the commit wrote the other buffer -: back to
        // it, the exchange again with exact counts, the estimate again.
        ctx->comm_overflows += 1;
        ctx->live ^= 1;
        ctx->weights_unit = false;  // (the old set again: normalised weights)
        stage_end(ctx, MCL_STAGE_ESTIMATE);  // (one stage open at a time: the retry's resampling is timed as resampling)
        stage_begin(ctx, MCL_STAGE_RESAMPLE);
        mcl_status retry = sharded_draw(ctx, 0.0, 0.0, d_intervals, first_slot, m, d_plan);
        if (retry == MCL_OK) retry = commit(m, false);
        stage_end(ctx, MCL_STAGE_RESAMPLE);
        if (retry != MCL_OK) return retry;
        stage_begin(ctx, MCL_STAGE_ESTIMATE);
        if (const mcl_status s = sharded_estimate_sums(ctx, nullptr, 0, sums)) {
          stage_end(ctx, MCL_STAGE_ESTIMATE);
          return s;
        }
      }
      stage_end(ctx, MCL_STAGE_ESTIMATE);
      if (const mcl_status s = mcl_estimate_from_sums(sums, &est)) return s;
    }
    stage_collect(ctx);
    if (std::isfinite(est.pose[2]) && std::isfinite(est.pose[3])) {
      ctx->pivot[0] = est.pose[2];
      ctx->pivot[1] = est.pose[3];
    }
    remember_cloud_estimate(ctx, est);
    if (estimate) *estimate = est;
    if (info) {
      info->updated = 1;
      info->resampled = fires ? 1 : 0;
      info->num_particles = n_total;
      info->weight_sum = ctx->h_scalars[0];
      info->effective_sample_size = -1.0;
      info->random_state_probability = ctx->h_scalars[kPolicySlot + 2];
    }
    return MCL_OK;
  }
  MCL_HIP(ctx, hipMemcpyAsync(h, d + 4, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipMemcpyAsync(h + 1, d_gather_stats, 3 * world * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->comm_host_syncs += 1;
  stage_collect(ctx);
  const double weight_sum = h[0];
  double norm_sum = 0.0, norm_sumsq = 0.0;
  for (uint32_t r = 0; r < world; ++r) {
    norm_sum += h[1 + 3 * r + 1];
    norm_sumsq += h[1 + 3 * r + 2];
  }
  // :179 ThrunRecoveryProbabilityEstimator on the normalised weights
  double random_state_probability = 0.0;
  {
    const double average = norm_sum / static_cast<double>(n_total);
    const double fast_average = ctx->fast(average), slow_average = ctx->slow(average);
    if (std::abs(slow_average) >= std::numeric_limits<double>::epsilon())
      random_state_probability = std::clamp(1.0 - fast_average / slow_average, 0.0, 1.0);
  }
  bool do_resampling = fires;
  double ess = -1.0;
  if (do_resampling && ap.selective_resampling) {  // :181 && on_effective_size_drop
    ess = norm_sum == 0.0 ? 0.0 : (norm_sum * norm_sum) / norm_sumsq;
    do_resampling = ess < static_cast<double>(n_total) * 0.5;
  }
  if (do_resampling) {
    if (random_state_probability > 0.0) {  // :184-186
      ctx->slow.reset();
      ctx->fast.reset();
    }
    stage_begin(ctx, MCL_STAGE_RESAMPLE);
    // intervals of the global CDF: ends[r] = inclusive end of shard r, offsets[r] = its start
    double* up = h + 1 + 3 * world;
    double run = 0.0;
    for (uint32_t r = 0; r < world; ++r) {
      up[world + r] = run;
      run += h[1 + 3 * r];
      up[r] = run;
    }
    const double total = run;
    MCL_HIP(ctx, hipMemcpyAsync(d_intervals, up, 2 * world * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (adaptive) {
      uint64_t kept = 0;
      if (const mcl_status s = sharded_resample_kld(ctx, random_state_probability, total, d_intervals, &kept)) return s;
      ctx->global_n = kept;
    } else {
      const uint64_t m = ctx->n, first_slot = ctx->cfg.shard_offset;
      if (const mcl_status s = sharded_draw(ctx, random_state_probability, total, d_intervals, first_slot, m)) return s;
      if (const mcl_status s = mcl_commit_routed(ctx, ctx->step, first_slot, m, ctx->d_replies_in.ptr, ctx->d_route_order.ptr, ctx->d_targets.ptr)) return s;
    }
    stage_end(ctx, MCL_STAGE_RESAMPLE);
  }
  ctx->force_update = false;  // :199
  // :200 estimate: nine sums per shard, gathered, added in rank order; or cluster_based_estimate over the gathered cells
  // (beluga_ros/src/amcl.cpp:125)
  stage_begin(ctx, MCL_STAGE_ESTIMATE);
  mcl_estimate est{};
  if (ctx->estimate_kind == 1) {
    if (const mcl_status s = do_cluster_estimate(ctx, ctx->cluster_params, &est)) return s;
    stage_end(ctx, MCL_STAGE_ESTIMATE);
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } else {
    double sums[12];
    if (const mcl_status s = sharded_estimate_sums(ctx, nullptr, 0, sums)) return s;
    stage_end(ctx, MCL_STAGE_ESTIMATE);
    if (const mcl_status s = mcl_estimate_from_sums(sums, &est)) return s;
  }
  stage_collect(ctx);
  if (std::isfinite(est.pose[2]) && std::isfinite(est.pose[3])) {
    ctx->pivot[0] = est.pose[2];
    ctx->pivot[1] = est.pose[3];
  }
  remember_cloud_estimate(ctx, est);
  if (estimate) *estimate = est;
  if (info) {
    info->updated = 1;
    info->resampled = do_resampling ? 1 : 0;
    info->num_particles = ctx->global_n ? ctx->global_n : ap.max_particles;
    info->weight_sum = weight_sum;
    info->effective_sample_size = ess;
    info->random_state_probability = random_state_probability;
  }
  return MCL_OK;
}

// ---- built-in RCCL transport (librccl.so loaded at run time) ------------------------------------------------------------
struct RcclId {  // ncclUniqueId: 128 opaque bytes, passed by value
  char internal[128];
};
struct RcclApi {
  void* lib{nullptr};
  int (*GetUniqueId)(void*){nullptr};
  int (*CommInitRank)(void**, int, RcclId, int){nullptr};
  int (*CommDestroy)(void*){nullptr};
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t){nullptr};
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t){nullptr};
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t){nullptr};
  int (*GroupStart)(){nullptr};
  int (*GroupEnd)(){nullptr};
  int (*CommCount)(void*, int*){nullptr};  // optional: how many ranks the communicator itself says it has
};
RcclApi* rccl_api(std::string* error) {
  static RcclApi api;
  static bool tried = false;
  static std::string load_error;
  if (!tried) {
    tried = true;
    const char* candidates[] = {std::getenv("BELUGA_MCL_RCCL"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    // a copy that is already in the process (torch ships one) is used first: two RCCLs in one process do not mix
    for (const char* name : {"librccl.so", "librccl.so.1"})
      if (!api.lib) api.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : candidates)
      if (!api.lib && name) api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) {
      load_error = "librccl.so not found (set BELUGA_MCL_RCCL to its path)";
    } else {
      auto sym = [&](const char* n) { return dlsym(api.lib, n); };
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
      api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
      api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
      api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
      api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
      api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
      api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
        load_error = "librccl.so lacks an expected symbol";
    }
  }
  if (!load_error.empty()) {
    if (error) *error = load_error;
    return nullptr;
  }
  return &api;
}
constexpr int kNcclChar = 0;  // ncclInt8 / ncclChar
using RcclUser = mcl_ctx::RcclUserStorage;
int32_t rccl_all_gather(void* user, const void* d_send, void* d_recv, uint64_t bytes, void* stream) {
  auto* u = static_cast<RcclUser*>(user);
  return rccl_api(nullptr)->AllGather(d_send, d_recv, bytes, kNcclChar, u->comm, static_cast<hipStream_t>(stream));
}
int32_t rccl_all_to_all(void* user, const void* d_send, const uint64_t* send_bytes, void* d_recv, const uint64_t* recv_bytes, void* stream) {
  auto* u = static_cast<RcclUser*>(user);
  RcclApi* api = rccl_api(nullptr);
  int rc = api->GroupStart();
  const char* out = static_cast<const char*>(d_send);
  char* in = static_cast<char*>(d_recv);
  for (uint32_t q = 0; q < u->world && rc == 0; ++q) {
    if (send_bytes[q]) rc = api->Send(out, send_bytes[q], kNcclChar, static_cast<int>(q), u->comm, static_cast<hipStream_t>(stream));
    if (rc == 0 && recv_bytes[q]) rc = api->Recv(in, recv_bytes[q], kNcclChar, static_cast<int>(q), u->comm, static_cast<hipStream_t>(stream));
    out += send_bytes[q];
    in += recv_bytes[q];
  }
  const int end = api->GroupEnd();
  return rc ? rc : end;
}

// (mcl_set_map_async: defined beside mcl_set_map)
void drop_pending_map(mcl_ctx* ctx);
mcl_status apply_pending_map(mcl_ctx* ctx, bool wait);

}  // namespace

extern "C" {

const char* mcl_version(void) { return "beluga_mcl 0.3 (gfx950)"; }

uint32_t mcl_debug_curve_index(uint32_t heading_bin, uint32_t y_bin, uint32_t x_bin, uint32_t bits) {
  if (bits < 1 || bits > 6) return 0xFFFFFFFFu;
  return hilbert_index_3(heading_bin, y_bin, x_bin, bits); }

void mcl_default_config(mcl_config* cfg) {
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->sensor_kind = MCL_SENSOR_LIKELIHOOD_FIELD;
  cfg->amcl.update_min_d = 0.25;
  cfg->amcl.update_min_a = 0.2;
  cfg->amcl.resample_interval = 1;
  cfg->amcl.selective_resampling = 0;
  cfg->amcl.min_particles = 500;
  cfg->amcl.max_particles = 2000;
  cfg->amcl.alpha_slow = 0.001;
  cfg->amcl.alpha_fast = 0.1;
  cfg->amcl.kld_epsilon = 0.05;
  cfg->amcl.kld_z = 3.0;
  cfg->amcl.spatial_resolution_x = 0.5;
  cfg->amcl.spatial_resolution_y = 0.5;
  cfg->amcl.spatial_resolution_theta = 10.0 * kPi / 180.0;
  cfg->motion.distance_threshold = 0.01;
  cfg->lf = mcl_lf_params{100.0, 2.0, 0.5, 0.5, 0.2, 0, 0};
  cfg->beam = mcl_beam_params{0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 60.0};
}

const char* mcl_last_error(const mcl_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

mcl_status mcl_create(const mcl_config* cfg, mcl_ctx** out) {
  if (!cfg || !out) return fail(nullptr, MCL_ERR_INVALID_ARGUMENT, "mcl_create: null argument");
  *out = nullptr;
  if (cfg->amcl.max_particles == 0 || cfg->amcl.resample_interval == 0)
    return fail(nullptr, MCL_ERR_INVALID_ARGUMENT, "mcl_create: max_particles and resample_interval must be > 0");
  if (cfg->sensor_kind < MCL_SENSOR_LIKELIHOOD_FIELD || cfg->sensor_kind > MCL_SENSOR_LIKELIHOOD_FIELD_PROB)
    return fail(nullptr, MCL_ERR_INVALID_ARGUMENT, "mcl_create: unknown sensor_kind");
  if (cfg->motion_kind < MCL_MOTION_DIFFERENTIAL || cfg->motion_kind > MCL_MOTION_STATIONARY)
    return fail(nullptr, MCL_ERR_INVALID_ARGUMENT, "mcl_create: unknown motion_kind");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    return fail(nullptr, MCL_ERR_NO_DEVICE, "mcl_create: no HIP device (this library has no CPU fallback)");
  if (cfg->device_id < 0 || cfg->device_id >= count) return fail(nullptr, MCL_ERR_INVALID_ARGUMENT, "mcl_create: bad device_id");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return fail(nullptr, MCL_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, MCL_ERR_NO_DEVICE, std::string("mcl_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);

  mcl_ctx* ctx = new (std::nothrow) mcl_ctx();
  if (!ctx) return fail(nullptr, MCL_ERR_OUT_OF_MEMORY, "mcl_create: host allocation failed");
  ctx->cfg = *cfg;
  ctx->device = cfg->device_id;
  ctx->slow.alpha = cfg->amcl.alpha_slow;
  ctx->fast.alpha = cfg->amcl.alpha_fast;
  mcl_status st = MCL_OK;
  auto init = [&]() -> mcl_status {
    MCL_HIP(ctx, hipSetDevice(ctx->device));
    if (cfg->hip_stream) {
      ctx->stream = static_cast<hipStream_t>(cfg->hip_stream);
    } else {
      MCL_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = true;
    }
    const uint64_t cap = cfg->shard_capacity ? cfg->shard_capacity : cfg->amcl.max_particles;
    if (const mcl_status s = ensure_capacity(ctx, cap)) return s;
    MCL_HIP(ctx, ctx->d_scalars.ensure(32));
    MCL_HIP(ctx, hipMemsetAsync(ctx->d_scalars.ptr, 0, 32 * sizeof(double), ctx->stream));  // incl. the recovery filters
    MCL_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_scalars), 32 * sizeof(double), hipHostMallocMapped));
    std::memset(ctx->h_scalars, 0, 32 * sizeof(double));  // host mirrors are read before their first kernel has written them
    MCL_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->hd_scalars), ctx->h_scalars, 0));
    MCL_HIP(ctx, ctx->d_kld_scalars.ensure(8));
    MCL_HIP(ctx, hipMemsetAsync(ctx->d_kld_scalars.ptr, 0, 8 * sizeof(unsigned long long), ctx->stream));
    MCL_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_kld_scalars), 8 * sizeof(unsigned long long)));
    for (auto& pair : ctx->ev)
      for (auto& e : pair) MCL_HIP(ctx, hipEventCreate(&e));
    MCL_HIP(ctx, hipEventCreateWithFlags(&ctx->points_event, hipEventDisableTiming));
    configure_device_kernels();
    {  // the device's compute units: the queue form of the LF patch kernel launches three workgroups per CU
      int count = 0;
      if (hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || count <= 0) count = 256;
      ctx->tuning.device_cus = count;
    }
    // Environment defaults of the per-context switches (mcl_set_option changes them at run time).
    for (const char* name : {"lf_variant", "lf_fast", "lf_table", "lf_patch", "lf_dispersed", "lf_far_tiles", "key_layout", "lf_loose_below", "lf_small_particles", "device_policy",
                             "sort_min_particles", "beam_sort_min_particles", "field_build", "key_curve", "key_warp", "key_bits_xy", "lf_margin", "lf_split", "lf_queue_grid", "shard_pad_permille", "lf_queue", "lf_ends_first", "beam_free_ahead", "beam_sectors", "lf_weight_sums", "beam_table", "cycle_spin", "scan_fused", "draw_fold", "lf_unit_weights", "small_fused", "norm_store", "noise_ahead", "order_ahead", "lf_far_beams_per_wave"}) {
      std::string env = "BELUGA_MCL_";
      for (const char* c = name; *c; ++c) env += static_cast<char>(std::toupper(static_cast<unsigned char>(*c)));
      if (const char* v = std::getenv(env.c_str())) {
        if (std::string(name) == "lf_table") (void)mcl_set_option(ctx, name, std::string(v) == "cube" ? 1 : std::atoi(v));
        else (void)mcl_set_option(ctx, name, std::atoi(v));
      }
    }
    return MCL_OK;
  };
  st = init();
  if (st != MCL_OK) {
    g_create_error = ctx->error;
    mcl_destroy(ctx);
    return st;
  }
  *out = ctx;
  return MCL_OK;
}

void mcl_destroy(mcl_ctx* ctx) {
  if (ctx) drop_pending_map(ctx);
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto& set : ctx->sets) set.release();
  ctx->d_field.release();
  ctx->d_field_scratch.release();
  ctx->d_cube.release();
  ctx->d_pal_idx.release();
  ctx->d_far_bits.release();
  ctx->d_far_linear.release();
  ctx->d_far_votes.release();
  ctx->d_pal_val.release();
  ctx->d_pal_keys.release();
  ctx->d_cells.release();
  ctx->d_nonfree_bits.release();
  ctx->d_free.release();
  ctx->d_points.release();
  ctx->d_beam_points.release();
  ctx->d_beam_table.release();
  ctx->d_chunk.release();
  ctx->d_scalars.release();
  ctx->d_cdf.release();
  ctx->d_cdf_tree.release();
  ctx->d_lf_wsum.release();
  ctx->d_scan_state.release();
  ctx->d_noise.release();
  ctx->d_cloud.release();
  ctx->d_est_partials.release();
  ctx->d_cloud_w.release();
  ctx->d_hashes.release();
  ctx->d_table_keys.release();
  ctx->d_table_first.release();
  ctx->d_flags.release();
  ctx->d_uchunk.release();
  ctx->d_kld_scalars.release();
  ctx->d_sort_u32.release();
  ctx->d_route_u32.release();
  ctx->d_cell_f64.release();
  ctx->d_cell_u32.release();
  ctx->d_cell_u64.release();
  ctx->d_cell_exchange.release();
  ctx->d_sort_u64.release();
  ctx->d_sort_f64.release();
  if (ctx->rccl_comm) {
    if (RcclApi* api = rccl_api(nullptr)) (void)api->CommDestroy(ctx->rccl_comm);
  }
  if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
  if (ctx->h_cells) (void)hipHostFree(ctx->h_cells);
  ctx->d_comm_f64.release();
  ctx->d_comm_i64.release();
  ctx->d_targets.release();
  ctx->d_send_targets.release();
  ctx->d_requests_in.release();
  ctx->d_replies_out.release();
  ctx->d_replies_in.release();
  ctx->d_route_order.release();
  ctx->d_cand_states.release();
  ctx->d_new_shard.release();
  ctx->d_cand_hashes.release();
  ctx->d_block_hashes.release();
  if (ctx->h_points) (void)hipHostFree(ctx->h_points);
  if (ctx->points_event) (void)hipEventDestroy(ctx->points_event);
  if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
  if (ctx->h_kld_scalars) (void)hipHostFree(ctx->h_kld_scalars);
  for (auto& pair : ctx->ev)
    for (auto& e : pair)
      if (e) (void)hipEventDestroy(e);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

}  // extern "C"
namespace {
void drop_pending_map(mcl_ctx* ctx) {
  if (!ctx->pending_map) return;
  if (ctx->pending_map->worker.joinable()) ctx->pending_map->worker.join();
  delete ctx->pending_map;
  ctx->pending_map = nullptr;
}
// prebuilt_field (mcl_set_map_async): the likelihood field of exactly these cells and the context's parameters, built ahead by the worker
mcl_status set_map_impl(mcl_ctx* ctx, const int8_t* cells, uint32_t width, uint32_t height, double resolution, const double origin[4],
                        const int8_t value_traits[3], std::vector<float>* prebuilt_field);
// The swap of a map built ahead, where one is ready (the start of mcl_update; mcl_map_commit).
mcl_status apply_pending_map(mcl_ctx* ctx, bool wait) {
  mcl_ctx::PendingMap* p = ctx->pending_map;
  if (!p) return MCL_OK;
  if (p->state.load(std::memory_order_acquire) != 2 && !wait) return MCL_OK;
  if (p->worker.joinable()) p->worker.join();
  const mcl_status s = set_map_impl(ctx, p->cells.data(), p->W, p->H, p->resolution, p->origin, p->traits, p->field.empty() ? nullptr : &p->field);
  drop_pending_map(ctx);
  return s;
}
}  // namespace
extern "C" {

mcl_status mcl_set_map(mcl_ctx* ctx, const int8_t* cells, uint32_t width, uint32_t height, double resolution,
                       const double origin[4], const int8_t value_traits[3]) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, cells && origin && value_traits && width > 0 && height > 0 && resolution > 0, "mcl_set_map: bad argument");
  MCL_REQUIRE(ctx, static_cast<uint64_t>(width) * height < 0xFFFFFFFFull, "mcl_set_map: grid too large");
  drop_pending_map(ctx);  // (a map given now replaces one that is still on its way)
  return set_map_impl(ctx, cells, width, height, resolution, origin, value_traits, nullptr);
}

mcl_status mcl_set_map_async(mcl_ctx* ctx, const int8_t* cells, uint32_t width, uint32_t height, double resolution,
                             const double origin[4], const int8_t value_traits[3]) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, cells && origin && value_traits && width > 0 && height > 0 && resolution > 0, "mcl_set_map_async: bad argument");
  MCL_REQUIRE(ctx, static_cast<uint64_t>(width) * height < 0xFFFFFFFFull, "mcl_set_map_async: grid too large");
  if (ctx->have_comm && ctx->comm_world > 1)
    return fail(ctx, MCL_ERR_UNSUPPORTED, "mcl_set_map_async: not on a sharded filter (the ranks would swap maps in different cycles)");
  drop_pending_map(ctx);
  auto* p = new (std::nothrow) mcl_ctx::PendingMap();
  if (!p) return fail(ctx, MCL_ERR_OUT_OF_MEMORY, "mcl_set_map_async: out of memory");
  try {
    p->cells.assign(cells, cells + static_cast<size_t>(width) * height);
  } catch (const std::bad_alloc&) {
    delete p;
    return fail(ctx, MCL_ERR_OUT_OF_MEMORY, "mcl_set_map_async: out of memory");
  }
  p->W = width;
  p->H = height;
  p->resolution = resolution;
  std::memcpy(p->origin, origin, sizeof p->origin);
  std::memcpy(p->traits, value_traits, sizeof p->traits);
  ctx->pending_map = p;
  if (ctx->cfg.sensor_kind != MCL_SENSOR_BEAM && ctx->tuning.field_build == 0) {
    p->state.store(1, std::memory_order_release);
    const mcl_lf_params lf = ctx->cfg.lf;
    p->worker = std::thread([p, lf] {
      build_likelihood_field(p->cells.data(), p->W, p->H, p->resolution, OccupancyTraits{p->traits[0], p->traits[1], p->traits[2]}, lf, p->field);
      p->state.store(2, std::memory_order_release);
    });
  } else {
    p->state.store(2, std::memory_order_release);  // (nothing to build on the host: the swap does it all)
  }
  return MCL_OK;
}

mcl_status mcl_map_pending(mcl_ctx* ctx, int32_t* state) {
  if (!ctx || !state) return MCL_ERR_INVALID_ARGUMENT;
  *state = ctx->pending_map ? ctx->pending_map->state.load(std::memory_order_acquire) : 0;
  return MCL_OK;
}

mcl_status mcl_map_commit(mcl_ctx* ctx, int32_t wait) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  return apply_pending_map(ctx, wait != 0);
}

}  // extern "C"
namespace {
mcl_status set_map_impl(mcl_ctx* ctx, const int8_t* cells, uint32_t width, uint32_t height, double resolution, const double origin[4],
                        const int8_t value_traits[3], std::vector<float>* prebuilt_field) {
  if (const mcl_status s = bind_device(ctx)) return s;
  const size_t n = static_cast<size_t>(width) * height;
  ctx->W = width;
  ctx->H = height;
  ctx->resolution = resolution;
  ctx->origin = pose_from(origin);
  ctx->origin_inverse = pose_inverse(ctx->origin);  // likelihood_field_model_base.hpp:99 ; raycasting.hpp:69
  ctx->traits = OccupancyTraits{value_traits[0], value_traits[1], value_traits[2]};
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MCL_HIP(ctx, ctx->d_cells.ensure(n));
  MCL_HIP(ctx, hipMemcpy(ctx->d_cells.ptr, cells, n, hipMemcpyHostToDevice));
  std::vector<uint32_t> free_cells;
  collect_free_cells(cells, width, height, ctx->traits, free_cells);
  ctx->n_free = free_cells.size();
  MCL_HIP(ctx, ctx->d_free.ensure(std::max<size_t>(free_cells.size(), 1)));
  if (!free_cells.empty())
    MCL_HIP(ctx, hipMemcpy(ctx->d_free.ptr, free_cells.data(), free_cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (ctx->cfg.sensor_kind == MCL_SENSOR_BEAM) {
    MCL_REQUIRE(ctx, n < (1ull << 31), "mcl_set_map: beam model grids are limited to 2^31 cells");
    MCL_HIP(ctx, ctx->d_nonfree_bits.ensure(nonfree_words(width, height)));
    launch_pack_nonfree(ctx->stream, ctx->d_cells.ptr, width, height, ctx->traits.free_value, ctx->d_nonfree_bits.ptr);
    MCL_HIP(ctx, hipGetLastError());
    {
      const mcl_beam_params& b = ctx->cfg.beam;
      // (the table itself - 32 bytes per squared cell distance up to the range, 46 MB at 60 m / 5 cm - is built by the first
      // launch of the ordered kernel that wants it: the usual 2000-particle filter never does; do_reweight)
      ctx->beam_table_count = beam_table_entries(b.beam_max_range, resolution);
      ctx->beam_table_ready = false;
    }
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (ctx->cfg.sensor_kind != MCL_SENSOR_BEAM) {
    MCL_HIP(ctx, ctx->d_field.ensure(n));
    bool on_device = false;
    ctx->field_build_ms = 0.0;
    if (ctx->tuning.field_build == 1) {
      // Exact Euclidean distance transform on the device (kernels.hip, "likelihood field on the device"); equal to the
      // reference's wavefront at all but a few cells.  The default below is the bit-identical host wavefront.
      const mcl_lf_params& lf = ctx->cfg.lf;
      const FieldBuildParams fp{lf.max_obstacle_distance, lf.max_laser_distance, lf.z_hit, lf.z_random, lf.sigma_hit,
                                lf.model_unknown_space, lf.only_obstacle_boundaries};
      MCL_HIP(ctx, ctx->d_field_scratch.ensure(n));
      MCL_HIP(ctx, hipEventRecord(ctx->ev[MCL_STAGE_REWEIGHT][0], ctx->stream));
      on_device = launch_build_field(ctx->stream, ctx->d_cells.ptr, width, height, resolution, ctx->traits.free_value, ctx->traits.unknown_value,
                                     ctx->traits.occupied_value, fp, reinterpret_cast<uint16_t*>(ctx->d_field_scratch.ptr),
                                     reinterpret_cast<int16_t*>(ctx->d_field_scratch.ptr) + n, ctx->d_field.ptr);
      MCL_HIP(ctx, hipGetLastError());
      if (on_device) {
        MCL_HIP(ctx, hipEventRecord(ctx->ev[MCL_STAGE_REWEIGHT][1], ctx->stream));
        ctx->h_field.resize(n);
        MCL_HIP(ctx, hipMemcpyAsync(ctx->h_field.data(), ctx->d_field.ptr, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[MCL_STAGE_REWEIGHT][0], ctx->ev[MCL_STAGE_REWEIGHT][1]) == hipSuccess) ctx->field_build_ms = ms;
      }
    }
    if (!on_device) {
      if (prebuilt_field) ctx->h_field.swap(*prebuilt_field);
      else build_likelihood_field(cells, width, height, resolution, ctx->traits, ctx->cfg.lf, ctx->h_field);
      MCL_HIP(ctx, hipMemcpy(ctx->d_field.ptr, ctx->h_field.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    ctx->field_built_on_device = on_device;
    if (const mcl_status s = rebuild_cube(ctx, ctx->h_field.data())) return s;
  }
  ctx->have_map = true;
  return MCL_OK;
}
}  // namespace
extern "C" {

mcl_status mcl_get_likelihood_field(mcl_ctx* ctx, float* out) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, out, "null output");
  if (!ctx->have_map || !ctx->d_field.ptr) return fail(ctx, MCL_ERR_NOT_READY, "no likelihood field");
  if (const mcl_status s = bind_device(ctx)) return s;
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MCL_HIP(ctx, hipMemcpy(out, ctx->d_field.ptr, static_cast<size_t>(ctx->W) * ctx->H * sizeof(float), hipMemcpyDeviceToHost));
  return MCL_OK;
}

mcl_status mcl_set_likelihood_field(mcl_ctx* ctx, const float* field) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, field, "null field");
  if (!ctx->have_map) return fail(ctx, MCL_ERR_NOT_READY, "set the map first");
  if (const mcl_status s = bind_device(ctx)) return s;
  const size_t n = static_cast<size_t>(ctx->W) * ctx->H;
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MCL_HIP(ctx, ctx->d_field.ensure(n));
  MCL_HIP(ctx, hipMemcpy(ctx->d_field.ptr, field, n * sizeof(float), hipMemcpyHostToDevice));
  return rebuild_cube(ctx, field);
}

mcl_status mcl_initialize_normal(mcl_ctx* ctx, const double mean_xytheta[3], const double cov[9]) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->lf_wsum_count = 0;  // (workgroup sums of an earlier reweight describe another set)
  ctx->order_valid = false;  // (and an order computed ahead describes the particles of another one)
  MCL_REQUIRE(ctx, mean_xytheta && cov, "null argument");
  double T[9];
  if (!covariance_to_transform(cov, T)) return fail(ctx, MCL_ERR_BAD_COVARIANCE, "Invalid covariance matrix");
  if (const mcl_status s = bind_device(ctx)) return s;
  const uint64_t n = std::min<uint64_t>(ctx->cfg.amcl.max_particles, ctx->capacity);  // take_exactly(max_particles)
  ctx->weights_unit = false;
  launch_init_normal(ctx->stream, ctx->cur(), n, mean_xytheta, T, ctx->cfg.seed, ctx->cfg.shard_offset);
  MCL_HIP(ctx, hipGetLastError());
  ctx->weights_unit = true;  // (k_init_normal writes 1.0 to every weight)
  ctx->n = n;
  ctx->global_n = 0;  // (shards: every rank holds its share of max_particles again)
  ctx->global_n_unknown = false;
  ctx->force_update = true;  // amcl_core.hpp:136
  for (int k = 0; k < 3; ++k) {
    ctx->cloud_mean[k] = mean_xytheta[k];
    ctx->cloud_sigma[k] = std::sqrt(std::max(cov[4 * k], 0.0));
  }
  ctx->have_cloud_estimate = true;
  // a new set: whatever earlier launches reported about the old one is history; the next LF launch finds out
  patch_totals(ctx, &ctx->patch_seen_planned, &ctx->patch_seen_through);
  ctx->patch_useful = true;
  return MCL_OK;
}

mcl_status mcl_set_particles(mcl_ctx* ctx, const double* states, const double* weights, uint64_t n) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->lf_wsum_count = 0;  // (workgroup sums of an earlier reweight describe another set)
  ctx->order_valid = false;  // (and an order computed ahead describes the particles of another one)
  MCL_REQUIRE(ctx, n == 0 || (states && weights), "null argument");
  MCL_REQUIRE(ctx, n <= ctx->capacity, "mcl_set_particles: n exceeds capacity");
  if (const mcl_status s = bind_device(ctx)) return s;
  if (n) {
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MCL_HIP(ctx, hipMemcpy(ctx->cur().pose, states, n * 4 * sizeof(double), hipMemcpyHostToDevice));  // same record layout
    MCL_HIP(ctx, hipMemcpy(ctx->cur().w, weights, n * sizeof(double), hipMemcpyHostToDevice));
  }
  ctx->weights_unit = false;
  ctx->n = n;
  ctx->global_n_unknown = ctx->have_comm && ctx->comm_world > 1;  // a shard loaded by the caller: the ranks compare notes first
  ctx->force_update = true;
  ctx->have_cloud_estimate = false;  // the ordering falls back to a bounding-box pass until the next estimate
  patch_totals(ctx, &ctx->patch_seen_planned, &ctx->patch_seen_through);  // (as in mcl_initialize_normal)
  ctx->patch_useful = true;
  return MCL_OK;
}

mcl_status mcl_num_particles(mcl_ctx* ctx, uint64_t* n) {
  if (!ctx || !n) return MCL_ERR_INVALID_ARGUMENT;
  *n = ctx->n;
  return MCL_OK;
}

mcl_status mcl_get_particles(mcl_ctx* ctx, double* states, double* weights, uint64_t capacity, uint64_t* n) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, capacity >= ctx->n, "mcl_get_particles: output too small");
  if (const mcl_status s = bind_device(ctx)) return s;
  if (ctx->n) {
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (states) MCL_HIP(ctx, hipMemcpy(states, ctx->cur().pose, ctx->n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (weights) MCL_HIP(ctx, hipMemcpy(weights, ctx->cur().w, ctx->n * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (n) *n = ctx->n;
  return MCL_OK;
}

mcl_status mcl_force_update(mcl_ctx* ctx) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->force_update = true;
  return MCL_OK;
}

mcl_status mcl_propagate(mcl_ctx* ctx, const double pose[4], const double previous_pose[4], uint32_t step) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, pose && previous_pose, "null argument");
  if (const mcl_status s = bind_device(ctx)) return s;
  return do_propagate(ctx, pose_from(pose), pose_from(previous_pose), step);
}

mcl_status mcl_reweight(mcl_ctx* ctx, const double* points_xy, uint64_t num_points) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, num_points == 0 || points_xy, "null points");
  if (const mcl_status s = bind_device(ctx)) return s;
  return do_reweight(ctx, points_xy, num_points);
}

mcl_status mcl_weight_sum(mcl_ctx* ctx, double* sum) {
  if (!ctx || !sum) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  launch_weight_sum(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(0), ctx->d_scalars.ptr + 0);
  MCL_HIP(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars.ptr, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *sum = ctx->h_scalars[0];
  return MCL_OK;
}

mcl_status mcl_normalize(mcl_ctx* ctx, double factor, mcl_weight_stats* stats) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  return do_normalize(ctx, factor, stats);
}

mcl_status mcl_resample(mcl_ctx* ctx, double random_state_probability, uint32_t step, uint64_t* n_out) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  const mcl_status s = do_resample(ctx, random_state_probability, step, n_out);
  if (s == MCL_OK) {
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    stage_collect(ctx);
  }
  return s;
}

mcl_status mcl_estimate_sums(mcl_ctx* ctx, const double pivot_xy[2], double sums[12]) {
  if (!ctx || !sums) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  const double zero[2] = {0, 0};
  return do_estimate_sums(ctx, pivot_xy ? pivot_xy : zero, sums);
}

// algorithm/estimation.hpp:436-475 from the single-pass sufficient statistics.
mcl_status mcl_estimate_from_sums(const double sums[12], mcl_estimate* out) {
  if (!sums || !out) return MCL_ERR_INVALID_ARGUMENT;
  const double sw = sums[0], sw2 = sums[1];
  const double mc = sums[2] / sw, ms = sums[3] / sw;
  const double mdx = sums[4] / sw, mdy = sums[5] / sw;
  const double sq = sw2 / (sw * sw);  // sum of squared normalised weights
  const double corr = 1.0 - sq;       // estimation.hpp:270
  const double cxx = (sums[6] / sw - mdx * mdx) / corr;
  const double cxy = (sums[7] / sw - mdx * mdy) / corr;
  const double cyy = (sums[8] / sw - mdy * mdy) / corr;
  for (double& v : out->covariance) v = 0.0;
  out->covariance[0] = cxx;
  out->covariance[1] = cxy;
  out->covariance[3] = cxy;
  out->covariance[4] = cyy;
  out->pose[2] = sums[9] + mdx;
  out->pose[3] = sums[10] + mdy;
  const double norm = std::sqrt(mc * mc + ms * ms);
  if (norm < std::numeric_limits<double>::epsilon()) {  // estimation.hpp:460-466
    out->covariance[8] = std::numeric_limits<double>::infinity();
    const Rot2 zero = rot_exp(0.0);
    out->pose[0] = zero.c;
    out->pose[1] = zero.s;
  } else {
    out->covariance[8] = -2.0 * std::log(norm);
    const Rot2 r = rot_from_complex(mc, ms);
    out->pose[0] = r.c;
    out->pose[1] = r.s;
  }
  return MCL_OK;
}

mcl_status mcl_estimate_pose(mcl_ctx* ctx, mcl_estimate* out) {
  if (!ctx || !out) return MCL_ERR_INVALID_ARGUMENT;
  if (ctx->n == 0) return fail(ctx, MCL_ERR_NOT_READY, "no particles");
  if (const mcl_status s = bind_device(ctx)) return s;
  double sums[12];
  if (const mcl_status s = do_estimate_sums(ctx, ctx->pivot, sums)) return s;
  const mcl_status s = mcl_estimate_from_sums(sums, out);
  if (s == MCL_OK && std::isfinite(out->pose[2]) && std::isfinite(out->pose[3])) {
    ctx->pivot[0] = out->pose[2];
    ctx->pivot[1] = out->pose[3];
  }
  return s;
}

mcl_status mcl_update(mcl_ctx* ctx, const double control_pose[4], const double* points_xy, uint64_t num_points,
                      mcl_estimate* estimate, mcl_update_info* info) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, control_pose && (num_points == 0 || points_xy), "null argument");
  const auto t_entry = std::chrono::steady_clock::now();
  if (info) {
    std::memset(info, 0, sizeof(*info));
    info->effective_sample_size = -1.0;
    info->num_particles = ctx->n;
  }
  if (ctx->pending_map)  // a map built ahead (mcl_set_map_async) takes over where its field is done
    if (const mcl_status s = apply_pending_map(ctx, false)) return s;
  if (ctx->n == 0) return MCL_OK;  // amcl_core.hpp:166-168 -> nullopt
  const Pose2 pose = pose_from(control_pose);
  // update_policy_ = on_motion (policies/on_motion.hpp:63-67,121-133); evaluated even when forced (:170)
  bool moved;
  const bool had_latest = ctx->have_latest;
  const Pose2 previous_latest = ctx->latest;
  if (!ctx->have_latest) {
    ctx->latest = pose;
    ctx->have_latest = true;
    moved = true;
  } else {
    const Pose2 delta = pose_mul(pose_inverse(ctx->latest), pose);
    moved = std::sqrt(delta.x * delta.x + delta.y * delta.y) > ctx->cfg.amcl.update_min_d ||
            std::abs(rot_log(delta.r)) > ctx->cfg.amcl.update_min_a;
    if (moved) ctx->latest = pose;
  }
  if (!moved && !ctx->force_update) return MCL_OK;
  // Everything that can fail without touching a particle is checked before the filter state moves: an update that
  // fails here leaves the motion unconsumed, as if it had not been called (the reference has no partial-update state).
  auto undo_policy = [&] {
    ctx->have_latest = had_latest;
    ctx->latest = previous_latest;
  };
  if (const mcl_status s = bind_device(ctx)) {
    undo_policy();
    return s;
  }
  if (const mcl_status s = reweight_preconditions(ctx, num_points)) {
    undo_policy();
    return s;
  }
  ctx->lf_mode.decided = false;  // whatever an earlier, failed cycle left behind
  if (ctx->have_comm && ctx->comm_world > 1) {
    if (const mcl_status s = sharded_preconditions(ctx)) {  // before any rank-local state moves: the ranks must not diverge
      undo_policy();
      return s;
    }
    return sharded_update(ctx, pose, points_xy, num_points, estimate, info);
  }
  if (const mcl_status s = stage_points(ctx, points_xy, num_points)) {
    undo_policy();
    return s;
  }

  // control_action_window_ << control (RollingWindow<SE2,2>: newest first, extrapolates when short)
  if (!ctx->have_window) {
    ctx->window0 = ctx->window1 = pose;
    ctx->have_window = true;
  } else {
    ctx->window1 = ctx->window0;
    ctx->window0 = pose;
  }
  ctx->step += 1;

  bool keys_ready = false;
  if (const mcl_status s = do_propagate(ctx, ctx->window0, ctx->window1, ctx->step, num_points, &keys_ready)) return s;  // :174-175
  const auto t_first = std::chrono::steady_clock::now();
  // With a fixed particle count and no selective resampling nothing in the cycle depends on a host-side decision: the
  // recovery estimator runs on the device as well and the cycle synchronises once, at the estimate.
  const mcl_amcl_params& ap = ctx->cfg.amcl;
  const bool device_policy = !ap.selective_resampling && ap.min_particles >= std::min<uint64_t>(ap.max_particles, ctx->capacity) &&
                             ctx->tuning.device_policy != 0;
  // (the normalisation follows at once: the LF kernel leaves the sums it is built on)
  if (const mcl_status s = do_reweight(ctx, points_xy, num_points, true, keys_ready, /*want_weight_sums=*/ctx->tuning.lf_weight_sums != 0)) return s;  // :176
  // Small sets (the reference's own sizes): everything behind the reweight in ONE launch of one workgroup and one synchronisation
  // (k_small_tail) - the policies are evaluated on the device, the host keeps the recovery filters' state.
  if (ctx->tuning.small_fused != 0 && ctx->n <= 4096 && std::min<uint64_t>(ap.max_particles, ctx->capacity) <= 4096) {
    ctx->every_n_current = (ctx->every_n_current + 1) % ap.resample_interval;  // :181
    SmallTail t{};
    t.src = ctx->cur();
    t.dst = ctx->other();
    t.n = static_cast<uint32_t>(ctx->n);
    t.max_particles = static_cast<uint32_t>(std::min<uint64_t>(ap.max_particles, ctx->capacity));
    t.min_particles = static_cast<uint32_t>(std::min<uint64_t>(ap.min_particles, t.max_particles));
    t.seed = ctx->cfg.seed;
    t.step = ctx->step;
    t.fires = ctx->every_n_current == 0;
    t.selective = ap.selective_resampling != 0;
    t.alpha_slow = ap.alpha_slow;
    t.alpha_fast = ap.alpha_fast;
    t.slow = ctx->slow.output;
    t.fast = ctx->fast.output;
    t.kld_epsilon = ap.kld_epsilon;
    t.kld_z = ap.kld_z;
    t.hp = HashParams{ap.spatial_resolution_x, ap.spatial_resolution_y, ap.spatial_resolution_theta};
    t.g = ctx->grid_view();
    t.fc = FreeCells{ctx->d_free.ptr, ctx->have_map ? ctx->n_free : 0};
    t.pivot_x = ctx->pivot[0];
    t.pivot_y = ctx->pivot[1];
    t.mirror = ctx->hd_scalars;
    t.d_scalars = ctx->d_scalars.ptr;
    // (the completion word only where asked for: measured 10 us per cycle SLOWER than the stream's signal at 2000 particles, round 6)
    ctx->done_armed = ctx->tuning.cycle_spin > 0 && !ctx->profile;
    if (ctx->done_armed) {
      t.done_flag = reinterpret_cast<unsigned long long*>(ctx->hd_scalars + 31);
      t.done_seq = ++ctx->done_seq;
    }
    stage_begin(ctx, MCL_STAGE_RESAMPLE);
    const bool launched = launch_small_tail(ctx->stream, t);
    stage_end(ctx, MCL_STAGE_RESAMPLE);
    if (launched) {
      MCL_HIP(ctx, hipGetLastError());
      ctx->lf_wsum_count = 0;
      ctx->weights_unit = false;
      if (const mcl_status s = wait_for_cycle(ctx)) return s;
      stage_collect(ctx);
      const double* h = ctx->h_scalars;
      const bool resampled = h[5] != 0.0;
      if (resampled) {
        ctx->live ^= 1;
        ctx->n = static_cast<uint64_t>(h[6]);
        ctx->weights_unit = true;  // particle_traits.hpp:105
      }
      ctx->slow.output = h[18];
      ctx->fast.output = h[19];
      ctx->force_update = false;  // :199
      // (what the info reports, before another kernel's mirrored values take their place)
      const double weight_sum = h[0], ess_seen = h[7], p_seen = h[22];
      mcl_estimate est{};
      if (ctx->estimate_kind == 1) {  // beluga_ros::Amcl returns cluster_based_estimate (beluga_ros/src/amcl.cpp:125): its own kernels
        if (const mcl_status s = mcl_cluster_based_estimate(ctx, &ctx->cluster_params, &est)) return s;
      } else {
        double sums[12];
        for (int k = 0; k < 9; ++k) sums[k] = h[8 + k];
        sums[9] = ctx->pivot[0];
        sums[10] = ctx->pivot[1];
        sums[11] = 0.0;
        if (const mcl_status s = mcl_estimate_from_sums(sums, &est)) return s;  // :200
      }
      if (std::isfinite(est.pose[2]) && std::isfinite(est.pose[3])) {
        ctx->pivot[0] = est.pose[2];
        ctx->pivot[1] = est.pose[3];
      }
      remember_cloud_estimate(ctx, est);
      if (estimate) *estimate = est;
      if (info) {
        info->updated = 1;
        info->resampled = resampled ? 1 : 0;
        info->num_particles = ctx->n;
        info->weight_sum = weight_sum;
        info->effective_sample_size = ess_seen;
        info->random_state_probability = p_seen;
      }
      return MCL_OK;
    }
    ctx->done_armed = false;
    ctx->every_n_current = (ctx->every_n_current + ap.resample_interval - 1) % ap.resample_interval;  // (not launched: the large path counts)
  }
  mcl_weight_stats stats{};
  double random_state_probability = 0.0;
  double ess = -1.0;
  bool do_resampling = false;
  bool estimate_enqueued = false;  // the estimate sums of the resampled set came out of the draw kernel
  constexpr int kPolicySlot = 20;  // d_scalars[20..23) = {slow, fast, p}
  if (device_policy) {
    // :177; the totals of the normalised weights and the recovery estimator (:179, :184-186) ride on the next kernel
    ctx->every_n_current = (ctx->every_n_current + 1) % ap.resample_interval;  // :181
    do_resampling = ctx->every_n_current == 0;
    const RecoveryPolicy policy{ap.alpha_slow, ap.alpha_fast, do_resampling ? 1 : 0, ctx->d_scalars.ptr + kPolicySlot,
                                ctx->hd_scalars + kPolicySlot};
    bool fused = false;  // :177 and the CDF of :188 in one launch (the normalised weights of a set that is resampled at once are not stored)
    if (do_resampling)
      if (const mcl_status s = do_normalize_cdf(ctx, policy, &fused)) return s;
    if (!fused)
      if (const mcl_status s = do_normalize(ctx, std::numeric_limits<double>::quiet_NaN(), nullptr, false, false,
                                            /*store_weights=*/!do_resampling || ctx->tuning.norm_store != 0)) return s;
    if (do_resampling) {
      if (const mcl_status s = do_resample(ctx, 0.0, ctx->step, nullptr, ctx->d_scalars.ptr + kPolicySlot + 2, true,
                                           ctx->estimate_kind == 0, &estimate_enqueued, &policy, true, fused)) return s;  // :188-196
    } else {
      launch_norm_finalize(ctx->stream, ctx->chunk_row(1), ctx->chunk_row(2), ctx->n, ctx->d_scalars.ptr + 1, ctx->hd_scalars + 1, &policy);
      MCL_HIP(ctx, hipGetLastError());
    }
  } else {
    if (const mcl_status s = do_normalize(ctx, std::numeric_limits<double>::quiet_NaN(), &stats)) return s;  // :177

    // :179 ThrunRecoveryProbabilityEstimator on the NORMALISED weights (thrun_..._estimator.hpp:69-89)
    {
      const double average = stats.norm_sum / static_cast<double>(ctx->n);
      const double fast_average = ctx->fast(average);
      const double slow_average = ctx->slow(average);
      if (std::abs(slow_average) >= std::numeric_limits<double>::epsilon())
        random_state_probability = std::clamp(1.0 - fast_average / slow_average, 0.0, 1.0);
    }
    // :181 every_n [&& on_effective_size_drop] (every_n.hpp:47-50, on_effective_size_drop.hpp:45-49)
    ctx->every_n_current = (ctx->every_n_current + 1) % ctx->cfg.amcl.resample_interval;
    do_resampling = ctx->every_n_current == 0;
    if (do_resampling && ctx->cfg.amcl.selective_resampling) {
      ess = stats.norm_sum == 0.0 ? 0.0 : (stats.norm_sum * stats.norm_sum) / stats.norm_sumsq;  // effective_sample_size.hpp:46-59
      do_resampling = ess < static_cast<double>(ctx->n) * 0.5;
    }
    if (do_resampling) {
      if (random_state_probability > 0.0) {  // :184-186
        ctx->slow.reset();
        ctx->fast.reset();
      }
      if (const mcl_status s = do_resample(ctx, random_state_probability, ctx->step, nullptr, nullptr, true, ctx->estimate_kind == 0,
                                           &estimate_enqueued)) return s;  // :188-196
    }
  }
  ctx->force_update = false;  // :199
  mcl_estimate est{};
  auto t_enqueued = t_first, t_waited = t_first;
  bool host_timed = false;
  if (ctx->estimate_kind == 1) {  // beluga_ros::Amcl returns cluster_based_estimate (beluga_ros/src/amcl.cpp:125)
    if (const mcl_status s = mcl_cluster_based_estimate(ctx, &ctx->cluster_params, &est)) return s;
    if (std::isfinite(est.pose[2]) && std::isfinite(est.pose[3])) {
      ctx->pivot[0] = est.pose[2];
      ctx->pivot[1] = est.pose[3];
    }
  } else if (estimate_enqueued) {  // :200, sums already produced by the draw kernel
    stage_begin(ctx, MCL_STAGE_ESTIMATE);
    stage_end(ctx, MCL_STAGE_ESTIMATE);
    t_enqueued = std::chrono::steady_clock::now();
    if (const mcl_status s = wait_for_cycle(ctx)) return s;
    t_waited = std::chrono::steady_clock::now();
    host_timed = true;
    stage_collect(ctx);
    double sums[12];
    for (int k = 0; k < 9; ++k) sums[k] = ctx->h_scalars[8 + k];
    sums[9] = ctx->pivot[0];
    sums[10] = ctx->pivot[1];
    sums[11] = 0.0;
    if (const mcl_status s = mcl_estimate_from_sums(sums, &est)) return s;
    if (std::isfinite(est.pose[2]) && std::isfinite(est.pose[3])) {
      ctx->pivot[0] = est.pose[2];
      ctx->pivot[1] = est.pose[3];
    }
  } else if (const mcl_status s = mcl_estimate_pose(ctx, &est)) {  // :200
    return s;
  }
  remember_cloud_estimate(ctx, est);  // where the ordering keys of the next cycle are centred
  if (device_policy) {  // read back together with the estimate
    stats.sum = ctx->h_scalars[0];
    stats.norm_sum = ctx->h_scalars[1];
    stats.norm_sumsq = ctx->h_scalars[2];
    random_state_probability = ctx->h_scalars[kPolicySlot + 2];
  }
  if (estimate) *estimate = est;
  if (info) {
    info->updated = 1;
    info->resampled = do_resampling ? 1 : 0;
    info->num_particles = ctx->n;
    info->weight_sum = stats.sum;
    info->effective_sample_size = ess;
    info->random_state_probability = random_state_probability;
  }
  if (host_timed) {
    const auto ns = [](auto a, auto b) { return static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count()); };
    ctx->host_ns[0] += ns(t_entry, t_first);
    ctx->host_ns[1] += ns(t_first, t_enqueued);
    ctx->host_ns[2] += ns(t_enqueued, t_waited);
    ctx->host_ns[3] += ns(t_waited, std::chrono::steady_clock::now());
    ctx->host_cycles += 1;
  }
  return MCL_OK;
}

mcl_status mcl_prepare_laser_scan(const mcl_laser_scan* scan, double* points_xy, uint64_t* num_points) {
  if (!scan || !num_points || (scan->num_ranges && (!scan->ranges || !points_xy))) return MCL_ERR_INVALID_ARGUMENT;
  const uint64_t n = scan->num_ranges, count = scan->max_beams;
  const double lo = std::max(static_cast<double>(scan->range_min), scan->min_range);  // laser_scan.hpp:61-62
  const double hi = std::min(static_cast<double>(scan->range_max), scan->max_range);
  const double qx = scan->origin_se3[0], qy = scan->origin_se3[1], qz = scan->origin_se3[2], qw = scan->origin_se3[3];
  const uint64_t taken = n == 0 ? 0 : std::min(n, count);  // take_evenly.hpp:47-57
  uint64_t m = 0;
  for (uint64_t k = 0; k < taken; ++k) {
    uint64_t i = k;  // take_evenly.hpp:126-148: ceil(k * (size - 1) / (count - 1))
    if (count <= n && k > 0) {
      if (count == 1) break;
      const int64_t a = static_cast<int64_t>(k) * (static_cast<int64_t>(n) - 1), b = static_cast<int64_t>(count) - 1;
      i = static_cast<uint64_t>(a / b + ((a % b == 0) ? 0 : 1));
    }
    if (i >= n) break;
    const double range = static_cast<double>(scan->ranges[i]);
    // float arithmetic first, then widened (laser_scan.hpp:73-77)
    const double theta = static_cast<double>(scan->angle_min + static_cast<float>(static_cast<int>(i)) * scan->angle_increment);
    if (std::isnan(range) || !(range >= lo) || !(range <= hi)) continue;  // sensor/data/laser_scan.hpp:79-83
    const double px = range * std::cos(theta), py = range * std::sin(theta), pz = 0.0;
    // origin * (x, y, 0): Sophus SO3 rotates with uv = 2 (q.vec x p); p + q.w uv + q.vec x uv, then adds the translation
    double ux = qy * pz - qz * py, uy = qz * px - qx * pz, uz = qx * py - qy * px;
    ux += ux;
    uy += uy;
    uz += uz;
    points_xy[2 * m] = (px + qw * ux + (qy * uz - qz * uy)) + scan->origin_se3[4];
    points_xy[2 * m + 1] = (py + qw * uy + (qz * ux - qx * uz)) + scan->origin_se3[5];
    ++m;
  }
  *num_points = m;
  return MCL_OK;
}

mcl_status mcl_update_laser_scan(mcl_ctx* ctx, const double control_pose[4], const mcl_laser_scan* scan, mcl_estimate* estimate,
                                 mcl_update_info* info) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, scan != nullptr, "null scan");
  std::vector<double> pts(2 * std::min<uint64_t>(scan->num_ranges, scan->max_beams) + 2);
  uint64_t m = 0;
  if (mcl_prepare_laser_scan(scan, pts.data(), &m) != MCL_OK) return fail(ctx, MCL_ERR_INVALID_ARGUMENT, "bad laser scan");
  return mcl_update(ctx, control_pose, pts.data(), m, estimate, info);
}

mcl_status mcl_cluster_based_estimate(mcl_ctx* ctx, const mcl_cluster_params* params, mcl_estimate* out) {
  if (!ctx || !out) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  stage_begin(ctx, MCL_STAGE_ESTIMATE);
  const mcl_status s = do_cluster_estimate(ctx, params ? *params : mcl_cluster_params{0.20, 0.524, 0.90}, out);
  stage_end(ctx, MCL_STAGE_ESTIMATE);
  if (s == MCL_OK) {
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    stage_collect(ctx);
  }
  return s;
}

mcl_status mcl_set_estimate_kind(mcl_ctx* ctx, int32_t kind, const mcl_cluster_params* params) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, kind == 0 || kind == 1, "estimate kind must be 0 (estimate) or 1 (cluster_based_estimate)");
  const int32_t kind_before = ctx->estimate_kind;
  const mcl_cluster_params params_before = ctx->cluster_params;
  ctx->estimate_kind = kind;
  if (params) ctx->cluster_params = *params;
  // On a sharded filter the estimate's kind selects the cycle's collectives: a COLLECTIVE call there (every rank, concurrently, the
  // same kind and parameters - entered whatever this rank's state was before); a mismatch leaves kind and parameters as they were.
  if (const mcl_status s = comm_agree(ctx, "mcl_set_estimate_kind")) {
    ctx->estimate_kind = kind_before;
    ctx->cluster_params = params_before;
    return s;
  }
  return MCL_OK;
}

mcl_status mcl_sample_particle_cloud(mcl_ctx* ctx, uint64_t size, uint32_t draw_id, double* states) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, size == 0 || states, "null output");
  if (size == 0 || ctx->n == 0) return MCL_OK;  // particle_cloud.hpp:141: an empty set yields an empty message
  if (const mcl_status s = bind_device(ctx)) return s;
  MCL_HIP(ctx, ctx->d_cloud.ensure(size));
  MCL_HIP(ctx, ctx->d_cloud_w.ensure(size));
  if (const mcl_status s = do_build_cdf(ctx)) return s;
  ResampleArgs ra{};
  ra.seed = ctx->cfg.seed;
  ra.step = 0x80000000u | draw_id;
  ra.random_state_probability = 0.0;
  ra.n_in = ctx->n;
  ra.first_candidate = 0;
  ra.count = size;
  ra.out_offset = 0;
  launch_resample_draw(ctx->stream, ctx->cur(), ctx->cdf_tree(), ctx->d_scalars.ptr + 4, Particles{ctx->d_cloud.ptr, ctx->d_cloud_w.ptr}, ra,
                       ctx->grid_view(), FreeCells{nullptr, 0}, HashParams{1.0, 1.0, 1.0}, nullptr);
  MCL_HIP(ctx, hipGetLastError());
  MCL_HIP(ctx, hipMemcpyAsync(states, ctx->d_cloud.ptr, size * sizeof(double4), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MCL_OK;
}

mcl_status mcl_get_device_view(mcl_ctx* ctx, mcl_device_view* view) {
  if (!ctx || !view) return MCL_ERR_INVALID_ARGUMENT;
  const Particles p = ctx->cur();
  ctx->weights_unit = false;  // (the caller gets writable pointers)
  view->states = reinterpret_cast<double*>(p.pose);
  view->w = p.w;
  view->cdf = ctx->d_cdf.ptr;
  view->n = ctx->n;
  view->capacity = ctx->capacity;
  view->hip_stream = ctx->stream;
  return MCL_OK;
}

mcl_status mcl_set_num_particles(mcl_ctx* ctx, uint64_t n) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->lf_wsum_count = 0;  // (workgroup sums of an earlier reweight describe another set)
  ctx->order_valid = false;  // (and an order computed ahead describes the particles of another one)
  MCL_REQUIRE(ctx, n <= ctx->capacity, "n exceeds capacity");
  if (n > ctx->n) ctx->weights_unit = false;  // (what lies beyond the set is whatever was there)
  ctx->n = n;
  return MCL_OK;
}

mcl_status mcl_build_cdf(mcl_ctx* ctx, double* total) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  if (ctx->n == 0) {
    if (total) *total = 0.0;
    return MCL_OK;
  }
  if (const mcl_status s = do_build_cdf(ctx)) return s;
  if (total) {
    MCL_HIP(ctx, hipMemcpyAsync(ctx->h_scalars + 4, ctx->d_scalars.ptr + 4, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total = ctx->h_scalars[4];
  }
  return MCL_OK;
}

mcl_status mcl_resample_targets(mcl_ctx* ctx, uint32_t step, double random_state_probability, double total,
                                uint64_t first_slot, uint64_t count, double* d_targets) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, count == 0 || d_targets, "null targets");
  if (const mcl_status s = bind_device(ctx)) return s;
  launch_resample_targets(ctx->stream, ctx->cfg.seed, step, random_state_probability, total, first_slot, count,
                          ctx->have_map ? ctx->n_free : 0, d_targets);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_route_targets(mcl_ctx* ctx, const double* d_targets, uint64_t count, const double* d_ends, const double* d_offsets,
                             uint32_t world, uint32_t self_rank, double* d_send_targets, uint32_t* d_order, int64_t* d_counts) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, world >= 1 && world <= 64 && self_rank < world, "world must be 1..64");
  MCL_REQUIRE(ctx, d_ends && d_offsets && d_counts && (count == 0 || (d_targets && d_send_targets && d_order)), "null argument");
  MCL_REQUIRE(ctx, count < (1ull << 32), "too many targets");
  if (const mcl_status s = bind_device(ctx)) return s;
  const size_t nblocks = num_chunks(count);
  const size_t hist = static_cast<size_t>(world) * nblocks;
  MCL_HIP(ctx, ctx->d_route_u32.ensure(hist + 2 * (hist / kChunk + 1) + (count + 3) / 4 + 4));
  uint32_t* block_hist = ctx->d_route_u32.ptr;
  uint32_t* chunk_sum = block_hist + hist;
  uint32_t* chunk_off = chunk_sum + (hist / kChunk + 1);
  uint8_t* dest = reinterpret_cast<uint8_t*>(chunk_off + (hist / kChunk + 1));
  launch_route_targets(ctx->stream, d_targets, count, d_ends, d_offsets, world, self_rank, dest, block_hist, chunk_sum, chunk_off,
                       d_send_targets, d_order, reinterpret_cast<long long*>(d_counts));
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_serve_requests(mcl_ctx* ctx, const double* d_requests, uint64_t m, double* d_replies) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, m == 0 || (d_requests && d_replies), "null argument");
  MCL_REQUIRE(ctx, m == 0 || ctx->n > 0, "empty shard cannot serve draws");
  if (const mcl_status s = bind_device(ctx)) return s;
  launch_gather_by_cdf_aos(ctx->stream, ctx->cur(), ctx->cdf_tree(), d_requests, m, d_replies);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_commit_routed(mcl_ctx* ctx, uint32_t step, uint64_t first_slot, uint64_t count, const double* d_replies,
                             const uint32_t* d_order, const double* d_targets) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, count <= ctx->capacity, "count exceeds shard capacity");
  MCL_REQUIRE(ctx, count == 0 || (d_replies && d_order && d_targets), "null argument");
  if (const mcl_status s = bind_device(ctx)) return s;
  stage_begin(ctx, MCL_STAGE_RESAMPLE);
  launch_commit_routed(ctx->stream, ctx->other(), ctx->cfg.seed, step, first_slot, count, d_replies, d_order, d_targets,
                       ctx->grid_view(), FreeCells{ctx->d_free.ptr, ctx->have_map ? ctx->n_free : 0});
  stage_end(ctx, MCL_STAGE_RESAMPLE);
  MCL_HIP(ctx, hipGetLastError());
  ctx->live ^= 1;
  ctx->n = count;
  ctx->weights_unit = true;
  return MCL_OK;
}

mcl_status mcl_finish_candidates(mcl_ctx* ctx, uint32_t step, uint64_t first_slot, uint64_t count, const double* d_replies,
                                 const uint32_t* d_order, const double* d_targets, double* d_states, uint64_t* d_hashes) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, count == 0 || (d_replies && d_order && d_targets && d_states && d_hashes), "null argument");
  if (const mcl_status s = bind_device(ctx)) return s;
  const mcl_amcl_params& a = ctx->cfg.amcl;
  stage_begin(ctx, MCL_STAGE_RESAMPLE);
  launch_finish_candidates(ctx->stream, ctx->cfg.seed, step, first_slot, count, d_replies, d_order, d_targets, ctx->grid_view(),
                           FreeCells{ctx->d_free.ptr, ctx->have_map ? ctx->n_free : 0},
                           HashParams{a.spatial_resolution_x, a.spatial_resolution_y, a.spatial_resolution_theta}, d_states,
                           reinterpret_cast<unsigned long long*>(d_hashes));
  stage_end(ctx, MCL_STAGE_RESAMPLE);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_kld_begin(mcl_ctx* ctx) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  MCL_REQUIRE(ctx, ctx->cfg.amcl.max_particles < 0xFFFFFFFFull, "max_particles too large for KLD resampling");
  return kld_begin(ctx);
}

mcl_status mcl_kld_feed(mcl_ctx* ctx, const uint64_t* d_hashes, uint64_t count, uint64_t* first_fail) {
  if (!ctx || !first_fail) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  MCL_REQUIRE(ctx, ctx->kld_capacity > 0, "mcl_kld_feed before mcl_kld_begin");
  MCL_REQUIRE(ctx, count == 0 || d_hashes, "null hashes");
  MCL_REQUIRE(ctx, ctx->kld_pos + count <= ctx->kld_capacity, "more candidates than max_particles");
  *first_fail = ~0ull;
  if (count == 0) return MCL_OK;
  MCL_HIP(ctx, hipMemcpyAsync(ctx->d_hashes.ptr + ctx->kld_pos, d_hashes, count * sizeof(unsigned long long), hipMemcpyDeviceToDevice,
                              ctx->stream));
  if (const mcl_status s = kld_grow_table(ctx, ctx->kld_pos + count)) return s;
  return kld_process(ctx, count, first_fail);
}

mcl_status mcl_load_shard(mcl_ctx* ctx, const double* d_states, uint64_t n, uint64_t shard_offset) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->lf_wsum_count = 0;
  ctx->order_valid = false;
  MCL_REQUIRE(ctx, n <= ctx->capacity, "n exceeds shard capacity");
  MCL_REQUIRE(ctx, n == 0 || d_states, "null states");
  if (const mcl_status s = bind_device(ctx)) return s;
  if (n) MCL_HIP(ctx, hipMemcpyAsync(ctx->cur().pose, d_states, n * sizeof(double4), hipMemcpyDeviceToDevice, ctx->stream));
  ctx->weights_unit = false;
  launch_fill(ctx->stream, ctx->cur().w, n, 1.0);  // particle_traits.hpp:105
  MCL_HIP(ctx, hipGetLastError());
  ctx->weights_unit = true;
  ctx->n = n;
  ctx->cfg.shard_offset = shard_offset;
  ctx->have_cloud_estimate = false;
  return MCL_OK;
}

mcl_status mcl_weight_sum_device(mcl_ctx* ctx, double* d_sum) {
  if (!ctx || !d_sum) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  launch_weight_sum(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(0), d_sum);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_normalize_device(mcl_ctx* ctx, const double* d_factor, double* d_stats) {
  if (!ctx || !d_factor || !d_stats) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  ctx->weights_unit = false;
  stage_begin(ctx, MCL_STAGE_NORMALIZE);
  launch_normalize(ctx->stream, ctx->cur().w, ctx->n, d_factor, ctx->chunk_row(1), ctx->chunk_row(2), d_stats);
  stage_end(ctx, MCL_STAGE_NORMALIZE);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_build_cdf_device(mcl_ctx* ctx, double* d_total) {
  if (!ctx || !d_total) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  if (ctx->n == 0) {
    MCL_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(double), ctx->stream));
    return MCL_OK;
  }
  launch_cdf(ctx->stream, ctx->cur().w, ctx->n, ctx->chunk_row(3), ctx->chunk_row(4), ctx->d_cdf.ptr, d_total, ctx->d_cdf_tree.ptr);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_estimate_sums_device(mcl_ctx* ctx, const double pivot_xy[2], double* d_sums) {
  if (!ctx || !pivot_xy || !d_sums) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  stage_begin(ctx, MCL_STAGE_ESTIMATE);
  launch_estimate_sums(ctx->stream, ctx->cur(), ctx->n, pivot_xy[0], pivot_xy[1], ctx->chunk_row(0), d_sums);
  stage_end(ctx, MCL_STAGE_ESTIMATE);
  MCL_HIP(ctx, hipGetLastError());
  return MCL_OK;
}

mcl_status mcl_initialize_from_map(mcl_ctx* ctx) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->lf_wsum_count = 0;  // (workgroup sums of an earlier reweight describe another set)
  ctx->order_valid = false;  // (and an order computed ahead describes the particles of another one)
  if (!ctx->have_map) return fail(ctx, MCL_ERR_NOT_READY, "mcl_initialize_from_map: no map set");
  MCL_REQUIRE(ctx, ctx->n_free > 0, "mcl_initialize_from_map: the map has no free cell");  // the reference asserts (:136)
  if (const mcl_status s = bind_device(ctx)) return s;
  const uint64_t n = std::min<uint64_t>(ctx->cfg.amcl.max_particles, ctx->capacity);  // take_exactly(max_particles)
  ctx->weights_unit = false;
  launch_init_from_map(ctx->stream, ctx->cur(), n, ctx->cfg.seed, ctx->cfg.shard_offset, ctx->grid_view(),
                       FreeCells{ctx->d_free.ptr, ctx->n_free});
  MCL_HIP(ctx, hipGetLastError());
  ctx->weights_unit = true;  // (k_init_from_map writes 1.0 to every weight)
  ctx->n = n;
  ctx->global_n = 0;
  ctx->global_n_unknown = false;
  ctx->force_update = true;  // beluga_ros/include/beluga_ros/amcl.hpp:197
  // the set covers the map: centre of the grid, the spread of a uniform distribution over its extent in the world's axes
  // (extent / sqrt(12)), every heading - what the estimate of such a set would say
  const double hx = 0.5 * ctx->W * ctx->resolution, hy = 0.5 * ctx->H * ctx->resolution;
  double cx, cy;
  rot_apply(ctx->origin.r, hx, hy, cx, cy);
  const double ex = 2.0 * (std::abs(ctx->origin.r.c) * hx + std::abs(ctx->origin.r.s) * hy);
  const double ey = 2.0 * (std::abs(ctx->origin.r.s) * hx + std::abs(ctx->origin.r.c) * hy);
  ctx->cloud_mean[0] = cx + ctx->origin.x;
  ctx->cloud_mean[1] = cy + ctx->origin.y;
  ctx->cloud_mean[2] = 0.0;
  ctx->cloud_sigma[0] = ex / std::sqrt(12.0);
  ctx->cloud_sigma[1] = ey / std::sqrt(12.0);
  ctx->cloud_sigma[2] = kPi;
  ctx->have_cloud_estimate = true;
  // A set spread over the whole map is what the patch kernel would report as dispersed after its first launch: say so now
  // (reports of earlier launches are history), so that the first cycle already takes the kernel for dispersed sets.
  patch_totals(ctx, &ctx->patch_seen_planned, &ctx->patch_seen_through);
  ctx->patch_useful = false;
  ctx->patch_probe_in = 16;
  return MCL_OK;
}

mcl_status mcl_has_likelihood_field(const mcl_ctx* ctx, int32_t* has) {
  if (!ctx || !has) return MCL_ERR_INVALID_ARGUMENT;
  *has = ctx->cfg.sensor_kind != MCL_SENSOR_BEAM ? 1 : 0;  // beam_model.hpp has no likelihood_field() (has_likelihood_field_v)
  return MCL_OK;
}

mcl_status mcl_get_likelihood_field_origin(mcl_ctx* ctx, double origin[4]) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, origin, "null output");
  if (ctx->cfg.sensor_kind == MCL_SENSOR_BEAM)
    return fail(ctx, MCL_ERR_UNSUPPORTED, "The current sensor model does not support likelihood field");
  if (!ctx->have_map) return fail(ctx, MCL_ERR_NOT_READY, "no likelihood field");
  // likelihood_field_model_base.hpp:105: world_to_likelihood_field_transform_.inverse(), i.e. inverse(inverse(grid.origin()))
  const Pose2 o = pose_inverse(ctx->origin_inverse);
  origin[0] = o.r.c;
  origin[1] = o.r.s;
  origin[2] = o.x;
  origin[3] = o.y;
  return MCL_OK;
}

mcl_status mcl_project_point_cloud(const float* points_xyz, uint64_t num_points, const double origin_se3[7], double* points_xy) {
  if (!origin_se3 || (num_points && (!points_xyz || !points_xy))) return MCL_ERR_INVALID_ARGUMENT;
  const double qx = origin_se3[0], qy = origin_se3[1], qz = origin_se3[2], qw = origin_se3[3];
  for (uint64_t i = 0; i < num_points; ++i) {
    // beluga_ros/src/amcl.cpp:73-76: origin * p.cast<double>(), keep x and y.  Sophus SO3 rotates with
    // uv = 2 (q.vec x p); p + q.w uv + q.vec x uv, then adds the translation.
    const double px = static_cast<double>(points_xyz[3 * i]), py = static_cast<double>(points_xyz[3 * i + 1]),
                 pz = static_cast<double>(points_xyz[3 * i + 2]);
    double ux = qy * pz - qz * py, uy = qz * px - qx * pz, uz = qx * py - qy * px;
    ux += ux;
    uy += uy;
    uz += uz;
    points_xy[2 * i] = (px + qw * ux + (qy * uz - qz * uy)) + origin_se3[4];
    points_xy[2 * i + 1] = (py + qw * uy + (qz * ux - qx * uz)) + origin_se3[5];
  }
  return MCL_OK;
}

mcl_status mcl_update_point_cloud(mcl_ctx* ctx, const double control_pose[4], const float* points_xyz, uint64_t num_points,
                                  const double origin_se3[7], mcl_estimate* estimate, mcl_update_info* info) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, origin_se3 && (num_points == 0 || points_xyz), "null argument");
  std::vector<double> pts(2 * num_points + 2);
  if (mcl_project_point_cloud(points_xyz, num_points, origin_se3, pts.data()) != MCL_OK) return fail(ctx, MCL_ERR_INVALID_ARGUMENT, "bad point cloud");
  return mcl_update(ctx, control_pose, pts.data(), num_points, estimate, info);
}

mcl_status mcl_set_option(mcl_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return MCL_ERR_INVALID_ARGUMENT;
  const std::string key(name);
  Tuning& t = ctx->tuning;
  if (key == "lf_variant") t.lf_variant = value == 0 ? kLfWavePerParticle : (value == 1 ? kLfLanePerParticle : (value == 3 ? kLfBeamLanes : kLfSortedLanes));
  else if (key == "lf_dispersed") t.lf_dispersed = value < 0 ? 0 : static_cast<int>(std::min<int64_t>(value, 2));
  else if (key == "lf_far_beams_per_wave") t.lf_far_beams_per_wave = value < 0 ? 0 : static_cast<int>(std::min<int64_t>(value, 4096));
  else if (key == "key_layout") t.key_layout = value < 0 ? -1 : (value ? 1 : 0);
  else if (key == "lf_far_tiles") t.lf_far_tiles = value < 0 ? 0 : static_cast<int>(std::min<int64_t>(value, 2));
  else if (key == "lf_loose_below") t.lf_loose_below = static_cast<int>(std::clamp<int64_t>(value, 0, 257));
  else if (key == "lf_small_particles") t.lf_small_particles = value < 0 ? 0 : static_cast<int>(std::min<int64_t>(value, INT32_MAX));
  else if (key == "lf_fast") t.lf_fast = value < 0 ? -1 : (value ? 1 : 0);
  else if (key == "lf_table") t.lf_table = value ? 1 : 0;
  else if (key == "lf_patch") t.lf_patch = value < 0 || value > 2 ? 1 : static_cast<int>(value);
  else if (key == "device_policy") {
    const int before = t.device_policy;
    t.device_policy = value ? 1 : 0;
    // On a sharded filter it selects the cycle's collectives: a COLLECTIVE call there - every rank makes it, concurrently, with the
    // same value, whatever its value was before (a rank that skipped the exchange because nothing changed for IT would leave the
    // others waiting).  A mismatch leaves the option as it was.
    if (const mcl_status s = comm_agree(ctx, "mcl_set_option(device_policy)")) {
      t.device_policy = before;
      return s;
    }
  }
  else if (key == "field_build") t.field_build = value ? 1 : 0;
  else if (key == "key_curve") t.key_curve = value ? 1 : 0;
  else if (key == "key_warp") t.key_warp = value ? 1 : 0;
  else if (key == "key_bits_xy") t.key_bits_xy = (value >= 4 && value <= 6) ? static_cast<int>(value) : 0;
  else if (key == "lf_margin") t.lf_margin = value ? 1 : 0;
  else if (key == "lf_queue") t.lf_queue = value ? 1 : 0;
  else if (key == "lf_ends_first") t.lf_ends_first = value ? 1 : 0;
  else if (key == "beam_free_ahead") t.beam_free_ahead = value ? 1 : 0;
  else if (key == "beam_sectors") t.beam_sectors = value ? 1 : 0;
  else if (key == "shard_pad_permille") {
    // The capacity every pair of ranks exchanges: on a sharded filter a COLLECTIVE call like device_policy (ranks with different
    // capacities would post all-to-alls of different sizes); a mismatch leaves the option as it was.
    const int before = t.shard_pad_permille;
    t.shard_pad_permille = static_cast<int>(std::clamp<int64_t>(value, 0, 8000));
    if (const mcl_status s = comm_agree(ctx, "mcl_set_option(shard_pad_permille)")) {
      t.shard_pad_permille = before;
      return s;
    }
  }
  else if (key == "lf_queue_grid") t.lf_queue_grid = static_cast<int>(std::clamp<int64_t>(value, 0, 1 << 20));
  else if (key == "cycle_spin") t.cycle_spin = value < 0 ? -1 : (value ? 1 : 0);
  else if (key == "beam_table") {
    t.beam_table = value ? 1 : 0;
    if (!t.beam_table && ctx->beam_table_ready) {  // its memory goes back at once
      if (bind_device(ctx) == MCL_OK && hipStreamSynchronize(ctx->stream) == hipSuccess) {
        ctx->d_beam_table.release();
        ctx->beam_table_ready = false;
      }
    }
  }
  else if (key == "lf_weight_sums") t.lf_weight_sums = value ? 1 : 0;
  else if (key == "scan_fused") t.scan_fused = static_cast<int>(std::clamp<int64_t>(value, 0, 2));
  else if (key == "draw_fold") t.draw_fold = static_cast<int>(std::clamp<int64_t>(value, 0, 2));
  else if (key == "lf_unit_weights") t.lf_unit_weights = value ? 1 : 0;
  else if (key == "small_fused") t.small_fused = value ? 1 : 0;
  else if (key == "norm_store") t.norm_store = value ? 1 : 0;
  else if (key == "order_ahead") t.order_ahead = value ? 1 : 0;
  else if (key == "noise_ahead") t.noise_ahead = static_cast<int>(std::clamp<int64_t>(value, 0, 2));
  else if (key == "lf_split") t.lf_split = static_cast<int>(value & 3);  // 1: side by side only, 2: stacked only, 3: both
  else if (key == "sort_min_particles") t.sort_min_particles = static_cast<int>(std::clamp<int64_t>(value, 0, 1ll << 30));
  else if (key == "beam_sort_min_particles") t.beam_sort_min_particles = static_cast<int>(std::clamp<int64_t>(value, 0, 1ll << 30));
  else return fail(ctx, MCL_ERR_INVALID_ARGUMENT, "mcl_set_option: unknown option " + key);
  return MCL_OK;
}

mcl_status mcl_get_counter(mcl_ctx* ctx, const char* name, uint64_t* value) {
  if (!ctx || !name || !value) return MCL_ERR_INVALID_ARGUMENT;
  const std::string key(name);
  if (key == "lf_fast_launches") *value = ctx->lf_fast_launches;
  else if (key == "lf_patch_launches") *value = ctx->lf_patch_launches;
  else if (key == "lf_queue_launches") *value = ctx->lf_queue_launches;
  else if (key == "lf_beams_launches") *value = ctx->lf_beams_launches;
  else if (key == "lf_far_launches") *value = ctx->lf_far_launches;
  else if (key == "lf_far_beams_launches") *value = ctx->lf_far_beams_launches;
  else if (key == "lf_far_tiles") *value = ctx->far_tiles;
  else if (key == "noise_ahead_used") *value = ctx->noise_ahead_used;
  else if (key == "order_ahead_used") *value = ctx->order_ahead_used;
  else if (key == "order_ahead_missed") *value = ctx->order_ahead_missed;
  else if (key == "lf_patch_groups_planned" || key == "lf_patch_groups_through") {
    if (const mcl_status s = bind_device(ctx)) return s;
    MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t planned, through;
    patch_totals(ctx, &planned, &through, /*synchronised=*/true);
    *value = key == "lf_patch_groups_planned" ? planned : through;
  }
  else if (key == "host_ns_to_first_launch") *value = ctx->host_ns[0];
  else if (key == "host_ns_other_launches") *value = ctx->host_ns[1];
  else if (key == "host_ns_wait") *value = ctx->host_ns[2];
  else if (key == "host_ns_after_wait") *value = ctx->host_ns[3];
  else if (key == "host_cycles") *value = ctx->host_cycles;
  else if (key == "field_build_us") *value = static_cast<uint64_t>(ctx->field_build_ms * 1e3);  // kernels of the last device field build
  else if (key == "field_built_on_device") *value = ctx->field_built_on_device ? 1 : 0;
  else if (key == "cluster_cells") *value = ctx->cluster_cells;
  else if (key == "comm_bytes_out") *value = ctx->comm_bytes_out;
  else if (key == "comm_collectives") *value = ctx->comm_collectives;
  else if (key == "comm_host_syncs") *value = ctx->comm_host_syncs;
  else if (key == "comm_overflows") *value = ctx->comm_overflows;
  else if (key == "comm_ranks_seen") *value = ctx->comm_ranks_seen;
  else if (key == "comm_backend") *value = static_cast<uint64_t>(ctx->comm_backend);
  else return fail(ctx, MCL_ERR_INVALID_ARGUMENT, "mcl_get_counter: unknown counter " + key);
  return MCL_OK;
}

mcl_status mcl_debug_order(mcl_ctx* ctx, uint32_t* perm, uint32_t* keys) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, perm && keys, "null output");
  MCL_REQUIRE(ctx, ctx->n > 0 && ctx->n < (1ull << 32), "no particles");
  if (const mcl_status s = bind_device(ctx)) return s;
  const SortScratch sort = ctx->sort_scratch();
  KeyFrame frame{};
  const bool have_frame = predict_key_frame(ctx, nullptr, &frame);
  launch_order_particles(ctx->stream, ctx->cur(), ctx->n, &sort, have_frame ? &frame : nullptr, false, frame.layout);
  MCL_HIP(ctx, hipGetLastError());
  MCL_HIP(ctx, hipMemcpyAsync(perm, sort.perm, ctx->n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipMemcpyAsync(keys, sort.keys, ctx->n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MCL_OK;
}

mcl_status mcl_debug_set_recovery_filters(mcl_ctx* ctx, double slow, double fast) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  ctx->slow.output = slow;
  ctx->fast.output = fast;
  const double both[2] = {slow, fast};
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  MCL_HIP(ctx, hipMemcpy(ctx->d_scalars.ptr + 20, both, sizeof(both), hipMemcpyHostToDevice));  // d_scalars[20..23) = {slow, fast, p}
  return MCL_OK;
}

mcl_status mcl_comm_attach(mcl_ctx* ctx, uint32_t rank, uint32_t world, const mcl_transport* transport) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, world >= 1 && world <= 64 && rank < world, "mcl_comm_attach: world must be 1..64, rank < world");
  MCL_REQUIRE(ctx, world == 1 || (transport && transport->all_gather && transport->all_to_all), "mcl_comm_attach: incomplete transport");
  MCL_REQUIRE(ctx, world == 1 || ctx->cfg.shard_capacity > 0, "mcl_comm_attach: create the context with its shard_offset / shard_capacity");
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  ctx->transport = transport ? *transport : mcl_transport{};
  ctx->have_comm = true;
  if (ctx->comm_backend != 2) {
    ctx->comm_backend = world > 1 ? 1 : 0;
    ctx->comm_ranks_seen = world;
  }
  // the first collective of the communicator: do the ranks run the same filter?  (a rank with another
  // BELUGA_MCL_DEVICE_POLICY would otherwise take another sequence of collectives and block its peers for ever)
  if (const mcl_status s = comm_agree(ctx, "mcl_comm_attach")) {
    ctx->have_comm = false;
    return s;
  }
  return MCL_OK;
}

mcl_status mcl_comm_unique_id(uint8_t id[128]) {
  if (!id) return MCL_ERR_INVALID_ARGUMENT;
  std::string error;
  RcclApi* api = rccl_api(&error);
  if (!api) return fail(nullptr, MCL_ERR_UNSUPPORTED, "mcl_comm_unique_id: " + error);
  RcclId uid;
  if (api->GetUniqueId(&uid) != 0) return fail(nullptr, MCL_ERR_HIP, "ncclGetUniqueId failed");
  std::memcpy(id, uid.internal, sizeof(uid.internal));
  return MCL_OK;
}

mcl_status mcl_comm_attach_rccl(mcl_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t world) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  MCL_REQUIRE(ctx, id && world >= 1 && world <= 64 && rank < world, "mcl_comm_attach_rccl: bad argument");
  std::string error;
  RcclApi* api = rccl_api(&error);
  if (!api) return fail(ctx, MCL_ERR_UNSUPPORTED, "mcl_comm_attach_rccl: " + error);
  if (const mcl_status s = bind_device(ctx)) return s;
  RcclId uid;
  std::memcpy(uid.internal, id, sizeof(uid.internal));
  void* comm = nullptr;
  if (api->CommInitRank(&comm, static_cast<int>(world), uid, static_cast<int>(rank)) != 0 || !comm)
    return fail(ctx, MCL_ERR_HIP, "ncclCommInitRank failed");
  ctx->rccl_comm = comm;
  ctx->rccl_user = mcl_ctx::RcclUserStorage{comm, rank, world};
  ctx->comm_backend = 2;
  ctx->comm_ranks_seen = world;
  if (api->CommCount) {  // what the communicator itself says (the driver's check that RCCL saw every rank)
    int count = 0;
    if (api->CommCount(comm, &count) == 0 && count > 0) ctx->comm_ranks_seen = static_cast<uint64_t>(count);
  }
  const mcl_transport t{&ctx->rccl_user, rccl_all_gather, rccl_all_to_all};
  return mcl_comm_attach(ctx, rank, world, &t);
}

mcl_status mcl_sync(mcl_ctx* ctx) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  stage_collect(ctx);
  return MCL_OK;
}

mcl_status mcl_beam_cells_visited(mcl_ctx* ctx, uint64_t* cells, int32_t reset) {
  if (!ctx || !cells) return MCL_ERR_INVALID_ARGUMENT;
  if (const mcl_status s = bind_device(ctx)) return s;
  MCL_HIP(ctx, hipMemcpyAsync(ctx->h_kld_scalars + 1, ctx->d_kld_scalars.ptr + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                              ctx->stream));
  if (reset) MCL_HIP(ctx, hipMemsetAsync(ctx->d_kld_scalars.ptr + 1, 0, sizeof(unsigned long long), ctx->stream));
  MCL_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *cells = ctx->h_kld_scalars[1];
  return MCL_OK;
}

mcl_status mcl_profile_enable(mcl_ctx* ctx, int32_t on) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  ctx->profile_tick = 0;
  ctx->profile = on < 0 ? 0 : (on > 2 ? 2 : on);  // 1: the sensor kernel only, 2: every stage
  return MCL_OK;
}

mcl_status mcl_profile_read(mcl_ctx* ctx, double ms[MCL_NUM_STAGES], uint64_t counts[MCL_NUM_STAGES], int32_t reset) {
  if (!ctx) return MCL_ERR_INVALID_ARGUMENT;
  for (int s = 0; s < MCL_NUM_STAGES; ++s) {
    if (ms) ms[s] = ctx->prof_ms[s];
    if (counts) counts[s] = ctx->prof_count[s];
    if (reset) {
      ctx->prof_ms[s] = 0;
      ctx->prof_count[s] = 0;
    }
  }
  return MCL_OK;
}

}  // extern "C"
