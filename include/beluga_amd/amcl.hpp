// beluga_amd/amcl.hpp — header-only C++17 facade over the C ABI (include/beluga_mcl.h).
//
// Same public surface as `beluga::Amcl` (beluga/include/beluga/algorithm/amcl_core.hpp:81-233) for the
// SE(2) / DifferentialDriveModel / LikelihoodFieldModel|BeamSensorModel instantiation that
// `beluga_ros::Amcl` uses (beluga_ros/include/beluga_ros/amcl.hpp:102-282):
//   ctor(map, motion params, sensor params, AmclParams) ; particles() ; initialize(pose, covariance) ;
//   initialize(states) ; update_map(map) ; update(control_action, measurement) -> optional<pair<pose, cov>> ;
//   force_update() ; likelihood_field().
// Error behaviour follows the reference: `initialize` throws std::runtime_error on an invalid covariance
// (random/multivariate_normal_distribution.hpp:114-124), `update` returns std::nullopt when no update ran
// (amcl_core.hpp:166-172).  Any other failure of the device library throws std::runtime_error with its text.
//
// The reference's value types are Sophus::SE2d / Eigen::Matrix3d; neither library is required here.  `SE2d`
// below has Sophus' memory layout (cos, sin, x, y) and, when <sophus/se2.hpp> is available, converts both ways.
#ifndef BELUGA_AMD_AMCL_HPP
#define BELUGA_AMD_AMCL_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <execution>
#include <iterator>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "beluga_mcl.h"

#if defined(__has_include)
#if __has_include(<sophus/se2.hpp>)
#include <sophus/se2.hpp>
#define BELUGA_AMD_HAS_SOPHUS 1
#endif
#endif

namespace beluga_amd {

/// SE(2) pose with Sophus::SE2d's data() layout: unit complex (cos, sin) then translation (x, y).
struct SE2d {
  double c{1.0}, s{0.0}, x{0.0}, y{0.0};
  SE2d() = default;
  SE2d(double theta, double tx, double ty) : c(std::cos(theta)), s(std::sin(theta)), x(tx), y(ty) {}
  /// From any SE(2) type with Sophus::SE2d's interface (so2() + data() = cos, sin, x, y).
  template <class T, class = decltype(std::declval<const T&>().so2()), class = decltype(std::declval<const T&>().data())>
  SE2d(const T& other) : c(other.data()[0]), s(other.data()[1]), x(other.data()[2]), y(other.data()[3]) {}  // NOLINT
  [[nodiscard]] double angle() const { return std::atan2(s, c); }
  [[nodiscard]] const double* data() const { return &c; }
  [[nodiscard]] double* data() { return &c; }
#ifdef BELUGA_AMD_HAS_SOPHUS
  operator Sophus::SE2d() const {                                                                    // NOLINT
    Sophus::SE2d out;
    out.so2().data()[0] = c;
    out.so2().data()[1] = s;
    out.translation() = Eigen::Vector2d{x, y};
    return out;
  }
#endif
};
static_assert(sizeof(SE2d) == 4 * sizeof(double), "SE2d must be four packed doubles");

using Matrix3d = std::array<double, 9>;  ///< row-major 3x3

/// beluga::AmclParams (amcl_core.hpp:34-55) + the spatial hash resolutions of beluga_ros::AmclParams.
struct AmclParams {
  double update_min_d = 0.25;
  double update_min_a = 0.2;
  std::size_t resample_interval = 1UL;
  bool selective_resampling = false;
  std::size_t min_particles = 500UL;
  std::size_t max_particles = 2000UL;
  double alpha_slow = 0.001;
  double alpha_fast = 0.1;
  double kld_epsilon = 0.05;
  double kld_z = 3.0;
  double spatial_resolution_x = 0.5;
  double spatial_resolution_y = 0.5;
  double spatial_resolution_theta = 10.0 * 3.14159265358979323846 / 180.0;
};

/// beluga::DifferentialDriveModelParam (motion/differential_drive_model.hpp:40-68).
struct DifferentialDriveModelParam {
  double rotation_noise_from_rotation;
  double rotation_noise_from_translation;
  double translation_noise_from_translation;
  double translation_noise_from_rotation;
  double distance_threshold = 0.01;
};

/// beluga::OmnidirectionalDriveModelParam (motion/omnidirectional_drive_model.hpp:36-73).
struct OmnidirectionalDriveModelParam {
  double rotation_noise_from_rotation;
  double rotation_noise_from_translation;
  double translation_noise_from_translation;
  double translation_noise_from_rotation;
  double strafe_noise_from_translation;
  double distance_threshold = 0.01;
};

/// beluga::StationaryModel (motion/stationary_model.hpp:40-62) has no parameters.
struct StationaryModelParam {};

using MotionModelParam = std::variant<DifferentialDriveModelParam, OmnidirectionalDriveModelParam, StationaryModelParam>;

/// beluga::LikelihoodFieldModelParam (sensor/likelihood_field_model_base.hpp:42-64).
struct LikelihoodFieldModelParam {
  double max_obstacle_distance = 100.0;
  double max_laser_distance = 2.0;
  double z_hit = 0.5;
  double z_random = 0.5;
  double sigma_hit = 0.2;
  bool model_unknown_space = false;
  bool only_obstacle_boundaries = false;
};

/// beluga::BeamModelParam (sensor/beam_model.hpp:43-58).
struct BeamModelParam {
  double z_hit{0.5};
  double z_short{0.5};
  double z_max{0.05};
  double z_rand{0.05};
  double sigma_hit{0.2};
  double lambda_short{0.1};
  double beam_max_range{60};
};

/// beluga::LikelihoodFieldProbModelParam (sensor/likelihood_field_prob_model.hpp:34): same fields, other weighting.
struct LikelihoodFieldProbModelParam : LikelihoodFieldModelParam {};

using SensorModelParam = std::variant<LikelihoodFieldModelParam, BeamModelParam, LikelihoodFieldProbModelParam>;

/// A non-owning view of anything satisfying OccupancyGrid2 (sensor/data/occupancy_grid.hpp:39-75).
struct OccupancyGridView {
  const std::int8_t* cells{nullptr};  ///< row-major, height x width
  std::uint32_t width{0}, height{0};
  double resolution{0.0};
  SE2d origin{};
  std::int8_t free_value{0}, unknown_value{-1}, occupied_value{100};  ///< beluga_ros::OccupancyGrid::ValueTraits

  /// Adapts a grid type with width()/height()/resolution()/origin()/data() whose cell type is int8.  The value traits are
  /// the grid's own (`Grid::ValueTraits::kFreeValue / kUnknownValue / kOccupiedValue`, beluga_ros/occupancy_grid.hpp:48-64)
  /// when it declares them, the ROS trinary interpretation 0 / -1 / 100 otherwise.
  template <class Grid>
  static OccupancyGridView from(const Grid& grid) {
    OccupancyGridView v;
    v.cells = reinterpret_cast<const std::int8_t*>(&*std::begin(grid.data()));
    v.width = static_cast<std::uint32_t>(grid.width());
    v.height = static_cast<std::uint32_t>(grid.height());
    v.resolution = grid.resolution();
    v.origin = SE2d{grid.origin()};
    read_traits<Grid>(v, 0);
    return v;
  }

 private:
  template <class Grid>
  static auto read_traits(OccupancyGridView& v, int) -> decltype(Grid::ValueTraits::kFreeValue, void()) {
    v.free_value = static_cast<std::int8_t>(Grid::ValueTraits::kFreeValue);
    v.unknown_value = static_cast<std::int8_t>(Grid::ValueTraits::kUnknownValue);
    v.occupied_value = static_cast<std::int8_t>(Grid::ValueTraits::kOccupiedValue);
  }
  template <class Grid>
  static void read_traits(OccupancyGridView&, long) {}
};

/// What beluga_ros::LaserScan wraps (beluga_ros/include/beluga_ros/laser_scan.hpp:46-66): the sensor_msgs/LaserScan
/// fields, the laser origin in the base frame and the decimation / range limits.
struct LaserScan {
  std::vector<float> ranges;
  float angle_min{0.F}, angle_increment{0.F};
  float range_min{0.F}, range_max{0.F};
  std::array<double, 7> origin{0, 0, 0, 1, 0, 0, 0};  ///< Sophus::SE3d::data(): qx qy qz qw tx ty tz
  std::size_t max_beams{static_cast<std::size_t>(-1)};
  double min_range{2.2250738585072014e-308};
  double max_range{1.7976931348623157e308};
};

/// Host mirror of the particle set: what `beluga::TupleVector<std::tuple<SE2d, Weight>>` holds.  A sized random-access range
/// of (state, weight) tuples — `std::get<0>(p)` / `std::get<1>(p)` are what `beluga::state(p)` / `beluga::weight(p)` read
/// (type_traits/particle_traits.hpp) — plus the two component ranges `beluga::views::states / weights` project.
struct ParticleSet {
  using value_type = std::tuple<SE2d, double>;
  using reference = std::tuple<const SE2d&, const double&>;
  std::vector<SE2d> states;
  std::vector<double> weights;

  class const_iterator {
   public:
    using iterator_category = std::random_access_iterator_tag;
    using value_type = ParticleSet::value_type;
    using difference_type = std::ptrdiff_t;
    using pointer = void;
    using reference = ParticleSet::reference;
    const_iterator() = default;
    const_iterator(const ParticleSet* set, std::size_t index) : set_(set), index_(index) {}
    reference operator*() const { return reference{set_->states[index_], set_->weights[index_]}; }
    reference operator[](difference_type k) const { return *(*this + k); }
    const_iterator& operator++() { ++index_; return *this; }
    const_iterator operator++(int) { auto old = *this; ++index_; return old; }
    const_iterator& operator--() { --index_; return *this; }
    const_iterator operator--(int) { auto old = *this; --index_; return old; }
    const_iterator& operator+=(difference_type k) { index_ = static_cast<std::size_t>(static_cast<difference_type>(index_) + k); return *this; }
    const_iterator& operator-=(difference_type k) { return *this += -k; }
    friend const_iterator operator+(const_iterator it, difference_type k) { return it += k; }
    friend const_iterator operator+(difference_type k, const_iterator it) { return it += k; }
    friend const_iterator operator-(const_iterator it, difference_type k) { return it -= k; }
    friend difference_type operator-(const const_iterator& a, const const_iterator& b) {
      return static_cast<difference_type>(a.index_) - static_cast<difference_type>(b.index_);
    }
    friend bool operator==(const const_iterator& a, const const_iterator& b) { return a.index_ == b.index_; }
    friend bool operator!=(const const_iterator& a, const const_iterator& b) { return a.index_ != b.index_; }
    friend bool operator<(const const_iterator& a, const const_iterator& b) { return a.index_ < b.index_; }
    friend bool operator>(const const_iterator& a, const const_iterator& b) { return a.index_ > b.index_; }
    friend bool operator<=(const const_iterator& a, const const_iterator& b) { return a.index_ <= b.index_; }
    friend bool operator>=(const const_iterator& a, const const_iterator& b) { return a.index_ >= b.index_; }

   private:
    const ParticleSet* set_{nullptr};
    std::size_t index_{0};
  };
  using iterator = const_iterator;

  [[nodiscard]] std::size_t size() const { return weights.size(); }
  [[nodiscard]] bool empty() const { return weights.empty(); }
  [[nodiscard]] const_iterator begin() const { return {this, 0}; }
  [[nodiscard]] const_iterator end() const { return {this, size()}; }
  [[nodiscard]] reference operator[](std::size_t i) const { return reference{states[i], weights[i]}; }
};

/// `beluga::views::states(particles)` / `beluga::views::weights(particles)` for the host mirror (views/particles.hpp).
namespace views {
inline const std::vector<SE2d>& states(const ParticleSet& particles) { return particles.states; }
inline const std::vector<double>& weights(const ParticleSet& particles) { return particles.weights; }
}  // namespace views

/// What `likelihood_field()` returns: the accessors of `beluga::ValueGrid2<float>` (sensor/data/value_grid.hpp:36-69) that
/// `beluga_ros::assign_likelihood_field` reads (beluga_ros/include/beluga_ros/likelihood_field.hpp:31-66).
template <class T>
class ValueGrid2 {
 public:
  ValueGrid2() = default;
  ValueGrid2(std::vector<T> data, std::size_t width, double resolution) : data_(std::move(data)), width_(width), resolution_(resolution) {}
  [[nodiscard]] std::size_t size() const { return data_.size(); }
  [[nodiscard]] const std::vector<T>& data() const { return data_; }
  [[nodiscard]] std::size_t width() const { return width_; }
  [[nodiscard]] std::size_t height() const { return width_ ? data_.size() / width_ : 0; }
  [[nodiscard]] double resolution() const { return resolution_; }
  [[nodiscard]] const T& operator[](std::size_t i) const { return data_[i]; }

 private:
  std::vector<T> data_;
  std::size_t width_{0};
  double resolution_{1.0};
};

/// A contiguous shard of one logical filter's particles (include/beluga_mcl.h, "Particle shards"): one `Amcl` per GPU, every
/// instance with its shard, the same map, control actions and scans; after `attach` / `attach_rccl`, `update()` runs the cycle
/// over all shards (fixed-size and KLD-adaptive) and every instance returns the same estimate.  `particles()` is the shard.
struct Shard {
  std::uint64_t offset{0};    ///< global index of the shard's first particle
  std::uint64_t capacity{0};  ///< particles the shard can hold (its share of max_particles); 0: not sharded
  /// The balanced split the library itself uses when it re-balances: rank `rank` of `world`.
  static Shard of(std::uint64_t max_particles, unsigned rank, unsigned world) {
    const std::uint64_t base = max_particles / world, rem = max_particles % world;
    return Shard{rank * base + (rank < rem ? rank : rem), base + (rank < rem ? 1u : 0u)};
  }
};

class Amcl {
 public:
  using state_type = SE2d;
  using measurement_type = std::vector<std::pair<double, double>>;
  using estimation_type = std::pair<SE2d, Matrix3d>;

  /// `options`: library switches applied before the map is installed (mcl_set_option), e.g. {{"field_build", 1}} to build
  /// the likelihood field with the device's exact distance transform instead of the reference's wavefront on the host.
  Amcl(const OccupancyGridView& map, const MotionModelParam& motion, const SensorModelParam& sensor,
       const AmclParams& params = AmclParams{}, std::uint64_t seed = 0, int device = 0,
       const std::vector<std::pair<std::string, std::int64_t>>& options = {}, const Shard& shard = Shard{}) {
    mcl_config cfg;
    mcl_default_config(&cfg);
    cfg.device_id = device;
    cfg.seed = seed;
    cfg.shard_offset = shard.offset;
    cfg.shard_capacity = shard.capacity;
    cfg.amcl.update_min_d = params.update_min_d;
    cfg.amcl.update_min_a = params.update_min_a;
    cfg.amcl.resample_interval = params.resample_interval;
    cfg.amcl.selective_resampling = params.selective_resampling ? 1 : 0;
    cfg.amcl.min_particles = params.min_particles;
    cfg.amcl.max_particles = params.max_particles;
    cfg.amcl.alpha_slow = params.alpha_slow;
    cfg.amcl.alpha_fast = params.alpha_fast;
    cfg.amcl.kld_epsilon = params.kld_epsilon;
    cfg.amcl.kld_z = params.kld_z;
    cfg.amcl.spatial_resolution_x = params.spatial_resolution_x;
    cfg.amcl.spatial_resolution_y = params.spatial_resolution_y;
    cfg.amcl.spatial_resolution_theta = params.spatial_resolution_theta;
    if (const auto* dd = std::get_if<DifferentialDriveModelParam>(&motion)) {
      cfg.motion_kind = MCL_MOTION_DIFFERENTIAL;
      cfg.motion = mcl_diffdrive_params{dd->rotation_noise_from_rotation, dd->rotation_noise_from_translation,
                                        dd->translation_noise_from_translation, dd->translation_noise_from_rotation,
                                        dd->distance_threshold};
    } else if (const auto* om = std::get_if<OmnidirectionalDriveModelParam>(&motion)) {
      cfg.motion_kind = MCL_MOTION_OMNIDIRECTIONAL;
      cfg.motion = mcl_diffdrive_params{om->rotation_noise_from_rotation, om->rotation_noise_from_translation,
                                        om->translation_noise_from_translation, om->translation_noise_from_rotation,
                                        om->distance_threshold};
      cfg.strafe_noise_from_translation = om->strafe_noise_from_translation;
    } else {
      cfg.motion_kind = MCL_MOTION_STATIONARY;
    }
    const LikelihoodFieldModelParam* lf = std::get_if<LikelihoodFieldModelParam>(&sensor);
    if (!lf) lf = std::get_if<LikelihoodFieldProbModelParam>(&sensor);
    if (lf) {
      cfg.sensor_kind = std::holds_alternative<LikelihoodFieldProbModelParam>(sensor) ? MCL_SENSOR_LIKELIHOOD_FIELD_PROB
                                                                                     : MCL_SENSOR_LIKELIHOOD_FIELD;
      cfg.lf = mcl_lf_params{lf->max_obstacle_distance, lf->max_laser_distance, lf->z_hit, lf->z_random, lf->sigma_hit,
                             lf->model_unknown_space ? 1 : 0, lf->only_obstacle_boundaries ? 1 : 0};
    } else {
      const auto& b = std::get<BeamModelParam>(sensor);
      cfg.sensor_kind = MCL_SENSOR_BEAM;
      cfg.beam = mcl_beam_params{b.z_hit, b.z_short, b.z_max, b.z_rand, b.sigma_hit, b.lambda_short, b.beam_max_range};
    }
    max_particles_ = params.max_particles;
    const mcl_status st = mcl_create(&cfg, &ctx_);
    if (st != MCL_OK) throw std::runtime_error(std::string("beluga_amd::Amcl: ") + mcl_last_error(nullptr));
    try {
      for (const auto& [name, value] : options) check(mcl_set_option(ctx_, name.c_str(), value));
      update_map(map);
    } catch (...) {
      mcl_destroy(ctx_);
      ctx_ = nullptr;
      throw;
    }
  }
  Amcl(const Amcl&) = delete;
  Amcl& operator=(const Amcl&) = delete;
  Amcl(Amcl&& other) noexcept
      : ctx_(other.ctx_),
        width_(other.width_),
        height_(other.height_),
        max_particles_(other.max_particles_),
        resolution_(other.resolution_),
        has_field_(other.has_field_) {
    other.ctx_ = nullptr;
  }
  ~Amcl() { mcl_destroy(ctx_); }

  /// Returns a reference to the current set of particles (amcl_core.hpp:127). Downloaded lazily.
  [[nodiscard]] const ParticleSet& particles() const {
    if (dirty_) {
      std::uint64_t n = 0;
      check(mcl_num_particles(ctx_, &n));
      mirror_.states.resize(n);
      mirror_.weights.resize(n);
      if (n) check(mcl_get_particles(ctx_, mirror_.states.data()->data(), mirror_.weights.data(), n, &n));
      dirty_ = false;
    }
    return mirror_;
  }

  /// Initialize particles with a given pose and covariance (amcl_core.hpp:145-147).
  /// \throw std::runtime_error If the provided covariance is invalid.
  void initialize(const SE2d& pose, const Matrix3d& covariance) {
    const double mean[3] = {pose.x, pose.y, pose.angle()};
    const mcl_status st = mcl_initialize_normal(ctx_, mean, covariance.data());
    if (st == MCL_ERR_BAD_COVARIANCE) throw std::runtime_error("Invalid covariance matrix");
    check(st);
    dirty_ = true;
  }

  /// Initialize particles from caller-drawn states, weight 1 each (amcl_core.hpp:131-137).
  void initialize(const std::vector<SE2d>& states) {
    const std::vector<double> ones(states.size(), 1.0);
    check(mcl_set_particles(ctx_, states.empty() ? nullptr : states.data()->data(), ones.data(), states.size()));
    dirty_ = true;
  }

  /// Update the map used for localization (amcl_core.hpp:150).
  void update_map(const OccupancyGridView& map) {
    const std::int8_t traits[3] = {map.free_value, map.unknown_value, map.occupied_value};
    check(mcl_set_map(ctx_, map.cells, map.width, map.height, map.resolution, map.origin.data(), traits));
    have_pending_ = false;  // (a map given now replaces one that was still on its way)
    width_ = map.width;
    height_ = map.height;
    resolution_ = map.resolution;
    std::int32_t has = 0;
    check(mcl_has_likelihood_field(ctx_, &has));
    has_field_ = has != 0;
    field_.reset();
  }

  /// Extension (mcl_set_map_async): the new map's likelihood field is built on a worker thread - the reference's update_map blocks the
  /// caller for the build, seconds at 16 M cells - while the filter keeps running on the map it has; the swap happens at the start of the
  /// first update() after the build is done, or in map_commit().  `map` is copied before the call returns.
  void update_map_async(const OccupancyGridView& map) {
    const std::int8_t traits[3] = {map.free_value, map.unknown_value, map.occupied_value};
    check(mcl_set_map_async(ctx_, map.cells, map.width, map.height, map.resolution, map.origin.data(), traits));
    pending_ = {map.width, map.height, map.resolution};
    have_pending_ = true;
  }
  /// 0: no map on its way, 1: its field is being built, 2: built, waiting for the swap.
  int map_pending() {
    std::int32_t state = 0;
    check(mcl_map_pending(ctx_, &state));
    if (state == 0 && have_pending_) {  // the swap has happened (inside an update, or in map_commit)
      width_ = pending_.width;
      height_ = pending_.height;
      resolution_ = pending_.resolution;
      have_pending_ = false;
      field_.reset();
    }
    return state;
  }
  void map_commit(bool wait = true) {
    check(mcl_map_commit(ctx_, wait ? 1 : 0));
    (void)map_pending();
  }

  /// The C ABI writes points as a flat double[2 m]; the measurement type is a vector of pairs (no aliasing between the two).
  static measurement_type pairs_from(const std::vector<double>& flat, std::size_t m) {
    measurement_type points;
    points.reserve(m);
    for (std::size_t i = 0; i < m; ++i) points.emplace_back(flat[2 * i], flat[2 * i + 1]);
    return points;
  }

  /// Update particles based on motion and sensor information (amcl_core.hpp:165-201).  An empty particle set returns
  /// std::nullopt before the motion policy sees the control action (:166-168; mcl_update checks it first).
  auto update(const SE2d& control_action, const measurement_type& measurement) -> std::optional<estimation_type> {
    static_assert(sizeof(std::pair<double, double>) == 2 * sizeof(double), "measurement points must be packed pairs");
    mcl_estimate est;
    mcl_update_info info;
    check(mcl_update(ctx_, control_action.data(), measurement.empty() ? nullptr : &measurement.front().first, measurement.size(),
                     &est, &info));
    last_info_ = info;
    if (have_pending_) (void)map_pending();  // (a map given to update_map_async may have taken over inside this call)
    if (!info.updated) return std::nullopt;
    dirty_ = true;
    estimation_type out;
    out.first.c = est.pose[0];
    out.first.s = est.pose[1];
    out.first.x = est.pose[2];
    out.first.y = est.pose[3];
    for (int i = 0; i < 9; ++i) out.second[static_cast<std::size_t>(i)] = est.covariance[i];
    return out;
  }

  /// beluga_ros::Amcl::update(base_pose_in_odom, laser_scan) (beluga_ros/src/amcl.cpp:54-63).
  auto update(const SE2d& base_pose_in_odom, const LaserScan& laser_scan) -> std::optional<estimation_type> {
    mcl_laser_scan scan;
    scan.ranges = laser_scan.ranges.data();
    scan.num_ranges = laser_scan.ranges.size();
    scan.angle_min = laser_scan.angle_min;
    scan.angle_increment = laser_scan.angle_increment;
    scan.range_min = laser_scan.range_min;
    scan.range_max = laser_scan.range_max;
    for (std::size_t i = 0; i < 7; ++i) scan.origin_se3[i] = laser_scan.origin[i];
    scan.max_beams = laser_scan.max_beams;
    scan.min_range = laser_scan.min_range;
    scan.max_range = laser_scan.max_range;
    std::vector<double> flat(2 * (std::min<std::size_t>(laser_scan.ranges.size(), laser_scan.max_beams) + 1));
    std::uint64_t m = 0;
    check(mcl_prepare_laser_scan(&scan, flat.data(), &m));
    return update(base_pose_in_odom, pairs_from(flat, m));
  }

  /// Force a manual update of the particles on the next iteration of the filter (amcl_core.hpp:204).
  void force_update() { check(mcl_force_update(ctx_)); }

  /// Joins the communicator of a sharded filter (this instance was constructed with its `Shard`).  `transport`: the two
  /// collectives of the exchange over device buffers (MPI, threads of one process, ...); copied, its `user` must outlive this.
  /// A COLLECTIVE call for world > 1 (the ranks exchange a word of their configuration): every rank attaches, concurrently.
  void attach(unsigned rank, unsigned world, const mcl_transport& transport) { check(mcl_comm_attach(ctx_, rank, world, &transport)); }
  /// The same over RCCL / xGMI (librccl.so is loaded at run time): rank 0 calls rccl_unique_id() and hands the 128 bytes to
  /// the other ranks by whatever means the host has.
  void attach_rccl(const std::array<std::uint8_t, 128>& id, unsigned rank, unsigned world) {
    check(mcl_comm_attach_rccl(ctx_, id.data(), rank, world));
  }
  [[nodiscard]] static std::array<std::uint8_t, 128> rccl_unique_id() {
    std::array<std::uint8_t, 128> id{};
    if (mcl_comm_unique_id(id.data()) != MCL_OK) throw std::runtime_error(std::string("beluga_amd::Amcl: ") + mcl_last_error(nullptr));
    return id;
  }

  /// The pose sample behind beluga_ros::assign_particle_cloud(particles, size, PoseArray&)
  /// (beluga_ros/include/beluga_ros/particle_cloud.hpp:131-149): `size` states drawn with probability proportional to the
  /// weights (`views::sample | take_exactly(size)`); the set is not modified.  Pass a new draw_id per publication.
  [[nodiscard]] std::vector<SE2d> sample_particle_cloud(std::size_t size, std::uint32_t draw_id = 0) const {
    std::uint64_t n = 0;
    check(mcl_num_particles(ctx_, &n));
    std::vector<SE2d> out(n ? size : 0);
    if (!out.empty()) check(mcl_sample_particle_cloud(ctx_, out.size(), draw_id, reinterpret_cast<double*>(out.data())));
    return out;
  }

  /// beluga::cluster_based_estimate (algorithm/cluster_based_estimation.hpp:415-433) of the current particle set.
  [[nodiscard]] estimation_type cluster_based_estimate(double linear_hash_resolution = 0.20, double angular_hash_resolution = 0.524,
                                                       double weight_cap_percentile = 0.90) const {
    const mcl_cluster_params cp{linear_hash_resolution, angular_hash_resolution, weight_cap_percentile};
    mcl_estimate est;
    check(mcl_cluster_based_estimate(ctx_, &cp, &est));
    estimation_type out;
    out.first.c = est.pose[0];
    out.first.s = est.pose[1];
    out.first.x = est.pose[2];
    out.first.y = est.pose[3];
    for (int i = 0; i < 9; ++i) out.second[static_cast<std::size_t>(i)] = est.covariance[i];
    return out;
  }

  /// Makes update() return cluster_based_estimate, as beluga_ros::Amcl does (beluga_ros/src/amcl.cpp:125), instead of
  /// beluga::estimate, as beluga::Amcl does (amcl_core.hpp:200).  On a sharded filter (attach) a COLLECTIVE call: every rank, concurrently, alike.
  void use_cluster_based_estimate(bool enable) { check(mcl_set_estimate_kind(ctx_, enable ? 1 : 0, nullptr)); }

  /// beluga_ros::Amcl::likelihood_field() (beluga_ros/include/beluga_ros/amcl.hpp:141-158;
  /// LikelihoodFieldModelBase::likelihood_field(), likelihood_field_model_base.hpp:102), row-major height x width.
  /// \throw std::runtime_error If the sensor model has no likelihood field (the beam model).
  [[nodiscard]] const ValueGrid2<float>& likelihood_field() const {
    if (!has_field_) throw std::runtime_error("The current sensor model does not support likelihood field");
    if (!field_) {
      std::vector<float> data(static_cast<std::size_t>(width_) * height_);
      check(mcl_get_likelihood_field(ctx_, data.data()));
      field_.emplace(std::move(data), width_, resolution_);
    }
    return *field_;
  }

  /// beluga_ros::Amcl::likelihood_field_origin() (beluga_ros/include/beluga_ros/amcl.hpp:161-178).
  /// \throw std::runtime_error If the sensor model has no likelihood field.
  [[nodiscard]] SE2d likelihood_field_origin() const {
    SE2d origin;
    const mcl_status st = mcl_get_likelihood_field_origin(ctx_, origin.data());
    if (st == MCL_ERR_UNSUPPORTED) throw std::runtime_error("The current sensor model does not support likelihood field");
    check(st);
    return origin;
  }

  /// beluga_ros::Amcl::has_likelihood_field() (beluga_ros/include/beluga_ros/amcl.hpp:181-188).
  [[nodiscard]] bool has_likelihood_field() const { return has_field_; }

  /// beluga_ros::Amcl::initialize_from_map() (beluga_ros/include/beluga_ros/amcl.hpp:209): max_particles states drawn
  /// uniformly over the free cells of the map (random/multivariate_uniform_distribution.hpp:126-161).
  void initialize_from_map() {
    check(mcl_initialize_from_map(ctx_));
    dirty_ = true;
  }

  /// beluga_ros::Amcl::update(base_pose_in_odom, point_cloud) (beluga_ros/src/amcl.cpp:67-81): `points_xyz` are the cloud's
  /// points (3 floats each) in the sensor frame, `origin` the sensor pose in the base frame as Sophus::SE3d::data().
  auto update(const SE2d& base_pose_in_odom, const std::vector<float>& points_xyz, const std::array<double, 7>& origin)
      -> std::optional<estimation_type> {
    std::vector<double> flat(2 * (points_xyz.size() / 3 + 1));
    check(mcl_project_point_cloud(points_xyz.data(), points_xyz.size() / 3, origin.data(), flat.data()));
    return update(base_pose_in_odom, pairs_from(flat, points_xyz.size() / 3));
  }

  [[nodiscard]] const mcl_update_info& last_update_info() const { return last_info_; }
  [[nodiscard]] std::size_t max_particles() const { return max_particles_; }
  [[nodiscard]] mcl_ctx* native_handle() const { return ctx_; }

 private:
  void check(mcl_status st) const {
    if (st != MCL_OK) throw std::runtime_error(std::string("beluga_amd::Amcl: ") + mcl_last_error(ctx_));
  }
  mcl_ctx* ctx_{nullptr};
  struct PendingShape {
    std::uint32_t width{0}, height{0};
    double resolution{0.0};
  };
  PendingShape pending_{};  ///< of the map given to update_map_async, until its swap
  bool have_pending_{false};
  std::uint32_t width_{0}, height_{0};
  std::size_t max_particles_{0};
  double resolution_{0.0};
  bool has_field_{false};
  mutable ParticleSet mirror_;
  mutable bool dirty_{true};
  mutable std::optional<ValueGrid2<float>> field_;
  mcl_update_info last_info_{};
};

}  // namespace beluga_amd

#endif  // BELUGA_AMD_AMCL_HPP
