// beluga_amd/ros_amcl.hpp — the surface `beluga_amcl::AmclNode` programs against, backed by libbeluga_mcl.so.
//
// `beluga_amcl` builds its filter as `beluga_ros::Amcl{OccupancyGrid{map}, motion_model_variant, sensor_model_variant,
// AmclParams, execution_policy_variant}` (beluga_amcl/src/amcl_node.cpp:410-433) out of model OBJECTS
// (`beluga::DifferentialDriveModel{params}`, `beluga::LikelihoodFieldModel{params, grid}`, ... :350-408) and then calls
// `particles()`, `likelihood_field()`, `likelihood_field_origin()`, `has_likelihood_field()`, `initialize(pose, covariance)`,
// `initialize_from_map()`, `update_map()`, `update(pose, LaserScan | SparsePointCloud3f)`, `force_update()`
// (beluga_ros/include/beluga_ros/amcl.hpp:102-282).  This header provides exactly those names:
//
//   namespace beluga_amd::models  — parameter structs and model types with the reference's names; a model object only
//                                    carries its parameters (the model itself runs in the device library);
//   beluga_amd::ros::Amcl<Grid>   — the class; `Grid` is the occupancy-grid wrapper (beluga_ros::OccupancyGrid).
//
// so that the node's code compiles with two aliases changed (tests/cpp/amcl_node_bodies.cpp does precisely that):
//   namespace beluga = beluga_amd::models;   using Amcl = beluga_amd::ros::Amcl<beluga_ros::OccupancyGrid>;
// The execution-policy argument is accepted and ignored: every per-particle stage runs on the GPU.
#ifndef BELUGA_AMD_ROS_AMCL_HPP
#define BELUGA_AMD_ROS_AMCL_HPP

#include "beluga_amd/amcl.hpp"

namespace beluga_amd {

namespace models {

using beluga_amd::BeamModelParam;
using beluga_amd::DifferentialDriveModelParam;
using beluga_amd::LikelihoodFieldModelParam;
using beluga_amd::LikelihoodFieldProbModelParam;
using beluga_amd::OmnidirectionalDriveModelParam;

/// beluga::DifferentialDriveModel (motion/differential_drive_model.hpp:74-176).
struct DifferentialDriveModel {
  using param_type = DifferentialDriveModelParam;
  explicit DifferentialDriveModel(const param_type& p) : params(p) {}
  param_type params;
};
using DifferentialDriveModel2d = DifferentialDriveModel;
/// beluga::OmnidirectionalDriveModel (motion/omnidirectional_drive_model.hpp:80-150).
struct OmnidirectionalDriveModel {
  using param_type = OmnidirectionalDriveModelParam;
  explicit OmnidirectionalDriveModel(const param_type& p) : params(p) {}
  param_type params;
};
/// beluga::StationaryModel (motion/stationary_model.hpp:40-62).
struct StationaryModel {};

/// beluga::LikelihoodFieldModel (sensor/likelihood_field_model.hpp:40-97).  The grid is handed to the filter, which owns the map.
template <class OccupancyGrid>
struct LikelihoodFieldModel {
  using param_type = LikelihoodFieldModelParam;
  LikelihoodFieldModel(const param_type& p, const OccupancyGrid&) : params(p) {}
  param_type params;
};
/// beluga::LikelihoodFieldProbModel (sensor/likelihood_field_prob_model.hpp:40-96).
template <class OccupancyGrid>
struct LikelihoodFieldProbModel {
  using param_type = LikelihoodFieldProbModelParam;
  LikelihoodFieldProbModel(const param_type& p, const OccupancyGrid&) : params(p) {}
  param_type params;
};
/// beluga::BeamSensorModel (sensor/beam_model.hpp:60-160).
template <class OccupancyGrid>
struct BeamSensorModel {
  using param_type = BeamModelParam;
  BeamSensorModel(const param_type& p, const OccupancyGrid&) : params(p) {}
  param_type params;
};

}  // namespace models

namespace ros {

/// beluga_ros::AmclParams (beluga_ros/include/beluga_ros/amcl.hpp:54-98): same fields, same defaults.
using AmclParams = beluga_amd::AmclParams;

/// beluga_ros::Amcl (beluga_ros/include/beluga_ros/amcl.hpp:102-282).
template <class OccupancyGrid>
class Amcl {
 public:
  using particle_type = ParticleSet::value_type;
  using motion_model_variant = std::variant<models::DifferentialDriveModel2d, models::OmnidirectionalDriveModel, models::StationaryModel>;
  using sensor_model_variant = std::variant<models::LikelihoodFieldModel<OccupancyGrid>, models::LikelihoodFieldProbModel<OccupancyGrid>,
                                            models::BeamSensorModel<OccupancyGrid>>;
  using execution_policy_variant = std::variant<std::execution::sequenced_policy, std::execution::parallel_policy>;
  using estimation_type = beluga_amd::Amcl::estimation_type;

  /// Constructor (beluga_ros/src/amcl.cpp:28-46).
  Amcl(OccupancyGrid map, motion_model_variant motion_model, sensor_model_variant sensor_model, const AmclParams& params = AmclParams(),
       execution_policy_variant /*execution_policy*/ = std::execution::seq)
      : map_(std::move(map)),
        filter_(OccupancyGridView::from(map_), motion_params(motion_model), sensor_params(sensor_model), params, random_seed()) {
    filter_.use_cluster_based_estimate(true);  // beluga_ros/src/amcl.cpp:125
  }

  /// Returns a reference to the current set of particles (:138).
  [[nodiscard]] const auto& particles() const { return filter_.particles(); }
  /// Returns a reference to the current likelihood field (:141-158). \throw std::runtime_error for the beam model.
  [[nodiscard]] const auto& likelihood_field() const { return filter_.likelihood_field(); }
  /// Returns the current likelihood field origin transform (:161-178). \throw std::runtime_error for the beam model.
  [[nodiscard]] auto likelihood_field_origin() const { return filter_.likelihood_field_origin(); }
  /// Check if the sensor model bears a likelihood field (:181-188).
  [[nodiscard]] bool has_likelihood_field() const { return filter_.has_likelihood_field(); }

  /// Initialize particles using a custom distribution (:191-198): max_particles draws of `distribution(engine)`, each
  /// convertible to SE2d, weight 1.  The draws happen on the host (the distribution is arbitrary caller code).
  template <class Distribution, class = std::enable_if_t<!std::is_convertible_v<Distribution, SE2d>>>
  void initialize(Distribution distribution) {
    std::vector<SE2d> states;
    states.reserve(max_particles_);
    for (std::size_t i = 0; i < max_particles_; ++i) states.emplace_back(SE2d{distribution(engine_)});
    filter_.initialize(states);
  }
  /// Initialize particles with a given pose and covariance (:204-206). \throw std::runtime_error on an invalid covariance.
  template <class Matrix>
  void initialize(const SE2d& pose, const Matrix& covariance) {
    Matrix3d cov{};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) cov[static_cast<std::size_t>(3 * r + c)] = element(covariance, r, c);
    filter_.initialize(pose, cov);
  }
  /// Initialize particles using the default map distribution (:209).
  void initialize_from_map() { filter_.initialize_from_map(); }

  /// Update the map used for localization (:212, beluga_ros/src/amcl.cpp:48-51).
  void update_map(OccupancyGrid map) {
    map_ = std::move(map);
    filter_.update_map(OccupancyGridView::from(map_));
  }

  /// Update particles using laser scan data (:227-228, beluga_ros/src/amcl.cpp:54-63): the scan's points in cartesian
  /// coordinates, moved into the base frame with the scan's origin (an SE3: `origin() * Vector3d{x, y, 0}`).
  template <class LaserScan, class = decltype(std::declval<const LaserScan&>().points_in_cartesian_coordinates())>
  auto update(const SE2d& base_pose_in_odom, const LaserScan& laser_scan) -> std::optional<estimation_type> {
    return update_projected<false>(base_pose_in_odom, laser_scan.points_in_cartesian_coordinates(), se3_data(laser_scan.origin()));
  }
  /// Update particles using point cloud data (:243-244, beluga_ros/src/amcl.cpp:67-81).
  template <class PointCloud, class = decltype(std::declval<const PointCloud&>().points()), class = void>
  auto update(const SE2d& base_pose_in_odom, const PointCloud& point_cloud) -> std::optional<estimation_type> {
    return update_projected<true>(base_pose_in_odom, point_cloud.points(), se3_data(point_cloud.origin()));
  }
  /// Update particles based on motion and sensor information (:259-260, beluga_ros/src/amcl.cpp:83-126).
  auto update(const SE2d& base_pose_in_odom, std::vector<std::pair<double, double>>&& measurement) -> std::optional<estimation_type> {
    return filter_.update(base_pose_in_odom, measurement);
  }

  /// Force a manual update of the particles on the next iteration of the filter (:263).
  void force_update() { filter_.force_update(); }

  /// The underlying filter (device handle, cluster parameters, update info, particle-cloud sampling).
  [[nodiscard]] beluga_amd::Amcl& filter() { return filter_; }

 private:
  static MotionModelParam motion_params(const motion_model_variant& v) {
    return std::visit(
        [](const auto& m) -> MotionModelParam {
          using T = std::decay_t<decltype(m)>;
          if constexpr (std::is_same_v<T, models::StationaryModel>) return StationaryModelParam{};
          else return m.params;
        },
        v);
  }
  static SensorModelParam sensor_params(const sensor_model_variant& v) {
    return std::visit([](const auto& m) -> SensorModelParam { return m.params; }, v);
  }
  static std::uint64_t random_seed() {  // the reference's generators are seeded from the system's entropy source as well
    std::random_device device;
    return (static_cast<std::uint64_t>(device()) << 32) | device();
  }
  template <class Matrix>
  static auto element(const Matrix& m, int r, int c) -> decltype(m(r, c), double()) {  // Eigen::Matrix3d
    return m(r, c);
  }
  static double element(const Matrix3d& m, int r, int c) { return m[static_cast<std::size_t>(3 * r + c)]; }  // row-major array
  template <class SE3>
  static std::array<double, 7> se3_data(const SE3& origin) {  // Sophus::SE3d::data(): qx qy qz qw tx ty tz
    std::array<double, 7> out{};
    for (std::size_t i = 0; i < 7; ++i) out[i] = origin.data()[i];
    return out;
  }
  // What both sensor overloads do in the reference: every point is cast to double, moved by the sensor origin and its x, y
  // kept.  The projection itself is mcl_project_point_cloud (float points) for clouds and the same arithmetic in double for
  // laser points, which the reference computes in double from the start (sensor/data/laser_scan.hpp:73-77).
  template <bool kFloatPoints, class Points>
  auto update_projected(const SE2d& base_pose_in_odom, Points&& points, const std::array<double, 7>& origin)
      -> std::optional<estimation_type> {
    std::vector<std::pair<double, double>> measurement;
    if constexpr (kFloatPoints) {
      std::vector<float> xyz;
      for (const auto& p : points) {
        xyz.push_back(static_cast<float>(p.x()));
        xyz.push_back(static_cast<float>(p.y()));
        xyz.push_back(static_cast<float>(p.z()));
      }
      std::vector<double> flat(2 * (xyz.size() / 3 + 1));
      if (mcl_project_point_cloud(xyz.data(), xyz.size() / 3, origin.data(), flat.data()) != MCL_OK)
        throw std::runtime_error("beluga_amd::ros::Amcl: bad point cloud");
      measurement = beluga_amd::Amcl::pairs_from(flat, xyz.size() / 3);
    } else {
      const double qx = origin[0], qy = origin[1], qz = origin[2], qw = origin[3];
      for (const auto& p : points) {
        const double px = p.x(), py = p.y(), pz = 0.0;
        double ux = qy * pz - qz * py, uy = qz * px - qx * pz, uz = qx * py - qy * px;  // Sophus SO3 * point
        ux += ux;
        uy += uy;
        uz += uz;
        measurement.emplace_back((px + qw * ux + (qy * uz - qz * uy)) + origin[4], (py + qw * uy + (qz * ux - qx * uz)) + origin[5]);
      }
    }
    return filter_.update(base_pose_in_odom, measurement);
  }

  OccupancyGrid map_;
  beluga_amd::Amcl filter_;
  std::size_t max_particles_{filter_.max_particles()};
  std::mt19937_64 engine_{random_seed()};
};

}  // namespace ros
}  // namespace beluga_amd

#endif  // BELUGA_AMD_ROS_AMCL_HPP
