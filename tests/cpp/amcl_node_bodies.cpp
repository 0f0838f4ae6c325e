// The parts of beluga_amcl::AmclNode that touch the particle filter, compiled against beluga_amd with nothing but the
// filter's type changed:
//   get_motion_model / get_sensor_model        beluga_amcl/src/amcl_node.cpp:350-408
//   make_particle_filter                        :410-433
//   map_callback's filter + likelihood-field publishing block   :456-476
//   do_periodic_timer_callback's particle-cloud blocks          :500-514
//   initialize_from_estimate / initialize_from_map              :683-721
//   the update call of sensor_callback                          :603
// Everything ROS (messages, parameters, logging, beluga_ros' message helpers) is a small stand-in below; the member
// function bodies are the node's, with `beluga` / `beluga_ros::Amcl` resolving to the aliases at the top of this file.
// tests/test_cpp_facade.py compiles this with -Wall -Wextra -Werror and, on a GPU box, runs it.
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <string_view>

#include "beluga_amd/ros_amcl.hpp"

// ---- stand-ins for what the node includes ----------------------------------------------------------------------
namespace Sophus {
struct SO2d {
  double c{1}, s{0};
  [[nodiscard]] double log() const { return std::atan2(s, c); }
};
struct Vector2d {
  double vx{0}, vy{0};
  [[nodiscard]] double x() const { return vx; }
  [[nodiscard]] double y() const { return vy; }
};
struct SE2d {  // Sophus::SE2d's accessors the node uses; data() = (cos, sin, x, y)
  double v[4]{1, 0, 0, 0};
  SE2d() = default;
  SE2d(double theta, double x, double y) : v{std::cos(theta), std::sin(theta), x, y} {}
  SE2d(const beluga_amd::SE2d& p) : v{p.c, p.s, p.x, p.y} {}  // NOLINT
  [[nodiscard]] const double* data() const { return v; }
  [[nodiscard]] SO2d so2() const { return SO2d{v[0], v[1]}; }
  [[nodiscard]] Vector2d translation() const { return Vector2d{v[2], v[3]}; }
};
struct SE3d {  // data() = (qx, qy, qz, qw, tx, ty, tz)
  double v[7]{0, 0, 0, 1, 0, 0, 0};
  [[nodiscard]] const double* data() const { return v; }
};
}  // namespace Sophus
namespace Eigen {
struct Matrix3d {
  double m[9]{};
  double operator()(int r, int c) const { return m[3 * r + c]; }
  double& operator()(int r, int c) { return m[3 * r + c]; }
};
struct Vector3f {
  float v[3]{};
  [[nodiscard]] float x() const { return v[0]; }
  [[nodiscard]] float y() const { return v[1]; }
  [[nodiscard]] float z() const { return v[2]; }
};
struct Vector2d {
  double v[2]{};
  [[nodiscard]] double x() const { return v[0]; }
  [[nodiscard]] double y() const { return v[1]; }
};
}  // namespace Eigen

namespace geometry_msgs::msg {
struct Pose {
  double position[3]{}, orientation[4]{0, 0, 0, 1};
};
struct PoseArray {
  std::vector<Pose> poses;
};
}  // namespace geometry_msgs::msg
namespace nav_msgs::msg {
struct MapMetaData {
  unsigned int width{0}, height{0};
  float resolution{0.F};
  geometry_msgs::msg::Pose origin;
};
struct OccupancyGrid {
  using SharedPtr = std::shared_ptr<OccupancyGrid>;
  using ConstSharedPtr = std::shared_ptr<const OccupancyGrid>;
  MapMetaData info;
  std::vector<std::int8_t> data;
};
}  // namespace nav_msgs::msg

namespace beluga_ros {
// beluga_ros/include/beluga_ros/occupancy_grid.hpp:41-110
class OccupancyGrid {
 public:
  struct ValueTraits {
    static constexpr std::int8_t kFreeValue = 0;
    static constexpr std::int8_t kUnknownValue = -1;
    static constexpr std::int8_t kOccupiedValue = 100;
  };
  explicit OccupancyGrid(nav_msgs::msg::OccupancyGrid::ConstSharedPtr grid)
      : grid_(std::move(grid)),
        origin_(2.0 * std::atan2(grid_->info.origin.orientation[2], grid_->info.origin.orientation[3]), grid_->info.origin.position[0],
                grid_->info.origin.position[1]) {}
  [[nodiscard]] const Sophus::SE2d& origin() const { return origin_; }
  [[nodiscard]] std::size_t size() const { return grid_->data.size(); }
  [[nodiscard]] const auto& data() const { return grid_->data; }
  [[nodiscard]] std::size_t width() const { return grid_->info.width; }
  [[nodiscard]] std::size_t height() const { return grid_->info.height; }
  [[nodiscard]] double resolution() const { return grid_->info.resolution; }

 private:
  nav_msgs::msg::OccupancyGrid::ConstSharedPtr grid_;
  Sophus::SE2d origin_;
};
// beluga_ros/include/beluga_ros/laser_scan.hpp:46-100 (the two members Amcl::update reads)
struct LaserScan {
  std::vector<Eigen::Vector2d> points;
  Sophus::SE3d sensor_origin;
  [[nodiscard]] const auto& points_in_cartesian_coordinates() const { return points; }
  [[nodiscard]] const auto& origin() const { return sensor_origin; }
  [[nodiscard]] std::size_t size() const { return points.size(); }
};
// beluga_ros/include/beluga_ros/sparse_point_cloud.hpp:53-132
struct SparsePointCloud3f {
  std::vector<Eigen::Vector3f> cloud;
  Sophus::SE3d sensor_origin;
  [[nodiscard]] const auto& points() const { return cloud; }
  [[nodiscard]] const auto& origin() const { return sensor_origin; }
  [[nodiscard]] std::size_t size() const { return cloud.size(); }
};
// beluga_ros/include/beluga_ros/likelihood_field.hpp:31-66 (metadata + normalisation to [0, 100])
template <class Grid>
void assign_likelihood_field(const Grid& likelihood_field, const Sophus::SE2d& origin, nav_msgs::msg::OccupancyGrid& message) {
  message.info.width = static_cast<unsigned int>(likelihood_field.width());
  message.info.height = static_cast<unsigned int>(likelihood_field.height());
  message.info.resolution = static_cast<float>(likelihood_field.resolution());
  message.info.origin.position[0] = origin.translation().x();
  message.info.origin.position[1] = origin.translation().y();
  const auto& grid_data = likelihood_field.data();
  message.data.resize(likelihood_field.size());
  const auto [min_it, max_it] = std::minmax_element(grid_data.begin(), grid_data.end());
  const float min_val = *min_it, range = *max_it - *min_it;
  for (std::size_t i = 0; i < grid_data.size(); ++i)
    message.data[i] = range > 0.F ? static_cast<std::int8_t>((grid_data[i] - min_val) / range * 100.0F) : std::int8_t{0};
}
// beluga_ros/include/beluga_ros/particle_cloud.hpp:131-163: here every particle's state, in order (the reference samples
// `size(particles)` of them by weight; beluga_amd::Amcl::sample_particle_cloud is that sample).
template <class Particles>
geometry_msgs::msg::PoseArray& assign_particle_cloud(const Particles& particles, geometry_msgs::msg::PoseArray& message) {
  message.poses.clear();
  message.poses.reserve(particles.size());
  for (const auto& particle : particles) {
    const auto& state = std::get<0>(particle);
    auto& pose = message.poses.emplace_back();
    pose.position[0] = state.x;
    pose.position[1] = state.y;
    pose.orientation[2] = std::sin(0.5 * state.angle());
    pose.orientation[3] = std::cos(0.5 * state.angle());
  }
  return message;
}
template <class Message>
void stamp_message(const std::string&, int, Message&) {}
}  // namespace beluga_ros

// ---- the two aliases a maintainer changes ------------------------------------------------------------------------
namespace beluga = beluga_amd::models;  // beluga::DifferentialDriveModel{params}, beluga::LikelihoodFieldModel{params, grid}, ...
namespace beluga_ros {
using Amcl = beluga_amd::ros::Amcl<beluga_ros::OccupancyGrid>;  // was: class beluga_ros::Amcl
using AmclParams = beluga_amd::ros::AmclParams;
}  // namespace beluga_ros

#define RCLCPP_INFO(logger, ...) (void)std::snprintf(nullptr, 0, __VA_ARGS__)
#define RCLCPP_ERROR(logger, ...) (void)std::snprintf(nullptr, 0, __VA_ARGS__)

namespace beluga_amcl {

constexpr std::string_view kDifferentialModelName = "differential_drive";
constexpr std::string_view kOmnidirectionalModelName = "omnidirectional_drive";
constexpr std::string_view kStationaryModelName = "stationary";
constexpr std::string_view kNav2DifferentialModelName = "nav2_amcl::DifferentialMotionModel";
constexpr std::string_view kNav2OmnidirectionalModelName = "nav2_amcl::OmniMotionModel";
constexpr std::string_view kLikelihoodFieldModelName = "likelihood_field";
constexpr std::string_view kLikelihoodFieldProbModelName = "likelihood_field_prob";
constexpr std::string_view kBeamSensorModelName = "beam";

struct Parameter {
  std::string text;
  double number{0};
  [[nodiscard]] double as_double() const { return number; }
  [[nodiscard]] long as_int() const { return static_cast<long>(number); }
  [[nodiscard]] bool as_bool() const { return number != 0; }
  [[nodiscard]] std::string as_string() const { return text; }
};

class AmclNode {
 public:
  std::map<std::string, Parameter> parameters;
  std::unique_ptr<beluga_ros::Amcl> particle_filter_;
  bool enable_tf_broadcast_{false};
  nav_msgs::msg::OccupancyGrid last_likelihood_field_message;
  geometry_msgs::msg::PoseArray last_particle_cloud_message;

  [[nodiscard]] const Parameter& get_parameter(const std::string& name) const { return parameters.at(name); }
  [[nodiscard]] int get_logger() const { return 0; }
  [[nodiscard]] int now() const { return 0; }
  [[nodiscard]] auto get_execution_policy() const -> beluga_ros::Amcl::execution_policy_variant {
    if (get_parameter("execution_policy").as_string() == "par") return std::execution::par;
    return std::execution::seq;
  }

  // amcl_node.cpp:350-373
  auto get_motion_model(std::string_view name) const -> beluga_ros::Amcl::motion_model_variant {
    if (name == kDifferentialModelName || name == kNav2DifferentialModelName) {
      auto params = beluga::DifferentialDriveModelParam{};
      params.rotation_noise_from_rotation = get_parameter("alpha1").as_double();
      params.rotation_noise_from_translation = get_parameter("alpha2").as_double();
      params.translation_noise_from_translation = get_parameter("alpha3").as_double();
      params.translation_noise_from_rotation = get_parameter("alpha4").as_double();
      return beluga::DifferentialDriveModel{params};
    }
    if (name == kOmnidirectionalModelName || name == kNav2OmnidirectionalModelName) {
      auto params = beluga::OmnidirectionalDriveModelParam{};
      params.rotation_noise_from_rotation = get_parameter("alpha1").as_double();
      params.rotation_noise_from_translation = get_parameter("alpha2").as_double();
      params.translation_noise_from_translation = get_parameter("alpha3").as_double();
      params.translation_noise_from_rotation = get_parameter("alpha4").as_double();
      params.strafe_noise_from_translation = get_parameter("alpha5").as_double();
      return beluga::OmnidirectionalDriveModel{params};
    }
    if (name == kStationaryModelName) {
      return beluga::StationaryModel{};
    }
    throw std::invalid_argument(std::string("Invalid motion model: ") + std::string(name));
  }

  // amcl_node.cpp:375-408
  auto get_sensor_model(std::string_view name, nav_msgs::msg::OccupancyGrid::SharedPtr map) const
      -> beluga_ros::Amcl::sensor_model_variant {
    if (name == kLikelihoodFieldModelName) {
      auto params = beluga::LikelihoodFieldModelParam{};
      params.max_obstacle_distance = get_parameter("laser_likelihood_max_dist").as_double();
      params.max_laser_distance = get_parameter("laser_max_range").as_double();
      params.z_hit = get_parameter("z_hit").as_double();
      params.z_random = get_parameter("z_rand").as_double();
      params.sigma_hit = get_parameter("sigma_hit").as_double();
      params.model_unknown_space = get_parameter("model_unknown_space").as_bool();
      params.only_obstacle_boundaries = get_parameter("only_obstacle_boundaries").as_bool();
      return beluga::LikelihoodFieldModel{params, beluga_ros::OccupancyGrid{map}};
    }
    if (name == kLikelihoodFieldProbModelName) {
      auto params = beluga::LikelihoodFieldProbModelParam{};
      params.max_obstacle_distance = get_parameter("laser_likelihood_max_dist").as_double();
      params.max_laser_distance = get_parameter("laser_max_range").as_double();
      params.z_hit = get_parameter("z_hit").as_double();
      params.z_random = get_parameter("z_rand").as_double();
      params.sigma_hit = get_parameter("sigma_hit").as_double();
      return beluga::LikelihoodFieldProbModel{params, beluga_ros::OccupancyGrid{map}};
    }
    if (name == kBeamSensorModelName) {
      auto params = beluga::BeamModelParam{};
      params.z_hit = get_parameter("z_hit").as_double();
      params.z_short = get_parameter("z_short").as_double();
      params.z_max = get_parameter("z_max").as_double();
      params.z_rand = get_parameter("z_rand").as_double();
      params.sigma_hit = get_parameter("sigma_hit").as_double();
      params.lambda_short = get_parameter("lambda_short").as_double();
      params.beam_max_range = get_parameter("laser_max_range").as_double();
      return beluga::BeamSensorModel{params, beluga_ros::OccupancyGrid{map}};
    }
    throw std::invalid_argument(std::string("Invalid sensor model: ") + std::string(name));
  }

  // amcl_node.cpp:410-433
  auto make_particle_filter(nav_msgs::msg::OccupancyGrid::SharedPtr map) const -> std::unique_ptr<beluga_ros::Amcl> {
    auto params = beluga_ros::AmclParams{};
    params.update_min_d = get_parameter("update_min_d").as_double();
    params.update_min_a = get_parameter("update_min_a").as_double();
    params.resample_interval = static_cast<std::size_t>(get_parameter("resample_interval").as_int());
    params.selective_resampling = get_parameter("selective_resampling").as_bool();
    params.min_particles = static_cast<std::size_t>(get_parameter("min_particles").as_int());
    params.max_particles = static_cast<std::size_t>(get_parameter("max_particles").as_int());
    params.alpha_slow = get_parameter("recovery_alpha_slow").as_double();
    params.alpha_fast = get_parameter("recovery_alpha_fast").as_double();
    params.kld_epsilon = get_parameter("pf_err").as_double();
    params.kld_z = get_parameter("pf_z").as_double();
    params.spatial_resolution_x = get_parameter("spatial_resolution_x").as_double();
    params.spatial_resolution_y = get_parameter("spatial_resolution_y").as_double();
    params.spatial_resolution_theta = get_parameter("spatial_resolution_theta").as_double();

    return std::make_unique<beluga_ros::Amcl>(
        beluga_ros::OccupancyGrid{map},                                        //
        get_motion_model(get_parameter("robot_model_type").as_string()),       //
        get_sensor_model(get_parameter("laser_model_type").as_string(), map),  //
        params,                                                                //
        get_execution_policy());
  }

  // amcl_node.cpp:456-476 (filter construction / map update + the likelihood-field publishing block of map_callback)
  bool map_callback(nav_msgs::msg::OccupancyGrid::SharedPtr map) {
    bool publish = false;
    if (!particle_filter_) {
      try {
        RCLCPP_INFO(get_logger(), "Initializing particle filter instance");
        particle_filter_ = make_particle_filter(std::move(map));
        RCLCPP_INFO(get_logger(), "Particle filter initialization completed");
      } catch (const std::invalid_argument& error) {
        RCLCPP_ERROR(get_logger(), "Could not initialize particle filter: %s", error.what());
        return false;
      }
      if (get_parameter("debug").as_bool() && particle_filter_->has_likelihood_field()) {
        publish = true;
      }
    } else {
      particle_filter_->update_map(beluga_ros::OccupancyGrid{std::move(map)});
      publish = particle_filter_->has_likelihood_field();
    }

    if (publish) {
      auto message = nav_msgs::msg::OccupancyGrid{};
      beluga_ros::assign_likelihood_field(
          particle_filter_->likelihood_field(), particle_filter_->likelihood_field_origin(), message);
      beluga_ros::stamp_message(get_parameter("global_frame_id").as_string(), now(), message);
      last_likelihood_field_message = std::move(message);
    }
    return true;
  }

  // amcl_node.cpp:500-514
  void do_periodic_timer_callback() {
    if (!particle_filter_) {
      return;
    }
    {
      auto message = geometry_msgs::msg::PoseArray{};
      beluga_ros::assign_particle_cloud(particle_filter_->particles(), message);
      beluga_ros::stamp_message(get_parameter("global_frame_id").as_string(), now(), message);
      last_particle_cloud_message = std::move(message);
    }
  }

  // amcl_node.cpp:683-706
  bool initialize_from_estimate(const std::pair<Sophus::SE2d, Eigen::Matrix3d>& estimate) {
    RCLCPP_INFO(get_logger(), "Initializing particles from estimated pose and covariance");

    if (!particle_filter_) {
      RCLCPP_ERROR(get_logger(), "Could not initialize particles: The particle filter has not been initialized");
      return false;
    }

    const auto& [pose, covariance] = estimate;

    try {
      particle_filter_->initialize(pose, covariance);
    } catch (const std::runtime_error& error) {
      RCLCPP_ERROR(get_logger(), "Could not initialize particles: %s", error.what());
      return false;
    }

    enable_tf_broadcast_ = true;

    RCLCPP_INFO(
        get_logger(), "Particle filter initialized with %ld particles about initial pose x=%g, y=%g, yaw=%g",
        static_cast<long>(particle_filter_->particles().size()), pose.translation().x(), pose.translation().y(), pose.so2().log());

    return true;
  }

  // amcl_node.cpp:708-721
  bool initialize_from_map() {
    RCLCPP_INFO(get_logger(), "Initializing particles from map");

    if (!particle_filter_) {
      RCLCPP_ERROR(get_logger(), "Could not initialize particles: The particle filter has not been initialized");
      return false;
    }

    particle_filter_->initialize_from_map();
    enable_tf_broadcast_ = true;

    RCLCPP_INFO(
        get_logger(), "Particle filter initialized with %ld particles distributed across the map",
        static_cast<long>(particle_filter_->particles().size()));

    return true;
  }

  // amcl_node.cpp:603 (sensor_callback): one call for either measurement type
  template <class Measurement>
  auto update(const Sophus::SE2d& base_pose_in_odom, const Measurement& measurement) {
    return particle_filter_->update(base_pose_in_odom, measurement);
  }
};

}  // namespace beluga_amcl

int main(int argc, char** argv) {
  using beluga_amcl::Parameter;
  beluga_amcl::AmclNode node;
  const std::string sensor = argc > 1 ? argv[1] : "likelihood_field";
  for (const auto& [name, value] : std::map<std::string, double>{
           {"update_min_d", 0.25}, {"update_min_a", 0.2}, {"resample_interval", 1}, {"selective_resampling", 0}, {"min_particles", 500},
           {"max_particles", 2000}, {"recovery_alpha_slow", 0.001}, {"recovery_alpha_fast", 0.1}, {"pf_err", 0.05}, {"pf_z", 3.0},
           {"spatial_resolution_x", 0.5}, {"spatial_resolution_y", 0.5}, {"spatial_resolution_theta", 0.1745}, {"alpha1", 0.1},
           {"alpha2", 0.05}, {"alpha3", 0.1}, {"alpha4", 0.05}, {"alpha5", 0.1}, {"laser_likelihood_max_dist", 2.0},
           {"laser_max_range", 10.0}, {"z_hit", 0.5}, {"z_rand", 0.5}, {"z_short", 0.05}, {"z_max", 0.05}, {"sigma_hit", 0.2},
           {"lambda_short", 0.1}, {"model_unknown_space", 1}, {"only_obstacle_boundaries", 0}, {"debug", 1}})
    node.parameters[name] = Parameter{"", value};
  node.parameters["robot_model_type"] = Parameter{"differential_drive", 0};
  node.parameters["laser_model_type"] = Parameter{sensor, 0};
  node.parameters["execution_policy"] = Parameter{"seq", 0};
  node.parameters["global_frame_id"] = Parameter{"map", 0};

  auto map = std::make_shared<nav_msgs::msg::OccupancyGrid>();
  map->info.width = 96;
  map->info.height = 80;
  map->info.resolution = 0.1F;
  map->info.origin.position[0] = -2.0;
  map->info.origin.position[1] = -1.0;
  map->data.assign(96 * 80, 0);
  for (unsigned x = 0; x < 96; ++x) map->data[60 * 96 + x] = 100;
  for (unsigned y = 0; y < 80; ++y) map->data[y * 96 + 70] = 100;
  for (unsigned x = 0; x < 10; ++x) map->data[5 * 96 + x] = -1;

  try {
    if (!node.map_callback(map)) return 4;
    std::printf("has_likelihood_field %d\n", node.particle_filter_->has_likelihood_field() ? 1 : 0);
    if (sensor == "beam") {
      try {
        (void)node.particle_filter_->likelihood_field_origin();
        std::printf("beam_origin accepted\n");
        return 2;
      } catch (const std::runtime_error& e) {
        std::printf("beam_origin %s\n", e.what());
      }
    } else {
      std::printf("field_message %u %u %zu\n", node.last_likelihood_field_message.info.width,
                  node.last_likelihood_field_message.info.height, node.last_likelihood_field_message.data.size());
      const auto origin = node.particle_filter_->likelihood_field_origin();
      std::printf("field_origin %.17g %.17g\n", origin.x, origin.y);
    }
    node.initialize_from_map();
    node.do_periodic_timer_callback();
    std::printf("from_map_particles %zu %zu\n", node.particle_filter_->particles().size(), node.last_particle_cloud_message.poses.size());
    double wsum = 0;
    for (const double w : beluga_amd::views::weights(node.particle_filter_->particles())) wsum += w;
    std::printf("from_map_weight_sum %.17g\n", wsum);
    Eigen::Matrix3d covariance;
    covariance(0, 0) = covariance(1, 1) = 0.04;
    covariance(2, 2) = 0.01;
    node.initialize_from_estimate(std::make_pair(Sophus::SE2d{0.2, 1.0, 1.5}, covariance));
    Eigen::Matrix3d bad = covariance;
    bad(0, 1) = 2.0;
    std::printf("bad_covariance_rejected %d\n", node.initialize_from_estimate(std::make_pair(Sophus::SE2d{0.2, 1.0, 1.5}, bad)) ? 0 : 1);

    beluga_ros::LaserScan scan;
    for (int b = 0; b < 60; ++b) scan.points.push_back(Eigen::Vector2d{{2.0 * std::cos(-1.0 + b / 30.0), 2.0 * std::sin(-1.0 + b / 30.0)}});
    const auto first = node.update(Sophus::SE2d{0.0, 0.0, 0.0}, scan);
    std::printf("scan_update %d\n", first.has_value() ? 1 : 0);
    beluga_ros::SparsePointCloud3f cloud;
    for (int b = 0; b < 60; ++b)
      cloud.cloud.push_back(Eigen::Vector3f{{static_cast<float>(2.0 * std::cos(-1.0 + b / 30.0)), static_cast<float>(2.0 * std::sin(-1.0 + b / 30.0)), 0.3F}});
    const auto second = node.update(Sophus::SE2d{0.05, 0.3, 0.0}, cloud);
    std::printf("cloud_update %d\n", second.has_value() ? 1 : 0);
    if (second) std::printf("estimate %.6f %.6f\n", second->first.x, second->first.y);
    node.map_callback(map);  // a second map: update_map + republish
    std::printf("after_update_map %zu\n", node.particle_filter_->particles().size());
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error %s\n", e.what());
    return 3;
  }
  return 0;
}
