/* beluga_mcl.h — C ABI of libbeluga_mcl.so: the MI355X-native MCL particle-filter update.
 *
 * This is the drop-in boundary for ONE hot path of Ekumen-OS/beluga: `beluga::Amcl::update`
 * (beluga/include/beluga/algorithm/amcl_core.hpp:165-201) with DifferentialDriveModel +
 * LikelihoodFieldModel | BeamSensorModel, multinomial / KLD-adaptive resampling and the SE2
 * estimate.  The reference has no FFI for this path (it is header-only C++17 templates); these
 * entry points are what a binding for the path binds: one per reference call named below.
 * Plain pointers and sizes only, no C++/torch types, no exceptions across the boundary.
 *
 * Conventions
 *   - An SE2 pose is 4 doubles (cos, sin, x, y) = Sophus::SE2d::data() order, the layout of the
 *     reference's particle states (beluga/containers/tuple_vector.hpp:198 over Sophus::SE2d).
 *   - Particle sets cross the boundary as `states[n*4]` + `weights[n]` (host memory, caller owned).
 *     On the device the poses live as the same 4-double records (plus a weight array), owned by the context.
 *   - A measurement is `points_xy[B*2]` doubles: lidar hits in the robot base frame
 *     (`std::vector<std::pair<double,double>>`, sensor/likelihood_field_model.hpp:48).
 *   - Every call returns mcl_status (0 = OK, <0 = error; mcl_last_error() has the text).  Calls on
 *     one context must be serialised by the caller (the reference filter is not thread-safe either:
 *     beluga_amcl/src/ros2_common.cpp:407-409 uses one mutually exclusive callback group).
 *   - All work is enqueued on the context's HIP stream; calls that return host data synchronise it.
 *   - There is NO CPU fallback: mcl_create fails with MCL_ERR_NO_DEVICE when no gfx950 GPU is usable.
 */
#ifndef BELUGA_MCL_H
#define BELUGA_MCL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcl_ctx mcl_ctx;
typedef int32_t mcl_status;

enum {
  MCL_OK = 0,
  MCL_ERR_INVALID_ARGUMENT = -1,
  MCL_ERR_HIP = -2,
  MCL_ERR_OUT_OF_MEMORY = -3,
  MCL_ERR_NOT_READY = -4,      /* no map / no particles yet */
  MCL_ERR_BAD_COVARIANCE = -5, /* std::runtime_error of multivariate_normal_distribution.hpp:114-124 */
  MCL_ERR_NO_DEVICE = -6,
  MCL_ERR_UNSUPPORTED = -7     /* std::runtime_error "The current sensor model does not support likelihood field" (beluga_ros/amcl.hpp:154,174) */
};

/* LikelihoodFieldModel (sensor/likelihood_field_model.hpp), BeamSensorModel (sensor/beam_model.hpp),
 * LikelihoodFieldProbModel (sensor/likelihood_field_prob_model.hpp): beluga_ros::Amcl's sensor variants. */
enum { MCL_SENSOR_LIKELIHOOD_FIELD = 0, MCL_SENSOR_BEAM = 1, MCL_SENSOR_LIKELIHOOD_FIELD_PROB = 2 };
/* DifferentialDriveModel, OmnidirectionalDriveModel (motion/omnidirectional_drive_model.hpp:102-146),
 * StationaryModel (motion/stationary_model.hpp:55-61): beluga_ros::Amcl's motion variants. */
enum { MCL_MOTION_DIFFERENTIAL = 0, MCL_MOTION_OMNIDIRECTIONAL = 1, MCL_MOTION_STATIONARY = 2 };

/* beluga::AmclParams (algorithm/amcl_core.hpp:34-55) + the spatial-hash resolutions that
 * beluga_ros::AmclParams adds (beluga_ros/include/beluga_ros/amcl.hpp:90-97). Same defaults. */
typedef struct mcl_amcl_params {
  double update_min_d;          /* 0.25 */
  double update_min_a;          /* 0.2 */
  uint64_t resample_interval;   /* 1 */
  int32_t selective_resampling; /* 0 */
  int32_t reserved0;
  uint64_t min_particles; /* 500 */
  uint64_t max_particles; /* 2000 */
  double alpha_slow;      /* 0.001 */
  double alpha_fast;      /* 0.1 */
  double kld_epsilon;     /* 0.05 */
  double kld_z;           /* 3.0 */
  double spatial_resolution_x;     /* 0.5 */
  double spatial_resolution_y;     /* 0.5 */
  double spatial_resolution_theta; /* 10 deg in rad */
} mcl_amcl_params;

/* beluga::DifferentialDriveModelParam (motion/differential_drive_model.hpp:40-68). */
typedef struct mcl_diffdrive_params {
  double rotation_noise_from_rotation;       /* alpha1 */
  double rotation_noise_from_translation;    /* alpha2 */
  double translation_noise_from_translation; /* alpha3 */
  double translation_noise_from_rotation;    /* alpha4 */
  double distance_threshold;                 /* 0.01 */
} mcl_diffdrive_params;

/* beluga::LikelihoodFieldModelBaseParam (sensor/likelihood_field_model_base.hpp:42-64). */
typedef struct mcl_lf_params {
  double max_obstacle_distance; /* 100.0 */
  double max_laser_distance;    /* 2.0 */
  double z_hit;                 /* 0.5 */
  double z_random;              /* 0.5 */
  double sigma_hit;             /* 0.2 */
  int32_t model_unknown_space;      /* 0 */
  int32_t only_obstacle_boundaries; /* 0 */
} mcl_lf_params;

/* beluga::BeamModelParam (sensor/beam_model.hpp:43-58). */
typedef struct mcl_beam_params {
  double z_hit, z_short, z_max, z_rand, sigma_hit, lambda_short, beam_max_range;
} mcl_beam_params;

typedef struct mcl_config {
  int32_t device_id;   /* HIP device ordinal */
  int32_t sensor_kind; /* MCL_SENSOR_* */
  uint64_t seed;       /* key of the counter-based Philox4x32-10 stream (DESIGN.md "RNG stream") */
  mcl_amcl_params amcl;
  mcl_diffdrive_params motion;
  mcl_lf_params lf;
  mcl_beam_params beam;
  /* Particle sharding across processes (one context per GPU).  A single-GPU filter uses
   * shard_offset = 0, shard_count = max_particles.  Random draws are addressed by GLOBAL index. */
  uint64_t shard_offset;
  uint64_t shard_capacity; /* 0 => amcl.max_particles */
  void* hip_stream;        /* optional external hipStream_t (e.g. torch's current stream); NULL => own stream */
  int32_t motion_kind;     /* MCL_MOTION_*; the four alphas in `motion` are shared by the differential and omni models */
  int32_t reserved1;
  double strafe_noise_from_translation; /* OmnidirectionalDriveModelParam alpha5 (omnidirectional_drive_model.hpp:64-70) */
} mcl_config;

/* Fills `cfg` with the reference defaults listed above. */
void mcl_default_config(mcl_config* cfg);

/* beluga::Amcl ctor (amcl_core.hpp:105-124). */
mcl_status mcl_create(const mcl_config* cfg, mcl_ctx** out);
void mcl_destroy(mcl_ctx* ctx);
const char* mcl_last_error(const mcl_ctx* ctx); /* ctx may be NULL: error of the last failed mcl_create */

/* Sensor-model ctor / Amcl::update_map (amcl_core.hpp:150; likelihood_field_model_base.hpp:96-99,113-116,
 * 130-185; beam_model.hpp:152-156).  `cells` is row-major H x W int8 with value traits
 * {free, unknown, occupied} (beluga_ros/occupancy_grid.hpp:48-64 uses {0,-1,100}).
 * Builds the likelihood field with the reference's algorithm and uploads grid, field and the
 * free-cell list used by the random-state generator (random/multivariate_uniform_distribution.hpp:126-161). */
mcl_status mcl_set_map(mcl_ctx* ctx, const int8_t* cells, uint32_t width, uint32_t height, double resolution,
                       const double origin[4], const int8_t value_traits[3]);
/* Extension beside mcl_set_map (the reference has no such call: Amcl::update_map, amcl_core.hpp:150, builds the new sensor model on the
 * caller's thread, which at 16 M cells is 1 - 3 s of priority-queue wavefront, distance_map.hpp:55-98, during which no update runs):
 * the arguments are copied, the likelihood field of the new grid is built on a WORKER thread - the same host wavefront, the same bits -
 * while the filter keeps running on the map it has, and the swap (uploads, derived tables: a few milliseconds) happens at the start of
 * the first mcl_update after the build is done, or in mcl_map_commit.  A second call, or mcl_set_map, replaces a pending one.
 * mcl_map_pending: *state = 0 nothing pending, 1 building, 2 built and waiting for its swap.
 * mcl_map_commit: swaps now if the build is done (wait = 0: otherwise returns with nothing changed; wait = 1: waits for it first).
 * Not on a filter with a communicator of several ranks (MCL_ERR_UNSUPPORTED): the ranks would swap in different cycles. */
mcl_status mcl_set_map_async(mcl_ctx* ctx, const int8_t* cells, uint32_t width, uint32_t height, double resolution,
                             const double origin[4], const int8_t value_traits[3]);
mcl_status mcl_map_pending(mcl_ctx* ctx, int32_t* state);
mcl_status mcl_map_commit(mcl_ctx* ctx, int32_t wait);
/* LikelihoodFieldModelBase::likelihood_field() (likelihood_field_model_base.hpp:102). out: H*W floats. */
mcl_status mcl_get_likelihood_field(mcl_ctx* ctx, float* out);
/* beluga_ros::Amcl::has_likelihood_field() (beluga_ros/include/beluga_ros/amcl.hpp:181-188): 1 for the two likelihood-field
 * models, 0 for the beam model. */
mcl_status mcl_has_likelihood_field(const mcl_ctx* ctx, int32_t* has);
/* beluga_ros::Amcl::likelihood_field_origin() (amcl.hpp:161-178; likelihood_field_model_base.hpp:105): the field's origin
 * in world coordinates as (cos, sin, x, y).  MCL_ERR_UNSUPPORTED for the beam model (the reference throws). */
mcl_status mcl_get_likelihood_field_origin(mcl_ctx* ctx, double origin[4]);
/* Replace the device field with a caller-built one (same W,H as the current map). */
mcl_status mcl_set_likelihood_field(mcl_ctx* ctx, const float* field);

/* Amcl::initialize(pose, covariance) (amcl_core.hpp:145-147): max_particles samples of
 * N(mean_xytheta, cov[3x3 row-major]) with weight 1; sets force_update. */
mcl_status mcl_initialize_normal(mcl_ctx* ctx, const double mean_xytheta[3], const double cov[9]);
/* beluga_ros::Amcl::initialize_from_map() (beluga_ros/include/beluga_ros/amcl.hpp:209, called at
 * beluga_amcl/src/amcl_node.cpp:716): max_particles draws of MultivariateUniformDistribution<SE2d, OccupancyGrid>
 * (random/multivariate_uniform_distribution.hpp:126-161: the centre of a uniformly chosen free cell in the world frame,
 * uniform heading), weight 1; sets force_update.  Drawn on the device from the counter-based stream (step 0). */
mcl_status mcl_initialize_from_map(mcl_ctx* ctx);
/* Amcl::initialize(distribution) with caller-drawn states (amcl_core.hpp:131-137); n <= capacity. */
mcl_status mcl_set_particles(mcl_ctx* ctx, const double* states, const double* weights, uint64_t n);
/* Amcl::particles() (amcl_core.hpp:127). */
mcl_status mcl_num_particles(mcl_ctx* ctx, uint64_t* n);
mcl_status mcl_get_particles(mcl_ctx* ctx, double* states, double* weights, uint64_t capacity, uint64_t* n);
/* Amcl::force_update() (amcl_core.hpp:204). */
mcl_status mcl_force_update(mcl_ctx* ctx);

typedef struct mcl_estimate {
  double pose[4];        /* (cos, sin, x, y) */
  double covariance[9];  /* row-major 3x3: xy block + circular variance at [8] */
} mcl_estimate;

typedef struct mcl_update_info {
  int32_t updated;    /* 0 <=> the reference returns std::nullopt (amcl_core.hpp:166-172) */
  int32_t resampled;  /* resample_policy_ fired (amcl_core.hpp:181) */
  uint64_t num_particles;          /* after the update */
  double weight_sum;               /* sum of weights before normalisation */
  double effective_sample_size;    /* -1 when the policy did not evaluate it */
  double random_state_probability; /* ThrunRecoveryProbabilityEstimator output (amcl_core.hpp:179) */
} mcl_update_info;

/* Amcl::update(control_action, measurement) (amcl_core.hpp:165-201): the whole cycle. */
mcl_status mcl_update(mcl_ctx* ctx, const double control_pose[4], const double* points_xy, uint64_t num_points,
                      mcl_estimate* estimate, mcl_update_info* info);

/* Caller-side scan preparation, i.e. what beluga_ros::Amcl::update(pose, LaserScan) does before the filter sees the
 * measurement (beluga_ros/src/amcl.cpp:54-63; beluga_ros/include/beluga_ros/laser_scan.hpp:46-100;
 * beluga/sensor/data/laser_scan.hpp:64-90; beluga/views/take_evenly.hpp:126-148): decimate to `max_beams` evenly
 * spaced beams, drop NaN and out-of-range readings, polar -> cartesian, move into the base frame with the laser origin
 * `origin_se3` = Sophus::SE3d::data() = (qx, qy, qz, qw, tx, ty, tz).  sensor_msgs/LaserScan fields are float32.
 * Pure host arithmetic (O(beams)); `points_xy` must hold 2 * min(n, max_beams) doubles. */
typedef struct mcl_laser_scan {
  const float* ranges;
  uint64_t num_ranges;
  float angle_min, angle_increment;
  float range_min, range_max; /* from the message */
  double origin_se3[7];
  uint64_t max_beams;         /* SIZE_MAX: all */
  double min_range, max_range; /* caller limits (laser_min_range / laser_max_range); combined with the message's */
} mcl_laser_scan;
mcl_status mcl_prepare_laser_scan(const mcl_laser_scan* scan, double* points_xy, uint64_t* num_points);
/* mcl_prepare_laser_scan followed by mcl_update: beluga_ros::Amcl::update(base_pose_in_odom, laser_scan). */
mcl_status mcl_update_laser_scan(mcl_ctx* ctx, const double control_pose[4], const mcl_laser_scan* scan, mcl_estimate* estimate,
                                 mcl_update_info* info);

/* beluga_ros::Amcl::update(base_pose_in_odom, SparsePointCloud3f) (beluga_ros/src/amcl.cpp:67-81): every point of the
 * cloud (float x, y, z in the sensor frame) is moved into the base frame with the sensor origin `origin_se3` =
 * Sophus::SE3d::data() = (qx, qy, qz, qw, tx, ty, tz) in double precision and projected onto z = 0. */
mcl_status mcl_project_point_cloud(const float* points_xyz, uint64_t num_points, const double origin_se3[7], double* points_xy);
mcl_status mcl_update_point_cloud(mcl_ctx* ctx, const double control_pose[4], const float* points_xyz, uint64_t num_points,
                                  const double origin_se3[7], mcl_estimate* estimate, mcl_update_info* info);

/* ---- Stage-level entry points (what update() composes; used by parity tests and by the
 * multi-GPU driver, which interleaves collectives between them). -------------------------------- */

/* actions::propagate with DifferentialDriveModel::operator() (actions/propagate.hpp:57-79;
 * motion/differential_drive_model.hpp:129-164). `step` is the cycle counter word of the RNG stream. */
mcl_status mcl_propagate(mcl_ctx* ctx, const double pose[4], const double previous_pose[4], uint32_t step);
/* actions::reweight with the configured sensor model (actions/reweight.hpp:53-60;
 * likelihood_field_model.hpp:68-91 or beam_model.hpp:104-150): w[i] *= model(state[i]). */
mcl_status mcl_reweight(mcl_ctx* ctx, const double* points_xy, uint64_t num_points);

typedef struct mcl_weight_stats {
  double sum;        /* sum of weights before normalisation (actions/normalize.hpp:70) */
  double norm_sum;   /* sum of the normalised weights (the Thrun estimator's total, Q1) */
  double norm_sumsq; /* sum of squared normalised weights: ESS = norm_sum^2 / norm_sumsq */
} mcl_weight_stats;
/* Local sums of this shard's weights, no normalisation (multi-GPU: all-reduce `sum` first). */
mcl_status mcl_weight_sum(mcl_ctx* ctx, double* sum);
/* actions::normalize (actions/normalize.hpp:54-85) by `factor` (NaN => this shard's own sum) and
 * the statistics the policies consume (effective_sample_size.hpp:46-59, thrun_..._estimator.hpp:79-80). */
mcl_status mcl_normalize(mcl_ctx* ctx, double factor, mcl_weight_stats* stats);

/* views::sample | random_intersperse | take_while_kld | actions::assign (amcl_core.hpp:188-196).
 * New weights are 1 (type_traits/particle_traits.hpp:105). */
mcl_status mcl_resample(mcl_ctx* ctx, double random_state_probability, uint32_t step, uint64_t* n_out);

/* Sufficient statistics of beluga::estimate (algorithm/estimation.hpp:436-475) for this shard:
 * {Sw, Sw2, Sw*c, Sw*s, Sw*dx, Sw*dy, Sw*dx*dx, Sw*dx*dy, Sw*dy*dy, pivot_x, pivot_y, 0} with
 * dx = x - pivot_x.  Sums over shards are additive when every shard uses the same pivot. */
mcl_status mcl_estimate_sums(mcl_ctx* ctx, const double pivot_xy[2], double sums[12]);
/* Finishes the estimate from (all-reduced) sums. Pure host arithmetic. */
mcl_status mcl_estimate_from_sums(const double sums[12], mcl_estimate* out);
/* Convenience: single-shard estimate (estimation.hpp:436-475). */
mcl_status mcl_estimate_pose(mcl_ctx* ctx, mcl_estimate* out);

/* beluga::cluster_based_estimate (algorithm/cluster_based_estimation.hpp:415-433) — what beluga_ros::Amcl::update
 * returns (beluga_ros/src/amcl.cpp:125): particles are grouped into clusters around local maxima of the cell-averaged
 * weight; the mean and covariance of the cluster with the highest total weight are returned, or the overall estimate
 * if no cluster has more than one particle.  ParticleClusterizerParam :243-259 defaults: 0.20 m, 0.524 rad, 0.90. */
typedef struct mcl_cluster_params {
  double linear_hash_resolution;
  double angular_hash_resolution;
  double weight_cap_percentile;
} mcl_cluster_params;
mcl_status mcl_cluster_based_estimate(mcl_ctx* ctx, const mcl_cluster_params* params /* NULL: defaults */, mcl_estimate* out);
/* Selects what mcl_update returns: 0 = beluga::estimate (beluga::Amcl), 1 = cluster_based_estimate (beluga_ros::Amcl). */
mcl_status mcl_set_estimate_kind(mcl_ctx* ctx, int32_t kind, const mcl_cluster_params* params /* NULL: defaults */);

/* beluga_ros::assign_particle_cloud(particles, size, PoseArray&) (beluga_ros/include/beluga_ros/particle_cloud.hpp:131-149):
 * `particles | views::sample | take_exactly(size)` — a weighted sample of `size` states of the current set, for
 * publication; the set itself is not modified.  states: size x 4 doubles (cos, sin, x, y), host memory.  The draws come
 * from the counter-based stream (seed; index, 0x80000000 | draw_id): pass a different draw_id per publication. */
mcl_status mcl_sample_particle_cloud(mcl_ctx* ctx, uint64_t size, uint32_t draw_id, double* states);

/* ---- Particle shards across the GPUs of one node (one context per GPU, one process or thread per context) -------------
 * The logical filter's particles are split into contiguous shards of the global index space (mcl_config.shard_offset /
 * shard_capacity); the map is replicated and every rank passes the same control action and scan to mcl_update.  With a
 * communicator attached, mcl_update runs the whole cycle over the sharded set: propagation and reweight are local; the weight
 * normaliser is the sum of the gathered shard sums; the shard totals of the normalised weights are gathered into the
 * intervals of the global CDF; every output slot's draw u_j * total (same counter-based stream as on one GPU) is routed to the
 * shard that owns it, which answers with the ancestor's state (views::sample | random_intersperse | actions::assign,
 * amcl_core.hpp:188-196); the estimate's nine sums are gathered and added in rank order.  Results do not depend on the
 * number of ranks beyond the rounding of these sums.  With min_particles < max_particles the resampling is KLD-adaptive over
 * the shards as well (views::take_while_kld over the GLOBAL candidate stream, take_while_kld.hpp:72-88,112-137): candidates are
 * drawn block by block through the same exchange, the spatial hashes of every block are all-gathered, every rank takes the same
 * cut, and the kept candidates are re-balanced into contiguous shards - each context's shard_capacity must hold its share of
 * max_particles; mcl_update_info.num_particles is the count over all shards.  (The same steps are available one by one through
 * the stage-level entry points below: beluga_amd/sharded.py drives them over torch.distributed.)
 *
 * The collectives go through a transport: RCCL over xGMI (mcl_comm_attach_rccl; librccl.so is loaded at run time, the
 * library has no link-time dependency on it), or caller-supplied functions (mcl_comm_attach: MPI, a shared-memory exchange
 * between the threads of one process, ...).  All buffers are DEVICE pointers; calls are made in the same order on all ranks
 * and must be complete (or stream-ordered on `hip_stream`) when they return. */
typedef struct mcl_transport {
  void* user;
  /* every rank contributes `bytes` bytes at d_send; d_recv receives world * bytes, in rank order */
  int32_t (*all_gather)(void* user, const void* d_send, void* d_recv, uint64_t bytes, void* hip_stream);
  /* this rank sends send_bytes[q] bytes to rank q (consecutive blocks of d_send, in rank order) and receives recv_bytes[q]
   * bytes from rank q (consecutive blocks of d_recv, in rank order) */
  int32_t (*all_to_all)(void* user, const void* d_send, const uint64_t* send_bytes, void* d_recv, const uint64_t* recv_bytes,
                        void* hip_stream);
} mcl_transport;
/* Attaches a communicator of `world` ranks to a context created with this rank's shard_offset / shard_capacity.  The transport
 * struct is copied; `user` must outlive the context.  world == 1 is allowed (the cycle then needs no exchange).
 * COLLECTIVE for world > 1: the attach all-gathers a word of the configuration that selects a cycle's collectives (particle bounds,
 * resampling policy, thresholds, recovery alphas, KLD parameters, models, seed, device_policy, estimate kind, shard_pad_permille) and blocks until every
 * rank has attached - the ranks attach CONCURRENTLY (one thread or process per rank; a host that attaches its ranks one after the
 * other from one thread through a rendezvous transport deadlocks) - and every rank fails alike on a mismatch.  On an attached
 * filter mcl_set_option("device_policy" | "shard_pad_permille", ...) and mcl_set_estimate_kind are collective in the same way: every rank calls them, concurrently,
 * with the same arguments, whatever its own previous value was; on a mismatch they return an error and leave the value unchanged. */
mcl_status mcl_comm_attach(mcl_ctx* ctx, uint32_t rank, uint32_t world, const mcl_transport* transport);
/* RCCL: rank 0 obtains an id (ncclGetUniqueId), hands its 128 bytes to the other ranks by any means, every rank attaches. */
mcl_status mcl_comm_unique_id(uint8_t id[128]);
mcl_status mcl_comm_attach_rccl(mcl_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t world);

/* ---- Device access for zero-copy interop (torch / RCCL hand-off) -------------------------------- */
typedef struct mcl_device_view {
  double* states;     /* n records of 4 doubles (cos, sin, x, y) */
  double* w;
  double* cdf;        /* inclusive scan of the normalised weights (valid after mcl_build_cdf) */
  uint64_t n;         /* live particles in this shard */
  uint64_t capacity;
  void* hip_stream;
} mcl_device_view;
mcl_status mcl_get_device_view(mcl_ctx* ctx, mcl_device_view* view);
mcl_status mcl_set_num_particles(mcl_ctx* ctx, uint64_t n);
/* Inclusive scan of the (normalised) weights into the cdf buffer; returns the shard total. */
mcl_status mcl_build_cdf(mcl_ctx* ctx, double* total);
/* Sharded views::sample | random_intersperse (views/sample.hpp:102,133-135; random_intersperse.hpp:90-115) for the
 * `count` output slots [first_slot, first_slot+count) of the GLOBAL particle index space:
 * d_targets[t] = u_j * total (a point of the global CDF, total = sum of all shards' weights), or NaN where the
 * slot receives an injected random state.  u_j comes from the same Philox stream as the single-GPU path. */
mcl_status mcl_resample_targets(mcl_ctx* ctx, uint32_t step, double random_state_probability, double total,
                                uint64_t first_slot, uint64_t count, double* d_targets);
/* Routing for the ancestor exchange (device pointers throughout, nothing synchronises):
 * groups the `count` targets of mcl_resample_targets by the shard that owns them. `d_ends[r]` / `d_offsets[r]` are the
 * inclusive end / exclusive start of shard r's interval of the global CDF (world <= 64).  Outputs: d_send_targets =
 * shard-local targets ordered by destination rank, d_order[k] = output slot answered by the k-th request,
 * d_counts[r] (int64) = requests for rank r.  NaN targets (injected slots) are routed to `self_rank`. */
mcl_status mcl_route_targets(mcl_ctx* ctx, const double* d_targets, uint64_t count, const double* d_ends, const double* d_offsets,
                             uint32_t world, uint32_t self_rank, double* d_send_targets, uint32_t* d_order, int64_t* d_counts);
/* The owning shard's side of the exchange: for each of the `m` requests t[j] (values in this shard's cdf range
 * [0, total]) the state of the first particle i with cdf[i] >= t[j] — std::discrete_distribution's lower_bound
 * (views/sample.hpp:133-135) — as one (x, y, cos, sin) record of 4 doubles. */
mcl_status mcl_serve_requests(mcl_ctx* ctx, const double* d_requests, uint64_t m, double* d_replies);
/* actions::assign (actions/assign.hpp:56-64) of the exchanged ancestors, replies in request order plus the d_order of
 * mcl_route_targets: slot d_order[k] takes reply k, or a random free-space state where its target is NaN; weights become
 * 1 (particle_traits.hpp:105); the shard then holds `count` particles. */
mcl_status mcl_commit_routed(mcl_ctx* ctx, uint32_t step, uint64_t first_slot, uint64_t count, const double* d_replies,
                             const uint32_t* d_order, const double* d_targets);
/* KLD-adaptive resampling across shards (views/take_while_kld.hpp:72-88,112-137 over the GLOBAL candidate stream).
 * The driver draws candidates block by block: each shard draws its slice of the block (mcl_resample_targets, routing,
 * mcl_serve_requests as above), then mcl_finish_candidates writes the slice's states (4 doubles cos, sin, x, y per
 * candidate, slot order) and spatial hashes (algorithm/spatial_hash.hpp:190-193) to caller buffers without touching the
 * live set.  Every shard is fed the hashes of the whole block in global candidate order (all-gather) through
 * mcl_kld_feed, which returns in *first_fail the global index of the first candidate failing kld_condition (it and
 * everything after it is dropped), or ~0 if the block passes; mcl_kld_begin starts a pass.  After the cut the driver
 * re-balances the kept candidates and installs each shard's slice with mcl_load_shard (weights 1,
 * particle_traits.hpp:105); shard_offset is the global index of its first particle (used by the RNG addressing). */
mcl_status mcl_finish_candidates(mcl_ctx* ctx, uint32_t step, uint64_t first_slot, uint64_t count, const double* d_replies,
                                 const uint32_t* d_order, const double* d_targets, double* d_states, uint64_t* d_hashes);
mcl_status mcl_kld_begin(mcl_ctx* ctx);
mcl_status mcl_kld_feed(mcl_ctx* ctx, const uint64_t* d_hashes, uint64_t count, uint64_t* first_fail);
mcl_status mcl_load_shard(mcl_ctx* ctx, const double* d_states, uint64_t n, uint64_t shard_offset);
/* Device-side variants for callers that keep scalars on the GPU (e.g. to feed RCCL collectives without a host
 * round trip).  All pointers are DEVICE pointers; nothing synchronises; work is enqueued on the context's stream. */
mcl_status mcl_weight_sum_device(mcl_ctx* ctx, double* d_sum);
/* d_factor: the (global) normalisation factor; d_stats[0] = sum of the new weights, d_stats[1] = sum of squares. */
mcl_status mcl_normalize_device(mcl_ctx* ctx, const double* d_factor, double* d_stats);
mcl_status mcl_build_cdf_device(mcl_ctx* ctx, double* d_total);
/* d_sums[0..8] = the nine sums of mcl_estimate_sums for the given pivot. */
mcl_status mcl_estimate_sums_device(mcl_ctx* ctx, const double pivot_xy[2], double* d_sums);
mcl_status mcl_sync(mcl_ctx* ctx);

/* ---- Measurement hooks (bench.py): HIP-event timing of each stage on the context's stream. ------ */
enum {
  MCL_STAGE_PROPAGATE = 0,
  MCL_STAGE_REWEIGHT = 1,
  MCL_STAGE_NORMALIZE = 2,
  MCL_STAGE_RESAMPLE = 3,
  MCL_STAGE_ESTIMATE = 4,
  MCL_STAGE_SENSOR_KERNEL = 5, /* the sensor-model kernel alone (inside MCL_STAGE_REWEIGHT) */
  MCL_NUM_STAGES = 6
};
/* on: 0 = off, 1 = the sensor kernel of every 4th cycle only, 2 = every stage of every cycle (each event record costs ~5 us
 * of stream time, so timed runs use 1). */
mcl_status mcl_profile_enable(mcl_ctx* ctx, int32_t on);
/* Accumulated milliseconds and launch counts per stage since the last reset. */
mcl_status mcl_profile_read(mcl_ctx* ctx, double ms[MCL_NUM_STAGES], uint64_t counts[MCL_NUM_STAGES], int32_t reset);

/* Beam model only: grid cells visited by the ray walks since the last reset (cells/s). */
mcl_status mcl_beam_cells_visited(mcl_ctx* ctx, uint64_t* cells, int32_t reset);

/* ---- Switches and hooks for A/B measurements and tests.  No option but field_build changes a result beyond the rounding of a
 * particle's sum over the scan: the likelihood-field kernels with a lane per particle add the beams as libstdc++'s
 * std::transform_reduce does (blocks of four; the reference's call, likelihood_field_model.hpp:76), the kernels with a wave per
 * particle (lf_variant 3, small sets, lf_dispersed) in a fixed tree - 1e-16 relative. ------------------------------------
 * Options (defaults in parentheses; BELUGA_MCL_<NAME> in the environment sets the default at mcl_create):
 *   lf_variant (2)  likelihood-field kernel family: 2 = spatially ordered lanes (small sets: see lf_small_particles), 1 (and 0) = a lane
 *                   per particle in index order over the f32 field (no ordering pass), 3 = wave per particle over the palette table
 *   lf_fast (-1)    FMA variant with exact fallback: nonzero = whenever its preconditions hold, 0 = never
 *   lf_table (0)    1 = force the 8-byte table instead of the palette
 *   lf_patch (1)    index table through per-workgroup LDS patches (dense sets): 1 = where the last launch found them useful
 *                   (a dispersed set - global localisation - has none, and is sent to the per-lane gathers, with a probe every
 *                   16th launch), 0 = never, 2 = always
 *   lf_loose_below (224)  LDS-patch kernel: a workgroup with fewer than this many 256ths of its beam groups fitting a patch
 *                   gathers every look-up (a gathered group inside a patched workgroup costs twice one of an all-gathering one)
 *   lf_dispersed (2)  a set reported as dispersed: 2 = the lanes of a wave over the beams of one pose, the poses in the position-major
 *                   order, the far-tile bitmap in LDS (where the bitmap takes at most 32 KB and the scan fits beside it, else as 0),
 *                   0 = the ordered-lanes gather kernel (a lane per particle; 15 % slower at 1M x 1080), 1 = a wave per particle, lanes
 *                   over the beams, no ordering pass (slower still).  2 and 1 add a pose's terms as a tree over 64 lane sums: the
 *                   weights of 0 up to rounding (1e-13 relative).  lf_far_beams_per_wave (0 = 32): poses per wave of 2.
 *   lf_far_tiles (1)  the per-lane gather kernel keeps a bitmap of the table's far tiles in LDS (8x8 cells uniformly at the
 *                   field's most common value: free space beyond max_obstacle_distance of anything) and skips the memory access
 *                   of a look-up into one: 1 = for sets reported as dispersed, 0 = never, 2 = whenever that kernel runs.  Same
 *                   values, same sums (1M x 1080 dispersed over the 4000^2 map: 3.7 -> 1.9 ms per launch)
 *   key_layout (-1)  spatial ordering key: -1 = position-major (y, x Morton-interleaved, heading last) for likelihood-field sets
 *                   reported as dispersed - the far-tile gather kernel hands each XCD one contiguous eighth of that order, so an
 *                   XCD's L2 serves the neighbourhood of one block of the map at a time - and heading-major otherwise; 0 / 1 force
 *   lf_small_particles (65536)  likelihood-field sets below this size: a wave per 1..16 particles, lanes over the beams, no
 *                   ordering pass (the measured crossover to the ordered kernels)
 *   beam_table (1)  beam model, ordered kernel: what depends on a beam's EXPECTED range alone (the range itself, the hit normaliser
 *                   with its two erf, the short-return normaliser with its exp) comes from a table over the squared cell distance
 *                   between the hit and the source, built on the device at mcl_set_map with the same expressions; 0 = per beam
 *   lf_weight_sums (1)  the whole cycle's normalisation factor is added up from the likelihood-field kernel's workgroup sums of the
 *                   new weights (fixed order: the spatial order is the sort by (key, index), the same in every run); 0 = a pass of
 *                   its own over the weights (k_chunk_sum)
 *   device_policy (1)  recovery estimator on the device when the cycle takes no host-side decision
 *   sort_min_particles (16384)  below this many particles the spatial ordering is skipped (likelihood-field models)
 *   beam_sort_min_particles (16384)  beam model: the ordered kernel from this size on; below it a wave per particle over the
 *                   whole-grid bit maps (both skip empty space by the block distance map)
 *   key_curve (1)   heading-major ordering key: 1 = along the Hilbert curve through the (heading, y, x) bins - any run of the
 *                   order is a compact, connected set of bins -, 0 = Morton order (round 2: a run that crosses a high-level boundary
 *                   of the Z curve is two pieces far apart, and its workgroup fits no LDS patch)
 *   key_warp (1)    heading-major ordering key over bins of equal MASS of a normal set (the key frame's +-4 sigma mapped through the
 *                   normal distribution function) instead of equal width: the ordering's 1024 first-pass buckets then hold about the
 *                   same number of particles; 0 = equal width (round 2)
 *   key_bits_xy (0) bits of the x and of the y bins of that key (the heading gets the other 20 - 2 b): 0 = chosen every cycle from
 *                   the cloud's spread and the scan's reach (4 .. 6), 4 / 5 / 6 = forced (round 2: 6)
 *   lf_queue (1)    LDS-patch kernel, launches with more blocks than the device keeps workgroups resident (three per CU): 1 = that many
 *                   workgroups, each taking blocks from a counter until none is left (an XCD that is ahead takes more:
 *                   2 - 4 % off the kernel at 1M particles), 0 = one workgroup per block.  Which workgroup computes a block changes
 *                   nothing in it.  lf_queue_grid (0): the number of resident workgroups, 0 = three per CU (tests: a few workgroups)
 *   shard_pad_permille (1063)  sharded fixed-size cycle (mcl_comm_attach): the ancestor exchange moves a FIXED number of entries per pair of
 *                   ranks - mean * permille / 1000 + 8 * sqrt(mean) + 64, mean = what a shard's output slots ask of one other shard on
 *                   average - so that no count has to be read by the host in the middle of the cycle (one host synchronisation per cycle);
 *                   a cycle whose requests do not fit runs the exchange again with exact counts (mcl_get_counter "comm_overflows").
 *                   0 = always exact counts (two host synchronisations per cycle).  Part of the configuration the ranks compare at
 *                   mcl_comm_attach; on an attached filter setting it is a COLLECTIVE call like device_policy (every rank, same value).
 *   lf_ends_first (1)  LDS-patch kernel: the blocks are taken from both ends of the spatial order inwards (0, N - 1, 1, N - 2, ...): the ends
 *                   are the cloud's fringe, whose blocks gather every look-up and take twice as long - taken first they are not the launch's
 *                   last; 0 = in order
 *   beam_sectors (1)  beam model, ordered kernel, scanners that reach beyond half the 1024-cell LDS window (448 .. 896 cells): the scan is taken
 *                   in four sectors, each with a window that holds its rays; 0 = one centred window, rays that leave it go on over the
 *                   whole-grid maps in global memory (same cells visited either way)
 *   beam_free_ahead (1)  beam model, ordered kernel: per workgroup and beam, the cells the middle particle's ray clearance proves free for
 *                   every lane are passed in one closed-form step before a lane's own walk (same cells visited, same weights)
 *   cycle_spin (-1)  fixed-size cycles (resample every cycle, mean / covariance estimate): 1 = the host waits for a completion word the
 *                   cycle's last kernel stores to mapped host memory instead of the stream's completion signal (the calling thread spins
 *                   for the length of the cycle); 0 = hipStreamSynchronize; -1 = the word for sets of 256K particles and more.
 *                   Measured: + 1 % at 1M particles, 4 us per cycle slower at 2000.  Never while stage profiling is enabled.
 *   scan_fused (1)  fixed-size cycle that resamples: normalisation, totals of the normalised weights, recovery estimator and CDF in ONE
 *                   launch (the chunk sums travel between the workgroups inside it): 1 = for sets of up to 64K particles, where the cycle
 *                   is bound by the host's launches; 2 = up to 2M particles (measured at 1M: no faster than the two launches); 0 = never.
 *                   Bit-identical.
 *   draw_fold (1)   the draw kernel's last workgroup to finish adds up the estimate sums instead of a launch of its own behind it: 1 = for
 *                   sets of up to 64K particles, 2 = up to 4M (measured at 1M: 2.6 us slower), 0 = never.  Bit-identical.
 *   lf_unit_weights (1)  LDS-patch kernel on a set whose weights are all 1.0 - fresh from a resampling or an initialisation
 *                   (particle_traits.hpp:105), which the library keeps track of -: the old weight is not loaded (1.0 x = x); 0 = always
 *                   loaded.  Bit-identical; 9 us of a 1M-particle cycle.
 *   noise_ahead (1)  fixed-size cycles, sets of 64K .. 2M particles: the NEXT cycle's propagation normals - a function of (seed, step, particle
 *                   index) alone - are drawn a cycle ahead: 1 = by the draw kernel (whose vector units wait for memory), 2 = by a kernel of its
 *                   own behind the cycle's last one, while the host is away (cycles that end on the completion word); 0 = by the propagation
 *                   kernel itself.  Bit-identical; counter noise_ahead_used.
 *   order_ahead (1)  fixed-size cycles that end on the completion word (cycle_spin), sets of 64K .. 2M particles: the draw kernel also leaves the
 *                   ordering key of where each particle will be after the NEXT propagation - with the control action of this cycle as the
 *                   prediction, first order, single precision -, the ordering passes run behind the cycle's last kernel while the host is
 *                   away, and the next cycle goes from its propagation straight into the reweight if the action it gets is close to the
 *                   predicted one (else it orders as before).  Only locality depends on the order.  0 = the ordering inside the cycle.
 *                   Counters order_ahead_used / order_ahead_missed.
 *   norm_store (0)  fixed-size cycle that resamples at once: 0 = the normalisation kernel does not store the normalised weights (nothing reads
 *                   them), the CDF kernel divides again; 1 = stored.  Bit-identical.
 *   small_fused (1)  sets of up to 4096 particles: everything behind the reweight - normalise, policies, fixed-size or KLD resampling, estimate
 *                   sums - in one launch of one workgroup and one host synchronisation (and two one-workgroup kernels around the host's pass of
 *                   cluster_based_estimate); 0 = the kernels of the large path
 *   lf_split (3)    LDS-patch planner: a group of 8 beams that fits no whole 64 x 64-cell patch (its end-points straddle a range
 *                   discontinuity: 3 - 5 % of the groups of an indoor scan, whatever the cloud) goes through two half patches -
 *                   beams [0, k) and [k, 8), 32 x 64 (bit 0) or 64 x 32 (bit 1) cells each, in the buffer of one whole patch; 0 = such
 *                   groups are gathered
 *   lf_margin (1)   LDS-patch planner, the rotation part of the bound on a workgroup's end-points: 1 = per axis
 *                   ((1 - cos d) |q'x| + |sin d| |q'y| in x, the transpose in y), 0 = |R_p - R_ref| |q| on both axes (round 2)
 *   field_build (0)  how the NEXT mcl_set_map builds the likelihood field: 0 = the reference's priority-queue wavefront on the
 *                    host (bit-identical field, seconds at 16 M cells), 1 = exact Euclidean distance transform on the device
 *                    (milliseconds; equal at all but the few cells where the wavefront does not find the nearest obstacle,
 *                    never farther from the truth; falls back to 0 when max_obstacle_distance spans more than 1024 cells).
 *                    This one DOES change the field where the two algorithms differ; everything downstream follows the field.
 * Counters: lf_beams_launches = launches of the wave-per-particle LF kernel (small sets, lf_dispersed, lf_variant 3);
 *   lf_far_launches = those of the gather kernels with the far-tile bitmap (lf_far_beams_launches of them: the lanes-over-beams form,
 *   lf_dispersed 2), lf_far_tiles = tiles in the bitmap (0 = none built);
 *   lf_fast_launches = launches of the FMA variant so far; lf_patch_launches = those of them sent to the LDS-patch
 *   kernel; lf_patch_groups_planned / lf_patch_groups_through = groups of 8 beams (per workgroup) that kernel has looked at /
 *   has read through a patch, running totals over a sample of the workgroups; lf_queue_launches = launches of the
 *   LDS-patch kernel with the queue of blocks; field_built_on_device, field_build_us = the last mcl_set_map;
 *   cluster_cells = occupied cells of the last cluster_based_estimate; comm_ranks_seen (ncclCommCount of the library's communicator, 0
 *   without one), comm_collectives, comm_bytes_out (running totals of this rank), comm_backend (0 = none, 1 = the caller's transport, 2 = RCCL inside the library). */
mcl_status mcl_set_option(mcl_ctx* ctx, const char* name, int64_t value);
mcl_status mcl_get_counter(mcl_ctx* ctx, const char* name, uint64_t* value);
/* Runs the spatial ordering on the current set and returns it: perm[t] = particle at position t, keys[i] = ordering key
 * of particle i (n entries each, host memory).  keys[perm[t]] is non-decreasing in t. */
mcl_status mcl_debug_order(mcl_ctx* ctx, uint32_t* perm, uint32_t* keys);

/* Test hook: sets the outputs of the two exponential filters of ThrunRecoveryProbabilityEstimator
 * (thrun_recovery_probability_estimator.hpp:69-89, exponential_filter.hpp:32-44) on the host and on the device.  With a fixed particle
 * count the estimator sees the average of NORMALISED weights - 1 / N in every cycle - and never leaves p = 0 by itself (amcl_core.hpp:177-179,
 * as in the reference); a test that wants a cycle with injected random states (random_intersperse over shards) puts the filters apart first.
 * On an attached filter every rank has to make the same call. */
mcl_status mcl_debug_set_recovery_filters(mcl_ctx* ctx, double slow, double fast);

/* Position of the cell (heading, y, x), `bits` bits each, along the 3-D Hilbert curve the heading-major ordering key follows
 * (kernels.h hilbert_index_3; pure host arithmetic, no device needed): consecutive positions are face neighbours. */
uint32_t mcl_debug_curve_index(uint32_t heading_bin, uint32_t y_bin, uint32_t x_bin, uint32_t bits /* per axis, 1 .. 6 */);

const char* mcl_version(void);
/* Nonzero for a measurement build of the kernels (tools/build_variant.sh: ablations compute nonsense by design, timing builds distort):
 * beluga_amd/capi.py refuses to load one as the product library unless BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1. */
int mcl_measurement_build(void);

#ifdef __cplusplus
}
#endif
#endif /* BELUGA_MCL_H */
