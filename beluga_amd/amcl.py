"""Host-side mirror of the reference's filter interface, backed by the HIP library.

`Amcl` has the surface of `beluga::Amcl` (beluga/include/beluga/algorithm/amcl_core.hpp:81-233) /
`beluga_ros::Amcl` (beluga_ros/include/beluga_ros/amcl.hpp:102-282): same constructor ingredients
(map, motion model params, sensor model params, AmclParams), `particles()`, `initialize(pose, covariance)`,
`update_map(map)`, `update(control_action, measurement)` returning `None` where the reference returns
`std::nullopt`, and `force_update()`.  Parameter classes carry the reference's field names and defaults.
All per-particle work happens in libbeluga_mcl.so on the GPU; this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

from . import capi


@dataclass
class AmclParams:
    """beluga::AmclParams (amcl_core.hpp:34-55) + beluga_ros spatial resolutions (beluga_ros/amcl.hpp:90-97)."""
    update_min_d: float = 0.25
    update_min_a: float = 0.2
    resample_interval: int = 1
    selective_resampling: bool = False
    min_particles: int = 500
    max_particles: int = 2000
    alpha_slow: float = 0.001
    alpha_fast: float = 0.1
    kld_epsilon: float = 0.05
    kld_z: float = 3.0
    spatial_resolution_x: float = 0.5
    spatial_resolution_y: float = 0.5
    spatial_resolution_theta: float = 10.0 * math.pi / 180.0


@dataclass
class DifferentialDriveModelParam:
    """motion/differential_drive_model.hpp:40-68."""
    rotation_noise_from_rotation: float
    rotation_noise_from_translation: float
    translation_noise_from_translation: float
    translation_noise_from_rotation: float
    distance_threshold: float = 0.01


@dataclass
class OmnidirectionalDriveModelParam:
    """motion/omnidirectional_drive_model.hpp:36-73."""
    rotation_noise_from_rotation: float
    rotation_noise_from_translation: float
    translation_noise_from_translation: float
    translation_noise_from_rotation: float
    strafe_noise_from_translation: float
    distance_threshold: float = 0.01


@dataclass
class StationaryModelParam:
    """motion/stationary_model.hpp:40-62 takes no parameters."""


@dataclass
class LikelihoodFieldModelParam:
    """sensor/likelihood_field_model_base.hpp:42-64."""
    max_obstacle_distance: float = 100.0
    max_laser_distance: float = 2.0
    z_hit: float = 0.5
    z_random: float = 0.5
    sigma_hit: float = 0.2
    model_unknown_space: bool = False
    only_obstacle_boundaries: bool = False


@dataclass
class LikelihoodFieldProbModelParam(LikelihoodFieldModelParam):
    """sensor/likelihood_field_prob_model.hpp:34 (= LikelihoodFieldModelBaseParam)."""


@dataclass
class BeamModelParam:
    """sensor/beam_model.hpp:43-58."""
    z_hit: float = 0.5
    z_short: float = 0.5
    z_max: float = 0.05
    z_rand: float = 0.05
    sigma_hit: float = 0.2
    lambda_short: float = 0.1
    beam_max_range: float = 60.0


@dataclass
class OccupancyGrid:
    """An OccupancyGrid2 (sensor/data/occupancy_grid.hpp:39-75): row-major int8 cells, resolution, origin, value traits."""
    cells: np.ndarray  # (H, W) int8
    resolution: float
    origin: Sequence[float] = (1.0, 0.0, 0.0, 0.0)  # SE2 as (cos, sin, x, y)
    value_traits: Tuple[int, int, int] = (0, -1, 100)  # free, unknown, occupied


def se2_from_xytheta(x: float, y: float, theta: float) -> np.ndarray:
    return np.array([math.cos(theta), math.sin(theta), x, y], dtype=np.float64)


def _dp(a: np.ndarray):
    return a.ctypes.data_as(capi.c_double_p)


class Amcl:
    def __init__(self, grid: OccupancyGrid, motion, sensor, params: AmclParams = AmclParams(), *,
                 seed: int = 0, device: int = 0, shard_offset: int = 0, shard_capacity: int = 0, hip_stream: int = 0,
                 options: Optional[dict] = None):
        """options: library switches applied before the map is installed (mcl_set_option), e.g. {"field_build": 1} to build
        the likelihood field with the device's exact distance transform instead of the reference's wavefront on the host."""
        self._lib = capi.load()
        cfg = capi.Config()
        self._lib.mcl_default_config(C.byref(cfg))
        cfg.device_id = device
        cfg.seed = seed
        for k in ("update_min_d", "update_min_a", "resample_interval", "min_particles", "max_particles", "alpha_slow",
                  "alpha_fast", "kld_epsilon", "kld_z", "spatial_resolution_x", "spatial_resolution_y", "spatial_resolution_theta"):
            setattr(cfg.amcl, k, getattr(params, k))
        cfg.amcl.selective_resampling = int(params.selective_resampling)
        if isinstance(motion, StationaryModelParam):
            cfg.motion_kind = capi.MCL_MOTION_STATIONARY
        else:
            for k in ("rotation_noise_from_rotation", "rotation_noise_from_translation", "translation_noise_from_translation",
                      "translation_noise_from_rotation", "distance_threshold"):
                setattr(cfg.motion, k, getattr(motion, k))
            if isinstance(motion, OmnidirectionalDriveModelParam):
                cfg.motion_kind = capi.MCL_MOTION_OMNIDIRECTIONAL
                cfg.strafe_noise_from_translation = motion.strafe_noise_from_translation
        if isinstance(sensor, LikelihoodFieldModelParam):
            cfg.sensor_kind = (capi.MCL_SENSOR_LIKELIHOOD_FIELD_PROB if isinstance(sensor, LikelihoodFieldProbModelParam)
                               else capi.MCL_SENSOR_LIKELIHOOD_FIELD)
            for k in ("max_obstacle_distance", "max_laser_distance", "z_hit", "z_random", "sigma_hit"):
                setattr(cfg.lf, k, getattr(sensor, k))
            cfg.lf.model_unknown_space = int(sensor.model_unknown_space)
            cfg.lf.only_obstacle_boundaries = int(sensor.only_obstacle_boundaries)
        elif isinstance(sensor, BeamModelParam):
            cfg.sensor_kind = capi.MCL_SENSOR_BEAM
            for k in ("z_hit", "z_short", "z_max", "z_rand", "sigma_hit", "lambda_short", "beam_max_range"):
                setattr(cfg.beam, k, getattr(sensor, k))
        else:
            raise ValueError("sensor must be LikelihoodFieldModelParam or BeamModelParam")
        cfg.shard_offset = shard_offset
        cfg.shard_capacity = shard_capacity
        cfg.hip_stream = hip_stream or None
        self.params = params
        self._cfg = cfg
        self._ctx = capi._ctx()
        st = self._lib.mcl_create(C.byref(cfg), C.byref(self._ctx))
        if st != capi.MCL_OK:
            msg = self._lib.mcl_last_error(None).decode()
            self._ctx = None
            raise capi.MclError(st, msg)
        self._shape = None
        self._est, self._info = capi.Estimate(), capi.UpdateInfo()
        self._est_ref, self._info_ref = C.byref(self._est), C.byref(self._info)
        self._est_view = np.frombuffer(self._est, dtype=np.float64, count=13)  # pose[4] | covariance[9]
        self._have_info = False
        # a second handle of mcl_update that takes the arrays as raw addresses (no per-call pointer objects)
        self._update_fn = self._lib["mcl_update"]
        self._update_fn.restype = C.c_int32
        self._update_fn.argtypes = [capi._ctx, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        for name, value in (options or {}).items():
            self.set_option(name, value)
        self.update_map(grid)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.mcl_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != capi.MCL_OK:
            raise capi.MclError(st, self._lib.mcl_last_error(self._ctx).decode())

    # -- reference surface -------------------------------------------------------------------------
    def update_map(self, grid: OccupancyGrid):
        """Amcl::update_map (amcl_core.hpp:150)."""
        cells = np.ascontiguousarray(grid.cells, dtype=np.int8)
        H, W = cells.shape
        origin = np.ascontiguousarray(grid.origin, dtype=np.float64)
        traits = (C.c_int8 * 3)(*grid.value_traits)
        self._check(self._lib.mcl_set_map(self._ctx, cells.ctypes.data_as(capi.c_i8_p), W, H, float(grid.resolution), _dp(origin),
                                          traits))
        self._shape = (H, W)
        self._pending_shape = None  # (a map given now replaces one that was still on its way)

    def update_map_async(self, grid: OccupancyGrid):
        """Extension (mcl_set_map_async): the new map's likelihood field is built on a worker thread while the filter keeps running on
        the map it has; the swap happens at the start of the first update() after the build is done, or in map_commit()."""
        cells = np.ascontiguousarray(grid.cells, dtype=np.int8)
        H, W = cells.shape
        origin = np.ascontiguousarray(grid.origin, dtype=np.float64)
        traits = (C.c_int8 * 3)(*grid.value_traits)
        self._check(self._lib.mcl_set_map_async(self._ctx, cells.ctypes.data_as(capi.c_i8_p), W, H, float(grid.resolution), _dp(origin),
                                                traits))
        self._pending_shape = (H, W)

    def map_pending(self) -> int:
        """0: no map on its way, 1: its field is being built, 2: built, waiting for the swap."""
        v = C.c_int32(0)
        self._check(self._lib.mcl_map_pending(self._ctx, C.byref(v)))
        if v.value == 0 and getattr(self, "_pending_shape", None) is not None:
            self._shape, self._pending_shape = self._pending_shape, None
        return v.value

    def map_commit(self, wait: bool = True):
        """Swaps to the map given to update_map_async now (wait: for its build first; otherwise only if it is done)."""
        self._check(self._lib.mcl_map_commit(self._ctx, 1 if wait else 0))
        self.map_pending()

    def likelihood_field(self) -> np.ndarray:
        self.map_pending()  # (a swap inside update() may have changed the map's shape)
        """LikelihoodFieldModelBase::likelihood_field() (likelihood_field_model_base.hpp:102)."""
        out = np.zeros(self._shape, dtype=np.float32)
        self._check(self._lib.mcl_get_likelihood_field(self._ctx, out.ctypes.data_as(capi.c_float_p)))
        return out

    def set_likelihood_field(self, field: np.ndarray):
        field = np.ascontiguousarray(field, dtype=np.float32)
        assert field.shape == self._shape
        self._check(self._lib.mcl_set_likelihood_field(self._ctx, field.ctypes.data_as(capi.c_float_p)))

    def initialize(self, pose_xytheta, covariance):
        """Amcl::initialize(pose, covariance) (amcl_core.hpp:145-147). Raises RuntimeError on a bad covariance."""
        m = np.ascontiguousarray(pose_xytheta, dtype=np.float64)
        cv = np.ascontiguousarray(covariance, dtype=np.float64).reshape(9)
        st = self._lib.mcl_initialize_normal(self._ctx, _dp(m), _dp(cv))
        if st == capi.MCL_ERR_BAD_COVARIANCE:
            raise RuntimeError("Invalid covariance matrix")  # multivariate_normal_distribution.hpp:114-124
        self._check(st)

    def initialize_from_map(self):
        """beluga_ros::Amcl::initialize_from_map() (beluga_ros/include/beluga_ros/amcl.hpp:209): max_particles states drawn
        uniformly over the free cells of the map (random/multivariate_uniform_distribution.hpp:126-161)."""
        self._check(self._lib.mcl_initialize_from_map(self._ctx))

    def has_likelihood_field(self) -> bool:
        """beluga_ros::Amcl::has_likelihood_field() (amcl.hpp:181-188)."""
        v = C.c_int32(0)
        self._check(self._lib.mcl_has_likelihood_field(self._ctx, C.byref(v)))
        return bool(v.value)

    def likelihood_field_origin(self) -> np.ndarray:
        """beluga_ros::Amcl::likelihood_field_origin() (amcl.hpp:161-178) as (cos, sin, x, y); RuntimeError for the beam model."""
        out = np.zeros(4)
        st = self._lib.mcl_get_likelihood_field_origin(self._ctx, _dp(out))
        if st == capi.MCL_ERR_UNSUPPORTED:
            raise RuntimeError("The current sensor model does not support likelihood field")
        self._check(st)
        return out

    def update_point_cloud(self, control_action, points_xyz, origin_se3=(0, 0, 0, 1, 0, 0, 0)):
        """beluga_ros::Amcl::update(base_pose_in_odom, SparsePointCloud3f) (beluga_ros/src/amcl.cpp:67-81)."""
        ctrl = np.ascontiguousarray(control_action, dtype=np.float64)
        pts = np.ascontiguousarray(points_xyz, dtype=np.float32).reshape(-1, 3)
        origin = np.ascontiguousarray(origin_se3, dtype=np.float64)
        self._check(self._lib.mcl_update_point_cloud(self._ctx, _dp(ctrl), pts.ctypes.data_as(capi.c_float_p), len(pts), _dp(origin),
                                                     self._est_ref, self._info_ref))
        self._have_info = True
        if not self._info.updated:
            return None
        out = self._est_view.copy()
        return out[:4], out[4:13].reshape(3, 3)

    def set_particles(self, states, weights):
        """Amcl::initialize(distribution) with caller-drawn states (amcl_core.hpp:131-137)."""
        s = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 4)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        assert len(s) == len(w)
        self._check(self._lib.mcl_set_particles(self._ctx, _dp(s), _dp(w), len(w)))

    def num_particles(self) -> int:
        n = C.c_uint64(0)
        self._check(self._lib.mcl_num_particles(self._ctx, C.byref(n)))
        return n.value

    def particles(self):
        """Amcl::particles() (amcl_core.hpp:127): (states[n,4] as (cos,sin,x,y), weights[n])."""
        n = self.num_particles()
        s, w = np.zeros((n, 4)), np.zeros(n)
        got = C.c_uint64(0)
        self._check(self._lib.mcl_get_particles(self._ctx, _dp(s), _dp(w), n, C.byref(got)))
        return s, w

    def sample_particle_cloud(self, size: int, draw_id: int = 0) -> np.ndarray:
        """beluga_ros::assign_particle_cloud(particles, size, message) (particle_cloud.hpp:131-149): `size` states drawn
        with probability proportional to the weights, (size, 4) as (cos, sin, x, y); the set is not modified."""
        out = np.zeros((size, 4))
        if size:
            self._check(self._lib.mcl_sample_particle_cloud(self._ctx, size, draw_id, _dp(out)))
        return out if self.num_particles() else out[:0]

    def force_update(self):
        """Amcl::force_update() (amcl_core.hpp:204)."""
        self._check(self._lib.mcl_force_update(self._ctx))

    def update(self, control_action, measurement) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """Amcl::update (amcl_core.hpp:165-201). Returns (pose (cos,sin,x,y), covariance 3x3) or None."""
        # This wrapper sits inside the measured cycle: no per-call ctypes objects, no dict, raw addresses for the arrays.
        ctrl = control_action if (type(control_action) is np.ndarray and control_action.dtype == np.float64
                                  and control_action.flags.c_contiguous) else np.ascontiguousarray(control_action, dtype=np.float64)
        pts = measurement if (type(measurement) is np.ndarray and measurement.dtype == np.float64
                              and measurement.flags.c_contiguous) else np.ascontiguousarray(measurement, dtype=np.float64)
        status = self._update_fn(self._ctx, ctrl.ctypes.data, pts.ctypes.data, pts.size // 2, self._est_ref, self._info_ref)
        if status != 0:
            self._check(status)
        self._have_info = True
        if not self._info.updated:
            return None
        out = self._est_view.copy()
        return out[:4], out[4:13].reshape(3, 3)

    @property
    def last_info(self):
        """mcl_update_info of the last update as a dict (None before the first one)."""
        if not self._have_info:
            return None
        info = self._info
        return {
            "updated": bool(info.updated), "resampled": bool(info.resampled), "num_particles": info.num_particles,
            "weight_sum": info.weight_sum, "ess": info.effective_sample_size,
            "random_state_probability": info.random_state_probability,
        }

    def update_laser_scan(self, control_action, scan):
        """beluga_ros::Amcl::update(base_pose_in_odom, laser_scan) (beluga_ros/src/amcl.cpp:54-63)."""
        ctrl = np.ascontiguousarray(control_action, dtype=np.float64)
        self._check(self._lib.mcl_update_laser_scan(self._ctx, _dp(ctrl), C.byref(scan), self._est_ref, self._info_ref))
        self._have_info = True
        if not self._info.updated:
            return None
        out = self._est_view.copy()
        return out[:4], out[4:13].reshape(3, 3)

    # -- stage-level entry points (parity tests, multi-GPU driver) ------------------------------------
    def propagate(self, pose, previous_pose, step: int):
        a = np.ascontiguousarray(pose, dtype=np.float64)
        b = np.ascontiguousarray(previous_pose, dtype=np.float64)
        self._check(self._lib.mcl_propagate(self._ctx, _dp(a), _dp(b), step))

    def reweight(self, measurement):
        pts = np.ascontiguousarray(measurement, dtype=np.float64).reshape(-1, 2)
        self._check(self._lib.mcl_reweight(self._ctx, _dp(pts), len(pts)))

    def weight_sum(self) -> float:
        v = C.c_double(0)
        self._check(self._lib.mcl_weight_sum(self._ctx, C.byref(v)))
        return v.value

    def normalize(self, factor: float = float("nan")):
        st = capi.WeightStats()
        self._check(self._lib.mcl_normalize(self._ctx, factor, C.byref(st)))
        return {"sum": st.sum, "norm_sum": st.norm_sum, "norm_sumsq": st.norm_sumsq}

    def resample(self, random_state_probability: float, step: int) -> int:
        n = C.c_uint64(0)
        self._check(self._lib.mcl_resample(self._ctx, random_state_probability, step, C.byref(n)))
        return n.value

    def estimate_sums(self, pivot=(0.0, 0.0)) -> np.ndarray:
        p = np.ascontiguousarray(pivot, dtype=np.float64)
        out = np.zeros(12)
        self._check(self._lib.mcl_estimate_sums(self._ctx, _dp(p), _dp(out)))
        return out

    def estimate(self):
        est = capi.Estimate()
        self._check(self._lib.mcl_estimate_pose(self._ctx, C.byref(est)))
        return np.array(est.pose), np.array(est.covariance).reshape(3, 3)

    def cluster_based_estimate(self, linear_hash_resolution=0.20, angular_hash_resolution=0.524, weight_cap_percentile=0.90):
        """beluga::cluster_based_estimate (algorithm/cluster_based_estimation.hpp:415-433)."""
        cp = capi.ClusterParams(linear_hash_resolution, angular_hash_resolution, weight_cap_percentile)
        est = capi.Estimate()
        self._check(self._lib.mcl_cluster_based_estimate(self._ctx, C.byref(cp), C.byref(est)))
        return np.array(est.pose), np.array(est.covariance).reshape(3, 3)

    def set_estimate_kind(self, cluster_based: bool, **cluster_params):
        """What update() returns: beluga::estimate (beluga::Amcl) or cluster_based_estimate (beluga_ros::Amcl).
        On a sharded filter (comm_attach_rccl) a COLLECTIVE call: every rank makes it, concurrently, with the same arguments."""
        cp = capi.ClusterParams(cluster_params.get("linear_hash_resolution", 0.20), cluster_params.get("angular_hash_resolution", 0.524),
                                cluster_params.get("weight_cap_percentile", 0.90))
        self._check(self._lib.mcl_set_estimate_kind(self._ctx, int(cluster_based), C.byref(cp)))

    def build_cdf(self) -> float:
        t = C.c_double(0)
        self._check(self._lib.mcl_build_cdf(self._ctx, C.byref(t)))
        return t.value

    def device_view(self) -> capi.DeviceView:
        v = capi.DeviceView()
        self._check(self._lib.mcl_get_device_view(self._ctx, C.byref(v)))
        return v

    def set_num_particles(self, n: int):
        self._check(self._lib.mcl_set_num_particles(self._ctx, n))

    def resample_targets(self, step: int, random_state_probability: float, total: float, first_slot: int, count: int, d_targets: int):
        self._check(self._lib.mcl_resample_targets(self._ctx, step, random_state_probability, total, first_slot, count, d_targets))

    def route_targets(self, d_targets, count, d_ends, d_offsets, world, self_rank, d_send, d_order, d_counts):
        self._check(self._lib.mcl_route_targets(self._ctx, d_targets, count, d_ends, d_offsets, world, self_rank, d_send, d_order, d_counts))

    def serve_requests(self, d_requests, m, d_replies):
        self._check(self._lib.mcl_serve_requests(self._ctx, d_requests, m, d_replies))

    def commit_routed(self, step, first_slot, count, d_replies, d_order, d_targets):
        self._check(self._lib.mcl_commit_routed(self._ctx, step, first_slot, count, d_replies, d_order, d_targets))

    def finish_candidates(self, step, first_slot, count, d_replies, d_order, d_targets, d_states, d_hashes):
        self._check(self._lib.mcl_finish_candidates(self._ctx, step, first_slot, count, d_replies, d_order, d_targets, d_states, d_hashes))

    def kld_begin(self):
        self._check(self._lib.mcl_kld_begin(self._ctx))

    def kld_feed(self, d_hashes: int, count: int):
        """-> global index of the first candidate failing kld_condition, or None if the whole block passes."""
        fail = C.c_uint64(0)
        self._check(self._lib.mcl_kld_feed(self._ctx, d_hashes, count, C.byref(fail)))
        return None if fail.value == 0xFFFFFFFFFFFFFFFF else int(fail.value)

    def load_shard(self, d_states: int, n: int, shard_offset: int):
        self._check(self._lib.mcl_load_shard(self._ctx, d_states, n, shard_offset))

    def weight_sum_device(self, d_sum: int):
        self._check(self._lib.mcl_weight_sum_device(self._ctx, d_sum))

    def normalize_device(self, d_factor: int, d_stats: int):
        self._check(self._lib.mcl_normalize_device(self._ctx, d_factor, d_stats))

    def build_cdf_device(self, d_total: int):
        self._check(self._lib.mcl_build_cdf_device(self._ctx, d_total))

    def estimate_sums_device(self, pivot, d_sums: int):
        p = np.ascontiguousarray(pivot, dtype=np.float64)
        self._check(self._lib.mcl_estimate_sums_device(self._ctx, _dp(p), d_sums))

    def sync(self):
        self._check(self._lib.mcl_sync(self._ctx))

    def beam_cells_visited(self, reset: bool = True) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.mcl_beam_cells_visited(self._ctx, C.byref(v), int(reset)))
        return v.value

    # -- measurement hooks ---------------------------------------------------------------------------
    def profile_enable(self, on=True):
        """True / 2: HIP events around every stage; 1: around the sensor kernel only (what a timed run can afford: an event
        record costs ~5 us of stream time); False / 0: off."""
        level = 2 if on is True else int(on)
        self._check(self._lib.mcl_profile_enable(self._ctx, level))

    def comm_attach_rccl(self, unique_id: bytes, rank: int, world: int):
        """Joins the RCCL communicator of a sharded filter (include/beluga_mcl.h, "Particle shards"): this context must have been
        created with its shard_offset / shard_capacity; update() then runs the cycle over all shards.  COLLECTIVE for world > 1: the
        ranks attach concurrently (they exchange a word of their configuration and fail alike on a mismatch); set_option("device_policy")
        and set_estimate_kind are collective on an attached filter as well."""
        assert len(unique_id) == 128
        self._check(self._lib.mcl_comm_attach_rccl(self._ctx, unique_id, rank, world))

    def set_option(self, name: str, value: int):
        """A/B switch of the library (include/beluga_mcl.h, mcl_set_option); no option but field_build changes a result
        beyond the rounding of a particle's sum over the scan."""
        self._check(self._lib.mcl_set_option(self._ctx, name.encode(), int(value)))

    def debug_set_recovery_filters(self, slow: float, fast: float):
        """Test hook (mcl_debug_set_recovery_filters): the outputs of the recovery estimator's two exponential filters."""
        self._check(self._lib.mcl_debug_set_recovery_filters(self._ctx, float(slow), float(fast)))

    def counter(self, name: str) -> int:
        v = C.c_uint64(0)
        self._check(self._lib.mcl_get_counter(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def debug_order(self):
        """(perm, keys) of the spatial ordering of the current set: keys[perm] is non-decreasing."""
        n = self.num_particles()
        perm, keys = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        self._check(self._lib.mcl_debug_order(self._ctx, perm.ctypes.data_as(capi.c_u32_p), keys.ctypes.data_as(capi.c_u32_p)))
        return perm, keys

    def profile_read(self, reset: bool = True):
        ms = (C.c_double * len(capi.STAGES))()
        cnt = (C.c_uint64 * len(capi.STAGES))()
        self._check(self._lib.mcl_profile_read(self._ctx, ms, cnt, int(reset)))
        return {name: (ms[i], cnt[i]) for i, name in enumerate(capi.STAGES)}


def make_laser_scan(ranges, angle_min, angle_increment, range_min, range_max, origin_se3=(0, 0, 0, 1, 0, 0, 0), max_beams=2 ** 64 - 1,
                    min_range=float(np.finfo(np.float64).tiny), max_range=float(np.finfo(np.float64).max)):
    """A beluga_ros::LaserScan (laser_scan.hpp:46-66): the message fields + origin + decimation / range limits."""
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    scan = capi.LaserScan()
    scan.ranges = r.ctypes.data_as(capi.c_float_p)
    scan.num_ranges = len(r)
    scan.angle_min, scan.angle_increment = angle_min, angle_increment
    scan.range_min, scan.range_max = range_min, range_max
    scan.origin_se3 = (C.c_double * 7)(*origin_se3)
    scan.max_beams = max_beams
    scan.min_range, scan.max_range = min_range, max_range
    scan._keepalive = r
    return scan


def prepare_laser_scan(scan) -> np.ndarray:
    """points in the base frame, as beluga_ros::Amcl::update builds them (beluga_ros/src/amcl.cpp:54-63)."""
    lib = capi.load()
    out = np.zeros((min(scan.num_ranges, scan.max_beams) + 1, 2))
    m = C.c_uint64(0)
    st = lib.mcl_prepare_laser_scan(C.byref(scan), _dp(out), C.byref(m))
    if st != capi.MCL_OK:
        raise capi.MclError(st, "mcl_prepare_laser_scan")
    return out[:m.value].copy()


def estimate_from_sums(sums: np.ndarray):
    lib = capi.load()
    s = np.ascontiguousarray(sums, dtype=np.float64)
    est = capi.Estimate()
    st = lib.mcl_estimate_from_sums(_dp(s), C.byref(est))
    if st != capi.MCL_OK:
        raise capi.MclError(st, "mcl_estimate_from_sums")
    return np.array(est.pose), np.array(est.covariance).reshape(3, 3)


def project_point_cloud(points_xyz, origin_se3=(0, 0, 0, 1, 0, 0, 0)) -> np.ndarray:
    """Points of a beluga_ros::SparsePointCloud3f in the base frame, projected onto z = 0 (beluga_ros/src/amcl.cpp:73-76)."""
    lib = capi.load()
    pts = np.ascontiguousarray(points_xyz, dtype=np.float32).reshape(-1, 3)
    origin = np.ascontiguousarray(origin_se3, dtype=np.float64)
    out = np.zeros((len(pts), 2))
    st = lib.mcl_project_point_cloud(pts.ctypes.data_as(capi.c_float_p), len(pts), _dp(origin), _dp(out))
    if st != capi.MCL_OK:
        raise capi.MclError(st, "mcl_project_point_cloud")
    return out


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to the other ranks)."""
    lib = capi.load()
    buf = C.create_string_buffer(128)
    st = lib.mcl_comm_unique_id(buf)
    if st != capi.MCL_OK:
        raise capi.MclError(st, lib.mcl_last_error(None).decode())
    return buf.raw
