"""ctypes view of the C ABI in include/beluga_mcl.h (libbeluga_mcl.so).

The library is the product; this file only declares its signatures.  Loading fails loudly if the
shared object has not been built (`python -m beluga_amd.build`) — there is no Python/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BELUGA_MCL_LIB: another build of the same library (measurement variants: tools/build_variant.sh)
LIB_PATH = os.environ.get("BELUGA_MCL_LIB") or os.path.join(_HERE, "lib", "libbeluga_mcl.so")

MCL_OK = 0
MCL_ERR_INVALID_ARGUMENT = -1
MCL_ERR_HIP = -2
MCL_ERR_OUT_OF_MEMORY = -3
MCL_ERR_NOT_READY = -4
MCL_ERR_BAD_COVARIANCE = -5
MCL_ERR_NO_DEVICE = -6
MCL_ERR_UNSUPPORTED = -7

MCL_SENSOR_LIKELIHOOD_FIELD = 0
MCL_SENSOR_BEAM = 1
MCL_SENSOR_LIKELIHOOD_FIELD_PROB = 2
MCL_MOTION_DIFFERENTIAL, MCL_MOTION_OMNIDIRECTIONAL, MCL_MOTION_STATIONARY = 0, 1, 2

STAGES = ("propagate", "reweight", "normalize", "resample", "estimate", "sensor_kernel")

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_i8_p = C.POINTER(C.c_int8)
c_u64_p = C.POINTER(C.c_uint64)
c_u32_p = C.POINTER(C.c_uint32)


class AmclParams(C.Structure):
    _fields_ = [
        ("update_min_d", C.c_double), ("update_min_a", C.c_double), ("resample_interval", C.c_uint64),
        ("selective_resampling", C.c_int32), ("reserved0", C.c_int32), ("min_particles", C.c_uint64),
        ("max_particles", C.c_uint64), ("alpha_slow", C.c_double), ("alpha_fast", C.c_double), ("kld_epsilon", C.c_double),
        ("kld_z", C.c_double), ("spatial_resolution_x", C.c_double), ("spatial_resolution_y", C.c_double),
        ("spatial_resolution_theta", C.c_double),
    ]


class DiffDriveParams(C.Structure):
    _fields_ = [
        ("rotation_noise_from_rotation", C.c_double), ("rotation_noise_from_translation", C.c_double),
        ("translation_noise_from_translation", C.c_double), ("translation_noise_from_rotation", C.c_double),
        ("distance_threshold", C.c_double),
    ]


class LfParams(C.Structure):
    _fields_ = [
        ("max_obstacle_distance", C.c_double), ("max_laser_distance", C.c_double), ("z_hit", C.c_double), ("z_random", C.c_double),
        ("sigma_hit", C.c_double), ("model_unknown_space", C.c_int32), ("only_obstacle_boundaries", C.c_int32),
    ]


class BeamParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("z_hit", "z_short", "z_max", "z_rand", "sigma_hit", "lambda_short", "beam_max_range")]


class Config(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32), ("sensor_kind", C.c_int32), ("seed", C.c_uint64), ("amcl", AmclParams),
        ("motion", DiffDriveParams), ("lf", LfParams), ("beam", BeamParams), ("shard_offset", C.c_uint64),
        ("shard_capacity", C.c_uint64), ("hip_stream", C.c_void_p), ("motion_kind", C.c_int32), ("reserved1", C.c_int32),
        ("strafe_noise_from_translation", C.c_double),
    ]


class Estimate(C.Structure):
    _fields_ = [("pose", C.c_double * 4), ("covariance", C.c_double * 9)]


class UpdateInfo(C.Structure):
    _fields_ = [
        ("updated", C.c_int32), ("resampled", C.c_int32), ("num_particles", C.c_uint64), ("weight_sum", C.c_double),
        ("effective_sample_size", C.c_double), ("random_state_probability", C.c_double),
    ]


class WeightStats(C.Structure):
    _fields_ = [("sum", C.c_double), ("norm_sum", C.c_double), ("norm_sumsq", C.c_double)]


class LaserScan(C.Structure):
    _fields_ = [
        ("ranges", c_float_p), ("num_ranges", C.c_uint64), ("angle_min", C.c_float), ("angle_increment", C.c_float),
        ("range_min", C.c_float), ("range_max", C.c_float), ("origin_se3", C.c_double * 7), ("max_beams", C.c_uint64),
        ("min_range", C.c_double), ("max_range", C.c_double),
    ]


class ClusterParams(C.Structure):
    _fields_ = [("linear_hash_resolution", C.c_double), ("angular_hash_resolution", C.c_double), ("weight_cap_percentile", C.c_double)]


class DeviceView(C.Structure):
    _fields_ = [
        ("states", C.c_void_p), ("w", C.c_void_p), ("cdf", C.c_void_p),
        ("n", C.c_uint64), ("capacity", C.c_uint64), ("hip_stream", C.c_void_p),
    ]


_ctx = C.c_void_p

_SIGNATURES = {
    "mcl_default_config": (None, [C.POINTER(Config)]),
    "mcl_create": (C.c_int32, [C.POINTER(Config), C.POINTER(_ctx)]),
    "mcl_destroy": (None, [_ctx]),
    "mcl_last_error": (C.c_char_p, [_ctx]),
    "mcl_set_map": (C.c_int32, [_ctx, c_i8_p, C.c_uint32, C.c_uint32, C.c_double, c_double_p, c_i8_p]),
    "mcl_set_map_async": (C.c_int32, [_ctx, c_i8_p, C.c_uint32, C.c_uint32, C.c_double, c_double_p, c_i8_p]),
    "mcl_map_pending": (C.c_int32, [_ctx, C.POINTER(C.c_int32)]),
    "mcl_map_commit": (C.c_int32, [_ctx, C.c_int32]),
    "mcl_get_likelihood_field": (C.c_int32, [_ctx, c_float_p]),
    "mcl_set_likelihood_field": (C.c_int32, [_ctx, c_float_p]),
    "mcl_initialize_normal": (C.c_int32, [_ctx, c_double_p, c_double_p]),
    "mcl_set_particles": (C.c_int32, [_ctx, c_double_p, c_double_p, C.c_uint64]),
    "mcl_num_particles": (C.c_int32, [_ctx, c_u64_p]),
    "mcl_get_particles": (C.c_int32, [_ctx, c_double_p, c_double_p, C.c_uint64, c_u64_p]),
    "mcl_force_update": (C.c_int32, [_ctx]),
    "mcl_update": (C.c_int32, [_ctx, c_double_p, c_double_p, C.c_uint64, C.POINTER(Estimate), C.POINTER(UpdateInfo)]),
    "mcl_prepare_laser_scan": (C.c_int32, [C.POINTER(LaserScan), c_double_p, c_u64_p]),
    "mcl_update_laser_scan": (C.c_int32, [_ctx, c_double_p, C.POINTER(LaserScan), C.POINTER(Estimate), C.POINTER(UpdateInfo)]),
    "mcl_propagate": (C.c_int32, [_ctx, c_double_p, c_double_p, C.c_uint32]),
    "mcl_reweight": (C.c_int32, [_ctx, c_double_p, C.c_uint64]),
    "mcl_weight_sum": (C.c_int32, [_ctx, c_double_p]),
    "mcl_normalize": (C.c_int32, [_ctx, C.c_double, C.POINTER(WeightStats)]),
    "mcl_resample": (C.c_int32, [_ctx, C.c_double, C.c_uint32, c_u64_p]),
    "mcl_estimate_sums": (C.c_int32, [_ctx, c_double_p, c_double_p]),
    "mcl_estimate_from_sums": (C.c_int32, [c_double_p, C.POINTER(Estimate)]),
    "mcl_estimate_pose": (C.c_int32, [_ctx, C.POINTER(Estimate)]),
    "mcl_cluster_based_estimate": (C.c_int32, [_ctx, C.POINTER(ClusterParams), C.POINTER(Estimate)]),
    "mcl_set_estimate_kind": (C.c_int32, [_ctx, C.c_int32, C.POINTER(ClusterParams)]),
    "mcl_sample_particle_cloud": (C.c_int32, [_ctx, C.c_uint64, C.c_uint32, c_double_p]),
    "mcl_get_device_view": (C.c_int32, [_ctx, C.POINTER(DeviceView)]),
    "mcl_set_num_particles": (C.c_int32, [_ctx, C.c_uint64]),
    "mcl_build_cdf": (C.c_int32, [_ctx, c_double_p]),
    "mcl_resample_targets": (C.c_int32, [_ctx, C.c_uint32, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_void_p]),
    "mcl_route_targets": (C.c_int32, [_ctx, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcl_serve_requests": (C.c_int32, [_ctx, C.c_void_p, C.c_uint64, C.c_void_p]),
    "mcl_commit_routed": (C.c_int32, [_ctx, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcl_finish_candidates": (C.c_int32, [_ctx, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mcl_kld_begin": (C.c_int32, [_ctx]),
    "mcl_kld_feed": (C.c_int32, [_ctx, C.c_void_p, C.c_uint64, c_u64_p]),
    "mcl_load_shard": (C.c_int32, [_ctx, C.c_void_p, C.c_uint64, C.c_uint64]),
    "mcl_weight_sum_device": (C.c_int32, [_ctx, C.c_void_p]),
    "mcl_normalize_device": (C.c_int32, [_ctx, C.c_void_p, C.c_void_p]),
    "mcl_build_cdf_device": (C.c_int32, [_ctx, C.c_void_p]),
    "mcl_estimate_sums_device": (C.c_int32, [_ctx, c_double_p, C.c_void_p]),
    "mcl_sync": (C.c_int32, [_ctx]),
    "mcl_profile_enable": (C.c_int32, [_ctx, C.c_int32]),
    "mcl_profile_read": (C.c_int32, [_ctx, c_double_p, c_u64_p, C.c_int32]),
    "mcl_beam_cells_visited": (C.c_int32, [_ctx, c_u64_p, C.c_int32]),
    "mcl_initialize_from_map": (C.c_int32, [_ctx]),
    "mcl_has_likelihood_field": (C.c_int32, [_ctx, C.POINTER(C.c_int32)]),
    "mcl_get_likelihood_field_origin": (C.c_int32, [_ctx, c_double_p]),
    "mcl_project_point_cloud": (C.c_int32, [c_float_p, C.c_uint64, c_double_p, c_double_p]),
    "mcl_update_point_cloud": (C.c_int32, [_ctx, c_double_p, c_float_p, C.c_uint64, c_double_p, C.POINTER(Estimate), C.POINTER(UpdateInfo)]),
    "mcl_comm_attach": (C.c_int32, [_ctx, C.c_uint32, C.c_uint32, C.c_void_p]),
    "mcl_comm_unique_id": (C.c_int32, [C.c_char_p]),
    "mcl_comm_attach_rccl": (C.c_int32, [_ctx, C.c_char_p, C.c_uint32, C.c_uint32]),
    "mcl_set_option": (C.c_int32, [_ctx, C.c_char_p, C.c_int64]),
    "mcl_get_counter": (C.c_int32, [_ctx, C.c_char_p, c_u64_p]),
    "mcl_debug_order": (C.c_int32, [_ctx, c_u32_p, c_u32_p]),
    "mcl_debug_set_recovery_filters": (C.c_int32, [_ctx, C.c_double, C.c_double]),
    "mcl_debug_curve_index": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "mcl_version": (C.c_char_p, []),
    "mcl_measurement_build": (C.c_int, []),
}

_OPTIONAL = {"mcl_measurement_build"}  # symbols a library may lack and still load
_lib = None


def load():
    """Loads libbeluga_mcl.so. Raises if it is missing: the HIP library is the only implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing. Build it with `python -m beluga_amd.build` (needs hipcc); "
                "beluga_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        if os.environ.get("BELUGA_MCL_LIB"):
            import sys
            print(f"[beluga_amd] BELUGA_MCL_LIB: loading {LIB_PATH} instead of the package's own library", file=sys.stderr)
        # A measurement build of the kernels (tools/build_variant.sh: ablations compute nonsense by design) is refused as the product
        # library - wherever it was loaded from, the package's own path included - unless the caller asks for exactly that.
        try:
            lib.mcl_measurement_build.restype = C.c_int
            measurement = bool(lib.mcl_measurement_build())
        except AttributeError:  # a build from before the tag existed
            measurement = False
        if measurement and os.environ.get("BELUGA_MCL_ALLOW_MEASUREMENT_BUILD") != "1":
            raise ImportError(f"{LIB_PATH} is a measurement build of the kernels (timing hooks / ablations): its results are not "
                              "the product's.  Set BELUGA_MCL_ALLOW_MEASUREMENT_BUILD=1 to load it anyway (tools/ only).")
        for name, (res, args) in _SIGNATURES.items():
            if name in _OPTIONAL and not hasattr(lib, name):
                continue  # (an older build named by BELUGA_MCL_LIB)
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def exported_names():
    return sorted(_SIGNATURES)


class MclError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[mcl status {status}] {message}")
        self.status = status
