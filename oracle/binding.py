"""ctypes binding of the CPU oracle (oracle/libbeluga_oracle.so).

TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (beluga_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbeluga_oracle.so")

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_i8_p = C.POINTER(C.c_int8)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_u32_p = C.POINTER(C.c_uint32)
c_u64_p = C.POINTER(C.c_uint64)

ROS_TRAITS = (0, -1, 100)  # free, unknown, occupied (beluga_ros/occupancy_grid.hpp:48-64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "beluga_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class AmclConfig(C.Structure):
    _fields_ = [
        ("update_min_d", C.c_double),
        ("update_min_a", C.c_double),
        ("resample_interval", C.c_uint64),
        ("selective_resampling", C.c_int32),
        ("sensor_kind", C.c_int32),
        ("min_particles", C.c_uint64),
        ("max_particles", C.c_uint64),
        ("alpha_slow", C.c_double),
        ("alpha_fast", C.c_double),
        ("kld_epsilon", C.c_double),
        ("kld_z", C.c_double),
        ("hash_res", C.c_double * 3),
        ("alphas", C.c_double * 4),
        ("distance_threshold", C.c_double),
        ("lf", C.c_double * 5),
        ("lf_model_unknown_space", C.c_int32),
        ("lf_only_obstacle_boundaries", C.c_int32),
        ("beam", C.c_double * 7),
        ("seed", C.c_uint64),
        ("threads", C.c_int32),
        ("motion_kind", C.c_int32),
        ("alpha5", C.c_double),
    ]


_lib = None
_libs = {}


def _configure(L):
    L.orc_so2_log.restype = C.c_double
    L.orc_normalize.restype = C.c_double
    L.orc_effective_sample_size.restype = C.c_double
    L.orc_thrun.restype = C.c_double
    L.orc_kld_target_size.restype = C.c_uint64
    L.orc_kld_target_size.argtypes = [C.c_uint64, C.c_double, C.c_double]
    L.orc_kld_take_while.restype = C.c_uint64
    L.orc_kld_take_while.argtypes = [c_u64_p, C.c_uint64, C.c_uint64, C.c_double, C.c_double]
    L.orc_spatial_hash.restype = C.c_uint64
    L.orc_spatial_hash_xyt.restype = C.c_uint64
    L.orc_spatial_hash_xyt.argtypes = [C.c_double, C.c_double, C.c_double, c_double_p]
    L.orc_resample.restype = C.c_uint64
    L.orc_amcl_create.restype = C.c_void_p
    L.orc_amcl_num_free.restype = C.c_uint64
    L.orc_amcl_num_particles.restype = C.c_uint64
    L.orc_amcl_beam_steps.restype = C.c_int64
    for name in (
        "orc_amcl_destroy", "orc_amcl_set_map", "orc_amcl_set_field", "orc_amcl_get_field", "orc_amcl_num_free",
        "orc_amcl_set_particles", "orc_amcl_num_particles", "orc_amcl_get_particles", "orc_amcl_init_normal", "orc_amcl_init_from_map",
        "orc_amcl_force_update", "orc_amcl_stage_times", "orc_amcl_beam_steps", "orc_amcl_update",
    ):
        getattr(L, name).argtypes = None
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _libs["checker"] = _lib = _configure(C.CDLL(_LIB_PATH))
    return _lib


def native_library_path() -> str:
    """The timing build (-O3 -march=native) for THIS host: the file name carries a digest of the CPU's
    model and flags, so a library built on another box is never loaded (it could hold instructions this CPU lacks)."""
    import hashlib
    ident = []
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith(("model name", "flags")):
                    ident.append(line.strip())
                    if len(ident) == 2:
                        break
    except OSError:
        pass
    return os.path.join(_HERE, "libbeluga_oracle_native_" + hashlib.sha256("|".join(ident).encode()).hexdigest()[:12] + ".so")


def use_timing_build(on: bool) -> str:
    """bench.py's cpu_baseline leg only: switches every call of this module between the checker (-O2 -ffp-contract=off, the
    default) and the timing build.  Objects created under one build must be dropped before switching.  Returns the flags."""
    global _lib
    lib()
    if not on:
        _lib = _libs["checker"]
        return "-O2 -ffp-contract=off -fopenmp"
    if "native" not in _libs:
        path = native_library_path()
        src = os.path.join(_HERE, "beluga_oracle.cpp")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "native", "OUT=" + path])
        _libs["native"] = _configure(C.CDLL(path))
    _lib = _libs["native"]
    return "-O3 -march=native -fopenmp"


def _d(a):
    return a.ctypes.data_as(c_double_p)


def _dbl(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def se2(x, y, theta):
    out = np.zeros(4)
    lib().orc_se2_from_xytheta(C.c_double(x), C.c_double(y), C.c_double(theta), _d(out))
    return out


def se2_mul(a, b):
    a, b, out = _dbl(a), _dbl(b), np.zeros(4)
    lib().orc_se2_mul(_d(a), _d(b), _d(out))
    return out


def se2_inverse(a):
    a, out = _dbl(a), np.zeros(4)
    lib().orc_se2_inverse(_d(a), _d(out))
    return out


def so2_log(a):
    a = _dbl(a)
    return lib().orc_so2_log(_d(a))


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data_as(c_u32_p), k.ctypes.data_as(c_u32_p), out.ctypes.data_as(c_u32_p))
    return out


def draw(seed, step, purpose, index):
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_draw(C.c_uint64(seed), C.c_uint32(step), C.c_uint32(purpose), C.c_uint64(index), out.ctypes.data_as(c_u32_p))
    return out


def _traits(t):
    return (C.c_int8 * 3)(*t)


def make_likelihood_field(cells, res, lf_params, model_unknown_space=False, only_obstacle_boundaries=False, traits=ROS_TRAITS):
    """cells: (H, W) int8. lf_params: (max_obstacle_distance, max_laser_distance, z_hit, z_random, sigma_hit)."""
    cells = np.ascontiguousarray(cells, dtype=np.int8)
    H, W = cells.shape
    out = np.zeros((H, W), dtype=np.float32)
    p = (C.c_double * 5)(*lf_params)
    lib().orc_make_likelihood_field(
        cells.ctypes.data_as(c_i8_p), C.c_int(W), C.c_int(H), C.c_double(res), _traits(traits), p,
        C.c_int(int(model_unknown_space)), C.c_int(int(only_obstacle_boundaries)), out.ctypes.data_as(c_float_p))
    return out


def distance_map_1d(mask, max_value):
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.zeros(len(m), dtype=np.int32)
    lib().orc_distance_map_1d(m.ctypes.data_as(c_u8_p), C.c_int(len(m)), C.c_int(max_value), out.ctypes.data_as(c_i32_p))
    return out


def lf_weights(field, res, origin, max_laser_distance, states, points, threads=1):
    field = np.ascontiguousarray(field, dtype=np.float32)
    H, W = field.shape
    states = _dbl(states).reshape(-1, 4)
    points = _dbl(points).reshape(-1, 2)
    origin = _dbl(origin)
    out = np.zeros(len(states))
    lib().orc_lf_weights(
        field.ctypes.data_as(c_float_p), C.c_int(W), C.c_int(H), C.c_double(res), _d(origin), C.c_double(max_laser_distance),
        _d(states), C.c_uint64(len(states)), _d(points), C.c_uint64(len(points)), C.c_int(threads), _d(out))
    return out


def lf_prob_weights(field, res, origin, max_laser_distance, states, points):
    field = np.ascontiguousarray(field, dtype=np.float32)
    H, W = field.shape
    states = _dbl(states).reshape(-1, 4)
    points = _dbl(points).reshape(-1, 2)
    origin = _dbl(origin)
    out = np.zeros(len(states))
    lib().orc_lf_prob_weights(
        field.ctypes.data_as(c_float_p), C.c_int(W), C.c_int(H), C.c_double(res), _d(origin), C.c_double(max_laser_distance),
        _d(states), C.c_uint64(len(states)), _d(points), C.c_uint64(len(points)), _d(out))
    return out


MOTION_KINDS = {"differential": 0, "omnidirectional": 1, "stationary": 2}


def propagate_kind(states, kind, pose, prev, alphas, seed, step, index_offset=0, distance_threshold=0.01):
    s = _dbl(states).reshape(-1, 4).copy()
    pose, prev = _dbl(pose), _dbl(prev)
    a = (C.c_double * 5)(*(list(alphas) + [0.0] * (5 - len(alphas))))
    lib().orc_propagate_kind(_d(s), C.c_uint64(len(s)), C.c_int(MOTION_KINDS[kind]), _d(pose), _d(prev), a,
                             C.c_double(distance_threshold), C.c_uint64(seed), C.c_uint32(step), C.c_uint64(index_offset))
    return s


def beam_weights(cells, res, origin, beam_params, states, points, traits=ROS_TRAITS, threads=1, return_steps=False):
    cells = np.ascontiguousarray(cells, dtype=np.int8)
    H, W = cells.shape
    states = _dbl(states).reshape(-1, 4)
    points = _dbl(points).reshape(-1, 2)
    origin = _dbl(origin)
    out = np.zeros(len(states))
    steps = C.c_int64(0)
    bp = (C.c_double * 7)(*beam_params)
    lib().orc_beam_weights(
        cells.ctypes.data_as(c_i8_p), C.c_int(W), C.c_int(H), C.c_double(res), _d(origin), _traits(traits), bp, _d(states),
        C.c_uint64(len(states)), _d(points), C.c_uint64(len(points)), C.c_int(threads), _d(out), C.byref(steps))
    return (out, steps.value) if return_steps else out


def ray_cast(cells, res, origin, pose, max_range, bearing_theta, traits=ROS_TRAITS):
    cells = np.ascontiguousarray(cells, dtype=np.int8)
    H, W = cells.shape
    origin, pose = _dbl(origin), _dbl(pose)
    out = C.c_double(0)
    hit = lib().orc_ray_cast(
        cells.ctypes.data_as(c_i8_p), C.c_int(W), C.c_int(H), C.c_double(res), _d(origin), _traits(traits), _d(pose),
        C.c_double(max_range), C.c_double(bearing_theta), C.byref(out))
    return out.value if hit else None


def bresenham(p0, p1, modified=False, max_points=4096):
    out = np.zeros((max_points, 2), dtype=np.int32)
    n = lib().orc_bresenham(C.c_int(p0[0]), C.c_int(p0[1]), C.c_int(p1[0]), C.c_int(p1[1]), C.c_int(int(modified)),
                            out.ctypes.data_as(c_i32_p), C.c_int(max_points))
    return out[:n].copy()


def diffdrive_sampler(pose, prev, alphas, distance_threshold=0.01):
    pose, prev = _dbl(pose), _dbl(prev)
    a = (C.c_double * 4)(*alphas)
    out = np.zeros(6)
    lib().orc_diffdrive_sampler(_d(pose), _d(prev), a, C.c_double(distance_threshold), _d(out))
    return out


def propagate(states, sampler, seed, step, index_offset=0, threads=1):
    s = _dbl(states).reshape(-1, 4).copy()
    sampler = _dbl(sampler)
    lib().orc_propagate(_d(s), C.c_uint64(len(s)), _d(sampler), C.c_uint64(seed), C.c_uint32(step), C.c_uint64(index_offset),
                        C.c_int(threads))
    return s


def normalize(w):
    w = _dbl(w).copy()
    s = lib().orc_normalize(_d(w), C.c_uint64(len(w)))
    return w, s


def effective_sample_size(w):
    w = _dbl(w)
    return lib().orc_effective_sample_size(_d(w), C.c_uint64(len(w)))


class Thrun:
    def __init__(self, alpha_slow, alpha_fast):
        self.state = np.zeros(2)
        self.alpha_slow, self.alpha_fast = alpha_slow, alpha_fast

    def reset(self):
        self.state[:] = 0

    def __call__(self, w):
        w = _dbl(w)
        return lib().orc_thrun(_d(self.state), C.c_double(self.alpha_slow), C.c_double(self.alpha_fast), _d(w), C.c_uint64(len(w)))


def kld_target_size(k, epsilon, z):
    return lib().orc_kld_target_size(k, epsilon, z)


def kld_take_while(hashes, min_, epsilon, z):
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    return lib().orc_kld_take_while(h.ctypes.data_as(c_u64_p), len(h), min_, epsilon, z)


def spatial_hash(state, res):
    s, r = _dbl(state), _dbl(res)
    return lib().orc_spatial_hash(_d(s), _d(r))


def spatial_hash_xyt(x, y, t, res):
    r = _dbl(res)
    return lib().orc_spatial_hash_xyt(x, y, t, _d(r))


def resample(states, w, min_particles, max_particles, kld_epsilon, kld_z, hash_res, random_state_probability, seed, step,
             free_xy=None):
    states = _dbl(states).reshape(-1, 4)
    w = _dbl(w)
    hr = _dbl(hash_res)
    if free_xy is None:
        free_xy = np.zeros((0, 2))
    free_xy = _dbl(free_xy).reshape(-1, 2)
    out = np.zeros((max_particles, 4))
    anc = np.zeros(max_particles, dtype=np.int64)
    n = lib().orc_resample(
        _d(states), _d(w), C.c_uint64(len(w)), C.c_uint64(min_particles), C.c_uint64(max_particles), C.c_double(kld_epsilon),
        C.c_double(kld_z), _d(hr), C.c_double(random_state_probability), C.c_uint64(seed), C.c_uint32(step), _d(free_xy),
        C.c_uint64(len(free_xy)), _d(out), anc.ctypes.data_as(c_i64_p))
    return out[:n].copy(), anc[:n].copy()


def estimate(states, w):
    states = _dbl(states).reshape(-1, 4)
    w = _dbl(w)
    mean, cov = np.zeros(4), np.zeros(9)
    lib().orc_estimate(_d(states), _d(w), C.c_uint64(len(w)), _d(mean), _d(cov))
    return mean, cov.reshape(3, 3)


def cluster_ids(states, w, linear_res=0.2, angular_res=0.524, percentile=0.9):
    states, w = _dbl(states).reshape(-1, 4), _dbl(w)
    out = np.zeros(len(w), dtype=np.uint64)
    lib().orc_cluster_ids(_d(states), _d(w), C.c_uint64(len(w)), C.c_double(linear_res), C.c_double(angular_res), C.c_double(percentile),
                          out.ctypes.data_as(c_u64_p))
    return out


def cluster_based_estimate(states, w, linear_res=0.2, angular_res=0.524, percentile=0.9):
    states, w = _dbl(states).reshape(-1, 4), _dbl(w)
    mean, cov = np.zeros(4), np.zeros(9)
    lib().orc_cluster_based_estimate(_d(states), _d(w), C.c_uint64(len(w)), C.c_double(linear_res), C.c_double(angular_res),
                                     C.c_double(percentile), _d(mean), _d(cov))
    return mean, cov.reshape(3, 3)


def covariance_transform(cov):
    cov = _dbl(cov).reshape(9)
    T = np.zeros(9)
    ok = lib().orc_covariance_transform(_d(cov), _d(T))
    return T.reshape(3, 3) if ok else None


def init_normal(n, mean_xytheta, cov, seed, index_offset=0):
    states, w = np.zeros((n, 4)), np.zeros(n)
    m = _dbl(mean_xytheta)
    cv = _dbl(cov).reshape(9)
    ok = lib().orc_init_normal(_d(states), _d(w), C.c_uint64(n), _d(m), _d(cv), C.c_uint64(seed), C.c_uint64(index_offset))
    if not ok:
        raise RuntimeError("Invalid covariance matrix")
    return states, w


class Amcl:
    """The oracle's beluga::Amcl (amcl_core.hpp:81-233) with DifferentialDriveModel + LF/beam sensor model."""

    def __init__(self, *, update_min_d=0.25, update_min_a=0.2, resample_interval=1, selective_resampling=False,
                 min_particles=500, max_particles=2000, alpha_slow=0.001, alpha_fast=0.1, kld_epsilon=0.05, kld_z=3.0,
                 hash_res=(0.5, 0.5, np.deg2rad(10.0)), alphas=(0.1, 0.05, 0.1, 0.05), distance_threshold=0.01,
                 sensor="likelihood_field", lf=(100.0, 2.0, 0.5, 0.5, 0.2), lf_model_unknown_space=False,
                 lf_only_obstacle_boundaries=False, beam=(0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 60.0), seed=0, threads=1,
                 motion="differential", alpha5=0.0):
        cfg = AmclConfig()
        cfg.update_min_d, cfg.update_min_a = update_min_d, update_min_a
        cfg.resample_interval = resample_interval
        cfg.selective_resampling = int(selective_resampling)
        cfg.sensor_kind = {"likelihood_field": 0, "beam": 1, "likelihood_field_prob": 2}[sensor]
        cfg.motion_kind = MOTION_KINDS[motion]
        cfg.alpha5 = alpha5
        cfg.min_particles, cfg.max_particles = min_particles, max_particles
        cfg.alpha_slow, cfg.alpha_fast = alpha_slow, alpha_fast
        cfg.kld_epsilon, cfg.kld_z = kld_epsilon, kld_z
        cfg.hash_res = (C.c_double * 3)(*hash_res)
        cfg.alphas = (C.c_double * 4)(*alphas)
        cfg.distance_threshold = distance_threshold
        cfg.lf = (C.c_double * 5)(*lf)
        cfg.lf_model_unknown_space = int(lf_model_unknown_space)
        cfg.lf_only_obstacle_boundaries = int(lf_only_obstacle_boundaries)
        cfg.beam = (C.c_double * 7)(*beam)
        cfg.seed = seed
        cfg.threads = threads
        self._cfg = cfg
        self._h = C.c_void_p(lib().orc_amcl_create(C.byref(cfg)))
        self._shape = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_amcl_destroy(self._h)
            self._h = None

    def set_map(self, cells, res, origin, traits=ROS_TRAITS):
        cells = np.ascontiguousarray(cells, dtype=np.int8)
        H, W = cells.shape
        self._shape = (H, W)
        origin = _dbl(origin)
        lib().orc_amcl_set_map(self._h, cells.ctypes.data_as(c_i8_p), C.c_int(W), C.c_int(H), C.c_double(res), _d(origin),
                               _traits(traits))

    def set_field(self, field):
        field = np.ascontiguousarray(field, dtype=np.float32)
        assert field.shape == self._shape
        lib().orc_amcl_set_field(self._h, field.ctypes.data_as(c_float_p))

    def get_field(self):
        out = np.zeros(self._shape, dtype=np.float32)
        lib().orc_amcl_get_field(self._h, out.ctypes.data_as(c_float_p))
        return out

    def num_free(self):
        return lib().orc_amcl_num_free(self._h)

    def set_particles(self, states, w):
        states = _dbl(states).reshape(-1, 4)
        w = _dbl(w)
        lib().orc_amcl_set_particles(self._h, _d(states), _d(w), C.c_uint64(len(w)))

    def particles(self):
        n = lib().orc_amcl_num_particles(self._h)
        states, w = np.zeros((n, 4)), np.zeros(n)
        lib().orc_amcl_get_particles(self._h, _d(states), _d(w))
        return states, w

    def initialize(self, mean_xytheta, cov):
        m = _dbl(mean_xytheta)
        cv = _dbl(cov).reshape(9)
        if not lib().orc_amcl_init_normal(self._h, _d(m), _d(cv)):
            raise RuntimeError("Invalid covariance matrix")

    def initialize_from_map(self):
        """beluga_ros::Amcl::initialize_from_map() (beluga_ros/include/beluga_ros/amcl.hpp:209)."""
        if not lib().orc_amcl_init_from_map(self._h):
            raise RuntimeError("the map has no free cell")

    def force_update(self):
        lib().orc_amcl_force_update(self._h)

    def update(self, control, points):
        control = _dbl(control)
        points = _dbl(points).reshape(-1, 2)
        mean, cov, info = np.zeros(4), np.zeros(9), np.zeros(4)
        ok = lib().orc_amcl_update(self._h, _d(control), _d(points), C.c_uint64(len(points)), _d(mean), _d(cov), _d(info))
        if not ok:
            return None
        self.last_info = {"resampled": bool(info[0]), "random_state_probability": info[1], "ess": info[2], "weight_sum": info[3]}
        return mean, cov.reshape(3, 3)

    def stage_times(self):
        out = np.zeros(5)
        lib().orc_amcl_stage_times(self._h, _d(out))
        return dict(zip(("propagate", "reweight", "normalize", "resample", "estimate"), out))

    def beam_steps(self):
        return lib().orc_amcl_beam_steps(self._h)


def take_evenly_indices(size, count):
    L = lib()
    L.orc_take_evenly_index.restype = C.c_uint64
    L.orc_take_evenly_index.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    taken = 0 if size == 0 else min(size, count)
    out = [L.orc_take_evenly_index(k, size, count) for k in range(taken)]
    return [i for i in out if i < size]


def prepare_laser_scan(ranges, angle_min, angle_increment, range_min, range_max, origin_se3=(0, 0, 0, 1, 0, 0, 0),
                       max_beams=2 ** 63, min_range=np.finfo(np.float64).tiny, max_range=np.finfo(np.float64).max):
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    o = _dbl(origin_se3)
    out = np.zeros((len(r), 2))
    L = lib()
    L.orc_prepare_laser_scan.restype = C.c_uint64
    m = L.orc_prepare_laser_scan(r.ctypes.data_as(c_float_p), C.c_uint64(len(r)), C.c_float(angle_min), C.c_float(angle_increment),
                                 C.c_float(range_min), C.c_float(range_max), _d(o), C.c_uint64(max_beams), C.c_double(min_range),
                                 C.c_double(max_range), _d(out))
    return out[:m].copy()


def max_threads():
    return lib().orc_max_threads()


def project_point_cloud(points_xyz, origin_se3=(0, 0, 0, 1, 0, 0, 0)):
    """beluga_ros::Amcl::update(pose, SparsePointCloud3f)'s measurement (beluga_ros/src/amcl.cpp:73-76)."""
    pts = np.ascontiguousarray(points_xyz, dtype=np.float32).reshape(-1, 3)
    origin = _dbl(origin_se3)
    out = np.zeros((len(pts), 2))
    lib().orc_project_point_cloud(pts.ctypes.data_as(c_float_p), C.c_uint64(len(pts)), _d(origin), _d(out))
    return out
