"""Pins the CPU oracle (oracle/beluga_oracle.cpp) against the reference's own unit-test golden vectors.

Each test names the reference test it restates (paths relative to /root/reference/beluga/test/beluga/).
Tolerances are the reference's own.
"""
import math

import numpy as np
import pytest

from oracle import binding as orc

F, T = 0, 100  # false/true of StaticOccupancyGrid<.., bool> mapped onto int8 free/occupied


def grid5(rows):
    return np.array(rows, dtype=np.int8).reshape(5, 5)


IDENTITY = np.array([1.0, 0.0, 0.0, 0.0])
LF_PARAMS = (2.0, 20.0, 0.5, 0.5, 0.2)  # LikelihoodFieldModelParam{2.0, 20.0, 0.5, 0.5, 0.2}


def test_philox_known_answers():
    # Random123 kat_vectors (philox4x32 10 rounds)
    assert list(orc.philox([0, 0, 0, 0], [0, 0])) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert list(orc.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert list(orc.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


# ---- sensor/test_likelihood_field_model_base.cpp -------------------------------------------------
def test_lf_base_likelihood_field():  # :34-60
    cells = grid5([F, F, F, F, T,
                   F, F, F, T, F,
                   F, F, T, F, F,
                   F, T, F, F, F,
                   T, F, F, F, F])
    expected = [0.025, 0.025, 0.025, 0.069, 1.022,
                0.025, 0.027, 0.069, 1.022, 0.069,
                0.025, 0.069, 1.022, 0.069, 0.025,
                0.069, 1.022, 0.069, 0.027, 0.025,
                1.022, 0.069, 0.025, 0.025, 0.025]
    field = orc.make_likelihood_field(cells, 0.5, LF_PARAMS)
    np.testing.assert_allclose(field.ravel(), expected, atol=0.003)


def _to_likelihood(sq, sigma=0.2, z_hit=0.5, z_random=0.5, max_laser=2.0):
    amplitude = z_hit / (sigma * math.sqrt(2 * math.pi))
    return amplitude * math.exp(-sq / (2 * sigma * sigma)) + z_random / max_laser


def test_lf_base_thick_walls_combinations():  # :62-150
    cells = grid5([F, F, F, F, F,
                   F, T, T, T, F,
                   F, T, T, T, F,
                   F, T, T, T, F,
                   F, F, F, F, F])
    p = (10.0, 2.0, 0.5, 0.5, 0.2)
    f = orc.make_likelihood_field(cells, 1.0, p, False, False)
    assert f[2, 2] == pytest.approx(_to_likelihood(0.0), abs=1e-6)
    assert f[1, 1] == pytest.approx(_to_likelihood(0.0), abs=1e-6)
    f = orc.make_likelihood_field(cells, 1.0, p, False, True)
    assert f[0, 0] == pytest.approx(_to_likelihood(2.0), abs=1e-6)
    assert f[2, 2] == pytest.approx(_to_likelihood(1.0), abs=1e-6)
    assert f[1, 1] == pytest.approx(_to_likelihood(0.0), abs=1e-6)
    f = orc.make_likelihood_field(cells, 1.0, p, True, False)
    assert f[2, 2] == pytest.approx(_to_likelihood(0.0), abs=1e-6)
    assert f[1, 1] == pytest.approx(_to_likelihood(0.0), abs=1e-6)
    f = orc.make_likelihood_field(cells, 1.0, p, True, True)
    assert f[2, 2] == pytest.approx(1.0 / 2.0, abs=1e-6)
    assert f[1, 1] == pytest.approx(_to_likelihood(0.0), abs=1e-6)


def test_lf_base_hollow_thick_walls():  # :152-201
    cells = np.array([F, F, F, F, F, F, F,
                      F, T, T, T, T, T, F,
                      F, T, T, T, T, T, F,
                      F, T, T, F, T, T, F,
                      F, T, T, T, T, T, F,
                      F, T, T, T, T, T, F,
                      F, F, F, F, F, F, F], dtype=np.int8).reshape(7, 7)
    f = orc.make_likelihood_field(cells, 1.0, (10.0, 2.0, 0.5, 0.5, 0.2), False, True)
    assert f[3, 3] == pytest.approx(_to_likelihood(1.0), abs=1e-6)
    assert f[2, 2] == pytest.approx(_to_likelihood(1.0), abs=1e-6)


# ---- sensor/test_lfm_with_unknown_space.cpp ------------------------------------------------------
def test_lf_unknown_space_fields():  # :34-137
    cells = grid5([-1, -1, -1, 100, 100,
                   -1, 0, 0, 0, 100,
                   -1, 0, 0, 0, 100,
                   100, 0, 0, 0, 100,
                   100, 100, 100, 100, 100])
    U = 1 / 20.0
    f = orc.make_likelihood_field(cells, 0.5, LF_PARAMS, True, False)
    np.testing.assert_allclose(f.ravel(), [U, U, U, 1.022, 1.022,
                                           U, 0.025, 0.027, 0.069, 1.022,
                                           U, 0.027, 0.025, 0.069, 1.022,
                                           1.022, 0.069, 0.069, 0.069, 1.022,
                                           1.022, 1.022, 1.022, 1.022, 1.022], atol=0.003)
    f = orc.make_likelihood_field(cells, 0.5, LF_PARAMS, True, True)
    np.testing.assert_allclose(f.ravel(), [U, U, U, 1.022, U,
                                           U, 0.025, 0.027, 0.069, 1.022,
                                           U, 0.027, 0.025, 0.069, 1.022,
                                           1.022, 0.069, 0.069, 0.069, 1.022,
                                           U, 1.022, 1.022, 1.022, U], atol=0.003)
    cells2 = grid5([-1, -1, -1, 100, 100,
                    -1, -1, -1, 0, 0,
                    -1, -1, -1, 0, 0,
                    -1, -1, -1, 0, 0,
                    -1, -1, -1, 100, 100])
    U = 1 / 100.0
    f = orc.make_likelihood_field(cells2, 0.5, (2.0, 100.0, 0.5, 0.5, 0.2), True, False)
    np.testing.assert_allclose(f.ravel(), [U, U, U, 1.002, 1.002,
                                           U, U, U, 0.049, 0.049,
                                           U, U, U, 0.005, 0.005,
                                           U, U, U, 0.049, 0.049,
                                           U, U, U, 1.002, 1.002], atol=0.003)


# ---- sensor/test_likelihood_field_model.cpp ------------------------------------------------------
CENTER = grid5([F, F, F, F, F,
                F, F, F, F, F,
                F, F, T, F, F,
                F, F, F, F, F,
                F, F, F, F, F])
CORNER = grid5([F] * 24 + [T])


def _lf_weight(cells, res, origin, state, points, params=LF_PARAMS):
    field = orc.make_likelihood_field(cells, res, params)
    return orc.lf_weights(field, res, origin, params[1], [state], points)[0]


def test_lf_importance_weight():  # :34-74
    assert _lf_weight(CENTER, 0.5, IDENTITY, IDENTITY, [(1.25, 1.25)]) == pytest.approx(2.068, abs=0.003)
    assert _lf_weight(CENTER, 0.5, IDENTITY, IDENTITY, [(2.25, 2.25)]) == pytest.approx(1.000, abs=0.003)
    assert _lf_weight(CENTER, 0.5, IDENTITY, IDENTITY, [(-50.0, 50.0)]) == pytest.approx(1.000, abs=0.003)
    assert _lf_weight(CENTER, 0.5, IDENTITY, IDENTITY, [(1.20, 1.20), (1.25, 1.25), (1.30, 1.30)]) == pytest.approx(4.205, abs=0.01)
    assert _lf_weight(CENTER, 0.5, IDENTITY, orc.se2(1.25, 1.25, 0.0), [(0.0, 0.0)]) == pytest.approx(2.068, abs=0.003)


def test_lf_grid_with_offset():  # :76-103
    origin = orc.se2(-5, -5, 0.0)
    assert _lf_weight(CORNER, 2.0, origin, IDENTITY, [(4.5, 4.5)]) == pytest.approx(2.068, abs=0.003)
    assert _lf_weight(CORNER, 2.0, origin, origin, [(9.5, 9.5)]) == pytest.approx(2.068, abs=0.003)


def test_lf_grid_with_rotation():  # :105-130
    origin = orc.se2(0.0, 0.0, math.pi / 2)
    assert _lf_weight(CORNER, 2.0, origin, IDENTITY, [(-9.5, 9.5)]) == pytest.approx(2.068, abs=0.003)
    assert _lf_weight(CORNER, 2.0, origin, origin, [(9.5, 9.5)]) == pytest.approx(2.068, abs=0.003)


def test_lf_grid_with_rotation_and_offset():  # :132-158
    c, s = math.cos(math.pi / 2), math.sin(math.pi / 2)
    origin = orc.se2(c * -5 - s * -5, s * -5 + c * -5, math.pi / 2)
    assert _lf_weight(CORNER, 2.0, origin, IDENTITY, [(-4.5, 4.5)]) == pytest.approx(2.068, abs=0.003)
    assert _lf_weight(CORNER, 2.0, origin, origin, [(9.5, 9.5)]) == pytest.approx(2.068, abs=0.003)


def test_lf_grid_updates():  # :160-197
    assert _lf_weight(CENTER, 0.5, IDENTITY, IDENTITY, [(1.0, 1.0)]) == pytest.approx(2.068577607986223, abs=1e-6)
    assert _lf_weight(CORNER, 0.5, IDENTITY, IDENTITY, [(1.0, 1.0)]) == pytest.approx(1.0, abs=1e-3)


# ---- sensor/test_beam_model.cpp ------------------------------------------------------------------
BEAM = (0.5, 0.05, 0.05, 0.5, 0.2, 0.1, 60.0)  # z_hit z_short z_max z_rand sigma_hit lambda_short max_range (:29-38)


def test_beam_importance_weight():  # :40-82
    w = lambda pts: orc.beam_weights(CENTER, 0.5, IDENTITY, BEAM, [IDENTITY], pts)[0]
    assert w([(1.0, 1.0)]) == pytest.approx(1.0171643824743635, abs=1e-6)
    assert w([(0.75, 0.75)]) == pytest.approx(0.015905891701088148, abs=1e-6)
    assert w([(2.25, 2.25)]) == pytest.approx(0.000, abs=1e-6)
    assert w([(60.0, 60.0)]) == pytest.approx(0.00012500000000000003, abs=1e-6)


def test_beam_grid_updates():  # :84-122
    empty = grid5([F] * 25)
    assert orc.beam_weights(CENTER, 0.5, IDENTITY, BEAM, [IDENTITY], [(1.0, 1.0)])[0] == pytest.approx(1.0171643824743635, abs=1e-6)
    assert orc.beam_weights(empty, 0.5, IDENTITY, BEAM, [IDENTITY], [(1.0, 1.0)])[0] == pytest.approx(0.0, abs=1e-3)


# ---- algorithm/test_raycasting.cpp ---------------------------------------------------------------
def test_raycasting_nominal():  # :31-104 (EXPECT_EQ => exact)
    rc = lambda pose, rng, th: orc.ray_cast(CENTER, 0.5, IDENTITY, pose, rng, th)
    assert rc(orc.se2(0.5, 0.0, 0.0), 5.0, 0.0) is None
    assert rc(orc.se2(0.0, 1.0, 0.0), 5.0, 0.0) == 1.0
    assert rc(orc.se2(0.0, 1.0, 0.0), 5.0, math.pi / 2) is None
    assert rc(orc.se2(1.0, 1.0, 0.0), 5.0, math.pi / 2) == 0.0
    assert rc(orc.se2(0.0, 0.0, math.pi / 2), 1.0, 0.0) is None
    assert rc(orc.se2(1.0, 0.0, 0.0), 5.0, math.pi / 2) == 1.0
    assert rc(orc.se2(0.0, 0.0, 0.0), 5.0, math.pi / 4) == math.sqrt(2)


def test_raycasting_non_identity_origin():  # :106-131
    origin = orc.se2(0.5, 0.0, -math.pi / 4)
    assert orc.ray_cast(CENTER, 0.5, origin, orc.se2(0.5, 0.0, 0.0), 5.0, 0.0) == math.sqrt(2)


# ---- algorithm/raycasting/test_bresenham.cpp -----------------------------------------------------
@pytest.mark.parametrize("p0,p1,expected", [
    ((0, 0), (0, 0), [(0, 0)]),
    ((0, 0), (1, 1), [(0, 0), (1, 1)]),
    ((1, 1), (0, 0), [(1, 1), (0, 0)]),
    ((0, 0), (2, 1), [(0, 0), (1, 0), (2, 1)]),
    ((2, 1), (0, 0), [(2, 1), (1, 1), (0, 0)]),
    ((0, 2), (0, 0), [(0, 2), (0, 1), (0, 0)]),
    ((3, 2), (0, 0), [(3, 2), (2, 1), (1, 1), (0, 0)]),
])
def test_bresenham_standard(p0, p1, expected):  # :47-123
    assert [tuple(p) for p in orc.bresenham(p0, p1, modified=False)] == expected


@pytest.mark.parametrize("p0,p1,expected", [
    ((0, 0), (0, 0), [(0, 0)]),
    ((0, 0), (1, 1), [(0, 0), (1, 0), (0, 1), (1, 1)]),
    ((1, 1), (0, 0), [(1, 1), (0, 1), (1, 0), (0, 0)]),
    ((0, 0), (2, 1), [(0, 0), (1, 0), (1, 1), (2, 1)]),
    ((2, 1), (0, 0), [(2, 1), (1, 1), (1, 0), (0, 0)]),
    ((0, 2), (0, 0), [(0, 2), (0, 1), (0, 0)]),
    ((3, 2), (0, 0), [(3, 2), (2, 2), (2, 1), (1, 1), (1, 0), (0, 0)]),
])
def test_bresenham_modified(p0, p1, expected):  # :125-205
    assert [tuple(p) for p in orc.bresenham(p0, p1, modified=True)] == expected


# ---- algorithm/test_distance_map.cpp -------------------------------------------------------------
@pytest.mark.parametrize("mask,maxv,expected", [
    ([0] * 6, 10, [10] * 6),
    ([1] * 6, 10, [0] * 6),
    ([0, 1, 0, 0, 0, 1], 10, [1, 0, 1, 2, 1, 0]),
    ([1, 1, 0, 0, 0, 0], 10, [0, 0, 1, 2, 3, 4]),
    ([0, 0, 0, 0, 0, 1], 10, [5, 4, 3, 2, 1, 0]),
    ([0, 0, 0, 0, 0, 1], 3, [3, 3, 3, 2, 1, 0]),
])
def test_distance_map(mask, maxv, expected):  # :51-85
    assert list(orc.distance_map_1d(mask, maxv)) == expected


# ---- motion/test_differential_drive_model.cpp (noise-free exact poses, tol 1e-3) ------------------
def _apply(control, prev, state):
    sampler = orc.diffdrive_sampler(control, prev, (0.0, 0.0, 0.0, 0.0))
    return orc.propagate([state], sampler, seed=7, step=1)[0]


def _se2_near(a, b, tol):
    assert abs(orc.so2_log(orc.se2_mul(orc.se2_inverse(a), b))) < tol or abs(abs(orc.so2_log(orc.se2_mul(orc.se2_inverse(a), b))) - 2 * math.pi) < tol
    assert abs(a[2] - b[2]) < tol and abs(a[3] - b[3]) < tol


def test_diffdrive_noise_free():  # :56-106
    pi = math.pi
    pose = orc.se2(2.0, 5.0, pi / 3)
    _se2_near(_apply(orc.se2(1.0, -2.0, pi), orc.se2(1.0, -2.0, pi), pose), pose, 1e-3)  # OneUpdate
    c, p = orc.se2(1.0, 0.0, 0.0), orc.se2(0.0, 0.0, 0.0)  # Translate
    _se2_near(_apply(c, p, orc.se2(2.0, 0.0, 0.0)), orc.se2(3.0, 0.0, 0.0), 1e-3)
    _se2_near(_apply(c, p, orc.se2(0.0, 3.0, 0.0)), orc.se2(1.0, 3.0, 0.0), 1e-3)
    c = orc.se2(0.0, 1.0, pi / 2)  # RotateTranslate
    _se2_near(_apply(c, p, orc.se2(0.0, 0.0, 0.0)), orc.se2(0.0, 1.0, pi / 2), 1e-3)
    _se2_near(_apply(c, p, orc.se2(2.0, 3.0, -pi / 2)), orc.se2(3.0, 3.0, 0.0), 1e-3)
    c = orc.se2(0.0, 0.0, pi / 4)  # Rotate
    _se2_near(_apply(c, p, orc.se2(0.0, 0.0, pi)), orc.se2(0.0, 0.0, pi * 5 / 4), 1e-3)
    _se2_near(_apply(c, p, orc.se2(0.0, 0.0, -pi / 2)), orc.se2(0.0, 0.0, -pi / 4), 1e-3)
    c = orc.se2(1.0, 2.0, -pi / 2)  # RotateTranslateRotate
    _se2_near(_apply(c, p, orc.se2(3.0, 4.0, pi)), orc.se2(2.0, 2.0, pi / 2), 1e-3)


def _samples(alphas, control, prev, start, n=100_000, seed=1234):
    sampler = orc.diffdrive_sampler(control, prev, alphas)
    return orc.propagate(np.tile(start, (n, 1)), sampler, seed=seed, step=3)


def test_diffdrive_statistics():  # :122-257 — the reference's five distributional cases, its tolerances
    pi, alpha = math.pi, 0.2
    zero = orc.se2(0, 0, 0)
    # Translate (:122-140) tol 0.015
    out = _samples((0, 0, alpha, 0), orc.se2(3.0, 0, 0), zero, orc.se2(5.0, 0, 0))
    assert out[:, 2].mean() == pytest.approx(8.0, abs=0.015)
    assert out[:, 2].std() == pytest.approx(math.sqrt(alpha * 9.0), abs=0.015)
    # RotateFirstQuadrant (:142-160) tol 0.01
    out = _samples((alpha, 0, 0, 0), orc.se2(0, 0, pi / 4), zero, orc.se2(0, 0, pi / 6))
    th = np.arctan2(out[:, 1], out[:, 0])
    assert th.mean() == pytest.approx(pi / 6 + pi / 4, abs=0.01)
    assert th.std() == pytest.approx(math.sqrt(alpha * (pi / 4) ** 2), abs=0.01)
    # RotateThirdQuadrant (:185-207): backward/forward symmetric noise
    out = _samples((alpha, 0, 0, 0), orc.se2(0, 0, -pi * 3 / 4), zero, orc.se2(0, 0, pi / 6))
    th = np.arctan2(out[:, 1], out[:, 0])
    assert th.mean() == pytest.approx(pi / 6 - pi * 3 / 4, abs=0.01)
    assert th.std() == pytest.approx(math.sqrt(alpha * (pi / 4) ** 2), abs=0.01)
    # RotateTranslateRotate First/Third quadrant (:209-257)
    for tgt in ((1.0, 1.0), (-1.0, -1.0)):
        out = _samples((0, 0, 0, alpha), orc.se2(tgt[0], tgt[1], 0), zero, zero)
        norm = np.hypot(out[:, 2], out[:, 3])
        assert norm.mean() == pytest.approx(1.41, abs=0.01)
        assert norm.std() == pytest.approx(math.sqrt(alpha * 2 * (pi / 4) ** 2), abs=0.01)


# ---- views/test_take_while_kld.cpp ---------------------------------------------------------------
P90, P99 = 1.28155156327703, 2.32634787735669


@pytest.mark.parametrize("z,k,expected", [
    (P90, 3, 228), (P90, 4, 311), (P90, 5, 388), (P90, 6, 461), (P90, 7, 531), (P90, 100, 5871),
    (P99, 3, 462), (P99, 4, 569), (P99, 5, 666), (P99, 6, 756), (P99, 7, 843), (P99, 100, 6733),
])
def test_kld_condition_table(z, k, expected):  # :112-148 ; GenerateDistinctHashes(k) cycles k distinct hashes
    hashes = np.arange(20000, dtype=np.uint64) % np.uint64(k)
    assert orc.kld_take_while(hashes, 0, 0.01, z) == expected


def test_kld_min_max_limit():  # :150-188
    ones = np.ones(5000, dtype=np.uint64)
    assert orc.kld_take_while(np.zeros(0, dtype=np.uint64), 2, 0.1, 3.0) == 0  # TakeZero
    assert min(orc.kld_take_while(ones, 200, 0.05, 3.0), 1200) == 1200  # TakeMaximum (k<=2 => never stops, cap at max)
    # generate(1) | intersperse(2) | intersperse(3): 1 3 2 3 1 3 2 3 ...
    pattern = np.array([1, 3, 2, 3] * 2000, dtype=np.uint64)
    assert min(orc.kld_take_while(pattern, 0, 0.05, 3.0), 1200) == 135  # TakeLimit
    assert min(orc.kld_take_while(pattern, 200, 0.05, 3.0), 1200) == 200  # TakeMinimum


# ---- test_spatial_hash.cpp -----------------------------------------------------------------------
def test_spatial_hash_properties():  # :33-95 (3-axis variant of the 2-axis cases + the exact no-collision sweep)
    r1 = (1.0, 1.0, 1.0)
    h = lambda x, y, t=0.0, r=r1: orc.spatial_hash_xyt(x, y, t, r)
    assert h(10.3, 5.0) == h(10.0, 5.0) == h(10.0, 5.3) == h(10.1, 5.1)
    assert h(10.3, 5.0) != h(9.1, 5.1) and h(10.3, 5.0) != h(10.1, 4.1)
    r01 = (0.1, 0.1, 1.0)
    assert h(10.3, 5.13, 0, r01) == h(10.33, 5.14, 0, r01) != h(10.0, 5.0, 0, r01)
    assert h(-10.3, -2.13, 0, r01) == h(-10.27, -2.14, 0, r01) != h(-10.0, -2.0, 0, r01)


def test_spatial_hash_no_collisions():  # :73-95 — all 201^3 integer triples hash distinctly
    kFib = np.uint64(11400714819323198485)
    v = np.arange(-100, 101, dtype=np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        hv = v * kFib

        def rotl(x, s):
            return (x << np.uint64(s)) | (x >> np.uint64(64 - s))

        hx, hy, hz = hv, rotl(hv, 21), rotl(hv, 42)
        allh = (hx[:, None, None] ^ hy[None, :, None] ^ hz[None, None, :]).ravel()
    assert len(np.unique(allh)) == 201 ** 3
    # and the numpy formula above IS the oracle's hash
    for (x, y, t) in [(-100, 3, 57), (0, 0, 0), (100, -100, 99), (-1, -1, -1)]:
        ix, iy, it = x + 100, y + 100, t + 100
        assert int(hx[ix] ^ hy[iy] ^ hz[it]) == orc.spatial_hash_xyt(float(x), float(y), float(t), (1.0, 1.0, 1.0))


# ---- algorithm/test_effective_sample_size.cpp ----------------------------------------------------
@pytest.mark.parametrize("w,expected", [
    ([], 0.0), ([0.0] * 5, 0.0), ([1.0] * 5, 5.0), ([0.1] * 5, 5.0), ([100.0] * 5, 5.0), ([1.0, 0.0], 1.0),
    ([1.0, 0.0, 0.0], 1.0), ([1.0, 1.0, 0.0], 2.0), ([1.0, 0.5, 0.0], 1.8), ([1.0, 0.5, 0.5], 2.66),
])
def test_effective_sample_size(w, expected):  # :23-79
    assert orc.effective_sample_size(np.array(w, dtype=np.float64)) == pytest.approx(expected, abs=0.01)


# ---- algorithm/test_thrun_recovery_probability_estimator.cpp -------------------------------------
def test_thrun():  # :46-97
    assert orc.Thrun(0.2, 0.4)([]) == 0.0
    assert orc.Thrun(0.2, 0.4)([0.0, 0.0]) == 0.0
    est = orc.Thrun(0.5, 1.0)
    assert est([1.0, 2.0, 3.0]) == 0.0
    assert est([0.5, 1.0, 1.5]) == pytest.approx(0.33, abs=0.01)
    assert est([0.5, 1.0, 1.5]) == pytest.approx(0.20, abs=0.01)
    est.reset()
    assert est([0.5, 1.0, 1.5]) == 0.0
    for w0, w1, p in [(1.0, 1.5, 0.00), (1.0, 2.0, 0.00), (1.0, 0.5, 0.05), (0.5, 0.1, 0.08), (0.5, 0.0, 0.10)]:
        est = orc.Thrun(0.001, 0.1)
        assert est([w0]) == pytest.approx(0.0, abs=0.01)
        assert est([w1]) == pytest.approx(p, abs=0.01)


# ---- algorithm/test_estimation.cpp ---------------------------------------------------------------
def _states(lst):
    return np.array([orc.se2(x, y, th) for (th, x, y) in lst])


def _check_estimate(states, weights, mean_theta, mean_xy, cov_rows, tol=0.001):
    mean, cov = orc.estimate(states, weights)
    assert math.atan2(mean[1], mean[0]) == pytest.approx(mean_theta, abs=tol)
    assert mean[2] == pytest.approx(mean_xy[0], abs=tol) and mean[3] == pytest.approx(mean_xy[1], abs=tol)
    np.testing.assert_allclose(cov, np.array(cov_rows), atol=tol)
    return mean, cov


def test_estimation_golden():  # :129-244
    pi = math.pi
    _check_estimate(_states([(0.0, 1.0, 2.0), (0.0, 0.0, 0.0)]), [1.0, 1.0], 0.0, (0.5, 1.0),
                    [[0.5, 1.0, 0], [1.0, 2.0, 0], [0, 0, 0]])
    _check_estimate(_states([(-pi / 2, 0, 0), (0.0, 0, 0)]), [1.0, 1.0], -pi / 4, (0, 0),
                    [[0, 0, 0], [0, 0, 0], [0, 0, 0.693]])
    _check_estimate(_states([(pi / 6, 0.0, -3.0), (pi / 2, 1.0, -2.0), (pi / 3, 2.0, -1.0), (0.0, 3.0, 0.0)]), [1.0] * 4,
                    pi / 4, (1.5, -1.5), [[1.666, 1.666, 0], [1.666, 1.666, 0], [0, 0, 0.357]])
    mean, cov = orc.estimate(_states([(pi / 2, 0, 0), (-pi / 2, 0, 0)]), [1.0, 1.0])
    assert cov[2, 2] == math.inf and math.atan2(mean[1], mean[0]) == 0.0
    walk = [(pi * 0.1, 0.0, -2.0), (pi * 0.2, 1.0, -1.0), (pi * 0.3, 2.0, 1.0), (pi * 0.2, 3.0, 2.0), (pi * 0.2, 2.0, 1.0),
            (pi * 0.2, 1.0, -1.0), (pi * 0.3, 2.0, -2.0), (pi * 0.4, 3.0, -1.0), (pi * 0.5, 2.0, 1.0), (pi * 0.4, 1.0, 2.0)]
    _check_estimate(_states(walk), [1.0] * 10, 0.8762, (1.7, 0.0),
                    [[0.9000, 0.5556, 0], [0.5556, 2.4444, 0], [0, 0, 0.1355]])
    _check_estimate(_states([(pi / 6, 0.0, -3.0), (pi / 2, 1.0, -2.0), (pi / 3, 2.0, -1.0), (pi / 2, 1.0, -2.0)]),
                    [0.0, 1.0, 0.0, 1.0], pi / 2, (1.0, -2.0), [[0, 0, 0], [0, 0, 0], [0, 0, 0]])
    _check_estimate(_states(walk), [0.1, 0.4, 0.7, 0.1, 0.9, 0.2, 0.2, 0.4, 0.1, 0.4], 0.8687, (1.8, 0.3143),
                    [[0.5946, 0.0743, 0], [0.0743, 1.8764, 0], [0, 0, 0.0855]])


# ---- views/test_sample.cpp, views/test_random_intersperse.cpp (distributional pins) ---------------
def test_multinomial_frequencies():  # test_sample.cpp:137-163 (tol 0.01 over 100 000 draws) + :89-103
    states = np.array([orc.se2(float(i), 0.0, 0.0) for i in range(4)])
    w = np.array([0.1, 0.4, 0.0, 0.5])
    out, anc = orc.resample(states, w, 100_000, 100_000, 0.05, 3.0, (0.5, 0.5, 0.17), 0.0, seed=99, step=1)
    assert len(out) == 100_000
    freq = np.bincount(anc, minlength=4) / len(anc)
    np.testing.assert_allclose(freq, [0.1, 0.4, 0.0, 0.5], atol=0.01)
    assert freq[2] == 0.0  # weight-zero never drawn


def test_random_intersperse_rate_and_first_element():  # test_random_intersperse.cpp:87-93,152-158
    states = np.array([orc.se2(1.0, 1.0, 0.0)])
    free = np.array([[50.0, 50.0]])
    for seed in range(20):
        out, anc = orc.resample(states, [1.0], 1000, 1000, 0.05, 3.0, (0.5, 0.5, 0.17), 0.9, seed=seed, step=1, free_xy=free)
        assert anc[0] == 0  # first element always from the source
    out, anc = orc.resample(states, [1.0], 100_000, 100_000, 0.05, 3.0, (0.5, 0.5, 0.17), 0.25, seed=5, step=2, free_xy=free)
    assert (anc == -1).mean() == pytest.approx(0.25, abs=0.01)


def test_normalize_semantics():  # actions/test_normalize.cpp:27-99
    w, s = orc.normalize([1.0, 2.0, 3.0, 4.0])
    assert s == 10.0 and np.allclose(w, [0.1, 0.2, 0.3, 0.4])
    w, s = orc.normalize([0.25, 0.25, 0.5])  # already normalised: untouched
    assert list(w) == [0.25, 0.25, 0.5]


# ---- "next" rows of SURVEY.md 8(f): the other model closures of beluga_ros::Amcl's variant set -------------------
def test_lf_prob_importance_weight():  # sensor/test_likelihood_field_prob_model.cpp:35-76
    field = orc.make_likelihood_field(CENTER, 0.5, LF_PARAMS)
    w = lambda pts, state=IDENTITY: orc.lf_prob_weights(field, 0.5, IDENTITY, LF_PARAMS[1], [state], pts)[0]
    assert w([(1.25, 1.25)]) == pytest.approx(1.022, abs=0.003)
    assert w([(2.25, 2.25)]) == pytest.approx(0.025, abs=0.003)
    assert w([(-50.0, 50.0)]) == pytest.approx(0.050, abs=0.003)
    assert w([(1.20, 1.20), (1.25, 1.25), (1.30, 1.30)]) == pytest.approx(1.068, abs=0.01)
    assert w([(0.0, 0.0)], orc.se2(1.25, 1.25, 0.0)) == pytest.approx(1.022, abs=0.003)


def _omni(control, prev, state, alphas=(0.0,) * 5, n=1, seed=7):
    return orc.propagate_kind(np.tile(state, (n, 1)), "omnidirectional", control, prev, alphas, seed=seed, step=1)


def test_omnidirectional_noise_free():  # motion/test_omnidirectional_drive_model.cpp:53-100
    pi = math.pi
    pose = orc.se2(2.0, 5.0, pi / 3)
    _se2_near(_omni(orc.se2(1.0, -2.0, pi), orc.se2(1.0, -2.0, pi), pose)[0], pose, 1e-3)
    c, p = orc.se2(1.0, 0.0, 0.0), orc.se2(0.0, 0.0, 0.0)
    _se2_near(_omni(c, p, orc.se2(2.0, 0.0, 0.0))[0], orc.se2(3.0, 0.0, 0.0), 1e-3)
    _se2_near(_omni(c, p, orc.se2(0.0, 3.0, 0.0))[0], orc.se2(1.0, 3.0, 0.0), 1e-3)
    c = orc.se2(0.0, 1.0, pi / 2)
    _se2_near(_omni(c, p, orc.se2(0.0, 0.0, 0.0))[0], orc.se2(0.0, 1.0, pi / 2), 1e-3)
    _se2_near(_omni(c, p, orc.se2(2.0, 3.0, -pi / 2))[0], orc.se2(3.0, 3.0, 0.0), 1e-3)
    c = orc.se2(0.0, 0.0, pi / 4)
    _se2_near(_omni(c, p, orc.se2(0.0, 0.0, pi))[0], orc.se2(0.0, 0.0, pi * 5 / 4), 1e-3)
    _se2_near(_omni(c, p, orc.se2(0.0, 0.0, -pi / 2))[0], orc.se2(0.0, 0.0, -pi / 4), 1e-3)
    _se2_near(_omni(orc.se2(0.0, 1.0, 0.0), p, orc.se2(0.0, 0.0, 0.0))[0], orc.se2(0.0, 1.0, 0.0), 1e-3)  # TranslateStrafe


def test_omnidirectional_statistics():  # :116-178
    pi, alpha, zero = math.pi, 0.2, orc.se2(0, 0, 0)
    out = _omni(orc.se2(3.0, 0, 0), zero, orc.se2(5.0, 0, 0), (0, 0, alpha, 0, 0), n=100_000, seed=3)
    assert out[:, 2].mean() == pytest.approx(8.0, abs=0.015) and out[:, 2].std() == pytest.approx(math.sqrt(alpha * 9.0), abs=0.015)
    out = _omni(orc.se2(0, 0, pi / 4), zero, orc.se2(0, 0, pi / 6), (alpha, 0, 0, 0, 0), n=100_000, seed=4)
    th = np.arctan2(out[:, 1], out[:, 0])
    assert th.mean() == pytest.approx(pi / 6 + pi / 4, abs=0.01) and th.std() == pytest.approx(math.sqrt(alpha * (pi / 4) ** 2), abs=0.01)
    out = _omni(orc.se2(0, 0, -pi * 3 / 4), zero, orc.se2(0, 0, pi / 6), (alpha, 0, 0, 0, 0), n=100_000, seed=5)
    th = np.arctan2(out[:, 1], out[:, 0])
    assert th.mean() == pytest.approx(pi / 6 - pi * 3 / 4, abs=0.01) and th.std() == pytest.approx(math.sqrt(alpha * (pi / 4) ** 2), abs=0.01)


def test_stationary_model_statistics():  # motion/stationary_model.hpp:55-61: N(0, 0.02) on (theta, x, y), control ignored
    start = orc.se2(1.0, -2.0, 0.5)
    out = orc.propagate_kind(np.tile(start, (100_000, 1)), "stationary", orc.se2(9, 9, 1), orc.se2(0, 0, 0), (0.0,) * 5, seed=11, step=2)
    local = np.array([orc.se2_mul(orc.se2_inverse(start), s) for s in out[:2000]])
    assert abs(local[:, 2].mean()) < 0.002 and local[:, 2].std() == pytest.approx(0.02, abs=0.002)
    assert abs(local[:, 3].mean()) < 0.002 and local[:, 3].std() == pytest.approx(0.02, abs=0.002)
    th = np.arctan2(out[:, 1], out[:, 0]) - 0.5
    assert abs(th.mean()) < 0.001 and th.std() == pytest.approx(0.02, abs=0.001)


# ---- caller-side scan preparation ------------------------------------------------------------
@pytest.mark.parametrize("size,count,expected", [
    (0, 0, []), (0, 1, []), (4, 0, []), (4, 1, [1]), (4, 10, [1, 2, 3, 4]), (4, 2, [1, 4]), (5, 3, [1, 3, 5]), (6, 3, [1, 4, 6]),
    (9, 3, [1, 5, 9]), (4, 3, [1, 3, 4]), (10, 6, [1, 3, 5, 7, 9, 10]),
])
def test_take_evenly(size, count, expected):  # beluga/test/beluga/views/test_take_evenly.cpp:72-147 (1-based values there)
    assert [i + 1 for i in orc.take_evenly_indices(size, count)] == expected


def test_laser_scan_preparation():  # beluga_ros/test/test_laser_scan.cpp:30-89 + sensor/data/laser_scan.hpp:64-90
    pts = orc.prepare_laser_scan([1.0, 2.0, 3.0], 0.0, 0.1, 0.0, 100.0)
    ang = np.arctan2(pts[:, 1], pts[:, 0])
    np.testing.assert_allclose(ang, [0.0, 0.1, 0.2], atol=0.001)  # AngleIncrements
    np.testing.assert_allclose(np.hypot(pts[:, 0], pts[:, 1]), [1.0, 2.0, 3.0], rtol=1e-12)
    assert len(orc.prepare_laser_scan([1.0, 2.0, 3.0], 0.0, 0.1, 0.0, 100.0, max_beams=2)) == 2  # LimitMaxBeams
    # message limits and constructor limits combine (MinMaxRangeFrom*): [max(10,15), min(100,95)]
    pts = orc.prepare_laser_scan([12.0, 20.0, 97.0, np.nan, 50.0], 0.0, 0.1, 10.0, 100.0, min_range=15.0, max_range=95.0)
    np.testing.assert_allclose(np.hypot(pts[:, 0], pts[:, 1]), [20.0, 50.0], rtol=1e-12)
    # laser origin: 90 deg yaw + offset, z ignored
    q = (0.0, 0.0, math.sin(math.pi / 4), math.cos(math.pi / 4), 0.5, -0.25, 0.3)
    pts = orc.prepare_laser_scan([2.0], 0.0, 0.1, 0.0, 100.0, origin_se3=q)
    np.testing.assert_allclose(pts[0], [0.5, 1.75], atol=1e-12)


# ---- cluster_based_estimate (algorithm/test_cluster_based_estimation.cpp) ---------------------
def _multicluster(xmin, xmax, ymin, ymax, step):  # make_particle_multicluster_dataset :67-94
    xw, yw = xmax - xmin, ymax - ymin
    states, weights = [], []
    x = step / 2.0
    while x <= xw:
        y = step / 2.0
        while y <= yw:
            k = (0.0 if 2 * x < xw else 1.0) + (0.0 if 2 * y < yw else 2.0) + 1.0
            wt = abs(math.sin(2.0 * math.pi * x / xw)) * abs(math.sin(2.0 * math.pi * y / yw)) * k
            states.append(orc.se2(x + xmin, y + ymin, 0.0))
            weights.append(max(0.0, wt - k / 2.0))
            y += step
        x += step
    return np.array(states), np.array(weights)


def test_cluster_state_estimation_step():  # :291-315 — four clusters, means at the four peaks
    states, w = _multicluster(0.0, 36.0, 0.0, 36.0, 1.0)
    ids = orc.cluster_ids(states, w, 1.0, math.pi / 2.0, 0.9)
    ests = []
    for cid in np.unique(ids):
        sel = ids == cid
        if sel.sum() > 1:
            mean, _ = orc.estimate(states[sel], w[sel])
            ests.append((w[sel].sum(), mean[2], mean[3]))
    assert len(ests) == 4
    ests.sort()
    for (_, x, y), (ex, ey) in zip(ests, [(9.0, 9.0), (27.0, 9.0), (9.0, 27.0), (27.0, 27.0)]):
        assert x == pytest.approx(ex, abs=1e-6) and y == pytest.approx(ey, abs=1e-6)


def test_cluster_estimation_heaviest_and_nightmare():  # :357-415
    states, w = _multicluster(-2.0, 2.0, -2.0, 2.0, 0.025)
    sel = (states[:, 2] >= 0.0) & (states[:, 3] >= 0.0)
    exp_mean, exp_cov = orc.estimate(states[sel], w[sel])
    mean, cov = orc.cluster_based_estimate(states, w)
    np.testing.assert_allclose(mean, exp_mean, atol=1e-6)
    np.testing.assert_allclose(cov, exp_cov, atol=0.001)
    far = np.array([orc.se2(-10, -10, 0), orc.se2(-10, 10, 0), orc.se2(10, -10, 0), orc.se2(10, 10, 0)])
    fw = np.full(4, 0.2)
    exp_mean, exp_cov = orc.estimate(far, fw)
    mean, cov = orc.cluster_based_estimate(far, fw)
    np.testing.assert_allclose(mean, exp_mean, atol=1e-6)
    np.testing.assert_allclose(cov, exp_cov, atol=0.001)


def test_frozen_update_cycles_fixture_is_reproduced():
    """tests/golden/update_cycles_config1.npz (20 cycles of BASELINE config 1: KLD + recovery on the turtlebot3 grid) is the
    oracle's frozen answer for the end-to-end level, which the reference pins only with smoke tests: the oracle must keep
    reproducing it bit for bit."""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_cycle_fixture", os.path.join(here, "golden", "make_cycle_fixture.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = np.load(gen.OUT)
    cells, res, origin, truth, steps = gen.scenario()
    f = orc.Amcl(seed=gen.SEED, **gen.PARAMS)
    f.set_map(cells, res, origin)
    f.initialize(truth, np.diag([0.25, 0.25, 0.0685]))
    outs = [f.update(c, p) for c, p in steps]
    assert [o is not None for o in outs] == list(want["updated"])
    assert np.array_equal(np.array([o[0] for o in outs if o is not None]), want["poses"])
    assert np.array_equal(np.array([o[1] for o in outs if o is not None]), want["covs"])
    assert np.array_equal(f.particles()[0], want["final_states"])
