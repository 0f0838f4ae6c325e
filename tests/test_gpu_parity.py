"""GPU parity: every stage of the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (stated once, used below):
  * integer / index work (cells, hashes, KLD cut length, resample counts): bit-exact;
  * f64 stage outputs: relative 1e-12 (the device sums beams in a different association than the
    reference's sequential transform_reduce, and libm implementations differ by an ulp);
  * multinomial ancestors: identical except where the uniform lands within 1e-9 of a CDF step (parallel
    prefix sums round differently from std::partial_sum) — such flips are counted and bounded.
"""
import math
import os

import numpy as np
import pytest

from beluga_amd import synth
from beluga_amd.amcl import (Amcl, AmclParams, BeamModelParam, DifferentialDriveModelParam, LikelihoodFieldModelParam,
                             OccupancyGrid, se2_from_xytheta)
from oracle import binding as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-12
MOTION = DifferentialDriveModelParam(0.1, 0.05, 0.1, 0.05)
MOTION_T = (0.1, 0.05, 0.1, 0.05)
LF = LikelihoodFieldModelParam(max_obstacle_distance=2.0, max_laser_distance=100.0, z_hit=0.5, z_random=0.5, sigma_hit=0.2,
                               model_unknown_space=True)
LF_T = (2.0, 100.0, 0.5, 0.5, 0.2)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rooms_grid(size=400, seed=3):
    cells = synth.make_rooms_map(size, size, seed=seed, n_rooms=12)
    return OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-size * 0.025, -size * 0.025, 0.0))


def turtlebot_grid():
    z = np.load(os.path.join(GOLDEN, "turtlebot3_world_grid.npz"))
    ox, oy, ot = z["origin_xytheta"]
    return OccupancyGrid(cells=z["cells"], resolution=float(z["resolution"]), origin=se2_from_xytheta(ox, oy, ot))


def make_scan(grid, pose, beams, max_range=30.0, fov=270.0, seed=1):
    angles = synth.lidar_angles(beams, fov)
    ranges = synth.cast_scan(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), pose, angles, max_range, 0.01, seed)
    return synth.scan_points(ranges, angles)


def new_filter(grid, n, sensor=LF, **kw):
    """The ordered kernels engage from 16 384 particles here (option lf_small_particles pinned to the ordering threshold: the
    library's own crossover to them, 65 536 particles, is exercised by test_mid_size_sets_take_the_kernel_for_small_sets)."""
    small = kw.pop("lf_small_particles", 16_384)
    params = AmclParams(min_particles=kw.pop("min_particles", n), max_particles=n, **kw)
    f = Amcl(grid, MOTION, sensor, params, seed=11)
    f.set_option("lf_small_particles", small)
    return f


# ---------------------------------------------------------------------------------------------------
def test_field_build_bit_exact():
    """mcl_set_map's field == oracle make_likelihood_field (likelihood_field_model_base.hpp:130-185), bit for bit."""
    for grid, lf, lft in [
        (rooms_grid(), LF, LF_T),
        (turtlebot_grid(), LF, LF_T),
        (turtlebot_grid(), LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True, True), LF_T),
        (rooms_grid(160, 5), LikelihoodFieldModelParam(100.0, 2.0, 0.5, 0.5, 0.2, False, False), (100.0, 2.0, 0.5, 0.5, 0.2)),
    ]:
        f = Amcl(grid, MOTION, lf, AmclParams(max_particles=64), seed=1)
        got = f.likelihood_field()
        want = orc.make_likelihood_field(grid.cells, grid.resolution, lft, lf.model_unknown_space, lf.only_obstacle_boundaries)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        f.close()


def test_reference_golden_weights_through_the_c_abi():
    """test_likelihood_field_model.cpp:34-74,160-197 and test_beam_model.cpp:40-82 replayed on the GPU path."""
    F, T = 0, 100
    center = np.array([F] * 12 + [T] + [F] * 12, dtype=np.int8).reshape(5, 5)
    grid = OccupancyGrid(center, 0.5)
    lf = LikelihoodFieldModelParam(2.0, 20.0, 0.5, 0.5, 0.2)
    ident = np.array([[1.0, 0.0, 0.0, 0.0]])

    def lf_w(points, state=ident):
        f = Amcl(grid, MOTION, lf, AmclParams(max_particles=4), seed=1)
        f.set_particles(state, [1.0])
        f.reweight(points)
        w = f.particles()[1][0]
        f.close()
        return w

    assert lf_w([(1.25, 1.25)]) == pytest.approx(2.068, abs=0.003)
    assert lf_w([(2.25, 2.25)]) == pytest.approx(1.000, abs=0.003)
    assert lf_w([(-50.0, 50.0)]) == pytest.approx(1.000, abs=0.003)
    assert lf_w([(1.20, 1.20), (1.25, 1.25), (1.30, 1.30)]) == pytest.approx(4.205, abs=0.01)
    assert lf_w([(0.0, 0.0)], np.array([[1.0, 0.0, 1.25, 1.25]])) == pytest.approx(2.068, abs=0.003)
    assert lf_w([(1.0, 1.0)]) == pytest.approx(2.068577607986223, abs=1e-6)

    beam = BeamModelParam(0.5, 0.05, 0.05, 0.5, 0.2, 0.1, 60.0)

    def beam_w(points):
        f = Amcl(grid, MOTION, beam, AmclParams(max_particles=4), seed=1)
        f.set_particles(ident, [1.0])
        f.reweight(points)
        w = f.particles()[1][0]
        f.close()
        return w

    assert beam_w([(1.0, 1.0)]) == pytest.approx(1.0171643824743635, abs=1e-6)
    assert beam_w([(0.75, 0.75)]) == pytest.approx(0.015905891701088148, abs=1e-6)
    assert beam_w([(2.25, 2.25)]) == pytest.approx(0.000, abs=1e-6)
    assert beam_w([(60.0, 60.0)]) == pytest.approx(0.00012500000000000003, abs=1e-6)


@pytest.mark.parametrize("variant", ["0", "1", "2", "3"])
@pytest.mark.parametrize("beams", [1, 63, 64, 65, 180, 1080])
def test_reweight_lf_matches_oracle(variant, beams):
    """All kernel variants (wave per particle / lane per particle / spatially ordered lanes = the default / wave per particle
    with a lane per beam over the palette table = the one for dispersed sets)."""
    grid = rooms_grid()
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, beams, max_range=12.0)
    n = 4097 if beams != 1080 else 1500  # ragged tail tile
    if variant == "2":
        n = 20_001  # the binned variant only engages above 16384 particles
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=5)
    states[:8, 2] += 100.0  # some particles far outside the map: every beam out of grid
    w0 = np.random.Generator(np.random.MT19937(3)).uniform(0.5, 1.5, n)
    f = new_filter(grid, n)
    f.set_option("lf_variant", int(variant))
    f.set_particles(states, w0)
    f.reweight(pts)
    _, got = f.particles()
    field = f.likelihood_field()
    want = w0 * orc.lf_weights(field, grid.resolution, grid.origin, LF.max_laser_distance, states, pts)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)
    unknown = float(np.float32(1.0 / 100.0))  # float(1 / max_laser_distance), likelihood_field_model.hpp:75
    assert got[0] == pytest.approx(w0[0] * (1.0 + beams * unknown ** 3), rel=1e-12)  # every beam out of the grid
    f.close()


def test_mid_size_sets_take_the_kernel_for_small_sets():
    """Default options: likelihood-field sets below 65 536 particles go to k_reweight_lf_beams (a wave per few particles, the
    lanes over the beams, no ordering pass - faster than the ordered kernels up to there), sets from there on to the ordered
    kernels; same weights up to the rounding of the lane sums."""
    grid = rooms_grid()
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, 360, max_range=12.0)
    for n, beams_kernel in ((30_000, True), (65_535, True), (65_536, False)):
        states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=5)
        f = Amcl(grid, MOTION, LF, AmclParams(min_particles=n, max_particles=n), seed=11)
        f.set_particles(states, np.ones(n))
        f.reweight(pts)
        assert f.counter("lf_beams_launches") == (1 if beams_kernel else 0)
        assert f.counter("lf_fast_launches") == (0 if beams_kernel else 1)
        sample = np.random.Generator(np.random.MT19937(2)).choice(n, 2048, replace=False)
        want = orc.lf_weights(f.likelihood_field(), grid.resolution, grid.origin, LF.max_laser_distance, states[sample], pts)
        np.testing.assert_allclose(f.particles()[1][sample], want, rtol=RTOL)
        f.close()


@pytest.mark.parametrize("case", ["rooms", "rotated_ragged", "turtlebot", "unknown_space"])
def test_far_tile_bitmap_changes_no_weight(case):
    """The gather kernel with the bitmap of far tiles (8x8-cell tiles uniformly at the field's most common value; look-ups into
    them skip the table) against the same kernel without it: bit for bit, on particles spread over and beyond the map with any
    heading - grids whose sides are no multiples of 8, a rotated origin, the unknown-space overlay (where the common value
    may be another one) - and against the oracle.  The same under the position-major ordering key that dispersed sets get."""
    sensor = LF
    if case == "rooms":
        grid = rooms_grid()
    elif case == "rotated_ragged":
        grid = OccupancyGrid(synth.make_rooms_map(203, 157, seed=9, n_rooms=5), 0.1, origin=se2_from_xytheta(3.0, -2.0, 0.7))
    elif case == "turtlebot":
        grid = turtlebot_grid()
    else:
        cells = synth.make_rooms_map(333, 250, seed=4, n_rooms=7).copy()
        cells[:, 200:] = -1  # a third of the map unknown
        grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-4.0, 1.0, -0.3))
        sensor = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
    H, W = grid.cells.shape
    n = 21_013  # 42 workgroups of the gather kernel: its far-tile form pads the grid to 48 and hands each XCD a contiguous run
    rng = np.random.Generator(np.random.MT19937(17))
    gx = rng.uniform(-0.15 * W, 1.15 * W, n) * grid.resolution  # in the grid frame, beyond its borders too
    gy = rng.uniform(-0.15 * H, 1.15 * H, n) * grid.resolution
    oc, os_, ox, oy = grid.origin
    x, y = ox + oc * gx - os_ * gy, oy + os_ * gx + oc * gy
    th = rng.uniform(-np.pi, np.pi, n)
    states = np.stack([np.cos(th), np.sin(th), x, y], axis=1)
    pts = synth.scan_points(rng.uniform(0.3, 9.0, 363), synth.lidar_angles(363, 300.0))
    f = new_filter(grid, n, sensor=sensor)
    f.set_option("lf_patch", 0)
    f.set_option("lf_far_tiles", 0)
    f.set_option("lf_dispersed", 0)  # the lane-per-particle gather kernel, with and without the bitmap: bit for bit
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    plain = f.particles()[1]
    assert f.counter("lf_far_launches") == 0 and f.counter("lf_fast_launches") == 1
    f.set_option("lf_far_tiles", 2)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    assert f.counter("lf_far_tiles") > 0 and f.counter("lf_far_launches") == 1
    assert np.array_equal(f.particles()[1], plain)
    # the position-major ordering key of dispersed sets (forced here): another order of the lanes, the same weights
    f.set_option("key_layout", 1)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    assert f.counter("lf_far_launches") == 2 and np.array_equal(f.particles()[1], plain)
    perm, keys = f.debug_order()
    assert np.array_equal(np.sort(perm), np.arange(n)) and np.all(np.diff(keys[perm].astype(np.int64)) >= 0)
    want = orc.lf_weights(f.likelihood_field(), grid.resolution, grid.origin, sensor.max_laser_distance, states, pts,
                          threads=orc.max_threads())
    np.testing.assert_allclose(plain, want, rtol=RTOL)
    # the default for dispersed sets (lf_dispersed = 2): the lanes over the beams of a pose (k_reweight_lf_far_beams; 363 beams: five full
    # rounds of 64 and one of 43), the bitmap by the tiles' linear index, the poses in either order - a lane adds its beams in scan
    # order and the lane sums are added in a tree: the same weights up to rounding, whatever the order of the poses (bit for bit)
    f.set_option("lf_dispersed", 2)
    got = []
    for layout, per_wave in ((1, 0), (0, 0), (1, 5)):
        f.set_option("key_layout", layout)
        f.set_option("lf_far_beams_per_wave", per_wave)
        before = f.counter("lf_far_beams_launches")
        f.set_particles(states, np.ones(n))
        f.reweight(pts)
        assert f.counter("lf_far_beams_launches") == before + 1
        got.append(f.particles()[1])
        np.testing.assert_allclose(got[-1], plain, rtol=1e-13)
        np.testing.assert_allclose(got[-1], want, rtol=RTOL)
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
    f.close()


def test_patch_probes_skip_hopelessly_sparse_sets():
    """A set reported as dispersed is probed with the LDS-patch kernel every 16th launch (has it converged?) - unless the last
    estimate says its poses are too sparse for any workgroup to fit a patch (the probe costs three far-tile launches)."""
    grid = rooms_grid()
    n = 40_000
    pts = make_scan(grid, synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6),
                    180, max_range=8.0)
    # a wide cloud the library knows nothing about: the patch kernel runs, reports that next to nothing fits, the next 15
    # launches gather (far-tile form), the 16th after the verdict is a probe
    f = new_filter(grid, n)
    f.set_particles(synth.normal_particles(n, (0.0, 0.0, 0.3), (3.0, 3.0, 1.5), seed=4), np.ones(n))
    for _ in range(16):
        f.reweight(pts)
        f.sync()
    assert f.counter("lf_patch_launches") == 1 and f.counter("lf_far_launches") == 15
    f.reweight(pts)
    assert f.counter("lf_patch_launches") == 2 and f.counter("lf_far_launches") == 15
    f.close()
    # initialize_from_map says what the set looks like (uniform over the map, every heading): too sparse here, never probed
    f = new_filter(grid, n)
    f.initialize_from_map()
    for _ in range(20):
        f.reweight(pts)
    assert f.counter("lf_patch_launches") == 0 and f.counter("lf_far_launches") == 20
    f.close()
    # full cycles over a uniform set (no resampling, empty scans: the weights stay uniform): every estimate says "sparse"
    f = new_filter(grid, n, resample_interval=1000, update_min_d=0.0, update_min_a=0.0)
    f.initialize_from_map()
    for c in range(20):
        assert f.update(se2_from_xytheta(0.01 * c, 0.0, 0.0), np.zeros((0, 2))) is not None
    assert f.counter("lf_patch_launches") == 0 and f.counter("lf_far_launches") == 20
    # a dense set: the verdict is reset, the patch kernel runs and stays
    f.set_particles(synth.normal_particles(n, (0.0, 0.0, 0.3), (0.3, 0.3, 0.1), seed=4), np.ones(n))
    for c in range(3):
        f.reweight(pts)
    assert f.counter("lf_patch_launches") == 3
    f.close()


def test_reweight_lf_rotated_origin_and_empty_scan():
    cells = synth.make_rooms_map(200, 150, seed=9, n_rooms=6)
    grid = OccupancyGrid(cells, 0.1, origin=se2_from_xytheta(3.0, -2.0, 0.7))
    states = synth.normal_particles(1000, (6.0, 5.0, 1.0), (2.0, 2.0, 1.0), seed=8)
    pts = synth.scan_points(np.linspace(0.5, 8.0, 360), synth.lidar_angles(360, 360.0))
    f = new_filter(grid, 1000)
    f.set_particles(states, np.ones(1000))
    f.reweight(pts)
    got = f.particles()[1]
    want = orc.lf_weights(f.likelihood_field(), 0.1, grid.origin, LF.max_laser_distance, states, pts)
    np.testing.assert_allclose(got, want, rtol=RTOL)
    f.reweight(np.zeros((0, 2)))  # empty measurement: transform_reduce over nothing = 1.0
    np.testing.assert_allclose(f.particles()[1], want, rtol=RTOL)
    f.close()


@pytest.mark.parametrize("n,beams", [(777, 181), (5_000, 181), (20_001, 61)])
def test_reweight_beam_matches_oracle(n, beams):
    """The wave-per-particle kernel over the whole-grid maps (777 particles), the ordered-lanes kernel with its scan split into
    segments (5000 particles, forced below its threshold; 20 001 by default)."""
    grid = rooms_grid(300, 4)
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, beams, max_range=10.0, fov=360.0)
    states = synth.normal_particles(n, truth, (0.3, 0.3, 0.2), seed=5)
    states[:3, 2] += 50.0  # source cell outside the grid: no trace at all
    beam = BeamModelParam(beam_max_range=10.0)
    f = new_filter(grid, n, sensor=beam)
    if n == 5_000:
        f.set_option("beam_sort_min_particles", 0)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    got = f.particles()[1]
    want = orc.beam_weights(grid.cells, grid.resolution, grid.origin, (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 10.0), states, pts)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)  # erf/exp differ by ulps between libms
    f.close()


@pytest.mark.parametrize("kind", ["empty", "salt", "stripes", "unknown_frame", "rotated_rooms"])
def test_beam_walk_closed_forms_on_adversarial_maps(kind):
    """The ordered beam kernel's empty-space skips (block distance map, closed-form Bresenham state), its head / tail shortcuts
    and the walks over the whole-grid maps (rays that leave a workgroup's 1024^2 LDS window, particles outside it altogether),
    on maps built to stress them: nothing at all (every ray runs to its end or off the grid in the longest skips), salt noise
    (distance 0 / 1 nearly everywhere: cell-by-cell examination), thin diagonal stripes, an unknown frame around free space, and
    a rooms map behind a rotated origin.  20 000 particles spread over tens of metres.  Weights against the oracle's cell-by-cell
    Bresenham (raycasting.hpp:97-107, bresenham.hpp:122-160) and the number of cells visited EXACTLY (an integer result)."""
    rng = np.random.Generator(np.random.MT19937(7))
    W, H, res = 1500, 1200, 0.05
    origin = se2_from_xytheta(-37.5, -30.0, 0.0)
    cells = np.zeros((H, W), dtype=np.int8)
    if kind == "salt":
        cells[rng.random((H, W)) < 0.003] = 100
    elif kind == "stripes":
        yy, xx = np.mgrid[0:H, 0:W]
        cells[((xx + yy) % 97 == 0) & ((xx // 61 + yy // 53) % 3 != 0)] = 100
    elif kind == "unknown_frame":
        cells[:] = -1
        cells[150:-150, 200:-200] = 0
        cells[600, 300:900] = 100
    elif kind == "rotated_rooms":
        cells = synth.make_rooms_map(W, H, seed=9, n_rooms=40)
        origin = se2_from_xytheta(-20.0, -45.0, 0.6)
    grid = OccupancyGrid(cells=cells, resolution=res, origin=origin)
    centre = (0.0, 0.0, 0.3)
    beams, max_range = 61, 40.0
    pts = make_scan(grid, centre, beams, max_range=max_range, fov=360.0)
    pts[::7] *= 3.0  # some measured ranges far beyond what the map returns
    n = 20_000
    states = synth.normal_particles(n, centre, (8.0, 8.0, 1.5), seed=5)
    states[:5, 2] += 500.0  # source cells outside the grid
    beam = BeamModelParam(beam_max_range=max_range)
    f = new_filter(grid, n, sensor=beam)
    f.set_particles(states, np.ones(n))
    f.beam_cells_visited(reset=True)
    f.reweight(pts)
    got = f.particles()[1]
    visited = f.beam_cells_visited()
    want, steps = orc.beam_weights(grid.cells, res, grid.origin, (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, max_range), states, pts,
                                   threads=orc.max_threads(), return_steps=True)
    assert visited == steps
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
    f.close()
    # the kernel of small sets (a wave per particle, the same closed forms over the whole-grid maps): the first 3000 particles
    m = 3000
    f = new_filter(grid, m, sensor=beam)
    f.set_particles(states[:m], np.ones(m))
    f.beam_cells_visited(reset=True)
    f.reweight(pts)
    got_small, visited_small = f.particles()[1], f.beam_cells_visited()
    want_small, steps_small = orc.beam_weights(grid.cells, res, grid.origin, (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, max_range), states[:m], pts,
                                               threads=orc.max_threads(), return_steps=True)
    assert visited_small == steps_small
    np.testing.assert_allclose(got_small, want_small, rtol=1e-10, atol=1e-300)
    f.close()


@pytest.mark.parametrize("sigma", [(0.05, 0.05, 0.01), (0.6, 0.6, 0.25)])
def test_beam_kernel_options_visit_the_same_cells_and_leave_the_same_weights(sigma):
    """The ordered beam kernel with and without its two workgroup-level shortcuts - the scan taken in four sector windows
    (beam_sectors; engaged by scanners that reach 448 .. 896 cells: 30 m at 5 cm = 600) and the free-ahead certificate of the
    workgroup's middle ray (beam_free_ahead; fires on a tight cloud) -, all four combinations on a tight and on a wide cloud:
    the cells visited (the reference's count, an integer) are IDENTICAL, the weights agree within the rounding of a particle's sum
    over the scan (with sectors the terms are added sector by sector, and which sector a beam falls into is decided by the
    workgroup's middle particle: the order of the additions, not a term, depends on the neighbours - 1e-12), and every combination
    agrees with the oracle."""
    cells = synth.make_rooms_map(2000, 2000, seed=11, n_rooms=60)
    res, origin = 0.05, se2_from_xytheta(-50.0, -50.0, 0.0)
    grid = OccupancyGrid(cells=cells, resolution=res, origin=origin)
    truth = synth.find_free_pose(cells, res, (-50.0, -50.0), seed=2, clearance_cells=12)
    max_range = 30.0
    pts = make_scan(grid, truth, 360, max_range=max_range, fov=270.0)
    n = 40_000
    states = synth.normal_particles(n, truth, sigma, seed=6)
    beam = BeamModelParam(beam_max_range=max_range)
    results = {}
    for sectors in (1, 0):
        for ahead in (1, 0):
            f = new_filter(grid, n, sensor=beam)
            f.set_option("beam_sectors", sectors)
            f.set_option("beam_free_ahead", ahead)
            f.set_particles(states, np.ones(n))
            f.beam_cells_visited(reset=True)
            f.reweight(pts)
            results[(sectors, ahead)] = (f.particles()[1].copy(), f.beam_cells_visited())
            f.close()
    want, steps = orc.beam_weights(grid.cells, res, grid.origin, (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, max_range), states, pts,
                                   threads=orc.max_threads(), return_steps=True)
    base_w, base_cells = results[(0, 0)]
    assert base_cells == steps
    for key, (w, visited) in results.items():
        assert visited == steps, (key, visited, steps)
        np.testing.assert_allclose(w, base_w, rtol=1e-12, atol=1e-300, err_msg=str(key))
        np.testing.assert_allclose(w, want, rtol=1e-10, atol=1e-300, err_msg=str(key))


def test_propagate_matches_oracle():
    grid = rooms_grid(64, 1)
    n = 10_000
    states = synth.normal_particles(n, (0.0, 0.0, 0.3), (1.0, 1.0, 1.0), seed=4)
    pose, prev = se2_from_xytheta(1.3, 0.4, 0.35), se2_from_xytheta(1.0, 0.3, 0.30)
    f = new_filter(grid, n)
    f.set_particles(states, np.ones(n))
    f.propagate(pose, prev, step=7)
    got, _ = f.particles()
    sampler = orc.diffdrive_sampler(pose, prev, MOTION_T)
    want = orc.propagate(states, sampler, seed=11, step=7)
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-13)
    # in-place rotation branch (distance <= 0.01)
    f.set_particles(states, np.ones(n))
    pose2 = se2_from_xytheta(1.0, 0.3, 0.9)
    f.propagate(pose2, prev, step=8)
    want = orc.propagate(states, orc.diffdrive_sampler(pose2, prev, MOTION_T), seed=11, step=8)
    np.testing.assert_allclose(f.particles()[0], want, rtol=1e-11, atol=1e-13)
    f.close()


def test_normalize_and_policy_statistics():
    grid = rooms_grid(64, 1)
    n = 50_001
    rng = np.random.Generator(np.random.MT19937(12))
    w = rng.gamma(0.7, 1.0, n)
    f = new_filter(grid, n)
    f.set_particles(synth.normal_particles(n, (0, 0, 0), (1, 1, 1)), w)
    st = f.normalize()
    want_w, want_sum = orc.normalize(w)
    assert st["sum"] == pytest.approx(want_sum, rel=RTOL)
    np.testing.assert_allclose(f.particles()[1], want_w, rtol=RTOL)
    ess = st["norm_sum"] ** 2 / st["norm_sumsq"]
    assert ess == pytest.approx(orc.effective_sample_size(want_w), rel=1e-11)
    assert st["norm_sum"] == pytest.approx(1.0, abs=1e-12)
    # already normalised: untouched (normalize.hpp:73)
    before = f.particles()[1]
    f.normalize()
    assert np.array_equal(before, f.particles()[1])
    f.close()


def _compare_resampled(got, want_states, src_states, cdf_ref, seed, step):
    """States equal row by row, except draws whose uniform sits within 1e-9 of a CDF step."""
    assert got.shape == want_states.shape
    diff = np.where(np.any(got != want_states, axis=1))[0]
    for j in diff:
        r = orc.draw(seed, step, 2, int(j))
        u = float((int(r[0]) << 32 | int(r[1])) >> 11) * 2.0 ** -53
        k = np.searchsorted(cdf_ref, u, side="left")
        near = min(abs(cdf_ref[min(k, len(cdf_ref) - 1)] - u), abs(cdf_ref[max(k - 1, 0)] - u))
        assert near < 1e-9, f"candidate {j}: mismatch not explained by a CDF rounding boundary (gap {near})"
    return len(diff)


def test_multinomial_resample_matches_oracle():
    grid = rooms_grid(64, 1)
    n = 100_000
    states = synth.normal_particles(n, (0.2, -0.1, 0.3), (0.5, 0.5, 0.2), seed=4)
    w = np.random.Generator(np.random.MT19937(5)).gamma(0.5, 1.0, n)
    w[::7] = 0.0  # zero-weight particles are never drawn (test_sample.cpp:98-103)
    f = new_filter(grid, n)
    f.set_particles(states, w)
    m = f.resample(0.0, step=3)
    assert m == n
    got, gw = f.particles()
    assert np.all(gw == 1.0)  # particle_traits.hpp:105
    want, anc = orc.resample(states, w, n, n, 0.05, 3.0, (0.5, 0.5, math.radians(10)), 0.0, seed=11, step=3)
    cdf = np.cumsum(w / w.sum())
    flips = _compare_resampled(got, want, states, cdf, 11, 3)
    assert flips <= 5
    # zero-weight ancestors never appear
    zero_states = {tuple(r) for r in states[::7][:50]}
    assert not any(tuple(r) in zero_states for r in got[:2000])
    f.close()


def test_random_intersperse_matches_oracle():
    grid = rooms_grid(128, 2)
    n = 20_000
    states = synth.normal_particles(n, (0.0, 0.0, 0.0), (0.3, 0.3, 0.2), seed=4)
    f = new_filter(grid, n)
    f.set_particles(states, np.ones(n))
    f.resample(0.25, step=9)
    got, _ = f.particles()
    # free cells in the world frame exactly as multivariate_uniform_distribution.hpp:157-159 lists them
    idx = np.flatnonzero(grid.cells.ravel() == 0)
    W = grid.cells.shape[1]
    free_xy = np.stack([(idx % W + 0.5) * grid.resolution + grid.origin[2], (idx // W + 0.5) * grid.resolution + grid.origin[3]], 1)
    want, anc = orc.resample(states, np.ones(n), n, n, 0.05, 3.0, (0.5, 0.5, math.radians(10)), 0.25, seed=11, step=9, free_xy=free_xy)
    assert anc[0] != -1  # first element is never interspersed (random_intersperse.hpp:90-100)
    inj = anc == -1
    assert inj.mean() == pytest.approx(0.25, abs=0.01)
    np.testing.assert_allclose(got[inj], want[inj], rtol=1e-12, atol=1e-12)  # same cells, same headings
    assert int(np.any(got[~inj] != want[~inj], axis=1).sum()) <= 5  # CDF rounding boundaries only
    f.close()


@pytest.mark.parametrize("spread,min_p", [((0.05, 0.05, 0.02), 100), ((2.0, 2.0, 1.0), 500), ((0.6, 0.6, 0.3), 5000)])
def test_kld_resample_matches_oracle(spread, min_p):
    """take_while_kld (take_while_kld.hpp:72-88,134-136): the cut length is an exact integer result."""
    grid = rooms_grid(128, 2)
    n, max_p = 30_000, 200_000
    states = synth.normal_particles(n, (0.0, 0.0, 0.0), spread, seed=6)
    w = np.random.Generator(np.random.MT19937(5)).gamma(2.0, 1.0, n)
    params = AmclParams(min_particles=min_p, max_particles=max_p)
    f = Amcl(grid, MOTION, LF, params, seed=11)
    f.set_particles(states, w)
    m = f.resample(0.0, step=2)
    want, anc = orc.resample(states, w, min_p, max_p, 0.05, 3.0, (0.5, 0.5, math.radians(10)), 0.0, seed=11, step=2)
    assert m == len(want)
    got, gw = f.particles()
    assert np.all(gw == 1.0)
    assert int(np.any(got != want, axis=1).sum()) <= 3
    f.close()


def test_kld_target_table_on_device():
    """test_take_while_kld.cpp:133-148 replayed through the device kernels: k distinct bins cycled forever."""
    grid = rooms_grid(64, 1)
    for z, k, expected in [(1.28155156327703, 3, 228), (1.28155156327703, 100, 5871), (2.32634787735669, 7, 843),
                           (2.32634787735669, 100, 6733)]:
        # k particles in k distinct spatial bins with equal weight.
        states = np.stack([np.ones(k), np.zeros(k), np.arange(k) * 1.0 + 0.25, np.zeros(k) + 0.25], axis=1)
        params = AmclParams(min_particles=0, max_particles=20_000, kld_epsilon=0.01, kld_z=z)
        f = Amcl(grid, MOTION, LF, params, seed=5)
        f.set_particles(states, np.ones(k))
        m = f.resample(0.0, step=1)
        want, _ = orc.resample(states, np.ones(k), 0, 20_000, 0.01, z, (0.5, 0.5, math.radians(10)), 0.0, seed=5, step=1)
        assert m == len(want)
        assert m == expected  # all k bins are seen long before target_size(k) candidates have been drawn
        f.close()


def test_estimate_matches_oracle():
    grid = rooms_grid(64, 1)
    n = 65_537
    states = synth.normal_particles(n, (57.3, -41.2, 2.9), (0.5, 0.7, 0.2), seed=4)
    w = np.random.Generator(np.random.MT19937(5)).gamma(1.5, 1.0, n)
    f = new_filter(grid, n)
    f.set_particles(states, w)
    pose, cov = f.estimate()
    want_pose, want_cov = orc.estimate(states, w)
    np.testing.assert_allclose(pose, want_pose, rtol=0, atol=1e-9)
    np.testing.assert_allclose(cov, want_cov, rtol=1e-9, atol=1e-12)
    # degenerate orientation (test_estimation.cpp:184-197)
    two = np.array([se2_from_xytheta(0, 0, math.pi / 2), se2_from_xytheta(0, 0, -math.pi / 2)])
    f.set_particles(two, [1.0, 1.0])
    pose, cov = f.estimate()
    assert cov[2, 2] == math.inf and pose[0] == 1.0 and pose[1] == 0.0
    f.close()


def test_initialize_matches_oracle_and_rejects_bad_covariance():
    grid = rooms_grid(64, 1)
    f = new_filter(grid, 5000)
    cov = np.array([[0.25, 0.05, 0.0], [0.05, 0.25, 0.0], [0.0, 0.0, 0.0685]])
    f.initialize((1.0, 2.0, 0.5), cov)
    got, w = f.particles()
    want, _ = orc.init_normal(5000, (1.0, 2.0, 0.5), cov, seed=11)
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12)
    assert np.all(w == 1.0)
    with pytest.raises(RuntimeError):
        f.initialize((0, 0, 0), np.array([[1.0, 2.0, 0], [0.0, 1.0, 0], [0, 0, 1.0]]))  # not symmetric
    with pytest.raises(RuntimeError):
        f.initialize((0, 0, 0), np.diag([1.0, -1.0, 1.0]))  # negative eigenvalue
    f.close()


def _run_both(grid, params, sensor, sensor_oracle_kw, cycles, beams, max_range, seed=21, init_sigma=(0.5, 0.5, 0.2), kidnap_at=None,
              options=None, from_map=False):
    origin_xy = (grid.origin[2], grid.origin[3])
    truth = synth.find_free_pose(grid.cells, grid.resolution, origin_xy, seed=4, clearance_cells=8)
    gpu = Amcl(grid, MOTION, sensor, params, seed=seed)
    for name, value in (options or {}).items():
        gpu.set_option(name, value)
    cpu = orc.Amcl(update_min_d=params.update_min_d, update_min_a=params.update_min_a, resample_interval=params.resample_interval,
                   selective_resampling=params.selective_resampling, min_particles=params.min_particles,
                   max_particles=params.max_particles, alpha_slow=params.alpha_slow, alpha_fast=params.alpha_fast,
                   kld_epsilon=params.kld_epsilon, kld_z=params.kld_z,
                   hash_res=(params.spatial_resolution_x, params.spatial_resolution_y, params.spatial_resolution_theta),
                   alphas=MOTION_T, seed=seed, **sensor_oracle_kw)
    cpu.set_map(grid.cells, grid.resolution, grid.origin)
    cov = np.diag([s * s for s in init_sigma])
    if from_map:
        gpu.initialize_from_map()
        cpu.initialize_from_map()
    else:
        gpu.initialize(truth, cov)
        cpu.initialize(truth, cov)
    results = []
    pose = truth
    odom = (0.0, 0.0, 0.0)
    for c in range(cycles):
        step_fwd, step_turn = (0.3, 0.05) if c % 5 != 4 else (0.02, 0.01)  # every 5th step is below update_min_d/a
        pose = synth.odometry_step(pose, step_fwd, step_turn)
        odom = synth.odometry_step(odom, step_fwd, step_turn)
        if kidnap_at is not None and c == kidnap_at:  # the robot is carried elsewhere: the scans stop matching the cloud
            pose = synth.find_free_pose(grid.cells, grid.resolution, origin_xy, seed=77, clearance_cells=8)
        pts = make_scan(grid, pose, beams, max_range=max_range, seed=100 + c)
        ctrl = se2_from_xytheta(*odom)
        g = gpu.update(ctrl, pts)
        o = cpu.update(ctrl, pts)
        assert (g is None) == (o is None), f"cycle {c}: update/no-update decisions differ"
        if g is None:
            continue
        assert gpu.last_info["resampled"] == cpu.last_info["resampled"], f"cycle {c}"
        assert gpu.last_info["num_particles"] == len(cpu.particles()[1]), f"cycle {c}: particle counts differ"
        results.append((c, g, o, gpu.last_info, cpu.last_info))
    return gpu, cpu, results, pose


def test_update_cycle_end_to_end_fixed_size():
    """amcl_core.hpp:165-201 for 12 cycles, multinomial fixed N (config-2 shape, small)."""
    grid = rooms_grid(400, 3)
    params = AmclParams(min_particles=20_000, max_particles=20_000)
    gpu, cpu, results, truth = _run_both(grid, params, LF, dict(lf=LF_T, lf_model_unknown_space=True), 12, 360, 12.0)
    assert len(results) >= 9
    for c, (gp, gc), (op, oc), gi, oi in results:
        np.testing.assert_allclose(gp, op, atol=1e-9, err_msg=f"cycle {c}")
        np.testing.assert_allclose(gc, oc, rtol=1e-8, atol=1e-11, err_msg=f"cycle {c}")
        assert gi["weight_sum"] == pytest.approx(oi["weight_sum"], rel=1e-11)
    gs, gw = gpu.particles()
    os_, ow = cpu.particles()
    assert int(np.any(np.abs(gs - os_) > 1e-9, axis=1).sum()) <= 3
    # and the filter actually localises (beluga_system_tests tolerance: 0.9 m / 30 deg)
    est = results[-1][1][0]
    assert math.hypot(est[2] - truth[0], est[3] - truth[1]) < 0.9
    gpu.close()


def test_cycle_completion_word_changes_no_result():
    """Option cycle_spin = 1: a fixed-size cycle ends on a completion word its last kernel stores to mapped host memory instead
    of a stream synchronisation (off by default: measured without gain).  300 cycles - past the periodic stream synchronisation
    of that path - give the same estimates bit for bit as the default, and the particle sets are equal."""
    grid = rooms_grid(400, 3)
    outs = []
    for spin in (0, 1):
        f = new_filter(grid, 20_000)
        f.set_option("cycle_spin", spin)
        truth = synth.find_free_pose(np.asarray(grid.cells), grid.resolution, (grid.origin[2], grid.origin[3]), seed=4, clearance_cells=8)
        pts = make_scan(grid, truth, 180, max_range=12.0)
        f.initialize(truth, np.diag([0.04, 0.04, 0.01]))
        est = []
        pose = np.asarray(truth, dtype=np.float64)
        for c in range(300):
            pose = pose + np.array([0.03 * math.cos(pose[2]), 0.03 * math.sin(pose[2]), 0.0 if c % 2 else 0.02])
            f.force_update()  # (steps below the motion policy's thresholds)
            e = f.update(se2_from_xytheta(*pose), pts)
            assert e is not None
            est.append(np.concatenate([e[0], e[1].ravel()]))
        outs.append((np.asarray(est), f.particles()))
        f.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1][0], outs[1][1][0]) and np.array_equal(outs[0][1][1], outs[1][1][1])


@pytest.mark.parametrize("device_policy", ["1", "0"])
def test_update_cycle_recovery_injection_fixed_size(device_policy):
    """Fixed N, no selective resampling: the recovery estimator (thrun_recovery_probability_estimator.hpp:69-89) runs on
    the device and the cycle synchronises once (option device_policy = 0: on the host).  A kidnapped robot makes the fast
    average drop below the slow one: random states are injected; probabilities, decisions and particles follow the oracle."""
    grid = rooms_grid(400, 3)
    params = AmclParams(min_particles=20_000, max_particles=20_000, alpha_slow=0.001, alpha_fast=0.1)
    gpu, cpu, results, truth = _run_both(grid, params, LF, dict(lf=LF_T, lf_model_unknown_space=True), 12, 180, 12.0, kidnap_at=5,
                                         options={"device_policy": int(device_policy)})
    injected = 0
    for c, (gp, gc), (op, oc), gi, oi in results:
        assert gi["random_state_probability"] == pytest.approx(oi["random_state_probability"], abs=1e-12), f"cycle {c}"
        assert gi["weight_sum"] == pytest.approx(oi["weight_sum"], rel=1e-11)
        np.testing.assert_allclose(gp, op, atol=1e-9, err_msg=f"cycle {c}")
        injected += oi["random_state_probability"] > 0.0
    assert injected >= 1
    gs, _ = gpu.particles()
    os_, _ = cpu.particles()
    assert int(np.any(np.abs(gs - os_) > 1e-9, axis=1).sum()) <= 3
    gpu.close()


def test_update_cycle_end_to_end_kld_selective_and_recovery():
    """KLD-adaptive size + selective resampling (ESS) + every_n=2 + Thrun recovery injection, 15 cycles (config-1 shape)."""
    grid = turtlebot_grid()
    params = AmclParams(min_particles=500, max_particles=2000, resample_interval=2, selective_resampling=True, alpha_slow=0.001,
                        alpha_fast=0.1)
    gpu, cpu, results, truth = _run_both(grid, params, LF, dict(lf=LF_T, lf_model_unknown_space=True), 15, 180, 3.5, seed=0xBE1A6A,
                                         init_sigma=(0.5, 0.5, 0.26))
    assert len(results) >= 11
    for c, (gp, gc), (op, oc), gi, oi in results:
        np.testing.assert_allclose(gp, op, atol=1e-9, err_msg=f"cycle {c}")
        np.testing.assert_allclose(gc, oc, rtol=1e-8, atol=1e-11, err_msg=f"cycle {c}")
        assert gi["random_state_probability"] == pytest.approx(oi["random_state_probability"], abs=1e-12)
        if oi["ess"] >= 0:
            assert gi["ess"] == pytest.approx(oi["ess"], rel=1e-10)
    gpu.close()


def test_update_cycle_end_to_end_beam_model():
    grid = rooms_grid(300, 4)
    params = AmclParams(min_particles=3000, max_particles=3000)
    beam = BeamModelParam(beam_max_range=10.0)
    gpu, cpu, results, truth = _run_both(grid, params, beam, dict(sensor="beam", beam=(0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 10.0)), 6, 90,
                                         10.0)
    for c, (gp, gc), (op, oc), gi, oi in results:
        np.testing.assert_allclose(gp, op, atol=1e-8, err_msg=f"cycle {c}")
        assert gi["weight_sum"] == pytest.approx(oi["weight_sum"], rel=1e-9)
    gpu.close()


def test_nullopt_semantics():
    grid = rooms_grid(64, 1)
    f = new_filter(grid, 100)
    assert f.update(se2_from_xytheta(0, 0, 0), [(1.0, 0.0)]) is None  # empty particle set: amcl_core.hpp:166-168
    f.initialize((0, 0, 0), np.diag([0.01, 0.01, 0.01]))
    assert f.update(se2_from_xytheta(0, 0, 0), [(1.0, 0.0)]) is not None  # first call: forced
    assert f.update(se2_from_xytheta(0.01, 0, 0), [(1.0, 0.0)]) is None  # below update_min_d / update_min_a
    f.force_update()
    assert f.update(se2_from_xytheta(0.01, 0, 0), [(1.0, 0.0)]) is not None
    assert f.update(se2_from_xytheta(0.5, 0, 0), [(1.0, 0.0)]) is not None
    f.close()


# ---- BASELINE-size checks through size-independent properties ---------------------------------------
def test_full_size_properties_1m_x_1080():
    """Config 2 shape: 1M particles x 1080 beams on a 4000x4000 grid.  The oracle cannot sweep 1e9 lookups in a
    test, so: (1) a 2048-particle sample is checked against it; (2) linearity of `sum pz^3` over a split of the
    scan; (3) permutation equivariance; (4) the resampled set only contains ancestors with non-zero weight."""
    size = 4000
    cells = synth.make_rooms_map(size, size, seed=42)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-100.0, -100.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-100.0, -100.0), seed=1)
    pts = make_scan(grid, truth, 1080, max_range=30.0)
    n = 1_000_000
    lf = LikelihoodFieldModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
    f = Amcl(grid, MOTION, lf, AmclParams(min_particles=n, max_particles=n), seed=3)
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=9)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    w_full = f.particles()[1]
    sample = np.random.Generator(np.random.MT19937(1)).choice(n, 2048, replace=False)
    field = f.likelihood_field()
    want = orc.lf_weights(field, 0.05, grid.origin, 100.0, states[sample], pts, threads=orc.max_threads())
    np.testing.assert_allclose(w_full[sample], want, rtol=RTOL)
    # linearity over the scan
    f.set_particles(states, np.ones(n))
    f.reweight(pts[:500])
    wa = f.particles()[1]
    f.set_particles(states, np.ones(n))
    f.reweight(pts[500:])
    wb = f.particles()[1]
    np.testing.assert_allclose((wa - 1.0) + (wb - 1.0), w_full - 1.0, rtol=1e-11, atol=1e-13)
    # permutation equivariance
    perm = np.random.Generator(np.random.MT19937(2)).permutation(n)
    f.set_particles(states[perm], np.ones(n))
    f.reweight(pts)
    assert np.array_equal(f.particles()[1], w_full[perm]) or np.allclose(f.particles()[1], w_full[perm], rtol=RTOL)
    # resample: every survivor is one of the inputs, none with zero weight
    w = w_full.copy()
    w[::2] = 0.0
    f.set_particles(states, w)
    assert f.resample(0.0, step=1) == n
    got, gw = f.particles()
    assert np.all(gw == 1.0)
    keys = {r.tobytes() for r in states[1::2]}
    assert all(r.tobytes() in keys for r in got[:20000])
    pose, cov = f.estimate()
    assert np.all(np.isfinite(pose)) and np.all(np.isfinite(cov))
    f.close()


def test_full_size_kld_cut_10m_is_exact():
    """Config 3 shape: 10M max particles, min 100k, KLD (eps .05, z 3).  The cut length of take_while_kld is an integer
    result; the sequential oracle can replay all 10M candidates in a few seconds, so it is compared exactly — once with
    a tight cloud (stops right after min) and once with a cloud spread over thousands of bins (stops much later)."""
    grid = rooms_grid(400, 3)
    n = 200_000
    for spread, expect_min in (((0.3, 0.3, 0.1), True), ((9.0, 9.0, 3.0), False)):
        states = synth.normal_particles(n, (0.0, 0.0, 0.0), spread, seed=6)
        w = np.random.Generator(np.random.MT19937(5)).gamma(2.0, 1.0, n)
        params = AmclParams(min_particles=100_000, max_particles=10_000_000)
        f = Amcl(grid, MOTION, LF, params, seed=11)
        f.set_particles(states, w)
        m = f.resample(0.0, step=2)
        want, _ = orc.resample(states, w, 100_000, 10_000_000, 0.05, 3.0, (0.5, 0.5, math.radians(10)), 0.0, seed=11, step=2)
        assert m == len(want)
        assert (m == 100_000) == expect_min
        got, gw = f.particles()
        assert np.all(gw == 1.0)
        assert int(np.any(got != want, axis=1).sum()) <= 5
        f.close()


def test_64m_particles_on_one_device_sampled_against_oracle():
    """The whole 8-GPU job of config 4 (8M particles per GPU) on one device: 64M x 1080 beams (2.4 GB of states; indices stay
    below 2^32, particle-beam products do not).  One full update cycle, then a reweight sampled against the oracle (first and
    last particle included), the weight sum against a long-double host sum, and a resample that may only draw from the
    particles with positive weight."""
    size = 4000
    cells = synth.make_rooms_map(size, size, seed=42)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-100.0, -100.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-100.0, -100.0), seed=1)
    pts = make_scan(grid, truth, 1080, max_range=30.0)
    n = 64_000_000
    f = Amcl(grid, MOTION, LF, AmclParams(min_particles=n, max_particles=n), seed=3)
    f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
    est = f.update(se2_from_xytheta(0.3, 0.0, 0.02), pts)
    assert est is not None and np.all(np.isfinite(est[0])) and np.all(np.isfinite(est[1]))
    assert f.last_info["resampled"] and f.last_info["num_particles"] == n
    states, w0 = f.particles()
    assert len(w0) == n and np.all(w0 == 1.0)
    f.reweight(pts)
    w = f.particles()[1]
    sample = np.concatenate([np.random.Generator(np.random.MT19937(1)).choice(n, 2048, replace=False), [0, n - 1]])
    want = orc.lf_weights(f.likelihood_field(), 0.05, grid.origin, 100.0, states[sample], pts, threads=orc.max_threads())
    np.testing.assert_allclose(w[sample], want, rtol=RTOL)
    assert abs(f.weight_sum() / float(np.sum(w, dtype=np.longdouble)) - 1.0) < 1e-12
    # resample from a set whose particles left of the estimate weigh nothing: no survivor lies there
    x_cut = est[0][2]
    w[states[:, 2] < x_cut] = 0.0
    assert 0 < np.count_nonzero(w) < n
    f.set_particles(states, w)
    assert f.resample(0.0, step=2) == n
    got, gw = f.particles()
    assert np.all(gw == 1.0) and np.all(got[:, 2] >= x_cut)
    assert got[:, 2].max() <= states[:, 2].max()
    f.close()


def test_full_size_beam_model_1m_sample_against_oracle():
    """Config 5 shape: 1M particles x 1080 beams, BeamSensorModel on the 4000x4000 int8 grid.  A 256-particle sample of
    the full launch is checked against the oracle (each oracle particle walks ~3.5e5 cells), plus the cells-visited
    count of the whole launch against the oracle's per-particle average."""
    size = 4000
    cells = synth.make_rooms_map(size, size, seed=42)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-100.0, -100.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-100.0, -100.0), seed=1)
    pts = make_scan(grid, truth, 1080, max_range=30.0)
    n = 1_000_000
    beam = BeamModelParam(beam_max_range=30.0)
    f = Amcl(grid, MOTION, beam, AmclParams(min_particles=n, max_particles=n), seed=3)
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=9)
    f.set_particles(states, np.ones(n))
    f.beam_cells_visited(reset=True)
    f.reweight(pts)
    w = f.particles()[1]
    visited = f.beam_cells_visited()
    sample = np.random.Generator(np.random.MT19937(1)).choice(n, 256, replace=False)
    want, steps = orc.beam_weights(cells, 0.05, grid.origin, (0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 30.0), states[sample], pts,
                                   threads=orc.max_threads(), return_steps=True)
    np.testing.assert_allclose(w[sample], want, rtol=1e-10, atol=1e-300)
    assert visited / n == pytest.approx(steps / 256, rel=0.05)
    assert np.all(np.isfinite(w)) and np.all(w >= 0)
    f.close()


# ---- the other model closures of beluga_ros::Amcl's variant set -----------------------------
def test_omnidirectional_and_stationary_propagation_match_oracle():
    from beluga_amd.amcl import OmnidirectionalDriveModelParam, StationaryModelParam
    grid = rooms_grid(64, 1)
    n = 10_000
    states = synth.normal_particles(n, (0.0, 0.0, 0.3), (1.0, 1.0, 1.0), seed=4)
    pose, prev = se2_from_xytheta(1.3, 0.4, 0.35), se2_from_xytheta(1.0, 0.3, 0.30)
    alphas = (0.1, 0.05, 0.1, 0.05, 0.08)
    f = Amcl(grid, OmnidirectionalDriveModelParam(*alphas), LF, AmclParams(min_particles=n, max_particles=n), seed=11)
    f.set_particles(states, np.ones(n))
    f.propagate(pose, prev, step=7)
    want = orc.propagate_kind(states, "omnidirectional", pose, prev, alphas, seed=11, step=7)
    np.testing.assert_allclose(f.particles()[0], want, rtol=1e-11, atol=1e-13)
    f.close()
    f = Amcl(grid, StationaryModelParam(), LF, AmclParams(min_particles=n, max_particles=n), seed=11)
    f.set_particles(states, np.ones(n))
    f.propagate(pose, prev, step=3)
    want = orc.propagate_kind(states, "stationary", pose, prev, (0.0,) * 5, seed=11, step=3)
    np.testing.assert_allclose(f.particles()[0], want, rtol=1e-11, atol=1e-13)
    f.close()


@pytest.mark.parametrize("n", [3000, 20_001])  # lane kernel / ordered-lanes kernel with the log table
def test_likelihood_field_prob_model_matches_oracle(n):
    from beluga_amd.amcl import LikelihoodFieldProbModelParam
    grid = rooms_grid()
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, 120, max_range=12.0)
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=5)
    prob = LikelihoodFieldProbModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True)
    f = Amcl(grid, MOTION, prob, AmclParams(min_particles=n, max_particles=n), seed=11)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    got = f.particles()[1]
    want = orc.lf_prob_weights(f.likelihood_field(), grid.resolution, grid.origin, 100.0, states, pts)
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=0)  # exp() of a sum of ~120 logs
    # reference golden (test_likelihood_field_prob_model.cpp:35-76) through the C ABI
    center = np.array([0] * 12 + [100] + [0] * 12, dtype=np.int8).reshape(5, 5)
    g = Amcl(OccupancyGrid(center, 0.5), MOTION, LikelihoodFieldProbModelParam(2.0, 20.0, 0.5, 0.5, 0.2), AmclParams(max_particles=4), seed=1)
    g.set_particles(np.array([[1.0, 0.0, 0.0, 0.0]]), [1.0])
    g.reweight([(1.20, 1.20), (1.25, 1.25), (1.30, 1.30)])
    assert g.particles()[1][0] == pytest.approx(1.068, abs=0.01)
    f.close()
    g.close()


def test_update_cycle_end_to_end_omni_and_prob_model():
    from beluga_amd.amcl import LikelihoodFieldProbModelParam, OmnidirectionalDriveModelParam
    grid = rooms_grid(400, 3)
    origin_xy = (grid.origin[2], grid.origin[3])
    truth = synth.find_free_pose(grid.cells, grid.resolution, origin_xy, seed=4, clearance_cells=8)
    params = AmclParams(min_particles=500, max_particles=5000)
    alphas = (0.1, 0.05, 0.1, 0.05, 0.08)
    gpu = Amcl(grid, OmnidirectionalDriveModelParam(*alphas), LikelihoodFieldProbModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True), params,
               seed=5)
    cpu = orc.Amcl(min_particles=500, max_particles=5000, alphas=alphas[:4], alpha5=alphas[4], motion="omnidirectional",
                   sensor="likelihood_field_prob", lf=LF_T, lf_model_unknown_space=True, seed=5)
    cpu.set_map(grid.cells, grid.resolution, grid.origin)
    cov = np.diag([0.04, 0.04, 0.01])
    gpu.initialize(truth, cov)
    cpu.initialize(truth, cov)
    pose, odom = truth, (0.0, 0.0, 0.0)
    for c in range(6):
        pose = synth.odometry_step(pose, 0.3, 0.05)
        odom = synth.odometry_step(odom, 0.3, 0.05)
        pts = make_scan(grid, pose, 60, max_range=8.0, seed=50 + c)  # few beams: the product of 60 likelihoods stays in range
        g = gpu.update(se2_from_xytheta(*odom), pts)
        o = cpu.update(se2_from_xytheta(*odom), pts)
        assert gpu.last_info["num_particles"] == len(cpu.particles()[1]), f"cycle {c}"
        np.testing.assert_allclose(g[0], o[0], atol=1e-8, err_msg=f"cycle {c}")
    gpu.close()


# ---- cluster_based_estimate ---------------------------------------------------------------------
def _multicluster(xmin, xmax, ymin, ymax, step):  # test_cluster_based_estimation.cpp:67-94
    xw, yw = xmax - xmin, ymax - ymin
    xs = np.arange(step / 2.0, xw + 1e-12, step)
    ys = np.arange(step / 2.0, yw + 1e-12, step)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    k = (2 * X >= xw) * 1.0 + (2 * Y >= yw) * 2.0 + 1.0
    wt = np.abs(np.sin(2.0 * np.pi * X / xw)) * np.abs(np.sin(2.0 * np.pi * Y / yw)) * k
    wt = np.maximum(0.0, wt - k / 2.0)
    states = np.stack([np.ones(X.size), np.zeros(X.size), X.ravel() + xmin, Y.ravel() + ymin], axis=1)
    return states, wt.ravel()


def test_cluster_based_estimate_matches_oracle_and_reference_vectors():
    grid = rooms_grid(64, 1)
    # HeaviestClusterSelectionTest (:357-386): the estimate equals the plain estimate of the heaviest quadrant
    states, w = _multicluster(-2.0, 2.0, -2.0, 2.0, 0.025)
    f = new_filter(grid, len(w))
    f.set_particles(states, w)
    pose, cov = f.cluster_based_estimate()
    sel = (states[:, 2] >= 0.0) & (states[:, 3] >= 0.0)
    exp_pose, exp_cov = orc.estimate(states[sel], w[sel])
    np.testing.assert_allclose(pose, exp_pose, atol=1e-6)
    np.testing.assert_allclose(cov, exp_cov, atol=1e-3)
    want_pose, want_cov = orc.cluster_based_estimate(states, w)
    np.testing.assert_allclose(pose, want_pose, atol=1e-9)
    np.testing.assert_allclose(cov, want_cov, rtol=1e-8, atol=1e-11)
    f.close()
    # NightmareDistributionTest (:388-415): isolated particles -> overall estimate
    far = np.array([se2_from_xytheta(-10, -10, 0), se2_from_xytheta(-10, 10, 0), se2_from_xytheta(10, -10, 0), se2_from_xytheta(10, 10, 0)])
    f = new_filter(grid, 4)
    f.set_particles(far, np.full(4, 0.2))
    pose, cov = f.cluster_based_estimate()
    exp_pose, exp_cov = orc.estimate(far, np.full(4, 0.2))
    np.testing.assert_allclose(pose, exp_pose, atol=1e-9)
    np.testing.assert_allclose(cov, exp_cov, rtol=1e-9, atol=1e-12)
    f.close()


def test_cluster_based_estimate_bimodal_cloud_and_update_path():
    """Two pose hypotheses (60 % / 40 %) with random headings and weights: the heavier mode wins; and update() returns the
    cluster-based estimate when asked to behave like beluga_ros::Amcl."""
    grid = rooms_grid(400, 3)
    n = 200_000
    a = synth.normal_particles(int(n * 0.6), (2.0, 1.0, 0.5), (0.3, 0.3, 0.15), seed=1)
    b = synth.normal_particles(n - len(a), (-4.0, -3.0, -2.0), (0.3, 0.3, 0.15), seed=2)
    states = np.concatenate([a, b])
    perm = np.random.Generator(np.random.MT19937(3)).permutation(n)
    states = states[perm]
    w = np.random.Generator(np.random.MT19937(4)).gamma(2.0, 1.0, n)
    f = new_filter(grid, n)
    f.set_particles(states, w)
    pose, cov = f.cluster_based_estimate()
    want_pose, want_cov = orc.cluster_based_estimate(states, w)
    np.testing.assert_allclose(pose, want_pose, atol=1e-9)
    np.testing.assert_allclose(cov, want_cov, rtol=1e-7, atol=1e-10)
    assert abs(pose[2] - 2.0) < 0.3 and abs(pose[3] - 1.0) < 0.3  # a sub-cluster of the heavier mode
    plain, _ = f.estimate()
    assert abs(plain[2] - 2.0) > 1.0  # the plain mean sits between the modes
    # update() path
    f.set_estimate_kind(True)
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=4, clearance_cells=8)
    f.initialize(truth, np.diag([0.04, 0.04, 0.01]))
    pts = make_scan(grid, truth, 90, max_range=8.0)
    est = f.update(se2_from_xytheta(0, 0, 0), pts)
    s, ww = f.particles()
    want_pose, want_cov = orc.cluster_based_estimate(s, ww)
    np.testing.assert_allclose(est[0], want_pose, atol=1e-9)
    np.testing.assert_allclose(est[1], want_cov, rtol=1e-7, atol=1e-10)
    f.close()


def test_cluster_based_estimate_beyond_the_host_cell_list():
    """Global localisation on the headline map: 200 000 particles from initialize_from_map() spread over the 4000 x 4000 grid
    occupy far more hash cells than the mapped host list holds (16 384): the estimate then goes through the second compaction
    into device arrays and the stream-ordered copies (context.hip, do_cluster_estimate) - the path a round-2 ordering bug sat
    in.  It is what beluga_ros::Amcl returns while the robot is lost (beluga_ros/src/amcl.cpp:125,
    cluster_based_estimation.hpp:415-433).  Three sets: uniform weights straight after the initialisation, random weights on
    the same states, and the set an update() leaves (estimate_kind = cluster based, i.e. the ROS facade's update)."""
    import bench
    cells, truth, odoms, scans, _poses = bench.make_workload(2)
    grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
    n = 200_000
    f = new_filter(grid, n)
    f.initialize_from_map()
    states, w = f.particles()
    assert len(w) == n
    pose, cov = f.cluster_based_estimate()
    cells_uniform = f.counter("cluster_cells")
    assert cells_uniform > 16_384, cells_uniform
    want_pose, want_cov = orc.cluster_based_estimate(states, w)
    np.testing.assert_allclose(pose, want_pose, atol=1e-9)
    np.testing.assert_allclose(cov, want_cov, rtol=1e-8, atol=1e-10)
    # random weights: the cells' priorities differ, the flood fill takes another order
    w2 = np.random.Generator(np.random.MT19937(8)).gamma(2.0, 1.0, n)
    f.set_particles(states, w2)
    pose, cov = f.cluster_based_estimate()
    assert f.counter("cluster_cells") > 16_384
    want_pose, want_cov = orc.cluster_based_estimate(states, w2)
    np.testing.assert_allclose(pose, want_pose, atol=1e-9)
    np.testing.assert_allclose(cov, want_cov, rtol=1e-8, atol=1e-10)
    # the update() path with selective resampling (no resampling while the ESS stays above N / 2: the set stays spread out)
    f.close()
    params = AmclParams(min_particles=n, max_particles=n, selective_resampling=True)
    f = Amcl(grid, MOTION, LF, params, seed=11)
    f.set_estimate_kind(True)
    f.initialize_from_map()
    est = f.update(se2_from_xytheta(*odoms[0]), scans[0][::6])
    assert est is not None
    s3, w3 = f.particles()
    if not f.last_info["resampled"]:
        assert f.counter("cluster_cells") > 16_384
    want_pose, want_cov = orc.cluster_based_estimate(s3, w3)
    np.testing.assert_allclose(est[0], want_pose, atol=1e-9)
    np.testing.assert_allclose(est[1], want_cov, rtol=1e-8, atol=1e-10)
    f.close()


def _free_in_both(grid_a, grid_b, seed, clearance=8):
    """A pose on a free cell (with clearance) of both maps; both grids share the world frame region around it."""
    for k in range(200):
        pose = synth.find_free_pose(grid_a.cells, grid_a.resolution, (grid_a.origin[2], grid_a.origin[3]), seed=seed + k,
                                    clearance_cells=clearance)
        bx = int(math.floor((pose[0] - grid_b.origin[2]) / grid_b.resolution))
        by = int(math.floor((pose[1] - grid_b.origin[3]) / grid_b.resolution))
        H, W = grid_b.cells.shape
        if clearance <= bx < W - clearance and clearance <= by < H - clearance and \
                np.all(grid_b.cells[by - clearance:by + clearance + 1, bx - clearance:bx + clearance + 1] == 0):
            return pose
    raise AssertionError("no pose free in both maps")


@pytest.mark.parametrize("model", ["likelihood_field", "beam"])
@pytest.mark.parametrize("n", [3_000, 40_000])
def test_update_map_on_a_live_filter_follows_the_oracle(model, n):
    """Amcl::update_map (amcl_core.hpp:150 -> likelihood_field_model_base.hpp:113-116 / beam_model.hpp:154): five cycles, a
    DIFFERENT map (other obstacles, other size, other origin) on the live filter, five more cycles.  mcl_set_map rebuilds the
    field, the palette, the far tiles, the bit maps and the beam table; the particles stay.  Every estimate, before and after,
    against the oracle that takes the same call."""
    grid_a = rooms_grid(400, 3)
    cells_b = synth.make_rooms_map(440, 360, seed=9, n_rooms=10)
    grid_b = OccupancyGrid(cells=cells_b, resolution=0.05, origin=se2_from_xytheta(-11.5, -8.0, 0.0))
    truth = _free_in_both(grid_a, grid_b, seed=4)
    if model == "beam":
        sensor, okw, beams, max_range, tol = BeamModelParam(beam_max_range=10.0), dict(sensor="beam", beam=(0.5, 0.5, 0.05, 0.05, 0.2, 0.1, 10.0)), 60, 10.0, 1e-8
    else:
        sensor, okw, beams, max_range, tol = LF, dict(lf=LF_T, lf_model_unknown_space=True), 180, 12.0, 1e-9
    params = AmclParams(min_particles=n, max_particles=n)
    gpu = Amcl(grid_a, MOTION, sensor, params, seed=5)
    gpu.set_option("lf_small_particles", 16_384)
    cpu = orc.Amcl(min_particles=n, max_particles=n, alphas=MOTION_T, seed=5, **okw)
    cpu.set_map(grid_a.cells, grid_a.resolution, grid_a.origin)
    cov = np.diag([0.04, 0.04, 0.01])
    gpu.initialize(truth, cov)
    cpu.initialize(truth, cov)
    pose, odom = truth, (0.0, 0.0, 0.0)
    grid = grid_a
    compared = 0
    for c in range(10):
        if c == 5:
            gpu.update_map(grid_b)
            cpu.set_map(grid_b.cells, grid_b.resolution, grid_b.origin)
            grid = grid_b
            if model == "likelihood_field":
                want = orc.make_likelihood_field(grid_b.cells, grid_b.resolution, LF_T, True, False)
                assert np.array_equal(gpu.likelihood_field().view(np.uint32), want.view(np.uint32))
        step = 0.26 if c % 2 == 0 else -0.26  # back and forth (beyond update_min_d each time): stays inside the free region of both maps
        pose = synth.odometry_step(pose, step, 0.21)
        odom = synth.odometry_step(odom, step, 0.21)
        pts = make_scan(grid, pose, beams, max_range=max_range, seed=300 + c)
        ctrl = se2_from_xytheta(*odom)
        g = gpu.update(ctrl, pts)
        o = cpu.update(ctrl, pts)
        assert (g is None) == (o is None), f"cycle {c}"
        if g is None:
            continue
        gi, oi = gpu.last_info, cpu.last_info
        assert gi["resampled"] == oi["resampled"] and gi["num_particles"] == len(cpu.particles()[1])
        assert gi["weight_sum"] == pytest.approx(oi["weight_sum"], rel=1e-9), f"cycle {c}"
        np.testing.assert_allclose(g[0], o[0], atol=tol, err_msg=f"cycle {c}: pose")
        np.testing.assert_allclose(g[1], o[1], rtol=1e-7, atol=1e-10, err_msg=f"cycle {c}: covariance")
        compared += 1
    assert compared >= 8
    gs, _ = gpu.particles()
    os_, _ = cpu.particles()
    assert int(np.any(np.abs(gs - os_) > 1e-9, axis=1).sum()) <= 3
    gpu.close()


# ---- the two table forms of the default kernel ---------------------------------------------------------------------
@pytest.mark.parametrize("table", ["palette", "cube"])
@pytest.mark.parametrize("field_kind", ["distance_map", "arbitrary"])
def test_reweight_lf_table_forms_match_oracle(table, field_kind):
    """The hot kernel reads the field through a palette (2-byte index per cell + exact f64 values in LDS) when the field
    has few distinct values, and through an 8-byte table otherwise (an arbitrary field pushed through
    mcl_set_likelihood_field has millions).  Both must give the reference's weights, including for beams that end far
    outside the grid (clamped to the border tiles / the table's "unknown" slot) and right on its edges."""
    grid = rooms_grid()
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, 200, max_range=12.0)
    pts = np.concatenate([pts, [[500.0, -300.0], [-1e4, 2e4], [0.0, 0.0]]])  # far outside / the sensor itself
    n = 30_011
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=5)
    h, w = grid.cells.shape
    # particles on the grid's corners and edges: their beams straddle cell -1 / 0 and W-1 / W
    edge = np.array([[-10.0, -10.0], [-10.0 + w * 0.05, -10.0], [-10.0, -10.0 + h * 0.05], [-10.0 + w * 0.05, -10.0 + h * 0.05]])
    states[:4, 2:] = edge
    states[4:8, 2] += 100.0
    w0 = np.random.Generator(np.random.MT19937(3)).uniform(0.5, 1.5, n)
    f = new_filter(grid, n)
    f.set_option("lf_table", 1 if table == "cube" else 0)
    if field_kind == "arbitrary":
        rng = np.random.Generator(np.random.MT19937(9))
        f.set_likelihood_field(rng.uniform(0.01, 1.2, size=grid.cells.shape).astype(np.float32))
    f.set_particles(states, w0)
    f.reweight(pts)
    _, got = f.particles()
    field = f.likelihood_field()
    want = w0 * orc.lf_weights(field, grid.resolution, grid.origin, LF.max_laser_distance, states, pts)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)
    f.close()


def test_particle_cloud_sample_matches_oracle():
    """beluga_ros::assign_particle_cloud(particles, size, PoseArray&) (particle_cloud.hpp:131-149): a weighted sample of
    `size` states (views::sample | take_exactly) that leaves the set alone.  Same Philox stream on both sides."""
    grid = rooms_grid()
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    n, size, seed = 50_000, 7_001, 11
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=5)
    w = np.random.Generator(np.random.MT19937(3)).uniform(0.0, 1.0, n) ** 6
    f = new_filter(grid, n)
    f.set_particles(states, w)
    cloud = f.sample_particle_cloud(size, draw_id=3)
    want, anc = orc.resample(states, w, size, size, 0.05, 3.0, (0.5, 0.5, 0.1), 0.0, seed, 0x80000000 | 3)
    assert cloud.shape == (size, 4) and want.shape == (size, 4)
    assert int(np.any(cloud != want, axis=1).sum()) <= 2  # CDF-rounding boundary draws at most
    assert len(np.unique(anc)) > size // 4                # a genuine weighted sample, not one particle repeated
    s2, w2 = f.particles()
    assert np.array_equal(s2, states) and np.array_equal(w2, w)  # the set is untouched
    assert f.sample_particle_cloud(0).shape == (0, 4)
    assert not np.array_equal(f.sample_particle_cloud(size, draw_id=4), cloud)
    f.close()


def test_frozen_update_cycles_fixture():
    """The HIP path against tests/golden/update_cycles_config1.npz — the committed end-to-end vectors (20 cycles of BASELINE
    config 1: turtlebot3 grid, 500..2000 particles, KLD + recovery): decisions and particle counts exactly, estimates 1e-9."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_cycle_fixture", os.path.join(GOLDEN, "make_cycle_fixture.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = np.load(gen.OUT)
    cells, res, origin, truth, steps = gen.scenario()
    p = gen.PARAMS
    params = AmclParams(min_particles=p["min_particles"], max_particles=p["max_particles"], alpha_slow=p["alpha_slow"],
                        alpha_fast=p["alpha_fast"], kld_epsilon=p["kld_epsilon"], kld_z=p["kld_z"], spatial_resolution_x=p["hash_res"][0],
                        spatial_resolution_y=p["hash_res"][1], spatial_resolution_theta=p["hash_res"][2])
    f = Amcl(OccupancyGrid(cells=cells, resolution=res, origin=origin), MOTION, LF, params, seed=gen.SEED)
    f.initialize(truth, np.diag([0.25, 0.25, 0.0685]))
    k = 0
    for c, (ctrl, pts) in enumerate(steps):
        out = f.update(ctrl, pts)
        assert (out is not None) == bool(want["updated"][c]), f"cycle {c}"
        if out is None:
            continue
        np.testing.assert_allclose(out[0], want["poses"][k], atol=1e-9, err_msg=f"cycle {c}")
        np.testing.assert_allclose(out[1], want["covs"][k], rtol=1e-8, atol=1e-11, err_msg=f"cycle {c}")
        assert f.num_particles() == want["counts"][k], f"cycle {c}"
        assert f.last_info["weight_sum"] == pytest.approx(want["weight_sums"][k], rel=1e-11)
        assert f.last_info["random_state_probability"] == pytest.approx(want["random_state_probability"][k], abs=1e-12)
        k += 1
    states, _ = f.particles()
    assert states.shape == want["final_states"].shape
    assert int(np.any(np.abs(states - want["final_states"]) > 1e-9, axis=1).sum()) <= 2
    f.close()


def test_update_cycles_are_bitwise_reproducible():
    """Two filters, same seed, same inputs: identical particles, weights and estimates bit for bit, although the spatial
    ordering uses LDS atomics (their arrival order changes which lane serves which particle, never a result)."""
    grid = rooms_grid(400, 3)
    origin_xy = (grid.origin[2], grid.origin[3])
    truth = synth.find_free_pose(grid.cells, grid.resolution, origin_xy, seed=4, clearance_cells=8)
    outs = []
    for _ in range(2):
        f = new_filter(grid, 60_000, min_particles=5_000)  # KLD-adaptive: exercises the hash table as well
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        pose, odom, est = truth, (0.0, 0.0, 0.0), []
        for c in range(10):
            pose = synth.odometry_step(pose, 0.3, 0.05)
            odom = synth.odometry_step(odom, 0.3, 0.05)
            est.append(f.update(se2_from_xytheta(*odom), make_scan(grid, pose, 360, max_range=12.0, seed=100 + c)))
        outs.append((est, f.particles()))
        f.close()
    (e0, (s0, w0)), (e1, (s1, w1)) = outs
    assert np.array_equal(s0, s1) and np.array_equal(w0, w1)
    for a, b in zip(e0, e1):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("fast", [1, 0])
def test_reweight_lf_fma_variant_is_bit_identical(fast):
    """The default variant of the LF kernel evaluates beam end-points with FMAs and falls back to the separately rounded
    evaluation whenever the two could disagree on the cell (end-point within 2^-33 of a cell boundary) or a particle is too
    far away for the error bound.  A power-of-two resolution puts many end-points exactly on cell boundaries (the
    fallback's trigger), some particles sit 10^6 cells away, some beams end far outside the map.  Option lf_fast = 0 runs
    the separately rounded kernel on the same input; the counter says which one ran."""
    cells = synth.make_rooms_map(512, 512, seed=5, n_rooms=14)
    res = 0.0625
    grid = OccupancyGrid(cells=cells, resolution=res, origin=se2_from_xytheta(-16.0, -16.0, 0.0))
    truth = synth.find_free_pose(cells, res, (-16.0, -16.0), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, 200, max_range=12.0)
    on_boundary = np.array([[k * res, (k % 7) * res] for k in range(1, 40)])  # multiples of the resolution
    pts = np.concatenate([pts, on_boundary, [[300.0, -200.0]]])
    n = 40_003
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=5)
    states[:2000, 0], states[:2000, 1] = 1.0, 0.0                     # heading exactly 0
    states[:2000, 2:] = np.round(states[:2000, 2:] / res) * res       # positions on the cell lattice
    states[2000:2004, 2] += 1e5                                        # beyond the fast path's range
    w0 = np.random.Generator(np.random.MT19937(3)).uniform(0.5, 1.5, n)
    f = new_filter(grid, n)
    f.set_option("lf_fast", fast)
    f.set_particles(states, w0)
    f.reweight(pts)
    assert f.counter("lf_fast_launches") == (1 if fast else 0)
    _, got = f.particles()
    want = w0 * orc.lf_weights(f.likelihood_field(), res, grid.origin, LF.max_laser_distance, states, pts)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=0)
    f.close()


def test_reweight_lf_far_beams_kernel_falls_back_to_the_exact_end_points():
    """k_reweight_lf_far_beams (dispersed sets: lanes over the beams of a pose) evaluates end-points with FMAs like the other kernels and
    runs a group of poses again with the separately rounded arithmetic when one of their end-points lies within 2^-33 of a cell boundary
    or a pose is too far away for the error bound: a power-of-two resolution, poses on the cell lattice with heading 0 and scan points on
    multiples of the resolution (many end-points exactly on boundaries), poses 10^6 cells away, beams that end far outside the map; prior
    weights that are not 1.  Against the oracle, and against the lane-per-particle gather kernel up to the rounding of the sum."""
    cells = synth.make_rooms_map(1024, 1024, seed=5, n_rooms=10)
    res = 0.0625
    grid = OccupancyGrid(cells=cells, resolution=res, origin=se2_from_xytheta(-32.0, -32.0, 0.0))
    truth = synth.find_free_pose(cells, res, (-32.0, -32.0), seed=2, clearance_cells=6)
    pts = make_scan(grid, truth, 200, max_range=12.0)
    on_boundary = np.array([[k * res, (k % 7) * res] for k in range(1, 40)])  # multiples of the resolution
    pts = np.concatenate([pts, on_boundary, [[300.0, -200.0]]])
    n = 40_003
    rng = np.random.Generator(np.random.MT19937(11))
    th = rng.uniform(-np.pi, np.pi, n)
    states = np.stack([np.cos(th), np.sin(th), rng.uniform(-34.0, 34.0, n), rng.uniform(-34.0, 34.0, n)], axis=1)
    states[:2000, 0], states[:2000, 1] = 1.0, 0.0                     # heading exactly 0
    states[:2000, 2:] = np.round(states[:2000, 2:] / res) * res       # positions on the cell lattice
    states[2000:2004, 2] += 1e5                                        # beyond the fast path's range
    w0 = rng.uniform(0.5, 1.5, n)
    f = new_filter(grid, n)
    f.set_option("lf_patch", 0)
    f.set_option("lf_far_tiles", 2)
    weights = {}
    for mode in (2, 0):
        f.set_option("lf_dispersed", mode)
        before = f.counter("lf_far_beams_launches")
        f.set_particles(states, w0)
        f.reweight(pts)
        assert f.counter("lf_far_tiles") > 0 and f.counter("lf_far_beams_launches") == before + (1 if mode == 2 else 0)
        weights[mode] = f.particles()[1]
    want = w0 * orc.lf_weights(f.likelihood_field(), res, grid.origin, LF.max_laser_distance, states, pts, threads=orc.max_threads())
    np.testing.assert_allclose(weights[0], want, rtol=RTOL, atol=0)
    np.testing.assert_allclose(weights[2], want, rtol=RTOL, atol=0)
    np.testing.assert_allclose(weights[2], weights[0], rtol=1e-13, atol=0)
    f.close()
    # LikelihoodFieldProbModel (likelihood_field_prob_model.hpp:77-90: the weight is the exponential of the sum of logs): the kernel's other instance
    from beluga_amd.amcl import LikelihoodFieldProbModelParam
    g = new_filter(grid, n, sensor=LikelihoodFieldProbModelParam(2.0, 100.0, 0.5, 0.5, 0.2, True))
    g.set_option("lf_patch", 0)
    g.set_option("lf_far_tiles", 2)
    inside = states.copy()
    inside[2000:2004, 2] -= 1e5
    got = {}
    for mode in (2, 0):
        g.set_option("lf_dispersed", mode)
        before = g.counter("lf_far_beams_launches")
        g.set_particles(inside, np.ones(n))
        g.reweight(pts[:200])
        assert g.counter("lf_far_beams_launches") == before + (1 if mode == 2 and g.counter("lf_far_tiles") > 0 else 0)
        got[mode] = g.particles()[1]
    want = orc.lf_prob_weights(g.likelihood_field(), res, grid.origin, 100.0, inside, pts[:200])
    np.testing.assert_allclose(got[2], want, rtol=1e-11, atol=0)
    np.testing.assert_allclose(got[2], got[0], rtol=1e-12, atol=0)
    g.close()


@pytest.mark.parametrize("n", [16_384, 66_667, 200_000, 300_000])
def test_reweight_lf_patch_kernel_equals_the_gather_kernel_bit_for_bit(n):
    """The default LF kernel reads the index table through per-workgroup LDS patches wherever a bound on the workgroup's
    spread proves the look-ups inside one (k_reweight_lf_patch, forced by option lf_patch = 2); lf_patch = 0 gathers every look-up.
    Same cells, same sums: the weights are identical bit for bit - here on a small map, where a wide initial cloud puts
    patches across all four grid edges (clamped columns and rows), over the segmented launches of small sets (57, 360 and
    1080 beams: 1 to 16 segments, with and without a tail of beams) - and both agree with the oracle."""
    cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    grid = OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
    # 1300 and 1700 beams (200 000 particles: one segment): more groups than there are plan entries (the groups beyond them are gathered)
    for beams in (57, 360, 1080) + ((1300, 1700) if n == 200_000 else ()):
        pts = make_scan(grid, truth, beams, max_range=12.0)
        weights = []
        for patch in (2, 0):  # always / never
            f = new_filter(grid, n)
            f.set_option("lf_patch", patch)
            f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
            states, w0 = f.particles()
            f.reweight(pts)
            weights.append(f.particles()[1].copy())
            if patch == 2 and n <= 66_667:
                want = w0 * orc.lf_weights(f.likelihood_field(), 0.05, grid.origin, LF.max_laser_distance, states, pts)
                np.testing.assert_allclose(weights[0], want, rtol=RTOL, atol=0)
            f.close()
        assert np.array_equal(weights[0], weights[1]), (n, beams, int((weights[0] != weights[1]).sum()))


@pytest.mark.parametrize("n,grid_wgs", [(300_000, 7), (300_000, 64), (300_000, 669), (1_000_003, 0), (1_000_003, 96)])
def test_reweight_lf_patch_kernel_with_a_queue_of_blocks_equals_the_gather_kernel_bit_for_bit(n, grid_wgs):
    """k_reweight_lf_patch<true> (option lf_queue = 1, the default where a launch has more blocks than the device keeps workgroups
    resident): the workgroups of the launch take their blocks from a counter instead of one block each - which workgroup computes a
    block changes nothing in it.  Weights identical to the gather kernel's bit for bit, with the default number of workgroups (three
    per CU) and with a few that take dozens of blocks each (option lf_queue_grid; 669 = one block short of a workgroup per block),
    over wide and tight clouds and scans with a tail of beams; launch after launch (the counter wraps to zero by itself); the blocks
    taken in order or from both ends of the order inwards (option lf_ends_first)."""
    cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    grid = OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
    cases = [(1080, (0.5, 0.5, 0.2)), (1080, (0.1, 0.1, 0.03)), (57, (0.3, 0.3, 0.1)), (1543, (0.1, 0.1, 0.03)), (8, (0.1, 0.1, 0.03))]
    if n > 500_000:
        cases = cases[:2]
    for beams, sigma in cases:
        pts = make_scan(grid, truth, beams, max_range=12.0)
        weights = []
        for patch in (2, 0):  # always / never
            f = new_filter(grid, n)
            f.set_option("lf_patch", patch)
            f.set_option("lf_queue", 1)
            f.set_option("lf_queue_grid", grid_wgs)
            f.set_option("lf_ends_first", 0 if grid_wgs == 64 else 1)  # the blocks in order / from both ends of the order inwards (the default)
            f.initialize(truth, np.diag([s * s for s in sigma]))
            f.reweight(pts)
            f.reweight(pts)  # a second launch finds the counter where the first one left it: at zero
            weights.append(f.particles()[1].copy())
            if patch == 2:
                assert f.counter("lf_queue_launches") == 2, (beams, n)
                planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
                assert planned > 0 and through <= planned
                if sigma[0] <= 0.1 and beams >= 57:
                    assert through > 0.8 * planned, (beams, sigma, through, planned)
            f.close()
        assert np.array_equal(weights[0], weights[1]), (n, beams, sigma, int((weights[0] != weights[1]).sum()))


def test_queue_of_blocks_leaves_the_same_cycle_as_a_workgroup_per_block():
    """Whole cycles (the fused mcl_update: the LF kernel's per-block sums of the new weights feed the normalisation) with the queue of
    blocks and with a workgroup per block (option lf_queue = 0): estimates, weights and particle sets identical bit for bit."""
    import bench
    cycles = 8
    cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
    grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
    n = 1_000_000
    outs = []
    for queue in (1, 0):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF),
                 AmclParams(min_particles=n, max_particles=n), seed=42)
        f.set_option("lf_queue", queue)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        est = []
        for c in range(cycles):
            e = f.update(se2_from_xytheta(*odoms[c]), scans[c])
            est.append(np.concatenate([e[0], e[1].ravel(), [f.last_info["weight_sum"]]]))
        assert f.counter("lf_queue_launches") == (cycles if queue else 0)
        outs.append((np.asarray(est), f.particles()))
        f.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1][0], outs[1][1][0]) and np.array_equal(outs[0][1][1], outs[1][1][1])


@pytest.mark.parametrize("n,beams", [(1_000_000, 1080), (300_001, 360), (70_000, 180), (2_000, 180), (2_097_152, 90)])
def test_one_launch_scan_and_folded_estimate_sums_leave_the_same_cycle(n, beams):
    """Round 6 built two fusions of the fixed-size cycle's tail: k_normalize_cdf (option scan_fused: normalisation, totals of the normalised
    weights, recovery estimator and CDF in one pass, the chunk sums handed from workgroup to workgroup inside the launch) instead of
    k_normalize + k_cdf, and the estimate sums added up by the draw kernel's last workgroup (option draw_fold) instead of k_final_rows -
    measured no faster at 1M particles, so by default they serve sets of up to 64K particles, where a cycle is bound by the host's launches;
    here they are forced on at every size (value 2) - and k_normalize without its store, the division repeated by k_cdf (option norm_store).
    Same threads, same elements, same order of additions (actions/normalize.hpp:54-85, views/sample.hpp:128-159,
    effective_sample_size.hpp:46-59): estimates, weight sums, recovery probabilities and the resampled sets are identical bit for bit,
    launch after launch (the tickets wrap to zero, the epochs move on), on sets of one chunk, of a ragged last chunk, of the largest size
    the fused kernel takes (1024 chunks) - with the LF patch kernel's workgroup sums (1M) and with chunk sums of the weights (the others)."""
    import bench
    cycles = 6
    cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
    grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
    keep = np.linspace(0, bench.BEAMS - 1, beams).astype(int)
    outs = []
    for fused, fold, store in ((1, 1, 0), (0, 0, 0), (0, 0, 1), (1, 0, 0), (0, 1, 1)):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF),
                 AmclParams(min_particles=n, max_particles=n), seed=42)
        f.set_option("small_fused", 0)         # (the 2000-particle case: these are the large path's kernels)
        f.set_option("scan_fused", 2 * fused)  # (2: wherever the kernel takes the set; the default takes it for small sets only)
        f.set_option("draw_fold", 2 * fold)
        f.set_option("norm_store", store)      # (0: k_normalize does not store the normalised weights, k_cdf divides again)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        est = []
        for c in range(cycles):
            e = f.update(se2_from_xytheta(*odoms[c]), scans[c][keep])
            est.append(np.concatenate([e[0], e[1].ravel(), [f.last_info["weight_sum"], f.last_info["random_state_probability"]]]))
        outs.append((np.asarray(est), f.particles()))
        f.close()
    for other in outs[1:]:
        assert np.array_equal(outs[0][0], other[0]), np.abs(outs[0][0] - other[0]).max()
        assert np.array_equal(outs[0][1][0], other[1][0]) and np.array_equal(outs[0][1][1], other[1][1])


@pytest.mark.parametrize("lo,hi,selective,interval,beams,cluster", [
    (2000, 2000, False, 1, 180, False), (500, 2000, False, 1, 180, False), (500, 2000, True, 1, 360, False), (4096, 4096, False, 2, 90, False),
    (100, 4096, True, 1, 180, False), (1, 1, False, 1, 8, False), (700, 700, True, 1, 180, False), (500, 2000, False, 1, 180, True)])
def test_small_sets_one_launch_tail_matches_the_kernels_of_the_large_path(lo, hi, selective, interval, beams, cluster):
    """Sets of up to 4096 particles - the reference's own sizes (amcl_core.hpp:44-46) - run everything behind the reweight in ONE launch of
    one workgroup (k_small_tail, option small_fused): normalisation, the recovery estimator, every_n [&& on_effective_size_drop], the
    fixed-size or KLD-adaptive resampling with random_intersperse, the estimate's sums; one host synchronisation per cycle instead of up
    to three.  Against the same filter on the kernels of the large path (small_fused = 0), cycle by cycle: the same decisions and particle
    counts (integers: exact - the KLD cut among them), weight sums / effective sample sizes / recovery probabilities within 1e-12,
    estimates within 1e-9, and the same particles up to draws that sit on a CDF rounding boundary.  One cycle runs with the recovery
    filters put apart (a random state probability of about a half: injected states)."""
    cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
    angles = synth.lidar_angles(beams, 270.0)
    outs = []
    for fused in (1, 0):
        f = Amcl(grid, MOTION, LF, AmclParams(min_particles=lo, max_particles=hi, selective_resampling=selective, resample_interval=interval), seed=11)
        f.set_option("small_fused", fused)
        if cluster:  # what beluga_ros::Amcl returns (beluga_ros/src/amcl.cpp:125): cluster_based_estimate, behind the same tail
            f.set_estimate_kind(True)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        pose, odom, rows = truth, (0.0, 0.0, 0.0), []
        for c in range(9):
            pose = synth.odometry_step(pose, 0.3, 0.05)
            odom = synth.odometry_step(odom, 0.3, 0.05)
            pts = synth.scan_points(synth.cast_scan(cells, 0.05, (-10.0, -10.0), pose, angles, 12.0, 0.01, seed=c), angles)
            if c == 5:
                n_now = f.num_particles()
                f.debug_set_recovery_filters(2.0 / n_now, 1.0 / n_now)
            e = f.update(se2_from_xytheta(*odom), pts)
            i = f.last_info
            rows.append((e, dict(i), f.particles()))
        outs.append(rows)
        f.close()
    flips = 0
    for c, (a, b) in enumerate(zip(*outs)):
        ia, ib = a[1], b[1]
        assert ia["resampled"] == ib["resampled"] and ia["num_particles"] == ib["num_particles"], (c, ia, ib)
        np.testing.assert_allclose(ia["weight_sum"], ib["weight_sum"], rtol=1e-12)
        np.testing.assert_allclose(ia["random_state_probability"], ib["random_state_probability"], rtol=0, atol=1e-12)
        if ib["ess"] >= 0:
            np.testing.assert_allclose(ia["ess"], ib["ess"], rtol=1e-12)
        sa, wa = a[2]
        sb, wb = b[2]
        assert sa.shape == sb.shape
        differ = int(np.any(sa != sb, axis=1).sum())
        flips = max(flips, differ)
        # (a draw on a CDF rounding boundary picks the neighbour: a particle differs from there on, the estimate by its share)
        tol = 1e-9 + 2.0 * differ / max(len(wa), 1)
        np.testing.assert_allclose(a[0][0], b[0][0], rtol=0, atol=tol)
        np.testing.assert_allclose(a[0][1], b[0][1], rtol=1e-7, atol=1e-9 + 4.0 * differ / max(len(wa), 1))
        if differ == 0:
            np.testing.assert_allclose(wa, wb, rtol=1e-12)
    assert flips <= 3, flips
    if lo == 500 and hi == 2000:
        assert any(r[1]["random_state_probability"] > 0.3 for r in outs[0]), [r[1]["random_state_probability"] for r in outs[0]]


@pytest.mark.parametrize("n", [1_000_000, 300_001])
def test_propagation_normals_drawn_ahead_change_nothing(n):
    """The propagation's standard normals depend on (seed, step, particle index) alone - differential_drive_model.hpp:156-163 scales them by
    the control action afterwards -, so a fixed-size cycle draws the NEXT cycle's a cycle ahead - inside the draw kernel, whose vector units
    wait for the fabric (option noise_ahead = 1), or by a kernel of its own behind the cycle's last one while the host is away
    (k_noise_ahead, 2) -, and k_propagate reads them.  Same expressions, same bits: whole cycles with
    and without are identical, and the counter says the normals drawn ahead were used from the second cycle on - also across a cycle in
    which the robot does not move (no update: the normals wait for the step they belong to)."""
    import bench
    cycles = 7
    cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
    grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
    outs = []
    for ahead in (1, 0, 2):  # by the draw kernel (the default) / by k_propagate itself / by k_noise_ahead behind the cycle's last kernel
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF),
                 AmclParams(min_particles=n, max_particles=n), seed=42)
        f.set_option("noise_ahead", ahead)
        f.set_option("order_ahead", 0)  # (the order computed ahead needs the normals of mode 1: with it the modes would differ in their ORDER)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        est = []
        for c in range(cycles):
            e = f.update(se2_from_xytheta(*odoms[c]), scans[c])
            est.append(np.concatenate([e[0], e[1].ravel(), [f.last_info["weight_sum"]]]))
            if c == 3:
                assert f.update(se2_from_xytheta(*odoms[c]), scans[c]) is None  # the same odometry again: on_motion says no
        assert f.counter("noise_ahead_used") == (cycles - 1 if ahead else 0)
        outs.append((np.asarray(est), f.particles()))
        f.close()
    for other in outs[1:]:
        assert np.array_equal(outs[0][0], other[0])
        assert np.array_equal(outs[0][1][0], other[1][0]) and np.array_equal(outs[0][1][1], other[1][1])


def test_spatial_order_computed_a_cycle_ahead_changes_locality_only():
    """Option order_ahead: behind a fixed-size cycle's last kernel, while the host is away, the NEXT cycle's spatial order is computed from
    where the new particles will be after the next propagation - their normals for that step are drawn already, the control action is
    predicted (this cycle's) - and the next cycle goes from its propagation straight into the reweight, if the action it gets is close to
    the prediction; else it orders by the real poses as before.  Only locality depends on the order (and the rounding of the sums taken in
    it): against the same filter without it - the same estimates within 1e-9, weight sums within 1e-12, the same particles up to draws on a CDF
    rounding boundary; the order was used in every cycle of a steady trajectory and refused where the robot suddenly turns; and it is as
    good an order: the share of beam groups through an LDS patch does not drop."""
    import bench
    cycles, n = 14, 1_000_000
    cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
    grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
    # cycle 9: the odometry jumps sideways and turns - nothing like the action before it (the scans stay what they are: the weights do not care)
    controls = [np.asarray(o, dtype=np.float64) for o in odoms]
    for c in range(9, cycles):
        controls[c] = controls[c] + np.array([0.35, -0.4, 0.6])
    outs = []
    for ahead in (1, 0):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF),
                 AmclParams(min_particles=n, max_particles=n), seed=42)
        f.set_option("order_ahead", ahead)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        est = []
        for c in range(cycles):
            e = f.update(se2_from_xytheta(*controls[c]), scans[c])
            est.append(np.concatenate([e[0], e[1].ravel(), [f.last_info["weight_sum"]]]))
        used, missed = f.counter("order_ahead_used"), f.counter("order_ahead_missed")
        share = f.counter("lf_patch_groups_through") / max(f.counter("lf_patch_groups_planned"), 1)
        outs.append((np.asarray(est), f.particles(), used, missed, share))
        f.close()
    (ea, pa, used, missed, share_a), (eb, pb, used_b, missed_b, share_b) = outs
    assert used_b == 0 and missed_b == 0
    # one prediction per cycle from the first on; refused: the first real motion (predicted from the filter's motionless first update), the
    # jump and the cycle behind it (predicted from the jump) - and used everywhere else
    assert used + missed == cycles - 1 and 3 <= missed <= 4 and used >= cycles - 5, (used, missed)
    np.testing.assert_allclose(ea[:, -1], eb[:, -1], rtol=1e-12)
    differ = int(np.any(pa[0] != pb[0], axis=1).sum())
    assert differ <= 40, differ  # (draws on a CDF rounding boundary, a handful per cycle at most, carried along)
    np.testing.assert_allclose(ea[:, :4], eb[:, :4], rtol=0, atol=1e-9 + 3.0 * differ / n)
    # (the cycles behind the jump spread the set tenfold: there the frame predicted from the estimate before last is a size too small, and the
    # order a little coarser - 0.723 against 0.746 of the groups over this sequence; on the steady trajectory of the bench the shares are equal)
    assert share_a > 0.95 * share_b, (share_a, share_b)


def test_map_built_ahead_on_a_worker_thread_swaps_in_between_two_updates():
    """mcl_set_map_async (an extension beside Amcl::update_map, amcl_core.hpp:150): the likelihood field of the next map is built on a
    worker thread - the reference's wavefront (distance_map.hpp:55-98), the same bits - while the filter keeps updating on the map it
    has; the swap happens at the start of the first update after the build is done.  Against a twin filter that is given the same map
    with the synchronous update_map at the cycle where the first one swaps: identical estimates and particles in every cycle, identical
    likelihood fields afterwards; before the swap both run on the old map."""
    import time
    cells_a = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    cells_b = synth.make_rooms_map(640, 480, seed=8, n_rooms=16)  # another shape as well
    origin = se2_from_xytheta(-10.0, -10.0, 0.0)
    grid_a, grid_b = OccupancyGrid(cells_a, 0.05, origin=origin), OccupancyGrid(cells_b, 0.05, origin=origin)
    truth = synth.find_free_pose(cells_a, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
    n = 50_000
    filters = [new_filter(grid_a, n) for _ in range(2)]
    cov = np.diag([0.25, 0.25, 0.04])
    for f in filters:
        f.initialize(truth, cov)
    first, twin = filters
    angles = synth.lidar_angles(360, 270.0)
    pose, odom = truth, (0.0, 0.0, 0.0)
    swapped_at = None
    for c in range(8):
        pose = synth.odometry_step(pose, 0.3, 0.05)
        odom = synth.odometry_step(odom, 0.3, 0.05)
        pts = synth.scan_points(synth.cast_scan(cells_a, 0.05, (-10.0, -10.0), pose, angles, 12.0, 0.01, seed=c), angles)
        if c == 2:
            first.update_map_async(grid_b)
            assert first.map_pending() in (1, 2)
            assert first.likelihood_field().shape == cells_a.shape  # still the old map
        if c == 4:  # by now the build is to be done: wait for it without swapping (the swap is the next update's)
            t0 = time.time()
            while first.map_pending() == 1 and time.time() - t0 < 60:
                time.sleep(0.01)
            assert first.map_pending() == 2
            twin.update_map(grid_b)
            swapped_at = c
        a = first.update(se2_from_xytheta(*odom), pts)
        b = twin.update(se2_from_xytheta(*odom), pts)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), c
        if swapped_at is not None:
            assert first.map_pending() == 0
    assert swapped_at == 4
    assert first.likelihood_field().shape == cells_b.shape
    assert np.array_equal(first.likelihood_field(), twin.likelihood_field())
    sa, wa = first.particles()
    sb, wb = twin.particles()
    assert np.array_equal(sa, sb) and np.array_equal(wa, wb)
    # a synchronous update_map replaces a map that is still on its way; map_commit(wait) swaps at once
    first.update_map_async(grid_a)
    first.update_map(grid_b)
    assert first.map_pending() == 0 and first.likelihood_field().shape == cells_b.shape
    first.update_map_async(grid_a)
    first.map_commit(wait=True)
    assert first.map_pending() == 0 and first.likelihood_field().shape == cells_a.shape
    for f in filters:
        f.close()


@pytest.mark.parametrize("interval", [1, 2])
def test_unit_weights_skip_the_old_weights_load_and_change_nothing(interval):
    """A set fresh from a resampling or an initialisation holds weights of exactly 1.0 (particle_traits.hpp:105), and the host knows: the
    LF patch kernel then takes the sensor term as the new weight without loading the old one (option lf_unit_weights; 1.0 x = x).  Whole
    cycles with and without: identical bit for bit - resampling every cycle (every reweight sees unit weights) and every second cycle
    (resample_interval 2: the reweights in between multiply into normalised weights, which must still be loaded)."""
    import bench
    cycles, n = 7, 1_000_000
    cells, truth, odoms, scans, _poses = bench.make_workload(cycles)
    grid = OccupancyGrid(cells, bench.RESOLUTION, origin=se2_from_xytheta(bench.ORIGIN[0], bench.ORIGIN[1], 0.0))
    outs = []
    for unit in (1, 0):
        f = Amcl(grid, DifferentialDriveModelParam(*bench.ALPHAS), LikelihoodFieldModelParam(**bench.LF),
                 AmclParams(min_particles=n, max_particles=n, resample_interval=interval), seed=42)
        f.set_option("lf_unit_weights", unit)
        f.initialize(truth, np.diag([0.25, 0.25, 0.04]))
        est = []
        for c in range(cycles):
            e = f.update(se2_from_xytheta(*odoms[c]), scans[c])
            est.append(np.concatenate([e[0], e[1].ravel(), [f.last_info["weight_sum"], float(f.last_info["resampled"])]]))
        outs.append((np.asarray(est), f.particles()))
        f.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1][0], outs[1][1][0]) and np.array_equal(outs[0][1][1], outs[1][1][1])
    if interval == 2:
        assert set(outs[0][0][:, -1]) == {0.0, 1.0}


@pytest.mark.parametrize("options", [
    dict(lf_split=3, lf_margin=1, key_curve=1),  # the defaults
    dict(lf_split=1), dict(lf_split=2), dict(lf_split=0, lf_margin=0, key_curve=0, key_bits_xy=6),
    dict(lf_margin=0, key_bits_xy=4), dict(key_bits_xy=5),
])
def test_reweight_lf_patch_planner_options_change_no_weight(options):
    """What the LDS-patch kernel's planner and the ordering decide - whole patches, half patches side by side / stacked for groups
    that straddle a range discontinuity, the per-axis or isotropic bound, Hilbert or Morton keys and their bit split, changes where a look-up is read from, never its value: the
    weights equal the gather kernel's bit for bit, on a cloud wide enough to put patches across the grid's edges, and on a
    tight one where nearly every group goes through a patch (and the planner's statistics say so)."""
    cells = synth.make_rooms_map(400, 400, seed=3, n_rooms=12)
    grid = OccupancyGrid(cells=cells, resolution=0.05, origin=se2_from_xytheta(-10.0, -10.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-10.0, -10.0), seed=4, clearance_cells=8)
    pts = make_scan(grid, truth, 1080, max_range=12.0)
    n = 200_000
    for sigma in ((0.5, 0.5, 0.2), (0.1, 0.1, 0.03)):
        weights, shares = [], []
        for patch in (2, 0):
            f = new_filter(grid, n)
            f.set_option("lf_patch", patch)
            for k, v in options.items():
                f.set_option(k, v)
            f.initialize(truth, np.diag([s * s for s in sigma]))
            f.reweight(pts)
            weights.append(f.particles()[1].copy())
            if patch == 2:
                shares.append(f.counter("lf_patch_groups_through") / max(f.counter("lf_patch_groups_planned"), 1))
            f.close()
        assert np.array_equal(weights[0], weights[1]), (options, sigma, int((weights[0] != weights[1]).sum()))
        if sigma[0] < 0.2:
            assert shares[0] > 0.9, (options, shares)


def test_degenerate_set_of_identical_poses_is_ordered_in_bounded_time():
    """A million particles on ONE pose (mcl_set_particles with copies; a zero-spread cloud): all keys are equal, the ordering's
    first pass puts the whole set into one of its 1024 buckets, and one workgroup used to walk it twice on its own.  Such a
    bucket keeps the first pass's order (by particle index) and every workgroup copies a slice of it; the weights are the
    oracle's, and the reweight (ordering + kernel) stays within a few milliseconds."""
    import time
    grid = rooms_grid(400, 3)
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=4, clearance_cells=8)
    pts = make_scan(grid, truth, 360, max_range=12.0)
    n = 1_000_000
    states = np.tile(se2_from_xytheta(*truth), (n, 1))
    states[n // 2:, 2] += 1e-3  # (two poses a millimetre apart: still one bucket)
    w0 = np.ones(n)
    f = new_filter(grid, n)
    f.set_particles(states, w0)
    f.reweight(pts)  # (first launch: allocations)
    f.set_particles(states, w0)
    f.sync()
    t0 = time.perf_counter()
    f.reweight(pts)
    f.sync()
    elapsed = time.perf_counter() - t0
    perm, _keys = f.debug_order()
    assert np.array_equal(np.sort(np.asarray(perm)), np.arange(n))  # still a permutation
    got = f.particles()[1]
    pick = np.array([0, 1, n // 2 - 1, n // 2, n - 1])
    want = orc.lf_weights(f.likelihood_field(), grid.resolution, grid.origin, LF.max_laser_distance, states[pick], pts)
    np.testing.assert_allclose(got[pick], want, rtol=RTOL, atol=0)
    assert np.all(got[:n // 2] == got[0]) and np.all(got[n // 2:] == got[n - 1])
    assert elapsed < 0.02, elapsed  # (0.3 ms of ordering + the kernel on a healthy set; the single-workgroup walk took milliseconds per pass)
    f.close()


def test_spatial_order_is_a_sorted_permutation():
    """The ordering pass (two-pass radix sort of the 20-bit keys, the second pass stable): perm is a permutation and
    keys[perm] is non-decreasing — with the key frame from the host's estimate, from a bounding-box pass (set_particles
    leaves the host without an estimate), for ragged sizes, a point-mass cloud and a cloud spread over the whole map."""
    grid = rooms_grid()
    truth = synth.find_free_pose(grid.cells, grid.resolution, (grid.origin[2], grid.origin[3]), seed=2, clearance_cells=6)
    for n, spread, via_initialize in [(20_001, (0.5, 0.5, 0.2), False), (100_000, (0.5, 0.5, 0.2), True), (65_536, (0.0, 0.0, 0.0), False),
                                      (300_017, (8.0, 8.0, 3.0), False), (1_000_000, (0.4, 0.2, 0.15), True)]:
        f = new_filter(grid, n)
        if via_initialize:
            f.initialize(truth, np.diag([max(s * s, 1e-6) for s in spread]))
        else:
            f.set_particles(synth.normal_particles(n, truth, spread, seed=5), np.ones(n))
        perm, keys = f.debug_order()
        assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32)), (n, spread)
        assert keys.max() < (1 << 20)
        ordered = keys[perm]
        assert np.all(ordered[1:] >= ordered[:-1]), (n, spread)
        if spread[0] > 0:
            assert len(np.unique(keys)) > 1000  # the frame resolves the cloud
        f.close()


def test_initialize_from_map_matches_oracle_and_localises():
    """beluga_ros::Amcl::initialize_from_map() (beluga_ros/include/beluga_ros/amcl.hpp:209): max_particles draws over the free
    cells (random/multivariate_uniform_distribution.hpp:126-161).  Same stream on both sides: the sets are identical; every
    state sits on a free cell centre; and a global localisation run from it follows the oracle cycle by cycle."""
    grid = rooms_grid(400, 3)
    n = 50_000
    f = new_filter(grid, n)
    o = orc.Amcl(min_particles=n, max_particles=n, alphas=MOTION_T, seed=11, lf=LF_T, lf_model_unknown_space=True)
    o.set_map(grid.cells, grid.resolution, grid.origin)
    f.initialize_from_map()
    o.initialize_from_map()
    gs, gw = f.particles()
    os_, ow = o.particles()
    assert len(gw) == n and np.all(gw == 1.0)
    np.testing.assert_allclose(gs, os_, rtol=0, atol=1e-12)
    ij = np.floor((gs[:, 2:] - np.array([grid.origin[2], grid.origin[3]])) / grid.resolution).astype(int)
    assert np.all(grid.cells[ij[:, 1], ij[:, 0]] == 0)
    np.testing.assert_allclose(np.hypot(gs[:, 0], gs[:, 1]), 1.0, atol=1e-15)
    f.close()
    params = AmclParams(min_particles=2_000, max_particles=60_000, selective_resampling=False)
    gpu, cpu, results, truth = _run_both(grid, params, LF, dict(lf=LF_T, lf_model_unknown_space=True), 10, 180, 12.0, from_map=True)
    assert len(results) >= 6
    for c, g, o_, gi, ci in results:
        np.testing.assert_allclose(g[0], o_[0], atol=1e-9, err_msg=f"cycle {c}")
    gpu.close()


def test_likelihood_field_accessors_and_point_cloud_update():
    """beluga_ros::Amcl's likelihood_field_origin() / has_likelihood_field() (amcl.hpp:161-188: the beam model throws
    std::runtime_error) and update(pose, SparsePointCloud3f) (beluga_ros/src/amcl.cpp:67-81): the projected points equal the
    oracle's and the update equals the update with those points."""
    from beluga_amd.amcl import project_point_cloud
    grid = OccupancyGrid(cells=rooms_grid().cells, resolution=0.05, origin=se2_from_xytheta(-3.0, 2.0, 0.3))
    f = new_filter(grid, 3000)
    assert f.has_likelihood_field()
    np.testing.assert_allclose(f.likelihood_field_origin(), grid.origin, atol=1e-15)
    b = Amcl(grid, MOTION, BeamModelParam(beam_max_range=10.0), AmclParams(min_particles=100, max_particles=100), seed=1)
    assert not b.has_likelihood_field()
    with pytest.raises(RuntimeError, match="does not support likelihood field"):
        b.likelihood_field_origin()
    b.close()
    rng = np.random.Generator(np.random.MT19937(4))
    cloud = rng.uniform(-6, 6, size=(300, 3)).astype(np.float32)
    origin = (0.0, 0.0, math.sin(0.2), math.cos(0.2), 0.1, -0.05, 0.3)
    pts = project_point_cloud(cloud, origin)
    assert np.array_equal(pts, orc.project_point_cloud(cloud, origin))
    truth = (2.0, 7.0, 0.4)
    g = new_filter(grid, 3000)
    cov = np.diag([0.04, 0.04, 0.01])
    f.initialize(truth, cov)
    g.initialize(truth, cov)
    a = f.update_point_cloud(se2_from_xytheta(0.3, 0.0, 0.0), cloud, origin)
    b2 = g.update(se2_from_xytheta(0.3, 0.0, 0.0), pts)
    assert a is not None and b2 is not None
    assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])
    f.close()
    g.close()


def test_ten_million_particles_sampled_against_oracle():
    """Config 3 shape at its full size, no switches: the kernel the library picks for a 10M-particle reweight on the 4000^2
    map (the FMA variant: the counter says so) against the oracle on 1024 sampled particles, and take_while_kld's cut from a
    10M-particle SOURCE set against the sequential oracle (an integer result: exact)."""
    size = 4000
    cells = synth.make_rooms_map(size, size, seed=42)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-100.0, -100.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-100.0, -100.0), seed=1)
    pts = make_scan(grid, truth, 1080, max_range=30.0)
    n = 10_000_000
    f = Amcl(grid, MOTION, LF, AmclParams(min_particles=100_000, max_particles=n), seed=3)
    states = synth.normal_particles(n, truth, (0.5, 0.5, 0.2), seed=9)
    f.set_particles(states, np.ones(n))
    f.reweight(pts)
    assert f.counter("lf_fast_launches") == 1
    w = f.particles()[1]
    sample = np.random.Generator(np.random.MT19937(1)).choice(n, 1024, replace=False)
    want = orc.lf_weights(f.likelihood_field(), 0.05, grid.origin, 100.0, states[sample], pts, threads=orc.max_threads())
    np.testing.assert_allclose(w[sample], want, rtol=RTOL)
    stats = f.normalize()
    assert stats["norm_sum"] == pytest.approx(1.0, abs=1e-9)
    m = f.resample(0.0, step=2)
    wn = w / stats["sum"]
    want_states, _ = orc.resample(states, wn, 100_000, n, 0.05, 3.0, (0.5, 0.5, math.radians(10)), 0.0, seed=3, step=2)
    assert m == len(want_states)
    got, gw = f.particles()
    assert np.all(gw == 1.0)
    flips = int(np.any(got != want_states, axis=1).sum())
    assert flips <= 5, flips
    f.close()


def test_dispersed_cloud_1m_sampled_against_oracle():
    """The worst case for the spatially ordered lanes: 1M particles from initialize_from_map on the 4000^2 map (global
    localisation; neighbours in any order are metres and radians apart).  initialize_from_map says so, and the reweight goes
    straight to the ordered-lanes gather kernel - in its far-tile form: look-ups into tiles of free space far from any
    obstacle (a bitmap in LDS) take the field's common value without a memory access; the LDS-patch kernel, tried on the same
    set, finds next to no group of beams that fits a patch and reports it, which sends the launches after it to the gather
    kernel as well (option lf_patch = 1).  Same weights bit for bit with and without the bitmap (option lf_far_tiles) and from
    the patch kernel; a 1024-particle sample against the oracle.  Option lf_dispersed = 1 sends such sets to
    k_reweight_lf_beams instead (wave per particle, lane per beam, no ordering pass: measured slower, kept as a switch); its
    lane sums are added in a tree: same weights up to rounding."""
    size = 4000
    cells = synth.make_rooms_map(size, size, seed=42)
    grid = OccupancyGrid(cells, 0.05, origin=se2_from_xytheta(-100.0, -100.0, 0.0))
    truth = synth.find_free_pose(cells, 0.05, (-100.0, -100.0), seed=1)
    pts = make_scan(grid, truth, 1080, max_range=30.0)
    n = 1_000_000
    f = Amcl(grid, MOTION, LF, AmclParams(min_particles=n, max_particles=n), seed=3)
    f.set_option("lf_dispersed", 0)  # (the lane-per-particle form first; the default, 2, below)
    f.initialize_from_map()
    states, w0 = f.particles()
    assert np.all(w0 == 1.0)
    f.reweight(pts)
    assert f.counter("lf_beams_launches") == 0 and f.counter("lf_patch_launches") == 0 and f.counter("lf_fast_launches") == 1
    tiles = (size // 8 + 2) ** 2
    assert f.counter("lf_far_launches") == 1 and tiles // 4 < f.counter("lf_far_tiles") < tiles
    w = f.particles()[1]
    # without the bitmap: every look-up goes to the table
    f.set_option("lf_far_tiles", 0)
    f.set_particles(states, w0)
    f.set_option("lf_patch", 0)
    f.reweight(pts)
    assert f.counter("lf_far_launches") == 1 and f.counter("lf_fast_launches") == 2 and f.counter("lf_patch_launches") == 0
    assert np.array_equal(f.particles()[1], w)
    f.set_option("lf_far_tiles", 1)
    f.set_option("lf_patch", 1)
    sample = np.random.Generator(np.random.MT19937(1)).choice(n, 1024, replace=False)
    want = orc.lf_weights(f.likelihood_field(), 0.05, grid.origin, 100.0, states[sample], pts, threads=orc.max_threads())
    np.testing.assert_allclose(w[sample], want, rtol=RTOL)
    # a set the library knows nothing about goes to the LDS-patch kernel first, which reports what it found
    f.set_particles(states, w0)
    f.reweight(pts)
    planned, through = f.counter("lf_patch_groups_planned"), f.counter("lf_patch_groups_through")
    assert f.counter("lf_patch_launches") == 1 and planned >= (n // 448 // 16) * 135 and through * 4 < planned
    assert np.array_equal(f.particles()[1], w)
    # ... and the launch after that report gathers again, with the bitmap (the weights multiply)
    f.reweight(pts)
    assert f.counter("lf_patch_launches") == 1 and f.counter("lf_beams_launches") == 0 and f.counter("lf_far_launches") == 2
    assert np.array_equal(f.particles()[1], w * w)
    # option lf_dispersed = 1: the wave-per-particle kernel for sets reported as dispersed
    f.set_option("lf_dispersed", 1)
    f.initialize_from_map()
    f.reweight(pts)
    assert f.counter("lf_patch_launches") == 1 and f.counter("lf_beams_launches") == 1
    np.testing.assert_allclose(f.particles()[1], w, rtol=1e-13)
    # the default, lf_dispersed = 2: the lanes over the beams of a pose, the poses in the position-major order, the bitmap by the
    # tiles' linear index (k_reweight_lf_far_beams; 1080 beams: 16 full rounds and one of 56)
    f.set_option("lf_dispersed", 2)
    f.initialize_from_map()
    f.reweight(pts)
    assert f.counter("lf_far_beams_launches") == 1 and f.counter("lf_beams_launches") == 1 and f.counter("lf_patch_launches") == 1
    got = f.particles()[1]
    np.testing.assert_allclose(got, w, rtol=1e-13)
    np.testing.assert_allclose(got[sample], want, rtol=RTOL)
    f.close()
