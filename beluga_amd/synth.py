"""Synthetic workloads for tests and bench.py: occupancy maps, laser scans, trajectories.

Pure numpy, no dependency on the oracle or on the reference tree (neither exists on the GPU box's
product path).  Everything is seeded and deterministic.
"""
from __future__ import annotations

import math

import numpy as np

FREE, UNKNOWN, OCCUPIED = 0, -1, 100


def make_rooms_map(width: int = 4000, height: int = 4000, seed: int = 42, n_rooms: int = 130, wall: int = 2) -> np.ndarray:
    """Config-2 map: border walls + seeded axis-aligned rectangular room outlines `wall` cells thick,
    each with a door gap, ~1-2 % occupied, rest free (no unknown space)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    g = np.zeros((height, width), dtype=np.int8)
    g[:wall, :] = OCCUPIED
    g[-wall:, :] = OCCUPIED
    g[:, :wall] = OCCUPIED
    g[:, -wall:] = OCCUPIED
    for _ in range(n_rooms):
        w = int(rng.integers(width // 40, width // 8))
        h = int(rng.integers(height // 40, height // 8))
        x0 = int(rng.integers(wall, max(wall + 1, width - w - wall)))
        y0 = int(rng.integers(wall, max(wall + 1, height - h - wall)))
        x1, y1 = min(x0 + w, width - 1), min(y0 + h, height - 1)
        g[y0:y0 + wall, x0:x1] = OCCUPIED
        g[y1 - wall:y1, x0:x1] = OCCUPIED
        g[y0:y1, x0:x0 + wall] = OCCUPIED
        g[y0:y1, x1 - wall:x1] = OCCUPIED
        # a door on a random side
        side = int(rng.integers(0, 4))
        door = max(4, min(w, h) // 5)
        if side == 0:
            a = int(rng.integers(x0 + wall, max(x0 + wall + 1, x1 - door)))
            g[y0:y0 + wall, a:a + door] = FREE
        elif side == 1:
            a = int(rng.integers(x0 + wall, max(x0 + wall + 1, x1 - door)))
            g[y1 - wall:y1, a:a + door] = FREE
        elif side == 2:
            a = int(rng.integers(y0 + wall, max(y0 + wall + 1, y1 - door)))
            g[a:a + door, x0:x0 + wall] = FREE
        else:
            a = int(rng.integers(y0 + wall, max(y0 + wall + 1, y1 - door)))
            g[a:a + door, x1 - wall:x1] = FREE
    return g


def find_free_pose(cells: np.ndarray, resolution: float, origin_xy, seed: int = 1, clearance_cells: int = 20):
    """A pose (x, y, theta) in the world frame whose neighbourhood is free (unrotated origin)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    H, W = cells.shape
    for _ in range(10000):
        xi = int(rng.integers(W // 4, 3 * W // 4))
        yi = int(rng.integers(H // 4, 3 * H // 4))
        y0, y1 = max(0, yi - clearance_cells), min(H, yi + clearance_cells)
        x0, x1 = max(0, xi - clearance_cells), min(W, xi + clearance_cells)
        if np.all(cells[y0:y1, x0:x1] == FREE):
            return (origin_xy[0] + (xi + 0.5) * resolution, origin_xy[1] + (yi + 0.5) * resolution, float(rng.uniform(-math.pi, math.pi)))
    raise RuntimeError("no free pose found")


def cast_scan(cells: np.ndarray, resolution: float, origin_xy, pose_xytheta, angles: np.ndarray, max_range: float,
              noise_sigma: float = 0.0, seed: int = 1) -> np.ndarray:
    """Ranges of a 2-D lidar at `pose` (world frame, unrotated grid origin) by marching each ray in
    half-cell steps until a non-free cell or the map edge; max_range if nothing is hit."""
    H, W = cells.shape
    x, y, th = pose_xytheta
    step = 0.5 * resolution
    n_steps = int(math.ceil(max_range / step))
    t = (np.arange(1, n_steps + 1) * step)[None, :]
    ca, sa = np.cos(th + angles)[:, None], np.sin(th + angles)[:, None]
    xi = np.floor((x + t * ca - origin_xy[0]) / resolution).astype(np.int64)
    yi = np.floor((y + t * sa - origin_xy[1]) / resolution).astype(np.int64)
    inside = (xi >= 0) & (yi >= 0) & (xi < W) & (yi < H)
    hit = np.ones_like(inside)
    hit[inside] = cells[yi[inside], xi[inside]] != FREE
    first = np.argmax(hit, axis=1)
    any_hit = hit.any(axis=1)
    ranges = np.where(any_hit, t[0, first], max_range)
    ranges = np.minimum(ranges, max_range)
    if noise_sigma > 0:
        rng = np.random.Generator(np.random.MT19937(seed))
        ranges = np.clip(ranges + rng.normal(0.0, noise_sigma, size=ranges.shape), 0.05, max_range)
    return ranges


def scan_points(ranges: np.ndarray, angles: np.ndarray) -> np.ndarray:
    """Polar -> cartesian hits in the robot base frame (sensor/data/laser_scan.hpp:64-90 without filtering)."""
    return np.stack([ranges * np.cos(angles), ranges * np.sin(angles)], axis=1)


def lidar_angles(num_beams: int = 1080, fov_deg: float = 270.0) -> np.ndarray:
    half = math.radians(fov_deg) / 2.0
    return np.linspace(-half, half, num_beams, endpoint=False) + (half / num_beams)


def normal_particles(n: int, mean_xytheta, sigmas, seed: int = 7) -> np.ndarray:
    """n SE2 states (cos, sin, x, y) ~ N(mean, diag(sigmas^2)) — host-drawn initial sets for stage tests."""
    rng = np.random.Generator(np.random.MT19937(seed))
    x = rng.normal(mean_xytheta[0], sigmas[0], n)
    y = rng.normal(mean_xytheta[1], sigmas[1], n)
    t = rng.normal(mean_xytheta[2], sigmas[2], n)
    return np.stack([np.cos(t), np.sin(t), x, y], axis=1)


def odometry_step(pose_xytheta, forward: float, turn: float):
    x, y, t = pose_xytheta
    return (x + forward * math.cos(t), y + forward * math.sin(t), t + turn)
