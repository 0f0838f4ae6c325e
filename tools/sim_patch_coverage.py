"""CPU model of the LDS-patch planner of k_reweight_lf_patch (no GPU): share of the beam groups that fit a 64 x 64-cell patch
for a Gaussian cloud of the bench's shape, under different ordering keys and bounds.  Decides nothing by itself - it ranks the
candidates that are then measured on the GPU (tools/gpu_r3_lf.sh).

    python tools/sim_patch_coverage.py [sigma_x sigma_y sigma_theta]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

RES = 0.05
WG = 448


def spread(v, gap):  # bit i -> position gap * i
    out = np.zeros_like(v)
    for i in range(12):
        out |= ((v >> i) & 1) << (gap * i)
    return out


def hilbert3(a0, a1, a2, b):
    x0, x1, x2 = a0.copy(), a1.copy(), a2.copy()
    q = 1 << (b - 1)
    while q > 1:
        p = q - 1
        x0 ^= np.where(x0 & q, p, 0)
        for x in (x1, x2):
            t = (x0 ^ x) & p
            inv = (x & q) != 0
            x0 ^= np.where(inv, p, t)
            x ^= np.where(inv, 0, t)
        q >>= 1
    x1 ^= x0
    x2 ^= x1
    t = np.zeros_like(x0)
    q = 1 << (b - 1)
    while q > 1:
        t ^= np.where(x2 & q, q - 1, 0)
        q >>= 1
    x0 ^= t
    x1 ^= t
    x2 ^= t
    return (spread(x0, 3) << 2) | (spread(x1, 3) << 1) | spread(x2, 3)


def keys(x, y, th, sig, bits_xy, curve):
    bits_t = 20 - 2 * bits_xy

    def bins(v, s, nb):
        u = v / (8.0 * s) + 0.5
        return np.clip((u * (1 << nb)).astype(np.int64), 0, (1 << nb) - 1)
    bx, by, bt = bins(x, sig[0], bits_xy), bins(y, sig[1], bits_xy), bins(th, sig[2], bits_t)
    slab = (bt >> bits_xy) << (3 * bits_xy)
    bt_in = bt & ((1 << bits_xy) - 1)
    if curve == "morton":
        return slab | spread(bx, 3) | (spread(by, 3) << 1) | (spread(bt_in, 3) << 2)
    return slab | hilbert3(bt_in, by, bx, bits_xy)


def coverage(x, y, th, order, pts, per_axis, patch=64):
    n = len(x) // WG * WG
    xs, ys, ts = (v[order][:n].reshape(-1, WG) for v in (x / RES, y / RES, th))
    ref_x, ref_y = 0.5 * (xs.min(1) + xs.max(1)), 0.5 * (ys.min(1) + ys.max(1))
    ref_t = np.arctan2(np.sin(ts).sum(1), np.cos(ts).sum(1))
    Dx = np.abs(xs - ref_x[:, None]).max(1) * 1.001 + 2
    Dy = np.abs(ys - ref_y[:, None]).max(1) * 1.001 + 2
    d = ts - ref_t[:, None]
    A, B = (1 - np.cos(d)).max(1), np.abs(np.sin(d)).max(1)
    iso = (2 * np.abs(np.sin(d / 2))).max(1)
    c, s = np.cos(ref_t)[:, None], np.sin(ref_t)[:, None]
    qx = (pts[None, :, 0] * c - pts[None, :, 1] * s) / RES  # [wg][beam]
    qy = (pts[None, :, 0] * s + pts[None, :, 1] * c) / RES
    g = qx.shape[1] // 8
    qxg, qyg = qx[:, :g * 8].reshape(len(qx), g, 8), qy[:, :g * 8].reshape(len(qy), g, 8)
    span_x = np.floor(qxg.max(2)) - np.floor(qxg.min(2))
    span_y = np.floor(qyg.max(2)) - np.floor(qyg.min(2))
    if per_axis:
        tx = A[:, None] * np.abs(qxg).max(2) + B[:, None] * np.abs(qyg).max(2)
        ty = B[:, None] * np.abs(qxg).max(2) + A[:, None] * np.abs(qyg).max(2)
    else:
        reach = np.sqrt(qxg ** 2 + qyg ** 2).max(2)
        tx = ty = iso[:, None] * reach
    mx, my = np.ceil(Dx[:, None] + tx), np.ceil(Dy[:, None] + ty)
    fits = (mx < 64) & (my < 64) & (span_x + 2 * mx < patch) & (span_y + 2 * my + 7 < patch)
    share = fits.mean(1)
    loose = share < 176 / 256
    return fits.mean(), np.where(loose, 0.0, share).mean(), loose.mean()


def main():
    sig = [float(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [0.33, 0.14, 0.11]
    n = 1_000_000
    rng = np.random.Generator(np.random.MT19937(5))
    x, y, th = (rng.normal(0.0, s, n) for s in sig)
    _, _, _, scans, _ = bench.make_workload(12)
    pts = np.asarray(scans[10])
    print(f"sigma {sig}, scan reach max {np.hypot(pts[:, 0], pts[:, 1]).max() / RES:.0f} cells, median {np.median(np.hypot(pts[:, 0], pts[:, 1])) / RES:.0f}")
    for bits_xy in (6, 5, 4):
        for curve in ("morton", "hilbert"):
            order = np.argsort(keys(x, y, th, sig, bits_xy, curve), kind="stable")
            for per_axis in (False, True):
                all_fit, through, loose = coverage(x, y, th, order, pts, per_axis)
                print(f"bits xy {bits_xy} theta {20 - 2 * bits_xy}  {curve:8s} {'per-axis ' if per_axis else 'isotropic'}: groups fitting {all_fit:.3f}, "
                      f"through a patch (loose workgroups gather everything) {through:.3f}, loose workgroups {loose:.3f}")


if __name__ == "__main__":
    main()


def extra():
    sig = [0.33, 0.14, 0.11]
    n = 1_000_000
    rng = np.random.Generator(np.random.MT19937(5))
    x, y, th = (rng.normal(0.0, s, n) for s in sig)
    _, _, _, scans, _ = bench.make_workload(12)
    pts = np.asarray(scans[10])
    order = np.argsort(keys(x, y, th, sig, 5, "hilbert"), kind="stable")
    global WG
    for wg in (448, 192, 960):
        WG = wg
        print("wg", wg, coverage(x, y, th, order, pts, True))
    WG = 448
    for patch in (64, 80, 96, 128):
        print("patch", patch, coverage(x, y, th, order, pts, True, patch))
    # no cloud at all: every particle at the reference pose (what the scan's own discontinuities cost)
    z = np.zeros(n)
    print("point cloud", coverage(z, z, z, np.arange(n), pts, True))
    # ideal order for this metric? sort by theta only / by theta then x
    for name, k in (("theta only", np.argsort(th)),):
        print(name, coverage(x, y, th, k, pts, True))


if __name__ == "__main__" and os.environ.get("SIM_EXTRA"):
    extra()
