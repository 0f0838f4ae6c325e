"""CPU model of k_reweight_lf_patch's planner on a REAL cloud (tools/dump_cloud.py: gpurun_out/cloud*.npy): which groups of 8 beams
fit a whole patch, two half patches, or nothing - by workgroup (448 consecutive particles of the device's order) and by beam
group -, and what the groups that fit nothing have in common.  Usage: python tools/sim_planner.py [prefix]"""
import os, sys
import numpy as np

prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cloud"
RES, WG, PW, PH = 0.05, 448, 64, 64
cloud = np.load(prefix + ".npy").astype(np.float64)
perm = np.load(prefix + "_perm.npy").astype(np.int64)
scan = np.load(prefix + "_scan.npy").astype(np.float64)
n = len(cloud) // WG * WG
x, y, th = (cloud[perm[:n], k].reshape(-1, WG) for k in range(3))
x, y = x / RES, y / RES
G = len(scan) // 8
q = scan[:G * 8]

ref_x, ref_y = 0.5 * (x.min(1) + x.max(1)), 0.5 * (y.min(1) + y.max(1))
sc, ss = np.cos(th).sum(1), np.sin(th).sum(1)
ln = np.hypot(sc, ss)
rc, rs = sc / ln, ss / ln
Dx = np.abs(x - ref_x[:, None]).max(1) * 1.001 + 2
Dy = np.abs(y - ref_y[:, None]).max(1) * 1.001 + 2
c, s = np.cos(th), np.sin(th)
A = np.abs(c * rc[:, None] + s * rs[:, None] - 1).max(1) * 1.001
B = np.abs(s * rc[:, None] - c * rs[:, None]).max(1) * 1.001
# reference end-points [wg][beam]
ex = (q[None, :, 0] * rc[:, None] - q[None, :, 1] * rs[:, None]) / RES
ey = (q[None, :, 0] * rs[:, None] + q[None, :, 1] * rc[:, None]) / RES
cx, cy = np.floor(ex + ref_x[:, None]), np.floor(ey + ref_y[:, None])
W = len(x)
cx, cy, ax, ay = (v.reshape(W, G, 8) for v in (cx, cy, np.abs(ex), np.abs(ey)))


def fits(lo_x, hi_x, lo_y, hi_y, qx, qy, pw, ph):
    tx = (A[:, None] * qx + B[:, None] * qy) * 1.001 * 1.001
    ty = (B[:, None] * qx + A[:, None] * qy) * 1.001 * 1.001
    mx, my = np.ceil(Dx[:, None] + tx), np.ceil(Dy[:, None] + ty)
    ok = (mx < 64) & (my < 64)
    x0 = lo_x - mx
    y0 = np.floor((lo_y - my) / 8) * 8
    return ok & (hi_x + mx - x0 < pw) & (hi_y + my - y0 < ph), mx, my


whole, mx, my = fits(cx.min(2), cx.max(2), cy.min(2), cy.max(2), ax.max(2), ay.max(2), PW, PH)
# split at the widest jump
jump = np.maximum(np.abs(np.diff(cx, axis=2)), np.abs(np.diff(cy, axis=2)))
k = jump.argmax(2) + 1
idx = np.arange(8)[None, None, :]
first = idx < k[:, :, None]


def half(mask, pw, ph):
    big = 1e9
    lo_x = np.where(mask, cx, big).min(2); hi_x = np.where(mask, cx, -big).max(2)
    lo_y = np.where(mask, cy, big).min(2); hi_y = np.where(mask, cy, -big).max(2)
    qx = np.where(mask, ax, 0).max(2); qy = np.where(mask, ay, 0).max(2)
    return fits(lo_x, hi_x, lo_y, hi_y, qx, qy, pw, ph)[0]


side = half(first, PW // 2, PH) & half(~first, PW // 2, PH)
stack = half(first, PW, PH // 2) & half(~first, PW, PH // 2)
ok = whole | side | stack
share = ok.mean(1)
loose = share < 176 / 256
print(f"workgroups {W}, groups {G}: whole {whole.mean():.4f}, + halves {ok.mean():.4f}; loose workgroups {loose.mean():.4f} "
      f"(their groups count as gathered: through a patch {np.where(loose[:, None], False, ok).mean():.4f})")
reach = np.hypot(ax, ay).max(2)
bad = ~ok & ~loose[:, None]
print("gathered groups inside patched workgroups:", bad.mean())
for lo, hi in ((0, 100), (100, 200), (200, 300), (300, 400), (400, 500), (500, 700)):
    sel = (reach >= lo) & (reach < hi)
    print(f"  reach {lo:3d}-{hi:3d} cells: {sel.mean():.3f} of the groups, {bad[sel].mean() if sel.any() else 0:.3f} of them gathered")
print("margin x of gathered groups: median", np.median(mx[bad]), " y:", np.median(my[bad]), "; of fitting groups:", np.median(mx[ok]), np.median(my[ok]))
span_x = cx.max(2) - cx.min(2); span_y = cy.max(2) - cy.min(2)
print("span of gathered groups: median x", np.median(span_x[bad]), "y", np.median(span_y[bad]), "; widest jump median", np.median(jump.max(2)[bad]))
wg_bad = bad.mean(1)
order = np.argsort(-wg_bad)
print("workgroups by share of gathered groups: top 5 %% hold %.2f of all gathered groups" % (bad[order[:W // 20]].sum() / max(bad.sum(), 1)))
print("A (1-cos) and B (sin) of workgroups, median / 95th pct:", np.median(A), np.percentile(A, 95), np.median(B), np.percentile(B, 95),
      " Dx, Dy median / 95th:", np.median(Dx), np.percentile(Dx, 95), np.median(Dy), np.percentile(Dy, 95))
print("B of the workgroups holding gathered groups (weighted):", np.average(B, weights=wg_bad + 1e-12), " Dx:", np.average(Dx, weights=wg_bad + 1e-12), " Dy:", np.average(Dy, weights=wg_bad + 1e-12))
