"""Forward blend with per-row entry lists over 4x4 blocks, as it would be built: per chunk of 64 tile-list entries the quadrant
survivors (exact footprint) are listed per block by the alpha bounding box alone (or exactly), the wave iterates max over its
four blocks, rounded up to the unroll of 4.  Compared with today's count (all quadrant survivors)."""
import numpy as np
rng = np.random.default_rng(2)
def gaussians(n):
    out = []
    py, px = np.mgrid[0:16, 0:16]
    while len(out) < n:
        cx, cy = rng.uniform(-12, 28, 2)
        s = np.exp(rng.uniform(np.log(0.5), np.log(4.0), 2)); th = rng.uniform(0, np.pi); o = rng.uniform(0.02, 0.98)
        c, sn = np.cos(th), np.sin(th)
        a = c * c * s[0] ** 2 + sn * sn * s[1] ** 2 + 0.3; b = c * sn * (s[0] ** 2 - s[1] ** 2); d = sn * sn * s[0] ** 2 + c * c * s[1] ** 2 + 0.3
        det = a * d - b * b
        A, B, C = d / det, -b / det, a / det
        dx, dy = px - cx, py - cy
        m = np.minimum(0.99, o * np.exp(-0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy)) >= 1 / 255
        t2 = max(2 * np.log(255 * o), 0)
        rx, ry = np.sqrt(t2 * a), np.sqrt(t2 * d)
        bb = (int(np.ceil(cx - rx)), int(np.floor(cx + rx)), int(np.ceil(cy - ry)), int(np.floor(cy + ry)))
        # the tile list holds every Gaussian whose 3-sigma rectangle touches the tile (a superset of the alpha footprint)
        out.append((m, bb))
    return out
cur = new_bb = new_exact = 0
T = 40
for _ in range(T):
    gs = gaussians(300)       # tile list (incl. entries that reach no pixel)
    for qy in range(2):
        for qx in range(2):
            for c0 in range(0, len(gs), 64):
                chunk = gs[c0:c0 + 64]
                surv = [(m[8 * qy:8 * qy + 8, 8 * qx:8 * qx + 8], bb) for m, bb in chunk]
                surv = [s_ for s_ in surv if s_[0].any()]
                cur += (len(surv) + 3) // 4 * 4 if False else len(surv)
                cb = [0] * 4; ce = [0] * 4
                for q, (x0, x1, y0, y1) in surv:
                    for r in range(4):
                        bx, by = 8 * qx + 4 * (r & 1), 8 * qy + 4 * (r >> 1)
                        if x0 <= bx + 3 and x1 >= bx and y0 <= by + 3 and y1 >= by: cb[r] += 1
                        if q[4 * (r >> 1):4 * (r >> 1) + 4, 4 * (r & 1):4 * (r & 1) + 4].any(): ce[r] += 1
                new_bb += (max(cb) + 3) // 4 * 4
                new_exact += (max(ce) + 3) // 4 * 4
print(f"iterations per tile: today {cur / T:.0f}; per-block lists by bbox {new_bb / T:.0f} ({cur / new_bb:.2f}x), exact {new_exact / T:.0f} ({cur / new_exact:.2f}x)")
