"""Forward blend, wave per 8x8 quadrant: iterations if each 16-lane row walks only the entries that reach its own pixels.
Row shapes: 8x2 strip (today's lane -> pixel map) or 4x4 block.  Same Monte-Carlo scene as row_granularity_sim.py."""
import numpy as np
rng = np.random.default_rng(1)
def tile(n):
    m_all = []
    while len(m_all) < n:
        cx, cy = rng.uniform(-12, 28, 2)
        s = np.exp(rng.uniform(np.log(0.5), np.log(4.0), 2)); th = rng.uniform(0, np.pi); o = rng.uniform(0.02, 0.98)
        c, sn = np.cos(th), np.sin(th)
        a = c * c * s[0] ** 2 + sn * sn * s[1] ** 2 + 0.3; b = c * sn * (s[0] ** 2 - s[1] ** 2); d = sn * sn * s[0] ** 2 + c * c * s[1] ** 2 + 0.3
        det = a * d - b * b
        A, B, C = d / det, -b / det, a / det
        py, px = np.mgrid[0:16, 0:16]
        dx, dy = px - cx, py - cy
        m = np.minimum(0.99, o * np.exp(-0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy)) >= 1 / 255
        if m.any():
            m_all.append(m)
    return m_all
cur = strip = block = 0
for _ in range(40):
    ms = tile(220)
    for qy in range(2):
        for qx in range(2):
            qs = [m[8 * qy:8 * qy + 8, 8 * qx:8 * qx + 8] for m in ms]
            qs = [q for q in qs if q.any()]
            cur += len(qs)
            strip += max(sum(q[2 * r:2 * r + 2].any() for q in qs) for r in range(4))
            block += max(sum(q[4 * (r >> 1):4 * (r >> 1) + 4, 4 * (r & 1):4 * (r & 1) + 4].any() for q in qs) for r in range(4))
print(f"per tile: quadrant lists {cur / 40:.0f} iterations; rows = 8x2 strips {strip / 40:.0f} ({cur / strip:.2f}x); rows = 4x4 blocks {block / 40:.0f} ({cur / block:.2f}x)")
# strips selected by the alpha bounding box's y-interval alone (what the kernel has per instance), quadrant survivors as before
rng = np.random.default_rng(1)
def tile_bb(n):
    out = []
    while len(out) < n:
        cx, cy = rng.uniform(-12, 28, 2)
        s = np.exp(rng.uniform(np.log(0.5), np.log(4.0), 2)); th = rng.uniform(0, np.pi); o = rng.uniform(0.02, 0.98)
        c, sn = np.cos(th), np.sin(th)
        a = c * c * s[0] ** 2 + sn * sn * s[1] ** 2 + 0.3; b = c * sn * (s[0] ** 2 - s[1] ** 2); d = sn * sn * s[0] ** 2 + c * c * s[1] ** 2 + 0.3
        det = a * d - b * b
        A, B, C = d / det, -b / det, a / det
        py, px = np.mgrid[0:16, 0:16]
        dx, dy = px - cx, py - cy
        m = np.minimum(0.99, o * np.exp(-0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy)) >= 1 / 255
        if m.any():
            t2 = 2 * np.log(255 * o)
            ry = np.sqrt(max(t2, 0) * d)
            out.append((m, int(np.ceil(cy - ry)), int(np.floor(cy + ry))))
    return out
cur = strip = 0
for _ in range(40):
    ms = tile_bb(220)
    for qy in range(2):
        for qx in range(2):
            qs = [(m[8 * qy:8 * qy + 8, 8 * qx:8 * qx + 8], lo, hi) for m, lo, hi in ms]
            qs = [q for q in qs if q[0].any()]
            cur += len(qs)
            strip += max(sum((lo <= 8 * qy + 2 * r + 1) and (hi >= 8 * qy + 2 * r) for _, lo, hi in qs) for r in range(4))
print(f"rows = 8x2 strips by bbox y-interval: {strip / 40:.0f} iterations ({cur / strip:.2f}x)")
