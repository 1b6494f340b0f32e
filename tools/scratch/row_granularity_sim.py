"""Estimate (CPU Monte-Carlo): how much shorter would the blend kernels' per-tile replay be if the four 16-lane rows of a wave walked
their own entry lists over 4x4-pixel blocks instead of the whole wave walking 8x8 quadrants?  Gaussians as in the bench scene
(log-uniform sigma 0.5..4 px per axis, random orientation, opacity U(0.02, 0.98), low-pass 0.3), alpha >= 1/255 footprint only."""
import numpy as np
rng = np.random.default_rng(0)
def one_tile(n_entries):
    cx = rng.uniform(-12, 28, 4 * n_entries); cy = rng.uniform(-12, 28, 4 * n_entries)
    s = np.exp(rng.uniform(np.log(0.5), np.log(4.0), (4 * n_entries, 2)))
    th = rng.uniform(0, np.pi, 4 * n_entries); o = rng.uniform(0.02, 0.98, 4 * n_entries)
    c, sn = np.cos(th), np.sin(th)
    a = c * c * s[:, 0] ** 2 + sn * sn * s[:, 1] ** 2 + 0.3
    b = c * sn * (s[:, 0] ** 2 - s[:, 1] ** 2)
    d = sn * sn * s[:, 0] ** 2 + c * c * s[:, 1] ** 2 + 0.3
    det = a * d - b * b
    A, B, C = d / det, -b / det, a / det
    py, px = np.mgrid[0:16, 0:16]
    quad_iters = 0
    row_work = np.zeros(4)
    hit_entries = 0
    for i in range(4 * n_entries):
        dx, dy = px - cx[i], py - cy[i]
        alpha = np.minimum(0.99, o[i] * np.exp(-0.5 * (A[i] * dx * dx + C[i] * dy * dy) - B[i] * dx * dy))
        m = alpha >= 1 / 255
        if not m.any():
            continue
        hit_entries += 1
        for qy in range(2):
            for qx in range(2):
                q = m[8 * qy:8 * qy + 8, 8 * qx:8 * qx + 8]
                if q.any():
                    quad_iters += 1
                    # 4x4 sub-blocks of the quadrant: sub-block r belongs to DPP row r
                    for r in range(4):
                        sb = q[4 * (r >> 1):4 * (r >> 1) + 4, 4 * (r & 1):4 * (r & 1) + 4]
                        row_work[r] += sb.any()
        if hit_entries >= n_entries:
            break
    return quad_iters, row_work, hit_entries
tot_q = tot_max = tot_sum = ent = 0
for _ in range(60):
    q, rw, e = one_tile(220)
    tot_q += q; tot_max += rw.max(); tot_sum += rw.sum(); ent += e
print(f"entries/tile {ent / 60:.0f}; quadrant replays/entry {tot_q / ent:.2f}; 4x4 block visits per quadrant replay {tot_sum / tot_q:.2f} of 4")
print(f"wave iterations: quadrant scheme {tot_q / 60:.0f}/tile, row scheme (max over rows) {tot_max / 60:.0f}/tile -> ratio {tot_q / tot_max:.2f}x")
