"""Dense O(P*H*W) float64 PyTorch-autograd statement of the raster spec (DESIGN.md "Raster spec").

Independent of oracle/raster_oracle.c: it is written from the spec as tensor algebra and differentiated
by autograd, so it cross-checks the oracle's hand-derived backward.  Tiny sizes only.
"""
import numpy as np
import torch

TILE = 16


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.view(-1, 3, 3)


def dense_render(settings, means3D, colors, opacities, scales, rotations, radii, uv_delta=None):
    """settings: dict(H,W,x_min,y_min,scale,threshold,viewmatrix[4,4],bg[3],scale_modifier[,flags,low_pass]); flags are the
    GSVC_RASTER_* convention switches of include/gsvc_hip.h (the slab test itself is in `radii`, which the oracle provides).
    radii: int array from the oracle (defines visibility and the tile rectangle, both non-differentiable).
    All tensor inputs float64.  Returns image [3,H,W]."""
    H, W = settings["H"], settings["W"]
    sc = settings["scale"]
    M = torch.as_tensor(np.asarray(settings["viewmatrix"], dtype=np.float64))
    bg = torch.as_tensor(np.asarray(settings["bg"], dtype=np.float64))
    Wm, t = M[:3, :3], M[:3, 3]
    pv = means3D @ Wm.T + t
    R = quat_to_rot(rotations)
    S = scales * settings.get("scale_modifier", 1.0)
    L = R * S[:, None, :]
    cov3 = L @ L.transpose(1, 2)
    T2 = sc * Wm[:2, :]
    cov2 = T2 @ cov3 @ T2.T
    flags = int(settings.get("flags", 0))
    lp = 0.0 if flags & 32 else (settings.get("low_pass", 0.0) or 0.3)
    off = 0.0 if flags & 2 else 0.5
    a = cov2[:, 0, 0] + lp
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + lp
    det = a * c - b * b
    A, B, Cc = c / det, -b / det, a / det
    u = (pv[:, 0] - settings["x_min"]) * sc - off
    v = (pv[:, 1] - settings["y_min"]) * sc - off
    if uv_delta is not None:
        u = u + uv_delta[:, 0]
        v = v + uv_delta[:, 1]
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    tile_x = (xs // TILE).long()
    tile_y = (ys // TILE).long()
    P = means3D.shape[0]
    radii = np.asarray(radii)
    # order: depth ascending (float32 value as the oracle sees it), ties by index
    depth32 = pv[:, 2].detach().numpy().astype(np.float32)
    if flags & 4:
        depth32 = -depth32
    order = sorted([i for i in range(P) if radii[i] > 0], key=lambda i: (depth32[i], i))
    T = torch.ones(H, W, dtype=torch.float64)
    C = torch.zeros(3, H, W, dtype=torch.float64)
    done = torch.zeros(H, W, dtype=torch.bool)
    u32 = u.detach().numpy().astype(np.float32)
    v32 = v.detach().numpy().astype(np.float32)
    for i in order:
        rf = np.float32(radii[i])

        def clampi(tf, g):
            tf = min(max(float(tf), -1.0), g + 1.0)
            return min(g, max(0, int(tf)))
        x0 = clampi((u32[i] - rf) / np.float32(TILE), gx)
        x1 = clampi((u32[i] + rf + np.float32(TILE - 1)) / np.float32(TILE), gx)
        y0 = clampi((v32[i] - rf) / np.float32(TILE), gy)
        y1 = clampi((v32[i] + rf + np.float32(TILE - 1)) / np.float32(TILE), gy)
        member = (tile_x >= x0) & (tile_x < x1) & (tile_y >= y0) & (tile_y < y1)
        dx = u[i] - xs
        dy = v[i] - ys
        power = -0.5 * (A[i] * dx * dx + Cc[i] * dy * dy) - B[i] * dx * dy
        raw = opacities[i] * torch.exp(power)
        if flags & 16:
            alpha = torch.clamp(raw, max=0.99)                          # a real clamp: no gradient where it is active
        else:
            alpha = raw + (torch.clamp(raw, max=0.99) - raw).detach()  # clamp value, pass-through gradient
        valid = member & (power <= 0) & (alpha >= 1.0 / 255.0) & (~done)
        test_T = T * (1 - alpha)
        stop = valid & (test_T < 1e-4)
        done = done | stop
        valid = valid & (~stop)
        w = torch.where(valid, alpha * T, torch.zeros_like(T))
        C = C + colors[i][:, None, None] * w
        T = torch.where(valid, test_T, T)
    return C + T * bg[:, None, None]
