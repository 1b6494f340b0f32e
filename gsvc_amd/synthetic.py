"""Synthetic UVG-shaped inputs for tests and bench.py (no datasets in this image).

Cube convention follows reference frame_cube/frame.py:98-101,156-162:
``scale = max(H, W, T)/2``, ``x_min = -W/2/scale``, ``y_min = -H/2/scale``, ``z_k = (k - T/2)/scale``.
Everything is generated with numpy ``default_rng(seed)`` so the same bytes appear on every box.
"""
from __future__ import annotations

import numpy as np


def cube(H: int, W: int, T: int):
    scale = max(H, W, T) / 2.0
    return dict(scale=scale, x_min=-W / 2.0 / scale, y_min=-H / 2.0 / scale, z_min=-T / 2.0 / scale)


def frame_z(frame_id: int, T: int, scale: float) -> float:
    return (frame_id - T / 2.0) / scale


def view_matrices(z_cam: float):
    """Row-major math matrices of the two glm.lookAt views of reference frame_cube/frame.py:18-43."""
    Mf = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -z_cam], [0, 0, 0, 1]], dtype=np.float32)
    Ms = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, z_cam], [0, 0, 0, 1]], dtype=np.float32)
    return Mf, Ms


def raster_scene(P: int, H: int = 1080, W: int = 1920, T: int = 600, seed: int = 2026, window_frames: float = 16,
                 frame_id: int | None = None, sigma_px=(0.5, 4.0), opacity=(0.02, 0.98)):
    """BASELINE.md config 2/5 raster-only set: P Gaussians uniform in the frame's z-slab."""
    rng = np.random.default_rng(seed)
    cb = cube(H, W, T)
    scale = cb["scale"]
    if frame_id is None:
        frame_id = T // 2
    z_cam = frame_z(frame_id, T, scale)
    thr = (window_frames / 2.0) / scale
    means = np.empty((P, 3), dtype=np.float32)
    means[:, 0] = rng.uniform(cb["x_min"], -cb["x_min"], P)
    means[:, 1] = rng.uniform(cb["y_min"], -cb["y_min"], P)
    means[:, 2] = rng.uniform(z_cam - thr, z_cam + thr, P)
    log_s = rng.uniform(np.log(sigma_px[0]), np.log(sigma_px[1]), (P, 3))
    scales = (np.exp(log_s) / scale).astype(np.float32)
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(opacity[0], opacity[1], (P, 1)).astype(np.float32)
    col = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    Mf, Ms = view_matrices(z_cam)
    settings = dict(H=H, W=W, x_min=cb["x_min"], y_min=cb["y_min"], scale=scale, threshold=thr, viewmatrix=Mf,
                    viewmatrix_s=Ms, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, z_cam=z_cam)
    return dict(means3D=means, scales=scales, rotations=q.astype(np.float32), opacities=opac, colors=col,
                settings=settings)
