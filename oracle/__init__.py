"""CPU oracle for the GSVC hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  Nothing under ``gsvc_amd/`` does, and the product path fails loudly when the HIP library is
missing instead of falling back to anything here.

Contents
  raster_oracle.c  restatement of the orthographic tile rasterizer (PARITY UNPINNED: source absent from
                   the reference, see the file header)
  grid_oracle.c    restatement of gridencoder.cu (reference submodules/gridencoder.zip)
  rate_oracle.py   numpy restatement of utils/entropy_models.py:32-68,159-175 and utils/encodings.py
                   quantisers
This module wraps the two C files through ctypes with numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsvc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.c", "grid_oracle.c")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.gsvc_oracle_raster_preprocess.restype = C.c_int64
        _lib.gsvc_oracle_raster_forward.restype = C.c_int64
        _lib.gsvc_oracle_raster_backward.restype = None
        _lib.gsvc_oracle_raster_backward_mt.restype = None
        _lib.gsvc_oracle_grid_forward.restype = C.c_int
        _lib.gsvc_oracle_grid_backward.restype = C.c_int
    return _lib


class RasterSettings(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("x_min", C.c_float),
        ("y_min", C.c_float),
        ("scale", C.c_float),
        ("threshold", C.c_float),
        ("scale_modifier", C.c_float),
        ("bg", C.c_float * 3),
        ("viewmatrix", C.c_float * 16),
        ("flags", C.c_uint32),
        ("low_pass", C.c_float),
    ]


def make_settings(H, W, x_min, y_min, scale, threshold, viewmatrix, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, flags=0,
                  low_pass=0.0):
    s = RasterSettings()
    s.image_height, s.image_width = int(H), int(W)
    s.x_min, s.y_min, s.scale, s.threshold = float(x_min), float(y_min), float(scale), float(threshold)
    s.scale_modifier = float(scale_modifier)
    s.bg[:] = [float(b) for b in bg]
    vm = np.asarray(viewmatrix, dtype=np.float32).reshape(16)
    s.viewmatrix[:] = [float(v) for v in vm]
    s.flags, s.low_pass = int(flags), float(low_pass)
    return s


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def view_matrices(z_cam: float):
    """The two view matrices of reference frame_cube/frame.py:18-43 (glm.lookAt towards -z and +z, up +y),
    written as row-major math matrices M with p_view = M[:3,:3] p + M[:3,3]."""
    Mf = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -z_cam], [0, 0, 0, 1]], dtype=np.float32)
    Ms = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, z_cam], [0, 0, 0, 1]], dtype=np.float32)
    return Mf, Ms


def raster_preprocess(st: RasterSettings, means3D, scales, rotations):
    means3D, scales, rotations = _f32(means3D), _f32(scales), _f32(rotations)
    P = means3D.shape[0]
    radii = np.zeros(P, dtype=np.int32)
    tiles = np.zeros(P, dtype=np.int32)
    total = lib().gsvc_oracle_raster_preprocess(C.byref(st), C.c_int64(P), _p(means3D), _p(scales), _p(rotations),
                                                _p(radii, C.c_int32), _p(tiles, C.c_int32))
    return radii, tiles, int(total)


@dataclass
class RasterForward:
    image: np.ndarray
    radii: np.ndarray
    num_rendered: int
    final_T: np.ndarray
    n_contrib: np.ndarray
    tile_ranges: np.ndarray
    point_list: np.ndarray
    geom: np.ndarray
    borderline: np.ndarray


def raster_forward(st: RasterSettings, means3D, colors, opacities, scales, rotations, num_threads: int = 0) -> RasterForward:
    means3D, colors, opacities = _f32(means3D), _f32(colors), _f32(opacities).reshape(-1)
    scales, rotations = _f32(scales), _f32(rotations)
    P = means3D.shape[0]
    H, W = st.image_height, st.image_width
    gx, gy = (W + 15) // 16, (H + 15) // 16
    _, _, total = raster_preprocess(st, means3D, scales, rotations)
    image = np.zeros((3, H, W), dtype=np.float32)
    radii = np.zeros(P, dtype=np.int32)
    final_T = np.zeros((H, W), dtype=np.float32)
    n_contrib = np.zeros((H, W), dtype=np.int32)
    ranges = np.zeros((gx * gy, 2), dtype=np.int32)
    plist = np.zeros(max(total, 1), dtype=np.int32)
    geom = np.zeros((P, 8), dtype=np.float32)
    border = np.zeros((H, W), dtype=np.uint8)
    n = lib().gsvc_oracle_raster_forward(
        C.byref(st), C.c_int64(P), _p(means3D), _p(colors), _p(opacities), _p(scales), _p(rotations),
        _p(image), _p(radii, C.c_int32), _p(final_T), _p(n_contrib, C.c_int32), _p(ranges, C.c_int32),
        C.c_int64(total), _p(plist, C.c_int32), _p(geom), _p(border, C.c_uint8), C.c_int(num_threads))
    # `total` counts every Gaussian the geometric test keeps; the forward also culls opacity <= 0, so n <= total
    assert 0 <= n <= total
    return RasterForward(image, radii, int(n), final_T, n_contrib, ranges, plist[:n], geom, border)


@dataclass
class RasterBackward:
    means3D: np.ndarray
    means2D: np.ndarray
    colors: np.ndarray
    opacities: np.ndarray
    scales: np.ndarray
    rotations: np.ndarray


def raster_backward(st: RasterSettings, means3D, colors, opacities, scales, rotations, fwd: RasterForward,
                    dL_dimage, num_threads: int = 1) -> RasterBackward:
    """``num_threads`` 1 (default): the serial statement every fixture was generated with.  > 1: tiles in parallel with one
    accumulator per (tile, Gaussian) instance, added per Gaussian in tile order (bench.py's CPU baseline; the same result for any
    thread count, a few 1e-16 relative from the serial sums)."""
    means3D, colors, opacities = _f32(means3D), _f32(colors), _f32(opacities).reshape(-1)
    scales, rotations, dL = _f32(scales), _f32(rotations), _f32(dL_dimage)
    P = means3D.shape[0]
    g3 = np.zeros((P, 3), np.float32)
    g2 = np.zeros((P, 3), np.float32)
    gc = np.zeros((P, 3), np.float32)
    go = np.zeros((P, 1), np.float32)
    gs = np.zeros((P, 3), np.float32)
    gq = np.zeros((P, 4), np.float32)
    plist = np.ascontiguousarray(fwd.point_list if fwd.point_list.size else np.zeros(1, np.int32))
    lib().gsvc_oracle_raster_backward_mt(
        C.byref(st), C.c_int64(P), _p(means3D), _p(colors), _p(opacities), _p(scales), _p(rotations),
        _p(fwd.radii, C.c_int32), _p(fwd.tile_ranges, C.c_int32), _p(plist, C.c_int32), _p(fwd.final_T),
        _p(fwd.n_contrib, C.c_int32), _p(dL), _p(g3), _p(g2), _p(gc), _p(go), _p(gs), _p(gq), C.c_int(int(num_threads)))
    return RasterBackward(g3, g2, gc, go, gs, gq)


# ------------------------------------------------------------------------------------------ hash grid
def grid_forward(inputs, embeddings, offsets, resolutions, calc_dy_dx: bool = False):
    """Returns outputs [L,N,C] (the reference's native layout) and dy_dx [N, L*D*C] or None."""
    inputs, embeddings = _f32(inputs), _f32(embeddings)
    offsets = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
    resolutions = np.ascontiguousarray(np.asarray(resolutions, dtype=np.int32))
    N, D = inputs.shape
    Cf = embeddings.shape[1]
    L = resolutions.shape[0]
    out = np.empty((L, N, Cf), dtype=np.float32)
    dy = np.empty((N, L * D * Cf), dtype=np.float32) if calc_dy_dx else None
    rc = lib().gsvc_oracle_grid_forward(_p(inputs), _p(embeddings), _p(offsets, C.c_int32), _p(resolutions, C.c_int32),
                                        _p(out), C.c_uint32(N), C.c_uint32(D), C.c_uint32(Cf), C.c_uint32(L),
                                        _p(dy) if dy is not None else None)
    if rc != 0:
        raise RuntimeError("GridEncoding: n_fearures must be 1, 2, 4, 8, 16 or 32." if rc == -2
                           else "GridEncoding: num_dim must be 1, 2, 3.")
    return out, dy


def grid_backward(grad, inputs, embeddings, offsets, resolutions, dy_dx=None):
    """grad [L,N,C] -> (grad_embeddings [rows,C], grad_inputs [N,D] or None)."""
    grad, inputs, embeddings = _f32(grad), _f32(inputs), _f32(embeddings)
    offsets = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
    resolutions = np.ascontiguousarray(np.asarray(resolutions, dtype=np.int32))
    N, D = inputs.shape
    Cf = embeddings.shape[1]
    L = resolutions.shape[0]
    ge = np.zeros_like(embeddings)
    gi = np.zeros_like(inputs) if dy_dx is not None else None
    dy = _f32(dy_dx) if dy_dx is not None else None
    rc = lib().gsvc_oracle_grid_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets, C.c_int32),
                                         _p(resolutions, C.c_int32), _p(ge), C.c_uint32(N), C.c_uint32(D),
                                         C.c_uint32(Cf), C.c_uint32(L), _p(dy) if dy is not None else None,
                                         _p(gi) if gi is not None else None)
    if rc != 0:
        raise RuntimeError("grid backward: unsupported num_dim / n_features")
    return ge, gi
