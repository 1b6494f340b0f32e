"""Drop-in for the native module ``_gridencoder`` (reference submodules/gridencoder.zip, bindings.cpp:5-13).

Same two entry points, same positional arguments and the same caller-allocates contract as the reference's
pybind module (``gridencoder.h:12-36``; call sites reference utils/encodings.py:529-553,582-610):

    grid_encode_forward(inputs, embeddings, offsets_list, resolutions_list, outputs, N, num_dim, n_features,
                        n_levels, max_level, Rb, PV, dy_dx, binary_vxl, min_level_id)
    grid_encode_backward(grad, inputs, embeddings, offsets_list, resolutions_list, grad_embeddings, N, num_dim,
                         n_features, n_levels, max_level, Rb, dy_dx, grad_inputs, binary_vxl, min_level_id)

Error behaviour mirrors the TORCH_CHECKs of gridencoder.cu:15-18,1032-1050 (RuntimeError for non-CUDA,
non-contiguous, wrong dtype) and the std::runtime_error of :909,937 for unsupported n_features / num_dim.
``binary_vxl`` and a tensor ``min_level_id`` are never passed by GSVC (encodings.py:497,528-540) and raise
NotImplementedError.  Only float32 embeddings are supported (GSVC never enables autocast).
"""
from __future__ import annotations

import torch

from . import _lib


def _check(t, name, *, floating=False, integer=False):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if floating and t.dtype not in (torch.float32, torch.float16, torch.float64):
        raise RuntimeError(f"{name} must be a floating tensor")
    if integer and t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def _f32_only(t, name):
    if t.dtype != torch.float32:
        raise NotImplementedError(f"{name}: only float32 is implemented on gfx950 (GSVC never uses half/double here)")


def grid_encode_forward(inputs, embeddings, offsets_list, resolutions_list, outputs, N, num_dim, n_features,
                        n_levels, max_level, Rb, PV, dy_dx=None, binary_vxl=None, min_level_id=None):
    for t, n in ((inputs, "inputs"), (embeddings, "embeddings"), (outputs, "outputs")):
        _check(t, n, floating=True)
    for t, n in ((offsets_list, "offsets_list"), (resolutions_list, "resolutions_list")):
        _check(t, n, integer=True)
    if binary_vxl is not None or min_level_id is not None:
        raise NotImplementedError("binary_vxl / tensor min_level_id are not used by GSVC and not implemented")
    for t, n in ((inputs, "inputs"), (embeddings, "embeddings"), (outputs, "outputs")):
        _f32_only(t, n)
    if dy_dx is not None:
        _check(dy_dx, "dy_dx", floating=True)
        _f32_only(dy_dx, "dy_dx")
    rc = _lib.lib().gsvc_grid_forward(_lib.ptr(inputs), _lib.ptr(embeddings), _lib.ptr(offsets_list),
                                      _lib.ptr(resolutions_list), _lib.ptr(outputs), int(N), int(num_dim),
                                      int(n_features), int(n_levels), _lib.ptr(dy_dx), _lib.current_stream(inputs.device))
    _lib.check(rc, "grid_encode_forward")


def grid_encode_backward(grad, inputs, embeddings, offsets_list, resolutions_list, grad_embeddings, N, num_dim,
                         n_features, n_levels, max_level, Rb, dy_dx=None, grad_inputs=None, binary_vxl=None,
                         min_level_id=None):
    for t, n in ((grad, "grad"), (inputs, "inputs"), (embeddings, "embeddings"), (grad_embeddings, "grad_embeddings")):
        _check(t, n, floating=True)
        _f32_only(t, n)
    for t, n in ((offsets_list, "offsets_list"), (resolutions_list, "resolutions_list")):
        _check(t, n, integer=True)
    if binary_vxl is not None or min_level_id is not None:
        raise NotImplementedError("binary_vxl / tensor min_level_id are not used by GSVC and not implemented")
    if dy_dx is not None:
        _check(dy_dx, "dy_dx", floating=True)
        _check(grad_inputs, "grad_inputs", floating=True)
    rc = _lib.lib().gsvc_grid_backward(_lib.ptr(grad), _lib.ptr(inputs), _lib.ptr(embeddings), _lib.ptr(offsets_list),
                                       _lib.ptr(resolutions_list), _lib.ptr(grad_embeddings), int(N), int(num_dim),
                                       int(n_features), int(n_levels), _lib.ptr(dy_dx),
                                       _lib.ptr(grad_inputs) if dy_dx is not None else None,
                                       _lib.current_stream(inputs.device))
    _lib.check(rc, "grid_encode_backward")
