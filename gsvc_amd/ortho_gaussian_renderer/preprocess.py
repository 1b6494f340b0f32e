"""``prefilter_voxel`` — which anchors can reach the frame's z-slab and the screen.

Same signature and result as reference ortho_gaussian_renderer/preprocess.py:30-118: a bool[A] mask,
``rasterizer.visible_filter(anchors, exp(scaling)[:, :3], rotation) > 0``.  The reference also allocates an
unused [A,3] screen-space tensor on every call (:44-49); that is dropped.
"""
from __future__ import annotations

import torch

from ..rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def raster_settings_for(frame, pc, pipe, bg_color, scaling_modifier=1.0):
    """GaussianRasterizationSettings exactly as the reference builds them (renderer.py:63-83): the view matrix
    passed is ``frame.view_matrix.permute(1, 0)``.  Host tensors are kept on the host (the kernels take the
    16 floats by value), which removes the per-render H2D copy + sync of the reference.
    One addition: GSVC_RASTER_TIGHT_BINNING (include/gsvc_hip.h) — the lists hold a Gaussian only in the tiles its alpha >= 1/255 box
    touches; image, radii, num_rendered and gradients are those of the 3-sigma lists (tests/test_raster_gpu.py
    test_tight_binning_*), the binning, the sort and the backward's per-instance rows shrink by the 15 % (early) to 40 % (late in a
    fit) of instances that can contribute nothing.  GSVC_RASTER_LOOSE_BINNING=1 turns it off."""
    from .. import switches
    from .._lib import RASTER_TIGHT_BINNING
    tight = 0 if switches.RASTER_LOOSE_BINNING else RASTER_TIGHT_BINNING
    return GaussianRasterizationSettings(
        image_height=int(frame.image_height), image_width=int(frame.image_width), x_min=frame.x_min, y_min=frame.y_min,
        scale=frame.scale, threshold=pc.model_config.threshold, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=frame.view_matrix.permute(1, 0), sh_degree=pc.model_config.sh_degree, campos=frame.cam_pos,
        prefiltered=False, debug=getattr(pipe, "debug", False), flags=int(getattr(pipe, "raster_flags", 0) or 0) | tight,
        low_pass=float(getattr(pipe, "raster_low_pass", 0.0) or 0.0))


class RawGeometry(tuple):
    """(anchors, raw ``_scaling`` [A, 6], raw ``_rotation`` [A, 4]) for the multi-view visibility kernel, which applies the
    activations itself (``scale_exp``: exp; rotations: normalize).  Element 0 is the anchors as the generation pass reads them."""
    scale_exp = True


def _quantised_anchors(pc):
    """``pc.get_anchor`` (a dozen elementwise launches over all anchors); cached while the tensors it is computed from are the same
    objects at the same version — only where the owner declares the anchors static (``pc.anchor_static``: a Trainer that does not
    train them; an optimizer that writes through raw pointers does not bump versions)."""
    if pc.decoded_version:
        return pc._anchor.contiguous()
    if not getattr(pc, "anchor_static", False):
        return pc.get_anchor.contiguous()
    src = (pc._anchor, pc.x_bound_min, pc.x_bound_max)
    key = tuple((id(t), t._version) for t in src)
    hit = getattr(pc, "_anchor_q_cache", None)
    if hit is None or hit[0] != key:
        hit = (key, src, pc.get_anchor.contiguous())        # src: keeps the ids from being reused
        pc._anchor_q_cache = hit
    return hit[2]


def prefilter_geometry(pc):
    """(anchors, scales[:, :3], rotations) of all anchors as the visibility test reads them, without autograd.
    They do not change within a step, so a step that tests several views evaluates the getters once.  CUDA models get a
    ``RawGeometry``: the raw parameter tensors, whose activations the visibility kernel applies."""
    with torch.no_grad():
        sc, rot = getattr(pc, "_scaling", None), getattr(pc, "_rotation", None)
        if (sc is not None and rot is not None and pc._anchor.is_cuda and sc.dtype == rot.dtype == pc._anchor.dtype == torch.float32
                and sc.dim() == 2 and sc.shape[1] >= 3 and sc.is_contiguous() and rot.is_contiguous() and rot.shape[1] == 4):
            g = RawGeometry((_quantised_anchors(pc), sc, rot))
            g.scale_exp = not pc.decoded_version
            return g
        return pc.get_anchor.contiguous(), pc.get_scaling[:, :3].contiguous(), pc.get_rotation.contiguous()


def prefilter_voxels_many(frames, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, geometry=None):
    """``prefilter_voxel`` of several frames / views over the same anchors: a list of bool[A] masks (rows of one tensor), eight
    views per launch (csrc/raster_fwd.hip k_visible_masks)."""
    if getattr(pipe, "compute_cov3D_python", False):
        raise NotImplementedError("compute_cov3D_python is False in GSVC")
    geometry = geometry if geometry is not None else prefilter_geometry(pc)
    if not isinstance(geometry, RawGeometry):
        return [prefilter_voxel(f, pc, pipe, bg_color, scaling_modifier, geometry=geometry) for f in frames]
    import ctypes as C
    from .. import _lib
    from ..rasterizer import settings_to_c
    anchors, sc, rot = geometry
    A, R = int(anchors.shape[0]), len(frames)
    masks = torch.empty(R, A, dtype=torch.bool, device=anchors.device)
    for r0 in range(0, R, 8):
        cs = [settings_to_c(raster_settings_for(f, pc, pipe, bg_color, scaling_modifier)) for f in frames[r0:r0 + 8]]
        ptrs = (C.c_void_p * len(cs))(*[C.addressof(c) for c in cs])
        _lib.check(_lib.lib().gsvc_raster_visible_masks(ptrs, len(cs), A, _lib.ptr(anchors), _lib.ptr(sc), int(sc.shape[1]),
                                                        int(bool(geometry.scale_exp)), _lib.ptr(rot), 1, _lib.ptr(masks[r0:r0 + 8]),
                                                        _lib.current_stream(anchors.device)), "gsvc_raster_visible_masks")
    return list(masks.unbind(0))


def prefilter_voxel(frame, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, geometry=None):
    if getattr(pipe, "compute_cov3D_python", False):
        raise NotImplementedError("compute_cov3D_python is False in GSVC")
    geometry = geometry if geometry is not None else prefilter_geometry(pc)
    if isinstance(geometry, RawGeometry):
        return prefilter_voxels_many([frame], pc, pipe, bg_color, scaling_modifier, geometry=geometry)[0]
    rasterizer = GaussianRasterizer(raster_settings=raster_settings_for(frame, pc, pipe, bg_color, scaling_modifier))
    with torch.no_grad():
        means3D, scales, rotations = geometry
        radii_pure = rasterizer.visible_filter(means3D=means3D, scales=scales, rotations=rotations, cov3D_precomp=None)
    return radii_pure > 0
