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
    16 floats by value), which removes the per-render H2D copy + sync of the reference."""
    return GaussianRasterizationSettings(
        image_height=int(frame.image_height), image_width=int(frame.image_width), x_min=frame.x_min, y_min=frame.y_min,
        scale=frame.scale, threshold=pc.model_config.threshold, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=frame.view_matrix.permute(1, 0), sh_degree=pc.model_config.sh_degree, campos=frame.cam_pos,
        prefiltered=False, debug=getattr(pipe, "debug", False), flags=int(getattr(pipe, "raster_flags", 0) or 0),
        low_pass=float(getattr(pipe, "raster_low_pass", 0.0) or 0.0))


def prefilter_geometry(pc):
    """(anchors, scales[:, :3], rotations) of all anchors as the visibility test reads them, without autograd.
    They do not change within a step, so a step that tests several views evaluates the getters once."""
    with torch.no_grad():
        return pc.get_anchor.contiguous(), pc.get_scaling[:, :3].contiguous(), pc.get_rotation.contiguous()


def prefilter_voxel(frame, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, geometry=None):
    if getattr(pipe, "compute_cov3D_python", False):
        raise NotImplementedError("compute_cov3D_python is False in GSVC")
    rasterizer = GaussianRasterizer(raster_settings=raster_settings_for(frame, pc, pipe, bg_color, scaling_modifier))
    with torch.no_grad():
        means3D, scales, rotations = geometry if geometry is not None else prefilter_geometry(pc)
        radii_pure = rasterizer.visible_filter(means3D=means3D, scales=scales, rotations=rotations, cov3D_precomp=None)
    return radii_pure > 0
