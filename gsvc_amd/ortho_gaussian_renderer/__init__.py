"""Drop-in for GSVC's ``ortho_gaussian_renderer`` package (reference ortho_gaussian_renderer/__init__.py:1-3):
``render``, ``prefilter_voxel``, ``generate_neural_gaussians``, ``GenerateMode``, ``GeneratedGaussians``,
``RatePack``, ``calc_sampled_rate`` with the reference's signatures.
"""
from ..generate import (GenerateMode, GeneratedGaussians, RatePack, calc_sampled_rate,  # noqa: F401
                        generate_neural_gaussians)
from .preprocess import prefilter_voxel  # noqa: F401
from .renderer import plan_views, render, render_frames, render_many, render_pair  # noqa: F401
