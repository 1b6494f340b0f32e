"""API-name alias of the legacy ``gaussian_renderer`` package.

The reference's gaussian_renderer/__init__.py is dead code (nothing imports it; it references an undefined
PyGaussianRasterizer at :295 and a never-created pc.mlp_grid).  BASELINE.json names the
package, so the three public names exist here and delegate to the orthographic path.
"""
from ..ortho_gaussian_renderer import generate_neural_gaussians, prefilter_voxel, render  # noqa: F401
