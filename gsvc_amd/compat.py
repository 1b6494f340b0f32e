"""Make unmodified GSVC import this package in place of its two native extensions.

    import gsvc_amd.compat; gsvc_amd.compat.install()
    # from now on, in GSVC's own files:
    #   import _gridencoder as _backend                                   (utils/encodings.py:21)
    #   from diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer import
    #        GaussianRasterizationSettings, GaussianRasterizer            (ortho_gaussian_renderer/renderer.py:6)
    # resolve to the gfx950 HIP kernels of libgsvc_hip.so.

``install(renderer=True)`` additionally aliases GSVC's ``ortho_gaussian_renderer`` / ``gaussian_renderer``
package names to this package's renderer (same signatures), for callers that want the whole render path.
"""
from __future__ import annotations

import sys
import types


def install(renderer: bool = False):
    from . import gridencoder_backend, rasterizer
    sys.modules["_gridencoder"] = gridencoder_backend
    pkg = types.ModuleType("diff_gaussian_rasterization")
    pkg.__path__ = []  # mark as package
    pkg.cuda_ortho_gaussian_rasterizer = rasterizer
    sys.modules["diff_gaussian_rasterization"] = pkg
    sys.modules["diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer"] = rasterizer
    if renderer:
        from . import gaussian_renderer, ortho_gaussian_renderer
        sys.modules["ortho_gaussian_renderer"] = ortho_gaussian_renderer
        sys.modules["gaussian_renderer"] = gaussian_renderer
