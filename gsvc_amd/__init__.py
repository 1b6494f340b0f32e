"""gsvc_amd — MI355X-native hot path of GSVC (orthographic sliding-window Gaussian rasterizer, hash-grid
encoder, entropy-rate estimator and the fitting step around them) behind GSVC's own renderer API.

The compute path is hand-written HIP for gfx950 in ``gsvc_amd/csrc`` behind the C-ABI of
``include/gsvc_hip.h``; there is no CPU fallback: importing a compute entry point without the built
library raises.
"""
__version__ = "0.1.0"
