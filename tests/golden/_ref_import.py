"""Import the reference's Python on CPU (THIS container only) to generate golden vectors.

Nothing here travels as reference code: it stubs the third-party modules the reference imports at module
load (loguru, plyfile, torchac, ...), points `_gridencoder` and the rasterizer slot at OUR backends, and
patches `.cuda()` to a no-op so the reference's own arithmetic runs on PyTorch-CPU.  Used only by
make_golden.py; the GPU box has no /root/reference and never imports this file's products except the
committed .npz/.pt fixtures.
"""
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, k):
        return _Any()


class CpuMode(TorchFunctionMode):
    """Rewrites device='cuda' keyword arguments to 'cpu'."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = kwargs or {}
        if str(kwargs.get("device", "")).startswith("cuda"):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def _oracle_gridencoder_module():
    """`_gridencoder` backed by oracle/grid_oracle.c (numpy <-> torch CPU tensors)."""
    import oracle
    m = types.ModuleType("_gridencoder")

    def grid_encode_forward(inputs, embeddings, offsets_list, resolutions_list, outputs, N, num_dim, n_features,
                            n_levels, max_level, Rb, PV, dy_dx, binary_vxl, min_level_id):
        assert binary_vxl is None and min_level_id is None
        out, dy = oracle.grid_forward(inputs.detach().numpy(), embeddings.detach().numpy(), offsets_list.numpy(),
                                      resolutions_list.numpy(), calc_dy_dx=dy_dx is not None)
        outputs.copy_(torch.from_numpy(out))
        if dy_dx is not None:
            dy_dx.copy_(torch.from_numpy(dy))

    def grid_encode_backward(grad, inputs, embeddings, offsets_list, resolutions_list, grad_embeddings, N, num_dim,
                             n_features, n_levels, max_level, Rb, dy_dx, grad_inputs, binary_vxl, min_level_id):
        assert binary_vxl is None and min_level_id is None
        ge, gi = oracle.grid_backward(grad.detach().numpy(), inputs.detach().numpy(), embeddings.detach().numpy(),
                                      offsets_list.numpy(), resolutions_list.numpy(),
                                      dy_dx.detach().numpy() if dy_dx is not None else None)
        grad_embeddings.add_(torch.from_numpy(ge))
        if grad_inputs is not None:
            grad_inputs.copy_(torch.from_numpy(gi))

    m.grid_encode_forward = grid_encode_forward
    m.grid_encode_backward = grid_encode_backward
    return m


def _oracle_rasterizer_module(flags: int = 0, low_pass: float = 0.0):
    """(``flags`` / ``low_pass``: conventions the module keeps to ITSELF — its settings type has no such fields, like the real
    extension's — for the calibrator's self-test: tools/calibrate_conventions.py must recover them from results alone.)
    `diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer` backed by oracle/raster_oracle.c: the settings type with the
    fields the reference's call sites construct it from (renderer.py:63-83, preprocess.py:58-83), the rasterizer as an autograd
    node (image, radii, num_rendered; means2D receives the screen-space gradient) and visible_filter.  With this in the slot
    the UNMODIFIED reference render() / prefilter_voxel() run on PyTorch-CPU (make_golden_prod.py)."""
    from typing import NamedTuple
    import oracle
    m = types.ModuleType("diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer")

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        x_min: float
        y_min: float
        scale: float
        threshold: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    def _settings(rs):
        return oracle.make_settings(rs.image_height, rs.image_width, rs.x_min, rs.y_min, rs.scale, rs.threshold,
                                    rs.viewmatrix.detach().contiguous().numpy(), bg=[float(v) for v in rs.bg],
                                    scale_modifier=rs.scale_modifier, flags=flags, low_pass=low_pass)

    class _Rasterize(torch.autograd.Function):
        @staticmethod
        def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, rs, holder):
            st = _settings(rs)
            a = [t.detach().contiguous().numpy() for t in (means3D, colors, opacities, scales, rotations)]
            fwd = oracle.raster_forward(st, *a)
            ctx.st, ctx.arrays, ctx.fwd = st, a, fwd
            holder["forward"] = fwd
            radii = torch.from_numpy(fwd.radii.copy())
            ctx.mark_non_differentiable(radii)
            return torch.from_numpy(fwd.image.copy()), radii

        @staticmethod
        def backward(ctx, g_image, _g_radii):
            b = oracle.raster_backward(ctx.st, *ctx.arrays, ctx.fwd, g_image.detach().contiguous().numpy())
            t = torch.from_numpy
            return t(b.means3D), t(b.means2D), t(b.colors), t(b.opacities), t(b.scales), t(b.rotations), None, None

    class GaussianRasterizer(torch.nn.Module):
        last = {}                                  # the oracle's forward state of the most recent call (borderline mask)

        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
            assert cov3D_precomp is None
            radii, _, _ = oracle.raster_preprocess(_settings(self.raster_settings), means3D.detach().contiguous().numpy(),
                                                   scales.detach().contiguous().numpy(), rotations.detach().contiguous().numpy())
            return torch.from_numpy(radii.copy())

        def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3D_precomp=None):
            assert shs is None and cov3D_precomp is None
            image, radii = _Rasterize.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self.raster_settings,
                                            GaussianRasterizer.last)
            return image, radii, GaussianRasterizer.last["forward"].num_rendered

    m.GaussianRasterizationSettings = GaussianRasterizationSettings
    m.GaussianRasterizer = GaussianRasterizer
    return m


def install(rasterizer: bool = False):
    """Install stubs + patches; returns the CpuMode context to wrap reference calls in.  ``rasterizer``: fill the rasterizer slot
    with the oracle (the reference's render() / prefilter_voxel() then run); otherwise it stays a permissive stub."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    names = ["loguru", "plyfile", "torchac", "constriction", "gsvc_cuda_ans", "colorama", "glm", "dahuffman",
             "torch_scatter", "simple_knn", "simple_knn._C", "diff_gaussian_rasterization",
             "diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer", "torchvision", "torchvision.transforms",
             "torchvision.utils", "lpips", "pytorch_msssim", "skimage", "skimage.metrics", "simple_parsing"]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["loguru"].logger = _Any()
    sys.modules["plyfile"].PlyData = _Any
    sys.modules["plyfile"].PlyElement = _Any
    sys.modules["gsvc_cuda_ans"].ANSCoder = _Any
    sys.modules["dahuffman"].HuffmanCodec = _Any
    sys.modules["torch_scatter"].scatter_max = _Any()
    sys.modules["simple_knn._C"].distCUDA2 = lambda pts: torch.full((pts.shape[0],), 1e-4)
    sys.modules["simple_knn"]._C = sys.modules["simple_knn._C"]
    sys.modules["colorama"].Fore = _Any()
    sys.modules["colorama"].Style = _Any()
    sys.modules["colorama"].init = _Any()
    sys.modules["torchvision.transforms"].ToTensor = _Any
    sys.modules["torchvision.transforms"].transforms = _Any()
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.utils"].save_image = _Any()
    sys.modules["torchvision.utils"].make_grid = _Any()
    sys.modules["lpips"].LPIPS = _Any
    sys.modules["pytorch_msssim"].ms_ssim = _Any()
    sys.modules["skimage.metrics"].peak_signal_noise_ratio = _Any()
    sys.modules["skimage.metrics"].structural_similarity = _Any()
    sys.modules["skimage"].metrics = sys.modules["skimage.metrics"]
    sp = sys.modules["simple_parsing"]
    sp.ArgumentParser = _Any
    sp.field = lambda *a, **k: k.get("default", None)
    sys.modules["diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer"].GaussianRasterizationSettings = _Any
    sys.modules["diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer"].GaussianRasterizer = _Any
    sys.modules["_gridencoder"] = _oracle_gridencoder_module()
    if rasterizer:
        sys.modules["diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer"] = _oracle_rasterizer_module()
        sys.modules["glm"].vec3 = _Any                      # frame_cube/frame.py is imported for its Frame dataclass only
        sys.modules["glm"].lookAt = _Any()

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    return CpuMode()
