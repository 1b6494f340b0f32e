"""Seeded inputs shared by the golden-vector generator that drives the reference (make_golden_prod.py, build container only) and by
the GPU tests that replay it (tests/test_prod_fixture_gpu.py): both sides regenerate the SAME model parameters, anchors and random
draws from seeds instead of carrying megabytes of inputs in the fixture.  Own code — nothing here comes from the reference.

All generation happens on the CPU with explicit ``torch.Generator`` objects (the CPU generator's stream for a given seed is a
property of the torch build, which is the same image on both boxes); the caller moves the tensors where it needs them.
"""
import zlib

import numpy as np
import torch

PROD = dict(feat_dim=50, n_offsets=10, n_features_per_level=8, log2_hashmap_size=9, log2_hashmap_size_2D=9,
            resolutions_list=(18, 24, 33, 44, 59, 80, 108, 148, 201, 275, 376, 514), resolutions_list_2D=(130, 258, 514, 1026))
SCENE = dict(A=6000, H=96, W=160, T=64, frame=40, threshold=0.08, seed=2024)


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))


def frame_numbers(H: int, W: int, T: int, idx: int):
    """The frame's numbers by the formulas of reference frame_cube/frame.py:92-101,156-190 and the two view matrices of
    frame.py:18-43 (glm.lookAt towards -z / +z with up +y; np.array(glm.mat4) is column-major, so the STORED tensor is the
    transpose of the row-major math matrix — SURVEY G13)."""
    scale = max(H, W, T) / 2
    z = (idx - T / 2) / scale
    Mf = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -z], [0, 0, 0, 1]], dtype=np.float32)
    Ms = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, z], [0, 0, 0, 1]], dtype=np.float32)
    return dict(x_min=-W / 2 / scale, y_min=-H / 2 / scale, z_min=-T / 2 / scale, scale=scale, z=z,
                view_matrix=torch.from_numpy(Mf.T.copy()), view_matrix_s=torch.from_numpy(Ms.T.copy()),
                cam_pos=torch.tensor([0.0, 0.0, z], dtype=torch.float32))


def anchors(A: int, fn: dict, threshold: float, seed: int) -> dict:
    """Per-anchor parameters: four fifths of the anchors inside the frame's z-slab, the rest outside."""
    g = _gen("anchors", seed)
    xy = (torch.rand(A, 2, generator=g) * 2 - 1) * torch.tensor([-fn["x_min"], -fn["y_min"]], dtype=torch.float32)
    u = torch.rand(A, generator=g)
    inside = fn["z"] + (torch.rand(A, generator=g) * 2 - 1) * 0.9 * threshold
    outside = fn["z"] + torch.sign(torch.rand(A, generator=g) - 0.5) * threshold * (1.5 + 2 * torch.rand(A, generator=g))
    z = torch.where(u < 0.8, inside, outside).clamp(fn["z_min"] * 0.99, -fn["z_min"] * 0.99)
    rot = torch.zeros(A, 4)
    rot[:, 0] = 1
    return {"_anchor": torch.cat([xy, z.unsqueeze(1)], dim=1).float(),
            "_offset": torch.randn(A, PROD["n_offsets"], 3, generator=g) * 0.5,
            "_mask": torch.randn(A, PROD["n_offsets"], 1, generator=g) * 3,
            "_anchor_feat": torch.randn(A, PROD["feat_dim"], generator=g),
            "_scaling": torch.randn(A, 6, generator=g) * 0.3 - 4.0,
            "_rotation": rot, "_opacity": torch.zeros(A, 1)}


def anchors_uniform(A: int, fn: dict, seed: int) -> dict:
    """Per-anchor parameters with z uniform over the whole cube (every 0.01 z-slab of the stream codec is populated: the
    reference's encoder fails on an empty slab) and a third of the offset masks switched off."""
    d = anchors(A, fn, 1.0, seed + 1)
    g = _gen("anchors_uniform", seed)
    d["_anchor"][:, 2] = (torch.rand(A, generator=g) * 2 - 1) * (-fn["z_min"]) * 0.98
    return d


def fill_parameters(module, seed: int):
    """Every floating-point tensor of ``module.state_dict()`` that is not a per-anchor tensor, filled from a generator seeded by
    its NAME (the two implementations share the state_dict keys, not the construction order): linear weights U(+-1.7/sqrt(fan_in)),
    biases U(+-0.1), hash tables U(+-1.2)."""
    with torch.no_grad():
        for name, t in sorted(module.state_dict().items()):
            if name.startswith("_") or not t.is_floating_point() or t.numel() == 0:
                continue
            g = _gen(name, seed)
            if name.endswith("params"):                     # a hash table (binarised by STE_binary at use)
                v = torch.rand(t.shape, generator=g) * 2.4 - 1.2
            elif t.dim() == 2:
                v = (torch.rand(t.shape, generator=g) * 2 - 1) * (1.7 / t.shape[1] ** 0.5)
            elif t.dim() == 1 and name.endswith("bias"):
                v = (torch.rand(t.shape, generator=g) * 2 - 1) * 0.1
            else:
                continue                                    # buffers (frequency bands, offsets): as constructed
            t.copy_(v.to(t.device, t.dtype))


class SeededDraws:
    """``torch.rand_like`` / ``Tensor.uniform_`` answer the i-th draw of the block from generator(seed + i), whatever the
    device and the shape (a draw is numel() values laid out row-major): the reference on the CPU and the HIP path on the GPU see
    the same noise without a tape in the fixture.  ``count`` = draws taken so far."""

    def __init__(self, seed: int):
        self.seed, self.count = seed, 0
        self._rand_like, self._uniform = torch.rand_like, torch.Tensor.uniform_

    def _next(self, like):
        u = torch.rand(like.numel(), generator=torch.Generator().manual_seed(self.seed + self.count))
        self.count += 1
        return u.view(like.shape).to(like.device)

    def __enter__(self):
        me = self

        def rand_like(x, *a, **k):
            return me._next(x)

        def uniform_(t, lo=0.0, hi=1.0, **k):
            return t.copy_(me._next(t) * (hi - lo) + lo)

        torch.rand_like, torch.Tensor.uniform_ = rand_like, uniform_
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.Tensor.uniform_ = self._rand_like, self._uniform


def image_weights(H: int, W: int, seed: int) -> torch.Tensor:
    """The fixed dL/dimage of the fixture's backward passes."""
    return torch.randn(3, H, W, generator=_gen("dL_dimage", seed))


def gt_image(H: int, W: int, idx: int, seed: int) -> torch.Tensor:
    """Ground-truth picture [3, H, W] of frame ``idx`` in [0, 1]: a coarse random grid bilinearly enlarged (smooth, so that the
    structural-similarity term is not degenerate) plus a little fine noise."""
    g = _gen(f"gt_image_{idx}", seed)
    coarse = torch.rand(1, 3, max(2, H // 12), max(2, W // 12), generator=g)
    img = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True)[0]
    return (img + 0.05 * torch.randn(3, H, W, generator=g)).clamp(0, 1).contiguous()


def optical_flow(H: int, W: int, idx: int, seed: int) -> torch.Tensor:
    """Backward optical flow [2, H, W] in pixels per frame between frames ``idx`` and ``idx + 1`` (what the reference's
    FrameCubeDataset.get_optical_flow hands to calc_optical_loss)."""
    return torch.randn(2, H, W, generator=_gen(f"flow_{idx}", seed)) * 1.5


STEP = dict(lmbda=0.004, opacity_reg=0.003,            # opacity_reg is 0 by default: non-zero here so that its weight is pinned too
            iterations={0: 1601, 1: 12001, 2: 16001, 3: 35001})      # one iteration inside each phase of the default schedule


def rows(t: torch.Tensor, stride: int):
    return t[::stride]
