"""Drop-in for ``diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer`` (external CUDA extension of
GSVC, reference README.md:52) backed by the gfx950 HIP kernels of ``csrc/raster_{fwd,bwd}.hip``.

Mirrors the surface GSVC uses:
  * ``GaussianRasterizationSettings(image_height, image_width, x_min, y_min, scale, threshold, bg,
    scale_modifier, viewmatrix, sh_degree, campos, prefiltered, debug)`` — keyword construction,
    reference ortho_gaussian_renderer/renderer.py:63-83;
  * ``GaussianRasterizer(raster_settings=...)(means3D=, means2D=, shs=, colors_precomp=, opacities=,
    scales=, rotations=, cov3D_precomp=) -> (image[3,H,W], radii[P], num_rendered)``, renderer.py:85-98;
  * ``GaussianRasterizer.visible_filter(means3D=, scales=, rotations=, cov3D_precomp=) -> radii``,
    reference ortho_gaussian_renderer/preprocess.py:99-104;
  * an autograd node: ``means2D.grad`` receives the screen-space gradient (renderer.py:37-42,
    scene/gaussian_model.py:1311).

Error behaviour follows the 3DGS lineage: missing scales/rotations, or both/neither of shs and
colors_precomp, raise ``Exception``; shs and cov3D_precomp are not used by GSVC (it always passes
colors_precomp / scales+rotations) and raise NotImplementedError here.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import NamedTuple, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib, switches


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
    sh_degree: int = 0
    campos: Optional[torch.Tensor] = None
    prefiltered: bool = False
    debug: bool = False
    # not fields of the reference's settings: the convention switches of include/gsvc_hip.h (GSVC_RASTER_*, 0 = DESIGN.md's
    # raster spec) and the low-pass added to the 2-D covariance (0 = the default 0.3), see INTEGRATION.md
    flags: int = 0
    low_pass: float = 0.0


def _host_floats(t, n):
    if isinstance(t, torch.Tensor):
        a = t.detach().to("cpu", torch.float32).contiguous().view(-1).numpy()  # syncs if t lives on the GPU
    else:
        a = np.asarray(t, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


def settings_to_c(rs: GaussianRasterizationSettings) -> _lib.RasterSettingsC:
    s = _lib.RasterSettingsC()
    s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
    s.x_min, s.y_min, s.scale = float(rs.x_min), float(rs.y_min), float(rs.scale)
    s.threshold, s.scale_modifier = float(rs.threshold), float(rs.scale_modifier)
    s.bg[:] = _host_floats(rs.bg, 3).tolist()
    s.viewmatrix[:] = _host_floats(rs.viewmatrix, 16).tolist()
    s.flags, s.low_pass = int(getattr(rs, "flags", 0)), float(getattr(rs, "low_pass", 0.0))
    return s


def _as_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class RasterState:
    """Opaque state of one forward (geom / binning / image blobs) kept for backward and inspection."""

    def __init__(self, cs, P, max_instances, geom, binning, image_state, radii):
        self.cs, self.P, self.max_instances = cs, P, max_instances
        self.geom, self.binning, self.image_state, self.radii = geom, binning, image_state, radii
        self._counters = self._listed = None
        self._host = self._event = None     # early asynchronous read-back of the counters (raster_forward, sync=False)

    def counters(self):
        """(num_rendered, overflow, num_visible, max_tile_len) — one 16-byte D2H copy (synchronises), cached."""
        if self._counters is None:
            c = self.binning[:32].view(torch.int32).tolist()
            self._counters, self._listed = tuple(int(v) for v in c[:4]), int(c[7])
        return self._counters

    def tile_lists(self):
        """(tile_offsets[T+1], point_list[instances listed]) as int32 tensors (views into the binning blob).  The lists' length is
        tile_offsets[T]: num_rendered counts the 3-sigma rectangles' tiles, which is more under GSVC_RASTER_TIGHT_BINNING."""
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().gsvc_raster_binning_layout(C.byref(self.cs), self.P, self.max_instances, C.byref(a), C.byref(b)),
                   "gsvc_raster_binning_layout")
        H, W = self.cs.image_height, self.cs.image_width
        T = ((H + 15) // 16) * ((W + 15) // 16)
        if self.counters()[1]:
            raise _lib.GsvcError("tile_lists: the forward overflowed its instance buffer (no lists were written)")
        off = self.binning[a.value:a.value + 4 * (T + 1)].view(torch.int32)
        n = int(off[T])
        pl = self.binning[b.value:b.value + 4 * n].view(torch.int32)
        return off, pl

    def listed_instances(self) -> int:
        """(tile, Gaussian) instances the tile lists hold = tile_offsets[T]: what the sort, the compositing kernels and the backward's
        rows work on.  Equal to num_rendered for the 3-sigma lists, fewer under GSVC_RASTER_TIGHT_BINNING (one 4-byte copy: synchronises)."""
        if self._listed is None:          # (the counters' read-back carries it: gsvc_raster_counters.reserved[2])
            self.counters()
        return self._listed

    def image_aux(self):
        a, b = C.c_uint64(), C.c_uint64()
        _lib.check(_lib.lib().gsvc_raster_image_layout(C.byref(self.cs), C.byref(a), C.byref(b)), "gsvc_raster_image_layout")
        H, W = self.cs.image_height, self.cs.image_width
        gy, gx = (H + 15) // 16, (W + 15) // 16
        n = gy * gx * 256

        def rows(t):      # the blob keeps them tile-major [tile_y, tile_x, quad_y, quad_x, lane_y, lane_x]
            return t.view(gy, gx, 2, 2, 8, 8).permute(0, 2, 4, 1, 3, 5).reshape(gy * 16, gx * 16)[:H, :W].contiguous()
        fT = rows(self.image_state[a.value:a.value + 4 * n].view(torch.float32))
        nc = rows(self.image_state[b.value:b.value + 4 * n].view(torch.int32))
        return fT, nc


_capacity_hint = {}


def raster_forward(cs: _lib.RasterSettingsC, means3D, colors, opacities, scales, rotations, max_instances=None,
                   sync=True, pair=False, readback=False, radii_out=None, side_stream=None):
    """Launch the forward pipeline.  Returns (image, radii, state).  With ``sync`` the instance counters
    are read back (16 B) and the call is repeated with a larger instance capacity if it overflowed; without
    it the caller must check ``state.counters()[1]`` itself (``readback``: the counters' copy to the host is queued right
    behind the forward, for resolve_deferred()).  ``pair=True`` returns the two-view frame
    (render(view) + flip(render(opposite view))) / 2 from one pass (inference only, see gsvc_raster_forward_pair)."""
    L = _lib.lib()
    P = int(means3D.shape[0])
    dev = means3D.device
    H, W = cs.image_height, cs.image_width
    key = (H, W)
    if max_instances is None:
        max_instances = max(_capacity_hint.get(key, 0), 4 * P, 1 << 16)
    # side_stream (a torch.cuda.Stream the caller has made wait for the inputs, and waits for afterwards): the kernels go there;
    # every buffer is still allocated on the CURRENT stream's pool (the caller's join orders any reuse behind the side work)
    stream = _lib.current_stream(dev) if side_stream is None else C.c_void_p(side_stream.cuda_stream)
    while True:
        sizes = _lib.RasterSizesC()
        _lib.check(L.gsvc_raster_sizes_query(C.byref(cs), P, max_instances, C.byref(sizes)), "gsvc_raster_sizes_query")
        geom = torch.empty(sizes.geom_bytes, dtype=torch.uint8, device=dev)
        binning = torch.empty(sizes.binning_bytes, dtype=torch.uint8, device=dev)
        image_state = torch.empty(sizes.image_bytes, dtype=torch.uint8, device=dev)
        image = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        radii = radii_out if radii_out is not None else torch.empty(P, dtype=torch.int32, device=dev)
        fn = L.gsvc_raster_forward_pair if pair else L.gsvc_raster_forward
        _lib.check(fn(C.byref(cs), P, max_instances, _lib.ptr(means3D), _lib.ptr(colors),
                                         _lib.ptr(opacities), _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(image),
                                         _lib.ptr(radii), _lib.ptr(geom), _lib.ptr(binning), _lib.ptr(image_state), stream),
                   "gsvc_raster_forward")
        state = RasterState(cs, P, max_instances, geom, binning, image_state, radii)
        if not sync and readback:
            # the counters are final once the forward kernels have run: their 16 bytes start travelling to the host
            # now, behind the forward only, so that resolve_deferred() later waits for THIS copy and not for
            # everything queued after it (the whole backward of a fitting step)
            state._host = torch.empty(8, dtype=torch.int32, pin_memory=True)
            src = binning[:32].view(torch.int32)
            with torch.cuda.stream(side_stream) if side_stream is not None else contextlib.nullcontext():
                state._host.copy_(src, non_blocking=True)
                state._event = torch.cuda.Event()
                state._event.record()
        if not sync:
            return image, radii, state
        n, overflow, _, _ = state.counters()
        n = state._listed or n          # the LISTS' length decides the capacity (shorter than num_rendered under tight binning: ADVICE r05)
        if not overflow:
            _capacity_hint[key] = max(_capacity_hint.get(key, 0), int(n * 1.25) + 1024)
            return image, radii, state
        max_instances = int(n * 1.25) + 1024


def resolve_deferred(states):
    """Read back, with ONE host synchronisation, the instance counters of forwards launched with ``sync=False``.
    Returns (num_rendered list, overflowed: bool); on overflow the capacity hint is raised so that repeating the
    forwards succeeds — their images (and anything computed from them) must be discarded."""
    if not states:
        return [], False
    if all(getattr(st, "_event", None) is not None for st in states):
        from .generate import _blocked_wait      # (what the host blocks here is the GPU's lead over it: Trainer's bound detector)
        for st in states:
            _blocked_wait(st._event)
        host = [st._host.tolist() for st in states]
    else:
        host = torch.stack([st.binning[:32].view(torch.int32) for st in states]).tolist()
    over = False
    for st, c in zip(states, host):
        st._counters, st._listed = tuple(int(v) for v in c[:4]), int(c[7])
        key = (st.cs.image_height, st.cs.image_width)
        _capacity_hint[key] = max(_capacity_hint.get(key, 0), int((c[7] or c[0]) * 1.25) + 1024)
        over = over or bool(c[1])
    return [c[0] for c in host], over


def backward_scratch_floats(P: int, max_instances: int) -> int:
    """Floats of scratch gsvc_raster_backward needs (one 64-byte row of partial sums per instance)."""
    return int(_lib.lib().gsvc_raster_backward_scratch_bytes(int(P), int(max_instances))) // 4


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, cs, holder, sync=True):
        means3D, colors = _as_f32(means3D, "means3D"), _as_f32(colors, "colors_precomp")
        opacities, scales, rotations = _as_f32(opacities, "opacities"), _as_f32(scales, "scales"), _as_f32(rotations, "rotations")
        image, radii, state = raster_forward(cs, means3D, colors, opacities, scales, rotations, sync=sync, readback=not sync)
        ctx.state = state
        ctx.save_for_backward(means3D, colors, opacities, scales, rotations)
        ctx.mark_non_differentiable(radii)
        holder["state"] = state
        return image, radii

    @staticmethod
    def backward(ctx, grad_image, _grad_radii):
        means3D, colors, opacities, scales, rotations = ctx.saved_tensors
        st = ctx.state
        P = st.P
        dev = means3D.device
        g = _as_f32(grad_image, "grad_image")
        d3 = torch.empty(P, 3, device=dev)
        d2 = torch.empty(P, 3, device=dev)
        dc = torch.empty(P, 3, device=dev)
        do = torch.empty(P, 1, device=dev)
        ds = torch.empty(P, 3, device=dev)
        dq = torch.empty(P, 4, device=dev)
        scratch = torch.empty(backward_scratch_floats(P, st.max_instances), device=dev)
        _lib.check(_lib.lib().gsvc_raster_backward(
            C.byref(st.cs), P, st.max_instances, _lib.ptr(means3D), _lib.ptr(colors), _lib.ptr(opacities),
            _lib.ptr(scales), _lib.ptr(rotations), _lib.ptr(st.radii), _lib.ptr(st.geom), _lib.ptr(st.binning),
            _lib.ptr(st.image_state), _lib.ptr(g), _lib.ptr(d3), _lib.ptr(d2), _lib.ptr(dc), _lib.ptr(do), _lib.ptr(ds),
            _lib.ptr(dq), _lib.ptr(scratch), _lib.current_stream(dev)), "gsvc_raster_backward")
        return d3, d2, dc, do, ds, dq, None, None, None


_SIDE = {}


def _side_streams(device, renders):
    """Two side streams per device for the independent renders of a step (GSVC_RASTER_STREAMS=1: everything on the current
    stream)."""
    n = switches.RASTER_STREAMS
    if n <= 1 or renders < 2 or torch.device(device).type != "cuda":
        return []
    key = (torch.device(device).index, n)
    if key not in _SIDE:
        _SIDE[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _SIDE[key]


class _RasterizeMany(torch.autograd.Function):
    """R rasterizations of consecutive row ranges of ONE set of Gaussian tensors (the un-compacted renders of a fitting step are
    slices of the batch the generation pass produced): the forward launches the R pipelines on the ranges in place, the backward
    writes every render's gradients straight into its rows of batch-sized tensors.  As R separate autograd functions the batch
    tensors were split on the way in and their six gradients concatenated on the way back (six cat launches, 84 us per step),
    with a zero-filled means2D leaf and a radii tensor per render."""

    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, cs_list, bounds, holder):
        means3D, colors = _as_f32(means3D, "means3D"), _as_f32(colors, "colors_precomp")
        opacities, scales, rotations = _as_f32(opacities, "opacities"), _as_f32(scales, "scales"), _as_f32(rotations, "rotations")
        N = int(means3D.shape[0])
        radii = torch.empty(N, dtype=torch.int32, device=means3D.device)
        images, states = [], []
        # the renders are independent pipelines of dependent kernels (bin, scatter, sort, composite): dealt to two side streams,
        # one render's kernel boundaries and tails are filled by the other's kernels
        side = _side_streams(means3D.device, len(cs_list))
        main = torch.cuda.current_stream(means3D.device)
        for sd in side:
            sd.wait_stream(main)
        for r, cs in enumerate(cs_list):
            a, b = bounds[r], bounds[r + 1]
            img, _, st = raster_forward(cs, means3D[a:b], colors[a:b], opacities[a:b], scales[a:b], rotations[a:b], sync=False,
                                        readback=True, radii_out=radii[a:b], side_stream=side[r % len(side)] if side else None)
            images.append(img)
            states.append(st)
        for sd in side:
            main.wait_stream(sd)
        ctx.states, ctx.bounds = states, tuple(bounds)
        ctx.save_for_backward(means3D, colors, opacities, scales, rotations)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        holder["states"] = states
        return (*images, radii)

    @staticmethod
    def backward(ctx, *grads):
        means3D, colors, opacities, scales, rotations = ctx.saved_tensors
        N, dev = means3D.shape[0], means3D.device
        d3, d2, dc = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
        do, ds, dq = torch.empty(N, 1, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, 4, device=dev)
        L = _lib.lib()
        side = _side_streams(dev, len(ctx.states))
        main = torch.cuda.current_stream(dev)
        # everything the side streams read is produced / allocated on the main stream BEFORE the fork: an image gradient that is
        # not contiguous fp32 (an expanded gradient of images[r].sum(), autocast) is converted by a copy kernel on main, and the
        # scratch rows are allocated there — converting after the fork left the side stream's backward reading the copy's
        # destination with nothing ordering the two
        gs = [None if g is None else _as_f32(g, "grad_image") for g in grads[:len(ctx.states)]]
        scratches = [None if g is None else torch.empty(backward_scratch_floats(st.P, st.max_instances), device=dev)
                     for g, st in zip(gs, ctx.states)]              # alive until the join below
        for r, g in enumerate(gs):
            if g is None:           # a render nothing was computed from: its Gaussians get no gradient
                a, b = ctx.bounds[r], ctx.bounds[r + 1]
                for t in (d3, d2, dc, do, ds, dq):
                    t[a:b].zero_()
        for sd in side:
            sd.wait_stream(main)
        for r, st in enumerate(ctx.states):
            a, b = ctx.bounds[r], ctx.bounds[r + 1]
            g, scratch = gs[r], scratches[r]
            if g is None:
                continue
            stream = C.c_void_p(side[r % len(side)].cuda_stream) if side else _lib.current_stream(dev)
            P = st.P
            _lib.check(L.gsvc_raster_backward(
                C.byref(st.cs), P, st.max_instances, _lib.ptr(means3D[a:b]), _lib.ptr(colors[a:b]), _lib.ptr(opacities[a:b]),
                _lib.ptr(scales[a:b]), _lib.ptr(rotations[a:b]), _lib.ptr(st.radii), _lib.ptr(st.geom), _lib.ptr(st.binning),
                _lib.ptr(st.image_state), _lib.ptr(g), _lib.ptr(d3[a:b]), _lib.ptr(d2[a:b]), _lib.ptr(dc[a:b]), _lib.ptr(do[a:b]),
                _lib.ptr(ds[a:b]), _lib.ptr(dq[a:b]), _lib.ptr(scratch), stream), "gsvc_raster_backward")
        for sd in side:
            main.wait_stream(sd)
        del scratches, gs
        return d3, d2, dc, do, ds, dq, None, None, None


def rasterize_many(cs_list, bounds, means3D, means2D, colors, opacities, scales, rotations):
    """(images [R], radii [N] int32, states [R]) of R un-synchronised rasterizations over the row ranges ``bounds`` (R + 1
    offsets) of the given tensors; ``means2D`` is the [N, 3] leaf whose .grad receives the screen-space gradients.  The caller
    resolves the instance counters with resolve_deferred(states)."""
    holder = {}
    out = _RasterizeMany.apply(means3D, means2D, colors, opacities, scales, rotations, list(cs_list), list(bounds), holder)
    return list(out[:-1]), out[-1], holder["states"]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings
        self._cs = None
        self.last_state: Optional[RasterState] = None
        self.deferred = False   # True: forward() does not read the counters back (third return value = RasterState)

    def _c_settings(self):
        if self._cs is None:
            self._cs = settings_to_c(self.raster_settings)
        return self._cs

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        if cov3D_precomp is not None:
            raise NotImplementedError("cov3D_precomp is not used by GSVC (pipe.compute_cov3D_python is False)")
        if scales is None or rotations is None:
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        with torch.no_grad():
            m, s, q = _as_f32(means3D, "means3D"), _as_f32(scales, "scales"), _as_f32(rotations, "rotations")
            P = int(m.shape[0])
            radii = torch.empty(P, dtype=torch.int32, device=m.device)
            _lib.check(_lib.lib().gsvc_raster_visible_filter(C.byref(self._c_settings()), P, _lib.ptr(m), _lib.ptr(s),
                                                             _lib.ptr(q), _lib.ptr(radii), _lib.current_stream(m.device)),
                       "gsvc_raster_visible_filter")
        return radii

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if shs is not None:
            raise NotImplementedError("SH colours are not used by GSVC (sh_degree=0, colors_precomp always given)")
        if cov3D_precomp is not None:
            raise NotImplementedError("cov3D_precomp is not used by GSVC")
        if scales is None or rotations is None:
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        holder = {}
        image, radii = _RasterizeGaussians.apply(means3D, means2D, colors_precomp, opacities, scales, rotations,
                                                 self._c_settings(), holder, not self.deferred)
        self.last_state = holder["state"]
        if self.deferred:      # counters stay on the device: the caller resolves them with resolve_deferred()
            return image, radii, self.last_state
        num_rendered = self.last_state.counters()[0]
        return image, radii, num_rendered
