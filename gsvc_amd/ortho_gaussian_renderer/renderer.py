"""``render`` — prefilter anchors, generate the frame's neural Gaussians, rasterize, pack RenderResults.

Same signature and RenderResults fields as reference ortho_gaussian_renderer/renderer.py:14-119.
"""
from __future__ import annotations

import os

import torch

from ..common.base import RenderResults
from ..generate import (GenerateMode, finish_deferred_rate, generate_neural_gaussians, generate_neural_gaussians_many, generator_trunks,
                        region)
from ..rasterizer import GaussianRasterizer, raster_forward, rasterize_many, settings_to_c
from .preprocess import prefilter_geometry, prefilter_voxel, prefilter_voxels_many, raster_settings_for


def render(frame, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, retain_grad=False,
           mode=GenerateMode.TRAINING_FULL_PRECISION):
    visible_mask = prefilter_voxel(frame, pc, pipe, bg_color)
    gss = generate_neural_gaussians(frame, pc, visible_mask, mode)
    # zero tensor whose .grad receives the screen-space gradient (densification statistics read it)
    screenspace_points = torch.zeros_like(gss.xyz, dtype=pc.get_anchor.dtype, requires_grad=True) + 0
    if retain_grad:
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    rasterizer = GaussianRasterizer(raster_settings=raster_settings_for(frame, pc, pipe, bg_color, scaling_modifier))
    rendered_image, radii, num_rendered = rasterizer(
        means3D=gss.xyz, means2D=screenspace_points, shs=None, colors_precomp=gss.color, opacities=gss.opacity,
        scales=gss.scaling, rotations=gss.rot, cov3D_precomp=None)
    return RenderResults(
        rendered_image=rendered_image, viewspace_points=screenspace_points, visibility_filter=radii > 0,
        visible_mask=visible_mask, radii=radii, active_gaussains=(radii > 0).sum(), num_rendered=num_rendered,
        selection_mask=gss.mask, neural_opacity=gss.neural_opacity, scaling=gss.scaling,
        bit_per_param=gss.bit_per_param, bit_per_feat_param=gss.bit_per_feat_param,
        bit_per_scaling_param=gss.bit_per_scaling_param, bit_per_offsets_param=gss.bit_per_offsets_param,
        entropy_constrained=(gss.bit_per_param is not None), generated_gaussians=gss, time_sub=gss.time_sub)


def plan_views(frames, pc, pipe, bg_color: torch.Tensor, mode=GenerateMode.TRAINING_FULL_PRECISION):
    """Visibility test of ``frames`` + every data-dependent index list of their batched generation pass, queued without a host
    synchronisation (gsvc_amd.generate.StepPlan); hand the result to ``render_many(plan=...)``.  Built at the tail of a
    fitting step for the next one, it replaces the step's six count read-backs by one wait."""
    from ..generate import StepPlan
    geometry = prefilter_geometry(pc)
    visible = prefilter_voxels_many(frames, pc, pipe, bg_color, geometry=geometry)
    return StepPlan(frames, pc, visible, geometry, sample=mode in (GenerateMode.TRAINING_ENTROPY, GenerateMode.TRAININ_STE_ENTROPY))


def render_many(frames, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, retain_grad=False,
                mode=GenerateMode.TRAINING_FULL_PRECISION, dense=False, anchor_grad=True, plan=None):
    """`render` for several frames/views of one step: the anchor -> Gaussian generation of all of them runs as
    one batch (gsvc_amd.generate.generate_neural_gaussians_many), then each view is rasterized.  Returns a list
    of RenderResults with the same fields `render` fills.

    ``dense=True``: no "opacity > 0" compaction and no read-back of the rasterizer's counters — the per-Gaussian
    fields (viewspace_points, radii, visibility_filter, scaling) cover all K slots of every visible anchor,
    ``selection_mask`` marks the slots with opacity > 0 (the others get radius 0), ``num_rendered`` is None until
    ``gsvc_amd.rasterizer.resolve_deferred([r.raster_state ...])`` is called, which the caller MUST do (it is the
    overflow check) before trusting the images.

    ``anchor_grad=False``: the anchor positions enter the generation pass detached (no gradient w.r.t. ``_anchor``:
    GSVC trains them with learning rate 0, `arguments/__init__.py` position_lr_*), which removes the backward
    through the positional embedding, the hash-grid input gradient and dy_dx."""
    if plan is not None:
        geometry, visible = plan.geometry, plan.visible_masks
    else:
        geometry = prefilter_geometry(pc)
        visible = prefilter_voxels_many(frames, pc, pipe, bg_color, geometry=geometry)
    with region('render.generate'):
        gss_list = generate_neural_gaussians_many(frames, pc, visible, mode, dense=dense,
                                                  anchors=None if anchor_grad else geometry[0], plan=plan, defer_rate=dense)
    results = []
    batch = getattr(gss_list[0], "batch", None) if (dense and gss_list) else None
    if batch is not None and getattr(batch, "xyz", None) is not None and len(batch.seg_offsets) == len(frames) + 1:
        # the R renders are row ranges of one set of tensors: one autograd function rasterizes them all (no split of the inputs,
        # no concatenation of their gradients, one means2D leaf, one radii tensor)
        bounds = batch.seg_offsets
        leaf = torch.empty(bounds[-1], 3, dtype=pc._anchor.dtype, device=batch.xyz.device, requires_grad=True)   # value never read
        cs_list = [settings_to_c(raster_settings_for(f, pc, pipe, bg_color, scaling_modifier)) for f in frames]
        if batch.xyz.is_cuda and getattr(batch, "small_work", False):
            # everything the generation pass produced is queued by now: work that reads only that (the sampled rate, the loss's
            # regularisers) may start behind this event on another stream while the rasterizer runs
            batch.generated_event = torch.cuda.current_stream(batch.xyz.device).record_event()
        with region('render.rasterize_many'):
            images, radii_all, states = rasterize_many(cs_list, bounds, batch.xyz, leaf, batch.color, batch.neural_opacity, batch.scaling,
                                                       batch.rot)
        finish_deferred_rate(gss_list)       # the sampled rate, behind the rasterizer's launches on its own stream (generate.py)
        seen_all = radii_all > 0
        batch.viewspace, batch.seen = leaf, seen_all
        for r, (frame, visible_mask, gss) in enumerate(zip(frames, visible, gss_list)):
            a, b = bounds[r], bounds[r + 1]
            results.append(RenderResults(
                rendered_image=images[r], viewspace_points=_GradRows(leaf, a, b), visibility_filter=seen_all[a:b],
                visible_mask=visible_mask, radii=radii_all[a:b], active_gaussains=states[r].binning[8:12].view(torch.int32)[0],
                num_rendered=None, selection_mask=gss.mask, neural_opacity=gss.neural_opacity, scaling=gss.scaling,
                bit_per_param=gss.bit_per_param, bit_per_feat_param=gss.bit_per_feat_param,
                bit_per_scaling_param=gss.bit_per_scaling_param, bit_per_offsets_param=gss.bit_per_offsets_param,
                entropy_constrained=(gss.bit_per_param is not None), generated_gaussians=gss, time_sub=gss.time_sub,
                dense=dense, visible_index=gss.visible_index, raster_state=states[r]))
        return results
    finish_deferred_rate(gss_list)
    for frame, visible_mask, gss in zip(frames, visible, gss_list):
        # the tensor whose .grad receives the screen-space gradient: a leaf (the reference's ``zeros_like(...) + 0`` with
        # retain_grad() holds the same numbers through two more kernels); pc._anchor.dtype, not pc.get_anchor.dtype — the
        # property re-quantises all anchors on every access
        screenspace_points = torch.zeros(gss.xyz.shape, dtype=pc._anchor.dtype, device=gss.xyz.device, requires_grad=True)
        rasterizer = GaussianRasterizer(raster_settings=raster_settings_for(frame, pc, pipe, bg_color, scaling_modifier))
        rasterizer.deferred = dense
        rendered_image, radii, num_rendered = rasterizer(
            means3D=gss.xyz, means2D=screenspace_points, shs=None, colors_precomp=gss.color, opacities=gss.opacity,
            scales=gss.scaling, rotations=gss.rot, cov3D_precomp=None)
        seen = radii > 0
        # un-compacted renders: the rasterizer already counted the Gaussians with radius > 0 (gsvc_raster_counters.num_visible)
        active = num_rendered.binning[8:12].view(torch.int32)[0] if dense else seen.sum()
        results.append(RenderResults(
            rendered_image=rendered_image, viewspace_points=screenspace_points, visibility_filter=seen,
            visible_mask=visible_mask, radii=radii, active_gaussains=active,
            num_rendered=None if dense else num_rendered,
            selection_mask=gss.mask, neural_opacity=gss.neural_opacity, scaling=gss.scaling,
            bit_per_param=gss.bit_per_param, bit_per_feat_param=gss.bit_per_feat_param,
            bit_per_scaling_param=gss.bit_per_scaling_param, bit_per_offsets_param=gss.bit_per_offsets_param,
            entropy_constrained=(gss.bit_per_param is not None), generated_gaussians=gss, time_sub=gss.time_sub,
            dense=dense, visible_index=gss.visible_index, raster_state=num_rendered if dense else None))
    return results


class _GradRows:
    """``viewspace_points`` of one render of a batched step: its ``.grad`` is the render's rows of the batch leaf's gradient
    (reference scene/gaussian_model.py:1311 reads ``viewspace_points.grad[:, :2]``)."""

    def __init__(self, leaf, a, b):
        self.leaf, self.a, self.b = leaf, a, b

    @property
    def grad(self):
        return None if self.leaf.grad is None else self.leaf.grad[self.a:self.b]

    @property
    def shape(self):
        return torch.Size((self.b - self.a, 3))


def render_pair(frame, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, mode=GenerateMode.DECODING_AS_IS):
    """The frame GSVC's evaluation / decoder outputs, (render(view) + flip_W(render(view_s))) / 2 (reference
    utils/report_utils.py:297-319), from ONE generation and ONE rasterization pass (gsvc_raster_forward_pair).
    Inference only and only for the deterministic modes (the noise modes draw different Gaussians per view)."""
    if mode not in (GenerateMode.DECODING_AS_IS, GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.TRAININ_STE_ENTROPY):
        raise ValueError("render_pair needs a deterministic GenerateMode")
    with torch.no_grad():
        cs = settings_to_c(raster_settings_for(frame, pc, pipe, bg_color, scaling_modifier))
        pairable = int(frame.image_width) % 16 == 0 and not (cs.flags & 3)
        if pairable and pc._anchor.is_cuda and not os.environ.get("GSVC_PAIR_PER_LAYER"):
            # The batched generation pass on ONE frame (round 6): the whole-network chain kernels (8 launches) instead of the 26 layer
            # launches + ~60 tensor operations of the per-frame path — a per-frame loop as the reference writes it (utils/report_utils.py:
            # 297-319) was HOST-bound at 1.5 ms per frame against 0.5 ms of kernels —, no "opacity > 0" compaction (the rasterizer culls
            # those itself): the per-Gaussian fields of the result cover all K slots of every visible anchor, selection_mask marks the
            # live ones.  Same image (tests/test_train_gpu.py::test_render_frames_equals_render_pair).
            geometry = prefilter_geometry(pc)
            visible_mask = prefilter_voxels_many([frame], pc, pipe, bg_color, scaling_modifier, geometry=geometry)[0]
            vis = visible_mask.nonzero(as_tuple=False).squeeze(1)          # (the one read-back of the frame besides the instance counters)
            if vis.shape[0] > 0:
                gss = generate_neural_gaussians_many([frame], pc, [vis], mode, dense=True, anchors=geometry[0])[0]
                gss.visable_mask = visible_mask
                image, radii, state = raster_forward(cs, gss.xyz.contiguous(), gss.color.contiguous(), gss.opacity.contiguous(),
                                                     gss.scaling.contiguous(), gss.rot.contiguous(), pair=True)
                return RenderResults(
                    rendered_image=image, viewspace_points=None, visibility_filter=radii > 0, visible_mask=visible_mask, radii=radii,
                    active_gaussains=(radii > 0).sum(), num_rendered=state.counters()[0], selection_mask=gss.mask,
                    neural_opacity=gss.neural_opacity, scaling=gss.scaling, generated_gaussians=gss, time_sub=gss.time_sub)
        visible_mask = prefilter_voxel(frame, pc, pipe, bg_color)
        gss = generate_neural_gaussians(frame, pc, visible_mask, mode)
        args = (gss.xyz.contiguous(), gss.color.contiguous(), gss.opacity.contiguous(), gss.scaling.contiguous(), gss.rot.contiguous())
        if int(frame.image_width) % 16 == 0 and not (cs.flags & 3):   # (one-sided slab / pixel-corner: the views do not mirror)
            image, radii, state = raster_forward(cs, *args, pair=True)
        else:   # the two views' tile grids do not mirror: two passes
            import copy
            back = copy.copy(frame)
            back.view_matrix, back.view_matrix_s = frame.view_matrix_s, frame.view_matrix
            f, radii, state = raster_forward(cs, *args)
            b, _, _ = raster_forward(settings_to_c(raster_settings_for(back, pc, pipe, bg_color, scaling_modifier)), *args)
            image = 0.5 * (f + torch.flip(b, dims=(-1,)))
    return RenderResults(
        rendered_image=image, viewspace_points=None, visibility_filter=radii > 0, visible_mask=visible_mask, radii=radii,
        active_gaussains=(radii > 0).sum(), num_rendered=state.counters()[0], selection_mask=gss.mask,
        neural_opacity=gss.neural_opacity, scaling=gss.scaling, generated_gaussians=gss, time_sub=gss.time_sub)


def render_frames(frames, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, mode=GenerateMode.DECODING_AS_IS, batch: int = 8):
    """The decoder's render loop (reference utils/report_utils.py:297-319 / stream decoding: one two-view frame per video
    frame) over a list of frames: the anchor -> Gaussian generation of `batch` frames runs as one un-compacted batch
    (frames of a video are independent), each frame is then composited by one gsvc_raster_forward_pair pass, and the
    instance counters of a batch are read back together.  Yields the images [3, H, W] in order.  Inference only."""
    if mode not in (GenerateMode.DECODING_AS_IS, GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.TRAININ_STE_ENTROPY):
        raise ValueError("render_frames needs a deterministic GenerateMode")
    from ..rasterizer import resolve_deferred
    frames = list(frames)
    for f in frames:
        if int(f.image_width) % 16 != 0 or (int(getattr(pipe, "raster_flags", 0) or 0) & 3):
            raise ValueError("render_frames: the two-view pass needs an image width that is a multiple of 16 and the symmetric-slab / "
                             "pixel-centre conventions (use render_pair)")
    from ..rasterizer import _side_streams
    dev = pc._anchor.device

    def launch(chunk, trunks):
        """Generation of the chunk's frames on the current stream, their two-view passes on the side streams; nothing is waited for."""
        geometry = prefilter_geometry(pc)
        visible = prefilter_voxels_many(chunk, pc, pipe, bg_color, geometry=geometry)
        gss_list = generate_neural_gaussians_many(chunk, pc, visible, mode, dense=True, anchors=geometry[0], trunks=trunks)
        images, states = [], []
        # the frames of a batch are independent pipelines: dealt to two side streams (rasterizer._side_streams), one frame's
        # kernel boundaries and tails are filled by the next frame's kernels
        side = _side_streams(dev, len(chunk))
        main = torch.cuda.current_stream(dev) if side else None
        args = [(gss.xyz.contiguous(), gss.color.contiguous(), gss.opacity.contiguous(), gss.scaling.contiguous(), gss.rot.contiguous())
                for gss in gss_list]
        for sd in side:
            sd.wait_stream(main)
        for k, (f, a) in enumerate(zip(chunk, args)):
            cs = settings_to_c(raster_settings_for(f, pc, pipe, bg_color, scaling_modifier))
            image, _, state = raster_forward(cs, *a, pair=True, sync=False, readback=True, side_stream=side[k % len(side)] if side else None)
            images.append(image)
            states.append(state)
        done = [sd.record_event() for sd in side]
        return chunk, images, states, done, args          # (args: the side streams read them; alive until the batch is finished)

    def finish(job, trunks):
        """The batch's images, once its passes have run and its instance counters say nothing overflowed (else: once more, alone)."""
        while True:
            chunk, images, states, done, _args = job
            for ev in done:
                torch.cuda.current_stream(dev).wait_event(ev)      # whoever reads the images does so on the current stream
            if not resolve_deferred(states)[1]:
                return images
            job = launch(chunk, trunks)      # an instance buffer overflowed, the capacity hint is raised: once more

    with torch.no_grad():
        # the generators' feature-only half does not depend on the frame: once per call for all anchors
        trunks = generator_trunks(pc) if mode in (GenerateMode.DECODING_AS_IS, GenerateMode.TRAINING_FULL_PRECISION) else None
        # software pipeline over the batches: batch i + 1's generation (MFMA / HBM: the current stream) is queued before batch i's
        # counters are waited for, so it runs next to batch i's compositing (vector ALU: the side streams)
        pending = None
        for i in range(0, len(frames), batch):
            job = launch(frames[i:i + batch], trunks)
            if pending is not None:
                yield from finish(pending, trunks)
            pending = job
        if pending is not None:
            yield from finish(pending, trunks)
