"""Anchor -> neural Gaussians for one frame (host orchestration around the MLPs, the hash grid and the rate
kernel).

Same interface and values as reference ortho_gaussian_renderer/guassian.py: ``GenerateMode`` (:21-26),
``RatePack`` (:34-39), ``GeneratedGaussians`` (:42-56), ``calc_sampled_rate`` (:73-132) and
``generate_neural_gaussians`` (:134-310).  The random draws (quantisation noise for feat / scaling / offsets,
then the 5 % anchor sample) happen in the reference's order so a seeded run reproduces its fixtures.

What is organised differently: the visible-anchor subset is taken once with an index gather, the
"opacity > 0" compaction uses one index list for all per-Gaussian tensors, and getters that the reference
re-evaluates over all anchors several times per call are evaluated once.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from enum import Enum

import torch

from .encodings import STE_multistep


class GenerateMode(Enum):
    TRAINING_FULL_PRECISION = 0
    TRAINING_QUANTIZED = 1
    TRAINING_ENTROPY = 2
    TRAININ_STE_ENTROPY = 3  # (sic) the reference's spelling is API
    DECODING_AS_IS = 4


@dataclass
class RatePack:
    bit_per_param: torch.Tensor = None
    bit_per_feat_param: torch.Tensor = None
    bit_per_scaling_param: torch.Tensor = None
    bit_per_offsets_param: torch.Tensor = None


@dataclass
class GeneratedGaussians:
    xyz: torch.Tensor
    color: torch.Tensor
    opacity: torch.Tensor
    scaling: torch.Tensor
    rot: torch.Tensor
    neural_opacity: torch.Tensor = None
    visable_mask: torch.Tensor = None  # (sic)
    mask: torch.Tensor = None
    bit_per_param: torch.Tensor = None
    bit_per_feat_param: torch.Tensor = None
    bit_per_scaling_param: torch.Tensor = None
    bit_per_offsets_param: torch.Tensor = None
    concatenated_all: torch.Tensor = None
    time_sub: float = None


BASE_Q_FEAT, BASE_Q_SCALING, BASE_Q_OFFSETS = 1, 0.001, 0.2
SAMPLE_RATE = 0.05


def calc_sampled_rate(pc, visible_mask, feat, grid_scaling, grid_offsets, Q_feat, Q_scaling, Q_offsets, entropy_context):
    """Bits per parameter estimated on a 5 % Bernoulli sample of the visible anchors that still own at least
    one live offset, scaled by the fraction of such anchors."""
    K = pc.n_offsets
    vis = _as_index(visible_mask)
    offset_masks = _visible_mask(pc, vis)
    with torch.no_grad():
        mask_anchor = (torch.sum(offset_masks, dim=1)[:, 0]) > 0     # get_mask_anchor restricted to the visible anchors
    keep_rate = (mask_anchor.sum() / mask_anchor.numel()).detach()
    # same draw as the reference's rand_like(anchor[:, 0]) (one uniform per visible anchor)
    chosen = (torch.rand_like(feat[:, 0]) <= SAMPLE_RATE) & mask_anchor.to(torch.bool)
    sel = chosen.nonzero(as_tuple=False).squeeze(1)
    take = lambda t: t.index_select(0, sel)  # noqa: E731
    ec = entropy_context
    bit_feat = pc.entropy_gaussian(take(feat), take(ec.mean_feat), take(ec.scale_feat), take(Q_feat), pc._anchor_feat.mean())
    bit_scaling = pc.entropy_gaussian(take(grid_scaling), take(ec.mean_scaling), take(ec.scale_scaling), take(Q_scaling),
                                      pc.get_scaling.mean())
    bit_offsets = pc.entropy_gaussian(take(grid_offsets).view(-1, 3 * K), take(ec.mean_offsets), take(ec.scale_offsets),
                                      take(Q_offsets), pc._offset.mean())
    bit_offsets = bit_offsets * take(offset_masks).repeat(1, 1, 3).view(-1, 3 * K)
    sf, ss, so = bit_feat.sum(), bit_scaling.sum(), bit_offsets.sum()
    nf, ns, no = bit_feat.numel(), bit_scaling.numel(), bit_offsets.numel()
    return RatePack(bit_per_param=(sf + ss + so) / (nf + ns + no) * keep_rate,
                    bit_per_feat_param=sf / nf * keep_rate,
                    bit_per_scaling_param=ss / ns * keep_rate,
                    bit_per_offsets_param=so / no * keep_rate)


def _visible_mask(pc, vis):
    """pc.get_mask[vis] evaluated on the gathered rows only (same values: the activation is elementwise)."""
    raw = pc._mask.index_select(0, vis)
    if pc.decoded_version:
        return raw
    s = torch.sigmoid(raw)
    return ((s > 0.01).float() - s).detach() + s


def _visible_scaling(pc, vis):
    raw = pc._scaling.index_select(0, vis)
    return raw if pc.decoded_version else 1.0 * pc.scaling_activation(raw)


def _as_index(mask_or_index):
    """Boolean mask -> int64 index list (one nonzero); index tensors pass through.  Gathers by index have a
    scatter-add backward (atomics) instead of the sort-based index_put a boolean mask triggers."""
    if mask_or_index.dtype == torch.bool:
        return mask_or_index.nonzero(as_tuple=False).squeeze(1)
    return mask_or_index


def _sync(t):
    if t.is_cuda:
        torch.cuda.synchronize()


def generate_neural_gaussians(frame, pc, visible_mask=None, mode=GenerateMode.TRAINING_FULL_PRECISION):
    time_sub = 0
    all_anchor = pc.get_anchor
    if visible_mask is None:
        visible_mask = torch.ones(all_anchor.shape[0], dtype=torch.bool, device=all_anchor.device)
    K = pc.n_offsets
    vis = _as_index(visible_mask)            # one nonzero for the five per-anchor gathers
    anchor = all_anchor.index_select(0, vis)
    feat = pc._anchor_feat.index_select(0, vis)
    grid_offsets = pc._offset.index_select(0, vis)
    grid_scaling = _visible_scaling(pc, vis)
    offset_masks = _visible_mask(pc, vis)
    rate = RatePack()
    Q_feat, Q_scaling, Q_offsets = BASE_Q_FEAT, BASE_Q_SCALING, BASE_Q_OFFSETS

    if mode in (GenerateMode.TRAINING_FULL_PRECISION, GenerateMode.DECODING_AS_IS):
        pass
    elif mode == GenerateMode.TRAINING_QUANTIZED:
        feat = pc.noise_quantizer(feat, Q_feat)
        grid_scaling = pc.noise_quantizer(grid_scaling, Q_scaling)
        grid_offsets = pc.noise_quantizer(grid_offsets, Q_offsets)
    elif mode == GenerateMode.TRAINING_ENTROPY:
        ec = pc.calc_entropy_context(anchor)
        Q_feat, Q_scaling, Q_offsets = Q_feat * ec.Q_feat_adj, Q_scaling * ec.Q_scaling_adj, Q_offsets * ec.Q_offsets_adj
        feat = pc.noise_quantizer(feat, Q_feat)
        grid_scaling = pc.noise_quantizer(grid_scaling, Q_scaling)
        grid_offsets = pc.noise_quantizer(grid_offsets, Q_offsets.unsqueeze(1))
        rate = calc_sampled_rate(pc, vis, feat, grid_scaling, grid_offsets, Q_feat, Q_scaling, Q_offsets, ec)
    elif mode == GenerateMode.TRAININ_STE_ENTROPY:
        _sync(anchor)
        t1 = time.time()
        ec = pc.calc_entropy_context(anchor)
        Q_feat = Q_feat * ec.Q_feat_adj.detach()
        Q_scaling = Q_scaling * ec.Q_scaling_adj.detach()
        Q_offsets = Q_offsets * ec.Q_offsets_adj.detach()
        feat = STE_multistep.apply(feat, Q_feat, pc._anchor_feat.mean()).detach()
        grid_scaling = STE_multistep.apply(grid_scaling, Q_scaling, pc.get_scaling.mean()).detach()
        grid_offsets = STE_multistep.apply(grid_offsets, Q_offsets.unsqueeze(1), pc._offset.mean()).detach()
        _sync(anchor)
        time_sub = time.time() - t1
        rate = calc_sampled_rate(pc, vis, feat, grid_scaling, grid_offsets, Q_feat, Q_scaling, Q_offsets, ec)
    else:
        raise ValueError(f"Unknown mode {mode}")

    # conditioning: embedding of the frame's z (same for every anchor) and of the anchor's z offset to it
    cam = frame.cam_pos.to(anchor.device)
    ob_view = (anchor - cam)[:, 2:]
    time_emb = pc.embed_time_fn(torch.zeros_like(ob_view) + cam[-1])
    z_emb = pc.embed_fn(ob_view)
    pe = torch.cat([time_emb, z_emb], dim=1)

    V = anchor.shape[0]
    neural_opacity = pc.get_opacity_mlp(feat, pe).reshape(-1, 1) * offset_masks.view(-1, 1)
    mask = (neural_opacity > 0.0).view(-1)
    color = pc.get_color_mlp(feat, pe).reshape(V * K, 3)
    scale_rot = pc.get_cov_mlp(feat, pe).reshape(V * K, 7)
    neural_offset = pc.get_deform_mlp(torch.cat([feat, pe], dim=1)).reshape(V * K, 3)
    offsets = grid_offsets.view(-1, 3) + neural_offset

    # per-Gaussian table [scaling(6) | anchor(3) | colour(3) | scale_rot(7) | offset(3)], kept because the
    # optical-flow loss of the training step reads it (reference utils/loss_utils.py:108-118)
    per_anchor = torch.cat([grid_scaling, anchor], dim=-1)
    concatenated_all = torch.cat([per_anchor.repeat_interleave(K, dim=0), color, scale_rot, offsets], dim=-1)
    alive_idx = mask.nonzero(as_tuple=False).squeeze(1)
    alive = concatenated_all.index_select(0, alive_idx)
    scaling_rep, anchor_rep = alive[:, 0:6], alive[:, 6:9]
    color_a, scale_rot_a, offsets_a = alive[:, 9:12], alive[:, 12:19], alive[:, 19:22]

    scaling = scaling_rep[:, 3:] * torch.sigmoid(scale_rot_a[:, :3])
    rot = pc.rotation_activation(scale_rot_a[:, 3:7])
    xyz = torch.clamp(anchor_rep + offsets_a * scaling_rep[:, :3], pc.x_bound_min, pc.x_bound_max)
    return GeneratedGaussians(
        xyz=xyz, color=color_a, opacity=neural_opacity.index_select(0, alive_idx),
        scaling=scaling, rot=rot,
        neural_opacity=neural_opacity, visable_mask=visible_mask, mask=mask,
        bit_per_param=rate.bit_per_param, bit_per_feat_param=rate.bit_per_feat_param,
        bit_per_scaling_param=rate.bit_per_scaling_param, bit_per_offsets_param=rate.bit_per_offsets_param,
        concatenated_all=concatenated_all, time_sub=time_sub)
