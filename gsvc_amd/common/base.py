"""RenderResults — field names equal reference common/base.py:9-27 (``active_gaussains`` (sic) is API)."""
import typing
from dataclasses import dataclass

import torch


@dataclass
class RenderResults:
    rendered_image: torch.Tensor
    viewspace_points: torch.Tensor
    visible_mask: torch.Tensor
    visibility_filter: torch.Tensor
    radii: torch.Tensor
    active_gaussains: int
    num_rendered: int
    time_sub: typing.Union[torch.Tensor, None] = None
    selection_mask: typing.Union[torch.Tensor, None] = None
    neural_opacity: typing.Union[torch.Tensor, None] = None
    scaling: typing.Union[torch.Tensor, None] = None
    bit_per_param: typing.Union[torch.Tensor, None] = None
    bit_per_feat_param: typing.Union[torch.Tensor, None] = None
    bit_per_scaling_param: typing.Union[torch.Tensor, None] = None
    bit_per_offsets_param: typing.Union[torch.Tensor, None] = None
    generated_gaussians: typing.Any = None
    entropy_constrained: bool = False
    # extras of the un-compacted ("dense") batched path (render_many(dense=True)); unset otherwise
    dense: bool = False                      # Gaussians = all K slots of every visible anchor; selection_mask marks opacity > 0
    visible_index: typing.Union[torch.Tensor, None] = None    # int64 indices of the visible anchors
    raster_state: typing.Any = None          # rasterizer state whose instance counters have not been read back yet
