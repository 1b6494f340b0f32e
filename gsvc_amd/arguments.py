"""Configuration of the hot path: the fields of reference arguments/__init__.py:50-244 that the renderer,
the model and the fitting step read (same names, same defaults), plus the YAML overlay of
cfgs/cfg_20240919.yaml.  The command-line parser (simple_parsing) is out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, fields


@dataclass
class ModelParams:
    sh_degree: int = 0
    threshold: float = 0.1            # render horizon: half-thickness of the z-slab a frame sees
    kernel_size: float = 0.3          # low-pass added to the 2-D covariance (pixels^2)
    anchor_feature_dim: int = 50
    n_offsets: int = 10
    voxel_size: float = 0.001
    update_depth: int = 3
    update_init_factor: int = 16
    update_hierarchy_factor: int = 4
    time_multi_res: int = 16
    offset_multi_res: int = 16
    log2: int = 13
    log2_2D: int = 15
    grid_feature_dim: int = 4
    use_feat_bank: bool = False
    resolution: int = -1
    white_background: bool = False


@dataclass
class PipelineParams:
    source_path: str = ""
    optical_path: str = ""
    model_path: str = ""
    tmc3_executable: str = None
    init_point_cloud: str = ""
    convert_SHs_python: bool = False
    compute_cov3D_python: bool = False
    debug: bool = False
    skip_prefetch: bool = False
    # not in the reference: rasterizer convention switches handed to every render (gsvc_raster_settings.flags / .low_pass,
    # include/gsvc_hip.h; 0 = DESIGN.md's raster spec) — how a maintainer matches the real extension, INTEGRATION.md
    raster_flags: int = 0
    raster_low_pass: float = 0.0


def _lr(init, final, delay_mult=0.01, max_steps=40_000):
    return dict(init=init, final=final, delay_mult=delay_mult, max_steps=max_steps)


@dataclass
class OptimizationParams:
    iterations: int = 40_000
    position_lr_init: float = 0.0
    position_lr_final: float = 0.0
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 40_000
    offset_lr_init: float = 0.01
    offset_lr_final: float = 0.0001
    offset_lr_delay_mult: float = 0.01
    offset_lr_max_steps: int = 40_000
    mask_lr_init: float = 0.01
    mask_lr_final: float = 0.0001
    mask_lr_delay_mult: float = 0.01
    mask_lr_max_steps: int = 40_000
    feature_lr: float = 0.0075
    opacity_lr: float = 0.02
    scaling_lr: float = 0.007
    rotation_lr: float = 0.002
    mlp_opacity_lr_init: float = 0.002
    mlp_opacity_lr_final: float = 0.00002
    mlp_opacity_lr_delay_mult: float = 0.01
    mlp_opacity_lr_max_steps: int = 40_000
    mlp_cov_lr_init: float = 0.004
    mlp_cov_lr_final: float = 0.004
    mlp_cov_lr_delay_mult: float = 0.01
    mlp_cov_lr_max_steps: int = 40_000
    mlp_color_lr_init: float = 0.008
    mlp_color_lr_final: float = 0.00005
    mlp_color_lr_delay_mult: float = 0.01
    mlp_color_lr_max_steps: int = 40_000
    mlp_featurebank_lr_init: float = 0.01
    mlp_featurebank_lr_final: float = 0.00001
    mlp_featurebank_lr_delay_mult: float = 0.01
    mlp_featurebank_lr_max_steps: int = 40_000
    encoding_xyz_lr_init: float = 0.005
    encoding_xyz_lr_final: float = 0.00001
    encoding_xyz_lr_delay_mult: float = 0.33
    encoding_xyz_lr_max_steps: int = 40_000
    mlp_grid_lr_init: float = 0.005
    mlp_grid_lr_final: float = 0.00001
    mlp_grid_lr_delay_mult: float = 0.01
    mlp_grid_lr_max_steps: int = 40_000
    mlp_deform_lr_init: float = 0.005
    mlp_deform_lr_final: float = 0.0005
    mlp_deform_lr_delay_mult: float = 0.01
    mlp_deform_lr_max_steps: int = 40_000
    mlp_entropy_net_lr_init: float = 0.005
    mlp_entropy_net_lr_final: float = 0.0005
    mlp_entropy_net_lr_delay_mult: float = 0.01
    mlp_entropy_net_lr_max_steps: int = 40_000
    init_anchor_num: int = 10_000
    lmbda: float = 0.001
    percent_dense: float = 0.01
    lambda_dssim: float = 0.2
    start_stat: int = 500
    update_from: int = 1500
    update_interval: int = 100
    update_until: int = 25_000
    pause_densification: int = 1_000
    scaling_reg: float = 0.01
    opacity_reg: float = 0
    optical_lambda: float = 5
    full_precision_training_total: int = 10_000
    quantized_training_total: int = 5_000
    entropy_constrained_train_total: int = 20_000
    ste_entropy_constrained_train_total: int = 5_000
    min_opacity: float = 0.005
    success_threshold: float = 0.8
    densify_grad_threshold: float = 0.0005


def apply_overrides(obj, overrides: dict):
    names = {f.name for f in fields(obj)}
    for k, v in (overrides or {}).items():
        if k not in names:
            raise KeyError(f"{type(obj).__name__} has no field {k!r}")
        setattr(obj, k, v)
    return obj


def load_yaml_config(path: str):
    """cfgs/*.yaml layout of the reference: top-level keys `model:` and `optimization:` override the defaults."""
    import yaml
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    return (apply_overrides(ModelParams(), cfg.get("model")), apply_overrides(OptimizationParams(), cfg.get("optimization")),
            PipelineParams())


def cfg_20240919():
    """The reference's shipped configuration (cfgs/cfg_20240919.yaml)."""
    m = apply_overrides(ModelParams(), dict(voxel_size=0.001, update_init_factor=16, update_hierarchy_factor=4,
                                            update_depth=3, grid_feature_dim=8, threshold=0.05))
    o = apply_overrides(OptimizationParams(), dict(iterations=40_000, lmbda=0.004, init_anchor_num=100_000, opacity_reg=0))
    return m, o, PipelineParams()
