"""ctypes binding of libgsvc_hip.so (the C-ABI declared in include/gsvc_hip.h).

There is no fallback: if the library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``
or ``make -C gsvc_amd/csrc``) every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSVC_LIB_PATH") or os.path.join(_HERE, "csrc", "libgsvc_hip.so")   # env: kernel experiments only
_lib = None


class GsvcError(RuntimeError):
    pass


class RasterSettingsC(C.Structure):
    _fields_ = [
        ("image_height", C.c_int32),
        ("image_width", C.c_int32),
        ("x_min", C.c_float),
        ("y_min", C.c_float),
        ("scale", C.c_float),
        ("threshold", C.c_float),
        ("scale_modifier", C.c_float),
        ("bg", C.c_float * 3),
        ("viewmatrix", C.c_float * 16),
        ("flags", C.c_uint32),
        ("low_pass", C.c_float),
    ]


# gsvc_raster_settings.flags (include/gsvc_hip.h)
RASTER_SLAB_ONE_SIDED, RASTER_PIXEL_CORNER, RASTER_DEPTH_DESCENDING = 1, 2, 4
RASTER_MEANS2D_PIXEL_UNITS, RASTER_CLAMP_STOPS_GRADIENT, RASTER_NO_LOW_PASS = 8, 16, 32
RASTER_TIGHT_BINNING = 64      # list a Gaussian only in the tiles its alpha >= 1/255 box touches (same results, shorter lists)


class AdamTensorC(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("bias_correction1", C.c_float), ("bias_correction2", C.c_float)]


class WgradReduceJobC(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p), ("slots", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32)]


class AnchorLevelC(C.Structure):
    _fields_ = [("states", C.c_void_p), ("words", C.c_void_p), ("freq", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64),
                ("n_words", C.c_int64), ("lanes", C.c_int32)]


class WgradPartialJobC(C.Structure):
    _fields_ = [("G", C.c_void_p), ("X", C.c_void_p), ("workspace", C.c_void_p), ("M", C.c_int64), ("workspace_floats", C.c_int64),
                ("want_db", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("slots_used", C.c_int32)]


class AccumJobC(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("K", C.c_int32), ("pad", C.c_int32)]


class SharedInputJobC(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("Y", C.c_void_p), ("Y2", C.c_void_p), ("N", C.c_int32), ("pad", C.c_int32)]


class RateSampleC(C.Structure):
    _fields_ = [("x", C.c_void_p * 3), ("mean", C.c_void_p * 3), ("scale", C.c_void_p * 3), ("Q", C.c_void_p * 3),
                ("mask", C.c_void_p), ("sel", C.c_void_p), ("sel_ctx", C.c_void_p), ("row_bounds", C.POINTER(C.c_int64)),
                ("x_mean", C.c_void_p), ("C", C.c_int32 * 3), ("K", C.c_int32), ("renders", C.c_int32), ("n_sel", C.c_int64)]


class AnsDecodeJobC(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("seg_offsets", C.c_void_p), ("mu", C.c_void_p), ("sigma", C.c_void_p), ("n", C.c_int64),
                ("min_symbol", C.c_int32), ("max_symbol", C.c_int32), ("seg_len", C.c_int32), ("symbols", C.c_void_p),
                ("error_flag", C.c_void_p), ("scratch", C.c_void_p)]


class GridIOC(C.Structure):
    _fields_ = [("feat_level_stride", C.c_int64), ("feat_point_stride", C.c_int64), ("in_stride", C.c_int32), ("in_col", C.c_int32 * 3)]


class GridManyJobC(C.Structure):
    _fields_ = [("embeddings", C.c_void_p), ("features", C.c_void_p), ("grad_embeddings", C.c_void_p), ("offsets", C.c_void_p),
                ("resolutions", C.c_void_p), ("D", C.c_uint32), ("L", C.c_uint32), ("layout", GridIOC)]


class GeneratorNetC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3", "Wg0", "bg0", "Wg1", "bg1", "Wb0", "bb0", "Wb1", "bb1")] + \
               [(n, C.c_int32) for n in ("feat_dim", "cond_dim", "hidden_dim", "out_dim", "out_act")]


class GeneratorGradsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W1", "b1", "W2", "b2", "W3", "b3", "Wg0", "bg0", "Wg1", "bg1", "Wb0", "bb0", "Wb1", "bb1")]


class FilmRowsC(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cond", C.c_void_p), ("row_of", C.c_void_p), ("src_a", C.c_void_p), ("src_b", C.c_void_p)]


class DeformNetC(C.Structure):
    _fields_ = [("W", C.c_void_p * 5), ("b", C.c_void_p * 5)] + [(n, C.c_int32) for n in ("feat_dim", "cond_dim", "hidden_dim", "out_dim")]


class QuantStepNetC(C.Structure):
    _fields_ = [("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p)]


class DeformGradsC(C.Structure):
    _fields_ = [("W", C.c_void_p * 5), ("b", C.c_void_p * 5)]


class RasterSizesC(C.Structure):
    _fields_ = [("geom_bytes", C.c_uint64), ("binning_bytes", C.c_uint64), ("image_bytes", C.c_uint64)]


_vp = C.c_void_p
_i64 = C.c_int64
_u32 = C.c_uint32
_f = C.c_float

_SIGNATURES = {
    "gsvc_last_error": (C.c_char_p, []),
    "gsvc_version": (C.c_char_p, []),
    "gsvc_profile_enable": (C.c_int, [C.c_int]),
    "gsvc_profile_collect": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int]),
    "gsvc_raster_sizes_query": (C.c_int, [C.POINTER(RasterSettingsC), _i64, _i64, C.POINTER(RasterSizesC)]),
    "gsvc_raster_visible_filter": (C.c_int, [C.POINTER(RasterSettingsC), _i64, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_raster_visible_masks": (C.c_int, [_vp, C.c_int32, _i64, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp]),
    "gsvc_raster_forward": (C.c_int, [C.POINTER(RasterSettingsC), _i64, _i64] + [_vp] * 11),
    "gsvc_raster_forward_pair": (C.c_int, [C.POINTER(RasterSettingsC), _i64, _i64] + [_vp] * 11),
    "gsvc_raster_backward_scratch_bytes": (_i64, [_i64, _i64]),
    "gsvc_raster_backward": (C.c_int, [C.POINTER(RasterSettingsC), _i64, _i64] + [_vp] * 18),
    "gsvc_raster_binning_layout": (C.c_int, [C.POINTER(RasterSettingsC), _i64, _i64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gsvc_raster_image_layout": (C.c_int, [C.POINTER(RasterSettingsC), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "gsvc_grid_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "gsvc_grid_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "gsvc_pack_sign_bits": (C.c_int, [_vp, _i64, _vp, _vp]),
    "gsvc_grid_forward_packed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, C.POINTER(GridIOC), _vp]),
    "gsvc_grid_forward_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, C.POINTER(GridIOC), _vp]),
    "gsvc_grid_backward_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, C.POINTER(GridIOC), _vp]),
    "gsvc_grid_forward_many": (C.c_int, [_vp, C.POINTER(GridManyJobC), C.c_int32, _u32, _u32, _vp]),
    "gsvc_grid_backward_many": (C.c_int, [_vp, C.POINTER(GridManyJobC), C.c_int32, _u32, _u32, _vp]),
    "gsvc_rate_forward": (C.c_int, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, C.c_int32, _i64, _i64, _vp, _vp, _vp]),
    "gsvc_ssim_l1_forward": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_ssim_l1_backward": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_ssim_l1_pair_forward": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_ssim_l1_pair_backward": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_ste_binary_count": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "gsvc_rate_normalise_scratch_bytes": (_i64, []),
    "gsvc_rate_normalise_forward": (C.c_int, [_vp, _vp, C.c_int32, _vp, _i64, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_float), _vp, _vp,
                                              _vp, _vp, _vp]),
    "gsvc_rate_normalise_backward": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "gsvc_training_statis": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _i64, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_anchor_rans_decode": (C.c_int, [C.POINTER(AnchorLevelC), C.c_int32, _vp, _vp]),
    "gsvc_octree_popcount": (C.c_int, [_vp, _i64, _vp, _vp]),
    "gsvc_octree_expand": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "gsvc_morton_decode": (C.c_int, [_vp, _i64, _vp, _vp]),
    "gsvc_q_rows_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _i64, C.c_int32, _vp, _vp]),
    "gsvc_q_rows_backward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _i64, _i64, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "gsvc_plan_scans_scratch_bytes": (_i64, [C.c_int32, _i64]),
    "gsvc_plan_scans": (C.c_int, [_vp, _vp, _vp, C.c_int32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_film_row_maps": (C.c_int, [_vp, C.POINTER(C.c_int64), C.c_int32, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_pair_rows_sum": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int32, _vp, _vp]),
    "gsvc_set_deterministic": (C.c_int, [C.c_int]),
    "gsvc_set_wgrad_stream": (C.c_int, [_vp]),
    "gsvc_wgrad_hold": (C.c_int, [C.c_int32]),
    "gsvc_wgrad_flush": (C.c_int, [_vp]),
    "gsvc_segment_rows_sum": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int32, _vp, C.c_int32, _vp]),
    "gsvc_ste_binary_count_many": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), C.c_int32, _vp, _vp]),
    "gsvc_ste_binary_backward_many": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), C.c_int32, _vp, C.c_int32, _vp, _vp]),
    "gsvc_table_bits": (C.c_int, [_vp, C.c_int32, _i64, _vp, _vp]),
    "gsvc_adam_step": (C.c_int, [C.c_int32, C.POINTER(AdamTensorC), C.c_double, C.c_double, C.c_double, _vp]),
    "gsvc_adam_step_guarded": (C.c_int, [C.c_int32, C.POINTER(AdamTensorC), C.c_double, C.c_double, C.c_double, C.POINTER(C.c_void_p),
                                         C.c_int32, _vp]),
    "gsvc_noise_quant_scratch_floats": (_i64, [C.POINTER(C.c_int64), C.c_int32]),
    "gsvc_noise_quant_forward": (C.c_int, [_vp, _vp, C.c_float, _vp, C.POINTER(C.c_int64), C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "gsvc_noise_quant_backward": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp, _vp, C.POINTER(C.c_int64), C.c_int32, C.c_int32, _vp, _vp,
                                            _vp]),
    "gsvc_ste_quant_forward": (C.c_int, [_vp, _vp, C.c_float, _vp, C.POINTER(C.c_int64), C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "gsvc_gen_tail_forward": (C.c_int, [_vp] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_float), _i64, C.c_int32] + [_vp] * 7),
    "gsvc_gen_tail_backward": (C.c_int, [_vp] * 7 + [C.POINTER(C.c_float), C.POINTER(C.c_float), _i64, C.c_int32] + [_vp] * 12),
    "gsvc_gather_rows_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_gather_rows_backward": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [_vp] * 9),
    "gsvc_gather_rows_backward_ranked": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32] + [_vp] * 9),
    "gsvc_plan_masks": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, _i64, _vp, C.c_int32, C.c_int32, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    "gsvc_compact_by_scan": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "gsvc_param_means_scratch_floats": (C.c_int64, []),
    "gsvc_param_means": (C.c_int, [_vp, _i64, _vp, _i64, C.c_int32, _vp, _i64, _vp, _vp, _vp]),
    "gsvc_ctx_post_forward": (C.c_int, [_vp, _vp, _i64, C.c_int32, _vp, _vp, _vp, _vp]),
    "gsvc_ctx_post_backward": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_film_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "gsvc_film_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gsvc_embed_pe": (C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_int32, C.c_int32, _vp, _vp]),
    "gsvc_optical_forward": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, C.c_int32, _i64, _vp, C.c_int32, C.c_int32,
                                       C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_optical_many_partial_floats": (_i64, [C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_int32]),
    "gsvc_optical_many_forward": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                            C.c_int32, _i64, _vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                            _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_optical_many_backward": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
                                             C.c_int32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_optical_backward": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_regs_partial_floats": (_i64, [C.POINTER(C.c_int64), C.c_int32]),
    "gsvc_regs_forward": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int64), C.c_int32, _vp, _vp, _vp, _vp]),
    "gsvc_regs_backward": (C.c_int, [_vp, _vp, C.POINTER(C.c_int64), C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_knn3_mean_dist2": (C.c_int, [_vp, _vp, C.POINTER(C.c_float), C.c_float, C.c_int32, C.c_int32, C.c_int32, _i64, _vp, _vp]),
    "gsvc_ans_segments": (_i64, [_i64, C.c_int32]),
    "gsvc_ans_table_checksum": (C.c_uint32, []),
    "gsvc_ans_scratch_bytes": (_i64, [_i64, C.c_int32]),
    "gsvc_ans_encode": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_ans_decode_scratch_bytes": (_i64, [_i64, C.c_int32]),
    "gsvc_ans_decode": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "gsvc_ans_decode_many": (C.c_int, [C.POINTER(AnsDecodeJobC), C.c_int32, _vp]),
    "gsvc_linear_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "gsvc_linear_forward_ex": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "gsvc_linear_wgrad": (C.c_int, [_vp, _vp, _vp, _vp, _i64, C.c_int32, C.c_int32, _vp, _i64, _vp]),
    "gsvc_linear_wgrad_partial": (C.c_int, [_vp, _vp, C.c_int32, _i64, C.c_int32, C.c_int32, _vp, _i64, C.POINTER(C.c_int32), _vp]),
    "gsvc_linear_wgrad_reduce_many": (C.c_int, [C.POINTER(WgradReduceJobC), C.c_int32, _vp]),
    "gsvc_linear_wgrad_partial_many": (C.c_int, [C.POINTER(WgradPartialJobC), C.c_int32, _vp]),
    "gsvc_linear_accumulate_many": (C.c_int, [C.POINTER(AccumJobC), C.c_int32, _vp, _i64, C.c_int32, _vp]),
    "gsvc_linear_forward_shared_input": (C.c_int, [_vp, _i64, C.c_int32, C.POINTER(SharedInputJobC), C.c_int32, _vp]),
    "gsvc_linear_wgrad_workspace": (_i64, [C.c_int32, C.c_int32]),
    "gsvc_generator_saved_floats": (_i64, [C.POINTER(GeneratorNetC), _i64, _i64]),
    "gsvc_generator_scratch_floats": (_i64, [C.POINTER(GeneratorNetC), _i64, _i64]),
    "gsvc_generator_forward": (C.c_int, [C.POINTER(GeneratorNetC), _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsvc_generator_backward": (C.c_int, [C.POINTER(GeneratorNetC), _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int32,
                                          C.POINTER(GeneratorGradsC), _vp]),
    "gsvc_quant_step_nets_forward": (C.c_int, [C.POINTER(QuantStepNetC), _vp, _i64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp]),
    "gsvc_quant_step_nets_backward": (C.c_int, [C.POINTER(QuantStepNetC), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i64, C.c_int32,
                                                C.c_int32, C.POINTER(C.c_void_p), _vp, _vp]),
    "gsvc_deform_saved_floats": (_i64, [C.POINTER(DeformNetC), _i64]),
    "gsvc_deform_scratch_floats": (_i64, [C.POINTER(DeformNetC), _i64]),
    "gsvc_deform_forward": (C.c_int, [C.POINTER(DeformNetC), _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsvc_deform_backward": (C.c_int, [C.POINTER(DeformNetC), _vp, _vp, _i64, _vp, _vp, _vp, _vp, C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                       C.POINTER(DeformGradsC), _vp]),
    "gsvc_generator_inference_floats": (_i64, [C.POINTER(GeneratorNetC), _i64, _i64]),
    "gsvc_generators_forward_inference": (C.c_int, [C.POINTER(GeneratorNetC), C.c_int32, _vp, _vp, _i64, C.POINTER(FilmRowsC), C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_void_p), _vp]),
    "gsvc_deform_forward_inference": (C.c_int, [C.POINTER(DeformNetC), _vp, _vp, _i64, _vp, _vp, _vp]),
    "gsvc_generators_forward": (C.c_int, [C.POINTER(GeneratorNetC), C.c_int32, _vp, _vp, _i64, C.POINTER(FilmRowsC), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), _vp]),
    "gsvc_generators_backward": (C.c_int, [C.POINTER(GeneratorNetC), C.c_int32, _vp, _vp, _i64, C.POINTER(FilmRowsC), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp, C.POINTER(C.c_void_p), C.POINTER(GeneratorGradsC),
                                           _vp]),
    "gsvc_rate_sample_scratch_floats": (_i64, [_i64]),
    "gsvc_rate_sample_forward": (C.c_int, [C.POINTER(RateSampleC), _vp, _vp, _vp]),
    "gsvc_rate_sample_backward": (C.c_int, [C.POINTER(RateSampleC), _vp, _vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp, _vp]),
    "gsvc_rate_backward": (C.c_int, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, C.c_int32, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}


def declared_symbols():
    """Every entry point include/gsvc_hip.h declares (checked by tests/test_abi.py against the header text)."""
    return sorted(_SIGNATURES)


def lib():
    """Load the HIP library, failing loudly when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsvcError(
                f"{LIB_PATH} is not built. Build the gfx950 HIP library first: `make -C gsvc_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). gsvc_amd has no CPU fallback.")
        import torch  # noqa: F401  (loads the HIP runtime this library links against)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if os.environ.get("GSVC_LIB_PATH") and not hasattr(L, name):
                continue          # kernel experiments against an older build: entry points it lacks stay unbound
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
        if _deterministic and hasattr(L, "gsvc_set_deterministic"):
            L.gsvc_set_deterministic(1)
    return _lib


_deterministic = False


def set_deterministic(on: bool):
    """GSVC_DETERMINISTIC (gsvc_amd.switches): told to the library now if it is loaded, else when it loads."""
    global _deterministic
    _deterministic = bool(on)
    if _lib is not None and hasattr(_lib, "gsvc_set_deterministic"):
        _lib.gsvc_set_deterministic(int(_deterministic))


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gsvc_last_error().decode("utf-8", "replace")
        raise GsvcError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL) as a plain integer: ctypes converts an int to the void* parameter itself,
    and a fitting step passes ~2 000 of them (building a c_void_p object for each cost ~0.4 ms of host time per step, round 6)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr()) if _PTR_OBJ else t.data_ptr()


_PTR_OBJ = bool(os.environ.get("GSVC_PTR_OBJ"))      # A/B timing only: the old c_void_p objects


def current_stream(device=None):
    """The current HIP stream of ``device`` as a void pointer (the raw handle: torch.cuda.current_stream() builds a Stream object per
    call, ~6 us each and ~50 calls per fitting step)."""
    import torch
    idx = getattr(device, "index", None) if device is not None and not isinstance(device, int) else device
    if idx is None:
        idx = torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))


def profile_enable(on):
    """bool, or the bit set of include/gsvc_hip.h (1 = per-kernel timing, 2 = the compositing backward's replay counters)."""
    check(lib().gsvc_profile_enable(int(on)), "gsvc_profile_enable")


def profile_collect(max_kernels: int = 64):
    """{kernel name: (launches, total_ms)} since the last collect (synchronises the recorded events)."""
    names = C.create_string_buffer(64 * max_kernels)
    launches = (C.c_int32 * max_kernels)()
    ms = (C.c_float * max_kernels)()
    n = lib().gsvc_profile_collect(names, launches, ms, max_kernels)
    if n < 0:
        check(n, "gsvc_profile_collect")
    out = {}
    for i in range(n):
        name = names.raw[64 * i:64 * (i + 1)].split(b"\0", 1)[0].decode()
        out[name] = (int(launches[i]), float(ms[i]))
    return out
