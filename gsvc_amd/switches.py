"""Run-time switches of the hot path, read from the environment ONCE (at import) — not on every call.

Every switch is a diagnostic: it selects the slower / reference-shaped path of one fused piece so that the two can be compared
(tests, same-process A/B timing: tools/ab/ab_env.py), or it sizes a resource.  The product runs with none of them set.
Code reads the module attributes (``switches.NO_MLP_CHAIN``); a process that changes ``os.environ`` afterwards calls ``reload()``
(tests/conftest.py does after every ``monkeypatch.setenv`` / ``delenv``).

    GSVC_NO_MLP_CHAIN      every MLP layer by layer (no whole-network chain kernels, no one-function networks)
    GSVC_NO_MLP_FUSED      same for the generators / mlp_deform chain kernels only
    GSVC_NO_QUANT_CHAIN    the three quant_step networks through the layer kernels (not k_quant_nets_*)
    GSVC_NO_SHARED_INPUT / GSVC_NO_ACCUM_MANY   no multi-product linear launches;  GSVC_MANY_MIN_ROWS  rows from which they are used
    GSVC_NO_DECODE_CHAIN   the decoder loop through cached trunks + per-layer heads
    GSVC_CTX_ALL_ROWS      the entropy context's dist networks on every distinct anchor (not only the rate sample's rows)
    GSVC_NO_FUSED_CTX / _GRID / _STATIS / _RATE / _GATHER / _PLAN / _STE, GSVC_NO_RANKED_GATHER, GSVC_NO_PACKED_GRID   tensor forms of those pieces
    GSVC_NO_GRID_MANY      Mix3d2dEncoding's four hash grids as four launches each way (one launch otherwise)
    GSVC_NO_FILM_SHARE     FiLM networks per (view, anchor) instead of per (frame, anchor)
    GSVC_NO_VIEW_SHARE     FULL_PRECISION / STE_ENTROPY steps: the generators per (view, anchor) although the two opposite views of a frame
                           have the same Gaussians (the generation pass per (frame, anchor) otherwise)
    GSVC_NO_RATE_OVERLAP   the sampled rate inside the generation pass, on the step's stream (not behind the rasterizer's launches on its own)
    GSVC_NO_WGRAD_OVERLAP  the generators' / mlp_deform's weight gradients on the step's stream (not on their own stream beside the feature
                           gradient's way back through the quantisers, gathers and hash grid)
    GSVC_RATE_EARLY        the sampled rate issued (on its own stream) BEFORE the rasterizer's launches: its backward then runs behind the rasterizer's
    GSVC_NO_LATE_ROWS      gather offsets / scalings / masks with the features (not behind the generators)
    GSVC_NO_PREFETCH / GSVC_NO_EARLY_PLAN / GSVC_EARLY_PLAN   step plan off / never from inside the backward / always
    GSVC_NO_ADAPTIVE_BOUND the step's "is the GPU or the host the bound" decisions by the row count only (not by the host's measured blocked time)
    GSVC_RASTER_LOOSE_BINNING   the renderer lists every Gaussian in all tiles of its 3-sigma rectangle (default: only where its alpha box reaches)
    GSVC_RASTER_STREAMS    side streams the step's renders are dealt to (default 2; 1 = all on the current stream)
    GSVC_DETERMINISTIC     every float sum of a fitting step in a fixed order: sort-based row scatters instead of float atomics, one workgroup
                           per hash-table slice, per-row launch choices by the row count only (no wall-clock measurement picks a kernel) ->
                           the same gradient bits run after run (tools/ab/determinism_check.py); the default keeps the faster forms
    GSVC_DP_SPARSE         data parallel: 0 = dense all-reduce always, 1 = row-sparse exchange always (default: whichever moves less)
    GSVC_DP_ZOWN / GSVC_DP_ZOWN_CHECK / GSVC_DP_FORCE   read by gsvc_amd/dist.py where the process group is known: per-anchor tensors owned by
                           z-range (halo exchange), its dropped-gradient check, the data-parallel path on a one-rank group (tests)
"""
import os

_FLAGS = ("NO_MLP_CHAIN", "NO_MLP_FUSED", "NO_QUANT_CHAIN", "NO_SHARED_INPUT", "NO_ACCUM_MANY", "NO_DECODE_CHAIN", "CTX_ALL_ROWS",
          "NO_FUSED_CTX", "NO_FUSED_GRID", "NO_FUSED_STATIS", "NO_FUSED_RATE", "NO_FUSED_GATHER", "NO_FUSED_PLAN", "NO_FUSED_STE", "NO_RANKED_GATHER",
          "NO_PACKED_GRID", "NO_GRID_MANY", "NO_FILM_SHARE", "NO_VIEW_SHARE", "NO_LATE_ROWS", "NO_RATE_OVERLAP", "NO_WGRAD_OVERLAP", "NO_PREFETCH", "NO_EARLY_PLAN", "EARLY_PLAN", "RATE_EARLY", "RASTER_LOOSE_BINNING", "NO_ADAPTIVE_BOUND", "DETERMINISTIC")


def reload():
    g = globals()
    for name in _FLAGS:
        g[name] = bool(os.environ.get("GSVC_" + name))
    g["RASTER_STREAMS"] = int(os.environ.get("GSVC_RASTER_STREAMS", "2"))
    g["MANY_MIN_ROWS"] = int(os.environ.get("GSVC_MANY_MIN_ROWS", "24576"))
    g["DP_SPARSE"] = os.environ.get("GSVC_DP_SPARSE")          # None | "0" | "1"
    from . import _lib
    _lib.set_deterministic(g["DETERMINISTIC"])          # the library's own fixed-order forms (hash-grid backward); applied at load if not loaded yet


reload()
