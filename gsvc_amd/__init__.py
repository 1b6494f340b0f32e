"""gsvc_amd — MI355X-native hot path of GSVC (orthographic sliding-window Gaussian rasterizer, hash-grid
encoder, entropy-rate estimator and the fitting step around them) behind GSVC's own renderer API.

The compute path is hand-written HIP for gfx950 in ``gsvc_amd/csrc`` behind the C-ABI of
``include/gsvc_hip.h``; there is no CPU fallback: importing a compute entry point without the built
library raises.
"""
__version__ = "0.1.0"


def _check_compiled_host():
    """The host modules of the fitting step may be compiled in place (``python setup_host.py build_ext --inplace``: Cython, the .py
    files stay the source).  The import system prefers the compiled module, so one built from an OLDER .py would silently run old
    code: refuse that.  ``compiled_host()`` lists what is loaded."""
    import glob
    import hashlib
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    built = glob.glob(os.path.join(here, "*.cpython-*.so")) + glob.glob(os.path.join(here, "*", "*.cpython-*.so"))
    if not built:
        return {}
    try:
        rec = json.load(open(os.path.join(here, "_host_build.json")))
    except Exception:  # noqa: BLE001
        rec = {}
    stale, ok = [], {}
    for so in built:
        mod = os.path.relpath(so, here).split(".cpython-")[0]
        src = os.path.join(here, mod + ".py")
        now = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] if os.path.exists(src) else None
        if rec.get(mod.replace(os.sep, "/")) != now:
            stale.append(mod)
        else:
            ok[mod.replace(os.sep, "/")] = now
    if stale:
        raise ImportError(f"gsvc_amd: compiled host modules {stale} were built from other sources than the .py files beside them: "
                          f"run `python setup_host.py build_ext --inplace` (or `python setup_host.py clean_host` to run the .py files)")
    return ok


_COMPILED_HOST = _check_compiled_host()


def compiled_host():
    """{module: source hash} of the host modules that run as compiled extensions in this process (empty: plain Python)."""
    return dict(_COMPILED_HOST)
