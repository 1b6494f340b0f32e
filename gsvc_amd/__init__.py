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
    code: such a module is removed (a build artefact; the .py is then imported).  ``compiled_host()`` lists what is loaded."""
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
        # a build artefact that no longer matches its source: take it out of the way (the .py beside it is then what is imported)
        import importlib
        import sys
        try:
            for so in built:
                if os.path.relpath(so, here).split(".cpython-")[0] in stale:
                    try:
                        os.remove(so)
                    except FileNotFoundError:      # another rank of the same launch was faster (every rank imports the package)
                        pass
            importlib.invalidate_caches()
            sys.stderr.write(f"gsvc_amd: removed compiled host modules built from older sources ({', '.join(sorted(stale))}): running the .py "
                             f"files; `python setup_host.py build_ext --inplace` compiles them again\n")
        except OSError as e:
            raise ImportError(f"gsvc_amd: compiled host modules {stale} were built from other sources than the .py files beside them and "
                              f"cannot be removed ({e}): run `python setup_host.py build_ext --inplace` or `python setup_host.py clean_host`")
    return ok


_COMPILED_HOST = _check_compiled_host()


def compiled_host():
    """{module: source hash} of the host modules that run as compiled extensions in this process (empty: plain Python)."""
    return dict(_COMPILED_HOST)
