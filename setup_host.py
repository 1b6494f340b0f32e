"""Compile the host side of the fitting step (the Python modules a step executes ~9 000 function calls of) to C extension modules with
Cython, in place: ``python setup_host.py build_ext --inplace`` (``__graft_entry__.build()`` runs it).  The sources stay plain Python —
the .py files are what is edited, read and tested; the compiled modules (git-ignored ``*.so`` beside them, which the import system
prefers) only take the bytecode interpreter out of the step's hot path.  ``GSVC_NO_COMPILED_HOST=1 python setup_host.py clean_host``
removes them again."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
MODULES = ["_lib", "generate", "mlp", "train", "rasterizer", "loss_utils", "optim", "model", "encodings", "entropy_models", "dist",
           "ortho_gaussian_renderer/renderer", "ortho_gaussian_renderer/preprocess"]


def compiled_files():
    out = []
    for m in MODULES:
        out += glob.glob(os.path.join(ROOT, "gsvc_amd", m + ".cpython-*.so"))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "clean_host":
        for f in compiled_files() + [os.path.join(ROOT, "gsvc_amd", "_host_build.json")]:
            if os.path.exists(f):
                os.remove(f)
                print("removed", os.path.relpath(f, ROOT))
        sys.exit(0)
    from Cython.Build import cythonize
    from setuptools import Extension, setup
    exts = [Extension("gsvc_amd." + m.replace("/", "."), [os.path.join("gsvc_amd", m + ".py")], extra_compile_args=["-O2", "-g0", "-w"])
            for m in MODULES]
    os.chdir(ROOT)
    setup(name="gsvc_amd_host", packages=[],
          ext_modules=cythonize(exts, language_level=3, build_dir="build/cython", nthreads=0,
                                compiler_directives={"binding": True, "always_allow_keywords": True, "annotation_typing": False}))
    # the generated C sources and objects are intermediates (4 MB of machine-written C per module): gone once the modules exist
    import shutil
    for d in glob.glob(os.path.join(ROOT, "build", "cython")) + glob.glob(os.path.join(ROOT, "build", "temp.*")) + \
            glob.glob(os.path.join(ROOT, "build", "lib.*")):
        shutil.rmtree(d, ignore_errors=True)
    try:
        os.rmdir(os.path.join(ROOT, "build"))
    except OSError:
        pass
    # what each compiled module was built from: gsvc_amd/__init__.py refuses to run a compiled module whose .py has changed since
    import hashlib
    import json
    rec = {m: hashlib.sha256(open(os.path.join(ROOT, "gsvc_amd", m + ".py"), "rb").read()).hexdigest()[:16] for m in MODULES}
    with open(os.path.join(ROOT, "gsvc_amd", "_host_build.json"), "w") as f:
        json.dump(rec, f, indent=1)
