"""CPU test of the drop-in boundary: libgsvc_hip.so loads without a GPU and exports every entry point that
include/gsvc_hip.h declares (no compute is called), argument validation works on the host side, and the
product package never imports the oracle."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_header():
    text = open(os.path.join(ROOT, "include", "gsvc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsvc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def hip_lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "gsvc_amd", "csrc", "libgsvc_hip.so")):
        g.build()
    from gsvc_amd import _lib
    return _lib


def test_header_and_binding_agree(hip_lib):
    names = _declared_in_header()
    assert names == hip_lib.declared_symbols()
    assert len(names) >= 16


def test_library_exports_every_declared_symbol(hip_lib):
    raw = C.CDLL(hip_lib.LIB_PATH)
    for name in _declared_in_header():
        assert hasattr(raw, name), name
    assert hip_lib.lib().gsvc_version().startswith(b"gsvc_hip")


def test_host_side_validation_without_gpu(hip_lib):
    L = hip_lib.lib()
    s = hip_lib.RasterSettingsC()
    s.image_height, s.image_width = 1080, 1920
    sz = hip_lib.RasterSizesC()
    assert L.gsvc_raster_sizes_query(C.byref(s), 200000, 800000, C.byref(sz)) == 0
    assert sz.geom_bytes >= 200000 * 80 and sz.binning_bytes >= 800000 * 20 and sz.image_bytes >= 1080 * 1920 * 8
    s.image_height = 0
    assert L.gsvc_raster_sizes_query(C.byref(s), 1, 1, C.byref(sz)) == -1
    assert b"image size must be positive" in L.gsvc_last_error()
    with pytest.raises(hip_lib.GsvcError):
        hip_lib.check(-1, "probe")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gsvc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libgsvc_oracle" not in text, f


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from gsvc_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GsvcError, match="no CPU fallback"):
        _lib.lib()
