"""tools/calibrate_conventions.py on the CPU: the module it calibrates is the oracle (oracle/raster_oracle.c behind the extension's
API: tests/golden/_ref_import._oracle_rasterizer_module) holding a flag combination the calibrator cannot see — the settings type has
the reference's fields only (ortho_gaussian_renderer/renderer.py:63-83) — and must read off results.  All 64 combinations of the six
convention switches of include/gsvc_hip.h, non-default low-pass values, and the fixture it records."""
import os

import numpy as np
import pytest

from tests.golden._ref_import import _oracle_rasterizer_module
from tools import calibrate_conventions as cal


@pytest.mark.parametrize("block", range(8))
def test_calibrator_recovers_hidden_flags_from_the_oracle(oracle_lib, block):
    for flags in range(8 * block, 8 * block + 8):
        res = cal.calibrate(_oracle_rasterizer_module(flags=flags), device="cpu", log=lambda *_: None)
        assert res["flags"] == flags and res["low_pass"] == 0.0, (flags, res)


@pytest.mark.parametrize("flags,low_pass", [(0, 0.55), (2 | 8, 0.1), (1 | 4 | 16, 1.5)])
def test_calibrator_recovers_a_non_default_low_pass(oracle_lib, flags, low_pass):
    res = cal.calibrate(_oracle_rasterizer_module(flags=flags, low_pass=low_pass), device="cpu", log=lambda *_: None)
    assert res["flags"] == flags and abs(res["low_pass"] - low_pass) <= 2e-3, res


def test_calibrator_refuses_what_no_switch_describes(oracle_lib):
    """A module whose images come out transposed is not one of the 64 conventions: an error, not a wrong answer."""
    import types
    m = _oracle_rasterizer_module()

    class Transposed(m.GaussianRasterizer):
        def forward(self, *a, **k):
            image, radii, n = super().forward(*a, **k)
            return image.transpose(1, 2).contiguous(), radii, n
    bad = types.SimpleNamespace(GaussianRasterizationSettings=m.GaussianRasterizationSettings, GaussianRasterizer=Transposed)
    with pytest.raises(cal.CalibrationError):
        cal.calibrate(bad, device="cpu", log=lambda *_: None)


def test_recorded_fixture_holds_results_only(oracle_lib, tmp_path):
    """The fixture = the scene's inputs + what the module made of them (numbers only: nothing of the module itself)."""
    m = _oracle_rasterizer_module(flags=2 | 8)
    fx = cal.record_fixture(m, device="cpu")
    assert fx["image"].shape == (3, 256, 256) and fx["radii"].dtype == np.int32 and int(fx["num_rendered"]) > 3000
    assert np.array_equal(fx["radii"], fx["radii_visible_filter"])
    for k in ("means3D", "means2D", "colors", "opacities", "scales", "rotations"):
        g = fx["grad_" + k]
        assert g.dtype == np.float32 and np.isfinite(g).all() and np.abs(g).max() > 0, k
    p = tmp_path / "cal.npz"
    np.savez_compressed(p, flags=np.int64(10), low_pass=np.float64(0.0), module=np.array("oracle"), **fx)
    z = np.load(p)
    assert all(z[k].dtype.kind in "fiuU" for k in z.files)          # floats, integers, one name string: data, not code
    assert p.stat().st_size < 2_000_000


def test_the_one_command_writes_the_fixture(oracle_lib, tmp_path):
    """`python tools/calibrate_conventions.py --module <dotted name>` as a maintainer runs it (here: the oracle-backed stand-in module,
    holding flags 4 | 8, on the CPU): prints the readings, returns the flags, writes the .npz with the module's results."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "raster_calibration.npz"
    run = subprocess.run([sys.executable, os.path.join(root, "tools", "calibrate_conventions.py"), "--module", "tests._oracle_extension_module",
                          "--device", "cpu", "--out", str(out)], cwd=root, env=dict(os.environ, ORACLE_EXT_FLAGS="12"), capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "flags = 12 (GSVC_RASTER_DEPTH_DESCENDING | GSVC_RASTER_MEANS2D_PIXEL_UNITS)" in run.stdout and "set pipe.raster_flags = 12" in run.stdout
    z = np.load(out)
    assert int(z["flags"]) == 12 and float(z["low_pass"]) == 0.0 and str(z["module"]) == "tests._oracle_extension_module"
    assert z["image"].shape == (3, 256, 256) and int(z["num_rendered"]) > 3000
