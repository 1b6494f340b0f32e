"""A module with the API of GSVC's external rasterizer extension (`GaussianRasterizationSettings`, `GaussianRasterizer`) backed by the CPU
oracle, holding conventions of its own that its settings type does not show (ORACLE_EXT_FLAGS / ORACLE_EXT_LOW_PASS, read at import):
the stand-in `tools/calibrate_conventions.py --module tests._oracle_extension_module` is run against in tests/test_calibrate_cpu.py."""
import os

from tests.golden._ref_import import _oracle_rasterizer_module

_m = _oracle_rasterizer_module(flags=int(os.environ.get("ORACLE_EXT_FLAGS", "0")), low_pass=float(os.environ.get("ORACLE_EXT_LOW_PASS", "0")))
GaussianRasterizationSettings = _m.GaussianRasterizationSettings
GaussianRasterizer = _m.GaussianRasterizer
