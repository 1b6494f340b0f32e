python tools/bench_grid.py
GSVC_GRID_DBG=1 python tools/bench_grid.py
GSVC_GRID_DBG=2 python tools/bench_grid.py
GSVC_GRID_DBG=3 python tools/bench_grid.py
GSVC_GRID_WGS=256 python tools/bench_grid.py
GSVC_GRID_WGS=1024 python tools/bench_grid.py
