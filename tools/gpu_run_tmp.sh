python tools/bench_grid.py
timeout 600 python -m pytest tests/test_grid_rate_gpu.py -q -m gpu --tb=short -x -k "grid" 2>&1 | tail -3 | cut -c1-250
