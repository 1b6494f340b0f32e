python -m pytest tests/test_grid_rate_gpu.py -q -m gpu --tb=short -k "linear" 2>&1 | grep -v "^$" | tail -40 | cut -c1-200
