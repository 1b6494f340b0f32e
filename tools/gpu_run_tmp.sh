timeout 900 python -m pytest tests/test_raster_gpu.py -q -m gpu --tb=short -x -k full_size 2>&1 | tail -12 | cut -c1-250
