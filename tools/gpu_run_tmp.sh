timeout 900 python -m pytest tests/test_train_gpu.py -q -m gpu --tb=short -x 2>&1 | tail -8 | cut -c1-250
