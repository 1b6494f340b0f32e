python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -6 | cut -c1-300
