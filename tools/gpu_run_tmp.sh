timeout 900 python -m pytest tests/test_raster_gpu.py -q -m gpu --tb=line -x 2>&1 | tail -4 | cut -c1-250
python bench.py --workload raster_fwdbwd --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
