timeout 900 python -m pytest tests -q -m gpu --tb=line -x 2>&1 | tail -3 | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --workload train_step --steps 30 --warmup 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
