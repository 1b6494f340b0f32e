python bench.py 2>&1 | tail -1 > gpurun_out/bench_headline.json; python -c "
import json; d=json.load(open('gpurun_out/bench_headline.json')); 
print(d['value'], d['ms_per_step'], d['render_fps'], d['roofline']); print(d['cpu_baseline']); print({k:v for k,v in d['train_step'].items() if k not in ('kernels','config')})"
