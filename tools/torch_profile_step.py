"""torch.profiler view of one fitting step (diagnostic): which ATen ops the GPU time belongs to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import ProfilerActivity, profile
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.train import Trainer
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, device=dev).materialize()
mp_.threshold = 8.0 / cube.scale
opt.full_precision_training_total = opt.quantized_training_total = 0
opt.entropy_constrained_train_total = 10 ** 9
opt.start_stat, opt.update_until, opt.pause_densification = 0, 10 ** 9, 0
pc = GaussianModel(mp_, 50, 10, 0.001, 3, 16, 4, False, n_features_per_level=8, log2_hashmap_size=13, log2_hashmap_size_2D=15, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (220000, 3)), 1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
pc.training_setup(opt)
tr = Trainer(pc, cube, opt, pipe, mp_)
for i in range(30):
    tr.step(i + 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(os.environ.get('STACKS')), record_shapes=bool(os.environ.get('SHAPES'))) as prof:
    for i in range(3):
        tr.step(10 + i, frame_idx=30)
    torch.cuda.synchronize()
if os.environ.get('GSVC_REGIONS'):
    for ev in sorted([e for e in prof.key_averages() if e.key.startswith(('gen.', 'step.'))], key=lambda e: e.key):
        print(f"{ev.key:24s} cpu_total {ev.cpu_time_total / 3e3:8.2f} ms/step   cuda_total {ev.device_time_total / 3e3:8.2f} ms/step")
elif os.environ.get('SHAPES'):
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=int(os.environ['SHAPES']), max_name_column_width=50))
elif os.environ.get('STACKS'):
    want = os.environ['STACKS'].split(',')
    from collections import Counter
    cnt = Counter()
    for ev in prof.events():
        if ev.name in want:
            st = [f for f in ev.stack if '/gsvc_amd/' in f or 'bench' in f][:3]
            cnt[(ev.name, ' <- '.join(x.split('/gsvc_amd/')[-1] for x in st))] += 1
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
        print(v, k[0], k[1])
else:
    print(prof.key_averages().table(sort_by=os.environ.get("SORT", "self_cuda_time_total"), row_limit=45, max_name_column_width=60))
