"""Where the stream codec's time goes (diagnostic): wall clock of encode / decode, per-kernel HIP-event times of the coder
kernels, and a torch.profiler table of one decode."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import ProfilerActivity, profile
from gsvc_amd import _lib
from gsvc_amd.arguments import cfg_20240919
from gsvc_amd.frame import SyntheticFrameCube
from gsvc_amd.model import GaussianModel
from gsvc_amd.stream_codec import conduct_stream_decoding, conduct_stream_encoding
dev = torch.device("cuda")
mp_, opt, pipe = cfg_20240919()
cube = SyntheticFrameCube(1080, 1920, 64, device=dev)
mp_.threshold = 8.0 / cube.scale
pc = GaussianModel(mp_, 50, 10, 0.001, 3, 16, 4, False, n_features_per_level=8, log2_hashmap_size=13, log2_hashmap_size_2D=15, device=dev)
rng = np.random.default_rng(0)
lim = np.array([cube.x_min, cube.y_min, cube.z_min]) * 1.1
pc.create_from_points(rng.uniform(lim, -lim, (int(os.environ.get("ANCHORS", 245000)), 3)), 1.0)
pc.update_anchor_bound(cube.x_min, cube.y_min, cube.z_min)
torch.manual_seed(0)
pc._anchor_feat.data.normal_(0, float(os.environ.get("FEAT_STD", 2.0)))
pc._offset.data.normal_(0, float(os.environ.get("OFF_STD", 0.5)))
pack = conduct_stream_encoding(pc)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pack = conduct_stream_encoding(pc)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    _lib.profile_enable(True)
    dec = conduct_stream_decoding(copy.deepcopy(pc), pack)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    prof = _lib.profile_collect(); _lib.profile_enable(False)
    print(f"rep {rep}: encode {1e3 * (t1 - t0):.1f} ms  decode {1e3 * (t2 - t1):.1f} ms (incl. deepcopy)", {k: (n, round(ms, 2)) for k, (n, ms) in prof.items() if "ans" in k})
pc2 = copy.deepcopy(pc)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    conduct_stream_decoding(pc2, pack)
    torch.cuda.synchronize()
if os.environ.get("LAUNCHES"):
    for ev in p.events():
        if "k_ans_decode" in ev.name and ev.device_time > 0:
            print("launch", ev.name[:30], f"{ev.device_time:.0f} us")
    print({k: (len(v), v[:80]) for k, v in (("feat", pack.feat), ("scaling", pack.scaling), ("offsets", pack.offsets))} if False else [(len(a), len(b), len(c)) for a, b, c in zip(pack.feat, pack.scaling, pack.offsets)], len(pack.masks), len(pack.hash))
else:
    print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=25, max_name_column_width=50))
