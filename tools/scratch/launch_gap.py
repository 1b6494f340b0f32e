"""Cost of one dependent launch in a stream: N tiny kernels back to back (eager, and replayed from a captured graph)."""
import torch, time
x = torch.zeros(64, device="cuda")
big = torch.zeros(64 * 1024 * 1024, device="cuda")
def run(n):
    for _ in range(n):
        x.add_(1.0)
for _ in range(3): run(100)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# keep the queue full: a long kernel first so that the host runs ahead
big.add_(1.0); big.add_(1.0); big.add_(1.0)
e0.record(); run(2000); e1.record(); torch.cuda.synchronize()
print(f"eager, queue pre-filled: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per tiny launch")
t0 = time.perf_counter(); run(2000); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host enqueue: {(t1 - t0) / 2000 * 1e6:.2f} us per launch")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(10)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        run(2000)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"graph replay: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per tiny launch")
