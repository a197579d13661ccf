"""Can HIP events recorded INSIDE a captured hipGraph be used for timing after a replay?  (bench.py: per-kernel durations of the replayed
step.)  python tools/graph_event_probe.py"""
import torch

x = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
y = torch.empty_like(x)
side = torch.cuda.Stream()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
for e in (e0, e1, e2):
    e.record()
torch.cuda.synchronize()
with torch.cuda.stream(side):
    torch.matmul(x, x, out=y)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
    e0.record()
    torch.matmul(x, x, out=y)
    e1.record()
    torch.matmul(x, x, out=y)
    torch.matmul(x, x, out=y)
    e2.record()
for i in range(3):
    g.replay()
    torch.cuda.synchronize()
    try:
        print('replay', i, 'one matmul %.1f us, two matmuls %.1f us' % (e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3))
    except Exception as ex:
        print('replay', i, 'elapsed_time failed:', type(ex).__name__, str(ex)[:200])
