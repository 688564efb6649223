"""GPU-side cost of a kernel boundary on this chip: 2000 dependent launches of a one-workgroup kernel captured into ONE graph and replayed
(no CPU launch cost in the figure), and the same with a 256-workgroup kernel that touches 64 MB (so that the caches have something to
write back at every boundary).  The encoder step has ~330 launches."""
import sys, os, torch
dev = "cuda:0"
x = torch.zeros(64, device=dev)
big = torch.zeros(16 * 1024 * 1024, device=dev)          # 64 MB
def run(fn, n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"tiny kernel (64 floats): {run(lambda: x.add_(1.0), 2000):.2f} us per dependent launch")
t_big = run(lambda: big.add_(1.0), 200)
print(f"64 MB read + 64 MB write kernel: {t_big:.1f} us per dependent launch = {2 * 64e6 * 1.048576 / t_big / 1e6:.2f} TB/s incl. its boundary")
