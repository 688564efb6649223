"""Does the encoder attention kernel slow down right after MFMA-dense GEMMs (board power / clock state), as it does inside the step
(470-490 us per layer in the headline profile against 420-430 alone)?  Times ONE attention call with HIP events directly after a burst of
n fc1-shaped GEMMs, for n = 0, 4, 16."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops, _lib as L
dev, dt = "cuda:0", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(32, 1500, 3 * 1280, device=dev, generator=g) * 0.5).to(dt)
qkv[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(dt)
a = torch.randn(48000, 1280, device=dev, dtype=dt)
w = torch.randn(5120, 1280, device=dev, dtype=dt) * 0.03
bias = torch.randn(5120, device=dev, dtype=dt)
out = torch.empty(48000, 5120, device=dev, dtype=dt)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): ops.attention_packed(qkv, 20, q_prescaled=True)
for n in (0, 4, 16, 0, 16):
    ts = []
    for rep in range(12):
        for _ in range(n): ops.gemm(a, w, bias=bias, act=L.ACT_GELU, out=out)
        e0.record(); ops.attention_packed(qkv, 20, q_prescaled=True); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    print(f"attention after {n:2d} GEMMs: median {ts[len(ts) // 2]:.0f} us  min {ts[0]:.0f}  max {ts[-1]:.0f}", flush=True)
