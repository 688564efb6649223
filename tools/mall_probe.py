#!/usr/bin/env python3
"""Does a weight slice that was just read (and so sits in the 256-MiB Infinity Cache) make the decode gate/up GEMM faster?
Times afhip_gemm_skinny (M = 8, N = 37888, K = 3584, SwiGLU epilogue, RMSNorm A) after (a) a 300-MB unrelated read (cold) and
(b) reading the first P MB of its own weights."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import _lib as L
lib = L.lib()
dev = "cuda:0"
M, N, K = 8, 37888, 3584
g = torch.Generator(device=dev).manual_seed(0)
w = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
other = torch.empty(300 * 1024 * 1024 // 4, device=dev, dtype=torch.int32).zero_()
a = L.GemmArgs()
a.A, a.W, a.C, a.M, a.N, a.K, a.lda, a.ldw, a.ldc = x.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, K, N // 2
a.dtype, a.act, a.a_norm_w, a.a_norm_eps = L.BF16, L.ACT_SWIGLU, nw.data_ptr(), 1e-6
def run():
    L.check(lib.afhip_gemm_skinny(C.byref(a), L.stream_ptr()))
wi = w.view(torch.int32).view(-1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for P in (0, 16, 32, 64, 96, 128, 192, 271):
    ts = []
    for rep in range(12):
        other.sum()                          # evict: 300 MB of unrelated traffic
        if P:
            wi[: P * 1024 * 1024 // 4].sum()   # warm the first P MB of the weights
        torch.cuda.synchronize()
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"warm {P:4d} MB: gate/up skinny GEMM median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f} us   ({N*K*2/ts[len(ts)//2]/1e6:.2f} TB/s)", flush=True)
