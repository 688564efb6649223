#!/usr/bin/env python3
"""Per-shape timing of afhip_gemm on the encoder / LLM problem sizes (bf16), for kernel tuning.
usage: python tools/gemm_bench.py [B]   (B = clips, default 32)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops, _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
dt = torch.bfloat16
M = 1500 * B
shapes = [("qkv", M, 3840, 1280, L.ACT_NONE, False), ("out+res", M, 1280, 1280, L.ACT_NONE, True),
          ("fc1+gelu", M, 5120, 1280, L.ACT_GELU, False), ("fc1 plain", M, 5120, 1280, L.ACT_NONE, False),
          ("fc2+res", M, 1280, 5120, L.ACT_NONE, True), ("square 4096", 4096, 4096, 4096, L.ACT_NONE, False),
          ("square 8192", 8192, 8192, 8192, L.ACT_NONE, False)]
for name, m, n, k, act, res in shapes:
    a = torch.randn(m, k, device=dev, dtype=dt)
    w = torch.randn(n, k, device=dev, dtype=dt) * 0.03
    bias = torch.randn(n, device=dev, dtype=dt)
    r = torch.randn(m, n, device=dev, dtype=dt) if res else None
    out = torch.empty(m, n, device=dev, dtype=dt)
    for _ in range(3):
        ops.gemm(a, w, bias=bias, act=act, residual=r, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    n_it = 10
    for _ in range(n_it):
        ops.gemm(a, w, bias=bias, act=act, residual=r, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n_it
    # vendor-library ceiling on the same device and data: torch.matmul -> hipBLASLt / rocBLAS (plain GEMM, no epilogue)
    wt = w.t()
    for _ in range(3):
        torch.matmul(a, wt, out=out)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n_it):
        torch.matmul(a, wt, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms_lib = e0.elapsed_time(e1) / n_it
    print(f"{name:14s} M={m:6d} N={n:5d} K={k:5d}  {ms:8.3f} ms  {2.0*m*n*k/ms/1e9:8.1f} TFLOP/s   | hipBLASLt plain {ms_lib:8.3f} ms {2.0*m*n*k/ms_lib/1e9:8.1f} TFLOP/s", flush=True)
