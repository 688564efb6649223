#!/usr/bin/env python3
"""Per-CU GEMM rate on problems of different footprint (is the K loop limited by beyond-L2 latency?)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
dt = torch.bfloat16
for m, n, k in [(2048, 2048, 8192), (4096, 4096, 8192), (1024, 4096, 16384), (8192, 8192, 8192), (256 * 16, 256 * 16, 1280), (16384, 16384, 1280), (16384, 16384, 5120)]:
    a = torch.randn(m, k, device="cuda", dtype=dt)
    w = torch.randn(n, k, device="cuda", dtype=dt) * 0.03
    out = torch.empty(m, n, device="cuda", dtype=dt)
    for _ in range(3):
        ops.gemm(a, w, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.gemm(a, w, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles = (m // 256) * (n // 256)
    rounds = -(-tiles // 256)
    cus = min(tiles, 256)
    tf = 2.0 * m * n * k / ms / 1e9
    print(f"M={m:6d} N={n:6d} K={k:6d} tiles={tiles:5d} rounds={rounds:3d}  {ms:8.3f} ms  {tf:8.1f} TFLOP/s  per-busy-CU {tf/cus:6.2f} TF  tile-time {ms/rounds*1e3:8.1f} us  (ideal tile {2*256*256*k/ (2500e12/256) *1e6:6.1f} us)")
