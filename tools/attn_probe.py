#!/usr/bin/env python3
"""Encoder attention alone (B = 32, 1500 positions, 20 heads of 64): ms per call, for in-process / env A-B experiments."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
for _ in range(5): ops.attention_packed(qkv, 20)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for r in range(5):
    e0.record()
    for _ in range(20): ops.attention_packed(qkv, 20)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
print(f"{os.environ.get('TAG', '')} attention B=32: {min(ts)*1e3:.1f} us (median {sorted(ts)[2]*1e3:.1f})  = {32*20*4*1500*1500*64/min(ts)/1e9:.0f} TFLOP/s")
