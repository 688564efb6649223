#!/usr/bin/env python3
"""Encoder attention alone (B = 32, 1500 positions, 20 heads of 64): us per call, plain form and the prescaled (lagged-maximum)
form the LayerNorm-folded encoder runs; for in-process / env A-B experiments."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
qkv_pre = qkv.clone()
qkv_pre[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for name, x, pre in (("plain", qkv, False), ("prescaled", qkv_pre, True), ("plain", qkv, False), ("prescaled", qkv_pre, True)):
    for _ in range(5): ops.attention_packed(x, 20, q_prescaled=pre)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0.record()
        for _ in range(20): ops.attention_packed(x, 20, q_prescaled=pre)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    res.setdefault(name, []).extend(ts)
for name, ts in res.items():
    ts = sorted(ts)
    print(f"{os.environ.get('TAG', '')} attention B=32 {name:10s}: min {ts[0]*1e3:.1f} us  median {ts[len(ts)//2]*1e3:.1f} us  = {32*20*4*1500*1500*64/ts[0]/1e9:.0f} TFLOP/s")
a = ops.attention_packed(qkv, 20).float()
b = ops.attention_packed(qkv_pre, 20, q_prescaled=True).float()
print("plain vs prescaled max abs diff", float((a - b).abs().max()))
