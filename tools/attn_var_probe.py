#!/usr/bin/env python3
"""A/B of the attn_enc64 variants (AFHIP_ENC64_VAR, read per call) in ONE process, interleaved rounds (cdna guide rule 24)."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
qkv[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,8,12".split(","))]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {v: [] for v in variants}
os.environ["AFHIP_ENC64_VAR"] = "0"
ref = ops.attention_packed(qkv, 20, q_prescaled=True).float()
for rnd in range(6):
    for v in variants:
        os.environ["AFHIP_ENC64_VAR"] = str(v)
        for _ in range(3): ops.attention_packed(qkv, 20, q_prescaled=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10): out = ops.attention_packed(qkv, 20, q_prescaled=True)
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 10)
for v in variants:
    os.environ["AFHIP_ENC64_VAR"] = str(v)
    d = float((ops.attention_packed(qkv, 20, q_prescaled=True).float() - ref).abs().max())
    ts = sorted(res[v])
    print(f"var {v:2d}: min {ts[0]*1e3:7.1f} us  median {ts[len(ts)//2]*1e3:7.1f} us  = {32*20*4*1500*1500*64/ts[0]/1e9:5.0f} TFLOP/s   max |diff vs var 0| {d:.3g}")
