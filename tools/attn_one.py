#!/usr/bin/env python3
"""A few launches of the encoder attention shape (B x 20 heads x 1500 x 64, bf16) for rocprofv3 runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
qkv = torch.randn(B, 1500, 3 * 1280, device="cuda", dtype=torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.attention_packed(qkv, 20)
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    ops.attention_packed(qkv, 20)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"attention B={B}: {ms:.3f} ms  {B*20*4*1500*1500*64/ms/1e9:.1f} TFLOP/s")
