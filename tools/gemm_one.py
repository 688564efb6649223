#!/usr/bin/env python3
"""Run a few launches of one GEMM shape (for rocprofv3 --pmc runs): python tools/gemm_one.py M N K [gelu]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops, _lib as L
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
act = L.ACT_GELU if len(sys.argv) > 4 else L.ACT_NONE
a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.03
b = torch.randn(n, device="cuda", dtype=torch.bfloat16)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(a, w, bias=b, act=act, out=out)
torch.cuda.synchronize()
