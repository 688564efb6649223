#!/usr/bin/env python3
"""Runs the 32-layer encoder at B = 32 in the e4m3-projection mode only (for rocprofv3 --kernel-trace --stats): per-kernel
times of the quantisation passes and the fp8 GEMMs.  `AFHIP_FP8_FC2=1` adds fc2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
enc = bench.build_encoder(dev, torch.bfloat16, None)
mode = os.environ.get("PROBE_MODE", "fp8")
enc.enable_fp8(mode in ("fp8", "fp8static"))
g = torch.Generator(device=dev).manual_seed(0)
mel = (torch.randn((32, 3000, 128), generator=g, device=dev) * 0.5).to(torch.bfloat16)
if mode == "fp8static":
    enc.calibrate_fp8((torch.randn((4, 3000, 128), generator=g, device=dev) * 0.5).to(torch.bfloat16))   # PROBE_MODE=fp8static: fc2 input quantised in fc1's epilogue
for _ in range(2):
    enc.encode_btc(mel)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    enc.encode_btc(mel)
e1.record(); torch.cuda.synchronize()
print(f"{mode}: {e0.elapsed_time(e1)/5:.2f} ms per forward")
