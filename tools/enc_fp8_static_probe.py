#!/usr/bin/env python3
"""32-layer encoder at B = 32: bf16, e4m3 with dynamic activation scales (out + fc1), and e4m3 with fc2's input statically quantised in fc1's
epilogue (calibrate_fp8): ms per forward of each and the output error of both e4m3 forms against the bf16 forward.  Then the two new GEMM
forms against their plain equivalents (out_fp8 epilogue vs bf16 output + host-side quantisation; a_scale_const vs a filled a_scale)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audio_intelligence_amd import ops, _lib as L
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
enc = bench.build_encoder(dev, torch.bfloat16, None)
g = torch.Generator(device=dev).manual_seed(0)
mel = (torch.randn((32, 3000, 128), generator=g, device=dev) * 0.5).to(torch.bfloat16)
mel2 = (torch.randn((32, 3000, 128), generator=g, device=dev) * 0.5).to(torch.bfloat16)     # a DIFFERENT batch than the calibration one

def timed(label):
    for _ in range(2): enc.encode_btc(mel2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = enc.encode_btc(mel2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label}: {ms:.2f} ms per forward", flush=True)
    return out.float(), ms

ref, ms_bf16 = timed("bf16")
enc.enable_fp8(True)
dyn, ms_dyn = timed("e4m3 out + fc1, dynamic row scales (round-2 form)")
sc = enc.calibrate_fp8(mel, attention_output=False)
print("fc2 input scales per layer (margin 2):", [round(float(x), 4) for x in sc[:4]], "...", [round(float(x), 4) for x in sc[-2:]])
sta, ms_sta = timed("e4m3 out + fc1 + fc2, fc2 input quantised in fc1's epilogue (static scale)")
enc.calibrate_fp8(mel, attention_output=True)
print("attention output scales per layer:", [round(float(x), 4) for x in enc._att_out_scale[:4]], "...")
sta2, ms_sta2 = timed("... + attention output written as e4m3 by the attention kernel (static scale)")
for name, o, ms in (("dynamic", dyn, ms_dyn), ("static fc2", sta, ms_sta), ("static fc2 + attention output", sta2, ms_sta2)):
    d = (o - ref).abs()
    print(f"  {name}: {ms_bf16 / ms:.3f}x bf16 | vs bf16 output max abs {float(d.max()):.4f} mean abs {float(d.mean()):.5f} (output std {float(ref.std()):.3f})")
