#!/usr/bin/env python3
"""A few launches of the log-mel kernel on B clips (for rocprofv3 runs): python tools/mel_one.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd.multimodal_io.feature_extraction import WhisperFeatureExtractorHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fe = WhisperFeatureExtractorHIP()
wav = torch.randn(B, 480000, device="cuda") * 0.1
for _ in range(5):
    fe.extract_device(wav, layout="btc", dtype=torch.bfloat16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ws = None
e0.record()
for _ in range(20):
    fe.extract_device(wav, layout="btc", dtype=torch.bfloat16)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"log-mel B={B}: {ms*1e3:.1f} us  {B*2688000/ms/1e6:.1f} GB/s on the bytes moved (f32 wav in + bf16 mel out)  ({B*30/ms*1e3:.0f} audio-s/s)")
