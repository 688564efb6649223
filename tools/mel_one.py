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
