#!/usr/bin/env python3
"""Decode-step probe for profiling: AF3-7B shape, B clips, ctx ~800, N greedy steps (bf16 then W8A16), no encoder.
usage: python tools/decode_probe.py [B] [steps] [ctx]     (wrap in rocprofv3 --kernel-trace --stats ... -- python3 tools/decode_probe.py)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PROBE_LIB"):            # same-box A/B of two builds of the library
    from audio_intelligence_amd import _lib as _L
    _L.load_library(os.environ["PROBE_LIB"])
import bench  # noqa: E402
from audio_intelligence_amd.multimodal_io.modeling_whisper import AFWhisperEncoder, AFWhisperEncoderConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 790
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = dict(bench.ENC_CFG)
cfg["encoder_layers"] = 1
enc = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(cfg)).to(dev, torch.bfloat16)
model, n_vocab = bench.build_llm_7b(dev, enc)
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.randn((B, ctx, 3584), generator=g, device=dev) * 0.5).to(torch.bfloat16)
hid, cache = model._forward_hidden(x, model.new_cache(B, ctx + 4 * steps + 64))
tok = model.text_token.expand(B, -1, -1).clone()
variants = [v for v in os.environ.get("PROBE_VARIANTS", "").split(",") if v] or [""]
for rnd in range(int(os.environ.get("PROBE_ROUNDS", "1"))):
  for var in variants:
    for kv in var.split("+"):
        if "=" in kv:
            os.environ[kv.split("=")[0]] = kv.split("=")[1]
    print("variant", var, "round", rnd)
    for fp8 in ((True,) if os.environ.get("PROBE_FP8_ONLY") else ((False, True) if not os.environ.get("PROBE_BF16_ONLY") else (False,))):
      model.enable_fp8_decode(fp8)
      cache.length = ctx
      hyp, _, cache = model._greedy_device_loop(tok, cache, "text", 4, poll=10 ** 9)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      hyp, _, cache = model._greedy_device_loop(hyp[:, -1:, :], cache, "text", steps, poll=10 ** 9)
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      print(f"B={B} ctx~{cache.length} {'fp8' if fp8 else 'bf16'}: {dt / steps * 1e3:.3f} ms/step, {B * steps / dt:.0f} tok/s", flush=True)
