"""Diagnostic build of the decode step's GEMM phases (decode_phases.hip + img_phase.h) with 100-MHz wall-clock stamps of wave 0 of every
workgroup -- never the product library.  For the launches of one layer (q|k|v, o, gate/up, down; the stamps share one clock, so the gaps
BETWEEN the launches show too): when a launch starts, when its weight window has been issued, first / last K step consumed, end of its K loop.
usage: python tools/phase_stamps.py [B] [layer]"""
import glob
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DAFHIP_STREAM_STAMPS"]
objs = []
for name in ("gemm_stream", "gemm_skinny", "decode_phases"):   # every unit that sees SkinnyP / ImgDesc
    obj = f"/tmp/{name}_stamps.o"
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(CSRC, name + ".hip"), "-o", obj], check=True)
    objs.append(obj)
lib = "/tmp/libafhip_phasestamps.so"
skip = ("gemm_stream.o", "gemm_skinny.o", "decode_phases.o")
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if os.path.basename(o) not in skip]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + others, check=True)
from audio_intelligence_amd import _lib as L  # noqa: E402
L.load_library(lib)
import bench  # noqa: E402
from audio_intelligence_amd.multimodal_io.modeling_whisper import AFWhisperEncoder, AFWhisperEncoderConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layer = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
cfg = dict(bench.ENC_CFG)
cfg["encoder_layers"] = 1
enc = AFWhisperEncoder(AFWhisperEncoderConfig.from_dict(cfg)).to(dev, torch.bfloat16)
model, n_vocab = bench.build_llm_7b(dev, enc)
g = torch.Generator(device=dev).manual_seed(1)
ctx = 790
x = (torch.randn((B, ctx, 3584), generator=g, device=dev) * 0.5).to(torch.bfloat16)
hid, cache = model._forward_hidden(x, model.new_cache(B, ctx + 200))
tok = model.text_token.expand(B, -1, -1).clone()
buf = torch.zeros(5 * 256 * 8, dtype=torch.int64, device=dev)
os.environ["AFHIP_STREAM_DBGPTR"] = hex(buf.data_ptr())
os.environ["AFHIP_PHASE_STAMP_LAYER"] = str(layer)
cache.length = ctx
hyp, _, cache = model._greedy_device_loop(tok, cache, "text", 40, poll=10 ** 9)     # graph replays: the stamps of the last step survive
torch.cuda.synchronize()
t = buf.cpu().reshape(5, 256, 8).double()
base = float(t[0][:, 0][t[0][:, 0] > 0].min())
names = ["launch begin", "window issued", "K loop about to start", "first step consumed", "last step consumed", "K loop end"]
order = [0, 1, 2, 3, 4, 5]
for pi, pname in enumerate(["o", "gate/up", "down", "q|k|v", "head"]):
    tt = t[pi]
    if float(tt.max()) == 0:
        continue
    print(f"{pname}:")
    for k in order:
        col = tt[:, k]
        col = col[col > 0]
        if len(col) == 0:
            continue
        col = ((col - base) / 100.0).sort().values
        print(f"    {names[k]:20s} min {float(col[0]):7.2f}  median {float(col[len(col) // 2]):7.2f}  max {float(col[-1]):7.2f}   us after the launch's first stamp")
