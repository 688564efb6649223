"""Diagnostic build of the encoder attention kernel (attention_enc.hip) with s_memtime stamps -- never the product library.
Prints, for waves 0 and 3 of workgroup 0, third block, key tiles 8..15: cycles spent waiting for the DMA of tile t+1, at the
workgroup barrier, in slot 1 (S'_b + O_b MFMAs beside softmax a and the fragment reads) and slot 2.  Read the SHARES."""
import glob, math, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
obj, lib = "/tmp/attention_enc_stamps.o", "/tmp/libafhip_encstamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-inline-asm", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-DAFHIP_ENC_STAMPS"] + sys.argv[1:] +
               ["-c", os.path.join(CSRC, "attention_enc.hip"), "-o", obj], check=True)
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("attention_enc.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
from audio_intelligence_amd import _lib as L
L.load_library(lib)
from audio_intelligence_amd import ops
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda") * 0.5).to(torch.bfloat16)
qkv[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
for _ in range(3): ops.attention_packed(qkv, 20, q_prescaled=True)
buf = torch.zeros(128, dtype=torch.int64, device="cuda")
os.environ["AFHIP_ENC_DBGPTR"] = hex(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.attention_packed(qkv, 20, q_prescaled=True); e1.record()
torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) * 1e3:.0f} us (stamped build)")
t = buf.cpu().reshape(2, 64).tolist()
for g in range(2):
    s = t[g]
    for j in range(8):
        b = s[5 * j: 5 * j + 5]
        nxt = s[5 * j + 5] if j < 7 else None
        print(f"  wave {'0' if g == 0 else '3'} tile {8 + j}: dma wait {b[1] - b[0]:>5} | barrier {b[2] - b[1]:>5} | slot 1 {b[3] - b[2]:>5} | slot 2 {b[4] - b[3]:>5}"
              + (f" | tile total {nxt - b[0]:>5}" if nxt else ""))
