"""Diagnostic build of the encoder attention kernel (attention_enc.hip) with s_memtime stamps -- never the product library.
Prints, for waves 0 and 3 of workgroup 0, third block, key tiles 8..15: cycles spent waiting for the DMA of tile t+1, at the
workgroup barrier, in slot 1 (S'_b + O_b MFMAs beside softmax a and the fragment reads) and slot 2.  Read the SHARES."""
import glob, math, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
obj, lib = "/tmp/attention_enc_stamps.o", "/tmp/libafhip_encstamps.so"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-inline-asm", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-fno-slp-vectorize",
         "-DAFHIP_ENC_STAMPS"] + sys.argv[1:]      # the Makefile's flags for this file + the stamp macro
SRC = os.path.join(CSRC, "attention_enc.hip")
# the same guard as the Makefile rule: the accumulator file is asm-owned, a build whose compiler-side code names an AGPR must not run
asm_out = "/tmp/attention_enc_stamps.s"
subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-S", SRC, "-o", asm_out], check=True, stderr=subprocess.DEVNULL)
import re
inside = False
for line in open(asm_out):
    if "#ASMSTART" in line: inside = True
    elif "#ASMEND" in line: inside = False
    elif not inside and (("v_accvgpr" in line) or re.search(r"[ ,\[]a\[?[0-9]+", line)):
        sys.exit("stamp build refused: hipcc parked a value in the asm-owned accumulator file: " + line.strip())
subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", SRC, "-o", obj], check=True)
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("attention_enc.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
from audio_intelligence_amd import _lib as L
L.load_library(lib)
from audio_intelligence_amd import ops
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda") * 0.5).to(torch.bfloat16)
qkv[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
for _ in range(3): ops.attention_packed(qkv, 20, q_prescaled=True)
buf = torch.zeros(128 + 256, dtype=torch.int64, device="cuda")
os.environ["AFHIP_ENC_DBGPTR"] = hex(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.attention_packed(qkv, 20, q_prescaled=True); e1.record()
torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1) * 1e3:.0f} us (stamped build)")
fin = buf[128:].cpu().tolist()
t = buf[:128].cpu().reshape(2, 64).tolist()
for g in range(2):
    s = t[g]
    for j in range(8):
        b = s[5 * j: 5 * j + 5]
        nxt = s[5 * j + 5] if j < 7 else None
        print(f"  wave {'0' if g == 0 else '3'} tile {8 + j}: dma wait {b[1] - b[0]:>5} | barrier {b[2] - b[1]:>5} | slot 1 {b[3] - b[2]:>5} | slot 2 {b[4] - b[3]:>5}"
              + (f" | tile total {nxt - b[0]:>5}" if nxt else ""))

dur = [x for x in fin if x > 0]
if dur:
    srt = sorted(dur)
    lo, hi, med = srt[0], srt[-1], srt[len(srt) // 2]
    print(f"duration of the {len(dur)} workgroups of this launch (s_memtime ticks, each on its own XCD's counter): min {lo}  median {med}  p90 {srt[int(len(srt) * 0.9)]}  max {hi}"
          f"  -> the launch lasts as long as its slowest workgroup: max / mean = {hi / (sum(dur) / len(dur)):.3f}")
    per_xcd = {}
    for i, x in enumerate(fin):
        if x > 0: per_xcd.setdefault(i & 7, []).append(x)
    print("per XCD (blockIdx & 7) mean duration / overall mean:", {k: round(sum(v) / len(v) / (sum(dur) / len(dur)), 3) for k, v in sorted(per_xcd.items())})
