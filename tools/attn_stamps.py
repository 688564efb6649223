"""Diagnostic build of the attention kernel with s_memtime stamps (never the product library): compiles attention.hip with
-DAFHIP_ATTN_STAMPS into /tmp, links it with the other objects, loads THAT library, and prints where waves 0 and 3 of workgroup 0
spend the cycles of key tiles 8..11 -- plain form and the prescaled (lagged-maximum) form.  Read the SHARES, not the length."""
import glob, math, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
obj = "/tmp/attention_stamps.o"
lib = "/tmp/libafhip_stamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-DAFHIP_ATTN_STAMPS",
                "-c", os.path.join(CSRC, "attention.hip"), "-o", obj], check=True)
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("attention.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
from audio_intelligence_amd import _lib as L
L.load_library(lib)
from audio_intelligence_amd import ops
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda") * 0.5).to(torch.bfloat16)
qkv_pre = qkv.clone()
qkv_pre[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
names = ["start", "QK done", "softmax done", "PV done", "stored", "after bar"]
for label, x, pre in (("plain", qkv, False), ("prescaled/LAG", qkv_pre, True)):
    buf = torch.zeros(64, dtype=torch.int64, device="cuda")
    os.environ.pop("AFHIP_ATTN_DBGPTR", None)
    for _ in range(3): ops.attention_packed(x, 20, q_prescaled=pre)
    os.environ["AFHIP_ATTN_DBGPTR"] = hex(buf.data_ptr())
    ops.attention_packed(x, 20, q_prescaled=pre)
    torch.cuda.synchronize()
    t = buf.cpu().reshape(2, 4, 8)
    print(label)
    for g in range(2):
        for j in range(4):
            base = int(t[g, j, 0])
            print(f"  wave {'0' if g == 0 else '3'} tile {8 + j}: " + "  ".join(f"{names[k]}=+{int(t[g, j, k]) - base}" for k in range(1, 6))
                  + (f"  | next tile starts +{int(t[g, j + 1, 0]) - base}" if j < 3 else ""))
