"""Diagnostic build of the lean decode attention kernel (attention.hip: attn_decode128_kernel) with 100-MHz wall-clock stamps of wave 0 of every
workgroup -- never the product library.  7B shape: B sequences x 4 kv heads x 7 query heads, context ctx, 128-key ranges.
usage: python tools/decode_attn_stamps.py [B] [ctx]"""
import glob, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "audio-intelligence_amd", "csrc")
obj, lib = "/tmp/attention_dstamps.o", "/tmp/libafhip_dstamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DAFHIP_ATTN_STAMPS", "-c", os.path.join(CSRC, "attention.hip"), "-o", obj], check=True)
others = [o for o in glob.glob(os.path.join(CSRC, "*.o")) if not o.endswith("attention.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, obj] + others, check=True)
from audio_intelligence_amd import _lib as L
L.load_library(lib)
from audio_intelligence_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 858
nq, nkv, hd, cap = 28, 4, 128, 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
qw = (nq + 2 * nkv) * hd
q = (torch.randn(B, qw, device=dev, generator=g)).to(torch.bfloat16)
caches = [((torch.randn(B, nkv, cap, hd, device=dev, generator=g)).to(torch.bfloat16), (torch.randn(B, nkv, cap, hd, device=dev, generator=g)).to(torch.bfloat16)) for _ in range(40)]
big = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
n_wg = ((ctx + 127) // 128) * B * nkv
buf = torch.zeros(n_wg * 8, dtype=torch.int64, device=dev)
for i in range(3): ops.attention_decode(q, caches[i][0], caches[i][1], nq, nkv, ctx, key_split=128, ld_q=qw)
big.zero_()                                                  # the K / V of the stamped launch come from HBM, not from a cache
torch.cuda.synchronize()
os.environ["AFHIP_ATTN_DBGPTR"] = hex(buf.data_ptr())
ops.attention_decode(q, caches[20][0], caches[20][1], nq, nkv, ctx, key_split=128, ld_q=qw)
torch.cuda.synchronize()
t = buf.cpu().reshape(n_wg, 8).double()
t = t[t[:, 0] > 0]
base = float(t[:, 0].min())
names = ["kernel start", "context length known", "all loads issued, q rotated", "S ready (K, q landed)", "softmax done, P written", "P.V done, partials in LDS", "barrier passed", "range partial stored"]
print(f"B={B} ctx={ctx}: {len(t)} workgroups")
for k in range(8):
    col = ((t[:, k] - base) / 100.0).sort().values
    print(f"    {names[k]:32s} min {float(col[0]):6.2f}  median {float(col[len(col) // 2]):6.2f}  max {float(col[-1]):6.2f}  us after the first workgroup's start")
