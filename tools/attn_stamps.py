import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
qkv = torch.randn(32, 1500, 3 * 1280, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
for _ in range(3): ops.attention_packed(qkv, 20)
os.environ["AFHIP_ATTN_DBGPTR"] = hex(buf.data_ptr())
ops.attention_packed(qkv, 20)
torch.cuda.synchronize()
t = buf.cpu().reshape(2, 4, 8)
t0 = int(t[0, 0, 0])
names = ["start", "QK done", "softmax done", "PV done", "stored", "after bar"]
for g in range(2):
    for j in range(4):
        print(f"group {g} tile {8+j}: " + "  ".join(f"{names[k]}={int(t[g,j,k])-t0}" for k in range(6)))
