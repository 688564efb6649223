import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
qkv = torch.randn(32, 1500, 3 * 1280, device="cuda", dtype=torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for dbg in sys.argv[1:]:
    os.environ["AFHIP_ATTN_DBG"] = dbg
    best = 1e9
    for r in range(3):
        ops.attention_packed(qkv, 20); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): ops.attention_packed(qkv, 20)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"dbg={dbg}: {best:.3f} ms", flush=True)
