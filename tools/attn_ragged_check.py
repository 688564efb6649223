import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
torch.manual_seed(0)
H, hd = 20, 64
d = H * hd
lens = torch.tensor([1500, 333, 1000, 64, 1201, 500], dtype=torch.int32)
B, T = len(lens), 1500
qkv = (torch.randn(B, T, 3 * d, device="cuda") * 0.5).to(torch.bfloat16)
qkv[:, :, :d] = (qkv[:, :, :d].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
kl = lens.cuda()
a1 = ops.attention_packed(qkv, H, key_len=kl, q_prescaled=True)
a2 = ops.attention_packed(qkv, H, key_len=kl, q_prescaled=True)
print("padded deterministic:", torch.equal(a1, a2))
rows = torch.cat([qkv[b, : int(lens[b])] for b in range(B)], dim=0).contiguous()
r1 = ops.attention_ragged(rows, H, lens, T, q_prescaled=True)
r2 = ops.attention_ragged(rows, H, lens, T, q_prescaled=True)
print("ragged deterministic:", torch.equal(r1, r2))
off = 0
for b in range(B):
    n = int(lens[b])
    x, y = r1[off: off + n], a1[b, :n]
    neq = (x != y)
    if neq.any():
        rws = neq.any(dim=1).nonzero().flatten()
        cols = neq.any(dim=0).nonzero().flatten()
        print(f"clip {b} len {n}: {int(neq.sum())} elements differ, rows {rws[:8].tolist()}..{rws[-3:].tolist()} ({len(rws)} rows), heads {sorted(set((cols // 64).tolist()))[:10]}, max diff {float((x.float() - y.float()).abs().max()):.4g}")
    else:
        print(f"clip {b} len {n}: identical")
    off += n
