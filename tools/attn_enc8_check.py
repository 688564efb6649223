"""The two encoder attention forms (attention_enc8.hip: 8 waves x 32 queries; attention_enc.hip: 4 waves x 64 queries) must agree BIT FOR BIT:
same MFMA sequences per accumulator, same partial-sum order, same lag rule.  Full clips, ragged key lengths, packed rows; then us per call."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops
torch.manual_seed(0)
def run(fn, on):
    os.environ["AFHIP_ATTN_ENC8"] = "1" if on else "0"   # the 8-wave form is opt-in
    out = fn()
    torch.cuda.synchronize()
    return out
bad = 0
for B, T, lens in ((2, 1500, None), (3, 1500, [1500, 777, 64]), (4, 700, [700, 1, 65, 300]), (32, 1500, None)):
    qkv = (torch.randn(B, T, 3 * 1280, device="cuda") * 0.7).to(torch.bfloat16)
    qkv[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e)) * 3.0).to(torch.bfloat16)
    kl = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens else None
    a = run(lambda: ops.attention_packed(qkv, 20, key_len=kl, q_prescaled=True), True)
    b = run(lambda: ops.attention_packed(qkv, 20, key_len=kl, q_prescaled=True), False)
    if lens:
        for i, n in enumerate(lens):
            a[i, n:] = 0; b[i, n:] = 0
    same = torch.equal(a, b)
    print(f"B={B} T={T} lens={lens}: bit-identical {same}  max abs diff {float((a.float() - b.float()).abs().max()):.3e}  finite {bool(torch.isfinite(a.float()).all())}", flush=True)
    bad += 0 if same else 1
    if lens:
        rows = torch.cat([qkv[i, :n] for i, n in enumerate(lens)], 0).contiguous()
        ln = torch.tensor(lens, dtype=torch.int32)
        a = run(lambda: ops.attention_ragged(rows, 20, ln, max(lens), q_prescaled=True), True)
        b = run(lambda: ops.attention_ragged(rows, 20, ln, max(lens), q_prescaled=True), False)
        same = torch.equal(a, b)
        print(f"   packed rows {tuple(rows.shape)}: bit-identical {same}", flush=True)
        bad += 0 if same else 1
qkv = (torch.randn(32, 1500, 3 * 1280, device="cuda") * 0.5).to(torch.bfloat16)
qkv[:, :, :1280] = (qkv[:, :, :1280].float() * (0.125 * math.log2(math.e))).to(torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(2):
    for on in (True, False):
        os.environ["AFHIP_ATTN_ENC8"] = "1" if on else "0"   # the 8-wave form is opt-in
        for _ in range(5): ops.attention_packed(qkv, 20, q_prescaled=True)
        ts = []
        for r in range(5):
            e0.record()
            for _ in range(20): ops.attention_packed(qkv, 20, q_prescaled=True)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        print(f"{'8 waves x 32' if on else '4 waves x 64'}: min {min(ts) * 1e3:.1f} us  median {sorted(ts)[2] * 1e3:.1f} us  = {32 * 20 * 4 * 1500 * 1500 * 64 / min(ts) / 1e9:.0f} TFLOP/s", flush=True)
sys.exit(1 if bad else 0)
