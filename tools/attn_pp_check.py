#!/usr/bin/env python3
"""Correctness + A/B timing of the ping-pong attention (attention_pp.hip) against the 4-wave flash kernel (attention.hip)
and an fp32 softmax reference.  usage: python tools/attn_pp_check.py [check|bench|all]"""
import os
import sys
import math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = "cuda:0"
torch.manual_seed(0)


def run(qkv, H, kl, pp):
    os.environ["AFHIP_ATTN_PP"] = "1" if pp else "0"
    return ops.attention_packed(qkv, H, key_len=kl)


def ref(qkv, H, kl):
    B, T, D3 = qkv.shape
    d = D3 // 3
    hd = d // H
    q, k, v = [x.float().reshape(B, T, H, hd).permute(0, 2, 1, 3) for x in qkv.split(d, dim=2)]
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if kl is not None:
        mask = torch.arange(T, device=qkv.device)[None, :] >= kl[:, None].long()
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, d)


if mode in ("check", "all"):
    for (B, T, H, lens, spike) in [(2, 1500, 20, None, False), (3, 1500, 6, [1500, 700, 65], False), (2, 333, 4, [333, 64], False),
                                   (2, 1500, 8, [1500, 1], True), (1, 256, 2, None, True)]:
        d = H * 64
        qkv = torch.randn(B, T, 3 * d, device=dev, dtype=torch.bfloat16)
        if spike:   # force the lagged-max rescale branch: one key far above the rest, late in the sequence
            qkv[:, T // 2, d:2 * d] *= 12.0
            qkv[:, : , :d] *= 2.0
        kl = torch.tensor(lens, device=dev, dtype=torch.int32) if lens else None
        o_pp = run(qkv, H, kl, True)
        o_old = run(qkv, H, kl, False)
        torch.cuda.synchronize()
        r = ref(qkv, H, kl)
        e_pp = (o_pp.float() - r).abs().max().item()
        e_old = (o_old.float() - r).abs().max().item()
        print(f"check B={B} T={T} H={H} lens={lens} spike={spike}: max err pp {e_pp:.4f} old {e_old:.4f} | mean err pp {(o_pp.float()-r).abs().mean().item():.5f} old {(o_old.float()-r).abs().mean().item():.5f}", flush=True)
        assert (e_pp < 0.03 or spike) and not torch.isnan(o_pp.float()).any(), "ping-pong attention mismatch"

if mode in ("bench", "all"):
    B, T, H = 32, 1500, 20
    qkv = torch.randn(B, T, 3 * H * 64, device=dev, dtype=torch.bfloat16)
    fl = 4.0 * T * T * 64 * H * B / 1e9
    best = {True: 1e9, False: 1e9}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rnd in range(4):
        for pp in (True, False):
            run(qkv, H, None, pp)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                run(qkv, H, None, pp)
            e1.record()
            torch.cuda.synchronize()
            best[pp] = min(best[pp], e0.elapsed_time(e1) / 5)
    print(f"encoder attention B=32 T=1500 H=20 hd=64: pp {best[True]:.3f} ms {fl/best[True]:.1f} TF | old {best[False]:.3f} ms {fl/best[False]:.1f} TF", flush=True)
