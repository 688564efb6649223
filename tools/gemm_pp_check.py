#!/usr/bin/env python3
"""Correctness + A/B timing of the ping-pong GEMM (gemm_pp.hip) against the 256x256 LDS-DMA kernel and hipBLASLt.
usage: python tools/gemm_pp_check.py [check|bench|all]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops, _lib as L

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = "cuda:0"
dt = torch.bfloat16
torch.manual_seed(0)


def run(a, w, bias, act, r, out, pp):
    os.environ["AFHIP_GEMM_PP"] = "1" if pp else "0"
    ops.gemm(a, w, bias=bias, act=act, residual=r, out=out)


if mode in ("check", "all"):
    cases = [(1024, 256, 128, L.ACT_NONE, False, False), (1024, 256, 256, L.ACT_NONE, True, False),
             (2048, 512, 1280, L.ACT_GELU, True, False), (3000, 1280, 1280, L.ACT_NONE, True, True),
             (48000, 1280, 5120, L.ACT_NONE, True, True), (48000, 5120, 1280, L.ACT_GELU, True, False),
             (6000, 3840, 1280, L.ACT_NONE, True, False), (1500 * 7, 1280, 1280, L.ACT_NONE, False, True)]
    for (m, n, k, act, hb, hr) in cases:
        a = torch.randn(m, k, device=dev, dtype=dt)
        w = torch.randn(n, k, device=dev, dtype=dt) * 0.05
        bias = torch.randn(n, device=dev, dtype=dt) if hb else None
        r = torch.randn(m, n, device=dev, dtype=dt) if hr else None
        out = torch.full((m, n), 7.0, device=dev, dtype=dt)
        for rep in range(3):
            run(a, w, bias, act, r, out, True)
        torch.cuda.synchronize()
        ref = a.float() @ w.float().t()
        if hb:
            ref += bias.float()
        if act == L.ACT_GELU:
            ref = torch.nn.functional.gelu(ref)
        if hr:
            ref += r.float()
        err = (out.float() - ref).abs()
        tol = 0.02 + 0.01 * ref.abs()
        bad = int((err > tol).sum())
        out2 = torch.empty_like(out)
        run(a, w, bias, act, r, out2, False)
        torch.cuda.synchronize()
        d_old = float((out.float() - out2.float()).abs().max())
        print(f"check M={m} N={n} K={k} act={act} bias={hb} res={hr}: max err {float(err.max()):.4f} bad {bad} | vs old kernel max diff {d_old:.4f}", flush=True)
        assert bad == 0, "ping-pong GEMM mismatch"

if mode in ("bench", "all"):
    M = 1500 * 32
    shapes = [("qkv", M, 3840, 1280, L.ACT_NONE, False), ("out+res", M, 1280, 1280, L.ACT_NONE, True),
              ("fc1+gelu", M, 5120, 1280, L.ACT_GELU, False), ("fc1 plain", M, 5120, 1280, L.ACT_NONE, False),
              ("fc2+res", M, 1280, 5120, L.ACT_NONE, True), ("square 4096", 4096, 4096, 4096, L.ACT_NONE, False),
              ("square 8192", 8192, 8192, 8192, L.ACT_NONE, False)]
    for name, m, n, k, act, res in shapes:
        a = torch.randn(m, k, device=dev, dtype=dt)
        w = torch.randn(n, k, device=dev, dtype=dt) * 0.03
        bias = torch.randn(n, device=dev, dtype=dt)
        r = torch.randn(m, n, device=dev, dtype=dt) if res else None
        out = torch.empty(m, n, device=dev, dtype=dt)
        wt = w.t()
        fns = {"pp": lambda: run(a, w, bias, act, r, out, True), "old": lambda: run(a, w, bias, act, r, out, False),
               "lib": lambda: torch.matmul(a, wt, out=out)}
        best = {k_: 1e9 for k_ in fns}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rnd in range(4):                     # interleaved rounds in one process
            for key, fn in fns.items():
                fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best[key] = min(best[key], e0.elapsed_time(e1) / 8)
        fl = 2.0 * m * n * k / 1e9
        print(f"{name:12s} M={m:6d} N={n:5d} K={k:5d}  pp {best['pp']:7.3f} ms {fl/best['pp']:7.1f} TF | old {best['old']:7.3f} ms {fl/best['old']:7.1f} TF | hipBLASLt plain {best['lib']:7.3f} ms {fl/best['lib']:7.1f} TF", flush=True)
