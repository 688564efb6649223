#!/usr/bin/env python3
"""fp8-operand GEMM probe: correctness vs torch on dequantised operands + timing vs the bf16 kernel on the same shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_intelligence_amd import ops, _lib as L
from audio_intelligence_amd.utils.quant import quantize_rows_e4m3
dev = "cuda:0"
shapes = [("enc qkv", 48000, 3840, 1280, L.ACT_NONE), ("enc fc1", 48000, 5120, 1280, L.ACT_GELU), ("enc fc2", 48000, 1280, 5120, L.ACT_NONE),
          ("llm qkv", 6320, 4608, 3584, L.ACT_NONE), ("llm gu", 6320, 37888, 3584, L.ACT_SWIGLU), ("llm down", 6320, 3584, 18944, L.ACT_NONE)]
for name, M, N, K, act in shapes:
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.03
    aq, sa = quantize_rows_e4m3(a)
    wq, sw = quantize_rows_e4m3(w)
    ad = (aq.view(torch.float8_e4m3fn).float() * sa[:, None])
    wd = (wq.view(torch.float8_e4m3fn).float() * sw[:, None])
    out = ops.gemm_fp8(aq, sa, wq, sw, act=act)
    rows = torch.arange(0, M, max(1, M // 64), device=dev)
    if act == L.ACT_SWIGLU:
        I = N // 2
        # interleaved rows: block j of 64 rows = 32 gate then 32 up of outputs [32 j, 32 j + 32)
        wg = wd.view(I // 32, 2, 32, K)[:, 0].reshape(I, K)
        wu = wd.view(I // 32, 2, 32, K)[:, 1].reshape(I, K)
        ref = torch.nn.functional.silu(ad[rows] @ wg.T) * (ad[rows] @ wu.T)
    else:
        ref = ad[rows] @ wd.T
        if act == L.ACT_GELU:
            ref = torch.nn.functional.gelu(ref)
    err = (out[rows].float() - ref).abs()
    lim = 3e-2 + 2e-2 * ref.abs()
    a16, w16 = a.to(torch.bfloat16), w.to(torch.bfloat16)
    o16 = torch.empty((M, N // 2 if act == L.ACT_SWIGLU else N), dtype=torch.bfloat16, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def t(fn):
        for _ in range(2): fn()
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5
    t8 = t(lambda: ops.gemm_fp8(aq, sa, wq, sw, act=act, out=out))
    t16 = t(lambda: ops.gemm(a16, w16, act=act, out=o16))
    tq = t(lambda: ops.quant_rows(a16))
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M} N={N} K={K}: bad {int((err > lim).sum())}/{err.numel()} max err {float(err.max()):.4f} | fp8 {t8:.3f} ms {fl/t8/1e9:.0f} TF | bf16 {t16:.3f} ms {fl/t16/1e9:.0f} TF | quant_rows {tq:.3f} ms", flush=True)
